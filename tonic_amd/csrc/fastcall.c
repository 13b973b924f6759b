/* tonic_amd._fastcall — the per-environment-step entry points of libtonic_hip.so behind the CPython vectorcall
 * convention instead of ctypes.
 *
 * Why: the collect loop (tonic/utils/trainer.py:44-56: agent.step -> environment.step -> agent.update) is a latency
 * chain of ~10 us per environment step in which two foreign calls sit on the critical path — the simulator's record
 * (tonic_collector_synthetic_step, which issues the GPU's next command) and the wait for the actions
 * (tonic_collector_wait_actions) — and three more run in the GPU's shadow (arm, claim, ppo_step).  A ctypes call
 * costs 0.35 - 0.6 us of argument conversion per call on this host; a METH_FASTCALL function ~0.1 us.
 *
 * This module holds NO logic: every function converts its arguments and calls the C ABI entry of the same name
 * (include/tonic_hip.h), resolved with dlsym from the library tonic_amd._lib has already loaded and version-checked.
 * Without this module the package binds the same entries through ctypes (TONIC_AMD_FASTCALL=0 forces that).
 *
 * Build: make -C tonic_amd/csrc fast   (gcc + Python.h; no HIP) */
#define PY_SSIZE_T_CLEAN
#include <Python.h>
#include <dlfcn.h>
#include <stdint.h>

#include "../../include/tonic_hip.h"      /* TONIC_ABI_VERSION: the signatures below are this version's */

typedef int (*synthetic_step_fn)(void*, const float*, const float*, int32_t);
typedef int (*wait_actions_fn)(void*, double);
typedef int (*arm_fn)(void*, int64_t, int32_t, int32_t);
typedef int (*claim_fn)(void*);
typedef int (*ring_fn)(void*);
typedef int (*q_act_fn)(void*, const float*, void*, int32_t, int32_t, int32_t, int32_t, float*, const void*, void*,
                        int64_t, void*);

static synthetic_step_fn p_synthetic_step;
static wait_actions_fn p_wait_actions;
static arm_fn p_arm, p_ppo_step;
static claim_fn p_claim;
static ring_fn p_ring;
static q_act_fn p_q_act;

/* int or None -> pointer (None: NULL); -1 + exception on anything else */
static int as_pointer(PyObject* o, void** out) {
  if (o == Py_None) { *out = NULL; return 0; }
  *out = PyLong_AsVoidPtr(o);
  return (*out == NULL && PyErr_Occurred()) ? -1 : 0;
}

static PyObject* bind(PyObject* self, PyObject* arg) {
  const char* path = PyUnicode_AsUTF8(arg);
  if (path == NULL) return NULL;
  void* lib = dlopen(path, RTLD_NOW | RTLD_NOLOAD);        /* the copy ctypes loaded: nothing is loaded here */
  if (lib == NULL) {
    PyErr_Format(PyExc_RuntimeError, "%s is not loaded: tonic_amd._lib.load() comes first", path);
    return NULL;
  }
  /* a shim built against another ABI would call the entries with the wrong arguments after a successful dlsym */
  int32_t (*abi)(void) = (int32_t (*)(void))dlsym(lib, "tonic_abi_version");
  if (abi == NULL || abi() != TONIC_ABI_VERSION) {
    PyErr_Format(PyExc_RuntimeError, "%s has ABI %d, tonic_amd._fastcall was built for %d: make -C tonic_amd/csrc fast",
                 path, abi ? (int)abi() : -1, TONIC_ABI_VERSION);
    return NULL;
  }
  p_synthetic_step = (synthetic_step_fn)dlsym(lib, "tonic_collector_synthetic_step");
  p_wait_actions = (wait_actions_fn)dlsym(lib, "tonic_collector_wait_actions");
  p_arm = (arm_fn)dlsym(lib, "tonic_collector_arm");
  p_ppo_step = (arm_fn)dlsym(lib, "tonic_collector_ppo_step");
  p_claim = (claim_fn)dlsym(lib, "tonic_collector_claim");
  p_ring = (ring_fn)dlsym(lib, "tonic_collector_ring");
  p_q_act = (q_act_fn)dlsym(lib, "tonic_collector_q_act");
  if (!p_synthetic_step || !p_wait_actions || !p_arm || !p_ppo_step || !p_claim || !p_ring || !p_q_act) {
    PyErr_Format(PyExc_RuntimeError, "%s lacks a tonic_collector_* entry", path);
    return NULL;
  }
  Py_RETURN_NONE;
}

#define REQUIRE_ARGS(n, name)                                                          \
  if (nargs != (n)) {                                                                  \
    PyErr_Format(PyExc_TypeError, name " takes %d arguments (%zd given)", (n), nargs); \
    return NULL;                                                                       \
  }

/* tonic_collector_synthetic_step(block, next_observations, actions | None, ring) */
static PyObject* synthetic_step(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
  REQUIRE_ARGS(4, "tonic_collector_synthetic_step")
  void *block, *next_observations, *actions;
  if (as_pointer(args[0], &block) || as_pointer(args[1], &next_observations) || as_pointer(args[2], &actions))
    return NULL;
  const int ring = PyObject_IsTrue(args[3]);
  if (ring < 0) return NULL;
  return PyLong_FromLong(p_synthetic_step(block, (const float*)next_observations, (const float*)actions, ring));
}

/* tonic_collector_wait_actions(collector, timeout_s): spins on the completion words — without the GIL, the noise
 * helper thread of the agent draws the next steps' noise meanwhile */
static PyObject* wait_actions(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
  REQUIRE_ARGS(2, "tonic_collector_wait_actions")
  void* collector;
  if (as_pointer(args[0], &collector)) return NULL;
  const double timeout = PyFloat_AsDouble(args[1]);
  if (timeout == -1.0 && PyErr_Occurred()) return NULL;
  int status;
  Py_BEGIN_ALLOW_THREADS
  status = p_wait_actions(collector, timeout);
  Py_END_ALLOW_THREADS
  return PyLong_FromLong(status);
}

/* tonic_collector_arm / tonic_collector_ppo_step(collector, row, eps_slot, store_previous) */
static PyObject* row_call(arm_fn fn, const char* name, PyObject* const* args, Py_ssize_t nargs) {
  if (nargs != 4) {
    PyErr_Format(PyExc_TypeError, "%s takes 4 arguments (%zd given)", name, nargs);
    return NULL;
  }
  void* collector;
  if (as_pointer(args[0], &collector)) return NULL;
  const long long row = PyLong_AsLongLong(args[1]);
  if (row == -1 && PyErr_Occurred()) return NULL;
  const long slot = PyLong_AsLong(args[2]);
  if (slot == -1 && PyErr_Occurred()) return NULL;
  const int store = PyObject_IsTrue(args[3]);
  if (store < 0) return NULL;
  return PyLong_FromLong(fn(collector, (int64_t)row, (int32_t)slot, (int32_t)store));
}
static PyObject* arm(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
  return row_call(p_arm, "tonic_collector_arm", args, nargs);
}
static PyObject* ppo_step(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
  return row_call(p_ppo_step, "tonic_collector_ppo_step", args, nargs);
}

static PyObject* claim(PyObject* self, PyObject* arg) {
  void* collector;
  if (as_pointer(arg, &collector)) return NULL;
  return PyLong_FromLong(p_claim(collector));
}

static PyObject* ring(PyObject* self, PyObject* arg) {
  void* block;
  if (as_pointer(arg, &block)) return NULL;
  return PyLong_FromLong(p_ring(block));
}

/* tonic_collector_q_act(collector, actor_params, actor_images, rebuild_images, kind, H, eps_slot, rows_out,
 *                        store | None (address of a tonic_q_store_t), workspace, workspace_bytes, stream) */
static PyObject* q_act(PyObject* self, PyObject* const* args, Py_ssize_t nargs) {
  REQUIRE_ARGS(12, "tonic_collector_q_act")
  void *collector, *params, *images, *rows_out, *store, *workspace, *stream;
  if (as_pointer(args[0], &collector) || as_pointer(args[1], &params) || as_pointer(args[2], &images) ||
      as_pointer(args[7], &rows_out) || as_pointer(args[8], &store) || as_pointer(args[9], &workspace) ||
      as_pointer(args[11], &stream))
    return NULL;
  long small[4];
  for (int i = 0; i < 4; ++i) {
    small[i] = PyLong_AsLong(args[3 + i]);
    if (small[i] == -1 && PyErr_Occurred()) return NULL;
  }
  const long long bytes = PyLong_AsLongLong(args[10]);
  if (bytes == -1 && PyErr_Occurred()) return NULL;
  return PyLong_FromLong(p_q_act(collector, (const float*)params, images, (int32_t)small[0], (int32_t)small[1],
                                 (int32_t)small[2], (int32_t)small[3], (float*)rows_out, store, workspace,
                                 (int64_t)bytes, stream));
}

static PyMethodDef methods[] = {
    {"bind", bind, METH_O, "bind(path of the loaded libtonic_hip.so): resolves the entries below"},
    {"tonic_collector_synthetic_step", (PyCFunction)(void (*)(void))synthetic_step, METH_FASTCALL, NULL},
    {"tonic_collector_wait_actions", (PyCFunction)(void (*)(void))wait_actions, METH_FASTCALL, NULL},
    {"tonic_collector_arm", (PyCFunction)(void (*)(void))arm, METH_FASTCALL, NULL},
    {"tonic_collector_ppo_step", (PyCFunction)(void (*)(void))ppo_step, METH_FASTCALL, NULL},
    {"tonic_collector_claim", claim, METH_O, NULL},
    {"tonic_collector_ring", ring, METH_O, NULL},
    {"tonic_collector_q_act", (PyCFunction)(void (*)(void))q_act, METH_FASTCALL, NULL},
    {NULL, NULL, 0, NULL}};

static struct PyModuleDef module = {PyModuleDef_HEAD_INIT, "_fastcall",
                                    "vectorcall bindings of the per-step tonic_collector_* entries", -1, methods};

PyMODINIT_FUNC PyInit__fastcall(void) { return PyModule_Create(&module); }
