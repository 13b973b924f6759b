// One-shot all-reduce of small float32 buffers between the GPUs of one node.
//
// replaces (new; nothing to mirror in the reference, which is single-process): the per-iteration
//   gradient exchange of the sharded learner.  RCCL's ring / tree all-reduce of a 44 KB - 711 KB
//   buffer is latency-bound: 2 (G - 1) hops over point-to-point xGMI links.  Here every rank
//   WRITES its contribution straight into a window of every peer (7 links used concurrently, one
//   hop), raises a flag there, waits for the 7 flags raised in its own window and adds the G
//   contributions it now holds locally IN RANK ORDER — every rank computes the same bits, and the
//   result does not depend on arrival order.
//
// Windows are plain device allocations exported with hipIpcGetMemHandle; the caller moves the
// handles between the processes (any bootstrap channel: the Python side uses torch.distributed's
// object gather once).  Two slot sets alternate with the call parity, so a rank that runs ahead
// never writes into a slot its peer is still reading.  No cache-wide fences on the hot path:
// contributions are pushed with plain stores + ONE system-scope release per workgroup, flags are
// system-scope atomics, and the slots are read back with system-scope loads (their lines may still
// sit in this XCD's L2 from two calls ago).
#include <stdlib.h>
#include <string.h>

#include <new>

#include "common.h"

namespace tonic {
namespace {

constexpr int kMaxRanks = 8;
constexpr int kCommBlocks = 16;            // workgroups of one all-reduce (each owns a slice)
constexpr int kCommThreads = 256;

struct WindowLayout {
  int64_t slot_floats;                     // capacity of one contribution, multiple of 4
  int64_t slots_bytes, flags_offset, status_offset, total_bytes;
  WindowLayout(int64_t max_floats, int world) {
    slot_floats = round_up(max_floats, 64);
    slots_bytes = 2 * (int64_t)world * slot_floats * 4;                   // [parity][source rank]
    flags_offset = round_up(slots_bytes, 256);
    status_offset = flags_offset + round_up(2 * (int64_t)world * kCommBlocks * 4, 256);
    total_bytes = round_up(status_offset + 256, 4096);
  }
};

struct ReduceArgs {
  char* window[kMaxRanks];                 // [rank] = that rank's window as mapped HERE
  float* buffer;
  int64_t n, slot_floats, flags_offset, status_offset;
  int rank, world;
  unsigned sequence;                       // > 0, increases by one per call
  unsigned long long timeout_ticks;        // 100 MHz wall clock
};

__device__ __forceinline__ float* slot_of(const ReduceArgs& a, int owner, int parity, int source) {
  return reinterpret_cast<float*>(a.window[owner]) +
         ((int64_t)parity * a.world + source) * a.slot_floats;
}
__device__ __forceinline__ unsigned* flag_of(const ReduceArgs& a, int owner, int parity, int source,
                                             int block) {
  return reinterpret_cast<unsigned*>(a.window[owner] + a.flags_offset) +
         ((int64_t)parity * a.world + source) * kCommBlocks + block;
}

__global__ __launch_bounds__(kCommThreads) void allreduce_oneshot_kernel(ReduceArgs a) {
  const int parity = (int)(a.sequence & 1u);
  const int tid = threadIdx.x, b = blockIdx.x;
  // this workgroup's slice, in 16-byte vectors
  const int64_t vecs = (a.n + 3) >> 2, per = (vecs + kCommBlocks - 1) / kCommBlocks;
  const int64_t v0 = min(vecs, (int64_t)b * per), v1 = min(vecs, v0 + per);
  // 1. push: my slice into slot [parity][my rank] of EVERY rank's window (my own included)
  for (int64_t v = v0 + tid; v < v1; v += kCommThreads) {
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * v + e < a.n) x[e] = a.buffer[4 * v + e];
    for (int p = 0; p < a.world; ++p)
      reinterpret_cast<f32x4*>(slot_of(a, p, parity, a.rank))[v] = x;
  }
  // 2. one system-scope release per wave that stored, then the flags: "block b of rank r is in"
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
  __syncthreads();
  if (tid < a.world)
    __hip_atomic_store(flag_of(a, tid, parity, a.rank, b), a.sequence, __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_SYSTEM);
  // 3. wait for the same slice of every rank in MY window
  __shared__ int failed;
  if (tid == 0) failed = 0;
  __syncthreads();
  if (tid < a.world) {
    const unsigned* flag = flag_of(a, a.rank, parity, tid, b);
    // (a communicator that has missed a peer once does not wait again: every later call returns at
    //  once with the status standing, so a caller that checks late has lost one timeout, not many)
    const unsigned* status = reinterpret_cast<const unsigned*>(a.window[a.rank] + a.status_offset);
    const bool broken = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
    const unsigned long long t0 = wall_clock64();
    while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != a.sequence) {
      if (broken || wall_clock64() - t0 > a.timeout_ticks) { failed = 1; break; }
      __builtin_amdgcn_s_sleep(2);
    }
  }
  __syncthreads();
  if (failed) {                            // a peer never arrived: say so instead of hanging
    if (tid == 0)
      __hip_atomic_store(reinterpret_cast<unsigned*>(a.window[a.rank] + a.status_offset),
                         a.sequence, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  // 4. the G contributions, added in rank order (system-scope loads: never a stale L2 line)
  for (int64_t v = v0 + tid; v < v1; v += kCommThreads) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < a.world; ++p) {
      __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(
          slot_of(a, a.rank, parity, p), 0, 0x7fffffff, 0x27000);
      const f32x4 x = __builtin_bit_cast(
          f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)(v * 16), 0, 17));
      if (p == 0) acc = x;
      else {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[e] = acc[e] + x[e];
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (4 * v + e < a.n) a.buffer[4 * v + e] = acc[e];
  }
}

}  // namespace
}  // namespace tonic

using namespace tonic;

struct tonic_comm {
  int rank, world;
  int64_t max_floats;
  char* window;                  // my window
  char* mapped[kMaxRanks];       // every rank's window as seen from this process
  bool opened[kMaxRanks];
  bool connected;
  unsigned sequence;
  unsigned long long timeout_ticks;   // 0: the process-wide default (comm_timeout_ticks)
};

#define TONIC_HIP(call, what)                                                       \
  do {                                                                              \
    hipError_t e__ = (call);                                                        \
    if (e__ != hipSuccess) {                                                        \
      set_error("%s: %s", (what), hipGetErrorString(e__));                          \
      return TONIC_ERR_LAUNCH;                                                      \
    }                                                                               \
  } while (0)

extern "C" int64_t tonic_comm_handle_bytes(void) { return (int64_t)sizeof(hipIpcMemHandle_t); }

extern "C" int tonic_comm_init(tonic_comm_t** out, int32_t rank, int32_t world, int64_t max_floats) {
  TONIC_REQUIRE(out != nullptr && world >= 1 && world <= kMaxRanks && rank >= 0 && rank < world &&
                    max_floats > 0,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_comm_init: rank %d of %d, %lld floats", rank,
                world, (long long)max_floats);
  *out = nullptr;
  tonic_comm* c = new (std::nothrow) tonic_comm();
  TONIC_REQUIRE(c != nullptr, TONIC_ERR_INVALID_ARGUMENT, "tonic_comm_init: out of memory");
  memset(c, 0, sizeof(*c));
  c->rank = rank; c->world = world; c->max_floats = max_floats;
  const WindowLayout l(max_floats, world);
  hipError_t e = hipMalloc(reinterpret_cast<void**>(&c->window), (size_t)l.total_bytes);
  if (e == hipSuccess) e = hipMemset(c->window, 0, (size_t)l.total_bytes);
  if (e != hipSuccess) {
    set_error("tonic_comm_init: window of %lld bytes: %s", (long long)l.total_bytes,
              hipGetErrorString(e));
    tonic_comm_destroy(c);
    return TONIC_ERR_LAUNCH;
  }
  c->mapped[rank] = c->window;
  c->connected = world == 1;
  *out = c;
  return TONIC_OK;
}

extern "C" int tonic_comm_export(tonic_comm_t* c, void* handle_out) {
  TONIC_REQUIRE(c && handle_out, TONIC_ERR_INVALID_ARGUMENT, "tonic_comm_export: bad argument");
  hipIpcMemHandle_t handle;
  TONIC_HIP(hipIpcGetMemHandle(&handle, c->window), "hipIpcGetMemHandle");
  memcpy(handle_out, &handle, sizeof(handle));
  return TONIC_OK;
}

extern "C" int tonic_comm_connect(tonic_comm_t* c, const void* all_handles) {
  TONIC_REQUIRE(c && all_handles, TONIC_ERR_INVALID_ARGUMENT, "tonic_comm_connect: bad argument");
  const char* bytes = static_cast<const char*>(all_handles);
  for (int r = 0; r < c->world; ++r) {
    if (r == c->rank || c->opened[r]) continue;
    hipIpcMemHandle_t handle;
    memcpy(&handle, bytes + (size_t)r * sizeof(handle), sizeof(handle));
    void* peer = nullptr;
    TONIC_HIP(hipIpcOpenMemHandle(&peer, handle, hipIpcMemLazyEnablePeerAccess),
              "hipIpcOpenMemHandle");
    c->mapped[r] = static_cast<char*>(peer);
    c->opened[r] = true;
  }
  c->connected = true;
  return TONIC_OK;
}

// How long a rank waits for its peers' flags before it gives up with an error status (never a
// hang): TONIC_AMD_ALLREDUCE_TIMEOUT_S seconds, default 120 — the first exchange of an update has
// to absorb whatever skew the ranks have (module load, a checkpoint save on rank 0, a slow
// simulator), and a rank that gives up leaves the sums unreduced, which the learner turns into an
// exception at its next read-back (tonic_comm_status).
static unsigned long long comm_timeout_ticks() {
  static const unsigned long long ticks = [] {
    double seconds = 120.0;
    if (const char* env = getenv("TONIC_AMD_ALLREDUCE_TIMEOUT_S")) {
      const double v = atof(env);
      if (v > 0.0) seconds = v;
    }
    return (unsigned long long)(seconds * 1e8);          // 100 MHz wall clock
  }();
  return ticks;
}

extern "C" int tonic_allreduce_f32(tonic_comm_t* c, float* d_buffer, int64_t n, void* stream) {
  TONIC_REQUIRE(c && d_buffer && n > 0, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_allreduce_f32: bad argument");
  TONIC_REQUIRE(c->connected, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_allreduce_f32: tonic_comm_connect first");
  TONIC_REQUIRE(n <= c->max_floats, TONIC_ERR_WORKSPACE,
                "tonic_allreduce_f32: %lld floats, the windows hold %lld", (long long)n,
                (long long)c->max_floats);
  TONIC_REQUIRE((reinterpret_cast<uintptr_t>(d_buffer) & 3) == 0, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_allreduce_f32: unaligned buffer");
  if (c->world == 1) return TONIC_OK;
  const WindowLayout l(c->max_floats, c->world);
  ReduceArgs a{};
  for (int r = 0; r < c->world; ++r) a.window[r] = c->mapped[r];
  a.buffer = d_buffer; a.n = n; a.slot_floats = l.slot_floats; a.flags_offset = l.flags_offset;
  a.status_offset = l.status_offset;
  a.rank = c->rank; a.world = c->world;
  a.sequence = ++c->sequence;
  a.timeout_ticks = c->timeout_ticks != 0 ? c->timeout_ticks : comm_timeout_ticks();
  hipLaunchKernelGGL(allreduce_oneshot_kernel, dim3(kCommBlocks), dim3(kCommThreads), 0,
                     as_stream(stream), a);
  TONIC_CHECK_LAUNCH("tonic_allreduce_f32");
  return TONIC_OK;
}

extern "C" int tonic_comm_set_timeout(tonic_comm_t* c, double seconds) {
  TONIC_REQUIRE(c != nullptr && seconds >= 0.0 && seconds <= 3600.0, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_comm_set_timeout: bad argument");
  c->timeout_ticks = (unsigned long long)(seconds * 1e8);   // 100 MHz wall clock; 0 = the default
  return TONIC_OK;
}

extern "C" int tonic_comm_can_access_peer(int32_t device, int32_t peer) {
  int count = 0;
  TONIC_HIP(hipGetDeviceCount(&count), "hipGetDeviceCount");
  TONIC_REQUIRE(device >= 0 && device < count && peer >= 0 && peer < count,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_comm_can_access_peer: devices %d, %d of %d",
                device, peer, count);
  if (device == peer) return 1;
  int can = 0;
  TONIC_HIP(hipDeviceCanAccessPeer(&can, device, peer), "hipDeviceCanAccessPeer");
  return can != 0 ? 1 : 0;
}

extern "C" int tonic_comm_status(tonic_comm_t* c) {
  TONIC_REQUIRE(c != nullptr, TONIC_ERR_INVALID_ARGUMENT, "tonic_comm_status: bad argument");
  const WindowLayout l(c->max_floats, c->world);
  unsigned failed_at = 0;
  TONIC_HIP(hipMemcpy(&failed_at, c->window + l.status_offset, 4, hipMemcpyDeviceToHost),
            "hipMemcpy");
  if (failed_at != 0) {
    set_error("tonic_allreduce_f32: a peer did not arrive within %.0f s (call number %u)",
              (double)(c->timeout_ticks != 0 ? c->timeout_ticks : comm_timeout_ticks()) / 1e8,
              failed_at);
    return TONIC_ERR_TIMEOUT;
  }
  return TONIC_OK;
}

extern "C" int tonic_comm_destroy(tonic_comm_t* c) {
  if (c == nullptr) return TONIC_OK;
  for (int r = 0; r < kMaxRanks; ++r)
    if (c->opened[r]) (void)hipIpcCloseMemHandle(c->mapped[r]);
  if (c->window) (void)hipFree(c->window);
  delete c;
  return TONIC_OK;
}
