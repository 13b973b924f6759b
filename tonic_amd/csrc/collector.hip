// Pinned-host batched collector (include/tonic_hip.h, "collector" section).
//
// replaces: the process boundary of tonic/environments/distributed.py:82-95,136-155 (one pickled
//   Pipe message per worker group and step, one shared Queue back) and the host <-> device hops of
//   tonic/torch/agents/a2c.py:41-73 (torch.as_tensor / .numpy() per step).
//
// One shared float32 BLOCK carries a whole environment step of all W workers:
//   [header | eps0 | observations | next_observations | rewards | resets | terminations | eps1 |
//    actions | resets_u8 | terminations_u8]
// The environment side (parent + forked workers, no HIP) synchronises on two futex words in the
// header: the parent bumps `go_seq` once per step for ALL worker groups, each group adds one to
// `done_count`, the last one wakes the parent — two system calls per step on the parent whatever
// the number of groups, against one Pipe.send per group + one Queue.get per group.
// The agent side page-locks the same block (hipHostRegister), so the workers' memory IS the DMA
// source: transport 0 lets the fused act kernel read observations / noise / the previous step's
// outcome straight from the block over PCIe and write the actions (and a completion word the host
// spins on) straight back; transport 1 moves the same bytes with hipMemcpyAsync on the
// collector's own stream around the kernel and waits on an event — no stream synchronisation
// either way.  transport 2 keeps that kernel resident for a rollout (a command word instead of a
// launch); transport 3 is transport 2 with the step's command, observations and noise PUSHED into a
// window of device memory by the host (write-combined stores through the GPU's BAR) instead of being
// pulled over PCIe by the kernel: the poll and the actor tiles' input loads stay inside the GPU.
#include <ctype.h>
#include <errno.h>
#include <immintrin.h>
#include <limits.h>
#include <linux/futex.h>
#include <stdlib.h>
#include <string.h>
#include <sys/syscall.h>
#include <time.h>
#include <unistd.h>

#include <pthread.h>

#include <new>

#include "collect16.h"
#include "collector_q.h"

namespace tonic {
namespace {

constexpr uint32_t kBlockMagic = 0x544f4e43u;        // "TONC"
constexpr int64_t kHeaderBytes = 4096;
constexpr int kFields = TONIC_COLLECTOR_FIELD_COUNT;
// transport 2, device memory zeroed at every launch of the resident kernel: the park notice (a
// 256-byte line of its own) and one claim word per workgroup slot (<= 4096 tiles + copy + record)
constexpr size_t kRelayBytes = 256 + 4 * (4096 + 8);

struct BlockHeader {
  uint32_t magic, version;
  int64_t W;
  int32_t O, A, groups;
  int32_t carry_over;          // the environment's promise (tonic_collector_block_carry_over)
  int64_t total_bytes;
  int64_t offset[kFields];                         // bytes from the start of the block
  alignas(64) uint32_t go_seq;                     // futex: bumped once per environment step
  alignas(64) uint32_t done_count;                 // futex: groups that finished the step
  alignas(64) uint32_t shutdown;
  alignas(64) uint32_t act_seq;                    // sequence number of the last act command
  alignas(64) uint64_t command;                    // transport 2: the host's next command word
  alignas(64) uint32_t parked;                     // transport 2: written by the resident kernel
  // transport 2: the NEXT command, prepared by the agent (tonic_collector_arm) and issued by
  // whoever completes the step record it acts on (tonic_collector_ring: the environment's step
  // call, or the last worker group in tonic_collector_worker_done); 0 = nothing armed
  alignas(64) uint64_t armed;
  // transport 3: the device window of the collector that owns this block, as the HOST address the owning
  // process stores through (0: none), and that process — forked workers share the header, not the mapping
  alignas(64) uint64_t push_window;
  int32_t push_pid;
};
static_assert(sizeof(BlockHeader) <= kHeaderBytes, "header does not fit its page");

int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

void layout(int64_t W, int O, int A, int64_t* offset, int64_t* total) {
  const int64_t sizes[kFields] = {
      W * A * 4,          // EPS0
      W * O * 4,          // OBSERVATIONS
      W * O * 4,          // NEXT_OBSERVATIONS
      W * 4, W * 4, W * 4,// REWARDS, RESETS, TERMINATIONS
      W * A * 4,          // EPS1
      W * A * 4,          // ACTIONS
      W, W,               // RESETS_U8, TERMINATIONS_U8
      (int64_t)collect16_blocks(W) * 4};   // DONE_FLAGS (one word per workgroup of the act launch)
  int64_t at = kHeaderBytes;
  for (int f = 0; f < kFields; ++f) {
    offset[f] = at;
    at = align_up(at + sizes[f], 256);
  }
  *total = align_up(at, 4096);
}

BlockHeader* header_of(void* block) {
  BlockHeader* h = static_cast<BlockHeader*>(block);
  return (h != nullptr && h->magic == kBlockMagic) ? h : nullptr;
}

long futex(uint32_t* word, int op, uint32_t value, const timespec* timeout) {
  return syscall(SYS_futex, word, op, value, timeout, nullptr, 0);
}

double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + ts.tv_nsec * 1e-9;
}

inline void cpu_relax() { __builtin_ia32_pause(); }

// Waits until *word != seen (returns true) or the deadline passes: a short spin first (the
// common case when everybody is fast), then FUTEX_WAIT slices.
bool wait_change(uint32_t* word, uint32_t seen, double timeout_s, int spin_iterations) {
  for (int i = 0; i < spin_iterations; ++i) {
    if (__atomic_load_n(word, __ATOMIC_ACQUIRE) != seen) return true;
    cpu_relax();
  }
  const double deadline = now_s() + timeout_s;
  for (;;) {
    if (__atomic_load_n(word, __ATOMIC_ACQUIRE) != seen) return true;
    const double left = deadline - now_s();
    if (left <= 0) return false;
    const double slice = left < 0.5 ? left : 0.5;
    timespec ts{(time_t)slice, (long)((slice - (time_t)slice) * 1e9)};
    futex(word, FUTEX_WAIT, seen, &ts);            // EAGAIN / EINTR / ETIMEDOUT: re-check
  }
}

// This process (cached; a forked child asks again).
pid_t g_pid = 0;
void forget_pid() { g_pid = 0; }
pid_t my_pid() {
  if (g_pid == 0) {
    static bool hooked = false;
    if (!hooked) { pthread_atfork(nullptr, nullptr, forget_pid); hooked = true; }
    g_pid = getpid();
  }
  return g_pid;
}

// The window of a transport-3 collector if THIS process may store through it.
char* push_window_of(const BlockHeader* h) {
  const uint64_t window = __atomic_load_n(&h->push_window, __ATOMIC_ACQUIRE);
  return window != 0 && h->push_pid == (int32_t)my_pid() ? reinterpret_cast<char*>(window) : nullptr;
}

// Block field -> the same field of the window: write-combined stores, nothing is ever read back.
void push_field(const BlockHeader* h, char* window, int f, int64_t bytes) {
  memcpy(window + h->offset[f], reinterpret_cast<const char*>(h) + h->offset[f], (size_t)bytes);
}

// The next command to the resident kernel.  Pull (transport 2): one release store to the block's header,
// the kernel finds it by polling over PCIe.  Push (transport 3, in the owning process): the step's
// observation rows go into the window first, a store fence keeps the write-combined stores in order, then
// the command word follows them into the window.  (A stop command, bit 3, acts on nothing; rows_pushed: the
// caller has put the observation rows into the window already.)
// `window`: where the pushed copy goes — the ISSUING collector's own window for the owner-side calls
// (issue_command_of: null for a pull collector, whose resident kernel polls the header — a second collector
// on a block whose first one pushes must not write its commands, or its stop word, into the first one's
// window: ADVICE r5), the header's window for the block-level ring of the environment's step.
void issue_command(BlockHeader* h, uint64_t word, bool rows_pushed, char* window) {
  __atomic_store_n(&h->command, word, __ATOMIC_RELEASE);
  if (window == nullptr) return;
  if ((word & 8u) == 0 && !rows_pushed)
    push_field(h, window, TONIC_COLLECTOR_OBSERVATIONS, h->W * h->O * 4);
  _mm_sfence();
  *reinterpret_cast<volatile uint64_t*>(window + offsetof(BlockHeader, command)) = word;
  _mm_sfence();
}

// Issues the command the agent has armed, if any (host stores only: callable from a forked worker —
// which leaves the command of a PUSH collector where it is: the owning process issues it, from its own
// ring or when the agent claims it back).
// The exchange makes ring and claim / cancel mutually exclusive: exactly one side gets the word.
int ring_armed(BlockHeader* h, bool rows_pushed = false) {
  if (__atomic_load_n(&h->armed, __ATOMIC_RELAXED) == 0) return 0;
  if (__atomic_load_n(&h->push_window, __ATOMIC_RELAXED) != 0 && push_window_of(h) == nullptr) return 0;
  const uint64_t word = __atomic_exchange_n(&h->armed, (uint64_t)0, __ATOMIC_ACQ_REL);
  if (word == 0) return 0;
  __atomic_store_n(&h->act_seq, (uint32_t)(word >> 32), __ATOMIC_RELEASE);
  issue_command(h, word, rows_pushed, push_window_of(h));
  return 1;
}

}  // namespace
}  // namespace tonic

using namespace tonic;

// ---------------------------------------------------------------- block: layout + worker sync

extern "C" int64_t tonic_collector_block_bytes(int64_t W, int32_t O, int32_t A) {
  if (W <= 0 || O <= 0 || A <= 0) return -1;
  int64_t offset[kFields], total;
  layout(W, O, A, offset, &total);
  return total;
}

extern "C" int tonic_collector_block_init(void* block, int64_t bytes, int64_t W, int32_t O,
                                          int32_t A, int32_t groups) {
  TONIC_REQUIRE(block != nullptr && W > 0 && O > 0 && A > 0 && groups > 0,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_collector_block_init: bad argument");
  TONIC_REQUIRE((reinterpret_cast<uintptr_t>(block) & 4095) == 0, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_collector_block_init: the block must be page aligned");
  BlockHeader h;
  memset(&h, 0, sizeof(h));
  layout(W, O, A, h.offset, &h.total_bytes);
  TONIC_REQUIRE(bytes >= h.total_bytes, TONIC_ERR_WORKSPACE,
                "tonic_collector_block_init: %lld bytes given, %lld needed", (long long)bytes,
                (long long)h.total_bytes);
  h.magic = kBlockMagic;
  h.version = 1;
  h.W = W; h.O = O; h.A = A; h.groups = groups;
  memset(block, 0, (size_t)h.total_bytes);
  memcpy(block, &h, sizeof(h));
  return TONIC_OK;
}

extern "C" int64_t tonic_collector_block_offset(const void* block, int32_t field) {
  const BlockHeader* h = header_of(const_cast<void*>(block));
  if (h == nullptr || field < 0 || field >= kFields) return -1;
  return h->offset[field];
}

// A vectorised simulator's step record in one call (tonic_amd.environments.SyntheticBatch: the
// zero-cost benchmark environment of SURVEY §8d, obs from a pre-generated pool, reward = -|a|^2):
// next_observations -> the NEXT_OBSERVATIONS and OBSERVATIONS fields, rewards[w] = -sum_a a[w][a]^2
// in float32, left to right.  actions == NULL: the block's ACTIONS field, where the act kernel
// wrote them.  Host code only (no HIP call): five NumPy calls per step otherwise.
extern "C" int tonic_collector_synthetic_step(void* block, const float* next_observations,
                                              const float* actions, int32_t ring) {
  BlockHeader* h = header_of(block);
  TONIC_REQUIRE(h != nullptr && next_observations != nullptr, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_collector_synthetic_step: bad argument");
  char* base = static_cast<char*>(block);
  const size_t bytes = (size_t)h->W * h->O * sizeof(float);
  memcpy(base + h->offset[TONIC_COLLECTOR_NEXT_OBSERVATIONS], next_observations, bytes);
  // A push collector's kernel reads the observation rows from its window: when this call issues the command
  // they go THERE first, the command follows, and the block's own copy (what the trainer is handed) is made
  // while the GPU is already at work.
  char* window = ring && __atomic_load_n(&h->armed, __ATOMIC_RELAXED) != 0 ? push_window_of(h) : nullptr;
  if (window == nullptr) memcpy(base + h->offset[TONIC_COLLECTOR_OBSERVATIONS], next_observations, bytes);
  const float* a = actions != nullptr ? actions
                                      : reinterpret_cast<const float*>(base + h->offset[TONIC_COLLECTOR_ACTIONS]);
  float* rewards = reinterpret_cast<float*>(base + h->offset[TONIC_COLLECTOR_REWARDS]);
  const int A = h->A;
  for (int64_t w = 0; w < h->W; ++w) {
    float sum = 0.f;
    for (int k = 0; k < A; ++k) sum += a[w * A + k] * a[w * A + k];
    rewards[w] = -sum;
  }
  // ring: the caller knows that the flags in the block are final too (nobody resets at this
  // step) -> the record is complete, the agent's armed command goes out from here
  if (window != nullptr) memcpy(window + h->offset[TONIC_COLLECTOR_OBSERVATIONS], next_observations, bytes);
  if (ring) ring_armed(h, window != nullptr);
  if (window != nullptr) memcpy(base + h->offset[TONIC_COLLECTOR_OBSERVATIONS], next_observations, bytes);
  return TONIC_OK;
}

extern "C" int tonic_collector_block_carry_over(void* block, int32_t promised) {
  BlockHeader* h = header_of(block);
  TONIC_REQUIRE(h != nullptr, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_collector_block_carry_over: bad block");
  __atomic_store_n(&h->carry_over, promised ? 1 : 0, __ATOMIC_RELEASE);
  return TONIC_OK;
}

extern "C" int tonic_collector_ring(void* block) {
  BlockHeader* h = header_of(block);
  TONIC_REQUIRE(h != nullptr, TONIC_ERR_INVALID_ARGUMENT, "tonic_collector_ring: bad block");
  return ring_armed(h);
}

extern "C" int64_t tonic_collector_worker_wait(void* block, int64_t seen, double timeout_s) {
  BlockHeader* h = header_of(block);
  if (h == nullptr) return -3;
  // Workers sleep: an environment step is far longer than a futex wake-up, and W Python
  // processes spinning would starve each other.
  const double deadline = now_s() + timeout_s;
  for (;;) {
    if (__atomic_load_n(&h->shutdown, __ATOMIC_ACQUIRE)) return -1;
    const uint32_t now = __atomic_load_n(&h->go_seq, __ATOMIC_ACQUIRE);
    if (now != (uint32_t)seen) return (int64_t)now;
    const double left = deadline - now_s();
    if (left <= 0) return -2;
    if (!wait_change(&h->go_seq, (uint32_t)seen, left < 0.5 ? left : 0.5, 200)) continue;
  }
}

extern "C" int tonic_collector_worker_done(void* block) {
  BlockHeader* h = header_of(block);
  TONIC_REQUIRE(h != nullptr, TONIC_ERR_INVALID_ARGUMENT, "tonic_collector_worker_done: bad block");
  const uint32_t before = __atomic_fetch_add(&h->done_count, 1u, __ATOMIC_ACQ_REL);
  if (before + 1 == (uint32_t)h->groups) {
    // the step record is complete: the GPU starts on it before the parent has even woken up
    ring_armed(h);
    futex(&h->done_count, FUTEX_WAKE, 1, nullptr);
  }
  return TONIC_OK;
}

extern "C" int tonic_collector_submit_actions(void* block) {
  BlockHeader* h = header_of(block);
  TONIC_REQUIRE(h != nullptr, TONIC_ERR_INVALID_ARGUMENT, "tonic_collector_submit_actions: bad block");
  __atomic_store_n(&h->done_count, 0u, __ATOMIC_RELEASE);       // every group is idle here
  __atomic_fetch_add(&h->go_seq, 1u, __ATOMIC_ACQ_REL);
  futex(&h->go_seq, FUTEX_WAKE, INT_MAX, nullptr);
  return TONIC_OK;
}

extern "C" int tonic_collector_wait_obs(void* block, double timeout_s) {
  BlockHeader* h = header_of(block);
  TONIC_REQUIRE(h != nullptr, TONIC_ERR_INVALID_ARGUMENT, "tonic_collector_wait_obs: bad block");
  const uint32_t groups = (uint32_t)h->groups;
  const double deadline = now_s() + timeout_s;
  for (;;) {
    const uint32_t count = __atomic_load_n(&h->done_count, __ATOMIC_ACQUIRE);
    if (count >= groups) return TONIC_OK;
    const double left = deadline - now_s();
    if (left <= 0) {
      set_error("tonic_collector_wait_obs: %u of %u worker groups answered within %.1f s", count,
                groups, timeout_s);
      return TONIC_ERR_TIMEOUT;
    }
    wait_change(&h->done_count, count, left, 2000);
  }
}

extern "C" int tonic_collector_shutdown(void* block) {
  BlockHeader* h = header_of(block);
  TONIC_REQUIRE(h != nullptr, TONIC_ERR_INVALID_ARGUMENT, "tonic_collector_shutdown: bad block");
  __atomic_store_n(&h->shutdown, 1u, __ATOMIC_RELEASE);
  __atomic_fetch_add(&h->go_seq, 1u, __ATOMIC_ACQ_REL);
  futex(&h->go_seq, FUTEX_WAKE, INT_MAX, nullptr);
  return TONIC_OK;
}

// ------------------------------------------------------------------------------ GPU side

struct tonic_collector {
  BlockHeader* host;            // the shared block (host address)
  char* mapped;                 // device-visible alias of the block (hipHostGetDevicePointer)
  char* staged;                 // transport 1: device copy of the block's fields
  char* window;                 // transport 3: fine-grained device memory the host stores into (block layout)
  int64_t W;
  int O, A, transport;
  bool counted;                 // among the users of its block (see BlockUse)
  hipStream_t stream;
  hipEvent_t learner_done, collect_done, actions_out;
  float* d_packed;
  float* seg[7];                // observations, actions, next_observations, rewards, resets,
  float* norm_acc;              //   terminations, log_probs
  // MeanStd.record sums per Segment row: hist[r] = the sums before step r, hist[r + 1] after it.
  // A step reads one entry and writes the next, so issuing a row twice (a speculative launch
  // that has to be repeated) accumulates nothing twice.  norm_acc -> hist[first row] when a
  // rollout starts, hist[last row + 1] -> norm_acc when it ends.
  float* d_norm_hist;
  int64_t hist_rows;
  bool hist_loaded;
  int64_t last_row;             // row of the last step issued
  bool wide;                    // shapes beyond the fused act kernel
  const float* actor_params;
  void* d_wide_ws;
  int64_t wide_ws_bytes;
  int64_t rows;
  unsigned seq;
  bool actor_packed, waiting;
  bool armed;                   // tonic_collector_arm left a command in the block's header
  int64_t armed_row;
  // transport 2: the resident collect kernel
  unsigned* d_relay;
  unsigned* d_tile_done;         // [4096] device words: see Collect16Args::tile_done
  bool live;
  double park_us;
  unsigned long long* d_stamps;   // developer probe (TONIC_AMD_COLLECTOR_STAMPS=1)
  unsigned long long launches, relaunches;
  int q_words;                    // > 0: the step in flight is an off-policy acting launch with that many completion words
};

namespace {

#define TONIC_HIP(call, what)                                                       \
  do {                                                                              \
    hipError_t e__ = (call);                                                        \
    if (e__ != hipSuccess) {                                                        \
      set_error("%s: %s", (what), hipGetErrorString(e__));                          \
      return TONIC_ERR_LAUNCH;                                                      \
    }                                                                               \
  } while (0)

// A command of collector `c` itself (step, stop): into ITS window if it pushes, else the header alone.
void issue_command_of(tonic_collector* c, uint64_t word) {
  char* window = c->window != nullptr && c->host->push_pid == (int32_t)my_pid() &&
                         c->host->push_window == (uint64_t)reinterpret_cast<uintptr_t>(c->window)
                     ? c->window : nullptr;
  issue_command(c->host, word, false, window);
}

// Takes the armed command back.  1: the environment has issued it meanwhile — the step is in
// flight and the handle's bookkeeping catches up; 0: it was never issued (or nothing was armed).
int claim_armed(tonic_collector* c) {
  if (!c->armed) return 0;
  c->armed = false;
  if (__atomic_exchange_n(&c->host->armed, (uint64_t)0, __ATOMIC_ACQ_REL) != 0) return 0;
  c->seq += 1;
  c->last_row = c->armed_row;
  c->waiting = true;
  return 1;
}

// Where the kernels read / write field `f`: the mapped block (transport 0 / 2), its device copy
// (transport 1) or, for what the host pushes (transport 3: observations and noise), the window.
float* field(tonic_collector* c, int f) {
  char* base = c->transport != 1 ? c->mapped : c->staged;
  if (c->transport == 3 &&
      (f == TONIC_COLLECTOR_OBSERVATIONS || f == TONIC_COLLECTOR_EPS0 || f == TONIC_COLLECTOR_EPS1))
    base = c->window;
  return reinterpret_cast<float*>(base + c->host->offset[f]);
}

// transport 3: the noise rows of slot `eps_slot` follow the block into the window (off the step's
// critical path when the command is armed ahead: the GPU is still busy with the step before)
void push_noise(tonic_collector* c, int eps_slot) {
  if (c->transport != 3 || eps_slot < 0) return;
  push_field(c->host, c->window, eps_slot == 0 ? TONIC_COLLECTOR_EPS0 : TONIC_COLLECTOR_EPS1,
             c->W * c->A * 4);
}

// The outcome of one environment step -> Segment row (the copy role of the collect kernel on its
// own, for the last step of a rollout).
__global__ void outcome_store_kernel(const float* next_obs, const float* rewards,
                                     const float* resets, const float* terminations,
                                     float* seg_next, float* seg_rew, float* seg_rst,
                                     float* seg_term, int64_t row, int64_t W, int O) {
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = tid; i < W * O; i += stride) seg_next[row * W * O + i] = next_obs[i];
  for (int64_t i = tid; i < W; i += stride) {
    seg_rew[row * W + i] = rewards[i];
    seg_rst[row * W + i] = resets[i];
    seg_term[row * W + i] = terminations[i];
  }
}

// transport 1: the step's inputs as ONE host-to-device copy (the fields are laid out so that
// either noise slot is contiguous with the observation / outcome fields).
int stage_inputs(tonic_collector* c, int eps_slot) {
  const BlockHeader* h = c->host;
  const int first = eps_slot == 0 ? TONIC_COLLECTOR_EPS0 : TONIC_COLLECTOR_OBSERVATIONS;
  const int last = eps_slot == 1 ? TONIC_COLLECTOR_EPS1 : TONIC_COLLECTOR_TERMINATIONS;
  const int64_t begin = h->offset[first], end = h->offset[last + 1];
  TONIC_HIP(hipMemcpyAsync(c->staged + begin, reinterpret_cast<char*>(c->host) + begin,
                           (size_t)(end - begin), hipMemcpyHostToDevice, c->stream),
            "tonic_collector: H2D of the step inputs");
  return TONIC_OK;
}

}  // namespace

extern "C" void* tonic_host_device_pointer(void* pinned_host) {
  void* device = nullptr;
  if (pinned_host == nullptr || hipHostGetDevicePointer(&device, pinned_host, 0) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  return device;
}

namespace {

// Collectors per block in this process (host code under the caller's serialisation, like the handles).  A
// block is page-locked by its first collector and released by its last: ROCm accepts a second registration
// of the same range, and the first hipHostUnregister would then take the mapping away from everybody.
struct BlockUse { const void* block; int users; bool locked_here; };
BlockUse g_blocks[256];
BlockUse* block_use(const void* block, bool make) {
  BlockUse* spare = nullptr;
  for (BlockUse& b : g_blocks) {
    if (b.users > 0 && b.block == block) return &b;
    if (b.users == 0 && spare == nullptr) spare = &b;
  }
  if (!make || spare == nullptr) return nullptr;
  spare->block = block; spare->locked_here = false;
  return spare;
}

// The NUMA node the current HIP device hangs off (-1: unknown / one node), from sysfs.
int device_numa_node() {
  int device = 0;
  char bus[32] = {0};
  if (hipGetDevice(&device) != hipSuccess || hipDeviceGetPCIBusId(bus, sizeof(bus), device) != hipSuccess) {
    (void)hipGetLastError();
    return -1;
  }
  for (char* p = bus; *p; ++p) *p = (char)tolower(*p);
  char path[128];
  snprintf(path, sizeof(path), "/sys/bus/pci/devices/%s/numa_node", bus);
  FILE* f = fopen(path, "r");
  if (f == nullptr) return -1;
  int node = -1;
  if (fscanf(f, "%d", &node) != 1) node = -1;
  fclose(f);
  return node;
}

// The shared block's pages next to the GPU that will poll and read them over PCIe every environment step.
// They sit where the thread that first touched them ran (the trainer's, wherever the scheduler had put it):
// on a two-socket host half of all processes got them on the far socket, and every one of the step's two or
// three PCIe round trips then crosses the socket interconnect as well — 13.0 against 10.7 us per environment
// step of 256 workers (profiles/r05_numa.md).  Best effort (mbind + MPOL_MF_MOVE on the mapping, before it
// is page-locked); TONIC_AMD_NUMA_BIND=0 leaves the pages where they are.
void move_block_near_device(void* block, size_t bytes) {
  const char* off = getenv("TONIC_AMD_NUMA_BIND");
  if (off != nullptr && off[0] == '0') return;
  const int node = device_numa_node();
  if (node < 0 || node >= 1024) return;
  unsigned long mask[16] = {0};
  mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
  constexpr int kMpolBind = 2, kMfMove = 2;           // <linux/mempolicy.h>: MPOL_BIND, MPOL_MF_MOVE
  (void)syscall(SYS_mbind, block, bytes, kMpolBind, mask, 8 * sizeof(mask) + 1, kMfMove);
}

// transport 3 needs the whole of device memory behind a CPU-visible BAR (every MI300 / MI350 server board;
// hipDeviceAttributeIsLargeBar), a block nobody else pushes into, and a step whose rows the host writes
// faster than the kernel pulls them: write-combined stores run at ~29 GB/s from one core, inside the
// environment's step, against ~2 us + bytes / 50 GB/s for the pull (scripts/ubench/pingpong.hip: 6.2 -> 4.9
// us per exchange at 28 KB with an otherwise idle host; the collect loop at W = 256, O = 28: 10.3 -> 9.7 us
// per environment step) — up to 48 KB of observations per step.  (Pushing the command word alone and
// pulling the rows is slower than the plain pull: 10.7 us, profiles/r05_push_transport.md.)
// TONIC_AMD_COLLECTOR_PUSH=0: never.
bool push_possible(const BlockHeader* h) {
  const char* off = getenv("TONIC_AMD_COLLECTOR_PUSH");
  if (off != nullptr && off[0] == '0') return false;
  if (h->push_window != 0) return false;
  if (h->W * h->O * 4 > 48 * 1024) return false;
  int device = 0, large_bar = 0;
  if (hipGetDevice(&device) != hipSuccess ||
      hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, device) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return large_bar == 1;
}

// Does a host store reach the window?  The attribute says the BAR covers device memory; whether THIS process may
// store through it (containers, IOMMU set-ups) is tried once per window WITHOUT touching it from user space: the
// kernel does the store — read() from a pipe into the window returns EFAULT where a user store would fault (no
// signal handlers swapped under a multi-threaded process, no longjmp across threads: ADVICE r5) — and the word
// comes back through hipMemcpy.
bool window_takes_host_stores(char* window) {
  const uint64_t word = 0x746f6e6963707573ull;
  int fd[2];
  if (pipe(fd) != 0) return false;
  bool stored = write(fd[1], &word, sizeof(word)) == (ssize_t)sizeof(word) &&
                read(fd[0], window + 64, sizeof(word)) == (ssize_t)sizeof(word);
  close(fd[0]);
  close(fd[1]);
  _mm_sfence();
  uint64_t back = 0;
  if (!stored || hipMemcpy(&back, window + 64, sizeof(back), hipMemcpyDeviceToHost) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  return back == word;
}

}  // namespace

extern "C" int tonic_collector_create(tonic_collector_t** out, void* block, int32_t transport) {
  TONIC_REQUIRE(out != nullptr, TONIC_ERR_INVALID_ARGUMENT, "tonic_collector_create: out is NULL");
  *out = nullptr;
  BlockHeader* h = header_of(block);
  TONIC_REQUIRE(h != nullptr, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_collector_create: not an initialised collector block");
  TONIC_REQUIRE(transport >= 0 && transport <= 3, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_collector_create: transport must be 0 (mapped, one launch per step), "
                "1 (hipMemcpyAsync), 2 (mapped, resident kernel) or 3 (resident kernel, inputs pushed)");
  TONIC_REQUIRE(h->O <= 384 && h->A <= 32, TONIC_ERR_UNSUPPORTED_SHAPE,
                "tonic_collector_create: the act kernels serve O <= 384, A <= 32 (got %d, %d)",
                h->O, h->A);
  tonic_collector* c = new (std::nothrow) tonic_collector();
  TONIC_REQUIRE(c != nullptr, TONIC_ERR_INVALID_ARGUMENT, "tonic_collector_create: out of memory");
  memset(c, 0, sizeof(*c));
  c->host = h; c->W = h->W; c->O = h->O; c->A = h->A; c->transport = transport;
  // beyond the fused act kernel (O > 32 or A > 8): layer-by-layer launches per step on the mapped
  // block (mlpwide.hip wide_collect_step), i.e. transport 0 whatever was asked for
  c->wide = h->O > 32 || h->A > 8;
  if (c->wide) c->transport = 0;
  if (c->transport == 3 && !push_possible(h)) c->transport = 2;
  auto fail = [&](const char* what, hipError_t e) {
    set_error("tonic_collector_create: %s: %s", what, hipGetErrorString(e));
    tonic_collector_destroy(c);
    return TONIC_ERR_LAUNCH;
  };
  hipError_t e = hipSuccess;
  BlockUse* use = block_use(block, true);
  if (use == nullptr) return fail("more than 256 live collector blocks", hipErrorOutOfMemory);
  if (use->users == 0) {
    move_block_near_device(block, (size_t)h->total_bytes);
    e = hipHostRegister(block, (size_t)h->total_bytes, hipHostRegisterMapped);
    if (e == hipErrorHostMemoryAlreadyRegistered) (void)hipGetLastError();     // (by the caller: theirs to release)
    else if (e != hipSuccess) return fail("hipHostRegister of the shared block", e);
    else use->locked_here = true;
  }
  use->users += 1;
  c->counted = true;
  void* mapped = nullptr;
  if ((e = hipHostGetDevicePointer(&mapped, block, 0)) != hipSuccess)
    return fail("hipHostGetDevicePointer", e);
  c->mapped = static_cast<char*>(mapped);
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess)
    return fail("hipStreamCreate", e);
  if ((e = hipEventCreateWithFlags(&c->learner_done, hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&c->collect_done, hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&c->actions_out, hipEventDisableTiming)) != hipSuccess)
    return fail("hipEventCreate", e);
  const int64_t packed = PackedActor(collect16_ks1(c->O), collect16_ap(c->A)).total;
  if ((e = hipMalloc(reinterpret_cast<void**>(&c->d_packed), packed * 4)) != hipSuccess ||
      (e = hipMalloc(reinterpret_cast<void**>(&c->staged), (size_t)h->total_bytes)) != hipSuccess ||
      (e = hipMalloc(reinterpret_cast<void**>(&c->d_relay), kRelayBytes)) != hipSuccess ||
      (e = hipMalloc(reinterpret_cast<void**>(&c->d_tile_done), 4096 * 4)) != hipSuccess ||
      (e = hipMemset(c->d_tile_done, 0, 4096 * 4)) != hipSuccess)
    return fail("hipMalloc of the collector scratch", e);
  if (c->wide) {
    c->wide_ws_bytes = wide_collect_workspace_bytes(c->W, c->O, c->A);
    if ((e = hipMalloc(&c->d_wide_ws, (size_t)c->wide_ws_bytes)) != hipSuccess)
      return fail("hipMalloc of the wide act workspace", e);
  }
  if (getenv("TONIC_AMD_COLLECTOR_STAMPS") != nullptr) {
    if ((e = hipMalloc(reinterpret_cast<void**>(&c->d_stamps), 32 * 8)) != hipSuccess ||
        (e = hipMemset(c->d_stamps, 0, 32 * 8)) != hipSuccess)
      return fail("hipMalloc of the stamp buffer", e);
  }
  if (c->transport == 3) {
    // the window: fine-grained device memory (the kernel's system-scope loads see the host's stores
    // whatever an L2 holds), zero = no command yet
    if ((e = hipExtMallocWithFlags(reinterpret_cast<void**>(&c->window), (size_t)h->total_bytes,
                                   hipDeviceMallocFinegrained)) != hipSuccess ||
        (e = hipMemsetAsync(c->window, 0, (size_t)h->total_bytes, c->stream)) != hipSuccess ||
        (e = hipStreamSynchronize(c->stream)) != hipSuccess)     // (this stream only: other agents' streams may be held)
      return fail("the device window of transport 3", e);
    if (!window_takes_host_stores(c->window)) {        // (the pull transport serves everybody)
      (void)hipFree(c->window);
      c->window = nullptr;
      c->transport = 2;
    }
  }
  if (c->transport == 3) {
    if ((e = hipMemsetAsync(c->window, 0, 4096, c->stream)) != hipSuccess ||
        (e = hipStreamSynchronize(c->stream)) != hipSuccess)
      return fail("the device window of transport 3", e);
    h->push_pid = (int32_t)my_pid();
    __atomic_store_n(&h->push_window, (uint64_t)reinterpret_cast<uintptr_t>(c->window), __ATOMIC_RELEASE);
  }
  const char* park = getenv("TONIC_AMD_COLLECTOR_PARK_US");
  c->park_us = park != nullptr ? atof(park) : 200.0;
  if ((e = hipMemset(c->staged, 0, (size_t)h->total_bytes)) != hipSuccess)
    return fail("hipMemset", e);
  c->seq = __atomic_load_n(&h->act_seq, __ATOMIC_ACQUIRE);
  *out = c;
  return TONIC_OK;
}

extern "C" int tonic_collector_destroy(tonic_collector_t* c) {
  if (c == nullptr) return TONIC_OK;
  // teardown: nothing useful can be done about a failing release
  if (claim_armed(c)) (void)tonic_collector_wait_actions(c, 1.0);   // (issued by the environment)
  if (c->live) {                     // a stop command (nothing to store) ends the resident kernel
    c->seq += 1;
    issue_command_of(c, ((uint64_t)c->seq << 32) | 8u);
    c->live = false;
  }
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->window != nullptr) {
    if (c->host->push_window == (uint64_t)reinterpret_cast<uintptr_t>(c->window))
      __atomic_store_n(&c->host->push_window, (uint64_t)0, __ATOMIC_RELEASE);
    (void)hipFree(c->window);
  }
  if (c->d_stamps) {
    unsigned long long t[32];
    if (hipMemcpy(t, c->d_stamps, sizeof(t), hipMemcpyDeviceToHost) == hipSuccess) {
      fprintf(stderr, "collector: %llu launches of the resident kernel (%llu after a park), %llu slots run "
              "by another workgroup, %llu commands skipped by a late workgroup, %llu park notices\n",
              c->launches, c->relaunches, t[24], t[25], t[26]);
      const char* role[3] = {"actor tile 0", "record", "copy 0"};
      for (int r = 0; r < 3; ++r) {
        const double n = t[r * 8 + 7] > 0 ? (double)t[r * 8 + 7] : 1.0;
        fprintf(stderr, "collector stamps, %s (us after the command was seen, %llu steps):", role[r],
                t[r * 8 + 7]);
        for (int p = 0; p < 7; ++p) fprintf(stderr, " %.2f", t[r * 8 + p] / n / 100.0);
        fprintf(stderr, "\n");
      }
    }
    (void)hipFree(c->d_stamps);
  }
  if (c->d_norm_hist) (void)hipFree(c->d_norm_hist);
  if (c->d_wide_ws) (void)hipFree(c->d_wide_ws);
  if (c->d_packed) (void)hipFree(c->d_packed);
  if (c->staged) (void)hipFree(c->staged);
  if (c->d_relay) (void)hipFree(c->d_relay);
  if (c->d_tile_done) (void)hipFree(c->d_tile_done);
  if (c->learner_done) (void)hipEventDestroy(c->learner_done);
  if (c->collect_done) (void)hipEventDestroy(c->collect_done);
  if (c->actions_out) (void)hipEventDestroy(c->actions_out);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  BlockUse* use = c->counted ? block_use(c->host, false) : nullptr;
  if (use != nullptr && --use->users == 0 && use->locked_here) {
    if (hipHostUnregister(c->host) != hipSuccess) (void)hipGetLastError();
  }
  delete c;
  return TONIC_OK;
}

extern "C" void* tonic_collector_stream(tonic_collector_t* c) {
  return c != nullptr ? c->stream : nullptr;
}

extern "C" int32_t tonic_collector_transport(tonic_collector_t* c) {
  return c != nullptr ? c->transport : -1;
}

extern "C" int tonic_collector_bind_segment(
    tonic_collector_t* c, float* d_seg_observations, float* d_seg_actions,
    float* d_seg_next_observations, float* d_seg_rewards, float* d_seg_resets,
    float* d_seg_terminations, float* d_seg_log_probs, float* d_norm_acc, int64_t rows) {
  TONIC_REQUIRE(c && d_seg_observations && d_seg_actions && d_seg_next_observations &&
                    d_seg_rewards && d_seg_resets && d_seg_terminations && d_seg_log_probs &&
                    rows > 0 && !c->live,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_collector_bind_segment: bad argument");
  float* seg[7] = {d_seg_observations, d_seg_actions, d_seg_next_observations, d_seg_rewards,
                   d_seg_resets, d_seg_terminations, d_seg_log_probs};
  memcpy(c->seg, seg, sizeof(seg));
  c->norm_acc = d_norm_acc;
  c->rows = rows;
  if (d_norm_acc != nullptr && c->hist_rows < rows + 1) {
    if (c->d_norm_hist) (void)hipFree(c->d_norm_hist);
    c->d_norm_hist = nullptr;
    TONIC_HIP(hipMalloc(reinterpret_cast<void**>(&c->d_norm_hist),
                        (size_t)(rows + 1) * 2 * c->O * sizeof(float)),
              "hipMalloc of the normaliser history");
    c->hist_rows = rows + 1;
  }
  return TONIC_OK;
}

extern "C" int tonic_collector_begin_rollout(tonic_collector_t* c, const float* d_actor_params,
                                             void* learner_stream) {
  TONIC_REQUIRE(c && d_actor_params, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_collector_begin_rollout: bad argument");
  // the parameters were written on the learner's stream: order the pack behind them
  TONIC_HIP(hipEventRecord(c->learner_done, as_stream(learner_stream)), "hipEventRecord");
  TONIC_HIP(hipStreamWaitEvent(c->stream, c->learner_done, 0), "hipStreamWaitEvent");
  c->actor_params = d_actor_params;               // (wide: read in place, they rest during a rollout)
  if (!c->wide) {
    const int status = launch_actor_pack(d_actor_params, c->d_packed, c->O, c->A, c->stream);
    if (status != TONIC_OK) return status;
  }
  c->actor_packed = true;
  c->hist_loaded = false;             // the first step of the rollout seeds the history
  return TONIC_OK;
}

namespace {

Collect16Args step_arguments(tonic_collector* c) {
  Collect16Args a{};
  a.packed = c->d_packed;
  a.obs = field(c, TONIC_COLLECTOR_OBSERVATIONS);
  a.next_obs = field(c, TONIC_COLLECTOR_NEXT_OBSERVATIONS);
  a.rewards = field(c, TONIC_COLLECTOR_REWARDS);
  a.resets = field(c, TONIC_COLLECTOR_RESETS);
  a.terminations = field(c, TONIC_COLLECTOR_TERMINATIONS);
  a.seg_obs = c->seg[0]; a.seg_act = c->seg[1]; a.seg_next = c->seg[2]; a.seg_rew = c->seg[3];
  a.seg_rst = c->seg[4]; a.seg_term = c->seg[5]; a.seg_lp = c->seg[6];
  a.norm_acc = c->norm_acc != nullptr ? c->d_norm_hist : nullptr;
  a.norm_stride = 2 * c->O;
  a.actions_out = field(c, TONIC_COLLECTOR_ACTIONS);
  a.W = c->W; a.O = c->O; a.A = c->A;
  if (c->transport != 1) {
    a.done_flags = reinterpret_cast<unsigned*>(c->mapped + c->host->offset[TONIC_COLLECTOR_DONE_FLAGS]);
    if ((int64_t)c->W * c->O >= kRecordFromSegment && c->norm_acc != nullptr) a.tile_done = c->d_tile_done;
    // (read per step: the environment may make its promise after the collector was created)
    a.next_from_obs = (int64_t)c->W * c->O >= kRecordFromSegment &&
                      __atomic_load_n(&c->host->carry_over, __ATOMIC_ACQUIRE) != 0;
  }
  a.stamps = c->d_stamps;
  return a;
}

// transport 2: (re)starts the resident kernel; it waits for command number `first_seq`.
int launch_resident(tonic_collector* c, unsigned first_seq) {
  TONIC_HIP(hipMemsetAsync(c->d_relay, 0, kRelayBytes, c->stream), "hipMemsetAsync");
  CollectResident r{};
  r.command = reinterpret_cast<const unsigned long long*>(
      (c->transport == 3 ? c->window : c->mapped) + offsetof(BlockHeader, command));
  r.relay = c->d_relay;
  r.claims = c->d_relay + 64;                     // (the park notice has a 256-byte line of its own)
  r.parked = reinterpret_cast<unsigned*>(c->mapped + offsetof(BlockHeader, parked));
  r.eps0 = field(c, TONIC_COLLECTOR_EPS0);
  r.eps1 = field(c, TONIC_COLLECTOR_EPS1);
  r.first_seq = first_seq;
  r.park_ticks = (unsigned long long)(c->park_us * 100.0);        // 100 MHz wall clock
  const char* pause = getenv("TONIC_AMD_COLLECTOR_POLL_SLEEP");
  r.poll_sleep = pause != nullptr ? atoi(pause) : 1;
  const int status = launch_collect_resident(step_arguments(c), r, c->stream);
  if (status == TONIC_OK) c->live = true;
  c->launches += 1;
  return status;
}

}  // namespace

extern "C" int tonic_collector_arm(tonic_collector_t* c, int64_t row, int32_t eps_slot,
                                   int32_t store_previous) {
  TONIC_REQUIRE(c && c->seg[0] && c->actor_packed && row >= 0 && row < c->rows && eps_slot >= -1 &&
                    eps_slot <= 1 && (!store_previous || row > 0) && !c->armed,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_collector_arm: bad row %lld / slot %d",
                (long long)row, eps_slot);
  // only a RESIDENT kernel can be commanded by somebody who cannot launch (a worker process)
  if (c->transport < 2 || !c->live || c->wide || (c->norm_acc != nullptr && !c->hist_loaded))
    return 0;
  // the block-level ring pushes an armed command into the header's window: only the collector that owns
  // that window may arm (a second collector on the block steps through its own calls: ADVICE r5)
  const uint64_t block_window = __atomic_load_n(&c->host->push_window, __ATOMIC_ACQUIRE);
  if (block_window != 0 && block_window != (uint64_t)reinterpret_cast<uintptr_t>(c->window)) return 0;
  push_noise(c, eps_slot);
  const uint64_t word = ((uint64_t)(c->seq + 1) << 32) | ((uint64_t)row << 8) |
                        (store_previous ? 4u : 0u) | (eps_slot >= 0 ? 2u : 0u) |
                        (eps_slot == 1 ? 1u : 0u);
  c->armed = true;
  c->armed_row = row;
  __atomic_store_n(&c->host->armed, word, __ATOMIC_RELEASE);
  return 1;
}

extern "C" int tonic_collector_claim(tonic_collector_t* c) {
  TONIC_REQUIRE(c != nullptr, TONIC_ERR_INVALID_ARGUMENT, "tonic_collector_claim: bad argument");
  return claim_armed(c);
}

extern "C" int tonic_collector_ppo_step(tonic_collector_t* c, int64_t row, int32_t eps_slot,
                                        int32_t store_previous) {
  TONIC_REQUIRE(c && c->seg[0] && c->actor_packed, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_collector_ppo_step: bind_segment / begin_rollout first");
  TONIC_REQUIRE(row >= 0 && row < c->rows && eps_slot >= -1 && eps_slot <= 1 &&
                    (!store_previous || row > 0),
                TONIC_ERR_INVALID_ARGUMENT, "tonic_collector_ppo_step: bad row %lld / slot %d",
                (long long)row, eps_slot);
  claim_armed(c);
  TONIC_REQUIRE(!c->waiting, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_collector_ppo_step: the previous step's actions were never waited for");
  if (c->transport == 1) {
    const int status = stage_inputs(c, eps_slot < 0 ? 0 : eps_slot);
    if (status != TONIC_OK) return status;
  }
  if (c->norm_acc != nullptr && !c->hist_loaded) {
    // (stream-ordered before the launch of this step; the resident kernel is not running yet)
    TONIC_REQUIRE(!c->live, TONIC_ERR_INVALID_ARGUMENT,
                  "tonic_collector_ppo_step: begin_rollout while the resident kernel runs");
    TONIC_HIP(hipMemcpyAsync(c->d_norm_hist + row * 2 * c->O, c->norm_acc,
                             (size_t)2 * c->O * sizeof(float), hipMemcpyDeviceToDevice, c->stream),
              "seeding the normaliser history");
    c->hist_loaded = true;
  }
  c->last_row = row;
  c->seq += 1;
  __atomic_store_n(&c->host->act_seq, c->seq, __ATOMIC_RELEASE);
  if (c->transport >= 2) {
    if (!c->live) {
      const int status = launch_resident(c, c->seq);
      if (status != TONIC_OK) return status;
    }
    const uint64_t word = ((uint64_t)c->seq << 32) | ((uint64_t)row << 8) |
                          (store_previous ? 4u : 0u) | (eps_slot >= 0 ? 2u : 0u) |
                          (eps_slot == 1 ? 1u : 0u);
    push_noise(c, eps_slot);
    issue_command_of(c, word);
    c->waiting = true;
    return TONIC_OK;
  }
  if (c->wide) {
    WideCollect w{};
    w.params = c->actor_params;
    w.obs = field(c, TONIC_COLLECTOR_OBSERVATIONS);
    w.eps = eps_slot < 0 ? nullptr
                         : field(c, eps_slot == 0 ? TONIC_COLLECTOR_EPS0 : TONIC_COLLECTOR_EPS1);
    w.next_obs = field(c, TONIC_COLLECTOR_NEXT_OBSERVATIONS);
    w.rewards = field(c, TONIC_COLLECTOR_REWARDS);
    w.resets = field(c, TONIC_COLLECTOR_RESETS);
    w.terminations = field(c, TONIC_COLLECTOR_TERMINATIONS);
    w.seg_obs = c->seg[0]; w.seg_act = c->seg[1]; w.seg_next = c->seg[2]; w.seg_rew = c->seg[3];
    w.seg_rst = c->seg[4]; w.seg_term = c->seg[5]; w.seg_lp = c->seg[6];
    w.norm_hist = c->norm_acc != nullptr ? c->d_norm_hist : nullptr;
    w.actions_out = field(c, TONIC_COLLECTOR_ACTIONS);
    w.done_flags = reinterpret_cast<unsigned*>(c->mapped + c->host->offset[TONIC_COLLECTOR_DONE_FLAGS]);
    w.done_seq = c->seq;
    w.row = row; w.outcome_row = store_previous ? row - 1 : -1; w.W = c->W; w.O = c->O; w.A = c->A;
    const int status = wide_collect_step(w, c->d_wide_ws, c->wide_ws_bytes, c->stream);
    if (status != TONIC_OK) return status;
    c->waiting = true;
    return TONIC_OK;
  }
  Collect16Args a = step_arguments(c);
  a.eps = eps_slot < 0 ? nullptr
                       : field(c, eps_slot == 0 ? TONIC_COLLECTOR_EPS0 : TONIC_COLLECTOR_EPS1);
  a.row = row;
  a.outcome_row = store_previous ? row - 1 : -1;
  a.done_seq = c->seq;
  const int status = launch_collect16(a, c->stream);
  if (status != TONIC_OK) return status;
  if (c->transport == 1) {
    const int64_t at = c->host->offset[TONIC_COLLECTOR_ACTIONS];
    TONIC_HIP(hipMemcpyAsync(reinterpret_cast<char*>(c->host) + at, c->staged + at,
                             (size_t)(c->W * c->A * 4), hipMemcpyDeviceToHost, c->stream),
              "tonic_collector: D2H of the actions");
    TONIC_HIP(hipEventRecord(c->actions_out, c->stream), "hipEventRecord");
  }
  c->waiting = true;
  return TONIC_OK;
}

extern "C" int tonic_collector_wait_actions(tonic_collector_t* c, double timeout_s) {
  TONIC_REQUIRE(c && c->waiting, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_collector_wait_actions: no step in flight");
  const double deadline = now_s() + timeout_s;
  const uint32_t* flags = reinterpret_cast<const uint32_t*>(
      reinterpret_cast<const char*>(c->host) + c->host->offset[TONIC_COLLECTOR_DONE_FLAGS]);
  const int words = c->q_words > 0 ? c->q_words : c->wide ? wide_collect_words(c->W) : collect16_blocks(c->W);
  int arrived = 0;                                   // words [0, arrived) already carry c->seq
  for (uint64_t spins = 0;; ++spins) {
    bool done;
    if (c->transport != 1) {
      while (arrived < words && __atomic_load_n(flags + arrived, __ATOMIC_ACQUIRE) == c->seq)
        ++arrived;
      done = arrived == words;
      const uint32_t parked = c->transport >= 2 && !done
                                  ? __atomic_load_n(&c->host->parked, __ATOMIC_ACQUIRE) : 0u;
      // (a notice for another command while this one is incomplete: the kernel may be gone all the
      //  same — a slot of this command gave up on rows of workgroups that had left — look at the stream)
      if (parked != 0 && (parked == c->seq || hipStreamQuery(c->stream) == hipSuccess)) {
        // the resident kernel parked while this command was on its way: start it again (the
        // command word still holds the command; work done twice is idempotent WHILE THE HOST WAITS,
        // see the kernel).  Some slots of this command were run before their workgroups left and
        // have their completion words out; the new launch runs every slot again, and none of them
        // may still be reading the block when this call returns: all words start over.
        TONIC_HIP(hipStreamSynchronize(c->stream), "hipStreamSynchronize");
        c->live = false;
        c->relaunches += 1;
        __atomic_store_n(&c->host->parked, 0u, __ATOMIC_RELEASE);
        for (int w = 0; w < words; ++w)
          __atomic_store_n(const_cast<uint32_t*>(flags) + w, 0u, __ATOMIC_RELAXED);
        __atomic_thread_fence(__ATOMIC_SEQ_CST);
        arrived = 0;
        const int status = launch_resident(c, c->seq);
        if (status != TONIC_OK) return status;
      }
    } else {
      const hipError_t e = hipEventQuery(c->actions_out);
      if (e != hipSuccess && e != hipErrorNotReady) {
        set_error("tonic_collector_wait_actions: %s", hipGetErrorString(e));
        return TONIC_ERR_LAUNCH;
      }
      done = e == hipSuccess;
    }
    if (done) break;
    cpu_relax();
    if ((spins & 0xfff) == 0xfff) {
      if (c->transport != 1) {          // a failed launch never writes the flags: look at the stream
        const hipError_t e = hipStreamQuery(c->stream);
        if (e != hipSuccess && e != hipErrorNotReady) {
          set_error("tonic_collector_wait_actions: %s", hipGetErrorString(e));
          return TONIC_ERR_LAUNCH;
        }
      }
      if (now_s() > deadline) {
        char missing[160] = "";
        int at = 0;
        for (int w = 0; w < words && at < 130 && c->transport != 1; ++w) {
          const uint32_t value = __atomic_load_n(flags + w, __ATOMIC_ACQUIRE);
          if (value != c->seq) at += snprintf(missing + at, sizeof(missing) - at, " [%d]=%u", w, value);
        }
        set_error("tonic_collector_wait_actions: no actions within %.1f s (command %u: %d of %d "
                  "completion words in order, others:%s; parked at %u, stream %s)", timeout_s, c->seq,
                  arrived, words, missing,
                  c->transport >= 2 ? __atomic_load_n(&c->host->parked, __ATOMIC_ACQUIRE) : 0u,
                  hipStreamQuery(c->stream) == hipSuccess ? "idle" : "busy");
        return TONIC_ERR_TIMEOUT;
      }
    }
  }
  c->waiting = false;
  c->q_words = 0;
  return TONIC_OK;
}

namespace tonic {
void collector_shape(tonic_collector_t* c, int64_t* W, int* O, int* A) { *W = c->W; *O = c->O; *A = c->A; }

int collector_begin_q_step(tonic_collector_t* c, int eps_slot, int extra_words, CollectorStep* out) {
  TONIC_REQUIRE(c != nullptr && out != nullptr && eps_slot >= -1 && eps_slot <= 0, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_collector_q_act: bad argument (noise slot %d)", eps_slot);
  TONIC_REQUIRE(!c->waiting && !c->live && !c->armed, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_collector_q_act: another step of this collector is in flight");
  const int words = (int)((c->W + 15) / 16) + extra_words;
  // (the block's completion field: collect16_blocks(W) words, rounded up to 256 bytes)
  TONIC_REQUIRE(words <= 64 || words <= collect16_blocks(c->W), TONIC_ERR_UNSUPPORTED_SHAPE,
                "tonic_collector_q_act: %d completion words do not fit the block", words);
  c->seq += 1;
  __atomic_store_n(&c->host->act_seq, c->seq, __ATOMIC_RELEASE);
  out->W = c->W; out->O = c->O; out->A = c->A;
  auto mapped = [&](int f) { return reinterpret_cast<float*>(c->mapped + c->host->offset[f]); };
  out->observations = mapped(TONIC_COLLECTOR_OBSERVATIONS);
  out->eps = eps_slot < 0 ? nullptr : mapped(TONIC_COLLECTOR_EPS0);
  out->actions_out = mapped(TONIC_COLLECTOR_EPS1);
  out->actions = mapped(TONIC_COLLECTOR_ACTIONS);
  out->next_observations = mapped(TONIC_COLLECTOR_NEXT_OBSERVATIONS);
  out->rewards = mapped(TONIC_COLLECTOR_REWARDS);
  out->resets = mapped(TONIC_COLLECTOR_RESETS);
  out->terminations = mapped(TONIC_COLLECTOR_TERMINATIONS);
  out->done_flags = reinterpret_cast<unsigned*>(c->mapped + c->host->offset[TONIC_COLLECTOR_DONE_FLAGS]);
  out->seq = c->seq;
  c->q_words = words;
  c->waiting = true;
  return TONIC_OK;
}
}  // namespace tonic

extern "C" int tonic_collector_end_rollout(tonic_collector_t* c, int64_t last_row,
                                           void* learner_stream) {
  if (c != nullptr) claim_armed(c);
  TONIC_REQUIRE(c && c->seg[0] && last_row < c->rows && !c->waiting, TONIC_ERR_INVALID_ARGUMENT,
                "tonic_collector_end_rollout: bad argument");
  const int64_t final_row = last_row;          // the last row that belongs to the rollout
  if (c->live) {
    // transport 2: a stop command — the copy workgroups store the pending outcome first — then
    // the resident kernel leaves and the stream drains
    c->seq += 1;
    const uint64_t word = ((uint64_t)c->seq << 32) | ((uint64_t)(last_row + 1) << 8) |
                          (last_row >= 0 ? 4u : 0u) | 8u;
    issue_command_of(c, word);
    c->waiting = true;
    const int status = tonic_collector_wait_actions(c, 60.0);
    if (status != TONIC_OK) return status;
    c->live = false;                   // (a kernel relaunched by the wait has left as well)
    last_row = -1;                     // stored
  }
  if (last_row >= 0) {
    if (c->transport == 1) {
      const int status = stage_inputs(c, 0);
      if (status != TONIC_OK) return status;
    }
    const int64_t items = c->W * c->O;
    const int blocks = (int)((items + 255) / 256 < 64 ? (items + 255) / 256 : 64);
    hipLaunchKernelGGL(outcome_store_kernel, dim3(blocks), dim3(256), 0, c->stream,
                       field(c, TONIC_COLLECTOR_NEXT_OBSERVATIONS),
                       field(c, TONIC_COLLECTOR_REWARDS), field(c, TONIC_COLLECTOR_RESETS),
                       field(c, TONIC_COLLECTOR_TERMINATIONS), c->seg[2], c->seg[3], c->seg[4],
                       c->seg[5], last_row, c->W, c->O);
    TONIC_CHECK_LAUNCH("outcome_store_kernel");
  }
  if (c->norm_acc != nullptr && c->hist_loaded && final_row >= 0) {   // the sums after that row
    TONIC_HIP(hipMemcpyAsync(c->norm_acc, c->d_norm_hist + (final_row + 1) * 2 * c->O,
                             (size_t)2 * c->O * sizeof(float), hipMemcpyDeviceToDevice, c->stream),
              "reading the normaliser history back");
    c->hist_loaded = false;
  }
  // The learner reads the Segment on its own stream; the block may be overwritten by the next
  // environment step as soon as this call returns.
  TONIC_HIP(hipEventRecord(c->collect_done, c->stream), "hipEventRecord");
  TONIC_HIP(hipStreamWaitEvent(as_stream(learner_stream), c->collect_done, 0),
            "hipStreamWaitEvent");
  TONIC_HIP(hipEventSynchronize(c->collect_done), "hipEventSynchronize");
  c->actor_packed = false;            // the learner is about to change the parameters
  return TONIC_OK;
}

// ---- tonic_stream_gate: one wave holds a stream until the host stores to a pinned word
namespace {
__global__ __launch_bounds__(64) void stream_gate_kernel(const unsigned* word, unsigned value,
                                                         unsigned long long ticks) {
  const unsigned long long t0 = wall_clock64();
  while (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) < value &&
         wall_clock64() - t0 < ticks)
    for (int k = 0; k < 3; ++k) __builtin_amdgcn_s_sleep(127);   // ~5 us between polls of the host's cache line
}
}  // namespace

extern "C" int tonic_stream_gate(const uint32_t* host_word, uint32_t value, double timeout_seconds,
                                 void* stream) {
  TONIC_REQUIRE(host_word != nullptr && timeout_seconds > 0.0 && timeout_seconds <= 3600.0,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_stream_gate: bad argument");
  void* device_word = nullptr;
  TONIC_HIP(hipHostGetDevicePointer(&device_word, const_cast<uint32_t*>(host_word), 0),
            "tonic_stream_gate: the word is not page-locked host memory");
  hipLaunchKernelGGL(stream_gate_kernel, dim3(1), dim3(64), 0, as_stream(stream),
                     static_cast<const unsigned*>(device_word), value,
                     (unsigned long long)(timeout_seconds * 1e8));
  TONIC_CHECK_LAUNCH("tonic_stream_gate");
  return TONIC_OK;
}

// ---- test tool (include/tonic_hip_dev.h)
namespace {
__global__ __launch_bounds__(256) void occupy_kernel(unsigned long long ticks) {
  extern __shared__ float held[];
  const unsigned long long t0 = wall_clock64();
  if (threadIdx.x == 0) held[0] = 0.f;
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
}  // namespace

extern "C" int tonic_debug_occupy(int32_t workgroups, double milliseconds, void* stream) {
  TONIC_REQUIRE(workgroups >= 1 && workgroups <= 4096 && milliseconds > 0.0 && milliseconds <= 2000.0,
                TONIC_ERR_INVALID_ARGUMENT, "tonic_debug_occupy: %d workgroups for %.1f ms",
                workgroups, milliseconds);
  constexpr int kBytes = 100 * 1024;
  static bool configured = false;
  if (!configured) {
    TONIC_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(occupy_kernel),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, kBytes),
              "hipFuncSetAttribute");
    configured = true;
  }
  hipLaunchKernelGGL(occupy_kernel, dim3(workgroups), dim3(256), kBytes, as_stream(stream),
                     (unsigned long long)(milliseconds * 1e5));
  TONIC_CHECK_LAUNCH("tonic_debug_occupy");
  return TONIC_OK;
}

