"""Single-node multi-GPU data parallelism: one process per GPU, RCCL over xGMI.

The worker axis W is sharded contiguously across ranks (rank r owns workers
[r*W/G, (r+1)*W/G), the reference's group-major order).  Rollout, store, critic evaluation and
the lambda-return scan need no communication (worker columns are independent); the learner
exchanges ONE flat float32 buffer per optimizer step — [gradient sums | 8 statistic sums] —
with a sum all-reduce, after which every rank applies the identical Adam step scaled by
1/N_global.  Advantage normalisation needs the global mean/std: ranks all-reduce
(sum, sum of squares, min, max-as-negated-min, count) once per update.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """Initialises torch.distributed from RANK / WORLD_SIZE / MASTER_* (torchrun contract)."""
    world = int(os.environ.get('WORLD_SIZE', '1'))
    exercise = os.environ.get('TONIC_AMD_EXERCISE_EXCHANGE') == '1'
    if (world <= 1 and not exercise) or dist.is_initialized():
        return rank(), world_size()
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    os.environ.setdefault('MASTER_PORT', '29500')
    if backend is None:
        backend = os.environ.get('TONIC_AMD_BACKEND') or (
            'nccl' if torch.cuda.is_available() else 'gloo')           # "nccl" is RCCL on ROCm
    if torch.cuda.is_available():
        torch.cuda.set_device(int(os.environ.get('LOCAL_RANK', 0)) % torch.cuda.device_count())
    dist.init_process_group(backend=backend)
    return rank(), world_size()


def rank():
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def world_size():
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def launch_rank():
    """(rank, world) of this process as the launcher announced them (RANK / WORLD_SIZE of
    ``python -m torch.distributed.run``), available before torch.distributed is initialised:
    the environments are seeded before any agent exists."""
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    world = int(os.environ.get('WORLD_SIZE', '1'))
    return (int(os.environ.get('RANK', '0')), world) if world > 1 else (0, 1)


def global_noise():
    """TONIC_AMD_GLOBAL_NOISE=1: every rank draws the action / exploration noise of ALL workers
    and keeps the rows of its own (cost grows with the number of ranks), so that N ranks with W / N
    workers each consume the random streams exactly like one process with W workers — what the
    multi-rank learning-curve tests use.  Default: each rank draws for its own workers from a
    generator re-seeded with seed + rank once the (replicated) parameters exist."""
    return os.environ.get('TONIC_AMD_GLOBAL_NOISE', '0') == '1'


def broadcast_from_first(tensors):
    """Rank 0's values everywhere (parameters at start-up: replicas must not depend on every rank
    having been given the same seed)."""
    if world_size() > 1:
        for tensor in tensors:
            dist.broadcast(tensor, src=0)


def exchanging():
    """True when the learner exchanges gradient sums between ranks.  Besides world_size > 1 that is
    a process group of ONE rank with TONIC_AMD_EXERCISE_EXCHANGE=1: a test hook that drives the
    whole exchange schedule (asynchronous RCCL all-reduces, stream hand-overs) on a single GPU —
    with one rank the reductions are identities, so results must equal the plain path bit for bit."""
    if world_size() > 1:
        return True
    return (dist.is_available() and dist.is_initialized()
            and os.environ.get('TONIC_AMD_EXERCISE_EXCHANGE') == '1')


def shard_bounds(total, r=None, world=None):
    """Contiguous [begin, end) of `total` items owned by rank r (remainder to the low ranks)."""
    r = rank() if r is None else r
    world = world_size() if world is None else world
    base, extra = divmod(total, world)
    begin = r * base + min(r, extra)
    return begin, begin + base + (1 if r < extra else 0)


def allreduce_sums(buffer):
    """In-place sum all-reduce of a flat buffer (no-op for a single process)."""
    if world_size() > 1:
        dist.all_reduce(buffer, op=dist.ReduceOp.SUM)
    return buffer


def combine_advantage_moments(total, total_sq, minimum, maximum, count):
    """Global advantage mean / population std from per-rank float64 moments
    (tonic/replays/segments.py:43-46 over the union of all shards)."""
    if world_size() > 1:
        sums = torch.tensor([total, total_sq, count], dtype=torch.float64)
        ext = torch.tensor([-minimum, maximum], dtype=torch.float64)
        if dist.get_backend() == 'nccl':
            sums, ext = sums.cuda(), ext.cuda()
        dist.all_reduce(sums, op=dist.ReduceOp.SUM)
        dist.all_reduce(ext, op=dist.ReduceOp.MAX)
        total, total_sq, count = sums.tolist()
        minimum, maximum = -float(ext[0]), float(ext[1])
    mean = total / count
    var = max(total_sq / count - mean * mean, 0.0)
    constant = minimum == maximum
    std = 0.0 if constant else var ** 0.5
    return mean, std, constant and minimum == 0.0


def _parse_cpulist(text):
    cpus = set()
    for part in text.strip().split(','):
        if '-' in part:
            first, last = part.split('-')
            cpus.update(range(int(first), int(last) + 1))
        elif part:
            cpus.add(int(part))
    return cpus


_bound = {}
_original_affinity = None       # the mask the process had before the first bind (restore_affinity puts it back)


_holders = 0


def hold_affinity():
    """An agent that asked for the binding and will say when it is done with it (release_affinity)."""
    global _holders
    _holders += 1


def release_affinity():
    global _holders
    _holders = max(_holders - 1, 0)
    if _holders == 0:
        restore_affinity()


def restore_affinity():
    """Undoes bind_near_gpu: the process (this thread and those started from now on) gets back the CPU mask it
    had before the first agent bound it — the agents' close() gets here when the last agent that holds the
    binding closes.  Threads and worker processes started while bound keep what they inherited."""
    global _original_affinity
    if _original_affinity is not None and hasattr(os, 'sched_setaffinity'):
        try:
            os.sched_setaffinity(0, _original_affinity)
        except OSError:
            pass
    _original_affinity = None
    _bound.clear()


def bind_near_gpu(device):
    """Keeps this process on the CPUs of the NUMA node its GPU hangs off (sched_setaffinity; threads and worker
    processes started afterwards inherit it).  The collect loop is a chain of PCIe round trips between this
    process's memory and the GPU: started on the far socket of a two-socket host — where the scheduler puts every
    second process — an environment step of 256 workers takes 13.0 us instead of 10.7 (profiles/r05_numa.md;
    the collector also moves the shared block's pages: tonic_collector_create).  One process per GPU binds to its
    own GPU's node, as launchers do with numactl.  No-op when the current affinity is already inside that node,
    when the node is unknown, or with TONIC_AMD_NUMA_BIND=0.  Returns the CPUs bound to (None: nothing done).
    A process-wide side effect of building an agent, so it is said once on the log, and `restore_affinity()`
    (called by the agents' close()) puts the original mask back."""
    global _original_affinity
    index = device.index if device.index is not None else torch.cuda.current_device()
    if index in _bound:
        return _bound[index]
    _bound[index] = None
    if os.environ.get('TONIC_AMD_NUMA_BIND', '1') == '0' or not hasattr(os, 'sched_setaffinity'):
        return None
    try:
        p = torch.cuda.get_device_properties(index)
        address = '%04x:%02x:%02x.0' % (getattr(p, 'pci_domain_id', 0), p.pci_bus_id, p.pci_device_id)
        local = _parse_cpulist(open(f'/sys/bus/pci/devices/{address}/local_cpulist').read())
        allowed = os.sched_getaffinity(0)
        target = local & allowed
        if target and not allowed <= local:
            if _original_affinity is None:
                _original_affinity = set(allowed)
            os.sched_setaffinity(0, target)
            _bound[index] = target
            from tonic_amd.utils import logger
            logger.log(f'tonic_amd: process bound to the {len(target)} CPUs of GPU {index}\'s NUMA node (of '
                       f'{len(allowed)} allowed; TONIC_AMD_NUMA_BIND=0 leaves the affinity alone, '
                       'agent.close() restores it)')
    except (OSError, ValueError, AttributeError):
        pass                                   # (no sysfs entry, a container without the right: stay put)
    return _bound[index]


_one_shot = None
_choice = None          # {'kind': 'oneshot' | 'rccl', 'reason': ...} once the ranks have decided


def allreduce_choice():
    """What the learner's gradient exchange uses and why (None before the first exchange):
    bench.py prints it next to the measured latencies of both."""
    return _choice


def _agree(ok, reason=''):
    """The AND over all ranks of a local verdict (with the first objecting rank's reason), so that
    every rank takes the same branch — a rank alone on the other side of a collective is a hang."""
    verdicts = [None] * world_size()
    dist.all_gather_object(verdicts, (bool(ok), str(reason)))
    for r, (fine, why) in enumerate(verdicts):
        if not fine:
            return False, f'rank {r}: {why}'
    return True, ''


def _peers_reachable():
    """tonic_allreduce_f32 writes into windows of the peers' device memory: every rank on a device of
    its own, all in one process namespace (one node), and hipDeviceCanAccessPeer for every pair.
    Every rank runs the SAME collectives here whatever it finds locally — one gather of (host, device,
    has a GPU), a verdict formed from the gathered list, one _agree — so a job whose ranks differ (one
    without a GPU, one on another host) cannot end up with mismatched collectives."""
    import socket

    from tonic_amd import _lib
    has_gpu = torch.cuda.is_available()
    me = (socket.gethostname(), torch.cuda.current_device() if has_gpu else -1, has_gpu)
    everyone = [None] * world_size()
    dist.all_gather_object(everyone, me)
    devices = [device for _, device, _ in everyone]
    ok, why = True, ''
    if not all(gpu for _, _, gpu in everyone):
        ok, why = False, 'a rank without a GPU'
    elif len({host for host, _, _ in everyone}) > 1:
        ok, why = False, 'ranks on several hosts'
    elif len(set(devices)) < len(devices):
        ok, why = False, f'ranks share devices {devices}'
    else:
        try:
            lib = _lib.load()
            for peer in devices:
                can = lib.tonic_comm_can_access_peer(me[1], peer)
                if can != 1:
                    reason = lib.tonic_last_error().decode() if can < 0 else 'hipDeviceCanAccessPeer = 0'
                    ok, why = False, f'device {me[1]} -> device {peer}: {reason}'
                    break
        except Exception as error:      # (a local failure is a verdict, not a missing collective)
            ok, why = False, str(error)
    return _agree(ok, why)


def one_shot(max_floats):
    """The process-wide OneShotAllReduce (tonic_allreduce_f32) when the gradient exchange should use
    it, else None: callers then use torch.distributed (RCCL).  TONIC_AMD_ALLREDUCE:
      rccl (default)  torch.distributed's process group (RCCL on a GPU node).  The one-shot exchange
                      has only ever run between processes that share ONE device (tests) — no multi-GPU
                      node was available to any round — so it is opt-in until a run on xGMI is recorded;
      auto            the ranks decide TOGETHER, once: every rank on a device of its own on one
                      host, peer access between every pair (hipDeviceCanAccessPeer), the windows
                      open (hipIpc*), and a self-test of exact sums over several sizes and both slot
                      parities passes on every rank — else RCCL, with the reason kept
                      (`allreduce_choice()`);
      oneshot         always (tests: also between processes that share one device).
    The windows grow on demand before first use only."""
    global _one_shot, _choice
    if world_size() == 1:
        return None
    if _one_shot is not None and _one_shot.max_floats >= max_floats:
        return _one_shot
    mode = os.environ.get('TONIC_AMD_ALLREDUCE', 'rccl')
    if mode == 'rccl' or (_choice is not None and _choice['kind'] == 'rccl'):
        if _choice is None:
            _choice = dict(kind='rccl', reason='TONIC_AMD_ALLREDUCE=rccl')
        return None
    if mode not in ('auto', 'oneshot'):
        raise ValueError(f'TONIC_AMD_ALLREDUCE={mode!r}: auto, oneshot or rccl')
    if _one_shot is not None:               # regrown windows: the old communicator is released
        torch.cuda.synchronize()
        _one_shot.check()
        _one_shot.close()
        _one_shot = None
    if mode == 'oneshot':
        _one_shot = OneShotAllReduce(max(max_floats, 1 << 18))
        _choice = dict(kind='oneshot', reason='TONIC_AMD_ALLREDUCE=oneshot')
        return _one_shot
    ok, why = _peers_reachable()
    candidate = None
    # From here every rank walks the same three _agree steps whatever happens locally: a rank that fails
    # (an exception included) carries its objection INTO the next _agree instead of skipping it, so no
    # rank is ever left alone in a collective.
    local, local_why = ok, why
    if local:
        try:
            candidate = OneShotAllReduce(max(max_floats, 1 << 18), tolerant=True)
            local, local_why = candidate.handle is not None, candidate.error
        except Exception as error:
            local, local_why = False, str(error)
    if ok:
        ok, why = _agree(local, local_why)
    if ok:
        try:
            local, local_why = candidate.self_test()
        except Exception as error:
            local, local_why = False, str(error)
        ok, why = _agree(local, local_why)
    if not ok:
        if candidate is not None:
            candidate.close()
        _choice = dict(kind='rccl', reason=why)
        return None
    _one_shot = candidate
    _choice = dict(kind='oneshot', reason='peer access between every pair of ranks, self-test passed')
    return _one_shot


def check_one_shot():
    """Raises if a peer missed some tonic_allreduce_f32 since the last check (the kernel then
    returned with its sums unreduced and an error status instead of hanging).  The learners call
    this once per update, right after the read-back that already synchronises the stream, so
    diverged replicas never survive an update silently."""
    if _one_shot is not None:
        _one_shot.check()


class OneShotAllReduce:
    """``tonic_allreduce_f32`` (include/tonic_hip.h): in-place sum all-reduce of small float32
    device buffers as ONE launch per rank — every rank writes its buffer into a window of every
    peer and adds the contributions in rank order (deterministic, identical on all ranks).
    ``torch.distributed`` is only the bootstrap channel that carries the IPC handles once.
    Validated between processes that share one GPU (tests/test_gpu_multirank.py); on a node with
    several GPUs ``parallel.one_shot`` selects it after its peer-access checks and ``self_test``.

    ``tolerant``: a step of the set-up that fails on THIS rank is remembered (``error``; ``handle``
    is then None) instead of raised, and the collectives of the set-up are still taken part in — the
    caller lets the ranks agree on the outcome."""

    def __init__(self, max_floats, tolerant=False):
        import ctypes

        from tonic_amd import _lib
        self._lib_module = _lib
        self.lib = lib = _lib.load()
        self.rank, self.world = rank(), world_size()
        self.max_floats = max_floats
        self.error = ''

        def step(status, what):
            if status != 0 and not self.error:
                self.error = f'{what}: {lib.tonic_last_error().decode()}'
                if not tolerant:
                    raise _lib.TonicHipError(self.error)
            return status == 0

        handle = ctypes.c_void_p()
        step(lib.tonic_comm_init(ctypes.byref(handle), self.rank, self.world, max_floats),
             'tonic_comm_init')
        self.handle = handle.value if not self.error else None
        if self.world > 1:
            size = lib.tonic_comm_handle_bytes()
            mine = ctypes.create_string_buffer(size)
            if self.handle is not None:
                step(lib.tonic_comm_export(self.handle, mine), 'tonic_comm_export')
            gathered = [None] * self.world
            dist.all_gather_object(gathered, (mine.raw, not self.error))
            if all(fine for _, fine in gathered) and self.handle is not None:
                everyone = ctypes.create_string_buffer(b''.join(raw for raw, _ in gathered),
                                                       size * self.world)
                step(lib.tonic_comm_connect(self.handle, everyone), 'tonic_comm_connect')
            elif not self.error:
                self.error = 'a peer could not export its window'
            dist.barrier()                  # every window is mapped everywhere before first use
        if self.error and self.handle is not None:
            lib.tonic_comm_destroy(self.handle)
            self.handle = None

    def self_test(self, calls=24, timeout_s=5.0):
        """Exact sums through the windows before the learner relies on them: integer-valued float32
        patterns (every partial sum is exact, so the expected bits are known without a second
        exchange) over several sizes — one float, ragged, the gradient buffers' order of
        magnitude, the whole window — `calls` back-to-back calls each (both slot parities, a rank
        running ahead of its peers), with a short timeout.  The FIRST call is checked by itself:
        peers that cannot see each other's stores cost one timeout, not one per call.
        Returns (ok, reason)."""
        _lib = self._lib_module
        if self.handle is None:
            return False, self.error or 'no communicator'
        _lib.check(self.lib.tonic_comm_set_timeout(self.handle, float(timeout_s)), 'set_timeout')
        device = torch.device('cuda', torch.cuda.current_device())
        ok, why = True, ''
        try:
            for n in (1, 1001, 11101, self.max_floats):
                index = torch.arange(n, device=device, dtype=torch.float32) % 251.0
                for call in range(calls):
                    buffer = index * float(self.rank + 1) + float(call)
                    self.all_reduce(buffer)
                    if call == 0 and n == 1:
                        torch.cuda.synchronize()
                        self.check()
                    want = index * float(self.world * (self.world + 1) // 2) + float(self.world * call)
                    if not torch.equal(buffer, want):
                        wrong = int((buffer != want).sum())
                        ok, why = False, f'{wrong} of {n} sums wrong in call {call}'
                        break
                if not ok:
                    break
                torch.cuda.synchronize()
                self.check()
        except _lib.TonicHipError as error:
            ok, why = False, str(error)
        if ok:
            _lib.check(self.lib.tonic_comm_set_timeout(self.handle, 0.0), 'set_timeout')
        return ok, why

    def all_reduce(self, tensor):
        _lib = self._lib_module
        _lib.check(self.lib.tonic_allreduce_f32(self.handle, _lib.ptr(tensor), tensor.numel(),
                                                _lib.current_stream()), 'tonic_allreduce_f32')
        return tensor

    def check(self):
        """Synchronous: raises if a peer failed to arrive in some earlier call."""
        self._lib_module.check(self.lib.tonic_comm_status(self.handle), 'tonic_comm_status')

    def close(self):
        if self.handle is not None:
            self.lib.tonic_comm_destroy(self.handle)
            self.handle = None
