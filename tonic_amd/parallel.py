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


_one_shot = None


def one_shot(max_floats):
    """The process-wide OneShotAllReduce when TONIC_AMD_ALLREDUCE=oneshot asks for it (windows grow
    on demand before first use only), else None: callers then use torch.distributed (RCCL)."""
    global _one_shot
    if os.environ.get('TONIC_AMD_ALLREDUCE', '') != 'oneshot' or world_size() == 1:
        return None
    if _one_shot is None or _one_shot.max_floats < max_floats:
        if _one_shot is not None:           # regrown windows: the old communicator is released
            torch.cuda.synchronize()
            _one_shot.check()
            _one_shot.close()
        _one_shot = OneShotAllReduce(max(max_floats, 1 << 18))
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
    Opt-in for the learner (``TONIC_AMD_ALLREDUCE=oneshot``): validated between processes that
    share one GPU (tests/test_gpu_multirank.py), not yet on an xGMI node."""

    def __init__(self, max_floats):
        import ctypes

        from tonic_amd import _lib
        self._lib_module = _lib
        self.lib = lib = _lib.load()
        self.rank, self.world = rank(), world_size()
        self.max_floats = max_floats
        handle = ctypes.c_void_p()
        _lib.check(lib.tonic_comm_init(ctypes.byref(handle), self.rank, self.world, max_floats),
                   'tonic_comm_init')
        self.handle = handle.value
        if self.world > 1:
            size = lib.tonic_comm_handle_bytes()
            mine = ctypes.create_string_buffer(size)
            _lib.check(lib.tonic_comm_export(self.handle, mine), 'tonic_comm_export')
            gathered = [None] * self.world
            dist.all_gather_object(gathered, mine.raw)
            everyone = ctypes.create_string_buffer(b''.join(gathered), size * self.world)
            _lib.check(lib.tonic_comm_connect(self.handle, everyone), 'tonic_comm_connect')
            dist.barrier()                  # every window is mapped everywhere before first use

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
