"""Python face of the pinned-host batched collector (``tonic_collector_*`` in
``include/tonic_hip.h``; kernels and futex protocol in ``csrc/collector.hip``).

``Block`` is the shared step record: one anonymous ``MAP_SHARED`` mapping created BEFORE the
environment workers are forked (``tonic/environments/distributed.py:97-109`` forks there too),
with NumPy views of its float32 fields.  The environments of ``tonic_amd.environments`` write
their step results straight into it and hand the views out as ``observations`` / ``infos``;
the agent recognises those views by identity and page-locks the very same memory, so between
the simulator's write and the GPU's read no byte is copied on the host.  Arrays that are not
block views (any other environment) are copied into a private block first — same path after.

``Collector`` is the agent side: the GPU handle around a block (``tonic_collector_create``).
"""
import ctypes
import mmap
import os
import weakref

import numpy as np

from tonic_amd import _lib

FIELDS = dict(eps0=0, observations=1, next_observations=2, rewards=3, resets=4, terminations=5,
              eps1=6, actions=7, resets_u8=8, terminations_u8=9)


class Block:
    """One environment step of W workers in shared, page-aligned host memory."""

    _live = weakref.WeakSet()

    def __init__(self, workers, observation_size, action_size, worker_groups=1):
        self.lib = lib = _lib.load()
        self.workers, self.observation_size, self.action_size = workers, observation_size, action_size
        nbytes = lib.tonic_collector_block_bytes(workers, observation_size, action_size)
        if nbytes <= 0:
            raise _lib.TonicHipError(f'bad collector block shape {workers, observation_size, action_size}')
        self.memory = mmap.mmap(-1, nbytes)          # MAP_SHARED | MAP_ANONYMOUS: survives fork
        self.nbytes = nbytes
        self._anchor = ctypes.c_char.from_buffer(self.memory)
        self.address = ctypes.addressof(self._anchor)
        _lib.check(lib.tonic_collector_block_init(self.address, nbytes, workers, observation_size,
                                                  action_size, worker_groups),
                   'tonic_collector_block_init')
        W, O, A = workers, observation_size, action_size

        def view(name, shape, dtype=np.float32):
            offset = lib.tonic_collector_block_offset(self.address, FIELDS[name])
            count = int(np.prod(shape))
            return np.frombuffer(self.memory, dtype, count, offset).reshape(shape)
        self.observations = view('observations', (W, O))
        self.next_observations = view('next_observations', (W, O))
        self.rewards = view('rewards', (W,))
        self.resets = view('resets', (W,))
        self.terminations = view('terminations', (W,))
        self.actions = view('actions', (W, A))
        self.eps = (view('eps0', (W, A)), view('eps1', (W, A)))
        self.resets_bool = view('resets_u8', (W,), np.bool_)
        self.terminations_bool = view('terminations_u8', (W,), np.bool_)
        # What `environment.step` hands out when it writes into this block: persistent READ-ONLY
        # views (the same objects every step).  They alias memory the next step — and the GPU —
        # overwrites in place: a caller that keeps step outputs across steps must copy them (or ask
        # the environment for `copy_outputs=True`); a caller that writes into them gets a
        # ValueError instead of silently corrupting the simulator's record.
        def handout(array):
            out = array.view()
            out.setflags(write=False)
            return out
        self.out_actions = handout(self.actions)
        self.out_observations = handout(self.observations)
        self.out_next_observations = handout(self.next_observations)
        self.out_rewards = handout(self.rewards)
        self.out_resets = handout(self.resets_bool)
        self.out_terminations = handout(self.terminations_bool)
        self.infos = dict(observations=self.out_next_observations, rewards=self.out_rewards,
                          resets=self.out_resets, terminations=self.out_terminations)
        # GPU handles that page-locked this mapping.  They MUST be destroyed (hipHostUnregister)
        # before the mapping goes away: a later mmap may reuse the address range, and a stale
        # registration would make the GPU read the old physical pages.  The finalizer runs
        # before the instance dictionary (and with it the mmap) is released.
        self._handles = []
        self._collectors = {}
        self._ring = _lib.hot('tonic_collector_ring')
        weakref.finalize(self, Block._release, self._handles, lib, os.getpid())
        Block._live.add(self)

    @staticmethod
    def _release(handles, lib, owner):
        # (a forked environment worker inherits this finalizer and runs it at ITS exit: the GPU
        #  handles belong to the process that made them — touching them there is a segmentation
        #  fault in the HIP runtime)
        if os.getpid() != owner:
            return
        for cell in handles:
            if cell[0] is not None:
                lib.tonic_collector_destroy(cell[0])
                cell[0] = None

    @classmethod
    def owner_of(cls, observations):
        """The live block whose `observations` view this very array is (None otherwise)."""
        for block in cls._live:
            if block.out_observations is observations or block.observations is observations:
                return block
        return None

    def set_flags(self, row, reset, termination):
        self.resets[row] = reset
        self.terminations[row] = termination
        self.resets_bool[row] = reset
        self.terminations_bool[row] = termination

    # -- environment-side synchronisation (futex words in the block header; no HIP) -----------
    def submit_actions(self):
        _lib.check(self.lib.tonic_collector_submit_actions(self.address),
                   'tonic_collector_submit_actions')

    def wait_obs(self, timeout):
        return self.lib.tonic_collector_wait_obs(self.address, float(timeout))

    def worker_wait(self, seen, timeout=3600.0):
        return self.lib.tonic_collector_worker_wait(self.address, seen, float(timeout))

    def worker_done(self):
        self.lib.tonic_collector_worker_done(self.address)

    def shutdown(self):
        self.lib.tonic_collector_shutdown(self.address)

    def promise_carry_over(self, promised=True):
        """The writer of this block guarantees: a worker's `observations` row IS its
        `next_observations` row of the step before unless its reset flag is set (what every
        environment of tonic_amd.environments writes).  Many-worker steps then move each
        observation row over PCIe once (tonic_collector_block_carry_over)."""
        if os.environ.get('TONIC_AMD_CARRY_OVER', '1') == '0':        # (developer switch: A/B runs)
            promised = False
        _lib.check(self.lib.tonic_collector_block_carry_over(self.address, 1 if promised else 0),
                   'tonic_collector_block_carry_over')

    def ring(self):
        """The step record is complete: issues the command the agent armed for this moment (if
        any).  An environment calls this once its observations, outcome and flags are in place."""
        return self._ring(self.address)


class Collector:
    """GPU side of a block: fused act + store launches on the collector's own stream."""

    def __init__(self, block, transport=0):
        self.lib = lib = block.lib
        handle = ctypes.c_void_p()
        _lib.check(lib.tonic_collector_create(ctypes.byref(handle), block.address, transport),
                   'tonic_collector_create')
        self._cell = [handle.value]          # shared with the block, which outlives the handle
        block._handles.append(self._cell)
        self.requested = transport
        self.transport = lib.tonic_collector_transport(handle)      # (what is in effect: wide -> 0, no window -> 2)
        self._step = _lib.hot('tonic_collector_ppo_step')          # bound once: the per-step hot calls
        self._wait = _lib.hot('tonic_collector_wait_actions')
        self._arm = _lib.hot('tonic_collector_arm')
        self._claim = _lib.hot('tonic_collector_claim')

    @property
    def handle(self):
        if self._cell[0] is None:
            raise _lib.TonicHipError('this collector was closed (or its block released)')
        return self._cell[0]

    @classmethod
    def for_block(cls, block, transport=0):
        """One collector per block and transport."""
        if transport not in block._collectors:
            block._collectors[transport] = cls(block, transport)
        return block._collectors[transport]

    def close(self):
        if self._cell[0] is not None:
            self.lib.tonic_collector_destroy(self._cell[0])
            self._cell[0] = None

    def bind_segment(self, buffers, norm_acc, rows):
        p = _lib.ptr
        _lib.check(self.lib.tonic_collector_bind_segment(
            self.handle, p(buffers['observations']), p(buffers['actions']),
            p(buffers['next_observations']), p(buffers['rewards']), p(buffers['resets']),
            p(buffers['terminations']), p(buffers['log_probs']), p(norm_acc), rows),
            'tonic_collector_bind_segment')

    def begin_rollout(self, actor_params, stream=None):
        """`stream`: the stream whose work the rollout is ordered behind (default: the current one)."""
        _lib.check(self.lib.tonic_collector_begin_rollout(
            self.handle, _lib.ptr(actor_params),
            _lib.current_stream() if stream is None else stream),
            'tonic_collector_begin_rollout')

    def ppo_step(self, row, eps_slot, store_previous):
        status = self._step(self._cell[0], row, eps_slot, store_previous)
        if status != 0:
            _lib.check(status, 'tonic_collector_ppo_step')

    def arm(self, row, eps_slot, store_previous):
        """Leaves the step's command with the environment (see Block.ring); False: not possible now."""
        status = self._arm(self._cell[0], row, eps_slot, store_previous)
        if status < 0:
            _lib.check(status, 'tonic_collector_arm')
        return status == 1

    def claim(self):
        """True: the environment issued the armed step (it is in flight); False: withdrawn."""
        return self._claim(self._cell[0]) == 1

    def wait_actions(self, timeout=float(os.environ.get('TONIC_AMD_COLLECTOR_TIMEOUT', '60'))):
        status = self._wait(self._cell[0], timeout)
        if status != 0:
            _lib.check(status, 'tonic_collector_wait_actions')

    def end_rollout(self, last_row):
        _lib.check(self.lib.tonic_collector_end_rollout(self.handle, last_row,
                                                        _lib.current_stream()),
                   'tonic_collector_end_rollout')
