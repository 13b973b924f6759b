"""HBM-resident off-policy ``Buffer`` — API of ``tonic/replays/buffers.py``.

Circular float32 store ``[max_size = size // W, W, ...]`` for ``observations, actions,
next_observations, rewards, resets, terminations, discounts``, NaN-initialised like the
reference (buffers.py:45), 936 B per transition at O=111, A=8 (0.94 GB for 1 M transitions).
``store`` is one ``tonic_buffer_store`` launch (row write, ``discounts = float32(1 -
terminations) * discount_factor``, observation-normaliser record) plus, for ``return_steps > 1``,
one ``tonic_buffer_accumulate_n_steps`` launch (buffers.py:58-79, bit-exact); ``get`` draws the indices
with the host ``RandomState`` exactly like buffers.py:86 (bit-exact stream) and gathers the
batch with ``tonic_buffer_gather`` (one wavefront per sampled transition,
``rows = idx // W``, ``cols = idx % W``).
"""
import numpy as np
import torch

from tonic_amd import _lib, parallel

KEYS = ('observations', 'actions', 'next_observations', 'rewards', 'resets', 'terminations',
        'discounts')
BATCH_KEYS = ('observations', 'actions', 'next_observations', 'rewards', 'discounts')


class Buffer:
    def __init__(self, size=int(1e6), return_steps=1, batch_iterations=50, batch_size=100,
                 discount_factor=0.99, steps_before_batches=int(1e4), steps_between_batches=50):
        self.full_max_size = size
        self.return_steps = return_steps
        self.batch_iterations = batch_iterations
        self.batch_size = batch_size
        self.discount_factor = discount_factor
        self.steps_before_batches = steps_before_batches
        self.steps_between_batches = steps_between_batches

    def initialize(self, seed=None, device=None):
        self.np_random = np.random.RandomState(seed)
        self.device = torch.device(device if device is not None else 'cuda')
        self.buffers = None
        self.index = 0
        self.size = 0
        self.last_steps = 0
        self.lib = _lib.load()

    def ready(self, steps):
        if steps < self.steps_before_batches:
            return False
        return (steps - self.last_steps) >= self.steps_between_batches

    def _allocate(self, num_workers, observation_size, action_size):
        # Multi-GPU: this rank holds a contiguous shard of the GLOBAL worker axis (equal shards);
        # capacity and the index stream are those of the global [max_size, W_global] buffer.
        self.num_workers = num_workers
        self.rank, self.world = parallel.rank(), parallel.world_size()
        self.global_workers = num_workers * self.world
        self.max_size = self.full_max_size // self.global_workers
        self.observation_size, self.action_size = observation_size, action_size
        R, W = self.max_size, num_workers

        def new(*shape):
            return torch.full(shape, float('nan'), dtype=torch.float32, device=self.device)
        self.buffers = dict(
            observations=new(R, W, observation_size), actions=new(R, W, action_size),
            next_observations=new(R, W, observation_size), rewards=new(R, W), resets=new(R, W),
            terminations=new(R, W), discounts=new(R, W))
        B = self.batch_size
        self.batch = dict(
            observations=torch.empty(B, observation_size, device=self.device),
            actions=torch.empty(B, action_size, device=self.device),
            next_observations=torch.empty(B, observation_size, device=self.device),
            rewards=torch.empty(B, device=self.device), discounts=torch.empty(B, device=self.device))

    def store(self, normalizer=None, **kwargs):
        """One time row from float32 device tensors [W, ...] (buffers.py:33-56)."""
        if self.buffers is None:
            self._allocate(kwargs['observations'].shape[0], kwargs['observations'].shape[1],
                           kwargs['actions'].shape[1])
        b, p = self.buffers, _lib.ptr
        sums = normalizer.device_sums if normalizer is not None else None
        _lib.check(self.lib.tonic_buffer_store(
            p(b['observations']), p(b['actions']), p(b['next_observations']), p(b['rewards']),
            p(b['resets']), p(b['terminations']), p(b['discounts']), p(kwargs['observations']),
            p(kwargs['actions']), p(kwargs['next_observations']), p(kwargs['rewards']),
            p(kwargs['resets']), p(kwargs['terminations']), p(sums), self.index,
            self.num_workers, self.observation_size, self.action_size,
            float(self.discount_factor), _lib.current_stream()), 'tonic_buffer_store')
        if normalizer is not None:
            normalizer.note_device_rows(self.num_workers)
        if self.return_steps > 1:                                    # buffers.py:52-53
            _lib.check(self.lib.tonic_buffer_accumulate_n_steps(
                p(b['next_observations']), p(b['rewards']), p(b['discounts']), p(b['resets']),
                p(kwargs['next_observations']), p(kwargs['rewards']), p(kwargs['terminations']),
                self.index, self.size, self.max_size, self.num_workers, self.observation_size,
                self.return_steps, float(self.discount_factor), _lib.current_stream()),
                'tonic_buffer_accumulate_n_steps')
        self.index = (self.index + 1) % self.max_size
        self.size = min(self.size + 1, self.max_size)

    def reserve_row(self, normalizer=None):
        """The host half of `store` for a transition whose device write is issued later (the off-policy agents'
        acting launch carries it: tonic_collector_q_act): the row it will occupy, the circular index and the size
        advanced (buffers.py:54-56), the normaliser's record count noted.  Needs an allocated Buffer and
        return_steps == 1 (the n-step accumulation reads the stored row right away)."""
        assert self.buffers is not None and self.return_steps == 1
        row = self.index
        if normalizer is not None:
            normalizer.note_device_rows(self.num_workers)
        self.index = (self.index + 1) % self.max_size
        self.size = min(self.size + 1, self.max_size)
        return row

    def store_at(self, row, normalizer=None, **kwargs):
        """The device half of a reserved transition on its own (tonic_buffer_store into `row`): what the acting
        launch would have carried, when no acting launch follows (an update is due first, the policy sits out)."""
        b, p = self.buffers, _lib.ptr
        sums = normalizer.device_sums if normalizer is not None else None
        _lib.check(self.lib.tonic_buffer_store(
            p(b['observations']), p(b['actions']), p(b['next_observations']), p(b['rewards']),
            p(b['resets']), p(b['terminations']), p(b['discounts']), p(kwargs['observations']),
            p(kwargs['actions']), p(kwargs['next_observations']), p(kwargs['rewards']),
            p(kwargs['resets']), p(kwargs['terminations']), p(sums), row,
            self.num_workers, self.observation_size, self.action_size,
            float(self.discount_factor), _lib.current_stream()), 'tonic_buffer_store')

    def store_arguments(self, row, observations, normalizer=None):
        """tonic_q_store_t for `row` (see reserve_row): the Buffer's arrays, the observation rows' device copy."""
        b, p = self.buffers, _lib.ptr
        sums = normalizer.device_sums if normalizer is not None else None
        return _lib.QStore(p(b['observations']), p(b['actions']), p(b['next_observations']), p(b['rewards']),
                           p(b['resets']), p(b['terminations']), p(b['discounts']), p(observations), p(sums),
                           row, float(self.discount_factor))

    def sample_indices(self, iterations=None):
        """The index stream of `iterations` successive Buffer.get draws (buffers.py:85-86)."""
        total = self.size * self.global_workers
        count = self.batch_iterations if iterations is None else iterations
        return np.stack([self.np_random.randint(total, size=self.batch_size)
                         for _ in range(count)])

    def shard_indices(self, indices):
        """Splits GLOBAL flat indices [iterations, B] (rows = idx // W_global, cols = idx %
        W_global, buffers.py:87-88) into this rank's part: returns (local flat indices into the
        [max_size, W_local] shard, positions inside the global batch, counts), the first two
        zero-padded to B per iteration.  Every rank draws the same global stream (same seed), so
        the union over ranks is exactly the single-process batch."""
        iterations, B = indices.shape
        rows, cols = indices // self.global_workers, indices % self.global_workers
        mine = (cols // self.num_workers) == self.rank
        local = np.zeros((iterations, B), np.int64)
        positions = np.zeros((iterations, B), np.int64)
        counts = mine.sum(axis=1)
        for it in range(iterations):
            pos = np.flatnonzero(mine[it])
            positions[it, :len(pos)] = pos
            local[it, :len(pos)] = (rows[it, pos] * self.num_workers
                                    + cols[it, pos] - self.rank * self.num_workers)
        return local, positions, counts

    def gather(self, device_indices, out=None):
        """Gathers one batch (int64 device indices [B]) into `out` (default: the reusable batch)."""
        out = out or self.batch
        b, p = self.buffers, _lib.ptr
        _lib.check(self.lib.tonic_buffer_gather(
            p(device_indices), p(b['observations']), p(b['actions']), p(b['next_observations']),
            p(b['rewards']), p(b['discounts']), p(out['observations']), p(out['actions']),
            p(out['next_observations']), p(out['rewards']), p(out['discounts']),
            self.num_workers, device_indices.shape[0], self.observation_size, self.action_size,
            _lib.current_stream()), 'tonic_buffer_gather')
        count = device_indices.shape[0]
        if count != out['rewards'].shape[0]:
            return {k: v[:count] for k, v in out.items()}
        return out

    def gather_many(self, device_indices):
        """All batches of one learner update in ONE launch: int64 device indices [iterations, B] ->
        {key: [iterations, B, ...]} (the store does not change while an update runs, so the
        gathers of its iterations need not be interleaved with them)."""
        iterations, count = device_indices.shape
        store = getattr(self, '_many', None)
        if store is None or store['rewards'].shape != (iterations, count):
            store = self._many = {
                k: torch.empty((iterations, count) + tuple(v.shape[1:]), dtype=v.dtype,
                               device=self.device) for k, v in self.batch.items()}
        b, p = self.buffers, _lib.ptr
        _lib.check(self.lib.tonic_buffer_gather(
            p(device_indices), p(b['observations']), p(b['actions']), p(b['next_observations']),
            p(b['rewards']), p(b['discounts']), p(store['observations']), p(store['actions']),
            p(store['next_observations']), p(store['rewards']), p(store['discounts']),
            self.num_workers, iterations * count, self.observation_size, self.action_size,
            _lib.current_stream()), 'tonic_buffer_gather')
        return store

    def get(self, *keys, steps):
        """Generator form of the reference API (buffers.py:81-91): yields device-tensor batches.
        With several ranks every rank draws the same global index stream and yields ITS part of
        each batch (the rows whose worker column it owns, in batch order; possibly none) — the
        fused learner reduces gradient sums over ranks, so the union is the reference's batch."""
        for _ in range(self.batch_iterations):
            indices = self.sample_indices(1)
            if self.world > 1:                 # this rank's part of the global batch
                local, _, counts = self.shard_indices(indices)
                indices = local[:, :counts[0]]
            count = indices.shape[1]
            out = {k: torch.empty((count,) + tuple(v.shape[1:]), dtype=v.dtype, device=self.device)
                   for k, v in self.batch.items()}
            if count > 0:
                device_indices = torch.as_tensor(indices[0], device=self.device)
                out = self.gather(device_indices, out)
            yield {k: out[k] for k in keys}
        self.last_steps = steps
