"""HBM-resident on-policy ``Segment`` — API of ``tonic/replays/segments.py``.

Layout in HBM (float32, like the reference which stores even the boolean flags as float32,
segments.py:33): ``observations, next_observations [T, W, O]``, ``actions [T, W, A]``,
``rewards, resets, terminations, log_probs, values, next_values, returns, advantages
[T, W]`` — time-major so that (a) one ``store`` call writes one contiguous row per buffer,
(b) the lambda-return scan walks T with lane = worker (coalesced), (c) the flattened batch
``[T*W, ...]`` the learner kernels read is a free view (``flatten_batch``, utils.py:22-25).
176 B per transition at O=17, A=6 (+16 B for the derived arrays) = 201 MB at W=256.

``store`` goes through ``tonic_segment_store`` (one launch per time step, which also advances
the observation normaliser's running sums); ``compute_returns`` through
``tonic_gae_lambda_returns`` which also produces the raw advantages and their mean / std;
the normalisation itself, (adv - mean) / std (segments.py:45), is applied in-register by the
actor kernel, so ``get_full('advantages')`` materialises it only when a caller asks.
"""
import os

import numpy as np
import torch

from tonic_amd import _lib, parallel

STORED_KEYS = ('observations', 'actions', 'next_observations', 'rewards', 'resets',
               'terminations', 'log_probs')


LEARNER_KEYS = ('observations', 'actions', 'advantages', 'log_probs', 'returns')


def flatten_batch(values):
    shape = values.shape
    return values.reshape((int(np.prod(shape[:2], dtype=int)),) + tuple(shape[2:]))


class Segment:
    def __init__(self, size=4096, batch_iterations=80, batch_size=None, discount_factor=0.99,
                 trace_decay=0.97, gae_chunks=None):
        """`gae_chunks` (not in the reference): 0 lets the library choose — below W = 65 536 the
        one-pass form (T cut into 128-row segments, carries composed as affine maps: returns within
        ~1e-6 relative of replays/utils.py:4-19, inside the 1e-5 the north star allows; 24 us instead
        of 104 us and 1.0 x instead of 1.73 x the algorithmic traffic at T = 4096, W = 256), the single
        chain above it; 1 runs ONE chain per worker in the reference's float32 operation order (returns
        bit-identical to the reference).  None (default): 0, or 1 with TONIC_AMD_GAE_EXACT=1."""
        if gae_chunks is None:
            gae_chunks = 1 if os.environ.get('TONIC_AMD_GAE_EXACT', '0') == '1' else 0
        self.max_size = size
        self.batch_iterations = batch_iterations
        self.batch_size = batch_size
        self.discount_factor = discount_factor
        self.trace_decay = trace_decay
        self.gae_chunks = gae_chunks

    def initialize(self, seed=None, device=None):
        self.np_random = np.random.RandomState(seed)
        self.device = torch.device(device if device is not None else 'cuda')
        self.buffers = None
        self.index = 0
        self.lib = _lib.load()

    def ready(self):
        return self.index == self.max_size

    def _allocate(self, num_workers, observation_size, action_size):
        T, W = self.max_size, num_workers
        self.num_workers = W
        self.observation_size, self.action_size = observation_size, action_size

        def new(*shape):
            return torch.zeros(shape, dtype=torch.float32, device=self.device)
        self.buffers = dict(
            observations=new(T, W, observation_size), actions=new(T, W, action_size),
            next_observations=new(T, W, observation_size), rewards=new(T, W), resets=new(T, W),
            terminations=new(T, W), log_probs=new(T, W), values=new(T, W),
            next_values=new(T, W), returns=new(T, W), advantages=new(T, W))
        self.adv_stats = torch.zeros(4, dtype=torch.float32, device=self.device)
        self.adv_moments = torch.zeros(5, dtype=torch.float64, device=self.device)
        nbytes = self.lib.tonic_gae_workspace_bytes(T, W, self.gae_chunks)
        self.gae_workspace = torch.empty(max(nbytes, 16), dtype=torch.uint8, device=self.device)

    def store(self, normalizer=None, **kwargs):
        """Writes one time row.  Values must be float32 device tensors of shape [W, ...]
        (the agent stages host data through its pinned buffers first)."""
        if self.buffers is None:
            self._allocate(kwargs['observations'].shape[0], kwargs['observations'].shape[1],
                           kwargs['actions'].shape[1])
        if self.index >= self.max_size:
            raise IndexError('Segment is full: call get()/get_full() before storing again')
        b = self.buffers
        sums = normalizer.device_sums if normalizer is not None else None
        p = _lib.ptr
        _lib.check(self.lib.tonic_segment_store(
            p(b['observations']), p(b['actions']), p(b['next_observations']), p(b['rewards']),
            p(b['resets']), p(b['terminations']), p(b['log_probs']),
            p(kwargs['observations']), p(kwargs['actions']), p(kwargs['next_observations']),
            p(kwargs['rewards']), p(kwargs['resets']), p(kwargs['terminations']),
            p(kwargs['log_probs']), p(sums), self.index, self.num_workers,
            self.observation_size, self.action_size, _lib.current_stream()),
            'tonic_segment_store')
        if normalizer is not None:
            normalizer.note_device_rows(self.num_workers)
        self.index += 1

    def compute_returns(self, values, next_values):
        """segments.py:67-78 + the raw-advantage half of get_full (segments.py:41-46)."""
        b = self.buffers
        shape = b['rewards'].shape
        for key, given in (('values', values), ('next_values', next_values)):
            # the reference passes NumPy arrays (segments.py:67-70); device tensors that already
            # ARE the segment's buffers (the fused path) cost nothing
            if not (torch.is_tensor(given) and given.is_cuda
                    and given.data_ptr() == b[key].data_ptr()):
                b[key].copy_(torch.as_tensor(np.asarray(given.cpu() if torch.is_tensor(given)
                                                        else given), dtype=torch.float32)
                             .reshape(shape))
        p = _lib.ptr
        _lib.check(self.lib.tonic_gae_lambda_returns(
            p(b['next_values']), p(b['rewards']), p(b['resets']), p(b['terminations']),
            p(b['values']), p(b['returns']), p(b['advantages']), p(self.adv_stats),
            p(self.adv_moments), shape[0], shape[1], float(self.discount_factor),
            float(self.trace_decay), self.gae_chunks, p(self.gae_workspace),
            self.gae_workspace.numel(), _lib.current_stream()), 'tonic_gae_lambda_returns')
        if parallel.exchanging():
            # global advantage statistics over every rank's worker shard
            m = self.adv_moments
            torch.distributed.all_reduce(m[0:2])
            torch.distributed.all_reduce(m[4:5])
            torch.distributed.all_reduce(m[2:4], op=torch.distributed.ReduceOp.MAX)
            _lib.check(self.lib.tonic_advantage_stats_from_moments(
                p(m), p(self.adv_stats), _lib.current_stream()),
                'tonic_advantage_stats_from_moments')

    def get_full(self, *keys):
        """Flattened [T*W, ...] device views.  'advantages' is returned NORMALISED like the
        reference (materialised with stock torch ops; the fused learner path never asks for
        it and reads the raw advantages + adv_stats instead)."""
        self.index = 0
        out = {}
        for k in keys:
            if k == 'advantages':
                mean, std, _, flag = self.adv_stats.tolist()
                adv = self.buffers['advantages']
                out[k] = flatten_batch((adv - mean) / std if flag else adv)
            else:
                out[k] = flatten_batch(self.buffers[k])
        return out

    def updates_per_get(self):
        """Number of batches one ``get`` / ``learner_batches`` pass yields."""
        if self.batch_size is None:
            return self.batch_iterations
        n = self.max_size * self.num_workers
        return self.batch_iterations * ((n + self.batch_size - 1) // self.batch_size)

    def learner_batches(self):
        """What the fused learner consumes: tuples ``(observations, actions, RAW advantages,
        log_probs, returns)`` of device tensors — the full flattened segment ``batch_iterations``
        times (segments.py:55-57), or shuffled minibatches (segments.py:58-65).

        Minibatch mode keeps the reference's index stream (one persistent ``arange`` shuffled
        in place by the replay's ``RandomState`` once per epoch) on the host, uploads the epoch's
        permutation and gathers the whole epoch with ONE ``tonic_segment_gather`` launch; the
        minibatches are then contiguous slices of that epoch image."""
        self.index = 0
        b = self.buffers
        full = tuple(flatten_batch(b[k]) for k in LEARNER_KEYS)
        if self.batch_size is None:
            for _ in range(self.batch_iterations):
                yield full
            return
        n, bs = full[0].shape[0], int(self.batch_size)
        if getattr(self, '_epoch', None) is None or self._epoch[0].shape[0] != n:
            self._epoch = tuple(torch.empty_like(v) for v in full)
            self._epoch_indices = torch.empty(n, dtype=torch.int64, device=self.device)
        order = np.arange(n)                                   # segments.py:59
        p = _lib.ptr
        for _ in range(self.batch_iterations):
            self.np_random.shuffle(order)                      # segments.py:61
            self._epoch_indices.copy_(torch.from_numpy(order))   # blocking H2D: `order` is reused
            _lib.check(self.lib.tonic_segment_gather(
                p(self._epoch_indices), *[p(v) for v in full], *[p(v) for v in self._epoch],
                n, n, self.observation_size, self.action_size, _lib.current_stream()),
                'tonic_segment_gather')
            for start in range(0, n, bs):                      # segments.py:62-65, ragged tail
                yield tuple(v[start:start + bs] for v in self._epoch)

    def get(self, *keys):
        """Drop-in generator (segments.py:49-65).  'advantages' come out normalised."""
        if self.batch_size is None:
            batch = self.get_full(*keys)
            for _ in range(self.batch_iterations):
                yield batch
            return
        mean, std, _, flag = self.adv_stats.tolist()
        extra = [k for k in keys if k not in LEARNER_KEYS]      # served by torch indexing
        for parts in self.learner_batches():
            batch = dict(zip(LEARNER_KEYS, parts))
            if 'advantages' in keys and flag:
                batch['advantages'] = (batch['advantages'] - mean) / std      # segments.py:45
            if extra:
                start = parts[0].data_ptr() - self._epoch[0].data_ptr()
                first = start // (4 * self.observation_size)
                rows = self._epoch_indices[first:first + parts[0].shape[0]]
                for k in extra:
                    batch[k] = flatten_batch(self.buffers[k])[rows]
            yield {k: batch[k] for k in keys}
