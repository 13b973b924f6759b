"""``MeanStd`` observation normaliser — API of ``tonic/torch/normalizers/mean_stds.py``.

The reference accumulates ``new_sum`` / ``new_sum_sq`` with a Python loop over worker rows
(mean_stds.py:44-48).  Here the running sums live in HBM (``device_sums``, float32[2*O]) and
are advanced by the ``tonic_segment_store`` kernel in the same sequential float32 order, so
they are bit-identical; ``update()`` downloads the 2*O floats once per learner update and
repeats the reference's host arithmetic (mean_stds.py:50-70: Python-float weights times
float32 arrays) before refreshing the ``_mean`` / ``_std`` parameters used by the kernels.
"""
import numpy as np
import torch


class MeanStd(torch.nn.Module):
    def __init__(self, mean=0, std=1, clip=None, shape=None):
        super().__init__()
        self.mean, self.std, self.clip = mean, std, clip
        self.count = 0
        self.new_sum = 0
        self.new_sum_sq = 0
        self.new_count = 0
        self.eps = 1e-2
        self.device_sums = None
        if shape:
            self.initialize(shape)

    def initialize(self, shape):
        def as_array(value):
            if isinstance(value, (int, float)):
                return np.full(shape, value, np.float32)
            return np.array(value, np.float32)
        self.mean, self.std = as_array(self.mean), as_array(self.std)
        self.mean_sq = np.square(self.mean)
        self._mean = torch.nn.Parameter(torch.as_tensor(self.mean), requires_grad=False)
        self._std = torch.nn.Parameter(torch.as_tensor(self.std), requires_grad=False)

    def attach(self, device):
        """Allocates the HBM accumulators read/written by tonic_segment_store."""
        size = int(np.prod(self.mean.shape))
        self.device_sums = torch.zeros(2 * size, dtype=torch.float32, device=device)
        return self

    def forward(self, val):
        with torch.no_grad():
            val = (val - self._mean) / self._std
            if self.clip is not None:
                val = torch.clamp(val, -self.clip, self.clip)
        return val

    def unnormalize(self, val):
        return val * self._std + self._mean

    def record(self, values):
        """Host-side record (only for callers outside the fused store kernel)."""
        for row in np.asarray(values, np.float32):
            self.new_sum = self.new_sum + row
            self.new_sum_sq = self.new_sum_sq + np.square(row)
            self.new_count += 1

    def note_device_rows(self, rows):
        """The store kernel recorded `rows` more observation rows into device_sums."""
        self.new_count += rows

    def update(self):
        if self.new_count == 0:
            # mean_stds.py:52 divides the integer 0 by the integer 0 here: same error, no NaNs
            raise ZeroDivisionError('MeanStd.update() without any recorded values')
        if self.device_sums is not None:
            from tonic_amd import parallel
            if parallel.exchanging():
                # every rank recorded its own worker shard: merge the running sums
                torch.distributed.all_reduce(self.device_sums)
                self.new_count *= torch.distributed.get_world_size()
            sums = self.device_sums.cpu().numpy()
            size = sums.shape[0] // 2
            self.new_sum = sums[:size].reshape(self.mean.shape).copy()
            self.new_sum_sq = sums[size:].reshape(self.mean.shape).copy()
            self.device_sums.zero_()
        total = self.count + self.new_count
        batch_mean = self.new_sum / self.new_count
        batch_mean_sq = self.new_sum_sq / self.new_count
        w_old, w_new = self.count / total, self.new_count / total
        self.mean = w_old * self.mean + w_new * batch_mean
        self.mean_sq = w_old * self.mean_sq + w_new * batch_mean_sq
        variance = np.maximum(self.mean_sq - np.square(self.mean), 0)
        self.std = np.maximum(np.sqrt(variance), self.eps)
        self.count = total
        self.new_count, self.new_sum, self.new_sum_sq = 0, 0, 0
        self._mean.data.copy_(torch.as_tensor(self.mean, dtype=torch.float32))
        self._std.data.copy_(torch.as_tensor(self.std, dtype=torch.float32))
