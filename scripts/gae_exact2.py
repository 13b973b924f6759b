"""Developer probe: exact GAE call with no resets at all vs resets everywhere (fast / slow chain)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tonic_amd import _lib  # noqa: E402

lib, p = _lib.load(), _lib.ptr
T, W = 4096, 256
for label, prob in (('no resets', 0.0), ('resets 1e-3', 1e-3), ('all resets', 1.1)):
    arrays = [torch.randn(T, W, device='cuda') for _ in range(3)]
    resets = (torch.rand(T, W, device='cuda') < prob).float()
    terms = resets * (torch.rand(T, W, device='cuda') < 0.5).float()
    outs = [torch.empty(T, W, device='cuda') for _ in range(2)]
    stats = torch.zeros(4, device='cuda')
    ws = torch.empty(max(lib.tonic_gae_workspace_bytes(T, W, 1), 16), dtype=torch.uint8, device='cuda')

    def run():
        _lib.check(lib.tonic_gae_lambda_returns(
            p(arrays[0]), p(arrays[1]), p(resets), p(terms), p(arrays[2]), p(outs[0]), p(outs[1]),
            p(stats), None, T, W, 0.99, 0.97, 1, p(ws), ws.numel(), _lib.current_stream()), 'gae')
    for mode in (1, 2, 4):
        _lib.check(lib.tonic_set_tuning(b'gae_stream', mode), 'tuning')
        print(label, 'mode', mode, 'us per call', round(bench.time_events(run, 20) * 1e3, 1))
