"""Developer probe: the exact (one chunk) GAE call at the BASELINE sizes, for rocprofv3."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from tonic_amd import _lib  # noqa: E402

lib, p = _lib.load(), _lib.ptr
for T, W in ((4096, 256), (4096, 1280)):
    arrays = [torch.randn(T, W, device='cuda') for _ in range(3)]
    resets = (torch.rand(T, W, device='cuda') < 1e-3).float()
    terms = resets * (torch.rand(T, W, device='cuda') < 0.5).float()
    outs = [torch.empty(T, W, device='cuda') for _ in range(2)]
    stats = torch.zeros(4, device='cuda')
    ws = torch.empty(max(lib.tonic_gae_workspace_bytes(T, W, 1), 16), dtype=torch.uint8, device='cuda')

    def run():
        _lib.check(lib.tonic_gae_lambda_returns(
            p(arrays[0]), p(arrays[1]), p(resets), p(terms), p(arrays[2]), p(outs[0]), p(outs[1]),
            p(stats), None, T, W, 0.99, 0.97, 1, p(ws), ws.numel(), _lib.current_stream()), 'gae')
    for mode in (1, 3):          # 2: developer probe, helpers idle (the chain's time alone; wrong results)
        _lib.check(lib.tonic_set_tuning(b'gae_stream', mode), 'tuning')
        print(T, W, 'mode', mode, 'us per call', round(bench.time_events(run, 20) * 1e3, 1))
    _lib.check(lib.tonic_set_tuning(b'gae_stream', 1), 'tuning')
