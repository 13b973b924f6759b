"""Developer probe: the exact GAE call right after heavy work (clocks up) vs after idling."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tonic_amd import _lib  # noqa: E402

lib, p = _lib.load(), _lib.ptr
T, W = 4096, 256
arrays = [torch.randn(T, W, device='cuda') for _ in range(3)]
resets = (torch.rand(T, W, device='cuda') < 1e-3).float()
terms = resets * (torch.rand(T, W, device='cuda') < 0.5).float()
outs = [torch.empty(T, W, device='cuda') for _ in range(2)]
stats = torch.zeros(4, device='cuda')
ws = torch.empty(max(lib.tonic_gae_workspace_bytes(T, W, 1), 16), dtype=torch.uint8, device='cuda')
big = torch.randn(8192, 8192, device='cuda')


def run():
    _lib.check(lib.tonic_gae_lambda_returns(
        p(arrays[0]), p(arrays[1]), p(resets), p(terms), p(arrays[2]), p(outs[0]), p(outs[1]),
        p(stats), None, T, W, 0.99, 0.97, 1, p(ws), ws.numel(), _lib.current_stream()), 'gae')


def timed(label, warm):
    run()
    torch.cuda.synchronize()
    if warm:
        for _ in range(30):
            big @ big
    else:
        time.sleep(0.5)
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(5):
        run()
    end.record()
    torch.cuda.synchronize()
    print(label, 'us per call', round(start.elapsed_time(end) / 5 * 1e3, 1))


for mode in (1, 0):
    _lib.check(lib.tonic_set_tuning(b'gae_stream', mode), 'tuning')
    timed(f'mode {mode} after 0.5 s idle', False)
    timed(f'mode {mode} after matmuls', True)
    timed(f'mode {mode} after 0.5 s idle', False)
    timed(f'mode {mode} after matmuls', True)
