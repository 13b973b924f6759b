import sys, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, torch
agent = bench.build_agent(steps=8, iterations=1)
from tonic_amd import _lib
lib, p = _lib.load(), _lib.ptr
stream = _lib.current_stream()
def gae_entry(t_steps, workers, chunks):
    dev = agent.device
    arrays = [torch.randn(t_steps, workers, device=dev) for _ in range(3)]
    resets = (torch.rand(t_steps, workers, device=dev) < 1e-3).float()
    terms = resets * (torch.rand(t_steps, workers, device=dev) < 0.5).float()
    outs = [torch.empty(t_steps, workers, device=dev) for _ in range(2)]
    stats = torch.zeros(4, device=dev)
    wsg = torch.empty(max(lib.tonic_gae_workspace_bytes(t_steps, workers, chunks), 16), dtype=torch.uint8, device=dev)
    def run():
        _lib.check(lib.tonic_gae_lambda_returns(p(arrays[0]), p(arrays[1]), p(resets), p(terms), p(arrays[2]), p(outs[0]), p(outs[1]), p(stats), None, t_steps, workers, 0.99, 0.97, chunks, p(wsg), wsg.numel(), stream), 'gae')
    ms = bench.time_events(run, 20)
    gbs = 28.0 * t_steps * workers / (ms * 1e-3) / 1e9
    return dict(T=t_steps, W=workers, chunks=chunks, us=round(ms*1e3, 1), GBs=round(gbs, 1), frac=round(gbs / 8000, 3))
for W in (256, 1280, 4096, 10240, 65536):
    for c in (1, 2):
        print(gae_entry(4096, W, c))
print(gae_entry(1000, 77, 2), gae_entry(4096, 1, 2))
