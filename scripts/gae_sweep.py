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
# exact chain: the streamed kernel against the lane = column scan (tuning key gae_stream), bit for bit
for (T_, W_) in ((4096, 256), (4096, 1280), (1000, 80), (1000, 77), (63, 5), (65, 16), (129, 32), (129, 33), (4096, 10240)):
    dev = agent.device
    arrays = [torch.randn(T_, W_, device=dev) for _ in range(3)]
    resets = (torch.rand(T_, W_, device=dev) < 1e-2).float()
    terms = resets * (torch.rand(T_, W_, device=dev) < 0.5).float()
    res = {}
    for mode in (0, 1, 3):
        _lib.check(lib.tonic_set_tuning(b'gae_stream', mode), 'tuning')
        outs = [torch.zeros(T_, W_, device=dev) for _ in range(2)]
        stats = torch.zeros(4, device=dev)
        wsg = torch.empty(max(lib.tonic_gae_workspace_bytes(T_, W_, 1), 16), dtype=torch.uint8, device=dev)
        _lib.check(lib.tonic_gae_lambda_returns(p(arrays[0]), p(arrays[1]), p(resets), p(terms), p(arrays[2]), p(outs[0]), p(outs[1]), p(stats), None, T_, W_, 0.99, 0.97, 1, p(wsg), wsg.numel(), stream), 'gae')
        torch.cuda.synchronize()
        res[mode] = (outs[0].cpu(), outs[1].cpu(), stats.cpu())
    print('stream == scan', (T_, W_), [bool(torch.equal(res[0][k], res[m][k])) for m in (1, 3) for k in (0, 1)], res[0][2].tolist()[:2], res[1][2].tolist()[:2])
_lib.check(lib.tonic_set_tuning(b'gae_stream', 1), 'tuning')
