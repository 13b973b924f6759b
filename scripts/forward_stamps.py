"""Developer probe: where one workgroup of the fused off-policy forward spends its time
(wall_clock64 stamps of workgroup (0, 0), 10 ns ticks).  usage: forward_stamps.py [B]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import tonic_amd, tonic_amd.torch as tt
from tonic_amd import _lib
from tonic_amd.environments import Box
B = int(sys.argv[1]) if len(sys.argv) > 1 else 100
O, A, iters, rows = 67, 21, 4, 100000
replay = tonic_amd.replays.Buffer(size=rows, batch_iterations=iters, batch_size=B)
agent = tt.agents.SAC(replay=replay)
agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=0)
replay._allocate(1, O, A)
for k, b in replay.buffers.items():
    b.copy_(torch.randn(b.shape, device='cuda') * (0.0 if k in ('resets', 'terminations') else 1.0))
replay.buffers['discounts'].fill_(0.99)
replay.size = rows
lib = _lib.load()
for _ in range(3):
    agent.enqueue_update(replay.sample_indices(), agent._draw_noise(iters), graph=False)
torch.cuda.synchronize()
stamps = torch.zeros(8 * 16 + 8 * 8, dtype=torch.int64, device='cuda')
lib.tonic_debug_forward_stamps(stamps.data_ptr())
agent.enqueue_update(replay.sample_indices(), agent._draw_noise(iters), graph=False)   # 16 forwards
torch.cuda.synchronize()
lib.tonic_debug_forward_stamps(None)
raw = stamps.cpu().numpy()
both, tn = raw[:128].reshape(8, 16), raw[128:].reshape(8, 8)
s, cycles = both[:, :8], both[:, 8:]
names = ['target actor', 'four critics', 'online actor', 'two critics'] * 2
labels = ['inputs->LDS', 'layer 1', 'epilogue 1', 'layer 2', 'epilogue 2', 'heads/out', 'tail']
print('B', B)
for i in range(8):
    t = s[i]
    if not t.any():
        continue
    d = [(t[j + 1] - t[j]) * 0.01 if t[j + 1] else 0.0 for j in range(7)]
    last = max(j for j in range(8) if t[j])
    print(f'{names[i]:13s}', ' '.join(f'{l} {x:5.2f}' for l, x in zip(labels, d)),
          f'| total {(t[last] - t[0]) * 0.01:5.2f} us',
          f'| shader clock {(cycles[i][last] - cycles[i][0]) / max((t[last] - t[0]) * 10.0, 1):5.2f} GHz')
for i in range(8):
    t = tn[i]
    if not t.any():
        continue
    print('weight gradients', ' '.join(f'{l} {(t[j + 1] - t[j]) * 0.01:5.2f}' for j, l in enumerate(
        ['requests+consts', 'main loop', 'exchange', 'fold+operands', 'adam+stores'])),
        f'| total {(t[5] - t[0]) * 0.01:5.2f} us')
