"""Per-tensor error of tonic_ppo_actor_grad_torso against float64 autograd for a list of torsos (debugging aid)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tonic_amd import _lib
lib = _lib.load()


def run(sizes, act, O, A, n):
    rng = np.random.RandomState(1)
    fn = torch.tanh if act == 1 else torch.relu
    arr = (ctypes.c_int32 * len(sizes))(*sizes)
    dims = (O,) + tuple(sizes)
    ps = []
    for fi, fo in zip(dims[:-1], dims[1:]):
        ps += [rng.normal(size=(fo, fi)) / np.sqrt(fi), rng.normal(size=fo) * 0.1]
    ps += [rng.normal(size=(1, A)) * 0.2, rng.normal(size=(A, sizes[-1])) / np.sqrt(sizes[-1]), rng.normal(size=A) * 0.1]
    ps = [p.astype(np.float32) for p in ps]
    flat = np.concatenate([p.reshape(-1) for p in ps])
    obs = rng.standard_normal((n, O)).astype(np.float32)
    actions = np.clip(rng.standard_normal((n, A)), -1, 1).astype(np.float32)
    adv = rng.standard_normal(n).astype(np.float32)
    pa = [torch.tensor(p.astype(np.float64), device='cuda', requires_grad=True) for p in ps]
    x = torch.tensor(obs.astype(np.float64), device='cuda')
    h = x
    for W, b in zip(pa[:-3][0::2], pa[:-3][1::2]):
        h = fn(h @ W.T + b)
    loc = torch.tanh(h @ pa[-2].T + pa[-1])
    dist = torch.distributions.Normal(loc, (torch.nn.functional.softplus(pa[-3]) + 1e-8).clamp(1e-4, 1.0))
    a_t = torch.tensor(actions.astype(np.float64), device='cuda')
    old = (dist.log_prob(a_t).sum(-1).detach().cpu().numpy() + rng.normal(size=n) * 0.1).astype(np.float32)
    ratio = torch.exp(dist.log_prob(a_t).sum(-1) - torch.tensor(old.astype(np.float64), device='cuda'))
    adv_t = torch.tensor(adv.astype(np.float64), device='cuda')
    loss = -torch.min(adv_t * ratio, adv_t * ratio.clamp(0.8, 1.2)).sum()
    grads = torch.autograd.grad(loss, pa)
    P = flat.size
    ws = torch.empty(lib.tonic_ppo_torso_workspace_bytes(n, O, A, 1, len(sizes), arr), dtype=torch.uint8, device='cuda')
    t = lambda v: torch.as_tensor(np.ascontiguousarray(v)).cuda()
    keep = [t(flat), t(obs), t(actions), t(adv), t(np.array([0, 1, 0, 0], np.float32)), t(old)]
    out = torch.zeros(P + 8, device='cuda')
    _lib.check(lib.tonic_ppo_actor_grad_torso(len(sizes), arr, act, *[k.data_ptr() for k in keep], out.data_ptr(),
                                              n, O, A, 0.2, 0.0, None, ws.data_ptr(), ws.numel(), None), 'actor')
    got = out.cpu().numpy()
    at, line = 0, []
    top = max(float(g.abs().max()) for g in grads)
    for g in grads:
        w = g.cpu().numpy().reshape(-1)
        e = np.abs(got[at:at + w.size] - w)
        bad = np.argwhere(e.reshape(g.shape) > 1e-4 * top)
        line.append(f'{tuple(g.shape)}:{e.max() / top:.1e}' + (f' rows {sorted(set(bad[:, 0]))[:6]}.. cols {sorted(set(bad[:, -1]))[:6]}..' if len(bad) else ''))
        at += w.size
    print(sizes, act, O, A, n, ' | '.join(line), flush=True)


for case in [((384, 300), 2, 40, 21, 1037), ((384, 300), 1, 40, 21, 1037), ((128, 300), 2, 40, 21, 1037), ((300, 128), 2, 40, 21, 1037),
             ((64, 44), 2, 40, 21, 1037), ((384, 256), 2, 40, 21, 1037), ((384, 304), 2, 40, 21, 1037), ((384, 300), 2, 40, 8, 1037),
             ((384, 300), 2, 40, 21, 4096)]:
    run(*case)
