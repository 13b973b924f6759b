"""Developer probe: are the fused grad launches bit-reproducible at every width?  Runs each
(kernel, variant, n, width) several times and reports distinct results, where they differ
(which tensor of the flat block) and by how much."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tonic_amd import _lib   # noqa: E402

if os.environ.get('PROBE_LIBRARY'):            # a variant build of the library (developer)
    _lib.LIBRARY_PATH = os.environ['PROBE_LIBRARY']
REPS = int(os.environ.get('PROBE_REPS', '6'))
QUICK = os.environ.get('PROBE_QUICK') == '1'

lib, p = _lib.load(), _lib.ptr
O, A = 17, 6
rng = np.random.RandomState(4)


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x, dtype=np.float32)).cuda()


def names(actor):
    out = [('W1', 64 * O), ('b1', 64), ('W2', 4096), ('b2', 64)]
    out += [('log_scale', A), ('W3', 64 * A), ('b3', A)] if actor else [('W3', 64), ('b3', 1)]
    out.append(('stats', 8))
    return out


def where(idx, actor):
    base = 0
    for name, count in names(actor):
        if idx < base + count:
            return f'{name}[{idx - base}]'
        base += count
    return str(idx)


def main():
    cparams = np.concatenate([(rng.normal(size=s) * c).reshape(-1) for s, c in (
        ((64, O), 0.3), ((64,), 0.1), ((64, 64), 0.15), ((64,), 0.1), ((1, 64), 0.3), ((1,), 1.0))])
    aparams = np.concatenate([(rng.normal(size=s) * c).reshape(-1) for s, c in (
        ((64, O), 0.3), ((64,), 0.1), ((64, 64), 0.15), ((64,), 0.1), ((1, A), 0.0),
        ((A, 64), 0.1), ((A,), 0.1))])
    for n in ((262144, 1048576) if QUICK else (300000, 262144, 1048576)):
        obs = dev(rng.standard_normal((n, O)))
        ret = dev(rng.standard_normal(n))
        act = dev(np.clip(rng.standard_normal((n, A)), -1, 1))
        adv = dev(rng.standard_normal(n))
        lp = dev(-6 + rng.standard_normal(n) * 0.2)
        stats = dev(np.array([0, 1, 0, 0]))
        mean, std = dev(np.zeros(O)), dev(np.ones(O))
        dc, da = dev(cparams), dev(aparams)
        ws = torch.empty(lib.tonic_ppo_workspace_bytes(n, O, A, 1), dtype=torch.uint8, device='cuda')
        Pc, Pa = lib.tonic_v_critic_param_count(O), lib.tonic_ppo_actor_param_count(O, A)
        for variant in ((3,) if QUICK else (3, 1)):
            _lib.check(lib.tonic_set_tuning(b'grad_variant', variant), 'tuning')
            for actor in (False, True):
                for width in ((0, 219, 64, 8) if QUICK else (0, 219, 64, 8, 1)):
                    if width == 1 and n > 300000:
                        continue
                    outs = []
                    for rep in range(REPS):
                        out = torch.zeros((Pa if actor else Pc) + 8, device='cuda')
                        if actor:
                            _lib.check(lib.tonic_ppo_actor_grad(
                                p(da), p(obs), p(act), p(adv), p(stats), p(lp), p(out), n, O, A, 0.2,
                                0.0, None, width, p(ws), ws.numel(), None), 'actor')
                        else:
                            _lib.check(lib.tonic_value_regression_grad(
                                p(dc), p(mean), p(std), 0.0, p(obs), p(ret), p(out), n, O, width,
                                p(ws), ws.numel(), None), 'critic')
                        torch.cuda.synchronize()
                        outs.append(out.cpu().numpy())
                    distinct = {o.tobytes() for o in outs}
                    line = (f'n={n:8d} variant={variant} {"actor " if actor else "critic"} '
                            f'width={width:3d}: {len(distinct)} distinct of {len(outs)}')
                    if len(distinct) > 1:
                        ref = outs[0]
                        other = next(o for o in outs if o.tobytes() != ref.tobytes())
                        bad = np.flatnonzero(other != ref)
                        rel = np.abs(other - ref)[bad] / max(np.abs(ref).max(), 1e-30)
                        line += (f'; {len(bad)} elements differ, max rel-to-max {rel.max():.2e}, first '
                                 + ', '.join(where(int(b), actor) for b in bad[:6]))
                    print(line, flush=True)
    _lib.check(lib.tonic_set_tuning(b'grad_variant', -1), 'tuning')


if __name__ == '__main__':
    main()
