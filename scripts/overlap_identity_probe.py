"""Developer probe: the critic's chain under a running rollout against the interleaved launches,
several runs of each mode — which rows differ between runs of the SAME mode (non-determinism)
and between the modes (a race)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tonic_amd import _lib as _lib_module   # noqa: E402
if os.environ.get('PROBE_LIBRARY'):
    _lib_module.LIBRARY_PATH = os.environ['PROBE_LIBRARY']
import tonic_amd          # noqa: E402
import tonic_amd.torch    # noqa: E402
from tonic_amd import environments   # noqa: E402

O, A, W, T = (int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (17, 6, 256, 1024)))
ITERATIONS = int(sys.argv[5]) if len(sys.argv) >= 6 else 80
EPISODE = int(sys.argv[6]) if len(sys.argv) >= 7 else 1000


def run(overlap, sleeps=True):
    os.environ['TONIC_AMD_CRITIC_OVERLAP'] = '1' if overlap else '0'
    env = environments.SyntheticBatch(W, O, A, max_episode_steps=EPISODE, pool=7 if EPISODE < 1000 else 5)
    env.initialize(seed=3)
    agent = tonic_amd.torch.agents.PPO(replay=tonic_amd.replays.Segment(size=T, batch_iterations=ITERATIONS))
    agent.initialize(env.observation_space, env.action_space, seed=9)
    observations = env.start()
    in_flight, first, trace = 0, None, []
    snapshots = []
    real_update = agent._update

    def snapshotting_update():
        agent._collector  # (the rollout has ended: end_rollout ran before _update)
        torch.cuda.synchronize()
        snap = {k: v.cpu().numpy().copy() for k, v in agent.replay.buffers.items()
                if k in ('observations', 'actions', 'next_observations', 'rewards', 'resets',
                         'terminations', 'log_probs')}
        snap['norm_sums'] = agent.model.observation_normalizer.device_sums.cpu().numpy().copy()
        snapshots.append(snap)
        real_update()
    agent._update = snapshotting_update
    for t in range(3 * T + 40):
        actions = agent.step(observations, t * W)
        if T <= 64 or t in (0, T - 1, T, T + 1, 2 * T - 1, 2 * T, 3 * T - 1, 3 * T + 5):
            trace.append(actions.copy())
        if t == T - 2:
            kept = {k: v.clone() for k, v in agent.replay.buffers.items()}
        observations, infos = env.step(actions)
        if t >= T and t % 16 == 0:
            pending = getattr(agent, '_critic_pending', None)
            in_flight += (pending is not None and pending['done'] is not None
                          and not pending['done'].query())
        if T <= t < T + 30 or 2 * T <= t < 2 * T + 30:
            if sleeps:
                time.sleep(0.0003)
        agent.update(**infos, steps=t * W)
        if t == 2 * T - 2:
            first = np.array(agent.last_infos)
    torch.cuda.synchronize()
    rows = np.array(agent.last_infos)
    state = {k: v.detach().cpu().numpy().copy() for k, v in agent.model.state_dict().items()}
    agent.close()
    return dict(first=first, rows=rows, state=state, in_flight=in_flight, trace=trace, snapshots=snapshots,
                kept={k: v.cpu().numpy() for k, v in kept.items()})


def compare(a, b, label):
    out = []
    for key in ('first', 'rows'):
        for net, name in ((0, 'actor'), (1, 'critic')):
            x, y = a[key][net], b[key][net]
            bad = np.flatnonzero((x != y).any(axis=1))
            if len(bad):
                out.append(f'{key}/{name}: rows {bad[:5].tolist()}.. ({len(bad)}) '
                           f'first diff {x[bad[0], :3]} vs {y[bad[0], :3]}')
    for key in ('observations', 'actions', 'rewards', 'resets', 'log_probs', 'next_observations'):
        x, y = a['kept'][key][:T - 2], b['kept'][key][:T - 2]
        if not np.array_equal(x, y, equal_nan=True):
            rows_ = np.flatnonzero((x != y).reshape(x.shape[0], -1).any(axis=1))
            out.append(f'Segment {key}: rows {rows_[:6].tolist()} differ')
    for u, (sa, sb) in enumerate(zip(a['snapshots'], b['snapshots'])):
        for key in sa:
            x, y = sa[key], sb[key]
            if not np.array_equal(x, y, equal_nan=True):
                if x.ndim >= 2:
                    bad = np.argwhere((x != y).reshape(x.shape[0], x.shape[1], -1).any(axis=2))
                    out.append(f'rollout {u} Segment {key}: {len(bad)} (row, worker) cells differ, first {bad[:4].tolist()}, '
                               f'workers {sorted(set(bad[:, 1].tolist()))[:8]}')
                else:
                    out.append(f'rollout {u} {key}: max |diff| {np.abs(x - y).max():.3e}')
        if any(o.startswith(f'rollout {u}') for o in out):
            break
    for i, (x, y) in enumerate(zip(a['trace'], b['trace'])):
        if not np.array_equal(x, y):
            out.append(f'actions of traced step {i} differ ({int((x != y).sum())} values, max {np.abs(x - y).max():.2e})')
            break
    for key in a['state']:
        if not np.array_equal(a['state'][key], b['state'][key]):
            out.append(f'{key}: max |diff| {np.abs(a["state"][key] - b["state"][key]).max():.2e}')
    print(label, '->', 'IDENTICAL' if not out else '; '.join(out[:5])[:900], flush=True)


def main():
    runs = {}
    for name, overlap, sleeps in (('plain1', False, True), ('plain2', False, True),
                                  ('over1', True, True), ('over2', True, True),
                                  ('over_nosleep', True, False)):
        runs[name] = run(overlap, sleeps)
        print(name, 'in flight:', runs[name]['in_flight'], flush=True)
    compare(runs['plain1'], runs['plain2'], 'plain vs plain')
    compare(runs['over1'], runs['over2'], 'overlap vs overlap')
    compare(runs['plain1'], runs['over1'], 'plain vs overlap')
    compare(runs['plain1'], runs['over_nosleep'], 'plain vs overlap (no sleeps)')


if __name__ == '__main__':
    main()
