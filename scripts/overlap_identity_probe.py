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
import tonic_amd          # noqa: E402
import tonic_amd.torch    # noqa: E402
from tonic_amd import environments   # noqa: E402

O, A, W, T = 17, 6, 256, 1024


def run(overlap, sleeps=True):
    os.environ['TONIC_AMD_CRITIC_OVERLAP'] = '1' if overlap else '0'
    env = environments.SyntheticBatch(W, O, A, max_episode_steps=1000, pool=5)
    env.initialize(seed=3)
    agent = tonic_amd.torch.agents.PPO(replay=tonic_amd.replays.Segment(size=T, batch_iterations=80))
    agent.initialize(env.observation_space, env.action_space, seed=9)
    observations = env.start()
    in_flight, first, trace = 0, None, []
    for t in range(3 * T + 40):
        actions = agent.step(observations, t * W)
        if t in (0, T - 1, T, T + 1, 2 * T - 1, 2 * T, 3 * T - 1, 3 * T + 5):
            trace.append(actions.copy())
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
    return dict(first=first, rows=rows, state=state, in_flight=in_flight, trace=trace)


def compare(a, b, label):
    out = []
    for key in ('first', 'rows'):
        for net, name in ((0, 'actor'), (1, 'critic')):
            x, y = a[key][net], b[key][net]
            bad = np.flatnonzero((x != y).any(axis=1))
            if len(bad):
                out.append(f'{key}/{name}: rows {bad[:5].tolist()}.. ({len(bad)}) '
                           f'first diff {x[bad[0], :3]} vs {y[bad[0], :3]}')
    for i, (x, y) in enumerate(zip(a['trace'], b['trace'])):
        if not np.array_equal(x, y):
            out.append(f'actions of traced step {i} differ')
    for key in a['state']:
        if not np.array_equal(a['state'][key], b['state'][key]):
            out.append(f'{key}: max |diff| {np.abs(a["state"][key] - b["state"][key]).max():.2e}')
    print(label, '->', 'IDENTICAL' if not out else '; '.join(out[:8]), flush=True)


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
