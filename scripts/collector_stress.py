"""Stress of the collector's completion protocol: the actions the host reads right after
wait_actions must be the ones the kernel stored in the Segment row, for many steps — and the
observation rows the kernel stored must be the ones the host wrote (or pushed: transport 3) for that
very step, never those of a step before."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(W=256, O=17, A=6, steps=20000, transport=0):
    import torch
    from tonic_amd.collector import Block, Collector
    block = Block(W, O, A)
    collector = Collector(block, transport)
    T = 512
    seg = {k: torch.zeros(T, W, *tail, device='cuda') for k, tail in (
        ('observations', (O,)), ('actions', (A,)), ('next_observations', (O,)), ('rewards', ()),
        ('resets', ()), ('terminations', ()), ('log_probs', ()))}
    sums = torch.zeros(2 * O, device='cuda')
    collector.bind_segment(seg, sums, T)
    flat = torch.randn(64 * O + 64 + 4096 + 64 + A + 64 * A + A, device='cuda') * 0.3
    torch.cuda.synchronize()
    collector.begin_rollout(flat)
    rng = np.random.RandomState(0)
    pool = rng.standard_normal((64, W, O)).astype(np.float32)
    eps = rng.standard_normal((64, W, A)).astype(np.float32)
    host_actions = np.zeros((T, W, A), np.float32)
    host_rewards = np.zeros((T, W), np.float32)
    host_observations = np.zeros((T, W, O), np.float32)
    bad = 0
    for i in range(steps):
        row = i % T
        block.observations[:] = pool[i % 64]
        block.observations[:, 0] = i                      # (no two steps alike)
        host_observations[row] = block.observations
        block.eps[i & 1][:] = eps[(i * 7) % 64]
        collector.ppo_step(row, i & 1, row > 0)
        collector.wait_actions()
        host_actions[row] = block.actions
        block.rewards[:] = i + np.arange(W)              # the outcome of this step
        host_rewards[row] = block.rewards
        block.next_observations[:] = pool[(i + 1) % 64]
        if row == T - 1:
            collector.end_rollout(T - 1)
            torch.cuda.synchronize()
            bad += int((seg['actions'].cpu().numpy() != host_actions).sum())
            bad += int((seg['rewards'].cpu().numpy() != host_rewards).sum())
            bad += int((seg['observations'].cpu().numpy() != host_observations).sum())
            collector.begin_rollout(flat)
    print(f'transport {transport} W {W}: {steps} steps, mismatching elements: {bad}')
    return bad


if __name__ == '__main__':
    total = 0
    for W in (256, 6, 1280):
        total += main(W=W)
    total += main(transport=1, steps=4000)
    for W in (256, 6, 1280):
        total += main(W=W, transport=2)
    for W in (256, 6):
        total += main(W=W, transport=3)
    sys.exit(1 if total else 0)
