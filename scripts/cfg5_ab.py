"""Developer probe: the collect loop of BASELINE config 5's per-GPU share (AntBullet shapes, 1 280
workers) with the environment-issued step (TONIC_AMD_ARM) and the carried-over observation rows
(TONIC_AMD_CARRY_OVER) switched on and off — one agent (one resident kernel) alive at a time, variants
alternating on one box: microseconds per environment step and the host-loop breakdown.
usage: cfg5_ab.py [rounds] [arm,carry;arm,carry;...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
bench.O, bench.A, bench.W = 28, 8, 1280

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 3
variants = [('0', '0'), ('1', '0'), ('0', '1'), ('1', '1')]
if len(sys.argv) > 2:
    variants = [tuple(v.split(',')) for v in sys.argv[2].split(';')]
times = {v: [] for v in variants}
parts = {}
for r in range(rounds):
    for v in variants:
        os.environ['TONIC_AMD_ARM'], os.environ['TONIC_AMD_CARRY_OVER'] = v
        agent, loop, rollout, out = bench.measure_job(1280, 0, 1, 1, 0, True, device_too=False)
        loop.run(bench.T - agent.replay.index)             # finish the segment: a learner update
        torch.cuda.synchronize()
        for _ in range(3):
            loop.run(64)
            t0 = time.perf_counter()
            loop.run(1024)
            times[v].append((time.perf_counter() - t0) / 1024 * 1e6)
        parts[v] = loop.breakdown(512)
        print('arm %s carry-over %s' % v, file=sys.stderr, flush=True)
        agent.close()
        del agent, loop, rollout
for v in variants:
    print('arm %s carry-over %s: us per environment step' % v, ' '.join(f'{t:.1f}' for t in times[v]),
          '| median', round(float(np.median(times[v])), 2), '|', parts[v])
