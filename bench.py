"""Headline benchmark: PPO rollout-collect + learner-update on synthetic HalfCheetah shapes.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong]

One "step" = one full pass of the hot path over one Segment of synthetic input, THROUGH the
drop-in API with the host in the loop: T=4096 environment steps x W workers per GPU, each one
`agent.step(observations)` (pinned-host collector: one fused launch = policy forward + sample +
log-prob + Segment store + MeanStd.record, actions back in the shared block),
`environment.step(actions)` (vectorised synthetic simulator writing into the same block) and
`agent.update(**infos)`; the T-th update runs ONE learner update (2 critic forwards, the GAE
scan, 80 x [actor fwd+loss+bwd, Adam, critic fwd+bwd, Adam] with the device-side KL early
stop), the statistics read-back and the normaliser update.
`value` = env steps/s of the whole job = (N_gpus * T * W_per_gpu * K) / time — PCIe and the
Python loop included.  `device_resident` repeats the measurement with the rollout inputs
pre-generated in HBM and no host in the loop (round 1's `value`).

Extra keys on the same JSON line: `roofline` (dominant kernel = fused actor forward+backward,
MFMA bound: `frac` against the ceiling of the kernel's own fp32 / fp16 instruction mix, the fp32-MFMA
figure beside it as `fp32_equivalent`), `roofline_gae` (HBM-bound scan, at the config size and at a bandwidth-bound
sweep), `cpu_baseline` (oracle/torch_port.py = the reference's torch-CPU path timed on this
box's cores on a bounded sample), `learner_updates_per_sec`, `host_loop` (where an environment
step goes), `strong_scaling` (N > 1: the metric's 256 workers split over the ranks).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

O, A, W, T, ITERATIONS = 17, 6, 256, 4096, 80          # cfg 2: HalfCheetah-v3, parallel=256
ACTOR_FLOP_PER_SAMPLE = 31232                          # fwd 11136 + bwd 20096
CRITIC_FLOP_PER_SAMPLE = 29312
FP32_MFMA_PEAK_TFLOPS = 157.3                          # MI355X_MICROARCH.md chip table
HBM_PEAK_GBS = 8000.0


def build_agent(steps=T, iterations=ITERATIONS, seed=0):
    import tonic_amd
    import tonic_amd.torch
    from tonic_amd.environments import Box
    agent = tonic_amd.torch.agents.PPO(
        replay=tonic_amd.replays.Segment(size=steps, batch_iterations=iterations))
    agent.initialize(Box(-np.inf, np.inf, (O,)), Box(-1, 1, (A,)), seed=seed)
    return agent


def time_events(fn, repeats):
    """Average GPU duration (ms) of fn() measured with events on the launch stream."""
    import torch
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    fn()
    torch.cuda.synchronize()
    start.record()
    for _ in range(repeats):
        fn()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) / repeats


# The arithmetic type of the path.  Everything is fp32 in and fp32 out with fp32 accumulation; the
# three 64x64 products of the fused grad kernels split each fp32 operand (scaled by a power of two into
# binary16's range) into two fp16 terms and sum the three fp16 MFMAs that matter at fp32 precision
# ("fp16x2"; tests/test_gpu_parity.py holds that variant to the error of the fp32-MFMA variants against
# float64, tonic_set_tuning selects them).
DTYPE = 'f32 (64x64 products of the grad kernels fp16x2-emulated: hi+lo fp16 split of the scaled fp32 operands, 23+ significant bits, fp32 accumulate)'


def pmc_traffic(prefix):
    """HBM bytes per launch from the committed PMC passes (profiles/rNN_traffic.json: FETCH_SIZE
    and WRITE_SIZE collected separately, gfx950 read correction applied) — PMC counters cannot be
    collected from inside the timed run, so the summary of the same kernels is quoted (the newest
    round's file that has the kernel)."""
    here = os.path.dirname(os.path.abspath(__file__))
    for name in ('r06_traffic.json', 'r05_traffic.json', 'r04_traffic.json', 'r03_traffic.json'):
        try:
            with open(os.path.join(here, 'profiles', name)) as f:
                table = json.load(f)['kernels']
            entry = next(v for k, v in table.items() if k.startswith(prefix))
        except (OSError, StopIteration, KeyError, ValueError):
            continue
        return dict(traffic=entry['read_bytes'] + entry['write_bytes'], traffic_unit='B/launch',
                    traffic_source='profiles/' + name)
    return dict(traffic=None)


def rocprof_us(prefix):
    """Average duration (us) of a kernel in the committed rocprofv3 --kernel-trace --stats summary of
    `python bench.py` (profiles/rNN_kernel_us.json, written by scripts/kernel_us.py from the
    stats csv of the same round; the newest round's), next to the HIP-event figure measured live."""
    here = os.path.dirname(os.path.abspath(__file__))
    for name in ('r06_kernel_us.json', 'r05_kernel_us.json', 'r04_kernel_us.json'):
        try:
            with open(os.path.join(here, 'profiles', name)) as f:
                table = json.load(f)['kernels']
            return next(v for k, v in table.items() if k.startswith(prefix))
        except (OSError, StopIteration, KeyError, ValueError):
            continue
    return None


# The ceiling the shipped arithmetic has: 12 % of the grad kernel's flops run at the fp32 MFMA rate, 88 %
# as three fp16 MFMAs per product = 16 / 3 x the fp32 rate (six bf16 MFMAs: 8 / 3 x)
def mixed_ceiling_tflops(split_share, rate=8.0 / 3.0):
    return 1.0 / ((1.0 - split_share) / FP32_MFMA_PEAK_TFLOPS
                  + split_share / (FP32_MFMA_PEAK_TFLOPS * rate))


def kernel_rooflines(agent):
    """Live roofline measurements of the dominant kernels (HIP events, launch stream)."""
    import torch
    from tonic_amd import _lib, replays
    lib, p = _lib.load(), _lib.ptr
    replay, actor, critic = agent.replay, agent.actor_updater, agent.critic_updater
    b = replay.buffers
    n = T * W
    obs = replays.flatten_batch(b['observations'])
    act = replays.flatten_batch(b['actions'])
    adv = replays.flatten_batch(b['advantages'])
    logp = replays.flatten_batch(b['log_probs'])
    ret = replays.flatten_batch(b['returns'])
    ws = actor._workspace_for(n)
    stream = _lib.current_stream()

    def actor_grad():
        _lib.check(lib.tonic_ppo_actor_grad(
            p(actor.flat.flat), p(obs), p(act), p(adv), p(replay.adv_stats), p(logp),
            p(actor.grad_sums), n, O, A, 0.2, 0.0, None, 0, p(ws), ws.numel(), stream), 'actor')

    mean, std = critic.norm_tensors()
    wsc = critic._workspace_for(n)

    def critic_grad():
        _lib.check(lib.tonic_value_regression_grad(
            p(critic.flat.flat), p(mean), p(std), 0.0, p(obs), p(ret), p(critic.grad_sums), n, O, 0,
            p(wsc), wsc.numel(), stream), 'critic')

    # The product library holds one form of the grad kernels (grad_variant 4: layer 1, the 64x64 products and
    # the head's forward product on fp16x2 terms; the fp32-MFMA / bf16x3 references live in the tests' dev build).
    shipped = ctypes.c_int32(-1)
    _lib.check(lib.tonic_get_tuning(b'grad_variant', ctypes.byref(shipped)), 'tuning')
    shipped = shipped.value
    # The shipped variant the way the JOB runs it: 80 launches behind a phase in which the chip is
    # lightly loaded for as long as a rollout takes (42 ms), twice = 160 launches.  The device's
    # clock follows its load with a lag of ~25 ms (profiles/r04_clock_ramp.md: the first launches
    # of an update take 320 - 350 us, the 80th 283 us), so a back-to-back loop — `steady_state`
    # below, 120 launches — shows what the kernel can do and not what the job pays; round 3's line
    # quoted such a loop (286 us) where rocprofv3 averaged 307 us over the job's launches.
    ws, wsc = actor._workspace_for(n), critic._workspace_for(n)

    def as_in_the_job(fn, launches=ITERATIONS, rounds=2, idle_s=0.042, before=None):
        start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        total = 0.0
        for _ in range(rounds):
            torch.cuda.synchronize()
            time.sleep(idle_s)
            for _ in range(launches if before is not None else 0):
                before()
            start.record()
            for _ in range(launches):
                fn()
            end.record()
            torch.cuda.synchronize()
            total += start.elapsed_time(end)
        return total / (rounds * launches)
    steady_a, steady_c = time_events(actor_grad, 120), time_events(critic_grad, 120)
    # (the PPO agent holds the critic's chain back to the END of the rollout — tonic_stream_gate — so the
    #  actor's launches follow 80 critic launches, and the critic's follow the lightly loaded phase)
    gated = os.environ.get('TONIC_AMD_CRITIC_GATE', '1') != '0'
    ms_a = as_in_the_job(actor_grad, idle_s=0.022, before=critic_grad) if gated else as_in_the_job(actor_grad)
    ms_c = as_in_the_job(critic_grad, idle_s=0.022 if gated else 0.042)
    tf_a = ACTOR_FLOP_PER_SAMPLE * n / (ms_a * 1e-3) / 1e12
    tf_c = CRITIC_FLOP_PER_SAMPLE * n / (ms_c * 1e-3) / 1e12
    # The ceiling is the kernel's OWN instruction mix: 88 % of the fp32-equivalent flops run as three fp16
    # MFMAs per product (16 / 3 x the fp32 MFMA rate), 12 % (dW1, dW3, the head's backward product) as fp32
    # MFMAs.  `frac` is measured against that; what the same arithmetic would cost un-split — the fp32 MFMA
    # peak, against which a split kernel can read above 1 — is kept beside it as a speed-up figure only.
    rate = 16.0 / 3.0
    arithmetic = ('fp32 in / out / accumulate; 88 % (critic: 91 %) of the flops (layer 1, h1->z2, dz2->dh1, dW2, the head\'s '
                  'forward product) as three fp16 MFMAs per product on hi+lo fp16 splits of power-of-two-scaled '
                  'fp32 operands (23+ significant bits), 12 % (dW1, dW3, the head\'s backward product) on fp32 MFMA')

    def entry(kernel, tf, ms, steady_ms, flop, split_flop):
        # split_flop: the flops per sample that run on fp16x2 terms (layer 1 2 O 64, three 64x64 products,
        # the actor's head forward 2 A 64)
        share = split_flop / flop
        ceiling = mixed_ceiling_tflops(share, rate)
        steady_tf = flop * n / (steady_ms * 1e-3) / 1e12
        return dict(bound='mfma', kernel=kernel, achieved=round(tf, 2), peak=round(ceiling, 1), unit='TFLOP/s',
                    frac=round(tf / ceiling, 4),
                    peak_is=f'the ceiling of the kernel\'s own instruction mix: {100 * (1 - share):.0f} % of the '
                            f'fp32-equivalent flops at the fp32 MFMA peak (157.3), {100 * share:.0f} % as 3 fp16 MFMAs '
                            f'per product at the fp16 peak (2 516.6)',
                    fp32_equivalent=dict(peak=FP32_MFMA_PEAK_TFLOPS, frac=round(tf / FP32_MFMA_PEAK_TFLOPS, 4),
                                         note='speed-up over the un-split arithmetic, not a roofline fraction'),
                    ms_per_launch=round(ms, 4), launches_timed=2 * ITERATIONS,
                    steady_state=dict(ms_per_launch=round(steady_ms, 4), launches_timed=120,
                                      achieved=round(steady_tf, 2), frac=round(steady_tf / ceiling, 4)),
                    samples_per_launch=n, flop_per_sample=flop)
    split = 2 * O * 64 + 3 * 2 * 64 * 64
    roof = entry('mlp64_grad16_kernel<actor> (+reduce_partials)', tf_a, ms_a, steady_a, ACTOR_FLOP_PER_SAMPLE,
                 split + 2 * A * 64)
    roof.update(pmc_traffic('mlp64_grad16_kernel<actor>'), rocprof_us=rocprof_us('mlp64_grad16_kernel<actor>'),
                grad_variant=shipped, arithmetic=arithmetic,
                timed_as=('80 launches behind a 22 ms lightly loaded phase and the critic\'s 80 launches, twice '
                          '(the job\'s pattern with the critic chain gated to the end of the rollout)' if gated
                          else '80 launches behind a 42 ms lightly loaded phase, twice (the job\'s pattern: the '
                               'clock ramps during them)'))
    roof_critic = entry('mlp64_grad16_kernel<critic> (+reduce_partials)', tf_c, ms_c, steady_c,
                        CRITIC_FLOP_PER_SAMPLE, split)
    roof_critic.update(rocprof_us=rocprof_us('mlp64_grad16_kernel<critic>'))

    # GAE scan: 28 algorithmic B / transition (read nv, r, reset, term, values; write ret, adv)
    def gae_entry(t_steps, workers, chunks):
        dev = agent.device
        arrays = [torch.randn(t_steps, workers, device=dev) for _ in range(3)]
        resets = (torch.rand(t_steps, workers, device=dev) < 1e-3).float()
        terms = resets * (torch.rand(t_steps, workers, device=dev) < 0.5).float()
        outs = [torch.empty(t_steps, workers, device=dev) for _ in range(2)]
        stats = torch.zeros(4, device=dev)
        wsg = torch.empty(max(lib.tonic_gae_workspace_bytes(t_steps, workers, chunks), 16),
                          dtype=torch.uint8, device=dev)

        def run():
            _lib.check(lib.tonic_gae_lambda_returns(
                p(arrays[0]), p(arrays[1]), p(resets), p(terms), p(arrays[2]), p(outs[0]),
                p(outs[1]), p(stats), None, t_steps, workers, 0.99, 0.97, chunks, p(wsg),
                wsg.numel(), stream), 'gae')
        ms = time_events(run, 10)
        gbs = 28.0 * t_steps * workers / (ms * 1e-3) / 1e9
        return dict(T=t_steps, W=workers, chunks=chunks, ms=round(ms, 4),
                    achieved=round(gbs, 1), frac=round(gbs / HBM_PEAK_GBS, 4))
    # chunks: 0 = the library's choice (the Segment's default since round 6) = 128-row segments in one pass below
    # W = 65 536; 1 = the bit-exact single chain per column (TONIC_AMD_GAE_EXACT=1); cfg-2 (256), cfg-5 per GPU
    # (1280), cfg-5 global (10 240) and the bandwidth-bound size SURVEY §8d asks for (65 536)
    sweep = [gae_entry(T, W, 0), gae_entry(T, W, 1), gae_entry(T, 1280, 1), gae_entry(T, 1280, 0),
             gae_entry(T, 4096, 0), gae_entry(T, 10240, 1), gae_entry(T, 10240, 0),
             gae_entry(T, 65536, 1), gae_entry(T, 65536, 2)]
    top = max(sweep, key=lambda e: e['achieved'])
    own = sweep[0]          # the metric's own size with the Segment's default (one pass, returns within 1e-5)
    roof_gae = dict(bound='hbm', kernel='gae_onepass_kernel (W = 256, the Segment\'s default); sweep: '
                                        'gae_stream16_kernel (bit-exact chain) / gae_scan_kernel (+gae_stats_kernel)',
                    achieved=own['achieved'], peak=HBM_PEAK_GBS, unit='GB/s', frac=own['frac'],
                    bytes_per_transition=28, at=dict(T=own['T'], W=own['W'], chunks=own['chunks']),
                    exact_chain_at_this_size=dict(achieved=sweep[1]['achieved'], frac=sweep[1]['frac'],
                                                  ms=sweep[1]['ms']),
                    sweep_top=dict(achieved=top['achieved'], frac=top['frac'],
                                   at=dict(T=top['T'], W=top['W'], chunks=top['chunks'])),
                    sweep=sweep, **pmc_traffic('gae_onepass_kernel'),
                    note='the headline figure is the metric\'s own size (T=4096, W=256: 29 MB, inside '
                         'the Infinity Cache: launch- and latency-bound, not bandwidth-bound); the kernel '
                         'reaches its HBM fraction on the sweep (sweep_top: W = 65 536, 1.9 GB)')
    return roof, roof_critic, roof_gae


class _PortEngine:
    """oracle/torch_port.py: the reference's torch-CPU operators restated (kind = "port")."""
    kind = 'port'

    def __init__(self, o_dim, a_dim, steps):
        import torch_port
        self.agent = torch_port.TorchPPO(o_dim, a_dim, steps=steps)

    def step_and_store(self, obs, workers):
        actions = self.agent.step(obs)
        self.agent.store(obs, -np.square(actions).sum(-1), np.zeros(workers, bool),
                         np.zeros(workers, bool))

    def evaluate_and_returns(self, data):
        self.agent.buffers = {k: v.copy() for k, v in data.items()}
        return self.agent.evaluate_and_returns()

    def iteration(self, batch):
        self.agent.actor_update(batch['observations'], batch['actions'], batch['advantages'],
                                batch['log_probs'])
        self.agent.critic_update(batch['observations'], batch['returns'])


class _ReferenceEngine:
    """The UNMODIFIED reference (tonic.torch.agents.PPO from /root/reference, imported through
    oracle/reference_loader.py) — only where the checkout exists, i.e. in the build container
    (kind = "reference")."""
    kind = 'reference'

    def __init__(self, o_dim, a_dim, steps):
        import reference_loader
        tonic = reference_loader.load_reference()
        self.agent = tonic.torch.agents.PPO(replay=tonic.replays.Segment(size=steps))
        self.agent.initialize(reference_loader.SyntheticSpace(-np.inf, np.inf, (o_dim,)),
                              reference_loader.SyntheticSpace(-1, 1, (a_dim,)), seed=0)

    def step_and_store(self, obs, workers):                      # trainer.py:44-50
        actions = self.agent.step(obs, 0)
        self.agent.update(observations=obs, rewards=-np.square(actions).sum(-1),
                          resets=np.zeros(workers, bool), terminations=np.zeros(workers, bool),
                          steps=0)

    def evaluate_and_returns(self, data):                        # ppo.py:20-31
        import torch
        agent, replay = self.agent, self.agent.replay
        replay.buffers = {k: v.copy() for k, v in data.items()}
        replay.num_workers = data['rewards'].shape[1]
        batch = replay.get_full('observations', 'next_observations')
        values, next_values = agent._evaluate(**batch)
        replay.compute_returns(values.numpy(), next_values.numpy())
        batch = replay.get_full('observations', 'actions', 'advantages', 'log_probs', 'returns')
        return {k: torch.as_tensor(v) for k, v in batch.items()}

    def iteration(self, batch):                                  # ppo.py:33-37
        self.agent._update_actor_critic(**batch)


def cpu_engines():
    """The CPU implementations that can be timed on this box: the reference itself where its
    checkout exists (never on the GPU box), the oracle's port always."""
    for path in (os.path.join(ROOT, 'oracle'),):
        if path not in sys.path:
            sys.path.insert(0, path)
    engines = [_PortEngine]
    try:
        import reference_loader
        if reference_loader.reference_available():
            engines.insert(0, _ReferenceEngine)
    except ImportError:
        pass
    return engines


def cpu_measure(engine_class, o_dim, a_dim, workers, steps, threads, sample_steps, sample_iters,
                iterations=ITERATIONS):
    """One bounded sample of the CPU path: `sample_steps` act + store steps, the full-size
    evaluate + lambda-returns, `sample_iters` full-batch actor + critic iterations; extrapolated
    to `steps` environment steps and `iterations` iterations."""
    import torch
    torch.set_num_threads(threads)
    n = steps * workers
    rng = np.random.RandomState(0)
    data = dict(
        observations=rng.standard_normal((steps, workers, o_dim)).astype(np.float32),
        next_observations=rng.standard_normal((steps, workers, o_dim)).astype(np.float32),
        actions=np.clip(rng.standard_normal((steps, workers, a_dim)), -1, 1).astype(np.float32),
        rewards=rng.standard_normal((steps, workers)).astype(np.float32),
        log_probs=(rng.standard_normal((steps, workers)) * 0.1 - 6).astype(np.float32))
    # the same episode statistics as the GPU rollout: resets ~ Bernoulli(1e-3), half terminal
    data['resets'] = (rng.uniform(size=(steps, workers)) < 1e-3).astype(np.float32)
    data['terminations'] = data['resets'] * (rng.uniform(size=(steps, workers)) < 0.5).astype(np.float32)
    obs = rng.standard_normal((workers, o_dim)).astype(np.float32)
    engine = engine_class(o_dim, a_dim, steps)
    engine.step_and_store(obs, workers)                  # (untimed: lazy initialisation, thread pool)
    engine.iteration(engine.evaluate_and_returns(data))
    engine = engine_class(o_dim, a_dim, steps)
    t0 = time.perf_counter()
    for _ in range(sample_steps):
        engine.step_and_store(obs, workers)
    t_step = (time.perf_counter() - t0) / sample_steps
    t0 = time.perf_counter()
    batch = engine.evaluate_and_returns(data)
    t_eval = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(sample_iters):
        engine.iteration(batch)
    t_iter = (time.perf_counter() - t0) / sample_iters
    cycle = steps * t_step + t_eval + iterations * t_iter
    return dict(kind=engine.kind, threads=threads, env_steps_per_sec=round(n / cycle, 1),
                learner_updates_per_sec=round(iterations / (t_eval + iterations * t_iter), 3),
                seconds=dict(per_env_step=round(t_step, 6), evaluate_and_gae=round(t_eval, 4),
                             per_iteration=round(t_iter, 4), cycle=round(cycle, 2)))


def cpu_baseline(o_dim=None, a_dim=None, workers=None, sample_steps=64, sample_iters=3):
    """The reference's torch-CPU PPO path on this box's cores, bounded sample (cpu_measure).  What
    is timed: the UNMODIFIED reference when its checkout is present (kind "reference" — the build
    container), oracle/torch_port.py otherwise (kind "port" — the GPU box, where /root/reference
    does not exist; profiles/r04_cpu_port_vs_reference.json holds both timed side by side in the
    build container).  Measured at torch's default thread count and at 16 threads (the default of
    128 threads on a big host is pathological for these small operators); `value` is the FASTER."""
    import torch
    o_dim, a_dim, workers = o_dim or O, a_dim or A, workers or W
    default_threads = torch.get_num_threads()
    engine = cpu_engines()[0]
    runs = [cpu_measure(engine, o_dim, a_dim, workers, T, default_threads, sample_steps, sample_iters)]
    if default_threads > 16:
        runs.append(cpu_measure(engine, o_dim, a_dim, workers, T, 16, sample_steps, sample_iters))
    torch.set_num_threads(default_threads)
    best = max(runs, key=lambda r: r['env_steps_per_sec'])
    return dict(value=best['env_steps_per_sec'], unit='env_steps/s', cores=best['threads'],
                kind=engine.kind, learner_updates_per_sec=best['learner_updates_per_sec'],
                sample=f'{sample_steps} act+store steps (W={workers}), full-size evaluate+GAE '
                       f'(N={T * workers}), {sample_iters} full-batch actor+critic iterations; '
                       f'extrapolated to T={T} steps and {ITERATIONS} iterations',
                runs=runs, os_cpu_count=os.cpu_count())


def build_offpolicy(kind, o_dim, a_dim, batch, workers, iterations=50, rows=1000000, seed=0,
                    fill_seed=0):
    """An off-policy agent of this package whose HBM Buffer (this rank's `workers` columns of the
    global [rows // W_global, W_global] store) is full of synthetic transitions (SURVEY §8d cfg 3:
    obs ~ N(0, 1), actions ~ U(-1, 1), rewards ~ N(0, 1), terminations ~ Bernoulli(1e-3))."""
    import torch
    import tonic_amd
    import tonic_amd.torch as tt
    from tonic_amd.environments import Box
    replay = tonic_amd.replays.Buffer(size=rows, batch_iterations=iterations, batch_size=batch)
    agent = dict(sac=tt.agents.SAC, td3=tt.agents.TD3, d4pg=tt.agents.D4PG,
                 mpo=tt.agents.MPO)[kind](replay=replay)
    agent.initialize(Box(-np.inf, np.inf, (o_dim,)), Box(-1, 1, (a_dim,)), seed=seed)
    replay._allocate(workers, o_dim, a_dim)
    gen = torch.Generator(device=agent.device)
    gen.manual_seed(fill_seed)
    for key, buf in replay.buffers.items():
        if key in ('resets', 'terminations'):
            buf.copy_((torch.rand(buf.shape, device=agent.device, generator=gen) < 1e-3).float())
        elif key == 'discounts':
            buf.copy_((1 - replay.buffers['terminations']) * 0.99)
        elif key == 'actions':
            buf.copy_(torch.rand(buf.shape, device=agent.device, generator=gen) * 2 - 1)
        else:
            buf.copy_(torch.randn(buf.shape, device=agent.device, generator=gen))
    replay.size, replay.index = replay.max_size, 0
    return agent, replay


def offpolicy_flop_per_iteration(kind, o_dim, a_dim, batch, H=256):
    """Dense fp32 contractions of one learner iteration, 2 FLOP per MAC; TD3 steps
    its actor every second iteration."""
    heads = 2 if kind == 'sac' else 1
    actor_fwd = 2 * (o_dim * H + H * H + heads * H * a_dim)
    critic_fwd = 2 * ((o_dim + a_dim) * H + H * H + H)
    critic_dx_hidden = 2 * (H * H + H)
    critic_step = actor_fwd + 4 * critic_fwd + 2 * critic_fwd + 2 * critic_dx_hidden
    used = 2 if kind == 'sac' else 1                     # critics behind the actor's loss
    actor_step = (actor_fwd + used * critic_fwd + used * 2 * (H + H * H + H * a_dim)
                  + actor_fwd + 2 * (H * H + heads * H * a_dim))
    return batch * (critic_step + (actor_step if kind == 'sac' else actor_step / 2))


def offpolicy_rates(kind='sac', o_dim=111, a_dim=8, batch=1024, workers=1, cpu=True):
    """cfg 3 / cfg 4 of BASELINE.json (not the headline metric).  Default: SAC, O=111, A=8, default
    256-wide networks, 1 M-transition HBM Buffer, B=1024, 50 iterations per update call.  Reports
    learner updates (batch iterations) per second on the GPU, eager and hipGraph-replayed, and
    the reference's torch-CPU path (oracle/torch_port.OffPolicyPort) on a 5-iteration sample."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, 'oracle'))
    import torch_port
    iterations, rows = 50, 1000000
    agent, replay = build_offpolicy(kind, o_dim, a_dim, batch, workers, iterations, rows)

    def one_update(graph):
        indices = replay.sample_indices()
        eps = agent._draw_noise(iterations)
        return agent.enqueue_update(indices, eps, graph=graph)

    out = {'workload': f'{kind.upper()} O={o_dim} A={a_dim} workers={workers} hidden=256 B={batch}, {iterations} iterations '
                       f'per update, {rows} transitions resident in HBM '
                       f'({sum(b.numel() for b in replay.buffers.values()) * 4 / 1e9:.2f} GB)'}
    for label, graph in (('eager', False), ('hip_graph', True)):
        one_update(graph)
        one_update(graph)
        reps, dt = 5, float('inf')
        for _ in range(3):                       # best of three groups: allocator / capture
            torch.cuda.synchronize()             # hiccups of a fresh agent are not the path
            t0 = time.perf_counter()
            for _ in range(reps):
                infos = one_update(graph)
            infos.cpu()
            dt = min(dt, (time.perf_counter() - t0) / reps)
        out[label] = {'learner_updates_per_sec': round(iterations / dt, 1),
                      'ms_per_update_call': round(dt * 1e3, 3)}
    if kind in ('d4pg', 'mpo'):      # (the rows added last: rates only)
        out['us_per_iteration'] = round(out['hip_graph']['ms_per_update_call'] * 1e3 / iterations, 1)
        return out
    # roofline of one learner iteration: dense fp32 contractions, 2 FLOP per MAC
    per_iteration = offpolicy_flop_per_iteration(kind, o_dim, a_dim, batch)
    seconds = out['hip_graph']['ms_per_update_call'] * 1e-3 / iterations
    tflops = per_iteration / seconds / 1e12
    out['roofline'] = dict(bound='mfma (latency-bound in practice: 5 dependent launches of '
                                 '15-40 us per iteration at B=1024, 3 when the actor is not due)',
                           flop_per_iteration=int(per_iteration), achieved=round(tflops, 2),
                           peak=FP32_MFMA_PEAK_TFLOPS, unit='TFLOP/s',
                           frac=round(tflops / FP32_MFMA_PEAK_TFLOPS, 4),
                           us_per_iteration=round(seconds * 1e6, 1))
    if not cpu:
        return out
    # CPU baseline: same path on torch-CPU, bounded sample
    state = {'pre/' + k: v.detach().cpu().numpy() for k, v in agent.model.state_dict().items()}
    port_ = torch_port.OffPolicyPort(kind, state, 'pre/')
    rng = np.random.RandomState(0)
    sample_rows = 4096
    host = dict(observations=rng.standard_normal((sample_rows, 1, o_dim)),
                actions=rng.uniform(-1, 1, (sample_rows, 1, a_dim)),
                next_observations=rng.standard_normal((sample_rows, 1, o_dim)),
                rewards=rng.standard_normal((sample_rows, 1)),
                discounts=np.full((sample_rows, 1), 0.99))
    host = {k: v.astype(np.float32) for k, v in host.items()}
    idx = rng.randint(sample_rows, size=(6, batch))
    eps = rng.standard_normal((6, 2 if kind == 'sac' else 1, batch, a_dim)).astype(np.float32)
    port_.update(host, 1, idx[:1], eps[:1])
    default_threads = torch.get_num_threads()
    runs = []
    for threads in ([default_threads, 16] if default_threads > 16 else [default_threads]):
        torch.set_num_threads(threads)
        port_.update(host, 1, idx[:1], eps[:1])
        t0 = time.perf_counter()
        port_.update(host, 1, idx[1:], eps[1:])
        dt = (time.perf_counter() - t0) / 5
        runs.append({'threads': threads, 'learner_updates_per_sec': round(1 / dt, 2)})
    torch.set_num_threads(default_threads)
    best = max(runs, key=lambda r: r['learner_updates_per_sec'])
    out['cpu_baseline'] = {'learner_updates_per_sec': best['learner_updates_per_sec'],
                           'kind': 'port', 'cores': best['threads'], 'runs': runs,
                           'sample': f'5 {kind.upper()} iterations (critic + actor + polyak) at '
                                     f'B={batch}; the faster of the default and 16 torch threads'}
    return out


def offpolicy_loop(kind='sac', o_dim=111, a_dim=8, batch=1024, workers=1, loop_iterations=2000, rows=1000000):
    """BASELINE configs 3 / 4 on their own metric: env steps/s AND learner updates/s of the REAL loop
    (tonic/utils/trainer.py:42-56: agent.step -> environment.step -> agent.update; ddpg.py:45-72,
    explorations/noisy.py:15-47, replays/buffers.py:28-56,81-91) over the vectorised synthetic simulator — the
    policy acts on every step (warm-up behind us: steps start at 100 000), every transition is stored in the
    HBM Buffer (pre-filled: 1 M transitions), and an update of 50 batch iterations fires whenever
    `steps_between_batches = 50` environment steps have passed (W = 1: every 50th loop iteration; W = 64: every
    loop iteration — the reference's schedule).  `host_loop`: where a loop iteration goes outside the learner."""
    import torch
    from tonic_amd.environments import SyntheticBatch
    iterations = 50
    agent, replay = build_offpolicy(kind, o_dim, a_dim, batch, workers, iterations, rows)
    env = SyntheticBatch(workers, o_dim, a_dim, max_episode_steps=1000, pool=64)
    env.initialize(seed=3)
    observations = env.start()
    first = 100000
    replay.last_steps = first
    clock = time.perf_counter
    learner = {'calls': 0, 'seconds': 0.0}
    inner = agent._update

    def timed_update(steps):
        t0 = clock()
        inner(steps)
        learner['calls'] += 1
        learner['seconds'] += clock() - t0
    agent._update = timed_update

    def run(count, steps, observations, parts=None):
        for _ in range(count):
            t0 = clock()
            actions = agent.step(observations, steps)
            t1 = clock()
            observations, infos = env.step(actions)
            t2 = clock()
            agent.update(**infos, steps=steps)
            if parts is not None:
                parts += (t1 - t0, t2 - t1, clock() - t2)
            steps += workers
        return steps, observations
    warm = max(2 * (50 // workers + 1), 8)                   # at least two updates: capture + one replay
    steps, observations = run(warm, first, observations)
    torch.cuda.synchronize()
    learner.update(calls=0, seconds=0.0)
    parts = np.zeros(3)
    t0 = clock()
    steps, observations = run(loop_iterations, steps, observations, parts)
    torch.cuda.synchronize()
    dt = clock() - t0
    agent._update = inner
    outside = dt - learner['seconds']
    out = dict(
        workload=f'{kind.upper()} O={o_dim} A={a_dim} workers={workers} B={batch}: agent.step -> environment.step -> '
                 f'agent.update over SyntheticBatch, {loop_iterations} loop iterations, an update of {iterations} '
                 f'iterations every {max(50 // workers, 1)} loop iteration(s), {rows} transitions resident in HBM',
        env_steps_per_sec=round(loop_iterations * workers / dt, 1),
        learner_updates_per_sec=round(learner['calls'] * iterations / dt, 1),
        update_calls=learner['calls'], ms_per_update_call=round(learner['seconds'] / max(learner['calls'], 1) * 1e3, 3),
        learner_share=round(learner['seconds'] / dt, 3),
        host_loop=dict(us_per_env_step=round(outside / loop_iterations * 1e6, 2),
                       agent_step_us=round(parts[0] / loop_iterations * 1e6, 2),
                       env_step_us=round(parts[1] / loop_iterations * 1e6, 2),
                       agent_update_us_without_learner=round((parts[2] - learner['seconds']) / loop_iterations * 1e6, 2),
                       workers=workers))
    close = getattr(agent, 'close', None)
    if close is not None:
        close()
    return out


class HostLoop:
    """The trainer's loop body (tonic/utils/trainer.py:44-56) over the vectorised synthetic
    simulator: agent.step -> environment.step -> agent.update, NumPy in / NumPy out."""

    def __init__(self, agent, workers, seed):
        from tonic_amd.environments import SyntheticBatch
        self.agent, self.workers = agent, workers
        self.env = SyntheticBatch(workers, O, A, max_episode_steps=1000, pool=64)
        self.env.initialize(seed=seed)
        self.observations = self.env.start()
        self.steps = 0

    def run(self, environment_steps):
        agent, env, observations, steps, W = (self.agent, self.env, self.observations,
                                              self.steps, self.workers)
        for _ in range(environment_steps):
            actions = agent.step(observations, steps)
            observations, infos = env.step(actions)
            agent.update(**infos, steps=steps)
            steps += W
        self.observations, self.steps = observations, steps

    def breakdown(self, environment_steps=1024):
        """Seconds per environment step spent in the three calls (no learner update inside:
        call it right after one)."""
        agent, env, W = self.agent, self.env, self.workers
        assert agent.replay.index + environment_steps < agent.replay.max_size
        observations, steps = self.observations, self.steps
        clock, parts = time.perf_counter, np.zeros(3)
        for _ in range(environment_steps):
            t0 = clock()
            actions = agent.step(observations, steps)
            t1 = clock()
            observations, infos = env.step(actions)
            t2 = clock()
            agent.update(**infos, steps=steps)
            parts += (t1 - t0, t2 - t1, clock() - t2)
            steps += W
        self.observations, self.steps = observations, steps
        us = parts / environment_steps * 1e6
        return dict(us_per_env_step=round(float(us.sum()), 2), agent_step_us=round(float(us[0]), 2),
                    env_step_us=round(float(us[1]), 2), agent_update_us=round(float(us[2]), 2),
                    env_steps_per_sec=round(W * 1e6 / float(us.sum()), 1),
                    steps=environment_steps, transport=getattr(agent, 'transport_in_effect', agent.transport))


def parallel_workers_loop(agent, groups=8, per_group=32, steps=512):
    """The trainer's loop body with the worker-PROCESS transport of the reference's `Parallel`
    (tonic/environments/distributed.py:136-155): distribute(builder, 8, 32) = 8 forked groups of 32
    Python environments writing into the shared block the GPU reads in place, two futex words per
    step.  us per environment step of 256 workers (the simulators are 256 Python objects here, so
    this is mostly their cost: 32 sequential `Synthetic.step` calls per group)."""
    from tonic_amd import environments
    env = environments.distribute(lambda: environments.Synthetic(O, A, max_episode_steps=1000),
                                  groups, per_group)
    env.initialize(seed=3)
    observations = env.start()
    replay = agent.replay
    assert replay.index == 0, 'call right after a learner update'
    clock, parts, count = time.perf_counter, np.zeros(3), 0
    for t in range(steps + 32):
        t0 = clock()
        actions = agent.step(observations, t * groups * per_group)
        t1 = clock()
        observations, infos = env.step(actions)
        t2 = clock()
        agent.update(**infos, steps=t * groups * per_group)
        if t >= 32:
            parts += (t1 - t0, t2 - t1, clock() - t2)
            count += 1
    us = parts / count * 1e6
    env.close()
    # finish the segment through the in-process loop's environment is the caller's business:
    # the agent is dropped after this measurement
    return dict(worker_groups=groups, workers_per_group=per_group,
                us_per_env_step=round(float(us.sum()), 1), agent_step_us=round(float(us[0]), 1),
                env_step_us=round(float(us[1]), 1), agent_update_us=round(float(us[2]), 1),
                env_steps_per_sec=round(groups * per_group * 1e6 / float(us.sum()), 1), steps=count)


def cfg1_plumbing(steps=3):
    """BASELINE config 1 — PPO at Pendulum-v1 shapes (O = 3, A = 1), parallel = 1, sequential = 1,
    Segment T = 4096, 80 iterations — through the same host-in-the-loop path: what the plumbing
    costs when there is ONE worker (a step is a GPU round trip whatever W is), with its CPU twin
    (the reference's torch-CPU path at the same sizes) beside it."""
    import torch
    global O, A, W
    saved = (O, A, W)
    O, A, W = 3, 1, 1
    try:
        agent = build_agent(seed=0)
        loop = HostLoop(agent, 1, seed=1)
        loop.run(T)                                       # warm-up incl. the first learner update
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop.run(steps * T)
        agent.settle()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
        host_loop = loop.breakdown(1024)
        loop.run(T - agent.replay.index)
        out = dict(workload='PPO Pendulum-v1 shapes (O=3, A=1), parallel=1 sequential=1, Segment '
                            'T=4096 (N=4096), 80 full-batch iterations; host in the loop',
                   env_steps_per_sec=round(steps * T / elapsed, 1),
                   ms_per_step=round(elapsed / steps * 1e3, 3), steps=steps, host_loop=host_loop,
                   critic_under_next_rollout=getattr(agent, '_critic_stream', None) is not None,
                   cpu_baseline=cpu_baseline(3, 1, 1, sample_steps=512, sample_iters=20))
        out['speedup_vs_cpu_baseline'] = round(out['env_steps_per_sec'] / out['cpu_baseline']['value'], 1)
        agent.close()
        return out
    finally:
        O, A, W = saved


def timed_steps(run_one, steps, warmup, world, finish=None):
    """W untimed + K timed steps between barriers; the MAX over ranks of the elapsed time.  `finish`
    (agent.settle): whatever the last step left running is waited for INSIDE the timed region — PPO's
    critic iterations, which a running job finishes under its next rollout, are finished here."""
    import torch

    def barrier():
        if finish is not None:
            finish()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()
    for _ in range(warmup):
        run_one()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        run_one()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device='cuda')
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(t)
    return elapsed


def measure_job(workers, rank, world, steps, warmup, capture, device_too=True):
    """Builds one agent with `workers` workers on this rank and times (a) the host-in-the-loop
    path and (b) the device-resident path; returns (agent, loop, rollout, results)."""
    from tonic_amd.rollout import DeviceRollout
    agent = build_agent(seed=0)                      # same seed: replicated parameters
    loop = HostLoop(agent, workers, seed=1 + rank)
    elapsed = timed_steps(lambda: loop.run(T), steps, warmup, world, finish=agent.settle)
    out = dict(elapsed=elapsed, ms_per_step=elapsed / steps * 1e3,
               value=world * T * workers * steps / elapsed,
               actor_iterations=int((agent.last_infos[0][:, 6] > 0).sum()),
               # device time of the LAST timed update's two chains (80 x [grad + fold + Adam] each; the critic's
               # ran under the rollout that followed, or inside `finish`), by events
               actor_chain_ms=round(getattr(agent, 'actor_chain_ms', None) or 0.0, 3),
               critic_chain_ms=round(getattr(agent, 'critic_chain_ms', None) or 0.0, 3))
    rollout = None
    if device_too:
        rollout = DeviceRollout(agent, workers, T, seed=1 + rank)

        def device_step():
            rollout.collect(capture=capture)
            agent._update()                          # enqueue + one read-back + normaliser
        d = timed_steps(device_step, steps, 1, world)
        out['device_resident'] = dict(
            env_steps_per_sec=round(world * T * workers * steps / d, 1),
            ms_per_step=round(d / steps * 1e3, 3),
            note='rollout inputs pre-generated in HBM, one fused launch per environment step '
                 'replayed from a hipGraph, no host in the loop')
    return agent, loop, rollout, out


def spawn_ranks(gpus):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, exactly as the
    driver does (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...), pass rank 0's
    JSON line through and fail loudly unless it reports n_gpus == N.  On a box with fewer devices
    than ranks the ranks share the devices under the gloo backend (a functional run of the N-rank
    path, not a scaling measurement: the line then says `ranks_share_devices`)."""
    import socket
    import subprocess
    import torch
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    devices = torch.cuda.device_count()
    if devices < gpus:
        env.setdefault('TONIC_AMD_BACKEND', 'gloo')
        env['TONIC_AMD_BENCH_SHARED_DEVICES'] = str(max(devices, 1))
    command = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node',
               str(gpus), '--master-addr', '127.0.0.1', '--master-port', str(port),
               os.path.abspath(__file__)] + sys.argv[1:]
    done = subprocess.run(command, env=env, stdout=subprocess.PIPE, text=True)
    lines = [line for line in done.stdout.splitlines() if line.startswith('{')]
    if done.returncode != 0 or len(lines) != 1:
        sys.stdout.write(done.stdout)
        sys.exit(f'bench.py --gpus {gpus}: the {gpus}-rank launch failed (exit code '
                 f'{done.returncode}, {len(lines)} JSON lines)')
    n = json.loads(lines[0]).get('n_gpus')
    if n != gpus:
        sys.exit(f'bench.py --gpus {gpus}: the line reports n_gpus = {n}')
    print(lines[0])


def allreduce_latency(floats, world):
    """us per in-place sum all-reduce of `floats` float32 on this job's process group (RCCL over
    xGMI on a multi-GPU node), and of tonic_allreduce_f32 (one-shot peer windows) beside it."""
    import torch
    from tonic_amd import parallel
    if world == 1:
        return None
    buffer = torch.zeros(floats, device='cuda')
    out = dict(floats=floats, bytes=4 * floats, backend=torch.distributed.get_backend())

    def rccl():
        torch.distributed.all_reduce(buffer)
    for _ in range(5):
        rccl()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        rccl()
    torch.cuda.synchronize()
    out['process_group_us'] = round((time.perf_counter() - t0) / 50 * 1e6, 1)
    # (a set-up step that fails on ONE rank — no peer access on this box, ... — is agreed on by all)
    comm = parallel.OneShotAllReduce(floats, tolerant=True)
    ok, why = parallel._agree(comm.handle is not None, comm.error)
    if ok:
        out['one_shot_us'] = round(time_events(lambda: comm.all_reduce(buffer), 50) * 1e3, 1)
        try:
            comm.check()
        except Exception as error:
            ok, why = False, str(error)
        ok, why = parallel._agree(ok, why)
        torch.distributed.barrier()
    if not ok:
        out['one_shot_error'] = why[:200]
    comm.close()
    out['learner_uses'] = parallel.allreduce_choice()
    return out


def cfg4_main(args, rank, world):
    """BASELINE config 4: TD3 on humanoid-walk shapes (O=67, A=21), parallel=512 workers SHARDED
    over the ranks: every rank holds 512 / N worker columns of the global
    [1953, 512] Buffer in HBM, draws the same global index / noise streams, gathers the samples of
    its own columns (binomial split of the batch), all-reduces the gradient SUMS of every
    optimizer step and scales by 1 / B_global.  One step = one learner update call = 50 batch
    iterations (25 actor steps, TD3's delay of 2).  value = iterations / s of the whole job —
    strong scaling: the global batch is fixed."""
    import torch
    kind, o_dim, a_dim, global_workers, iterations = 'td3', 67, 21, 512, 50
    assert global_workers % world == 0, (global_workers, world)
    workers = global_workers // world
    results = {}
    agent = None
    for batch in (100, 1024):             # the reference's default batch, and the batch of cfg 3
        del agent
        torch.cuda.empty_cache()
        agent, replay = build_offpolicy(kind, o_dim, a_dim, batch, workers, iterations,
                                        fill_seed=rank)

        def run_one():
            indices = replay.sample_indices()              # the GLOBAL stream, same on every rank
            eps = agent._draw_noise(iterations)
            agent.enqueue_update(indices, eps).cpu()       # one read-back per update call
        elapsed = timed_steps(run_one, args.steps, max(args.warmup, 2), world)
        results[batch] = dict(elapsed=elapsed, ms_per_step=elapsed / args.steps * 1e3,
                              value=iterations * args.steps / elapsed)
    main_run = results[100]
    critic_floats = agent.critic_updater.count + 8
    actor_floats = agent.actor_updater.count + 8
    flop = offpolicy_flop_per_iteration(kind, o_dim, a_dim, 100)
    seconds = main_run['elapsed'] / (args.steps * iterations)
    result = {
        'metric': 'learner updates/sec, TD3 humanoid-walk parallel=512 sharded',
        'value': round(main_run['value'], 1), 'unit': 'updates/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': max(args.warmup, 2),
        'ms_per_step': round(main_run['ms_per_step'], 3), 'higher_is_better': True,
        'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': f'TD3 dm_control humanoid-walk shapes (O={o_dim}, A={a_dim}), 256-wide '
                               f'networks, parallel={global_workers} workers sharded = {workers} '
                               f'columns per GPU of the global [{replay.max_size}, {global_workers}] '
                               f'Buffer (1 M transitions, {sum(b.numel() for b in replay.buffers.values()) * 4 / 1e9:.2f} '
                               'GB per GPU resident), global batch B=100 (reference default) drawn as '
                               'one global index stream and split by worker column, 50 iterations + '
                               '25 actor steps per update call, gradient sums all-reduced per '
                               'optimizer step',
                   'workers_per_gpu': workers, 'global_workers': global_workers,
                   'global_batch': 100, 'batch_iterations': iterations,
                   'parallelism': f'dp{world} (worker-axis shard of the Buffer, all-reduce of flat '
                                  'gradient sums)'},
        'us_per_iteration': round(seconds * 1e6, 1),
        # the exchange of one iteration: the critics' sums every iteration, the actor's every second one
        # (TD3's delay) — 1.5 all-reduces of 4 x these floats; one rank: none
        'allreduce_bytes_per_iteration': 4 * critic_floats + 2 * actor_floats,
        'allreduces_per_iteration': 1.5 if world > 1 else 0,
        # one rank: the whole update call (50 x 5 / 3 chained launches) replays from one hipGraph; several
        # ranks: the same chained launches in two halves around the exchange (tonic_q_iteration_t.phase) + the
        # two optimizer launches, eager (the exchange is a host call; per-rank batch sizes vary by iteration)
        'hip_graph': bool(getattr(agent, '_graph', None) is not None),
        'launches_per_iteration': 5 if world == 1 else 7,
        'fused_in_phases': bool(getattr(agent, '_phased_update', False)),
        'allreduce_floats': dict(critics_every_iteration=critic_floats,
                                 actor_every_second=actor_floats),
        'roofline': dict(bound='mfma (latency-bound at B=100: 7 row tiles)', flop_per_iteration=int(flop),
                         achieved=round(flop / seconds / 1e12, 3), peak=FP32_MFMA_PEAK_TFLOPS,
                         unit='TFLOP/s', frac=round(flop / seconds / 1e12 / FP32_MFMA_PEAK_TFLOPS, 5),
                         traffic=None),
        'B=1024': dict(updates_per_sec=round(results[1024]['value'], 1),
                       us_per_iteration=round(results[1024]['elapsed'] / (args.steps * iterations) * 1e6, 1)),
    }
    if world > 1:
        flat = agent.model.flat_online
        low, high = flat.clone(), flat.clone()
        torch.distributed.all_reduce(low, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(high, op=torch.distributed.ReduceOp.MAX)
        identical = bool(torch.equal(low, high))
        assert identical, 'parameters diverged across ranks'
        result['ranks_hold_identical_parameters'] = identical
        result['rccl_ranks'] = world if torch.distributed.get_backend() == 'nccl' else 0
        result['backend'] = torch.distributed.get_backend()
        result['allreduce_us'] = allreduce_latency(critic_floats, world)
        if os.environ.get('TONIC_AMD_BENCH_SHARED_DEVICES'):
            result['ranks_share_devices'] = int(os.environ['TONIC_AMD_BENCH_SHARED_DEVICES'])
    elif not args.no_extras:
        sys.path.insert(0, os.path.join(ROOT, 'oracle'))
        import torch_port
        state = {'pre/' + k: v.detach().cpu().numpy() for k, v in agent.model.state_dict().items()}
        port_ = torch_port.OffPolicyPort(kind, state, 'pre/')
        rng = np.random.RandomState(0)
        sample_rows = 4096
        host = dict(observations=rng.standard_normal((sample_rows, 1, o_dim)),
                    actions=rng.uniform(-1, 1, (sample_rows, 1, a_dim)),
                    next_observations=rng.standard_normal((sample_rows, 1, o_dim)),
                    rewards=rng.standard_normal((sample_rows, 1)),
                    discounts=np.full((sample_rows, 1), 0.99))
        host = {k: v.astype(np.float32) for k, v in host.items()}
        count = 201
        idx = rng.randint(sample_rows, size=(count, 100))
        eps = rng.standard_normal((count, 1, 100, a_dim)).astype(np.float32)
        threads = min(torch.get_num_threads(), 16)
        before = torch.get_num_threads()
        torch.set_num_threads(threads)
        port_.update(host, 1, idx[:1], eps[:1])
        t0 = time.perf_counter()
        port_.update(host, 1, idx[1:], eps[1:])
        dt = (time.perf_counter() - t0) / (count - 1)
        torch.set_num_threads(before)
        result['cpu_baseline'] = dict(value=round(1 / dt, 1), unit='updates/s', cores=threads,
                                      kind='port',
                                      sample=f'{count - 1} TD3 iterations (critic step, actor + '
                                             'polyak every second) at B=100 through '
                                             'oracle/torch_port.OffPolicyPort')
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument('--gpus', type=int, default=1)
    # (8 timed steps = 0.6 s: the critic's iterations of the LAST timed update are not hidden under a
    #  next rollout — 23 ms that a steady-state job pays once, not once per three steps)
    parser.add_argument('--steps', type=int, default=8)
    parser.add_argument('--warmup', type=int, default=1)
    parser.add_argument('--no-graph', action='store_true', help='eager launches, no hipGraph')
    parser.add_argument('--quick-extras', action='store_true',
                        help='only the phase split and the host-loop breakdown')
    parser.add_argument('--no-extras', action='store_true',
                        help='skip roofline / cpu_baseline / off-policy measurements')
    parser.add_argument('--scaling', default='weak', choices=('weak', 'strong'),
                        help='weak (default): the configured workers PER GPU; strong: the '
                             'configured workers are the global count, split over the ranks')
    parser.add_argument('--workload', default='cfg2', choices=('cfg2', 'cfg4', 'cfg5'),
                        help='cfg2 (default, the headline metric): HalfCheetah shapes, 256 workers; '
                             'cfg5: AntBullet shapes (O=28, A=8), 1280 workers per GPU under weak '
                             'scaling = BASELINE config 5 at 8 GPUs (10 240 global under strong); '
                             'cfg4: TD3 humanoid-walk shapes, 512 workers sharded over the ranks, '
                             'learner updates/s (BASELINE config 4)')
    args = parser.parse_args()
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return spawn_ranks(args.gpus)                 # no launcher: start the ranks ourselves
    global O, A, W
    if args.workload == 'cfg5':
        O, A, W = 28, 8, 1280
        args.no_extras = True

    # RCCL prints a version banner on STDOUT at NCCL_DEBUG=VERSION (set in this image): the contract
    # is ONE JSON line on rank 0's stdout
    if os.environ.get('NCCL_DEBUG', 'VERSION').upper() == 'VERSION':
        os.environ['NCCL_DEBUG'] = 'WARN'
    import torch
    from tonic_amd import _lib, parallel
    rank, world = parallel.init_from_env()
    if world != max(args.gpus, 1):
        sys.exit(f'bench.py --gpus {args.gpus} runs as {world} rank(s): the line would claim '
                 f'n_gpus = {world} (WORLD_SIZE={os.environ.get("WORLD_SIZE")})')
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    torch.cuda.set_device(local_rank % torch.cuda.device_count())
    from tonic_amd.utils import logger
    logger.get_current_logger().store = lambda *a, **k: None      # no log accumulation here
    if args.workload == 'cfg4':
        return cfg4_main(args, rank, world)
    capture = not args.no_graph
    global_workers = W * world if args.scaling == 'weak' else (W if args.workload == 'cfg2' else 10240)
    assert global_workers % world == 0, (global_workers, world)
    workers = global_workers // world

    # (the device-resident leg replays a hipGraph: single-process only — a capture next to
    #  RCCL's watchdog threads is not worth risking the multi-GPU line for)
    agent, loop, rollout, main_run = measure_job(workers, rank, world, args.steps, args.warmup,
                                                 capture, device_too=world == 1)
    name = 'HalfCheetah-v3' if args.workload == 'cfg2' else 'AntBulletEnv-v0'
    result = {
        'metric': f'env steps/sec (+ learner updates/sec), PPO {name.split("-")[0]} '
                  f'parallel={global_workers if args.scaling == "strong" else W}'
                  + ('' if args.scaling == 'strong' else ' per GPU'),
        'value': round(main_run['value'], 1), 'unit': 'env_steps/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup,
        'ms_per_step': round(main_run['ms_per_step'], 3), 'higher_is_better': True,
        'scaling': args.scaling, 'vs_baseline': None, 'dtype': DTYPE, 'data': 'synthetic',
        'config': {'workload': f'PPO {name} shapes (O={O}, A={A}), parallel={workers} workers per '
                               f'GPU ({global_workers} global), Segment T={T} (N={T * workers} '
                               'transitions per GPU per step), 80 full-batch iterations, 1 learner '
                               'update per step; host in the loop: agent.step / environment.step / '
                               'agent.update through the pinned-host collector',
                   'workers_per_gpu': workers, 'segment_steps': T, 'batch_iterations': ITERATIONS,
                   'global_workers': global_workers,
                   'parallelism': f'dp{world} (worker-axis shard, RCCL all-reduce of flat '
                                  'gradient sums)',
                   'collector_transport': getattr(agent, 'transport_in_effect', agent.transport),
                   # what `environment.step` is here: the vectorised synthetic simulator of this
                   # package, whose whole step (observations ~ N(0,1) from a pool, reward, episode
                   # lengths, resets) is ONE C entry point of the product library writing into the
                   # shared block; with the reference's transport — 8 forked groups of 32 Python
                   # environments — the same loop is bound by the simulators (`parallel_workers`)
                   'simulator': 'SyntheticBatch: library C step (tonic_collector_synthetic_step)',
                   # one GPU, full-batch iterations: the critic's 80 iterations of an update run on a
                   # second stream UNDER the next rollout (agents.PPO._update; every step still
                   # contains one whole update, the last one's tail lies inside the timed region)
                   'critic_under_next_rollout': getattr(agent, '_critic_stream', None) is not None,
                   # the process (and the shared block's pages) on the NUMA node the GPU hangs off: a collect step
                   # is PCIe round trips with this process's memory (profiles/r05_numa.md; TONIC_AMD_NUMA_BIND=0: off)
                   'numa_bind': os.environ.get('TONIC_AMD_NUMA_BIND', '1') != '0',
                   'process_was_moved_to_the_gpus_node': any(v is not None for v in parallel._bound.values()),
                   # how the per-environment-step C entries are bound (tonic_amd/_fastcall: csrc/fastcall.c)
                   'per_step_binding': 'vectorcall shim' if getattr(_lib.hot('tonic_collector_ring'), '__module__', '') == 'tonic_amd._fastcall' else 'ctypes'},
        'learner_updates_per_sec': round(ITERATIONS * args.steps / main_run['elapsed'], 2),
        'actor_iterations_last_update': main_run['actor_iterations'],
    }
    if 'device_resident' in main_run:
        result['device_resident'] = main_run['device_resident']

    if world > 1:
        # RCCL really saw every rank: after the updates above the replicated parameters must be
        # bit-identical everywhere (every rank applied the same all-reduced sums)
        flat = torch.cat([agent.model.flat_actor.flat, agent.model.flat_critic.flat])
        low, high = flat.clone(), flat.clone()
        torch.distributed.all_reduce(low, op=torch.distributed.ReduceOp.MIN)
        torch.distributed.all_reduce(high, op=torch.distributed.ReduceOp.MAX)
        identical = bool(torch.equal(low, high))
        assert identical, 'parameters diverged across ranks'
        result['ranks_hold_identical_parameters'] = identical
        result['backend'] = torch.distributed.get_backend()
        result['rccl_ranks'] = world if result['backend'] == 'nccl' else 0
        if os.environ.get('TONIC_AMD_BENCH_SHARED_DEVICES'):
            result['ranks_share_devices'] = int(os.environ['TONIC_AMD_BENCH_SHARED_DEVICES'])
        # the per-iteration exchange of this job: [actor sums | 8 | critic sums | 8] floats
        result['allreduce_us'] = allreduce_latency(
            agent.actor_updater.count + agent.critic_updater.count + 16, world)
        if args.scaling == 'weak' and args.workload == 'cfg2' and W % world == 0:
            # the metric's own configuration: 256 workers in total, split over the ranks
            # (the first job's agent — Segment, collector, pinned block — is released first: with it
            #  alive, the second agent's exchanges crawled under two gloo ranks sharing one GPU)
            import gc
            agent.close()
            del agent, loop, rollout, flat, low, high
            gc.collect()
            torch.cuda.empty_cache()
            _, _, _, strong = measure_job(W // world, rank, world, args.steps, 1, capture,
                                          device_too=False)
            result['strong_scaling'] = dict(
                global_workers=W, workers_per_gpu=W // world,
                env_steps_per_sec=round(strong['value'], 1),
                ms_per_step=round(strong['ms_per_step'], 3))

    if rank == 0 and not args.no_extras and world == 1:
        # phase split (untimed extras): where a step goes
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loop.run(T - 1)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        loop.run(1)                                       # the T-th update() runs the learner
        agent.settle()                                    # (incl. the critic's iterations it left running)
        torch.cuda.synchronize()
        result['collect_ms'] = round((t1 - t0) * 1e3 * T / (T - 1), 3)
        result['update_ms'] = round((time.perf_counter() - t1) * 1e3, 3)
        result['host_loop'] = loop.breakdown()
        loop.run(T - loop.agent.replay.index)             # finish the segment
        agent.settle()
        result['critic_chain_ms'] = main_run['critic_chain_ms']
        result['actor_chain_ms'] = main_run['actor_chain_ms']
        result['config']['critic_chain_gated'] = bool(
            getattr(agent, '_gate_ticket', 0)) and os.environ.get('TONIC_AMD_CRITIC_GATE', '1') != '0'
        if result['config']['critic_chain_gated']:
            # the same job with the critic's chain started at once instead of held back to the end of the
            # rollout (tonic_stream_gate only arms behind rollouts of < 100 ms, i.e. zero-cost simulators):
            # the headline beside its un-tuned twin, same agent, same harness
            os.environ['TONIC_AMD_CRITIC_GATE'] = '0'
            try:
                off = timed_steps(lambda: loop.run(T), args.steps, 1, world, finish=agent.settle)
            finally:
                os.environ['TONIC_AMD_CRITIC_GATE'] = '1'
            result['critic_gate_off'] = dict(env_steps_per_sec=round(world * T * workers * args.steps / off, 1),
                                             ms_per_step=round(off / args.steps * 1e3, 3), steps=args.steps)
    if rank == 0 and not args.no_extras and not args.quick_extras and world == 1:
        roof, roof_c, roof_g = kernel_rooflines(agent)
        result['roofline'] = roof
        result['roofline_critic'] = roof_c
        result['roofline_gae'] = roof_g
        result['cpu_baseline'] = cpu_baseline()
        result['parallel_workers'] = parallel_workers_loop(agent)
        result['speedup_vs_cpu_baseline'] = round(
            main_run['value'] / result['cpu_baseline']['value'], 1)
        # BASELINE config 5's per-GPU share (AntBullet shapes, 1 280 workers): 2 steps, same harness —
        # BEFORE the off-policy legs: every agent built in this process creates HIP streams, the
        # streams of a process share a handful of hardware queues, and a collector that lands on the
        # queue of the critic's stream loses the overlap this leg is about
        agent.close()
        del agent, loop, rollout
        import gc
        gc.collect()
        torch.cuda.empty_cache()
        O, A, W = 28, 8, 1280
        _, _, _, share = measure_job(W, rank, world, 2, 1, capture, device_too=False)
        result['cfg5_share'] = dict(
            workload='PPO AntBulletEnv-v0 shapes (O=28, A=8), 1280 workers on this GPU (config 5 = '
                     '10 240 workers over 8 GPUs), T=4096: N = 5 242 880 transitions per step',
            env_steps_per_sec=round(share['value'], 1), ms_per_step=round(share['ms_per_step'], 3),
            steps=2)
        O, A, W = 17, 6, 256
        del share
        gc.collect()
        torch.cuda.empty_cache()
        result['cfg1_plumbing'] = cfg1_plumbing()
        result['offpolicy_sac'] = offpolicy_rates()
        # configs 3 / 4 on their own metric (env steps/s + learner updates/s of the whole loop, acting included)
        result['offpolicy_sac']['loop'] = offpolicy_loop('sac', 111, 8, 1024, workers=1, loop_iterations=2000)
        # cfg 4 per-GPU share: TD3, humanoid-walk shapes, 64 of the 512 workers, the
        # reference's default batch of 100 and the batch of cfg 3
        result['offpolicy_td3'] = {}
        for b in (100, 1024):
            rates = offpolicy_rates('td3', 67, 21, b, workers=64, cpu=False)
            result['offpolicy_td3'][f'B={b}'] = dict(rates['hip_graph'], roofline=rates['roofline'])
        result['offpolicy_td3']['loop'] = offpolicy_loop('td3', 67, 21, 100, workers=64, loop_iterations=300)
        # D4PG (51 atoms) and MPO (20 sampled actions per state) on the same shapes, default B=100
        for other in ('d4pg', 'mpo'):
            rates = offpolicy_rates(other, 67, 21, 100, workers=64, cpu=False)
            result['offpolicy_' + other] = dict(rates['hip_graph'], workload=rates['workload'],
                                                us_per_iteration=rates['us_per_iteration'])
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
