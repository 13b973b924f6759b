"""Batched environment collector — API of ``tonic/environments/distributed.py``.

``distribute(builder, worker_groups, workers_per_group)`` returns an object with
``initialize(seed)``, ``start()``, ``step(actions)``, ``observation_space``, ``action_space``
exactly like the reference (worker order group-major, seeds ``seed + i*S + j``
distributed.py:18-20,109; time-outs reset but are not terminations :40).

What changed versus the reference's ``Parallel`` (pickle over one Pipe per group + one shared
Queue, distributed.py:136-155):

  * all workers write observations / rewards / flags straight into ONE shared float32 block
    (``tonic_amd.collector.Block``: an anonymous shared mapping created before ``fork``) — no
    serialisation, no per-step allocation;
  * ``start`` / ``step`` return persistent READ-ONLY VIEWS of that block (the same array objects
    every step, like a simulator that re-uses its output buffers; the next step — and the GPU —
    overwrite the memory behind them in place, so a caller that keeps step outputs across steps
    must copy them).  The agents of ``tonic_amd.torch``
    recognise them by identity and page-lock the block (``tonic_collector_create``), so the
    GPU reads the workers' memory and writes the actions back without a host copy.  Callers
    that keep step outputs across steps ask for ``copy_outputs=True`` (or set
    ``TONIC_AMD_COPY_OUTPUTS=1``) and get fresh arrays like the reference returns;
  * signalling is two futex words in the block header (``tonic_collector_submit_actions`` /
    ``tonic_collector_wait_obs`` in the parent, ``tonic_collector_worker_wait`` / ``_done`` in
    the workers): two system calls per step in the parent whatever the number of groups, and
    the parent waits with a timeout, so a dead worker raises instead of hanging forever.
"""
import multiprocessing
import os

import numpy as np

from tonic_amd.collector import Block


def _copy_default():
    return os.environ.get('TONIC_AMD_COPY_OUTPUTS', '0') == '1'


def _step_group(environments, lengths, max_episode_steps, block, first, actions):
    """One synchronous step of a group of environments (distributed.py:28-58) writing into
    rows [first, first + len(environments)) of the shared block."""
    for j, environment in enumerate(environments):
        row = first + j
        observation, reward, termination, _ = environment.step(actions[row])
        lengths[j] += 1
        reset = termination or lengths[j] == max_episode_steps      # time-outs are resets
        block.next_observations[row] = observation
        block.rewards[row] = reward
        block.set_flags(row, reset, termination)
        if reset:
            observation = environment.reset()
            lengths[j] = 0
        block.observations[row] = observation


def _outputs(block, copy):
    if not copy:
        return block.out_observations, dict(block.infos)      # fresh dict, persistent read-only views
    return block.observations.copy(), {k: v.copy() for k, v in block.infos.items()}


def _rank_offset(workers):
    from tonic_amd import parallel
    return parallel.launch_rank()[0] * workers


class Sequential:
    """A group of environments stepped in sequence (distributed.py:8-67)."""

    def __init__(self, environment_builder, max_episode_steps, workers, copy_outputs=None):
        self.environments = [environment_builder() for _ in range(workers)]
        self.max_episode_steps = max_episode_steps
        self.observation_space = self.environments[0].observation_space
        self.action_space = self.environments[0].action_space
        self.name = self.environments[0].name
        self.copy_outputs = _copy_default() if copy_outputs is None else copy_outputs

    def initialize(self, seed):
        # worker i of rank r is worker r * W + i of the whole job (the reference seeds worker i with
        # seed + i, distributed.py:18-20): N ranks with W / N workers each see what one process
        # with W workers would
        seed += _rank_offset(len(self.environments))
        for i, environment in enumerate(self.environments):
            environment.seed(seed + i)

    def start(self):
        workers = len(self.environments)
        self.block = Block(workers, self.observation_space.shape[0], self.action_space.shape[0])
        self.block.promise_carry_over()              # (_step_group writes exactly that)
        for i, environment in enumerate(self.environments):
            self.block.observations[i] = environment.reset()
        self.lengths = np.zeros(workers, int)
        return self.block.observations.copy() if self.copy_outputs else self.block.out_observations

    def step(self, actions):
        _step_group(self.environments, self.lengths, self.max_episode_steps, self.block, 0,
                    np.asarray(actions))
        self.block.ring()           # the record is complete (Parallel: the last worker group rings)
        return _outputs(self.block, self.copy_outputs)

    def render(self, mode='human', *args, **kwargs):
        outs = [env.render(mode=mode, *args, **kwargs) for env in self.environments]
        if mode != 'human':
            return np.array(outs)


def _worker(*args):
    """Process entry of one worker group.  The process is a fork of the trainer: it has inherited
    GPU handles, page-locked blocks and device tensors that it must never touch — not even to
    release them (the HIP runtime of a forked child is not usable: a finalizer or the garbage
    collector running at interpreter exit ends in a segmentation fault).  So the worker leaves
    through os._exit once its loop is over; what it owns is released with the process."""
    status = 1
    try:
        _worker_loop(*args)
        status = 0
    except BaseException:
        import traceback
        traceback.print_exc()
    finally:
        os._exit(status)


def _worker_loop(builder, max_episode_steps, workers, first, seed, block):
    environments = [builder() for _ in range(workers)]
    for j, environment in enumerate(environments):
        environment.seed(seed + j)
        block.observations[first + j] = environment.reset()
    lengths = np.zeros(workers, int)
    sequence = 0
    block.worker_done()
    while True:
        answer = block.worker_wait(sequence)
        if answer == -2:            # nothing for an hour (the parent may be learning): keep waiting
            continue
        if answer < 0:              # shutdown
            return
        sequence = answer
        _step_group(environments, lengths, max_episode_steps, block, first, block.actions)
        block.worker_done()


class Parallel:
    """Groups of sequential environments stepped by forked worker processes."""

    def __init__(self, environment_builder, worker_groups, workers_per_group,
                 max_episode_steps, timeout=600.0, copy_outputs=None):
        self.environment_builder = environment_builder
        self.worker_groups = worker_groups
        self.workers_per_group = workers_per_group
        self.max_episode_steps = max_episode_steps
        self.timeout = timeout
        self.copy_outputs = _copy_default() if copy_outputs is None else copy_outputs

    def initialize(self, seed):
        dummy = self.environment_builder()
        self.observation_space = dummy.observation_space
        self.action_space = dummy.action_space
        self.name = getattr(dummy, 'name', None)
        del dummy
        self.started = False
        workers = self.worker_groups * self.workers_per_group
        seed += _rank_offset(workers)                       # (see Sequential.initialize)
        self.block = Block(workers, self.observation_space.shape[0],
                           self.action_space.shape[0], worker_groups=self.worker_groups)
        self.block.promise_carry_over()              # (_step_group writes exactly that)
        context = multiprocessing.get_context('fork')      # builders are closures (Q12)
        self.processes = []
        for i in range(self.worker_groups):
            first = i * self.workers_per_group
            process = context.Process(
                target=_worker, daemon=True,
                args=(self.environment_builder, self.max_episode_steps, self.workers_per_group,
                      first, seed + first, self.block))
            process.start()
            self.processes.append(process)

    def _wait(self):
        status = self.block.wait_obs(self.timeout)
        if status != 0:
            dead = [i for i, p in enumerate(self.processes) if not p.is_alive()]
            raise RuntimeError(f'environment workers did not answer within {self.timeout}s '
                               f'(dead worker groups: {dead or "none"})')

    def start(self):
        assert not self.started
        self.started = True
        self._wait()
        return self.block.observations.copy() if self.copy_outputs else self.block.out_observations

    def step(self, actions):
        if actions is not self.block.out_actions and actions is not self.block.actions:
            self.block.actions[:] = actions
        self.block.submit_actions()
        self._wait()
        # (a push collector's armed command is the parent's to issue: the workers cannot store through
        #  the device window; with any other collector the last worker group has taken it already)
        self.block.ring()
        return _outputs(self.block, self.copy_outputs)

    def close(self):
        if getattr(self, 'block', None) is not None:
            self.block.shutdown()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def distribute(environment_builder, worker_groups=1, workers_per_group=1, copy_outputs=None):
    """Distributes workers over parallel and sequential groups (distributed.py:158-172)."""
    dummy = environment_builder()
    max_episode_steps = dummy.max_episode_steps
    del dummy
    if worker_groups < 2:
        return Sequential(environment_builder, max_episode_steps=max_episode_steps,
                          workers=workers_per_group, copy_outputs=copy_outputs)
    return Parallel(environment_builder, worker_groups=worker_groups,
                    workers_per_group=workers_per_group, max_episode_steps=max_episode_steps,
                    copy_outputs=copy_outputs)
