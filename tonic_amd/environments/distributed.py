"""Batched environment collector — API of ``tonic/environments/distributed.py``.

``distribute(builder, worker_groups, workers_per_group)`` returns an object with
``initialize(seed)``, ``start()``, ``step(actions)``, ``observation_space``, ``action_space``
exactly like the reference (worker order group-major, seeds ``seed + i*S + j``
distributed.py:18-20,109; time-outs reset but are not terminations :40).

What changed versus the reference's ``Parallel`` (pickle over one Pipe per group + one shared
Queue, distributed.py:136-155):

  * all workers write observations / rewards / flags straight into ONE shared float32 block
    (anonymous shared mapping created before ``fork``), laid out
    ``[observations | next_observations | rewards | resets | terminations | actions]`` with the
    worker axis outermost inside each field — no serialisation, no per-step allocation;
  * the block is page-locked with ``hipHostRegister`` when a GPU is present, so the agent's
    ``hipMemcpyAsync`` of the observation rows is a true DMA from the workers' memory;
  * signalling is a pair of counting semaphores per group (go / done) and the parent waits
    with a timeout, so a dead worker raises instead of hanging forever.
"""
import ctypes
import mmap
import multiprocessing

import numpy as np


class _Block:
    """Shared float32 block with named [W, ...] fields."""

    def __init__(self, workers, observation_size, action_size, shared):
        fields = (('observations', (workers, observation_size)),
                  ('next_observations', (workers, observation_size)),
                  ('rewards', (workers,)), ('resets', (workers,)), ('terminations', (workers,)),
                  ('actions', (workers, action_size)))
        total = sum(int(np.prod(shape)) for _, shape in fields)
        nbytes = max(total * 4, mmap.PAGESIZE)
        if shared:
            self.memory = mmap.mmap(-1, nbytes)      # MAP_SHARED | MAP_ANONYMOUS: survives fork
            flat = np.frombuffer(self.memory, np.float32, total)
        else:
            self.memory = None
            flat = np.zeros(total, np.float32)
        self.flat = flat
        self.views, offset = {}, 0
        for name, shape in fields:
            size = int(np.prod(shape))
            self.views[name] = flat[offset:offset + size].reshape(shape)
            offset += size
        self.pinned = False

    def pin(self):
        """Best effort hipHostRegister of the block (no-op without a GPU)."""
        try:
            import torch
            if torch.cuda.is_available() and not self.pinned:
                address = self.flat.ctypes.data
                status = torch.cuda.cudart().cudaHostRegister(address, self.flat.nbytes, 0)
                self.pinned = int(status) == 0
        except Exception:           # registration is an optimisation, never a requirement
            self.pinned = False
        return self.pinned

    def __getitem__(self, name):
        return self.views[name]


def _step_group(environments, lengths, max_episode_steps, block, first, actions):
    """One synchronous step of a group of environments (distributed.py:28-58) writing into
    rows [first, first + len(environments)) of the shared block."""
    for j, environment in enumerate(environments):
        row = first + j
        observation, reward, termination, _ = environment.step(actions[row])
        lengths[j] += 1
        reset = termination or lengths[j] == max_episode_steps      # time-outs are resets
        block['next_observations'][row] = observation
        block['rewards'][row] = reward
        block['resets'][row] = reset
        block['terminations'][row] = termination
        if reset:
            observation = environment.reset()
            lengths[j] = 0
        block['observations'][row] = observation


class Sequential:
    """A group of environments stepped in sequence (distributed.py:8-67)."""

    def __init__(self, environment_builder, max_episode_steps, workers):
        self.environments = [environment_builder() for _ in range(workers)]
        self.max_episode_steps = max_episode_steps
        self.observation_space = self.environments[0].observation_space
        self.action_space = self.environments[0].action_space
        self.name = self.environments[0].name

    def initialize(self, seed):
        for i, environment in enumerate(self.environments):
            environment.seed(seed + i)

    def start(self):
        workers = len(self.environments)
        self.block = _Block(workers, self.observation_space.shape[0],
                            self.action_space.shape[0], shared=False)
        self.block.pin()
        for i, environment in enumerate(self.environments):
            self.block['observations'][i] = environment.reset()
        self.lengths = np.zeros(workers, int)
        return self.block['observations'].copy()

    def step(self, actions):
        _step_group(self.environments, self.lengths, self.max_episode_steps, self.block, 0,
                    np.asarray(actions))
        return _outputs(self.block)

    def render(self, mode='human', *args, **kwargs):
        outs = [env.render(mode=mode, *args, **kwargs) for env in self.environments]
        if mode != 'human':
            return np.array(outs)


def _outputs(block):
    infos = dict(observations=block['next_observations'].copy(),
                 rewards=block['rewards'].copy(),
                 resets=block['resets'] != 0, terminations=block['terminations'] != 0)
    return block['observations'].copy(), infos


def _worker(builder, max_episode_steps, workers, first, seed, block, go, done):
    environments = [builder() for _ in range(workers)]
    for j, environment in enumerate(environments):
        environment.seed(seed + j)
        block['observations'][first + j] = environment.reset()
    lengths = np.zeros(workers, int)
    done.release()
    while True:
        go.acquire()
        _step_group(environments, lengths, max_episode_steps, block, first, block['actions'])
        done.release()


class Parallel:
    """Groups of sequential environments stepped by forked worker processes."""

    def __init__(self, environment_builder, worker_groups, workers_per_group,
                 max_episode_steps, timeout=600.0):
        self.environment_builder = environment_builder
        self.worker_groups = worker_groups
        self.workers_per_group = workers_per_group
        self.max_episode_steps = max_episode_steps
        self.timeout = timeout

    def initialize(self, seed):
        dummy = self.environment_builder()
        self.observation_space = dummy.observation_space
        self.action_space = dummy.action_space
        self.name = getattr(dummy, 'name', None)
        del dummy
        self.started = False
        workers = self.worker_groups * self.workers_per_group
        self.block = _Block(workers, self.observation_space.shape[0],
                            self.action_space.shape[0], shared=True)
        context = multiprocessing.get_context('fork')      # builders are closures (Q12)
        self.go, self.done, self.processes = [], [], []
        for i in range(self.worker_groups):
            go, done = context.Semaphore(0), context.Semaphore(0)
            first = i * self.workers_per_group
            process = context.Process(
                target=_worker, daemon=True,
                args=(self.environment_builder, self.max_episode_steps, self.workers_per_group,
                      first, seed + first, self.block, go, done))
            process.start()
            self.go.append(go)
            self.done.append(done)
            self.processes.append(process)

    def _wait(self):
        for i, done in enumerate(self.done):
            if not done.acquire(timeout=self.timeout):
                alive = self.processes[i].is_alive()
                raise RuntimeError(f'environment worker group {i} did not answer within '
                                   f'{self.timeout}s (process alive: {alive})')

    def start(self):
        assert not self.started
        self.started = True
        self._wait()
        self.block.pin()          # after the fork: registration is per process
        return self.block['observations'].copy()

    def step(self, actions):
        self.block['actions'][:] = actions
        for go in self.go:
            go.release()
        self._wait()
        return _outputs(self.block)


def distribute(environment_builder, worker_groups=1, workers_per_group=1):
    """Distributes workers over parallel and sequential groups (distributed.py:158-172)."""
    dummy = environment_builder()
    max_episode_steps = dummy.max_episode_steps
    del dummy
    if worker_groups < 2:
        return Sequential(environment_builder, max_episode_steps=max_episode_steps,
                          workers=workers_per_group)
    return Parallel(environment_builder, worker_groups=worker_groups,
                    workers_per_group=workers_per_group, max_episode_steps=max_episode_steps)
