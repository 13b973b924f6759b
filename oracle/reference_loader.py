"""TEST INFRASTRUCTURE ONLY — loader for the *unmodified* reference (fabiopardo/tonic).

Nothing in the product path (``tonic_amd/``) may import this file.  It is used by
``oracle/make_golden.py`` (run in the build container, where ``/root/reference`` is
mounted read-only) to execute the reference's own ``tonic.torch`` CPU path and snapshot
golden input/output vectors under ``tests/golden/``.

The reference cannot be imported as shipped here because ``gym`` and ``termcolor`` are
not installed (``tonic/environments/builders.py:5``, ``tonic/utils/logger.py:6``).  Two
stub modules are injected into ``sys.modules`` *before* ``import tonic``; the reference
tree itself is never touched.  Only the attributes the reference dereferences at import
time or on the hot path are provided (``gym.Wrapper``, ``gym.ActionWrapper``,
``gym.core.Env``, ``gym.spaces.Box``, ``gym.wrappers.TimeLimit``, ``termcolor.colored``).
"""
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = os.environ.get('TONIC_REFERENCE_ROOT', '/root/reference')


def reference_available():
    return os.path.isdir(os.path.join(REFERENCE_ROOT, 'tonic'))


class _Box:
    def __init__(self, low, high, shape=None, dtype=np.float32):
        if shape is not None:
            low = np.full(shape, low, dtype)
            high = np.full(shape, high, dtype)
        self.low = np.asarray(low, dtype)
        self.high = np.asarray(high, dtype)
        self.shape = self.low.shape
        self.dtype = np.dtype(dtype)


def _install_stubs():
    if 'gym' not in sys.modules:
        gym = types.ModuleType('gym')

        class Env:
            pass

        class Wrapper(Env):
            def __init__(self, env):
                self.env = env

            def __getattr__(self, name):
                return getattr(self.env, name)

        class ActionWrapper(Wrapper):
            pass

        class TimeLimit(Wrapper):
            pass

        gym.Env, gym.Wrapper, gym.ActionWrapper = Env, Wrapper, ActionWrapper
        for name, attrs in (('core', {'Env': Env}), ('spaces', {'Box': _Box}),
                            ('wrappers', {'TimeLimit': TimeLimit})):
            sub = types.ModuleType('gym.' + name)
            sub.__dict__.update(attrs)
            setattr(gym, name, sub)
            sys.modules['gym.' + name] = sub
        sys.modules['gym'] = gym
    if 'termcolor' not in sys.modules:
        termcolor = types.ModuleType('termcolor')
        termcolor.colored = lambda s, *a, **k: s
        sys.modules['termcolor'] = termcolor


def load_reference():
    """Returns the reference's ``tonic`` package (with ``tonic.torch`` imported)."""
    if not reference_available():
        raise RuntimeError(f'reference checkout not found at {REFERENCE_ROOT}')
    _install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import tonic  # noqa: the reference package, NOT tonic_amd
    import tonic.torch  # noqa
    assert os.path.realpath(tonic.__file__).startswith(os.path.realpath(REFERENCE_ROOT))
    return tonic


class SyntheticSpace(_Box):
    pass


class SyntheticEnvironment:
    """Zero-cost fixed-shape environment: obs ~ N(0,1) f32, reward
    = -||a||^2, never terminates (time-outs only).  Only what ``distribute`` /
    ``Sequential`` touch (``tonic/environments/distributed.py:12-20,24,36,47,161``)."""

    def __init__(self, observation_size, action_size, max_episode_steps=1000,
                 name='synthetic'):
        self.observation_space = _Box(-np.inf, np.inf, (observation_size,))
        self.action_space = _Box(-1, 1, (action_size,))
        self.max_episode_steps = max_episode_steps
        self.name = name
        self.random = np.random.RandomState(0)

    def seed(self, seed):
        self.random = np.random.RandomState(seed)

    def reset(self):
        return self.random.normal(size=self.observation_space.shape).astype(np.float32)

    def step(self, action):
        obs = self.random.normal(size=self.observation_space.shape).astype(np.float32)
        return obs, -float(np.sum(np.square(action))), False, {}
