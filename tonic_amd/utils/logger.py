"""Epoch logger with the key names and CSV semantics of ``tonic/utils/logger.py``.

The key names (``train/steps_per_second``, ``actor/kl`` ...) are part of the drop-in contract:
``tonic.plot`` and downstream tooling read ``log.csv`` / ``config.yaml`` written here.
``termcolor`` is optional (not installed in the ROCm image): plain text is used without it.
"""
import datetime
import os
import sys
import time

import numpy as np

try:
    import termcolor
except ImportError:          # pragma: no cover - depends on the image
    termcolor = None

current_logger = None


def _paint(text, color=None, on_color=None, attrs=None):
    if termcolor is None:
        return text
    return termcolor.colored(text, color, on_color, attrs=attrs)


class Logger:
    def __init__(self, path=None, width=60, script_path=None, config=None):
        self.path = path or str(time.time())
        # one process per GPU: rank 0 owns the experiment folder, the others write next to it
        # (their episode statistics are those of their own workers)
        from tonic_amd import parallel
        rank = parallel.launch_rank()[0]
        if rank > 0:
            self.path = os.path.join(self.path, f'rank{rank}')
        self.log_file_path = os.path.join(self.path, 'log.csv')
        if script_path:
            with open(script_path) as source:
                text = source.read()
            os.makedirs(self.path, exist_ok=True)
            target = os.path.join(self.path, 'script.py')
            with open(target, 'w') as out:
                out.write(text)
            log(f'Script file saved to {target}')
        if config:
            import yaml
            os.makedirs(self.path, exist_ok=True)
            target = os.path.join(self.path, 'config.yaml')
            with open(target, 'w') as out:
                yaml.dump(config, out)
            log(f'Config file saved to {target}')
        self.known_keys = set()
        self.stat_keys = set()
        self.epoch_dict = {}
        self.width = width
        self.last_epoch_progress = None
        self.start_time = time.time()

    def store(self, key, value, stats=False):
        """Keeps named values during an epoch (logger.py:51-59).

        The reference keeps the object it is given; its agents and environments return fresh
        arrays every step.  Here a step's outputs may be persistent views of the collector block
        (deviation 2), overwritten in place by the next step: an array that does not
        own its memory is kept as a copy, so an epoch's statistics are those of every step."""
        if isinstance(value, np.ndarray) and not value.flags.owndata:
            value = value.copy()
        bucket = self.epoch_dict.get(key)
        if bucket is None:
            self.epoch_dict[key] = [value]
            if stats:
                self.stat_keys.add(key)
        else:
            bucket.append(value)

    def _reduce(self):
        for key in list(self.epoch_dict):
            values = self.epoch_dict[key]
            if key in self.stat_keys:
                del self.epoch_dict[key]
                flat = np.concatenate([np.ravel(v) for v in values]) if len(values) else values
                # same reductions as logger.py:66-71 (np.mean/std/min/max over the stored list)
                self.epoch_dict[key + '/mean'] = np.mean(flat)
                self.epoch_dict[key + '/std'] = np.std(flat)
                self.epoch_dict[key + '/min'] = np.min(flat)
                self.epoch_dict[key + '/max'] = np.max(flat)
                self.epoch_dict[key + '/size'] = len(values)
            else:
                self.epoch_dict[key] = np.mean(values)

    def dump(self):
        """Prints the epoch table and appends a row to log.csv (rewritten when new keys appear)."""
        _run_before_dump()
        self._reduce()
        new_keys = [k for k in self.epoch_dict if k not in self.known_keys]
        first_row = not self.known_keys
        if new_keys:
            if not first_row:
                warning(f'Logging new keys {new_keys}')
            self.known_keys.update(new_keys)
            self.final_keys = sorted(self.known_keys)
        print()
        shown = set()
        for key in self.final_keys:
            *groups, leaf = key.split('/')
            for depth in range(len(groups)):
                prefix = '/'.join(groups[:depth + 1])
                if prefix not in shown:
                    shown.add(prefix)
                    print('  ' * depth + groups[depth].replace('_', ' '))
            value = self.epoch_dict.get(key)
            if isinstance(value, (float, np.floating)):
                text = f'{value:8.3g}'
            elif isinstance(value, (int, np.integer)):
                text = f'{value:,}'
            else:
                text = str(value)
            left = '  ' * len(groups) + leaf.replace('_', ' ')
            print(left + ' ' * max(1, self.width - len(left) - len(text)) + text)
        print()
        values = [self.epoch_dict.get(k) for k in self.final_keys]
        os.makedirs(self.path, exist_ok=True)
        if new_keys and not first_row:
            with open(self.log_file_path) as old:
                old_header = old.readline().strip().split(',')
                old_rows = [line.strip().split(',') for line in old if line.strip()]
            with open(self.log_file_path, 'w') as out:
                out.write(','.join(self.final_keys) + '\n')
                for row in old_rows:
                    by_key = dict(zip(old_header, row))
                    out.write(','.join(by_key.get(k, 'None') for k in self.final_keys) + '\n')
                out.write(','.join(map(str, values)) + '\n')
        else:
            with open(self.log_file_path, 'a') as out:
                if first_row:
                    out.write(','.join(self.final_keys) + '\n')
                out.write(','.join(map(str, values)) + '\n')
        self.epoch_dict.clear()
        self.last_epoch_progress = None

    def show_progress(self, steps, num_epoch_steps, num_steps, color='white', on_color='on_blue'):
        epoch_steps = (steps - 1) % num_epoch_steps + 1
        progress = int(self.width * epoch_steps / num_epoch_steps)
        if progress == self.last_epoch_progress:
            return
        per_step = (time.time() - self.start_time) / steps
        left_epoch = datetime.timedelta(seconds=int(max((num_epoch_steps - epoch_steps) * per_step, 0)))
        left_total = datetime.timedelta(seconds=int(max((num_steps - steps) * per_step, 0)))
        msg = f'Time left:  epoch {left_epoch}  total {left_total}'.center(self.width)
        print(_paint('\r' + msg[:progress], color, on_color), end='')
        print(msg[progress:], sep='', end='', flush=True)
        self.last_epoch_progress = progress


# Agents that log part of an update late (the PPO critic's rows, whose iterations run under the
# next rollout) register here: their rows are stored before the epoch they belong to is reduced —
# also under the REFERENCE's logger, whose module-level `dump` gets the same prologue the first time
# this module forwards to it.
_before_dump = []


def before_dump(owner, method):
    import weakref
    _before_dump.append((weakref.ref(owner), method))


def _run_before_dump():
    for ref, method in list(_before_dump):
        owner = ref()
        if owner is None:
            _before_dump.remove((ref, method))
        else:
            getattr(owner, method)()


def _hook_host(host):
    if getattr(host, '_tonic_amd_dump_hooked', False) or not hasattr(host, 'dump'):
        return
    plain = host.dump

    def dump(*args, **kwargs):
        _run_before_dump()
        return plain(*args, **kwargs)
    host.dump = dump
    host._tonic_amd_dump_hooked = True


def initialize(*args, **kwargs):
    global current_logger
    current_logger = Logger(*args, **kwargs)
    return current_logger


def use(module):
    """Forwards every call of this module to `module` (any object with the functions of
    tonic/utils/logger.py); ``use(None)`` restores the automatic choice below."""
    global _injected
    _injected = module


_injected = None


def _host():
    """Where the calls of this module go when they are not handled here.

    Under ``python -m tonic.train --header 'import tonic_amd as amd' --agent 'amd...'`` the
    REFERENCE package owns the run: tonic/train.py:117 initialises ``tonic.logger`` and the
    reference's Trainer dumps it every epoch (tonic/utils/trainer.py:92).  The agents of this
    package log through this module, so as long as this logger has not been initialised itself
    everything is forwarded to the reference's — the learner statistics land in the one
    ``log.csv``, checkpoints are announced and placed under the one experiment path."""
    if _injected is not None:
        return _injected
    if current_logger is not None:
        return None
    package = sys.modules.get('tonic')
    other = getattr(package, 'logger', None) if package is not None else None
    if other is None or other is sys.modules[__name__]:
        return None
    return other if hasattr(other, 'store') and hasattr(other, 'dump') else None


def get_current_logger():
    global current_logger
    host = _host()
    if host is not None:
        _hook_host(host)
        return host.get_current_logger()
    if current_logger is None:
        current_logger = Logger()
    return current_logger


def store(*args, **kwargs):
    return get_current_logger().store(*args, **kwargs)


def dump(*args, **kwargs):
    return get_current_logger().dump(*args, **kwargs)


def show_progress(*args, **kwargs):
    return get_current_logger().show_progress(*args, **kwargs)


def get_path():
    return get_current_logger().path


def log(msg, color='green'):
    host = _host()
    if host is not None:
        return host.log(msg, color)
    print(_paint(msg, color, attrs=['bold']))


def warning(msg, color='yellow'):
    host = _host()
    if host is not None:
        return host.warning(msg, color)
    print(_paint('Warning: ' + msg, color, attrs=['bold']))


def error(msg, color='red'):
    host = _host()
    if host is not None:
        return host.error(msg, color)
    print(_paint('Error: ' + msg, color, attrs=['bold']))
