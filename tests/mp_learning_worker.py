"""One rank of a multi-process training run on the task of tests/reach_task.py (launched by
tests/test_gpu_multirank.py with RANK / WORLD_SIZE / MASTER_* set): trains `case` with this
package's Trainer and writes the rank's reward curve.  usage: mp_learning_worker.py CASE OUT"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import reach_task  # noqa: E402
import tonic_amd  # noqa: E402
import tonic_amd.torch  # noqa: E402


def main():
    case, out = sys.argv[1], sys.argv[2]
    agent = reach_task.build_agent(tonic_amd, tonic_amd.torch.agents, case)
    with tempfile.TemporaryDirectory() as path:
        curve = reach_task.train(tonic_amd, agent, case, path)
    state = {k: float(v.detach().float().abs().sum()) for k, v in agent.model.state_dict().items()}
    json.dump(dict(curve=curve, parameters=state),
              open(f'{out}.rank{os.environ.get("RANK", "0")}.json', 'w'))


if __name__ == '__main__':
    main()
