"""TEST INFRASTRUCTURE ONLY — trains the UNMODIFIED reference's eight torch agents (CPU) on the task
and in the cases of tests/reach_task.py and writes the mean training reward of every tenth of each run to
tests/golden/learning_curves.json: what tests/test_gpu_learning.py holds this package's agents
against (same task, same hyper-parameters, same seeds, same Trainer contract).
Run in the build container: python oracle/make_learning_curves.py [CASE ...]"""
import json
import os
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))

import reference_loader  # noqa: E402
import reach_task  # noqa: E402


def main():
    import torch
    torch.set_num_threads(8)
    tonic = reference_loader.load_reference()
    import tonic.torch
    out = os.path.join(ROOT, 'tests', 'golden', 'learning_curves.json')
    names = sys.argv[1:] or tuple(reach_task.CASES)
    curves = json.load(open(out))['curves'] if sys.argv[1:] and os.path.exists(out) else {}
    for name in names:
        agent = reach_task.build_agent(tonic, tonic.torch.agents, name)
        with tempfile.TemporaryDirectory() as path:
            curves[name] = reach_task.train(tonic, agent, name, path)
        print(name, ' '.join(f'{x:.3f}' for x in curves[name]), flush=True)
    json.dump(dict(task='tests/reach_task.py', generator='oracle/make_learning_curves.py',
                   curves=curves), open(out, 'w'), indent=1)
    print('wrote', out)


if __name__ == '__main__':
    main()
