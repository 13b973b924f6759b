"""TEST INFRASTRUCTURE ONLY — developer check: the first learner updates of a case of tests/reach_task.py with the reference
(`ref`: build container, CPU) or this package (`amd`: GPU) — trains for `steps` environment steps
and dumps the model, so that the two can be compared parameter by parameter (how the target-support
semantics of the distributional critic step were found).
usage: reach_first_updates.py {ref|amd} CASE STEPS OUT.npz"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, ROOT)
import reach_task  # noqa: E402

which, case, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
if which == 'ref':
    import reference_loader
    tonic = reference_loader.load_reference()
    import tonic.torch as tt
else:
    import tonic_amd as tonic
    import tonic_amd.torch as tt
agent = reach_task.build_agent(tonic, tt.agents, case)
reach_task.ON_POLICY_RUN = dict(reach_task.ON_POLICY_RUN, steps=steps)
reach_task.OFF_POLICY_RUN = dict(reach_task.OFF_POLICY_RUN, steps=steps)
with tempfile.TemporaryDirectory() as path:
    curve = reach_task.train(tonic, agent, case, path)
np.savez(out, curve=np.array(curve),
         **{k: v.detach().cpu().numpy() for k, v in agent.model.state_dict().items()})
print(which, case, steps, 'reward curve', ' '.join(f'{x:.3f}' for x in curve))
