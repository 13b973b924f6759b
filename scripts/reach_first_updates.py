"""Developer check: the first learner updates of an agent on the task of tests/reach_task.py, with
the reference (`ref`, build container, CPU) or this package (`amd`, GPU): dumps the model after
`steps` environment steps.  usage: reach_first_updates.py {ref|amd} AGENT STEPS OUT.npz"""
import os, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
sys.path.insert(0, ROOT)
import reach_task
which, name, steps, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
if which == 'ref':
    import reference_loader
    tonic = reference_loader.load_reference()
    import tonic.torch as tt
else:
    import tonic_amd as tonic
    import tonic_amd.torch as tt
agent = reach_task.build_agent(tonic, tt.agents, name)
run = dict(reach_task.OFF_POLICY_RUN if name not in ('PPO', 'A2C', 'TRPO') else reach_task.ON_POLICY_RUN)
reach_task.OFF_POLICY_RUN = reach_task.ON_POLICY_RUN = dict(run, steps=steps)
with tempfile.TemporaryDirectory() as path:
    curve = reach_task.train(tonic, agent, name, path)
state = {k: v.detach().cpu().numpy() for k, v in agent.model.state_dict().items()}
np.savez(out, rewards=np.array(reach_task.Reach.rewards), **state)
print(which, name, steps, 'mean reward', np.mean(reach_task.Reach.rewards))
