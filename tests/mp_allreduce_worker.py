"""Worker of tests/test_gpu_multirank.py: one rank of a world_size-N group that exercises
tonic_allreduce_f32 (one-shot, IPC windows) — here between processes sharing one GPU."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tonic_amd import parallel        # noqa: E402


def run(out_path):
    rank, world = parallel.init_from_env()
    comm = parallel.OneShotAllReduce(max_floats=200000)
    results = []
    for call, n in enumerate((11101, 7, 177666, 4096, 11101, 11101, 1, 65536)):
        rng = np.random.RandomState(1000 * call + rank)
        mine = (rng.standard_normal(n) * 10.0 ** rng.randint(-3, 4)).astype(np.float32)
        buffer = torch.as_tensor(mine).cuda()
        comm.all_reduce(buffer)
        results.append(buffer.cpu().numpy())
    torch.cuda.synchronize()
    comm.check()
    # back-to-back calls on one buffer without any host sync in between (slot parity, ordering)
    chain = torch.full((5000,), float(rank + 1), device='cuda')
    for _ in range(25):
        comm.all_reduce(chain)
        chain.mul_(1.0 / world)
    torch.cuda.synchronize()
    comm.check()
    # what parallel.one_shot runs before the learner may rely on the windows
    passed, why = comm.self_test(calls=6)
    # ... and what the learner's exchange would pick on this box, with the reason
    #     (TONIC_AMD_ALLREDUCE=auto: the ranks decide together)
    os.environ.setdefault('TONIC_AMD_ALLREDUCE', 'auto')
    picked = parallel.one_shot(11101)
    choice = parallel.allreduce_choice() or {}
    np.savez(out_path + f'.rank{rank}.npz', chain=chain.cpu().numpy(),
             device=np.int64(torch.cuda.current_device()),
             self_test=np.array([int(passed)]), self_test_reason=np.array(why),
             choice=np.array(choice.get('kind', '')), choice_reason=np.array(choice.get('reason', '')),
             picked=np.array([int(picked is not None)]),
             **{f'call{i}': r for i, r in enumerate(results)})
    if world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()
    comm.close()


if __name__ == '__main__':
    run(sys.argv[1])
