"""Build container only (/root/reference present, no GPU needed): the CPU baseline bench.py reports
on the GPU box is oracle/torch_port.py (kind "port") because the reference checkout does not exist
there.  This script times the port AND the unmodified reference side by side on this container's
cores — the same bounded sample bench.cpu_measure takes (64 act + store steps at W = 256, the
full-size evaluate + lambda-returns, 3 full-batch iterations at N = 1 048 576), alternating, three
rounds — and writes profiles/r04_cpu_port_vs_reference.json: how far the port's SPEED is from the
reference's."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                     # noqa: E402
import torch                                     # noqa: E402


def main():
    engines = {e.kind: e for e in bench.cpu_engines()}
    assert 'reference' in engines, 'needs the reference checkout'
    threads = torch.get_num_threads()
    rounds = []
    for _ in range(3):
        for kind in ('reference', 'port'):
            rounds.append(bench.cpu_measure(engines[kind], bench.O, bench.A, bench.W, bench.T,
                                            threads, 64, 3))
            print(json.dumps(rounds[-1]), flush=True)

    def best(kind, key):
        return min(r['seconds'][key] for r in rounds if r['kind'] == kind)
    summary = {}
    for key in ('per_env_step', 'evaluate_and_gae', 'per_iteration', 'cycle'):
        ref, port = best('reference', key), best('port', key)
        summary[key] = dict(reference_s=ref, port_s=port, port_over_reference=round(port / ref, 3))
    out = dict(what='oracle/torch_port.py vs the unmodified reference (tonic.torch.agents.PPO), '
                    'CPU, PPO HalfCheetah shapes W=256 T=4096 N=1048576; best of 3 alternating rounds',
               cpu_count=os.cpu_count(), torch_threads=threads, torch=torch.__version__,
               summary=summary, rounds=rounds)
    with open(os.path.join(ROOT, 'profiles', 'r04_cpu_port_vs_reference.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == '__main__':
    main()
