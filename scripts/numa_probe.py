"""Developer probe: does the collect loop depend on WHERE the trainer's thread runs?  Prints the GPU's
PCI address, its NUMA node and local CPUs, the CPU this process runs on, then the microseconds per
environment step of the bench workload's collect loop (a) as started, (b) pinned to the GPU's local
CPUs, (c) pinned to the other CPUs — a fresh agent (and shared block) each time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes
import numpy as np, torch
import bench

_libc = ctypes.CDLL(None)


def current_cpu():
    return _libc.sched_getcpu()

def parse(cpulist):
    cpus = set()
    for part in cpulist.strip().split(','):
        if '-' in part:
            a, b = part.split('-')
            cpus.update(range(int(a), int(b) + 1))
        elif part:
            cpus.add(int(part))
    return cpus

p = torch.cuda.get_device_properties(0)
address = '%04x:%02x:%02x.0' % (getattr(p, 'pci_domain_id', 0), p.pci_bus_id, p.pci_device_id)
base = '/sys/bus/pci/devices/' + address
node = open(base + '/numa_node').read().strip() if os.path.exists(base + '/numa_node') else '?'
local = parse(open(base + '/local_cpulist').read()) if os.path.exists(base + '/local_cpulist') else set()
everything = os.sched_getaffinity(0)
print('GPU', address, 'numa node', node, 'local cpus', len(local), 'of', len(everything),
      '| running on cpu', current_cpu(), 'local' if current_cpu() in local else 'REMOTE')

def measure(label, cpus, bind='0', threads_too=True):
    """cpus: where the process is when the agent (and its shared block) is built; bind: TONIC_AMD_NUMA_BIND for
    the build (the collector then moves the block's pages to the GPU's node; threads_too: the process as well)."""
    from tonic_amd import parallel
    os.environ['TONIC_AMD_NUMA_BIND'] = bind
    parallel._bound.clear()
    keep = parallel.bind_near_gpu
    if not threads_too:
        parallel.bind_near_gpu = lambda device: None
    if cpus:
        os.sched_setaffinity(0, cpus)            # the scheduler put the process there ...
        if bind == '1':
            os.sched_setaffinity(0, everything)  # ... and it may run anywhere (it stays until something moves it)
    agent, loop, rollout, out = bench.measure_job(256, 0, 1, 1, 0, True, device_too=False)
    parallel.bind_near_gpu = keep
    loop.run(bench.T - agent.replay.index)
    torch.cuda.synchronize()
    times = []
    for _ in range(3):
        loop.run(64)
        t0 = time.perf_counter()
        loop.run(1024)
        times.append((time.perf_counter() - t0) / 1024 * 1e6)
    print(label, '| running on cpu', current_cpu(), 'local' if current_cpu() in local else 'REMOTE',
          '| us per environment step', ' '.join(f'{t:.2f}' for t in times), flush=True)
    agent.close()
    os.sched_setaffinity(0, everything)


if local and local != everything:
    far = everything - local
    measure('started on the far socket, nothing bound            ', far, '0')
    measure('started on the far socket, block pages moved        ', far, '1', threads_too=False)
    measure('started on the far socket, pages + process bound    ', far, '1')
    measure('started on the GPU\'s socket                         ', local & everything, '0')
    measure('started on the far socket, nothing bound (again)    ', far, '0')
else:
    measure('one NUMA node: as started', None)
