"""What a cyclic garbage collection INSIDE a stream capture does when it finalizes device objects that earlier work
left behind (an older captured graph in a reference cycle, tensors): the failure mode `_lib.capturing` exists for.
usage: capture_gc_probe.py {old-graph|tensor|event}   (exits 0 when the capture survives)"""
import gc
import sys

import torch

kind = sys.argv[1] if len(sys.argv) > 1 else 'old-graph'
x = torch.zeros(1024, device='cuda')


class Holder:
    pass


def garbage():
    h = Holder()
    h.me = h                                    # a cycle: only the cyclic collector frees it
    if kind == 'old-graph':
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            x.add_(1)
        g.replay()
        h.graph = g
    elif kind == 'tensor':
        h.t = torch.empty(64 << 20, device='cuda')
    else:
        h.e = torch.cuda.Event()
        h.e.record()


gc.disable()
garbage()
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    x.add_(1)
    gc.collect()                                # (what an allocation inside the capture may trigger)
    x.add_(1)
graph.replay()
torch.cuda.synchronize()
print(kind, 'survived', float(x[0]))
