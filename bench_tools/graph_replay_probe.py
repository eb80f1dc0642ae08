"""Does a whole PIR chunk response (scratch allocations included) capture into a HIP graph, and what does replaying it save?"""
import sys, time
import os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "swift-homomorphic-encryption_amd"), os.path.join(ROOT, "bench_tools")]
import torch, heamd
from path_bench import _uniform, _timed
degree = 8192
q = heamd.generate_primes([55] * 5, False, degree)
ctx = heamd.BfvContext(degree, 557057, q)
moduli = q[:-1]
d0, d1, chunks = 256, 64, 1
dim0 = _uniform(torch, moduli, (d0, 2), degree, 7)
rest = _uniform(torch, moduli, (d1, 2), degree, 8)
database = _uniform(torch, moduli, (chunks, d0 * d1), degree, 9)
key = _uniform(torch, q, (ctx.L, 2), degree, 10)
want = ctx.pir_compute_response([d0, d1], dim0, rest, database, chunks, relinearization_key=key)
torch.cuda.synchronize()
t_direct = _timed(torch, lambda: ctx.pir_compute_response([d0, d1], dim0, rest, database, chunks, relinearization_key=key), 10)
try:
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = ctx.pir_compute_response([d0, d1], dim0, rest, database, chunks, relinearization_key=key)
    graph.replay(); torch.cuda.synchronize()
    print("graph equal:", bool((out == want).all()))
    t_graph = _timed(torch, lambda: graph.replay(), 10)
    print(f"direct {t_direct*1e3:.3f} ms  graph {t_graph*1e3:.3f} ms")
except Exception as e:
    print("capture failed:", repr(e)[:500])
    print(f"direct {t_direct*1e3:.3f} ms")
