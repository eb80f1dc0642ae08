"""Profile target: the dim-0 ct x pt inner product with `queries` queries side by side (argv[1], default 4)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "swift-homomorphic-encryption_amd"), os.path.join(ROOT, "bench_tools")]
import torch  # noqa: E402

import heamd  # noqa: E402


heamd.set_scratch_cache()  # a server's setting: the library keeps its freed scratch (he_set_scratch_cache)
from path_bench import _uniform  # noqa: E402

queries = int(sys.argv[1]) if len(sys.argv) > 1 else 4
degree = 8192
q = heamd.generate_primes([55] * 5, False, degree)
ctx = heamd.BfvContext(degree, 557057, q)
count, columns = 256, 64
cts = _uniform(torch, q[:-1], (count, 2 * queries), degree, 5)
pts = _uniform(torch, q[:-1], (columns, count), degree, 6)
for _ in range(4):
    ctx.inner_product_plain(cts, pts, None, 2 * queries, columns)
torch.cuda.synchronize()
