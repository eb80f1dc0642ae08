"""PIR dim-0 workload for rocprofv3 passes: argv[1] = plain | packed (256 rows x 64 columns, a few launches)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "swift-homomorphic-encryption_amd"), os.path.join(ROOT, "bench_tools")]
import torch  # noqa: E402

import heamd  # noqa: E402


heamd.set_scratch_cache()  # a server's setting: the library keeps its freed scratch (he_set_scratch_cache)
from path_bench import _uniform  # noqa: E402

packed = len(sys.argv) > 1 and sys.argv[1] == "packed"
degree = 8192
q = heamd.generate_primes([55] * 5, False, degree)
ctx = heamd.BfvContext(degree, 557057, q)
count, columns = 256, 64
cts = _uniform(torch, q[:-1], (count, 2), degree, 5)
pts = _uniform(torch, q[:-1], (columns, count), degree, 6)
database = ctx.pack_plaintexts(pts) if packed else pts
for _ in range(4):
    if packed:
        ctx.inner_product_plain_packed(cts, database, None, 2, columns)
    else:
        ctx.inner_product_plain(cts, database, None, 2, columns)
torch.cuda.synchronize()
