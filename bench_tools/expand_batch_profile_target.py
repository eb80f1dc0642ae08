"""PirUtil.expand batched over queries: single-query and 16-query rates at two output counts (N = 8192, L = 4)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "swift-homomorphic-encryption_amd"), os.path.join(ROOT, "bench_tools")]
import torch  # noqa: E402

import heamd  # noqa: E402


heamd.set_scratch_cache()  # a server's setting: the library keeps its freed scratch (he_set_scratch_cache)
from path_bench import _timed, _uniform  # noqa: E402

degree = 8192
q = heamd.generate_primes([55] * 5, False, degree)
bfv = heamd.BfvContext(degree, 557057, q)
moduli = q[:-1]
elements = sorted({(degree >> level) + 1 for level in range(10)})
key_sets = [{e: _uniform(torch, q, (bfv.L, 2), degree, 100 * (c + 1) + i) for i, e in enumerate(elements)} for c in range(2)]
for outputs in (128, 320, 1024):
    single = _uniform(torch, moduli, (1, 2), degree, 14)
    t1 = _timed(torch, lambda: bfv.pir_expand(single, outputs, key_sets[0]), 5)
    for queries in (4, 16):
        batch = _uniform(torch, moduli, (queries, 1, 2), degree, 15)
        per_query = [key_sets[(i // 4) % 2] for i in range(queries)]
        tq = _timed(torch, lambda: bfv.pir_expand_batch(batch, outputs, per_query), 3)
        print(f"expand to {outputs:5d} outputs: 1 query {t1 * 1e3:7.3f} ms | {queries:2d} queries in one call "
              f"{tq * 1e3:8.3f} ms = {tq / queries * 1e3:7.3f} ms per query, {t1 * queries / tq:5.2f} x the single-query rate",
              flush=True)
