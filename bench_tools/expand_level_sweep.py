"""One level of PirUtil.expand (every query expands 1 -> 2 outputs: Q parents, one Galois key switch each, children written
to their output slots) per number of queries in the call, N = 8192, L = 4, one key for all -- where the expand end in the
key-MAC transform's store starts to pay (HEAMD_LIBRARY selects the variant library).   python bench_tools/expand_level_sweep.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "swift-homomorphic-encryption_amd"), os.path.join(ROOT, "bench_tools")]
import torch  # noqa: E402

import heamd  # noqa: E402
from path_bench import _timed, _uniform  # noqa: E402

heamd.set_scratch_cache()
degree = 8192
q = heamd.generate_primes([55] * 5, False, degree)
ctx = heamd.BfvContext(degree, 557057, q)
moduli = q[:-1]
key = _uniform(torch, q, (ctx.L, 2), degree, 3)
keys = {degree + 1: key}
for queries in (32, 64, 128, 256, 512, 1024, 2048, 4096):
    cts = _uniform(torch, moduli, (queries, 1, 2), degree, 1)
    t = _timed(torch, lambda: ctx.pir_expand_batch(cts, 2, [keys] * queries), 10)
    print("%5d parents: %7.1f us  (%5.0f k key switches/s)" % (queries, t * 1e6, queries / t / 1e3))
