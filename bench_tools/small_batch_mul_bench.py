"""ct x ct on small batches (the tail of a PIR response multiplies a handful of ciphertexts): microseconds per call.
python bench_tools/small_batch_mul_bench.py      (HEAMD_LIBRARY selects a variant library)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "swift-homomorphic-encryption_amd"), os.path.join(ROOT, "bench_tools")]
import torch  # noqa: E402

import heamd  # noqa: E402
from path_bench import _timed, _uniform  # noqa: E402

heamd.set_scratch_cache()
out = []
for degree, bits in [(8192, [55] * 5), (4096, [60, 60, 60])]:
    q = heamd.generate_primes(bits, False, degree)
    ctx = heamd.BfvContext(degree, 557057, q)
    for batch in (1, 2, 4, 8, 16, 28, 29, 64):
        lhs, rhs = _uniform(torch, q[:-1], (batch, 2), degree, 1), _uniform(torch, q[:-1], (batch, 2), degree, 2)
        seconds = _timed(torch, lambda: ctx.mul(lhs, rhs), 20)
        out.append("N=%d batch %3d: %7.1f us" % (degree, batch, seconds * 1e6))
print("\n".join(out))
