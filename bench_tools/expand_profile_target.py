"""PirUtil.expand (1 query ciphertext -> EXPAND_OUTPUTS outputs, default 1024; N = 8192, L = 4) for rocprofv3 --kernel-trace --stats."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "swift-homomorphic-encryption_amd"))
sys.path.insert(0, os.path.join(ROOT, "bench_tools"))
import torch  # noqa: E402

import heamd  # noqa: E402


heamd.set_scratch_cache()  # a server's setting: the library keeps its freed scratch (he_set_scratch_cache)
from path_bench import _timed, _uniform  # noqa: E402

degree = 8192
q = heamd.generate_primes([55] * 5, False, degree)
bfv = heamd.BfvContext(degree, 557057, q)
moduli = q[:-1]
elements = sorted({(degree >> level) + 1 for level in range(10)})
keys = {e: _uniform(torch, q, (bfv.L, 2), degree, 100 + i) for i, e in enumerate(elements)}
query = _uniform(torch, moduli, (1, 2), degree, 14)
outputs = int(os.environ.get("EXPAND_OUTPUTS", "1024"))
print("expand to %d outputs, ms:" % outputs, _timed(torch, lambda: bfv.pir_expand(query, outputs, keys), 5) * 1e3)
