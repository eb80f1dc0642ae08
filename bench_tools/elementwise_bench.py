"""PolyRq element-wise operators on one GPU: N = 8192, L = 4, 4096 polynomials (1 GiB operands)."""
import json
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

degree, batch = 8192, 4096
moduli = heamd.generate_primes([55] * 4, False, degree)
ctx = heamd.PolyContext(degree, moduli)
x = _uniform(torch, moduli, (batch,), degree, 1)
y = _uniform(torch, moduli, (batch,), degree, 2)
poly_bytes = 4 * degree * 8
scalars = [m // 3 for m in moduli]
for name, fn, streams in (("add", lambda: ctx.add_(x, y), 3), ("sub", lambda: ctx.sub_(x, y), 3),
                          ("neg", lambda: ctx.neg_(x), 2), ("mul (Eval)", lambda: ctx.mul_(x, y), 3),
                          ("mul by scalar", lambda: ctx.mul_scalar_(x, scalars), 2)):
    t = _timed(torch, fn, 10)
    print(json.dumps({"op": name, "poly_per_s": batch / t, "ms": t * 1e3, "GBps": streams * poly_bytes * batch / t / 1e9,
                      "frac_of_8TBps": streams * poly_bytes * batch / t / 8e12}), flush=True)
