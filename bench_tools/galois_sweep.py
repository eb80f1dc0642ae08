"""Bfv.applyGalois (out of place: the fused Galois key switch) and relinearize per batch size, N = 8192, L = 4: where the key
switch's end in the key-MAC transform's store starts to pay against the separate finish kernel (HEAMD_LIBRARY selects the
variant library).   python bench_tools/galois_sweep.py"""
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
out = []
for batch in (16, 32, 52, 64, 96, 128, 192, 256, 512, 1024):
    ct = _uniform(torch, moduli, (batch, 2), degree, 1)
    ct3 = _uniform(torch, moduli, (batch, 3), degree, 2)
    t_g = _timed(torch, lambda: ctx.apply_galois(ct, 2 * degree - 1, key), 20)
    t_r = _timed(torch, lambda: ctx.relinearize(ct3, key), 20)
    out.append("%4d: galois %6.1f us (%5.0f k/s)  relinearize %6.1f us (%5.0f k/s)" % (batch, t_g * 1e6, batch / t_g / 1e3, t_r * 1e6, batch / t_r / 1e3))
print("\n".join(out))
