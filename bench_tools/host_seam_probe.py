"""The host-pointer seam (he_ntt_forward / he_ntt_inverse on a pageable host slab) timed per call: upload + transform +
download + synchronise, in place on a buffer that stays allocated (the reference's benchmark loop times exactly that call,
PolyBenchmark.swift:148-158).  python bench_tools/host_seam_probe.py [polys ...]   (HEAMD_LIBRARY selects a variant)"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "swift-homomorphic-encryption_amd")]
import numpy as np  # noqa: E402

import heamd  # noqa: E402

degree, count = 8192, 4
moduli = heamd.generate_primes([55] * count, False, degree)
ctx = heamd.PolyContext(degree, moduli)
rng = np.random.default_rng(3)
for polys in [int(a) for a in sys.argv[1:]] or [16, 64, 256, 1024]:
    host = np.ascontiguousarray(np.stack([rng.integers(0, q, size=(polys, degree), dtype=np.uint64) for q in moduli], axis=1))
    original = host.copy()
    ctx.forward_ntt_host_(host)
    ctx.inverse_ntt_host_(host)
    assert np.array_equal(host, original), "round trip through the host seam"
    times = []
    for _ in range(8):
        t0 = time.perf_counter()
        ctx.forward_ntt_host_(host)
        times.append(time.perf_counter() - t0)
    best, median = min(times), sorted(times)[len(times) // 2]
    print(f"{polys:5d} polynomials ({host.nbytes >> 20:4d} MiB): median {median * 1e3:7.3f} ms = {2 * host.nbytes / median / 1e9:6.1f} GB/s of host "
          f"traffic, {polys / median / 1e3:7.1f} k poly-NTT/s; best {best * 1e3:7.3f} ms = {2 * host.nbytes / best / 1e9:6.1f} GB/s")
