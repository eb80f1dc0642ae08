"""ct x ct and relinearize on the reference's predefined parameter sets with 60-bit moduli (EncryptionParameters.swift:257-263:
n_8192_logq_29_60_60, n_8192_logq_40_60_60, n_4096 with 60-bit pairs) next to the BASELINE ring, 1024 ciphertext pairs each.

  python bench_tools/param_sets_bench.py            (on the GPU box; HEAMD_LIBRARY selects a variant library)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "swift-homomorphic-encryption_amd"), os.path.join(ROOT, "bench_tools")]

import torch  # noqa: E402

import heamd  # noqa: E402
from path_bench import _timed, _uniform  # noqa: E402


def run(degree, bits, batch=1024, reps=5):
    q = heamd.generate_primes(bits, False, degree)
    ctx = heamd.BfvContext(degree, 557057, q)
    moduli = q[:-1]
    lhs, rhs = _uniform(torch, moduli, (batch, 2), degree, 1), _uniform(torch, moduli, (batch, 2), degree, 2)
    key = _uniform(torch, q, (ctx.L, 2), degree, 3)
    state = {}

    def mul():
        state["ct3"] = ctx.mul(lhs, rhs)

    def relin():
        ctx.relinearize(state["ct3"], key)

    return batch / _timed(torch, mul, reps), batch / _timed(torch, relin, reps)


if __name__ == "__main__":
    heamd.set_scratch_cache()
    out = []
    # (N = 16384, L = 6: EncryptionParameters.swift:200-206 allows the degree; 256 pairs = 3.6 GB of workspace)
    for degree, bits, batch in [(8192, [29, 60, 60], 1024), (8192, [40, 60, 60], 1024), (4096, [60, 60, 60], 1024),
                                (8192, [55, 55, 55, 55, 55], 1024), (16384, [55] * 7, 256), (32768, [55] * 7, 128)]:
        mul, relin = run(degree, bits, batch)
        out.append("N=%d %s ct x ct %.1f k/s  relinearize %.1f k/s" % (degree, bits, mul / 1e3, relin / 1e3))
    print(" | ".join(out))
