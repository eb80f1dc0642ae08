"""Bfv<UInt32> on packed 4-byte slabs, the reference's n_4096_logq_27_28_28 parameter sets
(EncryptionParameters.swift:313-345: N = 4096, q = 2^27 - 40959, 2^28 - 65535 | 2^28 - 73727 as the key-switching
modulus): NTT, ct x ct + relinearize and the ct x pt inner loop, device-resident, HIP events.

  python bench_tools/word32_scheme_bench.py
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "swift-homomorphic-encryption_amd"), os.path.join(ROOT, "bench_tools")]
import torch  # noqa: E402

import heamd  # noqa: E402


heamd.set_scratch_cache()  # a server's setting: the library keeps its freed scratch (he_set_scratch_cache)
from path_bench import _timed  # noqa: E402

DEGREE = 4096
Q = [(1 << 27) - 40959, (1 << 28) - 65535, (1 << 28) - 73727]
T = (1 << 16) + 1  # n_4096_logq_27_28_28_logt_17


def uniform32(moduli, prefix, seed):
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    bound = torch.tensor(moduli, dtype=torch.int64, device="cuda").view(*([1] * len(prefix)), len(moduli), 1)
    x = torch.randint(0, 1 << 40, tuple(prefix) + (len(moduli), DEGREE), dtype=torch.int64, device="cuda", generator=gen)
    return (x % bound).to(torch.int32)


def main():
    ctx = heamd.BfvContext32(DEGREE, T, Q)
    moduli = Q[:-1]
    L = ctx.L
    out = {"parameters": "n_4096_logq_27_28_28_logt_17 (UInt32 words)", "degree": DEGREE, "moduli": Q}
    poly_ctx = heamd.PolyContext(DEGREE, moduli)
    batch = 16384
    slab = uniform32(moduli, (batch,), 1)
    for name, fn in (("forward", poly_ctx.forward_ntt_u32_), ("inverse", poly_ctx.inverse_ntt_u32_)):
        t = _timed(torch, lambda: fn(slab), 20)
        bytes_per = 2 * L * DEGREE * 4
        out[f"{name}_poly_ntt_per_s"] = batch / t
        out[f"{name}_frac_of_8TBps"] = bytes_per * batch / t / 8e12
    pairs = 2048
    lhs, rhs = uniform32(moduli, (pairs, 2), 2), uniform32(moduli, (pairs, 2), 3)
    key = uniform32(Q, (L, 2), 4)
    state = {}

    def mul():
        state["p"] = ctx.mul(lhs, rhs)

    def relin():
        ctx.relinearize(state["p"], key)

    mul()
    t_mul, t_relin = _timed(torch, mul, 5), _timed(torch, relin, 5)
    out["ct_mul_per_s"] = pairs / t_mul
    out["relinearize_per_s"] = pairs / t_relin
    out["ct_mul_relinearize_per_s"] = pairs / (t_mul + t_relin)
    count, columns = 1024, 256
    cts = uniform32(moduli, (count, 2), 5)
    pts = uniform32(moduli, (columns, count), 6)
    t = _timed(torch, lambda: ctx.inner_product_plain_resident(cts, pts, None, 2, columns), 3)
    db_bytes = count * columns * L * DEGREE * 4
    out["ct_pt_mac_per_s"] = count * columns / t
    out["database_GBps"] = db_bytes / t / 1e9
    out["database_frac_of_8TBps"] = db_bytes / t / 8e12
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
