"""Profile target: Bfv<UInt32> ct x ct + relinearize on packed 4-byte slabs (n_4096_logq_27_28_28), 2048 pairs."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "swift-homomorphic-encryption_amd"), os.path.join(ROOT, "bench_tools")]
import torch  # noqa: E402

import heamd  # noqa: E402


heamd.set_scratch_cache()  # a server's setting: the library keeps its freed scratch (he_set_scratch_cache)
from word32_scheme_bench import DEGREE, Q, T, uniform32  # noqa: E402

ctx = heamd.BfvContext32(DEGREE, T, Q)
moduli = Q[:-1]
pairs = 2048
lhs, rhs = uniform32(moduli, (pairs, 2), 2), uniform32(moduli, (pairs, 2), 3)
key = uniform32(Q, (ctx.L, 2), 4)
for _ in range(10):
    ctx.relinearize(ctx.mul(lhs, rhs), key)
torch.cuda.synchronize()
