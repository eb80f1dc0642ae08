"""Profile target: the 4-byte NTT kernels alone, N=4096, 3 moduli (27/28/28 bits), 16384 polynomials."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "swift-homomorphic-encryption_amd"), os.path.join(ROOT, "bench_tools")]
import torch  # noqa: E402

import heamd  # noqa: E402


heamd.set_scratch_cache()  # a server's setting: the library keeps its freed scratch (he_set_scratch_cache)
from word32_scheme_bench import DEGREE, Q, uniform32  # noqa: E402

moduli = Q[:-1]
ctx = heamd.PolyContext(DEGREE, moduli)
x = uniform32(moduli, (16384,), 5)
for _ in range(3):
    ctx.forward_ntt_u32_(x)
    ctx.inverse_ntt_u32_(x)
torch.cuda.synchronize()
