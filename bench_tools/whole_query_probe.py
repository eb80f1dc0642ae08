"""Where the time of he_pir_compute_response_to_query_device goes (one index, 8 chunks of 256 x 64, N=8192 L=4)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "swift-homomorphic-encryption_amd"), os.path.join(ROOT, "bench_tools")]
import torch  # noqa: E402

import heamd  # noqa: E402


heamd.set_scratch_cache()  # a server's setting: the library keeps its freed scratch (he_set_scratch_cache)
from path_bench import _timed, _uniform  # noqa: E402

degree, d0, d1, chunks = 8192, 256, 64, 8
q = heamd.generate_primes([55] * 5, False, degree)
ctx = heamd.BfvContext(degree, 557057, q)
qctx = ctx.ciphertext_context()
moduli = q[:-1]
total = d0 + d1
query = _uniform(torch, moduli, (1, 2), degree, 7)
elements = sorted({(degree >> level) + 1 for level in range((total - 1).bit_length())})
galois = {e: _uniform(torch, q, (ctx.L, 2), degree, 20 + i) for i, e in enumerate(elements)}
relin = _uniform(torch, q, (ctx.L, 2), degree, 10)
database = _uniform(torch, moduli, (chunks, d0 * d1), degree, 9)
expanded = ctx.pir_expand(query, total, galois)
dim0 = expanded[:d0].clone()
t_expand = _timed(torch, lambda: ctx.pir_expand(query, total, galois), 5)
t_ntt = _timed(torch, lambda: qctx.forward_ntt_(dim0), 5)
t_response = _timed(torch, lambda: ctx.pir_compute_response([d0, d1], dim0, expanded[d0:], database, chunks,
                                                            relinearization_key=relin), 5)
t_whole = _timed(torch, lambda: ctx.pir_compute_response_to_query([d0, d1], query, 1, galois, relin, database, chunks), 5)
print(f"expand {t_expand * 1e3:.3f} ms  dim-0 NTT {t_ntt * 1e3:.3f} ms  response {t_response * 1e3:.3f} ms  "
      f"sum {(t_expand + t_ntt + t_response) * 1e3:.3f} ms  whole call {t_whole * 1e3:.3f} ms")
