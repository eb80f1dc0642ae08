"""Per-call spread of he_pir_compute_response_to_query_device and of its parts (one index, 8 chunks of 256 x 64, N=8192
L=4): device milliseconds between events either side of every call, and the host milliseconds each call took to return.

    python bench_tools/whole_query_spread.py [alone|after_legs] [calls]

after_legs: the legs bench.py runs before it (path_bench.run_all's order) run first in the same process."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "swift-homomorphic-encryption_amd"), os.path.join(ROOT, "bench_tools")]
import torch  # noqa: E402

import heamd  # noqa: E402
import path_bench  # noqa: E402
from path_bench import _timed, _uniform  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "alone"
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 30
heamd.set_scratch_cache()
if mode == "after_legs":
    legs = path_bench.run_all(quick=False)
    print("whole query inside run_all:", json.dumps(legs["config5_pir_whole_query_1gpu"]))

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


def show(name, t):
    s = t.spread
    print(f"{name:28s} mean {t * 1e3:8.3f} ms  min {s['min_ms']:8.3f}  median {s['median_ms']:8.3f}  max {s['max_ms']:8.3f}"
          f"  calls {s['calls']}  host max {max(t.host_enqueue_ms):8.3f} ms")
    if s["max_ms"] > 1.5 * s["median_ms"]:
        print("   per call :", " ".join(f"{x:.2f}" for x in t.per_call_ms))
        print("   host     :", " ".join(f"{x:.2f}" for x in t.host_enqueue_ms))


show("expand", _timed(torch, lambda: ctx.pir_expand(query, total, galois), calls))
show("dim-0 forward NTT", _timed(torch, lambda: qctx.forward_ntt_(dim0), calls))
show("chunk loop", _timed(torch, lambda: ctx.pir_compute_response([d0, d1], dim0, expanded[d0:], database, chunks,
                                                                  relinearization_key=relin), calls))
for _ in range(3):
    show("whole query", _timed(torch, lambda: ctx.pir_compute_response_to_query([d0, d1], query, 1, galois, relin, database,
                                                                              chunks), calls))
