"""PIR chunk loop for a Bfv<UInt32> parameter set (n_4096_logq_27_28_28): packed 4-byte slabs
(he_pir_compute_response_device_u32) against the same context on 8-byte slabs (he_pir_compute_response_device)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "swift-homomorphic-encryption_amd"), os.path.join(ROOT, "bench_tools")]
import torch  # noqa: E402

import heamd  # noqa: E402


heamd.set_scratch_cache()  # a server's setting: the library keeps its freed scratch (he_set_scratch_cache)
from path_bench import _timed  # noqa: E402
from word32_scheme_bench import DEGREE, Q, T, uniform32  # noqa: E402

d0, d1, chunks = 256, 64, 32
moduli = Q[:-1]
narrow = heamd.BfvContext32(DEGREE, T, Q)
wide = heamd.BfvContext(DEGREE, T, Q, word_bits=32)
dim0, rest = uniform32(moduli, (d0, 2), 1), uniform32(moduli, (d1, 2), 2)
database = uniform32(moduli, (chunks, d0 * d1), 3)
key = uniform32(Q, (narrow.L, 2), 4)
out = {"parameters": "n_4096_logq_27_28_28", "dimensions": [d0, d1], "chunks": chunks}
t32 = _timed(torch, lambda: narrow.pir_compute_response([d0, d1], dim0, rest, database, chunks, relinearization_key=key), 3)
got32 = narrow.pir_compute_response([d0, d1], dim0, rest, database, chunks, relinearization_key=key)
args64 = [x.to(torch.int64) for x in (dim0, rest, database, key)]
t64 = _timed(torch, lambda: wide.pir_compute_response([d0, d1], args64[0], args64[1], args64[2], chunks,
                                                      relinearization_key=args64[3]), 3)
got64 = wide.pir_compute_response([d0, d1], args64[0], args64[1], args64[2], chunks, relinearization_key=args64[3])
out["same_words"] = bool((got32.to(torch.int64) == got64).all())
macs = chunks * d0 * d1
out["packed_4_byte"] = {"ms_per_chunk": t32 / chunks * 1e3, "ct_pt_mac_per_s": macs / t32,
                        "database_GBps": macs * len(moduli) * DEGREE * 4 / t32 / 1e9}
out["on_8_byte_slabs"] = {"ms_per_chunk": t64 / chunks * 1e3, "ct_pt_mac_per_s": macs / t64,
                          "database_GBps": macs * len(moduli) * DEGREE * 8 / t64 / 1e9}
# several queries in one call share the pass over the 4-byte database (he_pir_compute_response_queries_device_u32)
for queries in (2, 4):
    side = uniform32(moduli, (d0, queries, 2), 10 + queries)
    rests = uniform32(moduli, (queries, d1, 2), 20 + queries)
    keys = [key] * queries
    tq = _timed(torch, lambda: narrow.pir_compute_response_queries([d0, d1], side, rests, database, chunks, keys), 3)
    out[f"packed_4_byte_{queries}_queries"] = {"ms_per_chunk_per_query": tq / chunks / queries * 1e3,
                                               "chunk_responses_per_s": chunks * queries / tq,
                                               "ct_pt_mac_per_s": macs * queries / tq}
print(json.dumps(out, indent=1))
