"""Device-resident timings of the SURVEY.md 8(f) "next" rows on one GPU (N = 8192, L = 4, 55-bit moduli unless noted).

    python bench_tools/next_rows_bench.py

One JSON object per row: units per second and the algorithmic bytes per unit the HBM figure is computed from
(bytes a unit must read + write once).  Values are synthetic uniform words; keys are uniform words of the right shape.
"""
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

DEGREE = 8192
T = 557057
Q = heamd.generate_primes([55] * 5, False, DEGREE)
MODULI = Q[:-1]
L = len(MODULI)
POLY_BYTES = L * DEGREE * 8


def report(row, name, units, seconds, bytes_per_unit, note=""):
    rate = units / seconds
    print(json.dumps({"row": row, "op": name, "units_per_s": rate, "ms": seconds * 1e3,
                      "algorithmic_bytes_per_unit": bytes_per_unit, "GBps": rate * bytes_per_unit / 1e9,
                      "frac_of_8TBps": rate * bytes_per_unit / 8e12, "note": note}), flush=True)


def main():
    bfv = heamd.BfvContext(DEGREE, T, Q)
    poly = heamd.PolyContext(DEGREE, MODULI)

    # ---- N2: Galois automorphisms and the Galois key switch, query expansion
    batch = 2048
    slab = _uniform(torch, MODULI, (batch,), DEGREE, 11)
    for eval_format, label in ((False, "Coeff"), (True, "Eval")):
        t = _timed(torch, lambda: poly.apply_galois(slab, 3, eval_format), 10)
        report("N2", f"PolyRq.applyGalois ({label}), per polynomial", batch, t, 2 * POLY_BYTES)
    t = _timed(torch, lambda: poly.multiply_power_of_x(slab, -5), 10)
    report("N2", "PolyRq.multiplyPowerOfX, per polynomial", batch, t, 2 * POLY_BYTES)
    cts = _uniform(torch, MODULI, (1024, 2), DEGREE, 12)
    key = _uniform(torch, Q, (L, 2), DEGREE, 13)
    ws = torch.empty(bfv.apply_galois_workspace_bytes(1024) // 8, dtype=torch.int64, device="cuda")
    t = _timed(torch, lambda: bfv.apply_galois(cts, 3, key, workspace=ws), 5)
    report("N2", "Bfv.applyGalois (permute + key switch), per ciphertext", 1024, t, 4 * POLY_BYTES,
           "compulsory bytes: ciphertext in + out; the key (2.6 MB) is shared by the batch")
    outputs = 1024  # one query ciphertext -> 1024 indicator ciphertexts: log2 = 10 levels, 1023 key switches
    elements = sorted({(DEGREE >> level) + 1 for level in range(10)})
    keys = {e: _uniform(torch, Q, (L, 2), DEGREE, 100 + i) for i, e in enumerate(elements)}
    query = _uniform(torch, MODULI, (1, 2), DEGREE, 14)
    t = _timed(torch, lambda: bfv.pir_expand(query, outputs, keys), 3)
    report("N2", "PirUtil.expand 1 -> 1024 ciphertexts, per output ciphertext", outputs, t, 2 * POLY_BYTES,
           "1023 Galois key switches, level-batched; bytes: the output ciphertexts")

    # the same for 16 queries at once (two clients' keys alternate): every level is one batch over all queries
    key_sets = [keys, {e: _uniform(torch, Q, (L, 2), DEGREE, 200 + i) for i, e in enumerate(elements)}]
    batch_queries = _uniform(torch, MODULI, (16, 1, 2), DEGREE, 15)
    per_query = [key_sets[(i // 4) % 2] for i in range(16)]
    t16 = _timed(torch, lambda: bfv.pir_expand_batch(batch_queries, outputs, per_query), 3)
    report("N2", "PirUtil.expand, 16 queries x 1024 outputs in one call, per output ciphertext", 16 * outputs, t16,
           2 * POLY_BYTES, "%.1f x the single-query rate" % ((16 * outputs / t16) / (outputs / t)))

    # ---- N3: wire format and seeded polynomials
    t = _timed(torch, lambda: poly.serialize(slab), 10)
    packed = poly.serialization_byte_count()
    report("N3", "PolyRq.serialize (55-bit packing), per polynomial", batch, t, POLY_BYTES + packed)
    blob = poly.serialize(slab)
    t = _timed(torch, lambda: poly.deserialize(blob), 10)
    report("N3", "PolyRq(deserialize:), per polynomial", batch, t, POLY_BYTES + packed)
    seeds = torch.randint(0, 256, (batch, 32), dtype=torch.uint8, device="cuda")
    t = _timed(torch, lambda: poly.random_from_seeds(seeds), 3)
    report("N3", "seeded polynomial (NIST CTR_DRBG AES-128, 128 bits per coefficient), per polynomial", batch, t, POLY_BYTES,
           "AES-bound: re-key chain on 8 lanes per seed, then one wavefront per 4 KiB chunk")

    # ---- N4: plaintext lift / unlift and scaleAndRound
    values = torch.randint(0, T, (batch, DEGREE), dtype=torch.int64, device="cuda")
    t = _timed(torch, lambda: bfv.plaintext_to_eval(values), 10)
    report("N4", "Plaintext.convertToEvalFormat (lift + forward NTT), per plaintext", batch, t, DEGREE * 8 + POLY_BYTES)
    lifted = bfv.plaintext_to_eval(values)
    t = _timed(torch, lambda: bfv.plaintext_to_coeff(lifted), 10)
    report("N4", "Plaintext.convertToCoeffFormat (inverse NTT + unlift), per plaintext", batch, t,
           DEGREE * 8 + POLY_BYTES)
    t = _timed(torch, lambda: bfv.scale_and_round(slab), 10)
    report("N4", "_RnsTool.scaleAndRound, per polynomial", batch, t, POLY_BYTES + DEGREE * 8)


if __name__ == "__main__":
    main()
