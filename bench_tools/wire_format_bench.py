"""Timings of the wire-format rows for the library named by HEAMD_LIBRARY (default: production).

    python bench_tools/wire_format_bench.py            one line: seeded polynomials / s (batch 2048 and 1), serialize and
                                                       deserialize polynomials / s (N = 8192, L = 4, 55-bit moduli)
    python bench_tools/wire_format_bench.py ab NAME..  the same for lib/variants/libhe_amd_NAME.so, two rounds
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "swift-homomorphic-encryption_amd"))
sys.path.insert(0, os.path.join(ROOT, "bench_tools"))


def measure():
    import torch

    import heamd


    heamd.set_scratch_cache()  # a server's setting: the library keeps its freed scratch (he_set_scratch_cache)
    from path_bench import _timed, _uniform

    degree = 8192
    moduli = heamd.generate_primes([55] * 4, False, degree)
    poly = heamd.PolyContext(degree, moduli)
    out = {}
    for batch in (2048, 64, 1):
        seeds = torch.randint(0, 256, (batch, 32), dtype=torch.uint8, device="cuda")
        t = _timed(torch, lambda: poly.random_from_seeds(seeds), 10)
        out[f"seeded_batch{batch}_polys_per_s"] = batch / t
        out[f"seeded_batch{batch}_ms"] = t * 1e3
    batch = 2048
    slab = _uniform(torch, moduli, (batch,), degree, 3)
    packed = poly.serialize(slab)
    t = _timed(torch, lambda: poly.serialize(slab), 10)
    bytes_per_poly = slab[0].numel() * 8 + packed[0].numel()
    out["serialize_polys_per_s"] = batch / t
    out["serialize_frac_of_8TBps"] = batch / t * bytes_per_poly / 8e12
    t = _timed(torch, lambda: poly.deserialize(packed), 10)
    out["deserialize_polys_per_s"] = batch / t
    out["deserialize_frac_of_8TBps"] = batch / t * bytes_per_poly / 8e12
    return out


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "ab":
        variants = os.path.join(ROOT, "swift-homomorphic-encryption_amd", "lib", "variants")
        libs = {"production": None}
        for name in sys.argv[2:]:
            libs[name] = os.path.join(variants, f"libhe_amd_{name}.so")
        for round_index in range(2):
            for name, path in libs.items():
                env = dict(os.environ)
                if path:
                    env["HEAMD_LIBRARY"] = path
                r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True)
                line = r.stdout.strip().splitlines()[-1] if r.returncode == 0 and r.stdout.strip() else "FAILED " + r.stderr[-300:]
                print(f"round {round_index} {name:12s} {line}", flush=True)
        return
    print(json.dumps({k: round(v, 4) for k, v in measure().items()}))


if __name__ == "__main__":
    main()
