"""A/B timing of the ct x ct + relinearize pipeline (BASELINE configs[2]) over library variants, one process each.

  python bench_tools/ab_c3.py [NAME ...]     production library plus lib/variants/libhe_amd_NAME.so, two rounds
"""
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "swift-homomorphic-encryption_amd")
VARIANTS = os.path.join(PKG, "lib", "variants")
TIMER = "import sys; sys.path[:0] = [%r, %r, %r]; import torch, heamd, path_bench, json; " \
        "print(json.dumps(path_bench.config3_ct_mul(torch, heamd, batch=1024, reps=5)))" % (
            ROOT, PKG, os.path.join(ROOT, "bench_tools"))


def main():
    names = sys.argv[1:]
    libs = {"production": None}
    for name in names:
        libs[name] = os.path.join(VARIANTS, f"libhe_amd_{name}.so")
    for round_index in range(2):
        for name, path in libs.items():
            env = dict(os.environ)
            if path:
                env["HEAMD_LIBRARY"] = path
            result = subprocess.run([sys.executable, "-c", TIMER], env=env, capture_output=True, text=True)
            if result.returncode != 0:
                print(f"round {round_index} {name}: FAILED {result.stderr[-300:]}")
                continue
            r = json.loads(result.stdout.strip().splitlines()[-1])
            print(f"round {round_index} {name:16s} ct x ct {r['ct_mul_per_s'] / 1e3:7.1f} k/s   relinearize "
                  f"{r['relinearize_per_s'] / 1e3:7.1f} k/s   both {r['ct_mul_relinearize_per_s'] / 1e3:7.1f} k/s", flush=True)


if __name__ == "__main__":
    main()
