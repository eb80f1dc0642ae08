"""A/B of whole workloads (bench.py --workload ...) over library variants, one process each, two rounds.

  python bench_tools/ab_workloads.py c3,c4,c5 [NAME ...]     production + lib/variants/libhe_amd_NAME.so
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = os.path.join(ROOT, "swift-homomorphic-encryption_amd", "lib", "variants")


def main():
    workloads = sys.argv[1].split(",")
    libs = {"production": None}
    for name in sys.argv[2:]:
        libs[name] = os.path.join(VARIANTS, f"libhe_amd_{name}.so")
    for round_index in range(2):
        for workload in workloads:
            for name, path in libs.items():
                env = dict(os.environ)
                if path:
                    env["HEAMD_LIBRARY"] = path
                result = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", workload, "--steps",
                                         "8", "--warmup", "2", "--no-cpu-baseline", "--skip-other-configs"], env=env,
                                        capture_output=True, text=True)
                if result.returncode != 0:
                    print(f"round {round_index} {workload} {name}: FAILED {result.stderr[-300:]}")
                    continue
                r = json.loads(result.stdout.strip().splitlines()[-1])
                extra = {k: round(v) for k, v in r["extras"].items() if isinstance(v, float) and "per_s" in k}
                print(f"round {round_index} {workload} {name:14s} value {r['value']:14.1f} {r['unit']}  "
                      f"roofline frac {r['roofline']['frac']:.3f}  launch {r['roofline']['avg_launch_ms']:.4f} ms  {extra}",
                      flush=True)


if __name__ == "__main__":
    main()
