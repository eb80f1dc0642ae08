"""Fixed PIR dim-0 workload (the per-GPU shard of BASELINE configs[4]: 1024 rows x 128 columns) for rocprofv3 passes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "swift-homomorphic-encryption_amd"), os.path.join(ROOT, "bench_tools")]
import torch  # noqa: E402

import heamd  # noqa: E402


heamd.set_scratch_cache()  # a server's setting: the library keeps its freed scratch (he_set_scratch_cache)
import path_bench  # noqa: E402

print(path_bench.config5_inner_product(torch, heamd, count=1024, columns=128, reps=2))
