"""Fixed ct x ct + relinearize workload (BASELINE configs[2], batch 1024) for rocprofv3 --kernel-trace --stats."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "swift-homomorphic-encryption_amd"))
sys.path.insert(0, os.path.join(ROOT, "bench_tools"))
os.environ.setdefault("HEAMD_RECORDED_RATES", "1")  # no tool in a child process under the profiler
import torch  # noqa: E402

import heamd  # noqa: E402


heamd.set_scratch_cache()  # a server's setting: the library keeps its freed scratch (he_set_scratch_cache)
import path_bench  # noqa: E402

print(path_bench.config3_ct_mul(torch, heamd, batch=1024, reps=3))
