"""Fixed mod-switch workload (BASELINE configs[3]: N=16384, 6 -> 5 moduli, 8192 polynomials) for rocprofv3 passes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "swift-homomorphic-encryption_amd"), os.path.join(ROOT, "bench_tools")]
import torch  # noqa: E402

import heamd  # noqa: E402


heamd.set_scratch_cache()  # a server's setting: the library keeps its freed scratch (he_set_scratch_cache)
import path_bench  # noqa: E402

print(path_bench.config4_mod_switch(torch, heamd, batch=8192, reps=3))
