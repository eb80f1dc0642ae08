"""One PIR chunk response (256 x 64 database, N = 8192, L = 4) for rocprofv3 --kernel-trace --stats."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "swift-homomorphic-encryption_amd"))
sys.path.insert(0, os.path.join(ROOT, "bench_tools"))
import torch  # noqa: E402

import heamd  # noqa: E402


heamd.set_scratch_cache()  # a server's setting: the library keeps its freed scratch (he_set_scratch_cache)
import path_bench  # noqa: E402

print(path_bench.config5_pir_chunk(torch, heamd, d0=256, d1=64, reps=5))
