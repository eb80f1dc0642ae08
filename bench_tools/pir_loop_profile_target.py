"""The PIR chunk loop (8 chunks of a 256 x 64 database, N = 8192, L = 4, one query) for rocprofv3 --kernel-trace."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "swift-homomorphic-encryption_amd"), os.path.join(ROOT, "bench_tools")]
import torch  # noqa: E402

import heamd  # noqa: E402
import path_bench  # noqa: E402

heamd.set_scratch_cache()
print(path_bench.config5_pir_chunk_loop(torch, heamd, d0=256, d1=64, chunks=8, reps=5))
