"""ct x pt inner product (BASELINE configs[4] dim-0 step) over database shapes, including the per-GPU shard of the
reference shape (d0 = 1024 rows x d1 / 8 = 128 columns, 34 GB)."""
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
import path_bench  # noqa: E402

shapes = [(256, 64), (1024, 32), (1024, 128)] if len(sys.argv) < 2 else [tuple(map(int, a.split("x"))) for a in sys.argv[1:]]
for count, columns in shapes:
    print(json.dumps(path_bench.config5_inner_product(torch, heamd, count=count, columns=columns, reps=3)), flush=True)
    torch.cuda.empty_cache()
