"""Does the memory-side cache (256 MB) pay for running ct x ct + relinearize in CHUNKS of the batch?  The pipeline's
intermediate slabs are 2.3 MB (lifted) + 1.7 MB (tensor) + 1.3 MB (spread) per product: a chunk of a few dozen products
keeps them cache-resident between the kernel that writes them and the one that reads them, at the price of kernels that
no longer fill the device.  Times the same 1024 products as one batch and as chunks of 32 ... 512 through the same
workspaces (so every chunk reuses the same addresses).

  python bench_tools/chunk_probe.py
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "swift-homomorphic-encryption_amd"), os.path.join(ROOT, "bench_tools")]

import torch  # noqa: E402

import heamd  # noqa: E402
from path_bench import _timed, _uniform  # noqa: E402

heamd.set_scratch_cache()
degree, batch = 8192, 1024
q = heamd.generate_primes([55] * 5, False, degree)
ctx = heamd.BfvContext(degree, 557057, q)
moduli = q[:-1]
lhs, rhs = _uniform(torch, moduli, (batch, 2), degree, 1), _uniform(torch, moduli, (batch, 2), degree, 2)
key = _uniform(torch, q, (ctx.L, 2), degree, 3)
for chunk in (1024, 512, 256, 128, 96, 64, 48, 32):
    ws_mul = torch.empty(ctx.mul_workspace_bytes(chunk) // 8, dtype=torch.int64, device="cuda")
    ws_relin = torch.empty(ctx.relinearize_workspace_bytes(chunk) // 8, dtype=torch.int64, device="cuda")

    def mul_only():
        for start in range(0, batch, chunk):
            ctx.mul(lhs[start:start + chunk], rhs[start:start + chunk], workspace=ws_mul)

    def both():
        for start in range(0, batch, chunk):
            product = ctx.mul(lhs[start:start + chunk], rhs[start:start + chunk], workspace=ws_mul)
            ctx.relinearize(product, key, workspace=ws_relin)

    t_mul, t_both = _timed(torch, mul_only, 5), _timed(torch, both, 5)
    print("chunks of %4d: ct x ct %.1f k/s   ct x ct + relinearize %.1f k/s" % (chunk, batch / t_mul / 1e3, batch / t_both / 1e3))
