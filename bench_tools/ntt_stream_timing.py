"""Forward NTT timing of the production tiled kernel (variant 0) against the streaming kernel (variant 11)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "swift-homomorphic-encryption_amd"))
import torch  # noqa: E402

import heamd  # noqa: E402

degree, batch = 8192, 4096
moduli = heamd.generate_primes([55] * 4, False, degree)
ctx = heamd.PolyContext(degree, moduli)
bound = torch.tensor(moduli, dtype=torch.int64, device="cuda").view(1, -1, 1)
x = torch.randint(0, 1 << 62, (batch, 4, degree), dtype=torch.int64, device="cuda") % bound
for rnd in range(3):
    for variant in (0, 3, 11, 12):
        for _ in range(20):
            ctx.ntt_variant_(x, False, variant)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(50):
            ctx.ntt_variant_(x, False, variant)
        b.record(); b.synchronize()
        print(f"round {rnd} variant {variant:2d} forward {a.elapsed_time(b)/50:.4f} ms", flush=True)
