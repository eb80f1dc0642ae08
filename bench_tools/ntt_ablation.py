"""Ablation timings of the forward NTT kernel (N=8192, L=4, 4096 polys).  Variants 16+bits are measurement-only
kernels whose results are NOT transforms: bit0 uniform twiddles, bit1 no LDS exchange, bit2 no global load/store,
bit3 no conditional subtract, bit7 conflict-free linear LDS pattern instead of the transposes."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "swift-homomorphic-encryption_amd"))
import torch  # noqa: E402

import heamd  # noqa: E402

degree, bits, batch = 8192, [55] * 4, 4096
moduli = heamd.generate_primes(bits, False, degree)
ctx = heamd.PolyContext(degree, moduli)
bound = torch.tensor(moduli, dtype=torch.int64, device="cuda").view(1, len(moduli), 1)
x = torch.randint(0, 1 << 62, (batch, len(moduli), degree), dtype=torch.int64, device="cuda") % bound
names = {0: "full kernel", 16: "full kernel (again, via ablation entry)", 48: "no global load (store kept)", 80: "no global store (load kept)", 17: "uniform twiddles", 18: "no LDS exchange", 144: "conflict-free linear LDS pattern (wrong transposes)", 1040: "memory only: the kernel's loads and stores, no butterflies", 528: "16 words per lane, full tile: 2 rows per CU", 272: "16 words per lane, tile folded to 32 KB: 3 rows per CU (wrong transposes)", 19: "no LDS + uniform tw", 20: "no global ld/st",
         23: "compute only (no mem, no lds, uniform tw)", 24: "no csub", 25: "no csub + uniform tw",
         31: "compute only, no csub"}
for variant, name in names.items():
    for _ in range(10):
        ctx.ntt_variant_(x, False, variant)
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for _ in range(30):
        ctx.ntt_variant_(x, False, variant)
    stop.record()
    stop.synchronize()
    sec = start.elapsed_time(stop) * 1e-3 / 30
    print(f"{name:45s} {sec*1e3:7.3f} ms  {batch/sec/1e6:6.2f} M poly/s", flush=True)
    x = torch.randint(0, 1 << 62, (batch, len(moduli), degree), dtype=torch.int64, device="cuda") % bound
