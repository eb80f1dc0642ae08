"""N = 16384 with the two 55-bit primes that qualify for the shift-folded products (2^55 - d, d < 2^22): forward / inverse ms per
launch of 2048 polynomials of L = 2 -- the production library against HEAMD_LIBRARY variants (bench_tools/ab_variants.py)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "swift-homomorphic-encryption_amd")]
import torch  # noqa: E402

import heamd  # noqa: E402

degree, count, batch = 16384, 2, 2048
moduli = heamd.generate_primes([55] * count, False, degree)
ctx = heamd.PolyContext(degree, moduli)
bound = torch.tensor(moduli, dtype=torch.int64, device="cuda").view(1, -1, 1)
x = torch.randint(0, 1 << 62, (batch, count, degree), dtype=torch.int64, device="cuda") % bound
out = []
for inverse in (False, True):
    f = ctx.inverse_ntt_ if inverse else ctx.forward_ntt_
    for _ in range(20):
        f(x)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(50):
        f(x)
    b.record()
    b.synchronize()
    out.append("%s %.4f" % ("inv" if inverse else "fwd", a.elapsed_time(b) / 50))
print(os.path.basename(os.environ.get("HEAMD_LIBRARY", "production")), "N=16384 L=2:", "  ".join(out))
