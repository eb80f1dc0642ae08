"""Small fixed workload for rocprofv3 counter passes: N=8192, L=4, 4096 polys, forward+inverse, auto and wide
(NTT_DEGREE / NTT_BATCH in the environment: another ring size, e.g. 16384 / 1024)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "swift-homomorphic-encryption_amd"))
import torch  # noqa: E402

import heamd  # noqa: E402


heamd.set_scratch_cache()  # a server's setting: the library keeps its freed scratch (he_set_scratch_cache)

degree, bits, batch = int(os.environ.get("NTT_DEGREE", "8192")), [55] * 4, int(os.environ.get("NTT_BATCH", "4096"))
moduli = heamd.generate_primes(bits, False, degree)
ctx = heamd.PolyContext(degree, moduli)
bound = torch.tensor(moduli, dtype=torch.int64, device="cuda").view(1, len(moduli), 1)
x = torch.randint(0, 1 << 62, (batch, len(moduli), degree), dtype=torch.int64, device="cuda") % bound
variants = [int(v) for v in sys.argv[1:]] or [0, 3]
for variant in variants:
    for inverse in (False, True):
        for _ in range(3):
            ctx.ntt_variant_(x, inverse, variant)
torch.cuda.synchronize()
