"""Times the NTT kernel variants on the GPU (device-resident, HIP events).  Usage: python bench_tools/ntt_variants.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "swift-homomorphic-encryption_amd"))
import torch  # noqa: E402

import heamd  # noqa: E402


heamd.set_scratch_cache()  # a server's setting: the library keeps its freed scratch (he_set_scratch_cache)

NAMES = {0: "auto", 1: "auto-exact", 2: "generic", 3: "16 words/lane", 10: "auto-approx"}


def run(degree, bits, batch, variants=(0, 1, 3), reps=30):
    moduli = heamd.generate_primes(bits, False, degree)
    ctx = heamd.PolyContext(degree, moduli)
    bound = torch.tensor(moduli, dtype=torch.int64, device="cuda").view(1, len(moduli), 1)
    x = torch.randint(0, 1 << 62, (batch, len(moduli), degree), dtype=torch.int64, device="cuda") % bound
    bytes_per = 2 * len(moduli) * degree * 8 * batch
    for variant in variants:
        for inverse in (False, True):
            for _ in range(10):
                ctx.ntt_variant_(x, inverse, variant)
            start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            start.record()
            for _ in range(reps):
                ctx.ntt_variant_(x, inverse, variant)
            stop.record()
            stop.synchronize()
            sec = start.elapsed_time(stop) * 1e-3 / reps
            print(f"N={degree} L={len(moduli)} bits={bits[0]} batch={batch} {NAMES[variant]:14s} "
                  f"{'inv' if inverse else 'fwd'}: {sec*1e3:8.3f} ms  {batch/sec/1e6:7.3f} M poly/s  "
                  f"{bytes_per/sec/1e9:8.1f} GB/s", flush=True)


if __name__ == "__main__":
    run(8192, [55] * 4, 4096, variants=(0, 3, 0, 3, 10, 1))
    run(4096, [55] * 2, 8192, variants=(0, 3, 10, 1))
    run(16384, [55] * 4, 1024, variants=(0, 10, 1))
    run(32768, [55] * 4, 512, variants=(0, 10, 1, 2))
    run(8192, [60] * 3, 4096, variants=(0,))
    run(4096, [60] * 2, 8192, variants=(0,))
