"""PolyRq<UInt32> transform timings (device-resident, HIP events): python bench_tools/word32_bench.py"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "swift-homomorphic-encryption_amd"))
import torch  # noqa: E402

import heamd  # noqa: E402


heamd.set_scratch_cache()  # a server's setting: the library keeps its freed scratch (he_set_scratch_cache)


def run(degree, bits, batch, reps=20):
    moduli = heamd.generate_primes(bits, False, degree)
    ctx = heamd.PolyContext(degree, moduli)
    bound = torch.tensor(moduli, dtype=torch.int64, device="cuda").view(1, len(moduli), 1)
    x = (torch.randint(0, 1 << 40, (batch, len(moduli), degree), dtype=torch.int64, device="cuda") % bound).to(torch.int32)
    bytes_per = 2 * len(moduli) * degree * 4 * batch
    for inverse in (False, True):
        fn = ctx.inverse_ntt_u32_ if inverse else ctx.forward_ntt_u32_
        for _ in range(5):
            fn(x)
        start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        start.record()
        for _ in range(reps):
            fn(x)
        stop.record()
        stop.synchronize()
        sec = start.elapsed_time(stop) * 1e-3 / reps
        print(f"u32 N={degree} L={len(moduli)} batch={batch} {'inv' if inverse else 'fwd'}: {sec*1e3:8.3f} ms  "
              f"{batch/sec/1e6:7.3f} M poly/s  {bytes_per/sec/1e9:8.1f} GB/s", flush=True)


if __name__ == "__main__":
    run(4096, [27, 28, 28], 16384)
    run(8192, [30, 30, 30, 30], 4096)
