"""Does the dim-0 pass over a resident database (8 chunks of 256 x 64 plaintexts, N = 8192, L = 4: 34 GB) depend on WHERE the
database lies?  The same launch over databases allocated one after another in one process -- freed and re-allocated, behind
spacers of different sizes -- with the buffer's address and the median of 10 passes for each."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "swift-homomorphic-encryption_amd"), os.path.join(ROOT, "bench_tools")]
import torch  # noqa: E402

import heamd  # noqa: E402
from path_bench import _timed, _uniform  # noqa: E402

heamd.set_scratch_cache()
degree, d0, d1, chunks = 8192, 256, 64, 8
q = heamd.generate_primes([55] * 5, False, degree)
ctx = heamd.BfvContext(degree, 557057, q)
moduli = q[:-1]
cts = ctx.ciphertext_context().forward_ntt_(_uniform(torch, moduli, (d0, 2), degree, 5))
columns = chunks * d1
bound = torch.tensor(moduli, dtype=torch.int64, device="cuda").view(1, -1, 1)
spacers = [0, 0, 3 << 20, 1 << 30, (1 << 30) + (5 << 20), 0, 7 << 30, 0]
for attempt, spacer_bytes in enumerate(spacers):
    spacer = torch.empty(spacer_bytes, dtype=torch.uint8, device="cuda") if spacer_bytes else None
    database = torch.empty((columns * d0, len(moduli), degree), dtype=torch.int64, device="cuda")
    for first in range(0, columns * d0, 4096):  # canonical words, generated in pieces (no 34 GB temporaries)
        part = database[first:first + 4096]
        part.random_(0, 1 << 62)
        part.remainder_(bound)
    torch.cuda.synchronize()
    t = _timed(torch, lambda: ctx.inner_product_plain_resident(cts, database, None, 2, columns), 10)
    print("allocation %d  spacer %5d MiB  address 0x%x (mod 2 MiB: %d KiB)  dim-0 median %.3f ms  min %.3f  max %.3f  = %.3f of 8 TB/s" % (
        attempt, spacer_bytes >> 20, database.data_ptr(), (database.data_ptr() % (2 << 20)) >> 10, t.spread["median_ms"],
        t.spread["min_ms"], t.spread["max_ms"], database.numel() * 8 / (t.spread["median_ms"] * 1e-3) / 8e12), flush=True)
    del database, spacer
    torch.cuda.empty_cache()
