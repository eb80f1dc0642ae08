"""ct x ct then relinearize on CHAIN_BATCH ciphertexts (default 1; N = CHAIN_DEGREE, default 8192, with CHAIN_MODULI 55-bit moduli, default
5 = L 4), repeated, for rocprofv3 --kernel-trace:
the last repetitions' kernels are what bench_tools/timeline_digest.py prints."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "swift-homomorphic-encryption_amd"), os.path.join(ROOT, "bench_tools")]
import torch  # noqa: E402

import heamd  # noqa: E402
from path_bench import _uniform  # noqa: E402

heamd.set_scratch_cache()
degree, batch = int(os.environ.get("CHAIN_DEGREE", "8192")), int(os.environ.get("CHAIN_BATCH", "1"))
q = heamd.generate_primes([55] * int(os.environ.get("CHAIN_MODULI", "5")), False, degree)
bfv = heamd.BfvContext(degree, 557057, q)
moduli = q[:-1]
lhs, rhs = _uniform(torch, moduli, (batch, 2), degree, 1), _uniform(torch, moduli, (batch, 2), degree, 2)
key = _uniform(torch, q, (bfv.L, 2), degree, 3)
for _ in range(int(os.environ.get("CHAIN_REPS", "40"))):
    ct3 = bfv.mul(lhs, rhs)
    ct2 = bfv.relinearize(ct3, key)
torch.cuda.synchronize()
print("done", batch)
