"""Does the dim-0 pass over the database keep its rate on fewer CUs, and what do the remaining dimensions cost on the rest?
The PIR chunk loop's two stages (8 chunks of 256 x 64, N = 8192, L = 4) each timed on a stream restricted to a CU mask
(hipExtStreamCreateWithCUMask): if the database pass is bound by HBM and not by its multiplies it loses nothing on 7/8 of
the chip, and the remaining dimensions could run beside it on the other eighth.

    python bench_tools/cu_mask_probe.py"""
import ctypes
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "swift-homomorphic-encryption_amd"), os.path.join(ROOT, "bench_tools")]
import torch  # noqa: E402

import heamd  # noqa: E402
from path_bench import _timed, _uniform  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")
heamd.set_scratch_cache()
degree, d0, d1, chunks = 8192, 256, 64, 8
q = heamd.generate_primes([55] * 5, False, degree)
ctx = heamd.BfvContext(degree, 557057, q)
moduli = q[:-1]
dim0 = _uniform(torch, moduli, (d0, 2), degree, 7)
rest = _uniform(torch, moduli, (d1, 2), degree, 8)
database = _uniform(torch, moduli, (chunks * d1, d0), degree, 9)
key = _uniform(torch, q, (ctx.L, 2), degree, 10)
results = ctx.pir_dim0_columns(dim0, database)
one_chunk = results[:d1].clone()


def masked_stream(enabled):
    """A stream whose kernels run on the CUs whose bit is set in `enabled` (a list of 256 booleans)."""
    words = (ctypes.c_uint32 * 8)()
    for cu, on in enumerate(enabled):
        if on:
            words[cu // 32] |= 1 << (cu % 32)
    handle = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(handle), 8, words)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(handle.value)


def timed_on(stream, fn):
    with torch.cuda.stream(stream):
        t = _timed(torch, fn, 5)
    return t


patterns = {
    "first k": lambda k: [cu < k for cu in range(256)],
    "k/256 of every 32": lambda k: [(cu % 32) < k // 8 for cu in range(256)],
    "k/256 of every 8 (interleaved)": lambda k: [(cu % 8) < k // 32 for cu in range(256)],
}
print("dim-0 pass over 34.4 GB (ms), remaining dimensions of 8 chunks together (ms)")
for name, pattern in patterns.items():
    for k in (256, 224, 192, 128):
        if "every 8" in name and k % 32:
            continue
        stream = masked_stream(pattern(k))
        t_dim0 = timed_on(stream, lambda: ctx.pir_dim0_columns(dim0, database, stream=stream))
        print(f"  {name:32s} {k:3d} CUs: dim-0 {t_dim0 * 1e3:7.3f} ms ({34.359738368 / t_dim0 / 1e3:5.2f} TB/s)")
for name in ("first k", "k/256 of every 32"):
    for k in (256, 64, 32):
        stream = masked_stream(patterns[name](k))

        def tail():
            for c in range(chunks):
                ctx.pir_remaining_dimensions([d0, d1], one_chunk.clone(), rest, key, stream=stream)

        t_tail = timed_on(stream, tail)
        print(f"  {name:32s} {k:3d} CUs: remaining dimensions, chunk by chunk x 8: {t_tail * 1e3:7.3f} ms")
