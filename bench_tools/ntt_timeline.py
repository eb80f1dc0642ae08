"""Per-workgroup phase timeline of the tiled forward NTT (variant 32), N=8192, L=4, 4096 polys."""
import ctypes
import os
import sys

import numpy as np

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
rows = batch * len(moduli)
timeline = torch.zeros((rows, 16), dtype=torch.int64, device="cuda")
lib = heamd.load_library()
assert lib.he_debug_set_ntt_timeline(ctypes.c_void_p(timeline.data_ptr())) == 0
for _ in range(2):
    ctx.ntt_variant_(x, False, 32)
torch.cuda.synchronize()
t = timeline.cpu().numpy().astype(np.int64)
lib.he_debug_set_ntt_timeline(None)
names = ["global load", "pass A (5 st)", "lds store+barrier", "lds load", "pass B (5 st)+tw", "lds store+barrier",
         "lds load", "pass C (3 st)+tw+canon", "global store drain"]
d = np.diff(t[:, :10], axis=1).astype(np.float64)
total = (t[:, 9] - t[:, 0]).astype(np.float64)
print(f"rows={rows}  kernel span (cycles) = {t[:, 9].max() - t[:, 0].min()}")
print(f"per-workgroup lifetime: mean {total.mean():.0f} cycles, p10 {np.percentile(total, 10):.0f}, p90 {np.percentile(total, 90):.0f}")
for k, name in enumerate(names):
    print(f"  {name:26s} mean {d[:, k].mean():8.0f}  p10 {np.percentile(d[:, k], 10):8.0f}  p90 {np.percentile(d[:, k], 90):8.0f}  ({100 * d[:, k].mean() / total.mean():4.1f} %)")
# concurrency per CU: group by (xcc, hw_id CU bits)
hw = t[:, 15]
xcc = hw >> 32
hwid = hw & 0xFFFFFFFF
cu = (hwid >> 8) & 0xF
sh = (hwid >> 12) & 0x1
se = (hwid >> 13) & 0x7
key = xcc * 1000 + se * 100 + sh * 20 + cu
uniq = np.unique(key)
print(f"distinct (xcc,se,sh,cu) = {len(uniq)}")
# for the busiest CU print the first 8 workgroups' intervals
k0 = uniq[0]
sel = np.where(key == k0)[0]
order = sel[np.argsort(t[sel, 0])][:10]
base = t[order, 0].min()
for r in order:
    print(f"  row {r:6d} start {t[r,0]-base:8d} " + " ".join(f"{v-base:8d}" for v in t[r, 1:10]))
