"""Shader / memory clocks (rocm-smi) while one NTT kernel variant runs back to back for a few seconds.
Usage: python bench_tools/clock_probe.py <variant> [seconds]   (variant 0 production, 39 compute only, 1040 memory only)"""
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "swift-homomorphic-encryption_amd"))
import torch  # noqa: E402

import heamd  # noqa: E402

variant = int(sys.argv[1])
seconds = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
degree, batch = 8192, 4096
moduli = heamd.generate_primes([55] * 4, False, degree)
ctx = heamd.PolyContext(degree, moduli)
bound = torch.tensor(moduli, dtype=torch.int64, device="cuda").view(1, -1, 1)
x = torch.randint(0, 1 << 62, (batch, 4, degree), dtype=torch.int64, device="cuda") % bound
samples = []
stop = False


def poll():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        keep = [line.strip() for line in out.splitlines() if any(k in line for k in ("sclk", "mclk", "fclk", "Power"))]
        samples.append(" | ".join(keep))
        time.sleep(0.3)


thread = threading.Thread(target=poll)
thread.start()
t0 = time.time()
launches = 0
while time.time() - t0 < seconds:
    for _ in range(200):
        ctx.ntt_variant_(x, False, variant)
    torch.cuda.synchronize()
    launches += 200
elapsed = time.time() - t0
stop = True
thread.join()
print(f"variant {variant}: {elapsed / launches * 1e3:.4f} ms per launch over {launches} launches")
for s in samples[2:8]:
    print("  ", s[:300])
