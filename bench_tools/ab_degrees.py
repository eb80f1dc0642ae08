"""A/B of forward / inverse NTT over library variants at the other ring sizes (N = 4096 L = 2, N = 16384 L = 4), one
process per library, two rounds.   python bench_tools/ab_degrees.py [NAME ...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "swift-homomorphic-encryption_amd")
TIMER = r'''
import sys
sys.path.insert(0, %r)
import torch, heamd
out = []
for degree, count, batch in ((4096, 2, 8192), (16384, 4, 1024), (8192, 4, 4096)):
    moduli = heamd.generate_primes([55] * count, False, degree)
    ctx = heamd.PolyContext(degree, moduli)
    bound = torch.tensor(moduli, dtype=torch.int64, device="cuda").view(1, -1, 1)
    x = torch.randint(0, 1 << 62, (batch, count, degree), dtype=torch.int64, device="cuda") %% bound
    for inverse in (False, True):
        f = ctx.inverse_ntt_ if inverse else ctx.forward_ntt_
        for _ in range(20):
            f(x)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(50):
            f(x)
        b.record(); b.synchronize()
        out.append(a.elapsed_time(b) / 50)
print("N=4096 fwd %%.4f inv %%.4f | N=16384 fwd %%.4f inv %%.4f | N=8192 fwd %%.4f inv %%.4f" %% tuple(out))
''' % PKG


def main():
    libs = {"production": None}
    for name in sys.argv[1:]:
        libs[name] = os.path.join(PKG, "lib", "variants", f"libhe_amd_{name}.so")
    for round_index in range(2):
        for name, path in libs.items():
            env = dict(os.environ)
            if path:
                env["HEAMD_LIBRARY"] = path
            r = subprocess.run([sys.executable, "-c", TIMER], env=env, capture_output=True, text=True)
            print(f"round {round_index} {name:12s} {r.stdout.strip().splitlines()[-1] if r.returncode == 0 else 'FAILED ' + r.stderr[-300:]}", flush=True)


if __name__ == "__main__":
    main()
