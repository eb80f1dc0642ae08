"""A/B timing of compile-time kernel experiments.

  python bench_tools/ab_variants.py build NAME=[FILE.hip:]-DFLAG=1 [NAME2=...]   (no GPU needed)
      recompiles csrc/FILE.hip (default ntt_kernels.hip) with the extra flag, links it with the other objects of the
      current build into lib/variants/libhe_amd_NAME.so
  python bench_tools/ab_variants.py run [NAME ...]                              (on the GPU box)
      times forward / inverse NTT (N=8192, L=4, 4096 polynomials) for the production library and every variant, each in
      its own process (HEAMD_LIBRARY), interleaved over three rounds so that clock drift shows up as spread
"""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "swift-homomorphic-encryption_amd")
VARIANTS = os.path.join(PKG, "lib", "variants")

TIMER = r'''
import os, sys
sys.path.insert(0, %r)
import torch, heamd
degree, batch = 8192, 4096
moduli = heamd.generate_primes([55] * 4, False, degree)
ctx = heamd.PolyContext(degree, moduli)
bound = torch.tensor(moduli, dtype=torch.int64, device="cuda").view(1, -1, 1)
x = torch.randint(0, 1 << 62, (batch, 4, degree), dtype=torch.int64, device="cuda") %% bound
out = []
for variant in (0, 10):  # production schedule (split butterflies for these moduli), then pinned to the [0, 8p) schedule
    for inverse in (False, True):
        for _ in range(20):
            ctx.ntt_variant_(x, inverse, variant)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(50):
            ctx.ntt_variant_(x, inverse, variant)
        b.record(); b.synchronize()
        out.append(a.elapsed_time(b) / 50)
print("production %%.4f %%.4f   approx %%.4f %%.4f" %% tuple(out))
''' % PKG


def build(specs):
    sys.path.insert(0, PKG)
    import build as product_build
    product_build.build()
    os.makedirs(VARIANTS, exist_ok=True)
    def one(spec):
        name, flag = spec.split("=", 1)
        source = "ntt_kernels.hip"
        if ".hip:" in flag:
            source, flag = flag.split(":", 1)
        stem = os.path.splitext(source)[0]
        objects = [o for o in glob.glob(os.path.join(PKG, "csrc", "build", "*.o")) if not o.endswith(stem + ".o")]
        obj = os.path.join(VARIANTS, f"{stem}_{name}.o")
        subprocess.run([product_build._hipcc(), *product_build.FLAGS, *flag.split(), "-c",
                        os.path.join(PKG, "csrc", source), "-o", obj], check=True)
        subprocess.run([product_build._hipcc(), "-shared", "-fPIC", f"--offload-arch={product_build.ARCH}",
                        f"-Wl,--version-script={product_build.EXPORTS}", "-o",
                        os.path.join(VARIANTS, f"libhe_amd_{name}.so"), obj, *objects], check=True)
        os.unlink(obj)
        print("built", name, flush=True)

    import concurrent.futures
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as pool:
        list(pool.map(one, specs))


def run(names):
    libs = {"production": None}
    for path in sorted(glob.glob(os.path.join(VARIANTS, "libhe_amd_*.so"))):
        name = os.path.basename(path)[len("libhe_amd_"):-3]
        if not names or name in names:
            libs[name] = path
    for round_index in range(3):
        for name, path in libs.items():
            env = dict(os.environ)
            if path:
                env["HEAMD_LIBRARY"] = path
            result = subprocess.run([sys.executable, "-c", TIMER], env=env, capture_output=True, text=True)
            line = result.stdout.strip().splitlines()[-1] if result.returncode == 0 and result.stdout.strip() else (
                "FAILED " + result.stderr[-300:])
            print(f"round {round_index}  {name:24s} fwd/inv ms: {line}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(sys.argv[2:])
