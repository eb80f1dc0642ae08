"""A/B timing of kernel experiments kept OUT of the product sources.

An experiment is a module bench_tools/variants/NAME.py with
    DESCRIPTION = "..."
    COMPILE = ["ntt_kernels.hip", ...]            # translation units to rebuild (default: the files the edits touch)
    EDITS = [("file under csrc/", "exact old text", "new text"), ...]
The builder copies csrc/ to a scratch directory, applies the edits (an edit whose old text is not found exactly once is
an error: the experiment has drifted from the source and must be updated), recompiles the affected translation units
and links them with the production objects into lib/variants/libhe_amd_NAME.so.  The product sources carry no hooks.

  python bench_tools/ab_variants.py build NAME [NAME ...]     (no GPU needed; `all` = every module in variants/)
  python bench_tools/ab_variants.py run [--what ntt|degrees|large|c3|small|pir|c4|mid|script:PATH] [--rounds N] [NAME ...]     (on the GPU box)
      times the production library and each variant, one process per library (HEAMD_LIBRARY), interleaved over rounds
      so that clock drift shows up as spread
"""
import concurrent.futures
import glob
import importlib.util
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "swift-homomorphic-encryption_amd")
CSRC = os.path.join(PKG, "csrc")
VARIANTS = os.path.join(PKG, "lib", "variants")
SPECS = os.path.join(ROOT, "bench_tools", "variants")

NTT_TIMER = r'''
import sys
sys.path.insert(0, %r)
import torch, heamd
heamd.set_scratch_cache()
out = []
for degree, count, batch in %%s:
    moduli = heamd.generate_primes([55] * count, False, degree)
    ctx = heamd.PolyContext(degree, moduli)
    bound = torch.tensor(moduli, dtype=torch.int64, device="cuda").view(1, -1, 1)
    x = torch.randint(0, 1 << 62, (batch, count, degree), dtype=torch.int64, device="cuda") %%%% bound
    for inverse in (False, True):
        f = ctx.inverse_ntt_ if inverse else ctx.forward_ntt_
        for _ in range(20):
            f(x)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); a.record()
        for _ in range(50):
            f(x)
        b.record(); b.synchronize()
        out.append("N=%%%%d %%%%s %%%%.4f" %%%% (degree, "inv" if inverse else "fwd", a.elapsed_time(b) / 50))
print("  ".join(out))
''' % PKG
SHAPES = {"ntt": [(8192, 4, 4096)], "degrees": [(4096, 2, 8192), (8192, 4, 4096), (16384, 4, 1024)],
          "large": [(16384, 4, 1024), (32768, 4, 512), (32768, 8, 256)]}
C3_TIMER = ("import sys; sys.path[:0] = [%r, %r, %r]; import torch, heamd, path_bench, json; heamd.set_scratch_cache(); "
            "r = path_bench.config3_ct_mul(torch, heamd, batch=1024, reps=5); "
            "print('ct x ct %%.1f k/s  relinearize %%.1f k/s  both %%.1f k/s' %% (r['ct_mul_per_s'] / 1e3, "
            "r['relinearize_per_s'] / 1e3, r['ct_mul_relinearize_per_s'] / 1e3))" % (
                ROOT, PKG, os.path.join(ROOT, "bench_tools")))

# latency-bound chains of small launches: a single query's expansion, ct x ct and relinearize on 1 / 8 ciphertexts
SMALL_TIMER = r'''
import sys
sys.path[:0] = [%r, %r, %r]
import torch, heamd
from path_bench import _timed, _uniform
heamd.set_scratch_cache()
degree = 8192
q = heamd.generate_primes([55] * 5, False, degree)
bfv = heamd.BfvContext(degree, 557057, q)
moduli = q[:-1]
elements = sorted({(degree >> level) + 1 for level in range(10)})
keys = {e: _uniform(torch, q, (bfv.L, 2), degree, 100 + i) for i, e in enumerate(elements)}
query = _uniform(torch, moduli, (1, 2), degree, 14)
out = []
for outputs in (320, 1024):
    t = _timed(torch, lambda: bfv.pir_expand(query, outputs, keys), 10)
    out.append("expand %%d %%.4f ms" %% (outputs, float(t) * 1e3))
key = _uniform(torch, q, (bfv.L, 2), degree, 3)
for batch in (1, 8, 64):
    lhs = _uniform(torch, moduli, (batch, 2), degree, 1)
    rhs = _uniform(torch, moduli, (batch, 2), degree, 2)
    state = {}
    def mul(): state["ct3"] = bfv.mul(lhs, rhs)
    def relin(): state["ct2"] = bfv.relinearize(state["ct3"], key)
    t_mul = _timed(torch, mul, 20)
    t_relin = _timed(torch, relin, 20)
    out.append("batch %%d ct x ct %%.1f us relinearize %%.1f us" %% (batch, float(t_mul) * 1e6, float(t_relin) * 1e6))
print("  ".join(out))
''' % (ROOT, PKG, os.path.join(ROOT, "bench_tools"))

# the PIR server loop: 8 chunks of 256 x 64 (34 GB), one query -- chunk loop and whole query, median of 20 calls each
PIR_TIMER = r'''
import sys
sys.path[:0] = [%r, %r, %r]
import torch, heamd
from path_bench import _timed, _uniform
heamd.set_scratch_cache()
degree, d0, d1, chunks = 8192, 256, 64, 8
q = heamd.generate_primes([55] * 5, False, degree)
ctx = heamd.BfvContext(degree, 557057, q)
moduli = q[:-1]
total = d0 + d1
query = _uniform(torch, moduli, (1, 2), degree, 7)
elements = sorted({(degree >> level) + 1 for level in range((total - 1).bit_length())})
galois = {e: _uniform(torch, q, (ctx.L, 2), degree, 20 + i) for i, e in enumerate(elements)}
relin = _uniform(torch, q, (ctx.L, 2), degree, 10)
database = _uniform(torch, moduli, (chunks, d0 * d1), degree, 9)
expanded = ctx.pir_expand(query, total, galois)
dim0 = ctx.ciphertext_context().forward_ntt_(expanded[:d0].clone())
loop = _timed(torch, lambda: ctx.pir_compute_response([d0, d1], dim0, expanded[d0:], database, chunks, relinearization_key=relin), 20)
whole = _timed(torch, lambda: ctx.pir_compute_response_to_query([d0, d1], query, 1, galois, relin, database, chunks), 20)
print("chunk loop median %%.3f ms (min %%.3f max %%.3f)  whole query median %%.3f ms (min %%.3f max %%.3f)" %% (
    loop.spread["median_ms"], loop.spread["min_ms"], loop.spread["max_ms"],
    whole.spread["median_ms"], whole.spread["min_ms"], whole.spread["max_ms"]))
''' % (ROOT, PKG, os.path.join(ROOT, "bench_tools"))

C4_TIMER = ("import sys; sys.path[:0] = [%r, %r, %r]; import torch, heamd, path_bench; heamd.set_scratch_cache(); "
            "r = path_bench.config4_mod_switch(torch, heamd, batch=8192, reps=10); "
            "print('N=16384 6->5 moduli: %%.3f M poly/s  frac of 8 TB/s %%.4f  median %%.4f ms' %% (r['poly_per_s'] / 1e6, "
            "r['frac_of_8TBps'], r['spread_ms']['median_ms']))" % (ROOT, PKG, os.path.join(ROOT, "bench_tools")))

# ct x ct and relinearize on batches between the latency-bound and the throughput-bound ends
MID_TIMER = SMALL_TIMER.replace("for outputs in (320, 1024):", "for outputs in ():").replace("for batch in (1, 8, 64):", "for batch in (16, 32, 64, 128, 256, 512):")


def load_spec(name):
    path = os.path.join(SPECS, name + ".py")
    spec = importlib.util.spec_from_file_location("variant_" + name, path)
    module = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(module)
    return module


def apply_edits(name, scratch):
    module = load_spec(name)
    touched = []
    for file, old, new in module.EDITS:
        path = os.path.join(scratch, file)
        text = open(path).read()
        if text.count(old) != 1:
            raise SystemExit(f"variant {name}: expected exactly one occurrence in {file} of:\n{old}\n(found {text.count(old)})")
        open(path, "w").write(text.replace(old, new))
        touched.append(file)
    units = list(dict.fromkeys(getattr(module, "COMPILE", None) or [f for f in touched if f.endswith((".hip", ".cpp"))]))
    if not units:
        raise SystemExit(f"variant {name}: edits touch only headers; name the translation units in COMPILE")
    return units


def build_one(name, product_build):
    top = tempfile.mkdtemp(prefix=f"heamd_{name}_")
    scratch = os.path.join(top, "package", "csrc")  # the sources include "../../include/he_amd.h"
    try:
        os.makedirs(scratch)
        shutil.copytree(os.path.join(ROOT, "include"), os.path.join(top, "include"))
        for f in os.listdir(CSRC):
            if f.endswith((".hip", ".cpp", ".hpp", ".map")):
                shutil.copy(os.path.join(CSRC, f), scratch)
        units = apply_edits(name, scratch)
        stems = {os.path.splitext(u)[0] for u in units}
        objects = [o for o in glob.glob(os.path.join(CSRC, "build", "*.o"))
                   if os.path.splitext(os.path.basename(o))[0] not in stems]
        for unit in units:
            obj = os.path.join(scratch, os.path.splitext(unit)[0] + ".o")
            cmd = [product_build._hipcc(), *product_build.FLAGS, "-c", os.path.join(scratch, unit), "-o", obj]
            if unit.endswith(".cpp"):
                cmd[1:1] = ["-x", "hip"]
            subprocess.run(cmd, check=True)
            objects.append(obj)
        subprocess.run([product_build._hipcc(), "-shared", "-fPIC", f"--offload-arch={product_build.ARCH}",
                        f"-Wl,--version-script={product_build.EXPORTS}", "-o",
                        os.path.join(VARIANTS, f"libhe_amd_{name}.so"), *objects], check=True)
    finally:
        shutil.rmtree(top, ignore_errors=True)
    print("built", name, flush=True)


def build(names):
    sys.path.insert(0, PKG)
    import build as product_build
    product_build.build()
    os.makedirs(VARIANTS, exist_ok=True)
    if names == ["all"]:
        names = sorted(os.path.splitext(f)[0] for f in os.listdir(SPECS) if f.endswith(".py") and not f.startswith("_"))
    with concurrent.futures.ThreadPoolExecutor(max_workers=8) as pool:
        list(pool.map(lambda n: build_one(n, product_build), names))


def run(args):
    what, rounds, names = "ntt", 3, []
    while args:
        a = args.pop(0)
        if a == "--what":
            what = args.pop(0)
        elif a == "--rounds":
            rounds = int(args.pop(0))
        else:
            names.append(a)
    script = what[len("script:"):] if what.startswith("script:") else None  # any bench script: every line it prints
    if script:
        SHAPES[what] = []
    timer = ("import runpy, sys; sys.argv = [%r]; runpy.run_path(%r, run_name='__main__')" % (script, script)) if script else C3_TIMER if what == "c3" else SMALL_TIMER if what == "small" else PIR_TIMER if what == "pir" else MID_TIMER if what == "mid" else C4_TIMER if what == "c4" else NTT_TIMER % repr(SHAPES[what])
    libs = {"production": None}
    for path in sorted(glob.glob(os.path.join(VARIANTS, "libhe_amd_*.so"))):
        name = os.path.basename(path)[len("libhe_amd_"):-3]
        if not names or name in names:
            libs[name] = path
    for round_index in range(rounds):
        for name, path in libs.items():
            env = dict(os.environ)
            if path:
                env["HEAMD_LIBRARY"] = path
            result = subprocess.run([sys.executable, "-c", timer], env=env, capture_output=True, text=True)
            if script and result.returncode == 0:
                for line in result.stdout.strip().splitlines():
                    print(f"round {round_index}  {name:24s} {line}", flush=True)
                continue
            line = result.stdout.strip().splitlines()[-1] if result.returncode == 0 and result.stdout.strip() else (
                "FAILED " + result.stderr[-300:])
            print(f"round {round_index}  {name:24s} {line}", flush=True)


if __name__ == "__main__":
    if len(sys.argv) >= 2 and sys.argv[1] == "build":
        build(sys.argv[2:])
    else:
        run(sys.argv[2:] if len(sys.argv) >= 2 and sys.argv[1] == "run" else sys.argv[1:])
