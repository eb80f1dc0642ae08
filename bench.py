#!/usr/bin/env python3
"""bench.py -- benchmark of the MI355X-native BFV PolyRq/NTT engine.

Default workload (`--workload c2`, BASELINE.json configs[1], the headline): batched forward + inverse negacyclic NTT,
N = 8192, L = 4 RNS moduli (55-bit), 4096 polynomials per GPU (1 GiB device-resident slab per GPU), synthetic uniform
residues.  One "step" = one forward NTT of the whole batch followed by one inverse NTT of the whole batch, i.e.
2 x 4096 polynomial transforms per GPU.  `value` = polynomial transforms per second over all GPUs, inputs resident in
HBM when the timed region starts.

The other BASELINE.json configs run the same contract (weak scaling: the same per-GPU work on every rank, units sharded
by rank with heamd.sharding, no data-path collective; the RCCL all-gather of the result shards is timed separately):
    --workload c3   Bfv ct x ct + relinearize, N=8192, 4 moduli, 1024 ciphertext pairs per GPU     (configs[2])
    --workload c4   divideAndRoundQLast, N=16384, 6 -> 5 moduli, 8192 polynomials per GPU           (configs[3])
    --workload c5   PIR dim-0: 1024 query ciphertexts x 128 database columns per GPU (2^17 ct x pt products, 34 GB of
                    plaintexts per GPU); 8 GPUs = the 2^20 products of configs[4], the database sharded by column as
                    PirUtil.computeResponseForOneChunk groups columns (IndexPir/PirUtil.swift:427-445)

The contract's measurement (W warm-up steps, K timed steps) runs twice in the process.  The first pass, straight after
set-up, is reported as `value_without_pre_roll` / `ms_per_step_without_pre_roll`: the device idles while the host builds
contexts and inputs, and at ~1 ms per step W = 5 warm-up steps end before its clocks have settled under the power cap
(20 steps then read 5-8 % slower than after 100 warm-up steps).  Then the job runs untimed for `pre_roll_s` (0.25 s) and
the same W + K steps are measured again: `value` / `ms_per_step`, the steady state a server sees and the roofline leg
measures.

The JSON line carries
  * roofline     -- achieved algorithmic HBM bytes/s of the workload's dominant kernel, timed live with HIP events on
                    the launch stream, against the 8 TB/s HBM3E peak; `traffic` = the HBM bytes rocprofv3's counters
                    saw for the same launch (committed under profiles/), at this run's launch time;
  * cpu_baseline -- (c2, one GPU) the CPU oracle (a C port of the reference's Harvey NTT, oracle/he_oracle.c) timed
                    on this box's host cores on a bounded sample of the same workload, plus BASELINE configs[0]
                    (forwardNtt N=4096, 2 moduli, one thread);
  * extras       -- separately measured rates (not part of `value`).

Launch: `python bench.py --gpus 1` or, for N > 1,
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N`.
"""
import argparse
import json
import os
import re
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "swift-homomorphic-encryption_amd"))

DEGREE = 8192
MODULI_BITS = [55, 55, 55, 55]
BATCH = 4096
PRE_ROLL_S = 0.25  # untimed set-up run of the job before the W warm-up steps (run_benchmark)
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak
HBM_COPY_GBPS = 6290.0  # MI355X_MICROARCH.md: measured float4 streaming copy (79 % of peak)
TRAFFIC_PROFILE = os.path.join(ROOT, "profiles", "r06_pmc_traffic.json")
VALU_PROFILE = os.path.join(ROOT, "profiles", "r06_c3_valu.json")  # bench_tools/valu_json.py
# lane instructions per second of one opcode alone at 8 waves per SIMD, measured by bench_tools/microbench on an MI355X
# (profiles/r06i_microbench.txt) -- what valu_roofline() falls back to when the tool cannot be run beside the bench
RECORDED_RATES_T = {"v_mad_u64_u32": 35.743, "v_add_u32": 61.507}
ROOFLINE_WARMUPS, ROOFLINE_LAUNCHES = 10, 30  # >= 10 warm-ups, median of >= 30 launches


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", choices=["c2", "c3", "c4", "c5"], default="c2")
    ap.add_argument("--batch", type=int, default=0,
                    help="units per GPU (c2: polynomials, c3: ciphertext pairs, c4: polynomials, c5: database columns); "
                         "0 = the BASELINE.json size")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="polynomials in the CPU sample (0 = auto)")
    ap.add_argument("--skip-gather", action="store_true")
    ap.add_argument("--device-group", type=int, default=0,
                    help="c5 in ONE process over a he_device_group of this many members (the visible GPUs in turn; all on GPU 0 "
                         "when there is one): the C ABI's own multi-GPU split instead of one process per GPU")
    ap.add_argument("--skip-other-configs", action="store_true", help="do not time configs 3-5 (extras only)")
    return ap.parse_args()


def synthetic_slab(torch, moduli, prefix, degree, seed):
    """Uniform residues in [0, q_i), shape prefix + [L][N]: counter-based generator on the device."""
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    bound = torch.tensor(moduli, dtype=torch.int64, device="cuda").view(*([1] * len(prefix)), len(moduli), 1)
    x = torch.randint(0, 1 << 62, tuple(prefix) + (len(moduli), degree), dtype=torch.int64, device="cuda", generator=gen)
    return x % bound


def _mangled_kernel(name):
    """Itanium-mangled fragment of a kernel name as pmc_traffic.py prints it ('ntt_forward_tiled<13, 10, 4, 0, 2>',
    'floor_kernel<4, unsigned long, true>') -- what the library's symbol table holds for the instantiation."""
    base, _, args = name.partition("<")
    if not args:
        return base.encode()
    out = []
    for a in args.rstrip(">").split(","):
        a = a.strip()
        if a in ("true", "false"):
            out.append("Lb%dE" % (a == "true"))
        elif a == "unsigned long":
            out.append("m")
        elif a == "unsigned int":
            out.append("j")
        elif re.fullmatch(r"-?\d+", a):
            out.append("Li%sE" % a.replace("-", "n"))
        else:
            return base.encode()  # a type this table does not know: fall back to the bare name
    return ("%sI%sE" % (base, "".join(out))).encode()


_LIBRARY_IMAGE = None


def kernels_in_library(names):
    """True when every kernel the entry was counted on is an instantiation the CURRENT libhe_amd.so still carries (its
    symbol table lists the kernel stubs): an entry collected on kernels that no longer exist is stale."""
    global _LIBRARY_IMAGE
    if _LIBRARY_IMAGE is None:
        import heamd

        with open(heamd.library_path(), "rb") as f:
            _LIBRARY_IMAGE = f.read()
    return [n for n in names if _mangled_kernel(n) not in _LIBRARY_IMAGE]


def profiled_traffic(key, algorithmic_bytes_per_unit=None):
    """HBM bytes per unit (and per launch of the dominant kernel) counted by the rocprofv3 --pmc passes committed under
    profiles/ (bench.py cannot run rocprofv3 around itself); None when there is no entry -- or when the entry claims
    FEWER bytes than the algorithm must move (a stale or miscalibrated profile: counters cannot undercut the compulsory
    traffic of a kernel whose working set exceeds every cache), or names a kernel the current library does not contain."""
    try:
        with open(TRAFFIC_PROFILE) as f:
            entry = json.load(f).get(key)
    except OSError:
        return None
    if entry:
        names = list(entry.get("kernel") or []) + list((entry.get("per_kernel_bytes_per_unit") or {}).keys())
        missing = kernels_in_library(names)
        if missing:
            print(f"bench.py: ignoring {TRAFFIC_PROFILE}[{key}]: counted on kernels the library no longer has: {missing}",
                  file=sys.stderr)
            return None
    if entry and algorithmic_bytes_per_unit is not None:
        per_unit = entry.get("hbm_bytes_per_unit")
        if per_unit is None and entry.get("units_per_launch"):
            per_unit = entry["hbm_bytes_per_launch"] / entry["units_per_launch"]
        if per_unit is None or per_unit < algorithmic_bytes_per_unit:
            print(f"bench.py: ignoring {TRAFFIC_PROFILE}[{key}]: {per_unit} B/unit is below the algorithmic "
                  f"{algorithmic_bytes_per_unit} B/unit (FETCH_SIZE / WRITE_SIZE are calibrated for 16-byte streaming "
                  f"accesses only; MI355X_MICROARCH.md)", file=sys.stderr)
            return None
    return entry


_INSTRUCTION_RATES = None


def instruction_rates():
    """({opcode: T lane instructions / s at 8 waves per SIMD}, live?) from bench_tools/microbench run on this GPU now (a second
    of dependent-free chains per opcode); the recorded rates when the binary is missing or fails."""
    global _INSTRUCTION_RATES
    if _INSTRUCTION_RATES is None:
        import re
        import subprocess

        rates, live = dict(RECORDED_RATES_T), False
        tool = os.path.join(ROOT, "bench_tools", "microbench")
        try:
            if os.environ.get("HEAMD_RECORDED_RATES"):  # (profile targets: no second program under the profiler)
                raise RuntimeError("HEAMD_RECORDED_RATES is set")
            text = subprocess.run([tool], capture_output=True, text=True, timeout=120, check=True).stdout
            found = {m.group(1): float(m.group(2))
                     for m in re.finditer(r"^(\S+)\s+waves/SIMD=8\s.*Tlane_instr/s=([0-9.]+)", text, re.M)}
            if all(k in found for k in rates):
                rates, live = {k: found[k] for k in rates}, True
        except Exception as err:  # noqa: BLE001
            print(f"bench.py: {tool} did not run ({err}); recorded instruction rates used", file=sys.stderr)
        _INSTRUCTION_RATES = (rates, live)
    return _INSTRUCTION_RATES


def valu_roofline(products_per_s):
    """The vector-ALU roofline of ct x ct + relinearize: what binds the pipeline is 64-bit integer multiply issue, not HBM
    (DESIGN.md 4.3).  Numerator: the VALU wave instructions per product the rocprofv3 --pmc pass counted on the pipeline's
    seven kernels (SQ_INSTS_VALU, SQ_INSTS_VALU_INT64, SQ_INSTS_VALU_INT32 -- profiles/r06_c3_valu.json, replayed like the HBM
    counter bytes: `counts_live` false).  Denominator: this GPU's measured issue rates -- 64-bit integer instructions at the
    v_mad_u64_u32 rate, every other VALU instruction at the v_add_u32 rate.  `frac` is the 64-bit-integer instruction stream
    alone against its rate; `issue_frac` the whole VALU stream against the time the part needs to issue it."""
    try:
        with open(VALU_PROFILE) as f:
            counts = json.load(f)
    except OSError:
        return None
    missing = kernels_in_library(counts.get("kernel") or [])
    if missing:
        print(f"bench.py: ignoring {VALU_PROFILE}: counted on kernels the library no longer has: {missing}", file=sys.stderr)
        return None
    rates, live = instruction_rates()
    int64 = counts["int64_wave_instructions_per_product"] * 64
    other = (counts["valu_wave_instructions_per_product"] - counts["int64_wave_instructions_per_product"]) * 64
    achieved = int64 * products_per_s / 1e12
    issue_s_per_product = int64 / (rates["v_mad_u64_u32"] * 1e12) + other / (rates["v_add_u32"] * 1e12)
    return {
        "bound": "valu",
        "unit": "T lane-instr/s",
        "achieved": achieved,                       # 64-bit integer VALU instructions (multiply-adds, 64-bit adds / shifts)
        "peak": rates["v_mad_u64_u32"],
        "frac": achieved / rates["v_mad_u64_u32"],
        "issue_frac": issue_s_per_product * products_per_s,
        "products_per_s_at_issue_rate": 1.0 / issue_s_per_product,
        "int64_lane_instructions_per_product": int64,
        "valu_lane_instructions_per_product": int64 + other,
        "other_rate": rates["v_add_u32"],
        "rates_live": live,
        "counts_live": False,
        "counts_source": counts.get("source"),
    }


def time_kernel(torch, fn, reps, warmups=0):
    """Average duration (s) of fn() over reps back-to-back launches, HIP events on the current (launch) stream."""
    for _ in range(warmups):
        fn()
    start = torch.cuda.Event(enable_timing=True)
    stop = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for _ in range(reps):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) * 1e-3 / reps


def time_launches(torch, fn, warmups=ROOFLINE_WARMUPS, launches=ROOFLINE_LAUNCHES):
    """SURVEY.md 8(d)'s protocol for the roofline leg: `warmups` untimed launches, then `launches` launches in one
    back-to-back stream with a HIP event between consecutive ones (on the launch stream).  Returns (average, median, min)
    in seconds: the average is the whole region / launches (what rocprofv3's kernel trace averages), the median and the
    minimum are over the per-launch intervals."""
    for _ in range(warmups):
        fn()
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(launches + 1)]
    torch.cuda.synchronize()
    marks[0].record()
    for k in range(launches):
        fn()
        marks[k + 1].record()
    marks[-1].synchronize()
    each = [marks[k].elapsed_time(marks[k + 1]) * 1e-3 for k in range(launches)]
    return marks[0].elapsed_time(marks[-1]) * 1e-3 / launches, statistics.median(each), min(each)


def clocks_under_load(torch, fn, seconds=6.0, launches_per_sync=100):
    """Shader clock (MHz) and socket power (W) reported by rocm-smi while fn() runs back to back; None without rocm-smi.
    The NTT kernel runs at the socket power cap, so the clock it gets is part of the measurement."""
    import re
    import shutil
    import subprocess
    import threading

    if shutil.which("rocm-smi") is None:
        return None
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True,
                                     timeout=10).stdout
            except Exception:
                return
            sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
            power = re.search(r"Power \(W\): ([0-9.]+)", out)
            if sclk and power:
                samples.append((int(sclk.group(1)), float(power.group(1))))
            stop.wait(0.2)

    thread = threading.Thread(target=poll, daemon=True)
    thread.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(launches_per_sync):
            fn()
        torch.cuda.synchronize()
    stop.set()
    thread.join(timeout=15)
    steady = samples[1:] if len(samples) > 2 else samples
    if not steady:
        return None
    return {"sclk_mhz": sum(s[0] for s in steady) / len(steady), "socket_power_w": sum(s[1] for s in steady) / len(steady),
            "samples": len(steady)}


def cpu_baseline(moduli, sample_polys):
    """Times the CPU oracle (port of the reference's NTT) on a bounded sample: forward + inverse of sample_polys at the
    headline shape, and BASELINE configs[0] (PolyBenchmark forwardNtt, N=4096, 2 moduli, one thread)."""
    import numpy as np

    import oracle

    oracle.build()
    ctx = oracle.PolyContext(DEGREE, moduli)
    threads = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    rng = np.random.default_rng(0x5EED)
    if sample_polys <= 0:
        # calibrate on 16 polynomials single-threaded, then size the sample for ~10 s of wall time
        probe = np.stack([rng.integers(0, q, size=(16, DEGREE), dtype=np.uint64) for q in moduli], axis=1).copy()
        t0 = time.perf_counter()
        ctx.forward_ntt_inplace(probe, threads=1)
        per_poly = (time.perf_counter() - t0) / 16
        sample_polys = int(max(64, min(8192, 10.0 / (2 * per_poly) * max(1, threads) * 0.7)))
    slab = np.stack([rng.integers(0, q, size=(sample_polys, DEGREE), dtype=np.uint64) for q in moduli], axis=1).copy()
    original = slab.copy()
    t0 = time.perf_counter()
    ctx.forward_ntt_inplace(slab, threads=threads)
    ctx.inverse_ntt_inplace(slab, threads=threads)
    elapsed = time.perf_counter() - t0
    assert np.array_equal(slab, original), "oracle round trip failed"
    t0 = time.perf_counter()
    one = slab[: max(16, sample_polys // (4 * threads))].copy()
    ctx.forward_ntt_inplace(one, threads=1)
    single = one.shape[0] / (time.perf_counter() - t0)
    # BASELINE configs[0]: Benchmarks/PolyBenchmark/PolyBenchmark.swift:148-158 (forwardNtt of one polynomial, degree
    # 4096, two moduli, single thread)
    small_degree = 4096
    small_moduli = oracle.generate_primes([55, 55], False, small_degree)
    small_ctx = oracle.PolyContext(small_degree, small_moduli)
    small = np.stack([rng.integers(0, q, size=(512, small_degree), dtype=np.uint64) for q in small_moduli], axis=1).copy()
    t0 = time.perf_counter()
    small_ctx.forward_ntt_inplace(small, threads=1)
    config0 = small.shape[0] / (time.perf_counter() - t0)
    ct_mul = cpu_ct_mul_baseline(oracle, np, threads)
    return {
        "value": 2 * sample_polys / elapsed,
        "unit": "poly-NTT/s",
        "cores": threads,
        "kind": "port",
        "sample": f"forward+inverse NTT of {sample_polys} polynomials (N={DEGREE}, L={len(moduli)}), "
                  f"{threads} host threads, one polynomial per thread; C port of the reference's Harvey NTT "
                  f"(oracle/he_oracle.c), not the Swift binary",
        "single_thread_forward_poly_ntt_per_s": single,
        "config0_forward_ntt_n4096_l2_single_thread_per_s": config0,
        "config0_sample": "BASELINE configs[0]: forwardNtt N=4096, 2 x 55-bit moduli, 512 polynomials on one thread (port)",
        **ct_mul,
    }


def cpu_ct_mul_baseline(oracle, np, threads):
    """BASELINE.json's second metric on the host: Bfv ct x ct + relinearize at N=8192, L=4 (the reference's
    Benchmarks/RlweBenchmark/RlweBenchmark.swift:387-399,414-428 time `ciphertext * ciphertext` and `relinearize` one
    ciphertext at a time on one thread), with the CPU oracle -- one thread, and one product per thread on all host
    threads -- on a sample sized for a few seconds each."""
    q = oracle.generate_primes([55] * 5, False, DEGREE)
    ctx = oracle.BfvContext(DEGREE, 557057, q)
    moduli = q[:-1]
    rng = np.random.default_rng(0xC3)

    def uniform(prefix, row_moduli):
        return np.ascontiguousarray(np.stack(
            [rng.integers(0, m, size=tuple(prefix) + (DEGREE,), dtype=np.uint64) for m in row_moduli], axis=len(prefix)))

    key = uniform((ctx.L, 2), q)
    lhs, rhs = uniform((2, 2), moduli), uniform((2, 2), moduli)
    t0 = time.perf_counter()
    product = ctx.mul(lhs, rhs, threads=1)
    ctx.relinearize(product, key, threads=1)
    per_product = (time.perf_counter() - t0) / 2
    single = 1.0 / per_product
    count = int(max(threads, min(1024, 6.0 / per_product * threads * 0.7)))
    lhs, rhs = uniform((count, 2), moduli), uniform((count, 2), moduli)
    t0 = time.perf_counter()
    product = ctx.mul(lhs, rhs, threads=threads)
    t_mul = time.perf_counter() - t0
    t0 = time.perf_counter()
    ctx.relinearize(product, key, threads=threads)
    t_relin = time.perf_counter() - t0
    return {
        "ct_mul_relinearize_per_s": count / (t_mul + t_relin),
        "ct_mul_per_s": count / t_mul,
        "relinearize_per_s": count / t_relin,
        "ct_mul_relinearize_single_thread_per_s": single,
        "ct_mul_sample": f"Bfv ct x ct + relinearize of {count} ciphertext pairs (N={DEGREE}, L=4, t=557057), {threads} host "
                         f"threads, one product per thread, and 2 pairs on one thread; C port (oracle/he_oracle.c "
                         f"orc_bfv_mul_mt / orc_bfv_relinearize_mt), not the Swift binary",
    }


# ---------------------------------------------------------------------------------------------------------------------
# Workloads.  Each builds this rank's shard of device-resident synthetic inputs and exposes:
#   step()            one pass of the hot path over the shard (enqueue only)
#   units             units this rank processes per step
#   result()          the tensor holding this rank's results, dim 0 = units (what the final gather moves)
#   describe(world)   metric / unit / dtype / config for the JSON line
#   roofline(steps)   the dominant kernel timed on its own
class NttWorkload:
    """c2: BASELINE configs[1]."""
    key = "c2"

    def __init__(self, torch, heamd, sharding, args, rank, world):
        self.torch, self.heamd = torch, heamd
        self.batch = args.batch or BATCH
        self.moduli = heamd.generate_primes(MODULI_BITS, False, DEGREE)
        self.ctx = heamd.PolyContext(DEGREE, self.moduli)
        # weak scaling: the job is batch * world polynomials, rank r owns the contiguous shard [begin, end)
        self.total = self.batch * world
        begin, end = sharding.shard_bounds(self.total, world, rank)
        self.slab = synthetic_slab(torch, self.moduli, (end - begin,), DEGREE, seed=0x5EED + rank)
        self.units = 2 * (end - begin)
        self.bytes_per_transform = 2 * len(self.moduli) * DEGREE * 8  # read + write each word once

    def step(self):
        self.ctx.forward_ntt_(self.slab)
        self.ctx.inverse_ntt_(self.slab)

    def result(self):
        return self.slab, self.total

    def describe(self, world):
        return {
            "metric": "polynomial NTT throughput (forward+inverse, N=8192, L=4 RNS moduli)",
            "unit": "poly-NTT/s",
            "dtype": "u64",
            "config": {
                "workload": "BASELINE configs[1]: batched forward+inverse negacyclic NTT, N=8192, 4 RNS moduli "
                            "(55-bit), %d polynomials per GPU, device-resident" % self.batch,
                "degree": DEGREE,
                "moduli": self.moduli,
                "batch_per_gpu": self.batch,
                "parallelism": "batch sharded over %d GPU(s), no data-path collective" % world,
            },
        }

    def roofline(self, steps, rank):
        torch, ctx, slab = self.torch, self.ctx, self.slab
        launches = max(ROOFLINE_LAUNCHES, steps)
        forward_s, forward_median_s, forward_min_s = time_launches(torch, lambda: ctx.forward_ntt_(slab), launches=launches)
        inverse_s, inverse_median_s, _ = time_launches(torch, lambda: ctx.inverse_ntt_(slab), launches=launches)
        polys = slab.shape[0]
        achieved = self.bytes_per_transform * polys / forward_s / 1e9
        # the attainable figure next to the nominal peak: the library's own streaming copy of the same
        # slab -- 8 bytes per lane, non-temporal, the transforms' access width (he_words_copy_device)
        scratch = torch.empty_like(slab)
        copy_s = float("inf")
        for non_temporal in (False, True):  # the transforms' row policy and the default one: report the faster
            for _ in range(3):
                self.heamd.stream_copy(slab, scratch, non_temporal)
            copy_s = min(copy_s, time_kernel(torch, lambda: self.heamd.stream_copy(slab, scratch, non_temporal), max(5, steps)))
        copy_gbps = 2 * slab.numel() * 8 / copy_s / 1e9
        del scratch
        load_state = clocks_under_load(torch, lambda: ctx.forward_ntt_(slab)) if rank == 0 else None
        profile = profiled_traffic("c2_forward_ntt", self.bytes_per_transform)
        traffic = None
        if profile and profile.get("units_per_launch") == polys:
            traffic = profile["hbm_bytes_per_launch"] / forward_s / 1e9
        roofline = {
            "bound": "hbm",
            "kernel": "ntt_forward_tiled<13, 10, 4, 0, 2> (forward NTT: one 1024-lane workgroup per pair of residue rows "
                      "of one modulus, 8 words per lane per row, butterflies whose products fold by a shift at 2^(b+2) "
                      "for the moduli 2^b - d)",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic,
            "traffic_live": False,  # counter bytes per launch REPLAYED from the committed rocprofv3 --pmc passes (traffic_source) at this run's launch time; bench.py cannot run rocprofv3 around itself
            "traffic_source": (profile or {}).get("source"),
            "algorithmic_bytes_per_launch": self.bytes_per_transform * polys,
            "avg_launch_ms": forward_s * 1e3,
            "median_launch_ms": forward_median_s * 1e3,
            "min_launch_ms": forward_min_s * 1e3,
            "frac_at_median": self.bytes_per_transform * polys / forward_median_s / 1e9 / HBM_PEAK_GBPS,
            "launches": launches,
            "warmup_launches": ROOFLINE_WARMUPS,
            "copy_rate": copy_gbps,  # read + write rate of the library's streaming copy of the same 1 GiB slab
            "frac_of_copy_rate": achieved / copy_gbps,
            "frac_of_6.29TBps": achieved / HBM_COPY_GBPS,  # against the guide's measured streaming-copy rate
            "under_load": load_state,  # rocm-smi while the kernel runs back to back: it sits at the power cap
            # what the cap means for this kernel, measured once with bench_tools/power_probe.py: its butterflies alone
            # (registers only) and its slab's copy alone spend 0.360 J + 0.343 J per launch at 1 225 W / 875 W; the
            # transform spends 0.683 J at the 1 400 W cap -- time follows energy, not the slower of the two halves
            "power_bound": "profiles/r04f_power_probe.txt",
        }
        extras = {
            "forward_poly_ntt_per_s": polys / forward_s,
            "inverse_poly_ntt_per_s": polys / inverse_s,
            "forward_residue_ntt_per_s": polys * len(self.moduli) / forward_s,
            "inverse_avg_launch_ms": inverse_s * 1e3,
            "inverse_median_launch_ms": inverse_median_s * 1e3,
            "inverse_achieved_GBps": self.bytes_per_transform * polys / inverse_s / 1e9,
            "inverse_frac_of_8TBps": self.bytes_per_transform * polys / inverse_s / 1e9 / HBM_PEAK_GBPS,
        }
        if rank == 0:
            extras.update(self.host_pointer_rate())
        return roofline, extras

    def host_pointer_rate(self, polys=256, reps=8):
        """SURVEY.md 8(d)'s end-to-end figure: he_ntt_forward on HOST pointers (the B1 seam as the reference's benchmark
        loop times a call, PolyBenchmark.swift:148-158) -- upload, transform, download, synchronise, per call, IN PLACE on
        a host slab that stays allocated, as the reference's loop transforms one PolyRq over and over.  (Rounds 1-4 timed
        a binding that first copied the slab into a fresh numpy array: 64 MiB of first-touch page faults per call, 12.7 GB/s
        -- a property of the harness, not of the seam; profiles/r05d_host_seam_pipelined_vs_blocking.txt.)"""
        import numpy as np

        rng = np.random.default_rng(0xE2E)
        host = np.ascontiguousarray(np.stack([rng.integers(0, q, size=(polys, DEGREE), dtype=np.uint64) for q in self.moduli],
                                             axis=1))
        self.ctx.forward_ntt_host_(host)
        times = []
        for _ in range(reps):
            t0 = time.perf_counter()
            self.ctx.forward_ntt_host_(host)
            times.append(time.perf_counter() - t0)
        per_call = sorted(times)[len(times) // 2]
        return {
            "pcie_inclusive_forward_poly_ntt_per_s": polys / per_call,
            "pcie_inclusive_GBps": 2 * host.nbytes / per_call / 1e9,
            "pcie_inclusive_sample": f"he_ntt_forward in place on a pageable host slab of {polys} polynomials ({host.nbytes >> 20} MiB): "
                                     f"H2D + forward NTT + D2H + synchronise per call, median of {reps} calls; bytes = both directions",
        }


class CtMulWorkload:
    """c3: BASELINE configs[2] -- Bfv.mulAssign(ct, ct) + relinearize."""
    key = "c3"
    COMPULSORY = 1_572_864  # read 2 cts x 2 polys, write 2 polys

    def __init__(self, torch, heamd, sharding, args, rank, world):
        self.torch = torch
        self.batch = args.batch or 1024
        q = heamd.generate_primes([55] * 5, False, DEGREE)
        self.q = q
        self.ctx = heamd.BfvContext(DEGREE, 557057, q)
        moduli = q[:-1]
        self.total = self.batch * world
        begin, end = sharding.shard_bounds(self.total, world, rank)
        mine = end - begin
        self.lhs = synthetic_slab(torch, moduli, (mine, 2), DEGREE, 100 + rank)
        self.rhs = synthetic_slab(torch, moduli, (mine, 2), DEGREE, 200 + rank)
        self.key_ = synthetic_slab(torch, q, (self.ctx.L, 2), DEGREE, 3)  # the evaluation key is replicated
        self.ws_mul = torch.empty(self.ctx.mul_workspace_bytes(mine) // 8, dtype=torch.int64, device="cuda")
        self.ws_relin = torch.empty(self.ctx.relinearize_workspace_bytes(mine) // 8, dtype=torch.int64, device="cuda")
        self.units = mine
        self.out = None

    def step(self):
        product = self.ctx.mul(self.lhs, self.rhs, workspace=self.ws_mul)
        self.out = self.ctx.relinearize(product, self.key_, workspace=self.ws_relin)

    def result(self):
        return self.out, self.total

    def describe(self, world):
        return {
            "metric": "ciphertext-mul/s (Bfv ct x ct + relinearize, N=8192, L=4)",
            "unit": "ciphertext-mul/s",
            "dtype": "u64",
            "config": {
                "workload": "BASELINE configs[2]: Bfv<UInt64> ct x ct (lift, NTT, tensor, iNTT, floor) + relinearize, "
                            "N=8192, 4 ciphertext moduli + 1 key-switching modulus (55-bit), %d ciphertext pairs per "
                            "GPU, device-resident" % self.batch,
                "degree": DEGREE,
                "moduli": self.q,
                "batch_per_gpu": self.batch,
                "parallelism": "ciphertext pairs sharded over %d GPU(s), key replicated, no data-path collective" % world,
            },
        }

    def roofline(self, steps, rank):
        torch, ctx = self.torch, self.ctx
        state = {}

        def mul():
            state["p"] = ctx.mul(self.lhs, self.rhs, workspace=self.ws_mul)

        def relin():
            ctx.relinearize(state["p"], self.key_, workspace=self.ws_relin)

        reps = max(ROOFLINE_LAUNCHES, steps)
        mul()
        t_mul = time_kernel(torch, mul, reps, warmups=ROOFLINE_WARMUPS)
        t_relin = time_kernel(torch, relin, reps, warmups=ROOFLINE_WARMUPS)
        t_both = t_mul + t_relin
        achieved = self.COMPULSORY * self.units / t_both / 1e9
        profile = profiled_traffic("c3_ct_mul_relinearize", self.COMPULSORY)
        traffic = profile["hbm_bytes_per_unit"] * self.units / t_both / 1e9 if profile else None
        roofline = {
            "bound": "hbm",
            "kernel": "pipeline of 8 launches (lift x2, the row-fused BEHZ kernel x2 bands: four forward NTTs, tensor product and "
                      "three inverse NTTs per (item, [Q,Bsk] row) in one workgroup; floor, spread + forward NTT, key MAC + "
                      "inverse NTT of the q_ks row, key MAC + inverse NTT of the other rows with the key switch's end in its "
                      "store); achieved = compulsory bytes / time",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic,
            "traffic_live": False,  # counter bytes per launch REPLAYED from the committed rocprofv3 --pmc passes (traffic_source) at this run's launch time; bench.py cannot run rocprofv3 around itself
            "traffic_frac": traffic / HBM_PEAK_GBPS if traffic else None,  # what the pipeline really moves, against the peak
            "traffic_bytes_per_unit": (profile or {}).get("hbm_bytes_per_unit"),
            "traffic_source": (profile or {}).get("source"),
            "algorithmic_bytes_per_unit": self.COMPULSORY,
            "avg_launch_ms": t_both * 1e3,
        }
        if rank == 0:  # is the pipeline at the socket's power cap too?  (rocm-smi while it runs back to back)
            roofline["under_load"] = clocks_under_load(torch, lambda: (mul(), relin()), seconds=3.0, launches_per_sync=4)
        # the roofline that binds: vector-ALU issue (the HBM figures above are the contract's line, 0.04 of its peak)
        roofline["valu"] = valu_roofline(self.units / t_both)
        extras = {"ct_mul_per_s": self.units / t_mul, "relinearize_per_s": self.units / t_relin}
        return roofline, extras


class ModSwitchWorkload:
    """c4: BASELINE configs[3] -- divideAndRoundQLast."""
    key = "c4"
    DEGREE = 16384

    def __init__(self, torch, heamd, sharding, args, rank, world):
        self.torch = torch
        self.batch = args.batch or 8192
        self.moduli = heamd.generate_primes([55] * 6, False, self.DEGREE)
        self.ctx = heamd.PolyContext(self.DEGREE, self.moduli)
        self.total = self.batch * world
        begin, end = sharding.shard_bounds(self.total, world, rank)
        self.x = synthetic_slab(torch, self.moduli, (end - begin,), self.DEGREE, 400 + rank)
        self.units = end - begin
        self.bytes_per_poly = (6 + 5) * self.DEGREE * 8
        self.out = None

    def step(self):
        self.out = self.ctx.divide_and_round_q_last(self.x)

    def result(self):
        return self.out, self.total

    def describe(self, world):
        return {
            "metric": "RNS modulus-switch throughput (divideAndRoundQLast, N=16384, 6 -> 5 moduli)",
            "unit": "poly/s",
            "dtype": "u64",
            "config": {
                "workload": "BASELINE configs[3]: divideAndRoundQLast, N=16384, 6 -> 5 moduli (55-bit), %d polynomials "
                            "per GPU, device-resident" % self.batch,
                "degree": self.DEGREE,
                "moduli": self.moduli,
                "batch_per_gpu": self.batch,
                "parallelism": "polynomials sharded over %d GPU(s), no data-path collective" % world,
            },
        }

    def roofline(self, steps, rank):
        t, t_median, _ = time_launches(self.torch, self.step, launches=max(ROOFLINE_LAUNCHES, steps))
        achieved = self.bytes_per_poly * self.units / t / 1e9
        profile = profiled_traffic("c4_mod_switch", self.bytes_per_poly)
        traffic = profile["hbm_bytes_per_unit"] * self.units / t / 1e9 if profile else None
        return {
            "bound": "hbm",
            "kernel": "divide_and_round_q_last_rows_kernel<6> (16 B per lane, one coefficient pair per lane, the six row loads issued first)",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic,
            "traffic_live": False,  # counter bytes per launch REPLAYED from the committed rocprofv3 --pmc passes (traffic_source) at this run's launch time; bench.py cannot run rocprofv3 around itself
            "traffic_source": (profile or {}).get("source"),
            "algorithmic_bytes_per_launch": self.bytes_per_poly * self.units,
            "avg_launch_ms": t * 1e3,
            "median_launch_ms": t_median * 1e3,
            "under_load": clocks_under_load(self.torch, self.step, seconds=3.0, launches_per_sync=20) if rank == 0 else None,
        }, {}


class PirDim0Workload:
    """c5: the per-GPU column shard of BASELINE configs[4] (PirUtil.swift:427-445: per database column, the inner
    product of the expanded dim-0 query with the column's plaintexts)."""
    key = "c5"
    D0 = 1024

    def __init__(self, torch, heamd, sharding, args, rank, world):
        self.torch = torch
        self.columns_per_gpu = args.batch or 128
        q = heamd.generate_primes([55] * 5, False, DEGREE)
        self.q = q
        self.ctx = heamd.BfvContext(DEGREE, 557057, q)
        moduli = q[:-1]
        self.total_columns = self.columns_per_gpu * world
        begin, end = sharding.shard_bounds(self.total_columns, world, rank)
        self.columns = end - begin
        self.query = synthetic_slab(torch, moduli, (self.D0, 2), DEGREE, 5)  # replicated on every rank (same seed)
        # this rank's columns of the database: plaintext k of column c at [c][k] (MulPir.swift:547-555)
        self.database = synthetic_slab(torch, moduli, (self.columns, self.D0), DEGREE, 600 + rank)
        # --device-group K: the same columns split over the K members of a he_device_group in this one process (members on
        # the visible GPUs in turn; the per-GPU share of the database stays what the contract's c5 line streams per GPU)
        self.group = self.shards = None
        self.members = getattr(args, "device_group", 0)
        if self.members:
            if world != 1:
                raise SystemExit("--device-group is the single-process mode: run it without torchrun")
            visible = torch.cuda.device_count()
            devices = [m % visible for m in range(self.members)]
            gpus = len(set(devices))
            if gpus > 1:  # more GPUs: more columns, the same share per GPU
                self.total_columns = self.columns = self.columns_per_gpu * gpus
                self.database = None
            self.group = heamd.DeviceGroup(devices, DEGREE, 557057, q)
            self.shards = []
            for m, device in enumerate(devices):
                begin, end = self.group.bounds(self.columns, m)
                if end == begin:
                    self.shards.append(None)
                elif self.database is not None:
                    self.shards.append(self.database[begin:end])  # one GPU: views of the one slab
                else:
                    with torch.cuda.device(device):
                        self.shards.append(synthetic_slab(torch, moduli, (end - begin, self.D0), DEGREE, 600 + m))
            torch.cuda.set_device(devices[0])
            self.group_devices = devices
        self.units = self.columns * self.D0
        self.db_bytes = self.units * len(moduli) * DEGREE * 8
        self.out = None
        # the second half of configs[4], run on the gathered set (PirUtil.swift:448-485): the second dimension's query
        # ciphertexts (one per column of the whole database) and the relinearization key, replicated like the dim-0 query
        self.remaining = self.key_ = self.response = None
        if world > 1:
            self.remaining = synthetic_slab(torch, moduli, (self.total_columns, 2), DEGREE, 7)
            self.key_ = synthetic_slab(torch, q, (self.ctx.L, 2), DEGREE, 3)

    def step(self):
        # this rank's columns: the ct x pt inner products and their inverse NTT (he_pir_dim0_columns_device)
        if self.group is not None:
            # he_pir_dim0_columns_group: every member its columns on its own stream, gathered on member 0's device
            self.out = self.group.pir_dim0_columns(self.query, self.shards, self.columns)
        else:
            self.out = self.ctx.pir_dim0_columns(self.query, self.database)

    def result(self):
        return self.out, self.total_columns

    def consume(self, gathered):
        """What follows the all-gather in a column-sharded deployment: the remaining dimension over ALL columns'
        intermediate ciphertexts (he_pir_remaining_dimensions_device; it overwrites its input)."""
        self.response = self.ctx.pir_remaining_dimensions([self.D0, self.total_columns], gathered, self.remaining,
                                                          self.key_)

    def describe(self, world):
        return {
            "metric": "PIR dim-0 ciphertext x plaintext multiply-accumulates per second (N=8192, L=4)",
            "unit": "ct-pt-mac/s",
            "dtype": "u64",
            "config": {
                "workload": "BASELINE configs[4]: PIR server dim-0 inner products, %d query ciphertexts x %d database "
                            "columns per GPU (%d ct x pt products, %.1f GB of Eval plaintexts per GPU); 8 GPUs = the "
                            "2^20 products of the config" % (self.D0, self.columns_per_gpu,
                                                              self.D0 * self.columns_per_gpu, self.db_bytes / 1e9),
                "degree": DEGREE,
                "moduli": self.q,
                "rows": self.D0,
                "columns_per_gpu": self.columns_per_gpu,
                "parallelism": ("one process, he_device_group of %d members on devices %s: database sharded by column over the "
                                "members, query replicated, the members' columns gathered on member 0's device inside the "
                                "timed call" % (self.members, self.group_devices)) if self.group is not None else (
                               "database sharded by column over %d GPU(s), query replicated, no data-path collective; "
                               "the all-gather of the column results is timed separately" % world),
            },
        }

    def roofline(self, steps, rank):
        t = time_kernel(self.torch, self.step, max(ROOFLINE_LAUNCHES, steps), warmups=ROOFLINE_WARMUPS)
        achieved = self.db_bytes / t / 1e9
        profile = profiled_traffic("c5_inner_product_plain", self.db_bytes / self.units)
        traffic = profile["hbm_bytes_per_unit"] * self.units / t / 1e9 if profile else None
        return {
            "bound": "hbm",
            "kernel": "inner_product_plain_rows_kernel (each lane owns one word of 4 output columns and streams the "
                      "rows' (ciphertext, plaintext) pairs into carry-counting accumulators)",
            "achieved": achieved,
            "peak": HBM_PEAK_GBPS,
            "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBPS,
            "traffic": traffic,
            "traffic_live": False,  # counter bytes per launch REPLAYED from the committed rocprofv3 --pmc passes (traffic_source) at this run's launch time; bench.py cannot run rocprofv3 around itself
            "traffic_source": (profile or {}).get("source"),
            "algorithmic_bytes_per_launch": self.db_bytes,
            "avg_launch_ms": t * 1e3,
            "database_GBps_per_gpu": achieved,
            "under_load": clocks_under_load(self.torch, self.step, seconds=3.0, launches_per_sync=4) if rank == 0 else None,
        }, {}


WORKLOADS = {w.key: w for w in (NttWorkload, CtMulWorkload, ModSwitchWorkload, PirDim0Workload)}


def run_benchmark(args, make_job, rank, world, device="cuda", dist=None):
    """The timed part of the contract, independent of the workload and of the device: W warm-up steps, a barrier, exactly
    K steps bracketed by synchronisation, max over ranks, the units of all ranks; then the workload's roofline leg and the
    result gather.  `dist` is torch.distributed (initialised) when world > 1.  Returns the JSON line's dict on rank 0,
    None elsewhere.  (tests/test_sharding_gloo.py drives this function with two gloo ranks and a CPU workload.)"""
    import torch

    from heamd import sharding

    distributed = world > 1
    on_gpu = device == "cuda"

    def synchronize():
        if on_gpu:
            torch.cuda.synchronize()

    def barrier():
        synchronize()
        if distributed:
            dist.barrier()
        synchronize()

    job = make_job()
    # Pre-roll: the device sat idle while the host built contexts and inputs and is in its idle power state; at 1 ms per
    # step the contract's W warm-up steps end before the clocks have settled under the power cap (20 steps after 5
    # warm-ups read 5-8 % slower than after 100).  Both readings are reported (see timed_steps below).
    def timed_steps():
        """The contract's measurement: W warm-up steps, a barrier, exactly K steps bracketed by synchronisation, the
        maximum over ranks."""
        for _ in range(args.warmup):
            job.step()
        barrier()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            job.step()
        synchronize()
        seconds = time.perf_counter() - t0
        if distributed:
            dist.barrier()
            seconds = sharding.max_over_ranks(seconds, device=device)
        return seconds

    # the same measurement twice in one process: first as the bare contract reads it (W warm-ups straight after set-up:
    # `value_without_pre_roll`), then again after the pre-roll (`value`)
    synchronize()
    elapsed_cold = timed_steps()
    pre_roll_start = time.perf_counter()
    while on_gpu and time.perf_counter() - pre_roll_start < PRE_ROLL_S:
        job.step()
        synchronize()
    elapsed = timed_steps()
    units_all_ranks = job.units
    if distributed:
        counter = torch.tensor([job.units], dtype=torch.int64, device=device)
        dist.all_reduce(counter)
        units_all_ranks = int(counter.item())

    # ---- separately timed kernels: the workload's dominant kernel for the roofline figure
    roofline, extras = job.roofline(args.steps, rank)

    gather_ms = with_gather_elapsed = gather_bytes = None
    if distributed and not args.skip_gather:
        gather_bytes = job.result()[0].numel() * job.result()[0].element_size()  # what this rank contributes
        # the only collective on the path: gather the per-GPU result shards (RCCL all-gather over xGMI); also the same
        # K steps with the gather after every step, max over ranks
        local, total = job.result()
        out = sharding.gather_shards(local, total)
        synchronize()
        del out
        barrier()
        t1 = time.perf_counter()
        out = sharding.gather_shards(local, total)
        synchronize()
        gather_ms = sharding.max_over_ranks(time.perf_counter() - t1, device=device) * 1e3
        del out
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.steps):
            job.step()
            local, total = job.result()
            out = sharding.gather_shards(local, total)
            if hasattr(job, "consume"):  # c5: dim-0 -> all-gather -> remaining dimensions, the whole configs[4] flow
                job.consume(out)
        synchronize()
        with_gather_elapsed = sharding.max_over_ranks(time.perf_counter() - t1, device=device)
        del out

    if rank != 0:
        return None
    description = job.describe(world)
    return {
        "metric": description["metric"],
        "value": units_all_ranks * args.steps / elapsed,
        "unit": description["unit"],
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "pre_roll_s": PRE_ROLL_S if on_gpu else 0.0,
        "value_without_pre_roll": units_all_ranks * args.steps / elapsed_cold,
        "ms_per_step_without_pre_roll": elapsed_cold / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": description["dtype"],
        "data": "synthetic",
        "config": dict(description["config"],
                       pre_roll="W warm-ups + K timed steps are run twice in this process: straight after set-up "
                                "(value_without_pre_roll: the device is still leaving its idle power state) and again "
                                "after the job has run untimed for pre_roll_s (value: the steady state the roofline leg "
                                "measures); DESIGN.md section 6"),
        "roofline": roofline,
        "extras": dict(extras, all_gather_ms=gather_ms, all_gather_bytes_per_gpu=gather_bytes,
                       value_with_all_gather=(units_all_ranks * args.steps / with_gather_elapsed
                                              if with_gather_elapsed else None)),
    }


def main():
    args = parse_args()
    import torch

    import heamd
    from heamd import sharding

    rank, local_rank, world = sharding.rank_and_world()
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if distributed:
        import torch.distributed as dist

        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and distributed:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    # the library retains no scratch unless the host opts in (he_set_scratch_cache); a server does, and so does this job
    heamd.set_scratch_cache()
    result = run_benchmark(args, lambda: WORKLOADS[args.workload](torch, heamd, sharding, args, rank, world), rank, world,
                           "cuda", dist)
    if rank == 0:
        result["extras"]["library"] = heamd.version()
        if args.workload == "c2" and world == 1 and not args.skip_other_configs:
            # the other BASELINE.json configs on this GPU (ciphertext-mul/s, mod-switch, PIR inner loop); not `value`
            sys.path.insert(0, os.path.join(ROOT, "bench_tools"))
            import path_bench

            torch.cuda.empty_cache()
            result["extras"]["other_configs"] = path_bench.run_all(quick=False)
        if args.workload == "c2" and not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(heamd.generate_primes(MODULI_BITS, False, DEGREE), args.cpu_sample)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
