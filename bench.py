#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native BFV PolyRq/NTT engine.

Workload at N GPUs (BASELINE.json configs[1], weak scaling: the same batch on every GPU):
    batched forward + inverse negacyclic NTT, N = 8192, L = 4 RNS moduli (55-bit), 4096 polynomials per GPU
    (1 GiB device-resident slab per GPU), synthetic uniform residues.
One "step" = one forward NTT of the whole batch followed by one inverse NTT of the whole batch, i.e.
2 x 4096 polynomial transforms per GPU.  `value` = polynomial transforms per second over all GPUs, inputs resident in
HBM when the timed region starts.

The same JSON line carries
  * roofline     -- achieved algorithmic HBM bytes/s of the dominant kernel (forward NTT), timed live with HIP events
                    on the launch stream, against the 8 TB/s HBM3E peak;
  * cpu_baseline -- the CPU oracle (a C port of the reference's Harvey NTT, oracle/he_oracle.c) timed on this box's
                    host cores on a bounded sample of the same workload;
  * extras       -- forward / inverse / ct x ct rates measured separately (not part of `value`).

Launch: `python bench.py --gpus 1` or, for N > 1,
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N`.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "swift-homomorphic-encryption_amd"))

DEGREE = 8192
MODULI_BITS = [55, 55, 55, 55]
BATCH = 4096
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=BATCH, help="polynomials per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=0, help="polynomials in the CPU sample (0 = auto)")
    ap.add_argument("--skip-gather", action="store_true")
    ap.add_argument("--skip-other-configs", action="store_true", help="do not time configs 3-5 (extras only)")
    return ap.parse_args()


def synthetic_slab(torch, moduli, batch, degree, seed):
    """Uniform residues in [0, q_i): counter-based generator on the device."""
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed)
    bound = torch.tensor(moduli, dtype=torch.int64, device="cuda").view(1, len(moduli), 1)
    x = torch.randint(0, 1 << 62, (batch, len(moduli), degree), dtype=torch.int64, device="cuda", generator=gen)
    return x % bound


def pmc_traffic_gbps(batch, forward_seconds):
    """HBM traffic of one forward launch as counted by the PMC passes committed under profiles/ (bench.py cannot
    run rocprofv3 around itself); None when the committed profile was taken at another batch size."""
    path = os.path.join(ROOT, "profiles", "r01h_pmc_ntt_traffic.json")
    try:
        with open(path) as f:
            profile = json.load(f)
    except OSError:
        return None
    if profile.get("batch") != batch or profile.get("degree") != DEGREE:
        return None
    return profile["hbm_bytes_per_launch"] / forward_seconds / 1e9


def time_kernel(torch, fn, reps):
    """Average duration (s) of fn() over reps launches, HIP events on the current (launch) stream."""
    start = torch.cuda.Event(enable_timing=True)
    stop = torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    start.record()
    for _ in range(reps):
        fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop) * 1e-3 / reps


def clocks_under_load(torch, fn, seconds=1.5):
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
        for _ in range(100):
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
    """Times the CPU oracle (port of the reference's NTT) on a bounded sample: forward + inverse of sample_polys."""
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
    return {
        "value": 2 * sample_polys / elapsed,
        "unit": "poly-NTT/s",
        "cores": threads,
        "kind": "port",
        "sample": f"forward+inverse NTT of {sample_polys} polynomials (N={DEGREE}, L={len(moduli)}), "
                  f"{threads} host threads, one polynomial per thread; C port of the reference's Harvey NTT "
                  f"(oracle/he_oracle.c), not the Swift binary",
        "single_thread_forward_poly_ntt_per_s": single,
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
    if distributed:
        import torch.distributed as dist

        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    if args.gpus != world and rank == 0 and distributed:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)

    moduli = heamd.generate_primes(MODULI_BITS, False, DEGREE)
    ctx = heamd.PolyContext(DEGREE, moduli)
    # weak scaling: the job is batch * world polynomials, rank r owns the contiguous shard [begin, end)
    total_polys = args.batch * world
    begin, end = sharding.shard_bounds(total_polys, world, rank)
    slab = synthetic_slab(torch, moduli, end - begin, DEGREE, seed=0x5EED + rank)

    def step():
        ctx.forward_ntt_(slab)
        ctx.inverse_ntt_(slab)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if distributed:
        dist.barrier()
        elapsed = sharding.max_over_ranks(elapsed, device="cuda")

    # ---- separately timed kernels (rank 0 reports): forward is the dominant kernel for the roofline figure
    forward_s = time_kernel(torch, lambda: ctx.forward_ntt_(slab), max(5, args.steps))
    inverse_s = time_kernel(torch, lambda: ctx.inverse_ntt_(slab), max(5, args.steps))
    bytes_per_transform = 2 * len(moduli) * DEGREE * 8  # read + write each word once
    achieved_gbps = bytes_per_transform * args.batch / forward_s / 1e9

    # the attainable figure next to the nominal peak: a device-to-device copy of the same slab
    scratch = torch.empty_like(slab)
    copy_s = time_kernel(torch, lambda: scratch.copy_(slab), max(5, args.steps))
    copy_gbps = 2 * slab.numel() * 8 / copy_s / 1e9
    del scratch

    load_state = clocks_under_load(torch, lambda: ctx.forward_ntt_(slab)) if rank == 0 else None

    gather_ms = None
    if distributed and not args.skip_gather:
        # the only collective on the path: gather the per-GPU result shards (RCCL all-gather over xGMI)
        out = sharding.gather_shards(slab, total_polys)
        torch.cuda.synchronize()
        del out
        t1 = time.perf_counter()
        out = sharding.gather_shards(slab, total_polys)
        torch.cuda.synchronize()
        gather_ms = (time.perf_counter() - t1) * 1e3
        del out

    if rank == 0:
        total_transforms = 2 * args.batch * args.steps * world
        result = {
            "metric": "polynomial NTT throughput (forward+inverse, N=8192, L=4 RNS moduli)",
            "value": total_transforms / elapsed,
            "unit": "poly-NTT/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[1]: batched forward+inverse negacyclic NTT, N=8192, 4 RNS moduli "
                            "(55-bit), %d polynomials per GPU, device-resident" % args.batch,
                "degree": DEGREE,
                "moduli": moduli,
                "batch_per_gpu": args.batch,
                "parallelism": "batch sharded over %d GPU(s), no data-path collective" % world,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "ntt_forward_tiled<13, 10, 3, 0, 0> (forward NTT: one 1024-lane workgroup per residue row, "
                          "8 words per lane, headroom butterflies)",
                "achieved": achieved_gbps,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved_gbps / HBM_PEAK_GBPS,
                "traffic": pmc_traffic_gbps(args.batch, forward_s),
                "traffic_source": "profiles/r01h_pmc_ntt_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes, "
                                  "gfx950 FETCH_SIZE x2 correction), bytes per launch / this run's launch time",
                "algorithmic_bytes_per_launch": bytes_per_transform * args.batch,
                "avg_launch_ms": forward_s * 1e3,
                "copy_rate": copy_gbps,  # measured read + write rate of a plain copy of the same 1 GiB slab
                "frac_of_copy_rate": achieved_gbps / copy_gbps,
                "under_load": load_state,  # rocm-smi while the kernel runs back to back: it sits at the power cap
            },
            "extras": {
                "forward_poly_ntt_per_s": args.batch / forward_s,
                "inverse_poly_ntt_per_s": args.batch / inverse_s,
                "forward_residue_ntt_per_s": args.batch * len(moduli) / forward_s,
                "inverse_achieved_GBps": bytes_per_transform * args.batch / inverse_s / 1e9,
                "all_gather_ms": gather_ms,
                "library": heamd.version(),
            },
        }
        if world == 1 and not args.skip_other_configs:
            # the other BASELINE.json configs on this GPU (ciphertext-mul/s, mod-switch, PIR inner loop); not `value`
            sys.path.insert(0, os.path.join(ROOT, "bench_tools"))
            import path_bench

            del slab
            torch.cuda.empty_cache()
            result["extras"]["other_configs"] = path_bench.run_all(quick=False)
        if not args.no_cpu_baseline and world == 1:
            result["cpu_baseline"] = cpu_baseline(moduli, args.cpu_sample)
        else:
            result["cpu_baseline"] = None
        print(json.dumps(result))
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
