"""Is the forward NTT energy-bound?  Socket power and shader clock (rocm-smi) beside three workloads, five seconds each:
the transform's butterflies alone in registers, the streaming copy of its slab alone, and the transform itself
(he_ntt_forward_device, N = 8192, L = 4, 4096 polynomials).  Prints each leg's rate, power and clock, the energy each
spends on ONE launch's worth of work (872 M butterflies; 2 GiB read + written), and the launch time the socket's power
cap allows if the two energies simply add:  (E_butterflies + E_copy) / P_cap.

  python bench_tools/power_probe.py      (on the GPU box; builds bench_tools/power_probe on first use)
"""
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "swift-homomorphic-encryption_amd")]
HERE = os.path.dirname(os.path.abspath(__file__))
BINARY = os.path.join(HERE, "power_probe")
SECONDS = 5.0


def sampler():
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
            sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
            power = re.search(r"Power \(W\): ([0-9.]+)", out)
            if sclk and power:
                samples.append((time.perf_counter(), int(sclk.group(1)), float(power.group(1))))
            stop.wait(0.15)

    thread = threading.Thread(target=poll, daemon=True)
    thread.start()
    return samples, stop, thread


def measured(run):
    """run() -> rate string; returns (rate string, mean MHz, mean W) over the samples after the first second."""
    samples, stop, thread = sampler()
    t0 = time.perf_counter()
    rate = run()
    stop.set()
    thread.join(timeout=10)
    steady = [s for s in samples if s[0] - t0 > 1.0] or samples
    return rate, sum(s[1] for s in steady) / len(steady), sum(s[2] for s in steady) / len(steady), len(steady)


def main():
    if not os.path.exists(BINARY) or os.path.getmtime(BINARY) < os.path.getmtime(BINARY + ".hip"):
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", BINARY, BINARY + ".hip"])
    import torch

    import heamd

    def binary(mode):
        return lambda: subprocess.run([BINARY, mode, str(SECONDS)], capture_output=True, text=True).stdout.strip()

    moduli = heamd.generate_primes([55] * 4, False, 8192)
    ctx = heamd.PolyContext(8192, moduli)
    bound = torch.tensor(moduli, dtype=torch.int64, device="cuda").view(1, -1, 1)
    slab = torch.randint(0, 1 << 62, (4096, 4, 8192), dtype=torch.int64, device="cuda") % bound

    def transform():
        launches, t0 = 0, time.perf_counter()
        while time.perf_counter() - t0 < SECONDS:
            for _ in range(200):
                ctx.forward_ntt_(slab)
            torch.cuda.synchronize()
            launches += 200
        return "forward NTT %.4f ms per launch" % ((time.perf_counter() - t0) / launches * 1e3)

    variant = os.path.join(ROOT, "swift-homomorphic-encryption_amd", "lib", "variants", "libhe_amd_timing_only_no_butterflies.so")
    transform_script = (
        "import sys, time; sys.path[:0] = [%r, %r]; import torch, heamd\n"
        "moduli = heamd.generate_primes([55] * 4, False, 8192); ctx = heamd.PolyContext(8192, moduli)\n"
        "bound = torch.tensor(moduli, dtype=torch.int64, device='cuda').view(1, -1, 1)\n"
        "slab = torch.randint(0, 1 << 62, (4096, 4, 8192), dtype=torch.int64, device='cuda') %% bound\n"
        "n, t0 = 0, time.perf_counter()\n"
        "while time.perf_counter() - t0 < %f:\n"
        "    for _ in range(200): ctx.forward_ntt_(slab)\n"
        "    torch.cuda.synchronize(); n += 200\n"
        "print('forward NTT %%.4f ms per launch' %% ((time.perf_counter() - t0) / n * 1e3))\n"
        % (ROOT, os.path.join(ROOT, "swift-homomorphic-encryption_amd"), SECONDS))

    def rows_only():  # the transform's rows, exchanges and gathers without its butterflies (a TIMING-ONLY variant library)
        env = dict(os.environ, HEAMD_LIBRARY=variant)
        return subprocess.run([sys.executable, "-c", transform_script], capture_output=True, text=True, env=env).stdout.strip()

    legs = {}
    runs = [("butterflies", binary("butterflies")), ("fold", binary("fold")), ("copy", binary("copy")), ("ntt", transform)]
    if os.path.exists(variant):
        runs.append(("rows only", rows_only))
    for name, run in runs:
        legs[name] = measured(run)
        time.sleep(1.0)
        print("%-12s %-36s %6.0f MHz  %7.1f W  (%d samples)" % ((name,) + legs[name]))
    rate_c = float(re.search(r"([0-9.]+) TB/s", legs["copy"][0]).group(1)) * 1e12
    t_ntt = float(re.search(r"([0-9.]+) ms", legs["ntt"][0]).group(1)) * 1e-3
    butterflies_per_launch = 4096 * 4 * 13 * 4096  # rows x stages x N / 2
    bytes_per_launch = 2 * 4096 * 4 * 8192 * 8
    t_c = bytes_per_launch / rate_c
    e_c, e_ntt, cap = t_c * legs["copy"][2], t_ntt * legs["ntt"][2], legs["ntt"][2]
    print("the transform: %.3f ms, %.3f J | its slab copied alone: %.3f ms, %.3f J" % (t_ntt * 1e3, e_ntt, t_c * 1e3, e_c))
    for name, what in (("butterflies", "limb-wise products (rounds 2-4)"), ("fold", "shift-folded products (production)")):
        rate_b = float(re.search(r"([0-9.]+) T/s", legs[name][0]).group(1)) * 1e12
        t_b = butterflies_per_launch / rate_b
        e_b = t_b * legs[name][2]
        print("%s: one launch's butterflies alone %.3f ms, %.3f J; (E_butterflies + E_copy) / P(transform) = %.3f ms; "
              "max(t_butterflies, t_copy) = %.3f ms; measured %.3f ms" % (what, t_b * 1e3, e_b, (e_b + e_c) / cap * 1e3,
                                                                       max(t_b, t_c) * 1e3, t_ntt * 1e3))
    if "rows only" in legs:
        t_rows = float(re.search(r"([0-9.]+) ms", legs["rows only"][0]).group(1)) * 1e-3
        e_rows = t_rows * legs["rows only"][2]
        rate_b = float(re.search(r"([0-9.]+) T/s", legs["fold"][0]).group(1)) * 1e12
        e_b = butterflies_per_launch / rate_b * legs["fold"][2]
        print("rows, exchanges and gathers without butterflies: %.3f ms, %.3f J; + the fold butterflies' energy = %.3f J -> "
              "%.3f ms at the transform's power; measured %.3f ms" % (t_rows * 1e3, e_rows, e_rows + e_b, (e_rows + e_b) / cap * 1e3,
                                                                      t_ntt * 1e3))


if __name__ == "__main__":
    main()
