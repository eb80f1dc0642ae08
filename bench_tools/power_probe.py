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

    legs = {}
    for name, run in (("butterflies", binary("butterflies")), ("copy", binary("copy")), ("ntt", transform)):
        legs[name] = measured(run)
        time.sleep(1.0)
        print("%-12s %-36s %6.0f MHz  %7.1f W  (%d samples)" % ((name,) + legs[name]))
    rate_b = float(re.search(r"([0-9.]+) T/s", legs["butterflies"][0]).group(1)) * 1e12
    rate_c = float(re.search(r"([0-9.]+) TB/s", legs["copy"][0]).group(1)) * 1e12
    t_ntt = float(re.search(r"([0-9.]+) ms", legs["ntt"][0]).group(1)) * 1e-3
    butterflies_per_launch = 4096 * 4 * 13 * 4096  # rows x stages x N / 2
    bytes_per_launch = 2 * 4096 * 4 * 8192 * 8
    t_b, t_c = butterflies_per_launch / rate_b, bytes_per_launch / rate_c
    e_b, e_c, e_ntt = t_b * legs["butterflies"][2], t_c * legs["copy"][2], t_ntt * legs["ntt"][2]
    cap = legs["ntt"][2]
    print("one launch's butterflies alone: %.3f ms, %.3f J | its slab copied alone: %.3f ms, %.3f J | the transform: %.3f ms, %.3f J"
          % (t_b * 1e3, e_b, t_c * 1e3, e_c, t_ntt * 1e3, e_ntt))
    print("(E_butterflies + E_copy) / P(transform) = %.3f ms; max(t_butterflies, t_copy) = %.3f ms; measured %.3f ms"
          % ((e_b + e_c) / cap * 1e3, max(t_b, t_c) * 1e3, t_ntt * 1e3))


if __name__ == "__main__":
    main()
