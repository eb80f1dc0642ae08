#!/bin/bash
# Round 6: (a) what binds the N = 16384 transforms -- timing-only variants of the interleaved kernels against production;
# (b) the forward kernel's power model on the production arithmetic; (c) c5 through a device group in one process.
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
T=${1:-r06j}; O=gpurun_out/$T; mkdir -p $O
python bench_tools/ab_variants.py run --what degrees --rounds 3 timing_only_no_split_gathers timing_only_no_cross_stage timing_only_no_lds_exchange > $O/ab_16384.txt 2>&1
cat $O/ab_16384.txt
python bench_tools/power_probe.py > $O/power_probe.txt 2>&1; grep -v amdgpu.ids $O/power_probe.txt
timeout 600 python bench.py --workload c5 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c5.json 2> $O/bench_c5.err
timeout 600 python bench.py --workload c5 --device-group 2 --steps 10 --warmup 2 --no-cpu-baseline > $O/bench_c5_group2.json 2> $O/bench_c5_group2.err || tail -5 $O/bench_c5_group2.err
python - <<'PY'
import json
for n in ("bench_c5", "bench_c5_group2"):
    try:
        d = json.loads([l for l in open("gpurun_out/r06j/%s.json" % n) if l.startswith("{")][-1])
        print(n, "%.1f M ct-pt-mac/s  %.3f ms/step  frac %.3f" % (d["value"] / 1e6, d["ms_per_step"], d["roofline"]["frac"]), d["config"]["parallelism"][:90])
    except Exception as e:
        print(n, "failed", e)
PY
