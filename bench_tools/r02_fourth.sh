#!/bin/bash
# GPU pass 4: parity (new tests), bench workloads, nt-policy A/B, HBM counters for c2/c4/c5, C3 kernel stats
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest.log
tail -6 gpurun_out/pytest.log
timeout 600 python bench_tools/ab_variants.py run nt_load nt_store nt_both > gpurun_out/ab_nt.txt 2>&1
cat gpurun_out/ab_nt.txt
for w in c3 c4 c5; do
  timeout 600 python bench.py --workload $w --steps 10 --warmup 2 > gpurun_out/bench_$w.json 2> gpurun_out/bench_$w.err || tail -5 gpurun_out/bench_$w.err
  cut -c1-600 gpurun_out/bench_$w.json
done
pmc() {  # name target
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | cut -d' ' -f1)
    mkdir -p gpurun_out/pmc_$1
    timeout 600 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_$1/$n -- python $2 > gpurun_out/pmc_$1/$n.log 2>&1 || echo "pmc $1 $c failed"
  done
  python bench_tools/pmc_traffic.py gpurun_out/pmc_$1 | grep -v "at::\|rocclr"
}
pmc c2 "bench_tools/ntt_profile_target.py 0"
pmc c3 bench_tools/c3_profile_target.py
pmc c4 bench_tools/c4_profile_target.py
pmc c5 bench_tools/c5_profile_target.py
python bench_tools/traffic_json.py gpurun_out/pmc_c2 gpurun_out/pmc_c3 gpurun_out/pmc_c4 gpurun_out/pmc_c5 gpurun_out/r02_pmc_traffic.json > /dev/null
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c3_stats -- python bench_tools/c3_profile_target.py > gpurun_out/c3_stats.log 2>&1
f=$(find gpurun_out/c3_stats -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/c3_kernel_stats.csv; python bench_tools/kernel_stats_summary.py gpurun_out/c3_kernel_stats.csv | head -16
