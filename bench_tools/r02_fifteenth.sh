#!/bin/bash
# ct x ct with the Q band of the forward transform read from the ciphertexts (the lift no longer copies them)
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02za; mkdir -p $O; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_bfv.py tests/test_gpu_fuzz.py tests/test_gpu_pir.py -q -x -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 600 python bench.py --workload c3 --steps 10 --warmup 2 > $O/bench_c3.json 2> $O/bench_c3.err || tail -5 $O/bench_c3.err
cut -c1-600 $O/bench_c3.json
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_stats -- python bench_tools/c3_profile_target.py > $O/c3_stats.log 2>&1
f=$(find $O/c3_stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/c3_kernel_stats.csv; python bench_tools/kernel_stats_summary.py $O/c3_kernel_stats.csv | head -12
rm -rf $O/c3_stats
