#!/bin/bash
# GPU pass 5: parity with the streamed (non-temporal) policies, A/B against the cached policies per workload
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 1200 python -m pytest tests -m gpu -q > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest.log
tail -4 gpurun_out/pytest.log
timeout 600 python bench_tools/ab_variants.py run cached_rows > gpurun_out/ab_rows.txt 2>&1; cat gpurun_out/ab_rows.txt
timeout 900 python bench_tools/ab_workloads.py c3,c4,c5 cached_poly cached_rns cached_rows > gpurun_out/ab_workloads.txt 2>&1; cat gpurun_out/ab_workloads.txt
timeout 600 python bench_tools/elementwise_bench.py > gpurun_out/elementwise.txt 2>&1; tail -8 gpurun_out/elementwise.txt
