#!/bin/bash
# Round-4 A/B call: parity (all GPU tests, incl. the benched-size C4 / C5 ones), then the key-MAC and inverse-NTT variants
# against the production library, then a kernel trace of the C3 pipeline.   bash bench_tools/r04_ab.sh TAG
cd "$GRAFT_REPO_ROOT"
T=${1:-r04a}
O=gpurun_out/$T
mkdir -p $O
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
# every variant library under lib/variants/ is timed (C3_VARIANTS / NTT_VARIANTS restrict the lists; "none" skips a leg)
if [ "${C3_VARIANTS:-}" != "none" ]; then
  timeout 900 python bench_tools/ab_variants.py run --what c3 --rounds ${ROUNDS:-3} ${C3_VARIANTS:-} > $O/ab_c3.txt 2>&1; cat $O/ab_c3.txt
fi
if [ "${NTT_VARIANTS:-}" != "none" ]; then
  timeout 900 python bench_tools/ab_variants.py run --what ${NTT_WHAT:-ntt} --rounds ${ROUNDS:-3} ${NTT_VARIANTS:-} > $O/ab_ntt.txt 2>&1; cat $O/ab_ntt.txt
fi
if [ -n "${PARAM_SETS:-}" ]; then  # the 60-bit parameter sets, production and every variant library
  (echo "production: $(timeout 300 python bench_tools/param_sets_bench.py 2>&1 | tail -1)"
   for lib in swift-homomorphic-encryption_amd/lib/variants/libhe_amd_*.so; do
     [ -e "$lib" ] && echo "$(basename $lib .so | sed s/libhe_amd_//): $(HEAMD_LIBRARY=$PWD/$lib timeout 300 python bench_tools/param_sets_bench.py 2>&1 | tail -1)"
   done) > $O/param_sets.txt; cat $O/param_sets.txt
fi
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c3_stats -- python bench_tools/c3_profile_target.py > $O/c3_stats.log 2>&1
f=$(find $O/c3_stats -name "*kernel_stats.csv" | head -1); cp "$f" $O/c3_kernel_stats.csv; python bench_tools/kernel_stats_summary.py $O/c3_kernel_stats.csv | head -16
rm -rf $O/c3_stats
if [ -n "${TRACE_C5:-}" ]; then  # the timeline of one PIR chunk response: kernels against the gaps between them
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/c5_trace -- python bench_tools/pir_profile_target.py > $O/c5_trace.log 2>&1
  f=$(find $O/c5_trace -name "*kernel_trace.csv" | head -1)
  python bench_tools/kernel_gaps.py "$f" inner_product_plain 80 > $O/c5_chunk_timeline.txt 2>&1; tail -45 $O/c5_chunk_timeline.txt
  rm -rf $O/c5_trace
fi
