cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp; O=gpurun_out/r06d; mkdir -p $O
for k in 0 4; do
  HEAMD_PIR_PIECE_CHUNKS=$k timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace_$k -- python bench_tools/pir_loop_profile_target.py > $O/loop_$k.txt 2>&1
  f=$(find $O/trace_$k -name "*kernel_trace.csv" | head -1)
  python bench_tools/kernel_gaps.py "$f" inner_product_plain_rows_kernel 80 > $O/gaps_$k.txt 2>&1
  tail -3 $O/loop_$k.txt; cat $O/gaps_$k.txt
  rm -rf $O/trace_$k
done
