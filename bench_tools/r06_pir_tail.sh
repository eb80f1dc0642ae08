#!/bin/bash
# Round 6: the chunk loop's tail (remaining dimensions of 8 chunks of 256 x 64) kernel by kernel, and the loop's spread
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
T=${1:-r06y}; O=gpurun_out/$T; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/trace -- python bench_tools/pir_loop_profile_target.py > $O/loop.txt 2>&1
f=$(find $O/trace -name "*kernel_trace.csv" | head -1)
python bench_tools/kernel_gaps.py "$f" inner_product_plain_rows_kernel 80 > $O/gaps.txt 2>&1
tail -3 $O/loop.txt; cat $O/gaps.txt
rm -rf $O/trace
python bench_tools/whole_query_spread.py > $O/whole_query_spread.txt 2>&1; grep -v amdgpu.ids $O/whole_query_spread.txt | cut -c1-200
