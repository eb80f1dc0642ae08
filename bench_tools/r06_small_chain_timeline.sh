#!/bin/bash
# Round 6: the kernels of ct x ct + relinearize on 1 and 8 ciphertexts in launch order (chains of small launches)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
T=${1:-r06x}; O=gpurun_out/$T; mkdir -p $O
for b in 1 8; do
  CHAIN_BATCH=$b rocprofv3 --kernel-trace --output-format csv -d $O/trace_$b -- python bench_tools/small_chain_profile_target.py > $O/chain_$b.log 2>&1
  csv=$(find $O/trace_$b -name "*kernel_trace.csv" | head -1)
  python bench_tools/timeline_digest.py "$csv" 24 > $O/chain_timeline_$b.txt
  echo "== batch $b"; cat $O/chain_timeline_$b.txt
  rm -rf $O/trace_$b
done
