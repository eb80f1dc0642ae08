#!/bin/bash
# Bfv<UInt32> pipeline (n_4096_logq_27_28_28): kernel stats of production and of a variant library, one call.
#   bash bench_tools/w32_ab.sh [variant]        -> gpurun_out/w32_{production,variant}_kernel_stats.csv
O=gpurun_out
mkdir -p $O
for name in production ${1:-}; do
  lib=; [ $name != production ] && lib=$PWD/swift-homomorphic-encryption_amd/lib/variants/libhe_amd_$name.so
  HEAMD_LIBRARY=$lib timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/w32_$name -- python bench_tools/word32_profile_target.py > $O/w32_$name.log 2>&1
  cp $(find $O/w32_$name -name "*kernel_stats.csv" | head -1) $O/w32_${name}_kernel_stats.csv
  echo "== $name"; head -14 $O/w32_${name}_kernel_stats.csv | cut -c1-150
done
