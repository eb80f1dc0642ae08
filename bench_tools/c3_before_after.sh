#!/bin/bash
# ct x ct + relinearize kernel by kernel, BEFORE (a variant library: HEAMD_LIBRARY) and AFTER (the production library), from
# ONE call: rocprofv3 kernel stats (average microseconds per 1024 products) and FETCH_SIZE / WRITE_SIZE counters (bytes per
# product) of bench_tools/c3_profile_target.py for each.   bash bench_tools/c3_before_after.sh TAG VARIANT
cd "$GRAFT_REPO_ROOT"
T=${1:-r04h}; V=${2:-keymac_separate_finish}
O=gpurun_out/$T; mkdir -p $O
export TMPDIR=/tmp
run() {  # name [library]
  local name=$1 lib=$2
  local env=""; [ -n "$lib" ] && env="HEAMD_LIBRARY=$PWD/$lib"
  env $env timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${name}_stats -- python bench_tools/c3_profile_target.py > $O/${name}_stats.log 2>&1
  cp "$(find $O/${name}_stats -name '*kernel_stats.csv' | head -1)" $O/${name}_kernel_stats.csv
  for c in FETCH_SIZE WRITE_SIZE; do
    env $env timeout 600 rocprofv3 --pmc $c --output-format csv -d $O/${name}_pmc/$c -- python bench_tools/c3_profile_target.py > $O/${name}_pmc_$c.log 2>&1
  done
  python bench_tools/pmc_traffic.py $O/${name}_pmc 1024 > $O/${name}_pmc.txt
  cp $O/${name}_pmc/traffic.json $O/${name}_traffic.json
  rm -rf $O/${name}_stats $O/${name}_pmc
}
run before swift-homomorphic-encryption_amd/lib/variants/libhe_amd_$V.so
run after
python bench_tools/c3_table.py $O/before_kernel_stats.csv $O/before_traffic.json $O/after_kernel_stats.csv $O/after_traffic.json "$V" > $O/c3_before_after.txt
cat $O/c3_before_after.txt
