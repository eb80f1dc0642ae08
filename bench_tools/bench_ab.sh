#!/bin/bash
# The driver's command with production and variant libraries alternating in one call (HEAMD_LIBRARY).
#   bash bench_tools/bench_ab.sh [variant ...]     (libraries under swift-homomorphic-encryption_amd/lib/variants/)
for r in 1 2; do for name in production "$@"; do
  lib=; [ $name != production ] && lib=$PWD/swift-homomorphic-encryption_amd/lib/variants/libhe_amd_$name.so
  HEAMD_LIBRARY=$lib python bench.py --gpus 1 --steps 20 --warmup 5 --skip-other-configs --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; e=d['extras']
print('%-26s value %.3fM no-pre %.3fM | fwd %.4f (%.4f) | inv %.4f (%.4f) | sclk %.0f'%('$name',d['value']/1e6,d['value_without_pre_roll']/1e6,r['avg_launch_ms'],r['frac'],e['inverse_avg_launch_ms'],e['inverse_frac_of_8TBps'],r['under_load']['sclk_mhz']))"
done; done
