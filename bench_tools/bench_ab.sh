for r in 1 2; do for lib in "" swift-homomorphic-encryption_amd/lib/variants/libhe_amd_inv_unsigned_difference.so; do
  HEAMD_LIBRARY=${lib:+$PWD/$lib} python bench.py --gpus 1 --steps 20 --warmup 5 --skip-other-configs --no-cpu-baseline 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.readline()); r=d['roofline']; e=d['extras']
print('${lib:-production}'[-40:], 'value %.3fM no-pre %.3fM | fwd %.4f (%.4f) | inv %.4f (%.4f) | sclk %.0f'%(d['value']/1e6,d['value_without_pre_roll']/1e6,r['avg_launch_ms'],r['frac'],e['inverse_avg_launch_ms'],e['inverse_frac_of_8TBps'],r['under_load']['sclk_mhz']))"
done; done
