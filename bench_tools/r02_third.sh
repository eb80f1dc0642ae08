#!/bin/bash
# GPU pass 3: parity, inverse variants, NTT counters, C3 with/without XCD-local replica sets (+ HBM counters)
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/pmc_c3 gpurun_out/pmc_c3_noxcd
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest.log
tail -4 gpurun_out/pytest.log
V=swift-homomorphic-encryption_amd/lib/variants
for name in inv_late inv_noahead_late inv_noahead; do
  HEAMD_LIBRARY=$V/libhe_amd_$name.so timeout 300 python -m pytest tests/test_gpu_ntt.py -m gpu -x -q > gpurun_out/pytest_$name.log 2>&1
  echo "$name: $(tail -1 gpurun_out/pytest_$name.log)"
done
timeout 600 python bench_tools/ab_variants.py run inv_late inv_noahead_late inv_noahead > gpurun_out/ab.txt 2>&1
cat gpurun_out/ab.txt
timeout 600 python bench_tools/ab_c3.py no_xcd > gpurun_out/ab_c3.txt 2>&1
cat gpurun_out/ab_c3.txt
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  n=$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_c3/$n -- python bench_tools/c3_profile_target.py > gpurun_out/pmc_c3/$n.log 2>&1 || echo "pmc $c failed"
  HEAMD_LIBRARY=$V/libhe_amd_no_xcd.so timeout 300 rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmc_c3_noxcd/$n -- python bench_tools/c3_profile_target.py > gpurun_out/pmc_c3_noxcd/$n.log 2>&1 || echo "pmc noxcd $c failed"
done
echo "== XCD-local replica sets"; python bench_tools/pmc_traffic.py gpurun_out/pmc_c3 2>&1 | tail -30
echo "== plain order"; python bench_tools/pmc_traffic.py gpurun_out/pmc_c3_noxcd 2>&1 | tail -30
FILTER=ntt_ timeout 600 bash bench_tools/pmc_passes.sh gpurun_out/pmc_ntt 0 > gpurun_out/pmc_ntt.log 2>&1
tail -60 gpurun_out/pmc_ntt/summary.txt
