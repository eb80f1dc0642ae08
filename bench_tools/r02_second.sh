#!/bin/bash
# GPU pass: parity of the new inverse last stage / reducer, then A/B of kernel variants, then C3 HBM counters
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest.log 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest.log
tail -3 gpurun_out/pytest.log
V=swift-homomorphic-encryption_amd/lib/variants
for name in rows4 inv_late; do
  HEAMD_LIBRARY=$V/libhe_amd_$name.so timeout 300 python -m pytest tests/test_gpu_ntt.py -m gpu -x -q > gpurun_out/pytest_$name.log 2>&1
  echo "$name: $(tail -1 gpurun_out/pytest_$name.log)"
done
timeout 900 python bench_tools/ab_variants.py run > gpurun_out/ab.txt 2>&1
cat gpurun_out/ab.txt
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
  d=gpurun_out/pmc_c3/$(echo $c | cut -d' ' -f1)
  timeout 300 rocprofv3 --pmc $c --output-format csv -d $d -- python bench_tools/c3_profile_target.py > $d.log 2>&1 || echo "pmc $c failed"
done
python bench_tools/pmc_traffic.py gpurun_out/pmc_c3 2>&1 | tail -40
