#!/bin/bash
# Issue / stall / memory counters of the headline transforms (bench_tools/ntt_profile_target.py 0: N = 8192, L = 4, 4096 polynomials),
# production and the variant libraries named in NTT_PMC_VARIANTS -- separate --pmc passes, no trace domains alongside.
# bash bench_tools/ntt_pmc.sh TAG
cd "$GRAFT_REPO_ROOT"
T=${1:-r05pmc}
O=gpurun_out/$T
mkdir -p $O
export TMPDIR=/tmp
for v in production ${NTT_PMC_VARIANTS:-}; do
  lib=""; [ "$v" != production ] && lib=$PWD/swift-homomorphic-encryption_amd/lib/variants/libhe_amd_$v.so
  i=0
  for g in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD" \
           "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
    i=$((i+1))
    env ${lib:+HEAMD_LIBRARY=$lib} timeout 300 rocprofv3 --pmc $g --output-format csv -d $O/pmc_$v/pass$i -- python bench_tools/ntt_profile_target.py 0 > $O/pmc_${v}_pass$i.log 2>&1 || echo "pmc $v pass $i failed"
  done
  (echo "#### $v"; python bench_tools/pmc_summary.py $O/pmc_$v ntt_forward_tiled; python bench_tools/pmc_summary.py $O/pmc_$v ntt_inverse_tiled) >> $O/ntt_pmc_summary.txt 2>&1
  rm -rf $O/pmc_$v
done
cat $O/ntt_pmc_summary.txt
