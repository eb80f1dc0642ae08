#!/bin/bash
# Round 6: (a) dynamic vector-ALU instruction counts of the ct x ct + relinearize kernels (the numerator of bench.py's VALU
# roofline), (b) issue / stall / LDS / L2 counters of the N = 16384 transforms beside the N = 8192 pair, (c) the instruction
# rates of this box.   bash bench_tools/r06_counters.sh TAG
cd "$GRAFT_REPO_ROOT"
T=${1:-r06i}
O=gpurun_out/$T
mkdir -p $O
export TMPDIR=/tmp
./bench_tools/microbench > $O/microbench.txt 2>&1
grep -E "v_mad_u64_u32|v_add_u32|v_lshl_add_u64|v_cndmask" $O/microbench.txt | grep "waves/SIMD=8"
# (a)
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVES SQ_WAVE_CYCLES SQ_THREAD_CYCLES_VALU \
  --output-format csv -d $O/c3/pass1 -- python bench_tools/c3_profile_target.py > $O/c3_pass1.log 2>&1 || echo "c3 pass failed"
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY \
  --output-format csv -d $O/c3/pass2 -- python bench_tools/c3_profile_target.py > $O/c3_pass2.log 2>&1 || echo "c3 pass 2 failed"
(for k in lift_kernel behz_rows_fused floor_kernel ntt_forward_tiled ntt_inverse_tiled; do python bench_tools/pmc_summary.py $O/c3 $k; done) > $O/c3_valu_counters.txt 2>&1
rm -rf $O/c3
cat $O/c3_valu_counters.txt
# (b)
for shape in "16384 1024" "8192 4096"; do
  set -- $shape
  i=0
  for g in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM_RD SQ_WAVES SQ_LDS_ADDR_CONFLICT" \
           "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum" "FETCH_SIZE" "WRITE_SIZE"; do
    i=$((i+1))
    NTT_DEGREE=$1 NTT_BATCH=$2 timeout 300 rocprofv3 --pmc $g --output-format csv -d $O/ntt_$1/pass$i -- python bench_tools/ntt_profile_target.py 0 > $O/ntt_$1_pass$i.log 2>&1 || echo "ntt $1 pass $i failed"
  done
  (echo "#### N = $1, batch $2"; python bench_tools/pmc_summary.py $O/ntt_$1 ntt_forward; python bench_tools/pmc_summary.py $O/ntt_$1 ntt_inverse) >> $O/ntt_counters.txt 2>&1
  rm -rf $O/ntt_$1
done
cat $O/ntt_counters.txt
