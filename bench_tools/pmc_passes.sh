#!/bin/bash
# PMC passes (run on the GPU box):  [TARGET=script.py FILTER=substr] bash bench_tools/pmc_passes.sh <out-dir> <args ...>
# default target: the NTT kernels (bench_tools/ntt_profile_target.py <variant ...>)
# One rocprofv3 run per counter group (SQ has 8 slots, TCC 4); no trace domains next to --pmc.
set -u
export TMPDIR=/tmp
OUT=${1:-gpurun_out/pmc}; shift
VARIANTS=${*:-9}
mkdir -p "$OUT"
GROUPS_=(
  "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"
  "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
  "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_WAVES SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL"
  "TCC_HIT_sum TCC_MISS_sum"
  "FETCH_SIZE"
  "WRITE_SIZE"
  "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum"
)
i=0
for g in "${GROUPS_[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $g --output-format csv -d "$OUT/pass$i" -- python ${TARGET:-bench_tools/ntt_profile_target.py} $VARIANTS > "$OUT/pass$i.log" 2>&1 || echo "pass $i failed (see $OUT/pass$i.log)"
done
python bench_tools/pmc_summary.py "$OUT" ${FILTER:-ntt_} | tee "$OUT/summary.txt"
