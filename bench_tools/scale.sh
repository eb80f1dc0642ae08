#!/bin/bash
# Scaling sweep on ONE node: every BASELINE workload at 1 / 2 / 4 / 8 ranks (or "$RANKS"), one JSON line each, appended
# to $OUT (default gpurun_out/scale.jsonl).  Weak scaling: the per-GPU size of the config on every rank; `value` counts
# all ranks' units; extras.all_gather_ms / all_gather_bytes_per_gpu / value_with_all_gather report the only collective
# (the RCCL all-gather of the result shards -- C2 128 MiB, C3 64 MiB, C5 64 MiB per GPU at the BASELINE sizes).
#   bash bench_tools/scale.sh [workload ...]        e.g.  RANKS="1 2" bash bench_tools/scale.sh c2 c5
# UNMEASURED in this repository's build environment (gpurun hands out one GPU): run it where a node exists.
cd "$(dirname "$0")/.."
OUT=${OUT:-gpurun_out/scale.jsonl}
mkdir -p "$(dirname "$OUT")"
WORKLOADS=${*:-c2 c3 c4 c5}
export HSA_ENABLE_IPC_MODE_LEGACY=0
available=$(python -c 'import torch; print(torch.cuda.device_count())')
for workload in $WORKLOADS; do
  for n in ${RANKS:-1 2 4 8}; do
    if [ "$n" -gt "$available" ]; then echo "skip $workload x $n: $available GPU(s) visible" >&2; continue; fi
    if [ "$n" -eq 1 ]; then
      python bench.py --gpus 1 --workload $workload --steps 10 --warmup 3 --skip-other-configs --no-cpu-baseline
    else
      python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 \
        --master-port $((29500 + n)) bench.py --gpus $n --workload $workload --steps 10 --warmup 3
    fi | tail -1 | tee -a "$OUT" | python -c 'import json,sys; r=json.loads(sys.stdin.read()); print(r["config"]["workload"][:40], "n =", r["n_gpus"], "value =", "%.4g" % r["value"], r["unit"], "gather ms =", r["extras"].get("all_gather_ms"))'
  done
done
