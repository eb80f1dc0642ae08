"""Vector-ALU instruction counts per ct x ct + relinearize product from a `pmc_summary.py` digest of the rocprofv3 --pmc
pass SQ_INSTS_VALU / SQ_INSTS_VALU_INT64 / SQ_INSTS_VALU_INT32 over bench_tools/c3_profile_target.py (batch 1024):

    python bench_tools/valu_json.py gpurun_out/<call>/c3_valu_counters.txt profiles/r06_c3_valu.json

Counts are wave instructions per launch summed over the device; a kernel's launches per product call are given below (the lift
runs once per operand, the Bsk band and the floor in two parts of the batch).  What bench.py's VALU roofline replays
(`counts_live: false`) the way it replays the HBM counter bytes."""
import json
import re
import sys

BATCH = 1024
# launches per call of (ct x ct, relinearize) over the whole batch; the parts of a kernel that runs in parts add up to the batch
LAUNCHES = {
    "lift_kernel<4, unsigned long, true, 2>": 1,       # both operands in one launch (launch_lift_pair_q_to_qbsk_strided)
    "behz_rows_fused<13, 10, 4, 4>": 1,
    "behz_rows_fused<13, 10, 6, 6>": 1,                # two launches of half the batch each: per-launch averages x 2 / 2
    "floor_kernel<4, unsigned long, true, 1>": 1,      # likewise
    "ntt_forward_tiled<13, 10, 4, 1, 2>": 1,
    "ntt_inverse_tiled<13, 10, 4, 2, 1>": 1,           # the q_ks rows: 2048 rows, one per workgroup (kUngroupedBelowRows)
    "ntt_inverse_tiled<13, 10, 4, 4, 2>": 1,
}
HALVES = {"behz_rows_fused<13, 10, 6, 6>", "floor_kernel<4, unsigned long, true, 1>"}  # a launch covers half the batch


def main():
    text = open(sys.argv[1]).read()
    kernels = {}
    for block in re.split(r"^== ", text, flags=re.M)[1:]:
        name = block.split("  grid=")[0].strip()
        counters = {m.group(1): float(m.group(2)) for m in re.finditer(r"^\s+(\w+)\s+([0-9.]+)\s+\(n=", block, re.M)}
        kernels[name] = counters
    out = {"source": sys.argv[1], "batch": BATCH, "unit": "wave instructions per product (x 64 = lane instructions)",
           "per_kernel": {}}
    totals = {"valu": 0.0, "int64": 0.0, "int32": 0.0}
    for name, launches in LAUNCHES.items():
        c = kernels[name]
        units = BATCH / 2 if name in HALVES else BATCH  # products one launch covers
        row = {"valu": c["SQ_INSTS_VALU"] * launches / units, "int64": c["SQ_INSTS_VALU_INT64"] * launches / units,
               "int32": c["SQ_INSTS_VALU_INT32"] * launches / units, "waves_per_launch": c["SQ_WAVES"]}
        out["per_kernel"][name] = row
        for k in totals:
            totals[k] += row[k]
    out["kernel"] = list(LAUNCHES)
    out["valu_wave_instructions_per_product"] = totals["valu"]
    out["int64_wave_instructions_per_product"] = totals["int64"]
    out["int32_wave_instructions_per_product"] = totals["int32"]
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print(json.dumps({k: round(v) for k, v in totals.items()}))


if __name__ == "__main__":
    main()
