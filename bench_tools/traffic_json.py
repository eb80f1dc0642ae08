"""Builds profiles/r06_pmc_traffic.json (read by bench.py / path_bench.py) from the per-kernel traffic.json files that
bench_tools/pmc_traffic.py leaves in the PMC pass directories of the four workloads.

  python bench_tools/traffic_json.py <ntt-dir> <c3-dir> <c4-dir> <c5-dir> <out.json>

Units: c2 = one forward launch over 4096 polynomials; c3 = one ct x ct + relinearize; c4 = one polynomial; c5 = one
ct x pt multiply-accumulate.  Bytes = 2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), gfx950 correction as in
/opt/skills/guides/MI355X_MICROARCH.md.
"""
import json
import sys


def load(path):
    with open(path + "/traffic.json") as f:
        return json.load(f)


def bytes_per_dispatch(report, needle, exclude=()):
    total = 0.0
    hits = []
    for kernel, row in report.items():
        if kernel.startswith("_") or needle not in kernel or any(x in kernel for x in exclude):
            continue
        total += (row["fetch_KiB_per_dispatch"] + row["write_KiB_per_dispatch"]) * 1024.0
        hits.append(kernel)
    return total, hits


def main():
    ntt_dir, c3_dir, c4_dir, c5_dir, out = sys.argv[1:6]
    source = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (bench_tools/r06_final.sh -> pmc_traffic.py), FETCH_SIZE x2 on gfx950"
    result = {}
    ntt = load(ntt_dir)
    forward, names = bytes_per_dispatch(ntt, "ntt_forward_tiled")
    inverse, _ = bytes_per_dispatch(ntt, "ntt_inverse_tiled")
    result["c2_forward_ntt"] = {"kernel": names, "units_per_launch": 4096, "hbm_bytes_per_launch": forward,
                                "algorithmic_bytes_per_launch": 2 * 4 * 8192 * 8 * 4096,
                                "inverse_hbm_bytes_per_launch": inverse, "source": source}
    c3 = load(c3_dir)
    # How many times a kernel is launched per CALL of its pipeline comes from the dispatch counts themselves: the profile
    # target runs ct x ct and relinearize some number of times each; a pipeline's calls = the smallest dispatch count among its
    # kernels (the kernels launched once per call), a kernel's launches per call = its dispatches over that (the lift: 2; since
    # round 5 the Bsk band of the row-fused kernel and the floor go in two parts of the batch: 2 each).
    mul_kernels = ("lift_kernel", "floor_kernel", "behz_rows_fused", "ntt_forward_tiled<13, 10, 3, 3", "ntt_forward_tiled<13, 10, 4, 3", "ntt_forward_tiled<13, 10, 6, 0",
                   "ntt_inverse_tiled<13, 10, 6, 1", "ntt_inverse_tiled<13, 10, 7, 1", "ntt_inverse_tiled<13, 10, 4, 1", "tensor_kernel")
    # (only the pipeline's own kernels: the profile target may have run other device code beside it -- PyTorch's generators, a
    # tool in a child process)
    pipeline = ("lift_kernel", "floor_kernel", "behz_rows_fused", "ntt_forward_", "ntt_inverse_", "tensor_", "key_switch_")
    rows = {k: r for k, r in c3.items() if any(k.startswith(name) for name in pipeline)}
    is_mul = {k: any(m in k for m in mul_kernels) for k in rows}
    calls = {True: min((r["dispatches"] for k, r in rows.items() if is_mul[k]), default=1),
             False: min((r["dispatches"] for k, r in rows.items() if not is_mul[k]), default=1)}
    per_batch = 0.0
    detail, launches = {}, {}
    for kernel, row in rows.items():
        b = (row["fetch_KiB_per_dispatch"] + row["write_KiB_per_dispatch"]) * 1024.0
        times = row["dispatches"] / calls[is_mul[kernel]]
        per_batch += b * times
        detail[kernel] = b * times / 1024
        launches[kernel] = round(times, 3)
    result["c3_ct_mul_relinearize"] = {"hbm_bytes_per_unit": per_batch / 1024, "algorithmic_bytes_per_unit": 1572864,
                                       "per_kernel_bytes_per_unit": detail, "launches_per_call": launches, "batch": 1024,
                                       "source": source}
    c4 = load(c4_dir)
    b, names = bytes_per_dispatch(c4, "divide_and_round")
    result["c4_mod_switch"] = {"kernel": names, "hbm_bytes_per_unit": b / 8192,
                               "algorithmic_bytes_per_unit": 11 * 16384 * 8, "batch": 8192, "source": source}
    c5 = load(c5_dir)
    b, names = bytes_per_dispatch(c5, "inner_product_plain")
    result["c5_inner_product_plain"] = {"kernel": names, "hbm_bytes_per_unit": b / (1024 * 128),
                                        "algorithmic_bytes_per_unit": 4 * 8192 * 8, "rows": 1024, "columns": 128,
                                        "source": source}
    with open(out, "w") as f:
        json.dump(result, f, indent=1)
    print(json.dumps(result, indent=1))


if __name__ == "__main__":
    main()
