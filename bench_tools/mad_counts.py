"""64-bit-integer multiply instructions per lane of the BUILT kernels (no GPU needed): the numerator of the vector-ALU
roofline bench.py states for ct x ct + relinearize.

  python bench_tools/mad_counts.py [--json out.json]

For each kernel of the ct x ct + relinearize pipeline at N = 8192, L = 4 (the names rocprofv3 lists,
profiles/r05z_c3_kernel_stats.csv) the gfx950 code object of the library's own objects is disassembled and its multiply
instructions (v_mad_u64_u32, v_mad_i64_i32, v_mul_lo_u32, v_mul_hi_u32: all issue at the v_mad_u64_u32 rate or slower,
profiles/r01_microbench_instruction_rates.txt) are counted: outside any loop once, inside a loop (a backward branch and its
target) times the loop's trip count -- given below per kernel where there is one, from the kernel's source.  Multiplies per
lane x lanes per launch x launches per product = multiplies per product.
"""
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "bench_tools"))
import kernel_metadata  # noqa: E402

BUILD = kernel_metadata.BUILD
MULTIPLIES = ("v_mad_u64_u32", "v_mad_i64_i32", "v_mul_lo_u32", "v_mul_hi_u32", "v_mul_u32_u24", "v_mad_u32_u24")
N, L = 8192, 4
EXT = 2 * L + 1

# kernel (demangled, shortened) -> object, lanes per product, loop trip counts by order of appearance (None: no loops expected)
# Lanes per product follow the launch geometry in the sources (one lane per 8 words of a row in the tiled transforms, ...).
PIPELINE = [
    # the lift (rns_kernels.hip lift_kernel<L, W, LAZY, POLYS = 2>): one lane per coefficient of a pair of polynomials; two
    # launches per product (lhs, rhs), each over its 2 polynomials -> N lanes per launch
    {"kernel": "lift_kernel<4, unsigned long, true, 2>", "object": "rns_kernels.o", "lanes_per_product": 2 * N, "trips": {}},
    # Q band: one 1024-lane workgroup per (item, Q row): 4 forward transforms, tensor, 3 inverse transforms
    {"kernel": "behz_rows_fused<13, 10, 4, 4>", "object": "behz_kernels.o", "lanes_per_product": L * 1024, "trips": {}},
    # Bsk band: the same per (item, Bsk row), L + 1 rows
    {"kernel": "behz_rows_fused<13, 10, 6, 6>", "object": "behz_kernels.o", "lanes_per_product": (L + 1) * 1024, "trips": {}},
    # the floor: one lane per coefficient of each of the 3 product polynomials
    {"kernel": "floor_kernel<4, unsigned long, true, 1>", "object": "rns_kernels.o", "lanes_per_product": 3 * N, "trips": {}},
    # relinearize: spread + forward, two rows per 1024-lane workgroup, L (L + 1) rows
    {"kernel": "ntt_forward_tiled<13, 10, 4, 1, 2>", "object": "ntt_kernels.o", "lanes_per_product": L * (L + 1) * 512, "trips": {}},
    # the q_ks row of both update polynomials: key inner product (L terms: the loop) + inverse, two rows per workgroup
    {"kernel": "ntt_inverse_tiled<13, 10, 4, 2, 2>", "object": "ntt_kernels.o", "lanes_per_product": 2 * 512, "trips": "L"},
    # the L rows below it with the key switch's end fused
    {"kernel": "ntt_inverse_tiled<13, 10, 4, 4, 2>", "object": "ntt_kernels.o", "lanes_per_product": 2 * L * 512, "trips": "L"},
]


def disassemble(obj, workdir):
    code = kernel_metadata.code_object(os.path.join(BUILD, obj), workdir)
    text = subprocess.run([f"{kernel_metadata.LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", code], capture_output=True,
                          text=True, check=True).stdout
    kernels, name, body = {}, None, []
    for line in text.split("\n"):
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            if name:
                kernels[name] = body
            name, body = m.group(1), []
        elif name and line.startswith("\t"):
            body.append(line)
    if name:
        kernels[name] = body
    mangled = list(kernels)
    names = subprocess.run(["c++filt"], input="\n".join(mangled), capture_output=True, text=True, check=True).stdout.split("\n")
    return {kernel_metadata.short_name(n): kernels[m] for m, n in zip(mangled, names)}


def count(body, trips):
    """(multiplies per lane, [loop descriptions]).  A loop = a branch whose target lies before it; nested / overlapping loops
    are reported and counted with the product of their trip counts."""
    address = re.compile(r"//\s*([0-9A-Fa-f]+):")
    rows = []
    for line in body:
        m = address.search(line)
        op = line.strip().split()[0] if line.strip() else ""
        rows.append((int(m.group(1), 16) if m else None, op, line))
    index_of = {a: i for i, (a, _, _) in enumerate(rows) if a is not None}
    loops = []
    for i, (a, op, line) in enumerate(rows):
        if not op.startswith("s_cbranch") and op != "s_branch":
            continue
        m = re.search(r"<[^>]*\+0x([0-9a-fA-F]+)>", line)
        if not m or a is None:
            continue
        base = rows[0][0]
        target = base + int(m.group(1), 16)
        if target <= a and target in index_of:
            loops.append((index_of[target], i))
    weight = [1] * len(rows)
    described = []
    for k, (begin, end) in enumerate(loops):
        trip = trips if isinstance(trips, int) else 1
        inside = sum(1 for _, op, _ in rows[begin:end + 1] if op in MULTIPLIES)
        described.append({"instructions": end - begin + 1, "multiplies": inside, "trips": trip})
        for j in range(begin, end + 1):
            weight[j] *= trip
    total = sum(w for (_, op, _), w in zip(rows, weight) if op in MULTIPLIES)
    return total, described


def main():
    out = {"degree": N, "moduli": L, "kernels": [], "multiply_instructions": list(MULTIPLIES)}
    with tempfile.TemporaryDirectory() as workdir:
        cache = {}
        for spec in PIPELINE:
            if spec["object"] not in cache:
                cache[spec["object"]] = disassemble(spec["object"], workdir)
            body = cache[spec["object"]].get(spec["kernel"])
            if body is None:
                raise SystemExit("no kernel %s in %s" % (spec["kernel"], spec["object"]))
            trips = L if spec["trips"] == "L" else 1
            per_lane, loops = count(body, trips)
            per_product = per_lane * spec["lanes_per_product"]
            out["kernels"].append({"kernel": spec["kernel"], "multiplies_per_lane": per_lane,
                                   "lanes_per_product": spec["lanes_per_product"], "multiplies_per_product": per_product,
                                   "loops": loops})
            print("%-44s %6d multiplies per lane x %6d lanes = %10d per product   loops %s" % (
                spec["kernel"], per_lane, spec["lanes_per_product"], per_product, loops))
    out["multiplies_per_product"] = sum(k["multiplies_per_product"] for k in out["kernels"])
    print("multiplies per ct x ct + relinearize: %d" % out["multiplies_per_product"])
    if "--json" in sys.argv:
        json.dump(out, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)


if __name__ == "__main__":
    main()
