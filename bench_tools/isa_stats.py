"""Static instruction mix of the compiled gfx950 kernels (no GPU needed).

Usage: python bench_tools/isa_stats.py <file.hip> [name-substring ...]
Compiles the file to gfx950 assembly with the build's flags and prints, per kernel whose mangled name contains
every given substring: VGPR/SGPR/scratch use and the count of each opcode, with an estimate of VALU issue slots
(rates measured by bench_tools/microbench.hip: 64-bit and multiply ops take 2 slots, v_mul_hi_u32 3-4).
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "swift-homomorphic-encryption_amd", "csrc")

SLOTS = {
    "v_mad_u64_u32": 2, "v_mul_lo_u32": 2, "v_mul_hi_u32": 4, "v_lshl_add_u64": 2, "v_lshlrev_b64": 2,
    "v_lshrrev_b64": 2, "v_cmp_gt_u64_e32": 2, "v_cmp_lt_u64_e32": 2, "v_cmp_ge_u64_e32": 2, "v_cmp_le_u64_e32": 2,
    "v_fma_f64": 2, "v_mul_f64": 2, "v_add_f64": 2,
}


def assemble(source):
    out = tempfile.NamedTemporaryFile(suffix=".s", delete=False).name
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                    "-I" + CSRC, *os.environ.get("HEAMD_ISA_FLAGS", "").split(), "-o", out, source], check=True, stderr=subprocess.DEVNULL)
    text = open(out).read()
    os.unlink(out)
    return text


def main():
    source = sys.argv[1]
    needles = sys.argv[2:]
    text = assemble(source)
    meta = {}
    for block in re.findall(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", text, re.S):
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        meta[name] = {k: int(re.search(r"\.%s:\s+(\d+)" % k, block).group(1))
                      for k in ("vgpr_count", "sgpr_count", "private_segment_fixed_size", "group_segment_fixed_size")}
    for match in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", text, re.S | re.M):
        name, body = match.group(1), match.group(2)
        if not all(n in name for n in needles):
            continue
        ops = collections.Counter()
        for line in body.split("\n"):
            m = re.match(r"\s+([vs]_\w+|ds_\w+|global_\w+|buffer_\w+|scratch_\w+|flat_\w+)", line)
            if m:
                ops[m.group(1)] += 1
        valu = sum(c for o, c in ops.items() if o.startswith("v_"))
        slots = sum(c * SLOTS.get(o, 1) for o, c in ops.items() if o.startswith("v_"))
        print(f"== {name}\n   {meta.get(name)}\n   instructions {sum(ops.values())}  VALU {valu}  est. VALU slots {slots}"
              f"  s_nop {ops.get('s_nop', 0)}  s_waitcnt {ops.get('s_waitcnt', 0)}")
        print("   " + "  ".join(f"{o}:{c}" for o, c in ops.most_common(70)))


if __name__ == "__main__":
    main()
