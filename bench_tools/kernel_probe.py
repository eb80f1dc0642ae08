"""Compile a FEW instantiations of the NTT kernel templates on their own (seconds instead of the minutes the whole
translation unit takes) and print their register / scratch figures -- the inner loop of register-pressure work.

  python bench_tools/kernel_probe.py [--asm OUT.s] [--mix] 'ntt_inverse_tiled<13, 10, kModeSplit, kInverseFromKeyMacFinish, 2>' ...

--mix also prints each kernel's static instruction mix (opcode counts of its body).

The kernel definitions of csrc/ntt_kernels.hip (everything above the launchers) are copied to a scratch file followed by
explicit instantiations of the named kernels; nothing is linked into the library.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "swift-homomorphic-encryption_amd", "csrc")
SIGNATURES = {
    "ntt_forward_tiled": "(uint64_t*, const DeviceContext, const RowMap, const SpreadSource)",
    "ntt_inverse_tiled": "(uint64_t*, const DeviceContext, const RowMap, const InverseSource)",
    "ntt_forward_interleaved": "(uint64_t*, const DeviceContext, const RowMap, const SpreadSource)",
    "ntt_inverse_interleaved": "(uint64_t*, const DeviceContext, const RowMap, const InverseSource)",
}


def main():
    args = sys.argv[1:]
    asm_out, mix = None, False
    while args and args[0].startswith("--"):
        if args[0] == "--asm":
            asm_out, args = args[1], args[2:]
        elif args[0] == "--mix":
            mix, args = True, args[1:]
        else:
            raise SystemExit("unknown option " + args[0])
    source = open(os.path.join(CSRC, "ntt_kernels.hip")).read()
    head = source[:source.index("// Row pairs: where the register file allows it")]
    body = head + "\n".join(f"template __global__ void {k}{SIGNATURES[k.split('<')[0].strip()]};" for k in args)
    body += "\n}  // namespace\n}  // namespace heamd\n"
    with tempfile.TemporaryDirectory() as work:
        path = os.path.join(work, "probe.hip")
        open(path, "w").write(body)
        out = os.path.join(work, "probe.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        "-I" + CSRC, "-o", out, path], check=True)
        text = open(out).read()
        if asm_out:
            open(asm_out, "w").write(text)
    for block in re.findall(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", text, re.S):
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        get = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, block).group(1))  # noqa: E731
        demangled = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        short = re.sub(r"\(anonymous namespace\)::|heamd::|^void ", "", demangled).split("(")[0]
        print(f"{short:70s} vgpr {get('vgpr_count'):3d}  sgpr {get('sgpr_count'):3d}  scratch {get('private_segment_fixed_size'):4d} B"
              f"  spilled {get('vgpr_spill_count'):2d}")
        if mix:
            import collections

            body = re.search(r"^%s:[^\n]*\n(.*?)^\.Lfunc_end\d+:" % re.escape(name), text, re.S | re.M).group(1)
            ops = collections.Counter(m.group(1) for m in re.finditer(r"^\s+([vs]_\w+|ds_\w+|buffer_\w+|scratch_\w+|global_\w+)", body, re.M))
            print("   " + "  ".join(f"{o}:{c}" for o, c in ops.most_common(40)))


if __name__ == "__main__":
    main()
