"""Register / scratch / LDS figures of the kernels in the BUILT library's gfx950 code objects (no GPU needed).

  python bench_tools/kernel_metadata.py [object ...] [--filter SUBSTRING ...] [--spills-only]

Default objects: every csrc/build/*.o.  For each object the .hip_fatbin section is unbundled
(clang-offload-bundler, target hipv4-amdgcn-amd-amdhsa--gfx950) and the kernel descriptors' metadata notes are read with
llvm-readelf: VGPRs, SGPRs, scratch bytes per lane (private_segment_fixed_size), spilled VGPRs, static LDS.  Names are
demangled and shortened as bench_tools/pmc_traffic.py prints them.
"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "swift-homomorphic-encryption_amd", "csrc", "build")
LLVM = "/opt/rocm/lib/llvm/bin"
sys.path.insert(0, os.path.join(ROOT, "bench_tools"))
from pmc_traffic import short_name  # noqa: E402


def code_object(path, workdir):
    fatbin = os.path.join(workdir, os.path.basename(path) + ".fatbin")
    out = os.path.join(workdir, os.path.basename(path) + ".co")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", path, fatbin], check=True)
    if os.path.getsize(fatbin) == 0:
        return None
    subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", f"--input={fatbin}", "--type=o",
                    "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={out}"], check=True)
    return out


def kernels(code):
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", code], capture_output=True, text=True, check=True).stdout
    for block in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
        def field(key, cast=int):
            m = re.search(r"\.%s:\s+(\S+)" % key, block)
            return cast(m.group(1)) if m else 0
        yield {"name": field("name", str), "vgpr": field("vgpr_count"), "sgpr": field("sgpr_count"),
               "scratch": field("private_segment_fixed_size"), "spills": field("vgpr_spill_count"),
               "lds": field("group_segment_fixed_size"), "max_flat_workgroup_size": field("max_flat_workgroup_size")}


def main():
    args = sys.argv[1:]
    filters, objects, spills_only = [], [], False
    while args:
        a = args.pop(0)
        if a == "--filter":
            filters.append(args.pop(0))
        elif a == "--spills-only":
            spills_only = True
        else:
            objects.append(a)
    objects = objects or sorted(glob.glob(os.path.join(BUILD, "*.o")))
    with tempfile.TemporaryDirectory() as workdir:
        for path in objects:
            code = code_object(path, workdir)
            if code is None:
                continue
            rows = list(kernels(code))
            names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True,
                                   text=True).stdout.split("\n")
            shown = 0
            with_scratch = sum(1 for r in rows if r["scratch"])
            print(f"== {os.path.basename(path)}: {len(rows)} kernels, {with_scratch} with scratch")
            for r, name in zip(rows, names):
                name = short_name(name)
                if filters and not all(f in name for f in filters):
                    continue
                if spills_only and not r["scratch"]:
                    continue
                shown += 1
                print(f"   {name[:96]:96s} vgpr {r['vgpr']:3d}  sgpr {r['sgpr']:3d}  scratch {r['scratch']:4d} B  "
                      f"spilled {r['spills']:2d}  lds {r['lds']:6d}  lanes {r['max_flat_workgroup_size']}")


if __name__ == "__main__":
    main()
