"""Builds tests/device_probe/libarith_probe.so (test infrastructure: the device-side products of csrc/device_math.hpp behind a
C entry point) with hipcc for gfx950; called by __graft_entry__.build().  Cross-compiles without a GPU."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "swift-homomorphic-encryption_amd", "csrc")
SOURCE = os.path.join(HERE, "arith_probe.hip")
LIBRARY = os.path.join(HERE, "libarith_probe.so")


def build(force=False):
    newest = max(os.path.getmtime(p) for p in (SOURCE, os.path.join(CSRC, "device_math.hpp"), os.path.join(CSRC, "host_math.hpp")))
    if not force and os.path.exists(LIBRARY) and os.path.getmtime(LIBRARY) >= newest:
        return LIBRARY
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-I", CSRC, "-o", LIBRARY, SOURCE])
    return LIBRARY


if __name__ == "__main__":
    print(build(force=True))
