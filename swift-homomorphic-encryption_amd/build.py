"""Builds libhe_amd.so (HIP kernels + C ABI) for gfx950 with hipcc, in-tree.

    python swift-homomorphic-encryption_amd/build.py [--force]

hipcc cross-compiles without a GPU.  Objects go to csrc/build/, the library to lib/libhe_amd.so (git-ignored, but
shipped to the GPU box with the repo snapshot).
"""
import concurrent.futures
import hashlib
import os
import re
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ_DIR = os.path.join(CSRC, "build")
LIB_DIR = os.path.join(HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libhe_amd.so")
EXPORTS = os.path.join(CSRC, "exports.map")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function"]


def _hipcc():
    return shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".cpp")))


def _headers_mtime():
    newest = 0.0
    for root in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(root):
            if f.endswith((".hpp", ".h")):
                newest = max(newest, os.path.getmtime(os.path.join(root, f)))
    return newest


def _dependencies(depfile):
    """The prerequisites hipcc recorded for an object (-MD): the source and every header it really includes."""
    try:
        text = open(depfile).read()
    except OSError:
        return None
    text = text.replace("\\\n", " ")
    if ":" not in text:
        return None
    # make's escapes: a space inside a path is written "\\ " -- split on unescaped whitespace only
    names = [n.replace("\\ ", " ") for n in re.findall(r"(?:\\ |\S)+", text.split(":", 1)[1])]
    return [d if os.path.isabs(d) else os.path.join(CSRC, d) for d in names if d]


_COMPILER_ID = None


def _command_stamp(cmd):
    """What an object was built with: the full compile command and the compiler's own version string.  A change of FLAGS,
    ARCH or hipcc (a ROCm upgrade: -MMD does not list system headers) rebuilds the object even though no source moved."""
    global _COMPILER_ID
    if _COMPILER_ID is None:
        try:
            _COMPILER_ID = subprocess.run([_hipcc(), "--version"], capture_output=True, text=True).stdout.strip()
        except OSError:
            _COMPILER_ID = "unknown"
    return hashlib.sha256(("\0".join(cmd) + "\0" + _COMPILER_ID).encode()).hexdigest()


def _compile(src, force, header_time):
    obj = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".o")
    dep = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".d")
    stamp_path = os.path.join(OBJ_DIR, os.path.splitext(src)[0] + ".cmd")
    path = os.path.join(CSRC, src)
    cmd = [_hipcc(), *FLAGS, "-MMD", "-MF", dep, "-c", path, "-o", obj]
    if src.endswith(".cpp"):
        cmd[1:1] = ["-x", "hip"]
    stamp = _command_stamp(cmd)
    try:
        same_command = open(stamp_path).read() == stamp
    except OSError:
        same_command = False
    if not force and same_command and os.path.exists(obj):
        built = os.path.getmtime(obj)
        deps = _dependencies(dep)
        if deps is not None:
            # rebuild only when something this object was compiled from is newer (a header one unit includes does not
            # cost the four minutes of ntt_kernels.hip)
            if all(os.path.exists(d) and os.path.getmtime(d) <= built for d in deps):
                return obj, False
        elif built >= max(os.path.getmtime(path), header_time):
            return obj, False
    result = subprocess.run(cmd, capture_output=True, text=True)
    if result.returncode != 0:
        raise RuntimeError(f"hipcc failed on {src}:\n{result.stdout}\n{result.stderr}")
    with open(stamp_path, "w") as f:
        f.write(stamp)
    return obj, True


def build(force=False, verbose=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(LIB_DIR, exist_ok=True)
    header_time = _headers_mtime()
    sources = _sources()
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(8, len(sources))) as pool:
        results = list(pool.map(lambda s: _compile(s, force, header_time), sources))
    objects = [obj for obj, _ in results]
    rebuilt = any(changed for _, changed in results)
    if rebuilt or not os.path.exists(LIB_PATH):
        # only the C ABI (he_*) is exported: the C++ launchers behind it stay internal
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", f"-Wl,--version-script={EXPORTS}", "-o", LIB_PATH,
               *objects]
        result = subprocess.run(cmd, capture_output=True, text=True)
        if result.returncode != 0:
            raise RuntimeError(f"link failed:\n{result.stdout}\n{result.stderr}")
    if verbose:
        print(f"{'rebuilt' if rebuilt else 'up to date'}: {LIB_PATH}")
    return LIB_PATH


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
