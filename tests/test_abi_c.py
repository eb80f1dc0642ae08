"""The boundary is a C ABI: the header must be plain C, and a C program must be able to drive the library with no
Python or PyTorch in the process (that is what the Swift package would link)."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
INCLUDE = os.path.join(ROOT, "include")
LIBDIR = os.path.join(ROOT, "swift-homomorphic-encryption_amd", "lib")


@pytest.mark.parametrize("compiler,flags", [("gcc", ["-std=c99", "-x", "c"]), ("g++", ["-std=c++11", "-x", "c++"])])
def test_header_is_plain_c_and_cxx(compiler, flags, tmp_path):
    if shutil.which(compiler) is None:
        pytest.skip(f"{compiler} not installed")
    unit = tmp_path / "unit.c"
    unit.write_text('#include "he_amd.h"\nint main(void) { return HE_OK; }\n')
    subprocess.run([compiler, *flags, "-Wall", "-Wextra", "-Werror", "-pedantic", "-fsyntax-only", "-I", INCLUDE,
                    str(unit)], check=True)


@pytest.mark.gpu
def test_c_program_drives_the_library(tmp_path):
    if shutil.which("gcc") is None:
        pytest.skip("gcc not installed")
    binary = tmp_path / "abi_roundtrip"
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "c", "abi_roundtrip.c"),
                    "-I", INCLUDE, "-L", LIBDIR, "-lhe_amd", f"-Wl,-rpath,{LIBDIR}", "-o", str(binary)], check=True)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = LIBDIR + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    result = subprocess.run([str(binary)], capture_output=True, text=True, env=env, timeout=300)
    assert result.returncode == 0, result.stdout + result.stderr
    assert "abi round trip ok" in result.stdout
