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


@pytest.mark.gpu
def test_c_program_drives_the_scheme_level(tmp_path, oracle):
    """B3 from C: ct x ct, relinearize, modSwitchDownToSingle and a masked ct x pt inner product, each compared word for
    word with the oracle's output inside the C program (tests/c/abi_scheme.c)."""
    import numpy as np

    if shutil.which("gcc") is None:
        pytest.skip("gcc not installed")
    degree, batch, d0 = 256, 3, 5
    t = oracle.generate_primes([17], True, degree)[0]
    q = oracle.generate_primes([50, 50, 51], False, degree)
    ref = oracle.BfvContext(degree, t, q)
    L = ref.L
    moduli = ref.ciphertext_context().moduli
    rng = np.random.default_rng(333)

    def uniform(prefix, row_moduli):
        rows = [rng.integers(0, m, size=tuple(prefix) + (degree,), dtype=np.uint64) for m in row_moduli]
        return np.ascontiguousarray(np.stack(rows, axis=len(prefix)))

    lhs, rhs = uniform((batch, 2), moduli), uniform((batch, 2), moduli)
    key = uniform((L, 2), ref.key_switching_context().moduli)
    product = ref.mul(lhs, rhs)
    relinearized = ref.relinearize(product, key)
    single = relinearized
    for level in range(L, 1, -1):
        single = ref.mod_switch_down(single, 2, level)
    cts, pts = uniform((d0, 2), moduli), uniform((d0,), moduli)
    mask = np.array([1, 0, 1, 1, 0], dtype=np.uint64)
    inner = ref.inner_product_plain(cts, pts, present=mask.astype(np.uint8))
    head = np.array([degree, t, len(q), *q, batch, d0], dtype=np.uint64)
    fixture = tmp_path / "scheme.bin"
    with open(fixture, "wb") as out:
        for part in (head, lhs, rhs, key, product, relinearized, single, cts, pts, mask, inner):
            out.write(np.ascontiguousarray(part, dtype=np.uint64).tobytes())
    binary = tmp_path / "abi_scheme"
    subprocess.run(["gcc", "-std=c99", "-O1", "-Wall", "-Werror", os.path.join(ROOT, "tests", "c", "abi_scheme.c"),
                    "-I", INCLUDE, "-L", LIBDIR, "-lhe_amd", f"-Wl,-rpath,{LIBDIR}", "-o", str(binary)], check=True)
    env = dict(os.environ)
    env["LD_LIBRARY_PATH"] = LIBDIR + ":/opt/rocm/lib:" + env.get("LD_LIBRARY_PATH", "")
    result = subprocess.run([str(binary), str(fixture)], capture_output=True, text=True, env=env, timeout=300)
    assert result.returncode == 0, result.stdout + result.stderr
    assert "abi scheme ok" in result.stdout
