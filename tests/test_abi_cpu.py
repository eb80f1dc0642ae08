"""CPU-only checks of the drop-in boundary: the C-ABI library loads and exports every symbol include/he_amd.h
declares, the host-side setup math (validation order, primes, roots, twiddle tables, Bsk base) matches the oracle, and
compute entry points fail loudly without a device context.  No kernel is launched here."""
import os
import re

import numpy as np
import pytest

import heamd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, "include", "he_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(he_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = heamd.load_library()
    declared = _declared_functions()
    assert len(declared) > 40
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/he_amd.h but not exported by libhe_amd.so"
    bound = {name for name, _, _ in heamd.binding.SIGNATURES}
    assert set(declared) == bound, f"binding and header disagree: {set(declared) ^ bound}"
    assert "gfx950" in heamd.version()


def test_generate_primes_matches_reference_kats(kats):
    for c in kats["generate_primes"]["cases"]:
        if max(c["bits"]) > 62:
            continue
        assert heamd.generate_primes(c["bits"], c["preferring_small"], c["ntt_degree"]) == c["expected"], c
    for c in kats["generate_primes"]["error_cases"]:
        with pytest.raises(heamd.HeError) as err:
            heamd.generate_primes(c["bits"], c["preferring_small"], c["ntt_degree"])
        assert err.value.name == c["error"]


def test_poly_context_validation_order(kats):
    # PolyContextTests.swift:22-38 -- errors are raised before any device work, so this runs without a GPU
    for c in kats["poly_context_errors"]["cases"]:
        with pytest.raises(heamd.HeError) as err:
            heamd.PolyContext(c["degree"], c["moduli"])
        assert err.value.name == c["error"], c
        with pytest.raises(heamd.HeError) as err:
            heamd.PolyContext(c["degree"], c["moduli"], host_only=True)
        assert err.value.name == c["error"], c


def test_host_only_context_queries(kats):
    for c in kats["q_remainder"]["cases"]:
        ctx = heamd.PolyContext(c["degree"], c["moduli"], host_only=True)
        assert ctx.q_remainder(c["dividing_by"]) == c["expected"]
    c = kats["max_lazy_product_accumulation_count"]["cases"][1]
    ctx = heamd.PolyContext(c["degree"], c["moduli"], host_only=True)
    assert ctx.max_lazy_product_accumulation_count() == c["expected"]
    assert ctx.degree == c["degree"] and ctx.moduli == c["moduli"]


@pytest.mark.parametrize("degree,bits", [(8, [30]), (256, [60, 62]), (4096, [55, 55]), (8192, [55, 55, 55, 55])])
def test_ntt_tables_match_oracle(oracle, degree, bits):
    moduli = oracle.generate_primes(bits, False, degree)
    ours = heamd.PolyContext(degree, moduli, host_only=True)
    ref = oracle.PolyContext(degree, moduli)
    for i in range(len(moduli)):
        a, b = ours.ntt_tables(i), ref.ntt_tables(i)
        for key in ("root_powers", "root_factors", "inv_root_powers", "inv_root_factors"):
            assert np.array_equal(a[key], b[key]), (i, key)
        assert a["inverse_degree"] == b["inverse_degree"]
        assert a["inverse_degree_root"] == b["inverse_degree_root"]


def test_non_ntt_modulus_has_no_tables():
    ctx = heamd.PolyContext(4, [2, 3, 5], host_only=True)
    with pytest.raises(heamd.HeError) as err:
        ctx.ntt_tables(1)
    assert err.value.name == "invalidNttModulus"


def test_compute_without_device_context_fails_loudly():
    import ctypes

    ctx = heamd.PolyContext(8, heamd.generate_primes([30], False, 8), host_only=True)
    buf = np.zeros(8, dtype=np.uint64)
    lib = heamd.load_library()
    status = lib.he_ntt_forward(ctx.h, buf.ctypes.data_as(heamd.binding.U64P), 1)
    assert heamd.binding.STATUS_NAMES[status] == "deviceError"
    status = lib.he_ntt_forward_device(ctx.h, ctypes.c_void_p(0x1000), 1, None)
    assert heamd.binding.STATUS_NAMES[status] == "deviceError"
    assert b"host-only" in lib.he_last_error_message()
    # NTT on a context with a non-NTT modulus is rejected first (validateNttModuli), as in the reference
    bad = heamd.PolyContext(4, [2, 3, 5], host_only=True)
    status = lib.he_ntt_forward(bad.h, buf.ctypes.data_as(heamd.binding.U64P), 0)
    assert heamd.binding.STATUS_NAMES[status] == "invalidNttModulus"


def test_device_context_creation_needs_a_gpu():
    try:
        import torch

        has_gpu = torch.cuda.is_available()
    except Exception:  # pragma: no cover
        has_gpu = False
    if has_gpu:
        pytest.skip("a GPU is present")
    with pytest.raises(heamd.HeError) as err:
        heamd.PolyContext(8, heamd.generate_primes([30], False, 8))
    assert err.value.name == "deviceError"


def test_bfv_host_only_context_matches_oracle(oracle):
    degree = 8192
    q = oracle.generate_primes([55] * 5, False, degree)
    ours = heamd.BfvContext(degree, 557057, q, host_only=True)
    ref = oracle.BfvContext(degree, 557057, q)
    assert ours.L == ref.L == 4
    assert ours.bsk_moduli() == ref.rns_tool().bsk
    for k in (1, 2, 3, 4):
        assert ours.ciphertext_context(k).moduli == ref.ciphertext_context(k).moduli
        assert ours.key_switching_context(k).moduli == ref.key_switching_context(k).moduli
        assert ours.qbsk_context(k).moduli == ref.qbsk_context(k).moduli
    # the qBsk context's NTT tables (61-bit Bsk primes) match the oracle's
    a, b = ours.qbsk_context().ntt_tables(8), ref.qbsk_context().ntt_tables(8)
    assert np.array_equal(a["root_powers"], b["root_powers"]) and np.array_equal(a["inv_root_factors"], b["inv_root_factors"])


def test_bfv_context_errors(oracle):
    degree = 64
    t = oracle.generate_primes([17], True, degree)[0]
    q = oracle.generate_primes([40, 40], False, degree)
    for bad in ((degree, t, q + [q[0] + 2]), (degree + 1, t, q), (degree, q[0], q), (degree, t, [])):
        with pytest.raises(heamd.HeError) as err:
            heamd.BfvContext(*bad, host_only=True)
        assert err.value.name == "invalidEncryptionParameters", bad
        with pytest.raises(oracle.OracleError) as ref_err:
            oracle.BfvContext(*bad)
        assert ref_err.value.name == "invalidEncryptionParameters"
    # compute on a host-only context fails loudly
    ctx = heamd.BfvContext(degree, t, q, host_only=True)
    lib = heamd.load_library()
    status = lib.he_bfv_mul_device(ctx.h, 1, None, None, None, 1, None, 0, None)
    assert heamd.binding.STATUS_NAMES[status] == "deviceError"


def test_swift_package_carries_the_same_header():
    """swift/Sources/CHeAmd/include/he_amd.h is the header the SwiftPM C target exposes; it must be the library's."""
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with open(os.path.join(root, "include", "he_amd.h")) as ours, \
            open(os.path.join(root, "swift", "Sources", "CHeAmd", "include", "he_amd.h")) as copy:
        assert ours.read() == copy.read(), "run: cp include/he_amd.h swift/Sources/CHeAmd/include/he_amd.h"


def test_galois_element_helpers():
    """GaloisElement.rotatingColumns / swappingRows (PolyRq/Galois.swift:174-212) with the reference's own vectors
    (GaloisTests.swift:103-109: degree 8, elements 3, 9, 11 are right rotations by 3, 2, 1) and its round-trip property
    (:77-85: a rotation by s composed with one by N/2 - s is the identity, i.e. the elements multiply to 1 mod 2N)."""
    import heamd

    assert heamd.galois_element_swapping_rows(8) == 15 and heamd.galois_element_swapping_rows(8192) == 16383
    assert [heamd.galois_element_rotating_columns(s, 8) for s in (3, 2, 1)] == [3, 9, 11]
    for degree in (16, 32, 1024, 8192):
        seen = set()
        for step in range(1, degree // 2):
            forward = heamd.galois_element_rotating_columns(step, degree)
            backward = heamd.galois_element_rotating_columns(degree // 2 - step, degree)
            assert forward % 2 == 1 and forward * backward % (2 * degree) == 1
            assert heamd.galois_element_rotating_columns(-step, degree) == backward  # left by s = right by N/2 - s
            seen.add(forward)
        assert len(seen) == degree // 2 - 1
    for bad_step, degree in ((0, 8), (4, 8), (-4, 8), (1, 2)):
        with pytest.raises(heamd.HeError) as err:
            heamd.galois_element_rotating_columns(bad_step, degree)
        assert err.value.name == "invalidArgument"
    with pytest.raises(heamd.HeError) as err:
        heamd.galois_element_rotating_columns(1, 12)
    assert err.value.name == "invalidDegree"


def test_library_exports_only_the_c_abi():
    """`nm -D libhe_amd.so`: every defined text symbol is an entry point of include/he_amd.h (no C++ launcher, no
    measurement hook, nothing that is not declared in the header)."""
    import re
    import subprocess

    import heamd

    out = subprocess.run(["nm", "-D", "--defined-only", heamd.library_path()], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    assert exported and all(name.startswith("he_") for name in exported), sorted(n for n in exported if not n.startswith("he_"))
    import os

    with open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "he_amd.h")) as f:
        declared = set(re.findall(r"\b(he_[a-z0-9_]+)\s*\(", f.read()))
    assert exported <= declared, sorted(exported - declared)
