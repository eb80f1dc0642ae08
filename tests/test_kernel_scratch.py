"""No kernel of the production shapes keeps an ARRAY in scratch.  The tiled transforms sit at the 64-register cap and may
park a few registers (<= 64 B per lane is what the committed kernels do at most); hundreds of bytes mean the compiler gave up
promoting a register array -- a rolled loop indexing the rows at run time, which once cut relinearize on the reference's 60-bit
parameter sets from 1.5 M/s to 0.6 M/s without failing a single parity test.  Read from the built objects' kernel metadata
(bench_tools/kernel_metadata.py); no GPU involved."""
import glob
import os
import subprocess
import sys
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BUILD = os.path.join(ROOT, "swift-homomorphic-encryption_amd", "csrc", "build")
sys.path.insert(0, os.path.join(ROOT, "bench_tools"))


def _kernels(obj):
    import kernel_metadata

    with tempfile.TemporaryDirectory() as workdir:
        code = kernel_metadata.code_object(os.path.join(BUILD, obj), workdir)
        assert code is not None, obj
        rows = list(kernel_metadata.kernels(code))
        # the notes hold mangled names: demangle before shortening (short_name only strips namespaces and arguments)
        names = subprocess.run(["c++filt"], input="\n".join(r["name"] for r in rows), capture_output=True, text=True,
                               check=True).stdout.split("\n")
        return [(kernel_metadata.short_name(name), r["scratch"]) for r, name in zip(rows, names)]


def test_production_shapes_keep_no_array_in_scratch():
    if not glob.glob(os.path.join(BUILD, "ntt_kernels.o")):
        pytest.skip("the library's objects are built by __graft_entry__.build()")
    offenders, seen = [], 0
    for name, scratch in _kernels("ntt_kernels.o"):
        # the tiled kernels of the reference's degrees (N = 4096: <12, 9, ...>, N = 8192: <13, 10, ...>) and the interleaved ones
        production = name.startswith(("ntt_forward_tiled<12, 9,", "ntt_inverse_tiled<12, 9,", "ntt_forward_tiled<13, 10,",
                                      "ntt_inverse_tiled<13, 10,", "ntt_forward_interleaved<1,", "ntt_inverse_interleaved<1,"))
        seen += production
        # (round 6: EVERY transform kernel, not only the production shapes -- the 32-words-per-lane test kernels that kept
        # 272 B are gone)
        if scratch > 64:
            offenders.append((name, scratch))
    assert seen >= 100, seen  # (the names really are the demangled ones the prefixes above are written for)
    for obj in ("rns_kernels.o", "poly_kernels.o", "galois_kernels.o", "word32_kernels.o", "behz_kernels.o"):
        for name, scratch in _kernels(obj):
            if scratch > 64:
                offenders.append((name, scratch))
    assert not offenders, offenders


def test_shift_folded_two_row_kernels_keep_nothing_in_scratch():
    """The kernels every predefined parameter set of the reference runs on (shift-folded products, mode 4; two rows per
    workgroup; N = 4096 / 8192) -- the headline pair, the key switch's spread / key-MAC / finish transforms, the lifted and
    fused-load forms -- fit their 64 registers without a byte of scratch (round 5 left 8-20 B in five of them)."""
    import re

    if not glob.glob(os.path.join(BUILD, "ntt_kernels.o")):
        pytest.skip("the library's objects are built by __graft_entry__.build()")
    shape = re.compile(r"ntt_(forward|inverse)_tiled<(12, 9|13, 10), 4, \d, 2>")
    seen, offenders = 0, []
    for name, scratch in _kernels("ntt_kernels.o"):
        if shape.match(name):
            seen += 1
            if scratch != 0:
                offenders.append((name, scratch))
    assert seen >= 14, seen
    assert not offenders, offenders


def test_shift_folded_interleaved_kernels_of_the_slab_keep_nothing_in_scratch():
    """N = 16384 on the shift-folded products (two sub-rows per workgroup): the plain forward transform and the three inverse
    forms of the slab.  The inverse kept 20 B with three cross-stage twiddles in flight; with two it keeps none and measures
    4-5 % faster (profiles/r06v_interleaved_fold_inverse_ab.txt)."""
    if not glob.glob(os.path.join(BUILD, "ntt_kernels.o")):
        pytest.skip("the library's objects are built by __graft_entry__.build()")
    wanted = {"ntt_forward_interleaved<1, 4, 0>", "ntt_inverse_interleaved<1, 4, true, 0>",
              "ntt_inverse_interleaved<1, 4, false, 0>", "ntt_inverse_interleaved<1, 4, true, 1>"}
    found = {name: scratch for name, scratch in _kernels("ntt_kernels.o") if name in wanted}
    assert set(found) == wanted, sorted(found)
    assert not any(found.values()), found
