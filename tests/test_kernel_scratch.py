"""No kernel of the production shapes keeps an ARRAY in scratch.  The tiled transforms sit at the 64-register cap and may
park a few registers (<= 64 B per lane is what the committed kernels do at most); hundreds of bytes mean the compiler gave up
promoting a register array -- a rolled loop indexing the rows at run time, which once cut relinearize on the reference's 60-bit
parameter sets from 1.5 M/s to 0.6 M/s without failing a single parity test.  Read from the built objects' kernel metadata
(bench_tools/kernel_metadata.py); no GPU involved."""
import glob
import os
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
        return [(kernel_metadata.short_name(k["name"]), k["scratch"]) for k in kernel_metadata.kernels(code)]


def test_production_shapes_keep_no_array_in_scratch():
    if not glob.glob(os.path.join(BUILD, "ntt_kernels.o")):
        pytest.skip("the library's objects are built by __graft_entry__.build()")
    offenders = []
    for name, scratch in _kernels("ntt_kernels.o"):
        # the tiled kernels of the reference's degrees (N = 4096: <12, 9, ...>, N = 8192: <13, 10, ...>) and the interleaved ones
        production = name.startswith(("ntt_forward_tiled<12, 9,", "ntt_inverse_tiled<12, 9,", "ntt_forward_tiled<13, 10,",
                                      "ntt_inverse_tiled<13, 10,", "ntt_forward_interleaved<1,", "ntt_inverse_interleaved<1,"))
        if production and scratch > 64:
            offenders.append((name, scratch))
    for obj in ("rns_kernels.o", "poly_kernels.o", "galois_kernels.o", "word32_kernels.o", "behz_kernels.o"):
        for name, scratch in _kernels(obj):
            if scratch > 64:
                offenders.append((name, scratch))
    assert not offenders, offenders
