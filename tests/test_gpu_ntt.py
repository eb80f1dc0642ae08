"""GPU parity tests of the NTT kernels against the CPU oracle and the reference's KATs (through the C ABI).

Mirrors Tests/HomomorphicEncryptionTests/NttTests.swift.  Bit-exact: integer work, no tolerance.
"""
import random

import numpy as np
import pytest

import heamd

pytestmark = pytest.mark.gpu


def _rand_slab(rng, batch, moduli, degree):
    rows = [rng.integers(0, q, size=(batch, degree), dtype=np.uint64) for q in moduli]
    return np.ascontiguousarray(np.stack(rows, axis=1))


def test_ntt_known_answers(kats):
    # NttTests.swift:72-191 (N = 2..32, one and two moduli) -- exercises the generic kernel
    for c in kats["ntt"]["cases"]:
        coeff = np.array(c["coeff"], dtype=np.uint64)
        evals = np.array(c["eval"], dtype=np.uint64)
        ctx = heamd.PolyContext(coeff.shape[1], c["moduli"])
        got = heamd.to_host(ctx.forward_ntt_(heamd.to_device(coeff)))
        assert np.array_equal(got, evals), c
        back = heamd.to_host(ctx.inverse_ntt_(heamd.to_device(evals)))
        assert np.array_equal(back, coeff), c


def test_ntt_delta_and_zero(kats):
    for c in kats["ntt"]["delta_cases"]:
        n = c["degree"]
        ctx = heamd.PolyContext(n, [c["modulus"]])
        zeros = np.zeros((1, n), dtype=np.uint64)
        one_hot = zeros.copy()
        one_hot[0, 0] = 1
        ones = np.ones((1, n), dtype=np.uint64)
        assert np.array_equal(heamd.to_host(ctx.forward_ntt_(heamd.to_device(zeros))), zeros)
        assert np.array_equal(heamd.to_host(ctx.forward_ntt_(heamd.to_device(one_hot))), ones)
        assert np.array_equal(heamd.to_host(ctx.inverse_ntt_(heamd.to_device(ones))), one_hot)


@pytest.mark.parametrize(
    "degree,bits,batch",
    [
        (2, [30], 3),
        (16, [55, 52, 62, 58], 5),      # TestUtils.testCoefficientModuli shape (TestUtilities.swift:295-319)
        (256, [60, 62], 4),             # NttTests.swift:193-206
        (1024, [27, 28, 28], 3),
        (2048, [61, 61, 33], 2),
        (4096, [55, 55], 6),            # BASELINE config 1 shape (tiled kernel: 512 lanes x 8 words, row pairs + an odd row)
        (8192, [55, 55, 55, 55], 5),    # BASELINE config 2 shape (tiled kernel: 1024 lanes x 8 words, limb-wise butterflies)
        (8192, [62, 61, 60, 33], 3),    # a 62-bit modulus forces the exact-quotient butterflies
        (16384, [55, 61, 45], 2),       # two interleaved sub-rows of the 8192-point kernel
        (32768, [55, 40], 1),           # four interleaved sub-rows of the 8192-point kernel
    ],
)
def test_ntt_matches_oracle(oracle, degree, bits, batch):
    moduli = oracle.generate_primes(bits, False, degree)
    ours = heamd.PolyContext(degree, moduli)
    ref = oracle.PolyContext(degree, moduli)
    rng = np.random.default_rng(degree * 31 + len(bits))
    slab = _rand_slab(rng, batch, moduli, degree)
    # edge values: 0 and q-1 everywhere in the first polynomial's first/last rows
    slab[0, 0, :] = 0
    slab[0, -1, :] = moduli[-1] - 1
    expected_eval = ref.forward_ntt(slab)
    got_eval = heamd.to_host(ours.forward_ntt_(heamd.to_device(slab)))
    assert np.array_equal(got_eval, expected_eval)
    expected_coeff = ref.inverse_ntt(slab)
    got_coeff = heamd.to_host(ours.inverse_ntt_(heamd.to_device(slab)))
    assert np.array_equal(got_coeff, expected_coeff)


@pytest.mark.parametrize("degree,bits", [(4096, [55, 55]), (8192, [55, 55, 55, 55]), (16384, [55, 55])])
@pytest.mark.parametrize("variant", [1, 2, 3, 10])
def test_ntt_kernel_variants_agree(oracle, degree, bits, variant):
    """Every named schedule computes the reference transform: exact-quotient butterflies (1), generic radix-2 kernel
    (2), 16 words per lane (3), [0, 8p) butterflies (10)."""
    moduli = oracle.generate_primes(bits, False, degree)
    ours = heamd.PolyContext(degree, moduli)
    ref = oracle.PolyContext(degree, moduli)
    rng = np.random.default_rng(degree + variant)
    slab = _rand_slab(rng, 3, moduli, degree)
    assert np.array_equal(heamd.to_host(ours.ntt_variant_(heamd.to_device(slab), False, variant)), ref.forward_ntt(slab))
    assert np.array_equal(heamd.to_host(ours.ntt_variant_(heamd.to_device(slab), True, variant)), ref.inverse_ntt(slab))


@pytest.mark.parametrize("degree,bits,batch", [(8192, [55, 55, 55], 2), (8192, [55, 55, 55], 3), (8192, [50] * 5, 31),
                                               (8192, [61, 61], 64), (8192, [62], 77), (4096, [55, 41, 48], 66),
                                               (4096, [55] * 4, 25), (16384, [55, 55, 55], 9), (8192, [55] * 4, 683)])
def test_ntt_row_pairs(oracle, degree, bits, batch):
    """With 8 words per lane a workgroup transforms the same residue row of two consecutive polynomials (one modulus,
    every twiddle fetched once for both; ntt_kernels.hip kRowsPerWorkgroup): even and odd batches (the odd polynomial
    takes the one-row kernel), every butterfly schedule, modulus periods 1..5; against the oracle."""
    moduli = oracle.generate_primes(bits, False, degree)
    ours = heamd.PolyContext(degree, moduli)
    ref = oracle.PolyContext(degree, moduli)
    rng = np.random.default_rng(batch)
    slab = _rand_slab(rng, batch, moduli, degree)
    slab[-1, -1, :] = moduli[-1] - 1
    slab[0, 0, :] = 0
    for inverse in (False, True):
        got = heamd.to_host(ours.ntt_variant_(heamd.to_device(slab), inverse, 0))
        sample = sorted(set([0, 1, batch // 2, batch - 2, batch - 1]) & set(range(batch)))
        expected = ref.inverse_ntt(slab[sample]) if inverse else ref.forward_ntt(slab[sample])
        assert np.array_equal(got[sample], expected)


def test_ntt_unknown_variant_is_rejected(oracle):
    """The library exports no schedule that returns HE_OK with anything but the transform."""
    moduli = oracle.generate_primes([55, 55], False, 8192)
    ours = heamd.PolyContext(8192, moduli)
    slab = heamd.to_device(np.zeros((1, 2, 8192), dtype=np.uint64))
    for variant in (4, 8, 9, 12, 16, 17, 48, 1040, -1):
        with pytest.raises(heamd.HeError):
            ours.ntt_variant_(slab, False, variant)


@pytest.mark.parametrize("degree", [4096, 8192, 16384])
@pytest.mark.parametrize("bits", [[55, 55, 54], [41, 41], [48, 55]])
def test_ntt_headroom_mode_extremes(oracle, degree, bits):
    """Moduli in [2^40, 2^55) take the fold-free limb-wise Shoup schedule (ntt_common.hpp kModeSplit): drive it with the
    inputs that grow fastest (all q-1, alternating 0 / q-1, one-hot q-1) next to random rows and compare with the oracle."""
    moduli = oracle.generate_primes(bits, False, degree)
    ours = heamd.PolyContext(degree, moduli)
    ref = oracle.PolyContext(degree, moduli)
    rng = np.random.default_rng(degree + len(bits))
    slab = _rand_slab(rng, 6, moduli, degree)
    top = np.array(moduli, dtype=np.uint64)[:, None] - np.uint64(1)
    slab[0] = top
    slab[1] = 0
    slab[1, :, ::2] = top
    slab[2] = 0
    slab[2, :, 1::2] = top
    slab[3] = 0
    slab[3, :, degree - 1] = top[:, 0]
    slab[4, :, : degree // 2] = top
    for inverse, expected in ((False, ref.forward_ntt(slab)), (True, ref.inverse_ntt(slab))):
        got = heamd.to_host(ours.ntt_variant_(heamd.to_device(slab), inverse, 0))
        assert np.array_equal(got, expected)
        pinned = heamd.to_host(ours.ntt_variant_(heamd.to_device(slab), inverse, 10))
        assert np.array_equal(pinned, expected)


@pytest.mark.parametrize("degree", [4096, 8192, 16384])
def test_ntt_split_limb_edges(oracle, degree):
    """The limb-wise Shoup butterflies multiply the two 32-bit halves of a word separately: rows whose words sit on
    the limb edges (low half all ones / zero, high half at its maximum) and moduli at both ends of [2^40, 2^55)."""
    moduli = oracle.generate_primes([55, 41], False, degree) + oracle.generate_primes([41, 55], True, degree)
    ours = heamd.PolyContext(degree, moduli)
    ref = oracle.PolyContext(degree, moduli)
    rng = np.random.default_rng(degree)
    slab = _rand_slab(rng, 5, moduli, degree)
    q = np.array(moduli, dtype=np.uint64)[:, None]
    low_ones = np.uint64(0xFFFFFFFF)
    slab[0] = (slab[0] | low_ones) % q
    slab[1] = (slab[1] & ~low_ones) % q
    slab[2] = np.minimum(q - np.uint64(1), (q & ~low_ones) | (slab[2] & low_ones))
    slab[3] = ((q - np.uint64(1)) & ~low_ones) + np.uint64(0)
    slab[3, :, ::3] = np.uint64(0xFFFFFFFF)
    slab[3, :, 1::3] = np.uint64(1) << np.uint64(32)
    assert (slab < q).all()
    assert np.array_equal(heamd.to_host(ours.forward_ntt_(heamd.to_device(slab))), ref.forward_ntt(slab))
    assert np.array_equal(heamd.to_host(ours.inverse_ntt_(heamd.to_device(slab))), ref.inverse_ntt(slab))


@pytest.mark.parametrize("bits,batch", [([55, 55, 55], 100), ([55, 50, 61], 90), ([62, 45], 140), ([47], 300)])
def test_ntt_many_rows_16384(oracle, bits, batch):
    """N = 16384 with more rows than the chip holds workgroups at a time -- every butterfly schedule, row counts that do
    not divide by the CU count.  Word for word against the oracle."""
    degree = 16384
    moduli = oracle.generate_primes(bits, False, degree)
    ours, ref = heamd.PolyContext(degree, moduli), oracle.PolyContext(degree, moduli)
    rng = np.random.default_rng(len(bits) * 1000 + batch)
    slab = _rand_slab(rng, batch, moduli, degree)
    slab[0, :, :] = 0
    slab[-1, -1, :] = moduli[-1] - 1
    assert batch * len(moduli) > 256
    forward = ref.forward_ntt(slab)
    assert np.array_equal(heamd.to_host(ours.forward_ntt_(heamd.to_device(slab))), forward)
    assert np.array_equal(heamd.to_host(ours.inverse_ntt_(heamd.to_device(forward))), slab)


@pytest.mark.parametrize("degree", [16384, 32768])
@pytest.mark.parametrize("bits,batch", [([55, 55, 55], 3), ([55, 50, 61], 5), ([62, 45], 2), ([47], 9), ([61, 61], 4),
                                        ([50, 48, 44, 41], 1)])
def test_ntt_interleaved_rows(oracle, degree, bits, batch):
    """N = 16384 / 32768 as 2 / 4 interleaved sub-rows of 8192 words (ntt_forward_interleaved / ntt_inverse_interleaved):
    thirteen stages on the sub-rows with shared twiddles, the rest across them -- every butterfly schedule (limb-wise,
    [0, 8p), exact), limb-edge and extreme words; the oracle decides, word for word."""
    moduli = oracle.generate_primes(bits, False, degree)
    ours, ref = heamd.PolyContext(degree, moduli), oracle.PolyContext(degree, moduli)
    rng = np.random.default_rng(degree + len(bits) * 1000 + batch)
    slab = _rand_slab(rng, batch + 3, moduli, degree)
    q = np.array(moduli, dtype=np.uint64)[:, None]
    low_ones = np.uint64(0xFFFFFFFF)
    slab[0] = (slab[0] | low_ones) % q
    slab[1] = q - np.uint64(1)
    slab[2] = 0
    slab[2, :, 1] = 1  # x: the transform is the table of odd powers of psi
    forward = ref.forward_ntt(slab)
    assert np.array_equal(heamd.to_host(ours.forward_ntt_(heamd.to_device(slab))), forward)
    assert np.array_equal(heamd.to_host(ours.inverse_ntt_(heamd.to_device(forward))), slab)
    assert np.array_equal(heamd.to_host(ours.inverse_ntt_(heamd.to_device(slab))), ref.inverse_ntt(slab))


@pytest.mark.parametrize("degree", [4096, 8192])
@pytest.mark.parametrize("bits,small", [([60, 60, 60], False), ([56, 57, 58, 59], False), ([60], False), ([61, 61, 61, 61, 61], True),
                                        ([61], True), ([60, 55, 60], False), ([61, 60], True), ([58, 61], False)])
def test_ntt_fold_butterfly_moduli(oracle, degree, bits, small):
    """Moduli next to a power of two above 2^55 take the fold butterflies (ntt_common.hpp kModeFoldMinus / kModeFoldPlus):
    the largest 56..60-bit primes (what the reference's 60-bit parameter sets hold) fold by 2^(b+2) = 4d, the smallest 61-bit
    primes (the BEHZ auxiliary base, RnsTool.swift:28-66) by 2^62 = -4e; sets that mix the forms, or hold a smaller or a
    far-off prime, keep the [0, 8p) butterflies.  Limb-edge and extreme words; row counts that leave an odd row; the
    oracle decides."""
    moduli = oracle.generate_primes(bits, small, degree)
    ours, ref = heamd.PolyContext(degree, moduli), oracle.PolyContext(degree, moduli)
    batch = 5
    rng = np.random.default_rng(degree + sum(bits) + int(small))
    slab = _rand_slab(rng, batch, moduli, degree)
    q = np.array(moduli, dtype=np.uint64)[:, None]
    low_ones = np.uint64(0xFFFFFFFF)
    slab[0] = (slab[0] | low_ones) % q
    slab[1] = np.minimum(q - np.uint64(1), (q & ~low_ones) | (slab[1] & low_ones))
    slab[2] = q - np.uint64(1)
    slab[3] = 0
    slab[3, :, 1] = 1
    forward = ref.forward_ntt(slab)
    assert np.array_equal(heamd.to_host(ours.forward_ntt_(heamd.to_device(slab))), forward)
    assert np.array_equal(heamd.to_host(ours.inverse_ntt_(heamd.to_device(forward))), slab)
    assert np.array_equal(heamd.to_host(ours.inverse_ntt_(heamd.to_device(slab))), ref.inverse_ntt(slab))


@pytest.mark.parametrize("degree", [4096, 16384])
@pytest.mark.parametrize("bits", [[55, 54], [50, 48, 44], [42, 55], [41, 41]])
def test_ntt_shifted_factor_moduli(oracle, degree, bits):
    """Moduli just below a power of two (generatePrimes(preferringSmall: false)) take their gathered twiddles' quotient
    factors by a shift where that measures faster (forward N = 4096; DeviceModulus::split_shift);
    42-bit and smaller primes of an NTT-friendly form sit too far below their power of two and keep the table, as does
    a context that mixes the two.  Limb-edge words included; the oracle decides."""
    moduli = oracle.generate_primes(bits, False, degree)
    ours, ref = heamd.PolyContext(degree, moduli), oracle.PolyContext(degree, moduli)
    batch = 300 // len(moduli) + 1
    rng = np.random.default_rng(degree + sum(bits))
    slab = _rand_slab(rng, batch, moduli, degree)
    q = np.array(moduli, dtype=np.uint64)[:, None]
    low_ones = np.uint64(0xFFFFFFFF)
    slab[0] = (slab[0] | low_ones) % q
    slab[1] = np.minimum(q - np.uint64(1), (q & ~low_ones) | (slab[1] & low_ones))
    slab[2] = q - np.uint64(1)
    forward = ref.forward_ntt(slab)
    assert np.array_equal(heamd.to_host(ours.forward_ntt_(heamd.to_device(slab))), forward)
    assert np.array_equal(heamd.to_host(ours.inverse_ntt_(heamd.to_device(forward))), slab)


@pytest.mark.parametrize("moduli_count", [1, 2, 3, 5, 6, 7])
def test_ntt_row_map_periods(oracle, moduli_count):
    """A workgroup finds its row's modulus with one scalar multiply-high (ntt_kernels.hip locate): every period, with
    enough rows that a wrong quotient would pick a wrong modulus somewhere."""
    degree = 4096
    moduli = oracle.generate_primes([50] * moduli_count, False, degree)
    ours = heamd.PolyContext(degree, moduli)
    ref = oracle.PolyContext(degree, moduli)
    rng = np.random.default_rng(moduli_count)
    slab = _rand_slab(rng, 211, moduli, degree)
    assert np.array_equal(heamd.to_host(ours.forward_ntt_(heamd.to_device(slab))), ref.forward_ntt(slab))
    assert np.array_equal(heamd.to_host(ours.inverse_ntt_(heamd.to_device(slab))), ref.inverse_ntt(slab))


def test_ntt_multiplication_matches_schoolbook(oracle):
    # NttTests.swift:208-250
    degree = 128
    p = oracle.generate_primes([30], False, degree, word_bits=32)[0]
    ctx = heamd.PolyContext(degree, [p])
    rng = random.Random(2)
    x = [rng.randrange(p) for _ in range(degree)]
    y = [rng.randrange(p) for _ in range(degree)]
    xe = ctx.forward_ntt_(heamd.to_device(np.array([x], dtype=np.uint64)))
    ye = ctx.forward_ntt_(heamd.to_device(np.array([y], dtype=np.uint64)))
    prod = heamd.to_host(ctx.inverse_ntt_(ctx.mul_(xe, ye)))[0]
    expected = [0] * degree
    for i in range(degree):
        acc = 0
        for j in range(i + 1):
            acc += x[j] * y[i - j]
        for j in range(i + 1, degree):
            acc -= x[j] * y[degree + i - j]
        expected[i] = acc % p
    assert [int(v) for v in prod] == expected


def test_ntt_rows_seam(oracle):
    """PolyContext.forwardNtt(dataPtr:modulus:) (PolyRq+Ntt.swift:329-347): rows of one modulus."""
    degree = 4096
    moduli = oracle.generate_primes([55, 50, 45], False, degree)
    ours = heamd.PolyContext(degree, moduli)
    rng = np.random.default_rng(5)
    for index, q in enumerate(moduli):
        rows = rng.integers(0, q, size=(3, degree), dtype=np.uint64)
        single = oracle.PolyContext(degree, [q])
        got = heamd.to_host(ours.forward_ntt_rows_(q, heamd.to_device(rows)))
        assert np.array_equal(got, single.forward_ntt(rows.reshape(3, 1, degree)).reshape(3, degree)), index
        back = heamd.to_host(ours.inverse_ntt_rows_(q, heamd.to_device(got)))
        assert np.array_equal(back, rows)
    with pytest.raises(heamd.HeError) as err:
        ours.forward_ntt_rows_(97, heamd.to_device(np.zeros((1, degree), dtype=np.uint64)))
    assert err.value.name == "invalidPolyContext"


def test_ntt_host_pointer_seam(oracle):
    degree = 1024
    moduli = oracle.generate_primes([40, 41], False, degree)
    ours = heamd.PolyContext(degree, moduli)
    ref = oracle.PolyContext(degree, moduli)
    rng = np.random.default_rng(6)
    slab = _rand_slab(rng, 3, moduli, degree)
    assert np.array_equal(ours.forward_ntt_host(slab), ref.forward_ntt(slab))
    assert np.array_equal(ours.inverse_ntt_host(slab), ref.inverse_ntt(slab))


def test_ntt_rejects_non_ntt_moduli():
    ctx = heamd.PolyContext(4, [2, 3, 5])
    with pytest.raises(heamd.HeError) as err:
        ctx.forward_ntt_(heamd.to_device(np.zeros((1, 3, 4), dtype=np.uint64)))
    assert err.value.name == "invalidNttModulus"


def test_ntt_full_size_properties(oracle):
    """BASELINE config 2 at full size (N=8192, L=4, 4096 polys = 1 GiB): size-independent properties.

    * inverse(forward(x)) == x for the whole slab;
    * linearity: forward(x + y) == forward(x) + forward(y) (mod q) on the whole slab;
    * EVERY polynomial agrees word for word with the oracle's forward transform (NttTests.swift:193-206 compares whole
      polynomials), in slices that bound host memory -- the multi-threaded oracle does the slab in well under a second on
      the GPU boxes' hosts; on a host with fewer than 8 threads, a sample of 8 polynomials spread over the slab.
    """
    import torch
    from conftest import exhaustive_parity, host_threads

    degree, batch = 8192, 4096
    moduli = oracle.generate_primes([55] * 4, False, degree)
    ours = heamd.PolyContext(degree, moduli)
    ref = oracle.PolyContext(degree, moduli)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(0x5EED)
    bound = torch.tensor(moduli, dtype=torch.int64, device="cuda").view(1, len(moduli), 1)
    x = torch.randint(0, 1 << 62, (batch, len(moduli), degree), dtype=torch.int64, device="cuda", generator=gen) % bound
    original = x.clone()
    ours.forward_ntt_(x)
    if exhaustive_parity():
        compared = 0
        for first in range(0, batch, 512):  # 128 MiB per slice
            expected = heamd.to_host(original[first:first + 512])
            ref.forward_ntt_inplace(expected, threads=host_threads())
            assert np.array_equal(heamd.to_host(x[first:first + 512]), expected), first
            compared += expected.shape[0]
        assert compared == batch
        print(f"forward NTT: {compared} of {batch} polynomials compared with the oracle word for word")
    else:
        sample = [0, 1, 17, 1000, 2047, 2048, 4000, 4095]
        expected = ref.forward_ntt(heamd.to_host(original[sample]))
        assert np.array_equal(heamd.to_host(x[sample]), expected)
        print(f"forward NTT: {len(sample)} of {batch} polynomials compared with the oracle (small host)")
    assert int((x >= bound).sum()) == 0 and int((x < 0).sum()) == 0  # canonical outputs
    # linearity on the full slab
    y = torch.randint(0, 1 << 62, x.shape, dtype=torch.int64, device="cuda", generator=gen) % bound
    s = y.clone()
    ours.add_(s, original)            # s = x + y  (coefficient domain)
    ours.forward_ntt_(s)
    ours.forward_ntt_(y)
    ours.add_(y, x)                   # NTT(y) + NTT(x)
    assert torch.equal(s, y)
    del s, y
    ours.inverse_ntt_(x)
    assert torch.equal(x, original)


def _primes_below_power_of_two(oracle, bits, degree, eligible, count=1):
    """NTT primes p = 2^bits - d (p = 1 mod 2N) with d as LARGE as the shift-folded products allow (d < 2^(bits-32):
    csrc/poly_context.cpp split_shift) when `eligible`, or the first ones just past that bound otherwise."""
    step, top = 2 * degree, 1 << (bits - 32)
    found = []
    k = (top + 1) // step if eligible else (top + 1) // step + 1
    while len(found) < count and k >= 1 and k * step - 1 < (1 << (bits - 1)):
        d = k * step - 1
        p = (1 << bits) - d
        if (d < top) == eligible and oracle.is_prime(p):
            found.append(p)
        k += -1 if eligible else 1
    return found


@pytest.mark.parametrize("degree", [4096, 8192, 16384, 32768])
def test_shift_folded_products_at_the_edge_of_their_moduli(oracle, degree):
    """kModeFoldLazy (csrc/ntt_common.hpp): the transforms of moduli 2^b - d fold their products by a shift, which wants
    d < 2^(b-32).  The primes generatePrimes returns sit at the small end of d; these sit at the LARGE end (the product's
    bound 2^(b+2) + 2^33 d is nearly reached), next to primes just past the bound (limb-wise products) in the same context --
    the launch then falls back as a whole -- and alone.  Extreme words in every residue.  N = 16384 / 32768: the interleaved
    sub-row kernels on the same products (cross stages with three twiddles in flight)."""
    edge = [p for bits in (55, 54, 52, 50) for p in _primes_below_power_of_two(oracle, bits, degree, True)]
    past = [p for bits in (55, 52) for p in _primes_below_power_of_two(oracle, bits, degree, False)]
    # (NTT-friendly primes are 2N apart: the larger degrees have fewer of them inside the bound)
    assert len(edge) >= (3 if degree <= 8192 else 1) and len(past) == 2, (edge, past)
    rng = np.random.default_rng(degree)
    for moduli in (edge, edge[:1], edge[:2] + past[:1], past):
        ours, ref = heamd.PolyContext(degree, moduli), oracle.PolyContext(degree, moduli)
        slab = _rand_slab(rng, 5, moduli, degree)
        slab[0] = 0
        for i, m in enumerate(moduli):
            slab[1, i, :] = m - 1
            slab[2, i, ::2] = m - 1
            slab[2, i, 1::2] = 0
        assert np.array_equal(heamd.to_host(ours.forward_ntt_(heamd.to_device(slab))), ref.forward_ntt(slab)), moduli
        assert np.array_equal(heamd.to_host(ours.inverse_ntt_(heamd.to_device(slab))), ref.inverse_ntt(slab)), moduli
