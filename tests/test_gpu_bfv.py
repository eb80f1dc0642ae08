"""GPU parity tests of the BEHZ kernels and the Bfv<UInt64> scheme operations against the CPU oracle, plus the
reference's semantic (decrypt) checks (Sources/_TestUtilities/HeApiTestUtils.swift:494-720,1223-1285).  Bit-exact."""
import os
import random

import numpy as np
import pytest

import heamd
from bfv_helpers import BfvClient, negacyclic_multiply

pytestmark = pytest.mark.gpu


def _uniform(rng, shape_prefix, moduli, degree):
    """uint64 array [*shape_prefix][L][N] with row i uniform in [0, moduli[i])."""
    rows = [rng.integers(0, q, size=tuple(shape_prefix) + (degree,), dtype=np.uint64) for q in moduli]
    return np.ascontiguousarray(np.stack(rows, axis=len(shape_prefix)))


@pytest.fixture(scope="module")
def small(oracle):
    degree = 64
    t = oracle.generate_primes([17], True, degree)[0]
    q = oracle.generate_primes([40, 40, 40, 41], False, degree)
    return heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q), BfvClient(oracle, oracle.BfvContext(degree, t, q), seed=30)


@pytest.fixture(scope="module")
def config3(oracle):
    """BASELINE config 3 parameters: N=8192, t=557057, 4 ciphertext moduli + 1 key-switching modulus (55-bit)."""
    degree = 8192
    q = oracle.generate_primes([55] * 5, False, degree)
    return heamd.BfvContext(degree, 557057, q), oracle.BfvContext(degree, 557057, q)


def test_context_matches_oracle(oracle, config3):
    ours, ref = config3
    assert ours.L == ref.L == 4
    assert ours.bsk_moduli() == ref.rns_tool().bsk
    assert ours.ciphertext_context().moduli == ref.ciphertext_context().moduli
    assert ours.key_switching_context(2).moduli == ref.key_switching_context(2).moduli
    assert ours.qbsk_context(3).moduli == ref.qbsk_context(3).moduli


@pytest.mark.parametrize("level", [None, 2, 1])
def test_lift_and_floor_match_oracle(oracle, small, level):
    ours, ref, _ = small
    L = ours.L if level is None else level
    tool = ref.rns_tool(L)
    moduli = ref.ciphertext_context(L).moduli
    rng = np.random.default_rng(40 + L)
    x = _uniform(rng, (5,), moduli, ours.degree)
    x[0, :, :3] = 0
    x[0, :, 3] = [m - 1 for m in moduli]
    lifted = heamd.to_host(ours.lift_q_to_qbsk(heamd.to_device(x), L))
    expected = np.stack([tool.lift_q_to_qbsk(p) for p in x])
    assert np.array_equal(lifted, expected)
    qbsk_moduli = ref.qbsk_context(L).moduli
    y = _uniform(rng, (5,), qbsk_moduli, ours.degree)
    floored = heamd.to_host(ours.floor_qbsk_to_q(heamd.to_device(y), L))
    assert np.array_equal(floored, np.stack([tool.floor_qbsk_to_q(p) for p in y]))


def test_lift_matches_oracle_config3(oracle, config3):
    ours, ref = config3
    rng = np.random.default_rng(41)
    x = _uniform(rng, (3,), ref.ciphertext_context().moduli, ours.degree)
    got = heamd.to_host(ours.lift_q_to_qbsk(heamd.to_device(x)))
    assert np.array_equal(got, np.stack([ref.rns_tool().lift_q_to_qbsk(p) for p in x]))
    y = _uniform(rng, (3,), ref.qbsk_context().moduli, ours.degree)
    got = heamd.to_host(ours.floor_qbsk_to_q(heamd.to_device(y)))
    assert np.array_equal(got, np.stack([ref.rns_tool().floor_qbsk_to_q(p) for p in y]))


@pytest.mark.parametrize("level", [None, 2])
def test_mul_matches_oracle(oracle, small, level):
    """Bfv.mulAssign(ct, ct) word for word, top level and (literal restatement) a lower level."""
    ours, ref, _ = small
    L = ours.L if level is None else level
    moduli = ref.ciphertext_context(L).moduli
    rng = np.random.default_rng(50 + L)
    lhs, rhs = _uniform(rng, (4, 2), moduli, ours.degree), _uniform(rng, (4, 2), moduli, ours.degree)
    got = heamd.to_host(ours.mul(heamd.to_device(lhs), heamd.to_device(rhs), L))
    assert got.shape == (4, 3, L, ours.degree)
    assert np.array_equal(got, ref.mul(lhs, rhs, L))


def test_mul_relinearize_decrypts_to_product(oracle, small):
    # HeApiTestUtils.swift:494-556 with genuine encryptions and a genuine relinearization key
    ours, ref, client = small
    rng = random.Random(51)
    m1 = [rng.randrange(ours.t) for _ in range(ours.degree)]
    m2 = [rng.randrange(ours.t) for _ in range(ours.degree)]
    ct1, ct2 = client.encrypt(m1), client.encrypt(m2)
    product = ours.mul(heamd.to_device(ct1[None]), heamd.to_device(ct2[None]))
    expected = negacyclic_multiply(m1, m2, ours.t)
    assert client.decrypt(heamd.to_host(product)[0]) == expected
    key = client.relinearization_key()
    relin = ours.relinearize(product, heamd.to_device(key))
    assert np.array_equal(heamd.to_host(relin), ref.relinearize(heamd.to_host(product), key))
    assert client.decrypt(heamd.to_host(relin)[0]) == expected


@pytest.mark.parametrize("level", [None, 2, 1])
def test_relinearize_matches_oracle(oracle, small, level):
    ours, ref, _ = small
    L = ours.L if level is None else level
    moduli = ref.ciphertext_context(L).moduli
    rng = np.random.default_rng(60 + L)
    ct3 = _uniform(rng, (3, 3), moduli, ours.degree)
    key = _uniform(rng, (ours.L, 2), ref.key_switching_context().moduli, ours.degree)
    got = heamd.to_host(ours.relinearize(heamd.to_device(ct3), heamd.to_device(key), L))
    assert np.array_equal(got, ref.relinearize(ct3, key, L))
    with pytest.raises(heamd.HeError) as err:
        ours.relinearize(heamd.to_device(ct3), None, L)
    assert err.value.name == "missingRelinearizationKey"


@pytest.mark.parametrize("degree,bits,batch,level", [
    (8192, [55] * 5, 33, None),      # BASELINE config 3's moduli: Q band limb-wise (signed inverse), Bsk band fold (2^60 + e)
    (8192, [55] * 5, 61, 3),         # below the top level (seven [Q, Bsk] rows)
    (8192, [55] * 5, 130, 1),        # a single ciphertext modulus (three rows)
    (8192, [29, 60, 60], 60, None),  # the reference's n_8192_logq_29_60_60: 29 | 60 (fold, 2^60 - d) | Bsk
    (4096, [27, 28, 28], 300, None), # n_4096_logq_27_28_28: every ciphertext modulus below 2^40 -- the [0, 8p) butterflies
    (4096, [60, 60, 60], 90, None),  # all-60-bit moduli at N = 4096
    (4096, [50, 61, 55], 70, None),  # 50 (limb-wise) | 61 bits not next to a power of two ([0, 8p)) | Bsk
    (4096, [62, 62, 50], 64, None),  # 62-bit moduli: the exact butterflies for the whole record
])
def test_mul_row_fused_matches_oracle(oracle, monkeypatch, degree, bits, batch, level):
    """ct x ct on batches wide enough for behz_kernels.hip (HEAMD_BEHZ_FUSED_ABOVE=256 puts them through the row-fused kernels:
    in production those start at 1152 (item, row) workgroups, where they overtake the unfused launches) (one workgroup per (item, [Q, Bsk] row): four forward transforms,
    the tensor product, three scaled inverse transforms, the Eval rows never in HBM) -- every butterfly class the row bands
    take (limb-wise, fold of either form, [0, 8p), exact), levels below the top, odd batches; EVERY product word for word
    against the oracle's multiplyWithoutScaling + dropExtendedBase (Bfv+Multiply.swift:18-85), and the same words as the
    unfused pipeline computes for a batch too small for the fused kernel."""
    from conftest import host_threads

    monkeypatch.setenv("HEAMD_BEHZ_FUSED_ABOVE", "256")
    t = oracle.generate_primes([17], True, degree)[0]
    q = oracle.generate_primes(bits, False, degree)
    ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
    L = ours.L if level is None else level
    moduli = ref.ciphertext_context(L).moduli
    rng = np.random.default_rng(degree + batch)
    lhs, rhs = _uniform(rng, (batch, 2), moduli, degree), _uniform(rng, (batch, 2), moduli, degree)
    lhs[0, :, :, :2] = 0  # the extremes of every residue
    for i, m in enumerate(moduli):
        lhs[1, :, i, :] = m - 1
        rhs[1, :, i, :] = m - 1
    got = heamd.to_host(ours.mul(heamd.to_device(lhs), heamd.to_device(rhs), L))
    assert np.array_equal(got, ref.mul(lhs, rhs, L, threads=host_threads()))
    few = heamd.to_host(ours.mul(heamd.to_device(lhs[:3].copy()), heamd.to_device(rhs[:3].copy()), L))  # the unfused pipeline
    assert np.array_equal(few, got[:3])


def test_relinearize_refuses_overlapping_output(oracle, small):
    """he_bfv_relinearize_device: out overlapping ct3 is HE_ERR_INVALID_ARGUMENT whatever the batch size (the key switch's
    last kernel -- which one depends on the batch -- reads (c0, c1) of one item while it stores another's result); the one
    word-for-word in-place form, a single ciphertext onto its own (c0, c1), is allowed and equals the oracle."""
    ours, ref, _ = small
    L, n = ours.L, ours.degree
    moduli = ref.ciphertext_context(L).moduli
    rng = np.random.default_rng(77)
    ct3 = _uniform(rng, (3, 3), moduli, n)
    key = _uniform(rng, (ours.L, 2), ref.key_switching_context().moduli, n)
    lib = heamd.load_library()
    dev_ct3, dev_key = heamd.to_device(ct3), heamd.to_device(key)
    words = L * n
    for shift_words in (0, 2 * words, 3 * 3 * words - 1):  # the same start, inside, the last word
        out_ptr = dev_ct3.data_ptr() + 8 * shift_words
        assert lib.he_bfv_relinearize_device(ours.h, L, dev_ct3.data_ptr(), dev_key.data_ptr(), out_ptr, 3, None, 0, None) == 16
    assert np.array_equal(heamd.to_host(dev_ct3), ct3)  # nothing was launched
    single = heamd.to_device(ct3[:1].copy())
    assert lib.he_bfv_relinearize_device(ours.h, L, single.data_ptr(), dev_key.data_ptr(), single.data_ptr(), 1, None, 0, None) == 0
    assert np.array_equal(heamd.to_host(single)[0, :2], ref.relinearize(ct3[:1], key, L)[0])


def test_mul_and_relinearize_config3(oracle, config3):
    """BASELINE config 3 shape (N=8192, L=4): batch of 16, two items checked word for word against the oracle."""
    ours, ref = config3
    rng = np.random.default_rng(61)
    moduli = ref.ciphertext_context().moduli
    lhs, rhs = _uniform(rng, (16, 2), moduli, ours.degree), _uniform(rng, (16, 2), moduli, ours.degree)
    key = _uniform(rng, (ours.L, 2), ref.key_switching_context().moduli, ours.degree)
    product = ours.mul(heamd.to_device(lhs), heamd.to_device(rhs))
    relin = ours.relinearize(product, heamd.to_device(key))
    # every item word for word (the oracle spreads the 16 products over the host's threads)
    threads = min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1))
    expected_product = ref.mul(lhs, rhs, threads=threads)
    assert np.array_equal(heamd.to_host(product), expected_product)
    assert np.array_equal(heamd.to_host(relin), ref.relinearize(expected_product, key, threads=threads))
    # caller-provided workspace gives the same words
    import torch

    ws = torch.empty(ours.mul_workspace_bytes(16) // 8, dtype=torch.int64, device="cuda")
    again = ours.mul(heamd.to_device(lhs), heamd.to_device(rhs), workspace=ws)
    assert torch.equal(again, product)


@pytest.mark.parametrize("level", [3, 2, 1])
def test_mul_and_relinearize_below_the_top_level_config3(oracle, config3, level):
    """ct x ct + relinearize on the real ring (N=8192) BELOW the top level, where no reference test pins the words
    (SURVEY.md 4.3): the tiled fused loads (lifted-forward source rows, tensor + inverse, spread, key MAC) with L' < L
    rows per polynomial, the shared-m_sk constants of a lower-level tool -- every word against the oracle."""
    ours, ref = config3
    moduli = ref.ciphertext_context(level).moduli
    rng = np.random.default_rng(160 + level)
    lhs, rhs = _uniform(rng, (5, 2), moduli, ours.degree), _uniform(rng, (5, 2), moduli, ours.degree)
    key = _uniform(rng, (ours.L, 2), ref.key_switching_context().moduli, ours.degree)
    product = ours.mul(heamd.to_device(lhs), heamd.to_device(rhs), level)
    expected_product = ref.mul(lhs, rhs, level, threads=5)
    assert np.array_equal(heamd.to_host(product), expected_product)
    relin = ours.relinearize(product, heamd.to_device(key), level)
    assert np.array_equal(heamd.to_host(relin), ref.relinearize(expected_product, key, level, threads=5))


@pytest.mark.parametrize("count", [9, 12, 16])
def test_more_than_eight_ciphertext_moduli(oracle, count):
    """Context.init puts no bound on the number of coefficient moduli (Context.swift:94-143); the BEHZ kernels are
    specialised for 1..16 ciphertext moduli (16 x 55 bits is the N = 32768 security cap).  Lift, floor, ct x ct,
    relinearize and modSwitchDownToSingle word for word against the oracle with 9, 12 and 16 of them."""
    degree = 64
    t = oracle.generate_primes([17], True, degree)[0]
    q = oracle.generate_primes([36 + (i % 5) for i in range(count + 1)], False, degree)
    ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
    assert ours.L == ref.L == count
    moduli = ref.ciphertext_context().moduli
    rng = np.random.default_rng(900 + count)
    x = _uniform(rng, (3,), moduli, degree)
    assert np.array_equal(heamd.to_host(ours.lift_q_to_qbsk(heamd.to_device(x))),
                          np.stack([ref.rns_tool().lift_q_to_qbsk(p) for p in x]))
    y = _uniform(rng, (3,), ref.qbsk_context().moduli, degree)
    assert np.array_equal(heamd.to_host(ours.floor_qbsk_to_q(heamd.to_device(y))),
                          np.stack([ref.rns_tool().floor_qbsk_to_q(p) for p in y]))
    lhs, rhs = _uniform(rng, (2, 2), moduli, degree), _uniform(rng, (2, 2), moduli, degree)
    key = _uniform(rng, (count, 2), ref.key_switching_context().moduli, degree)
    product = ours.mul(heamd.to_device(lhs), heamd.to_device(rhs))
    expected = ref.mul(lhs, rhs)
    assert np.array_equal(heamd.to_host(product), expected)
    relin = heamd.to_host(ours.relinearize(product, heamd.to_device(key)))
    assert np.array_equal(relin, ref.relinearize(expected, key))
    single = relin
    for level in range(count, 1, -1):
        single = ref.mod_switch_down(single, 2, level)
    assert np.array_equal(heamd.to_host(ours.mod_switch_down_to_single(heamd.to_device(relin), 2)), single)


def test_mod_switch_down_matches_oracle(oracle, small):
    ours, ref, client = small
    rng = np.random.default_rng(70)
    ct = _uniform(rng, (3, 2), ref.ciphertext_context().moduli, ours.degree)
    got = heamd.to_host(ours.mod_switch_down(heamd.to_device(ct), 2))
    assert np.array_equal(got, ref.mod_switch_down(ct, 2))
    message = [int(v) for v in rng.integers(0, ours.t, size=ours.degree)]
    lower = heamd.to_host(ours.mod_switch_down(heamd.to_device(client.encrypt(message)[None]), 2))[0]
    assert client.decrypt(lower, moduli_count=ours.L - 1) == message
    with pytest.raises(heamd.HeError) as err:
        ours.mod_switch_down(heamd.to_device(ct[:, :, :1].copy()), 2, moduli_count=1)
    assert err.value.name == "invalidPolyContext"


def test_mul_plain_and_inner_product_plain(oracle, small):
    ours, ref, client = small
    poly_ctx = ref.ciphertext_context()
    rng = np.random.default_rng(80)
    count, columns = 6, 5
    cts = _uniform(rng, (count, 2), poly_ctx.moduli, ours.degree)
    pts = _uniform(rng, (columns, count), poly_ctx.moduli, ours.degree)
    present = rng.integers(0, 2, size=(columns, count), dtype=np.uint8)
    present[0, :] = 1
    present[1, :] = 0  # a column whose plaintexts are all nil: the reference returns the zero accumulator
    got = heamd.to_host(ours.inner_product_plain(heamd.to_device(cts), heamd.to_device(pts), present, 2, columns))
    for col in range(columns):
        assert np.array_equal(got[col], ref.inner_product_plain(cts, pts[col], present[col])), col
    no_mask = heamd.to_host(ours.inner_product_plain(heamd.to_device(cts), heamd.to_device(pts), None, 2, columns))
    assert np.array_equal(no_mask[2], ref.inner_product_plain(cts, pts[2], None))
    # Bfv.mulAssign(ct, pt)
    ct = heamd.to_device(cts[:3])
    ours.mul_plain_(ct, heamd.to_device(pts[0, :3]), 2)
    assert np.array_equal(heamd.to_host(ct), ref.mul_plain(cts[:3], pts[0, :3], 2))


@pytest.mark.parametrize("t_bits", [17, 41, 60])  # t^2 below 2^64, above it, and t next to the 62-bit moduli
def test_add_and_sub_plain_match_oracle(oracle, t_bits):
    """Bfv.addAssignCoeff / subAssignCoeff(ciphertext, plaintext) (Bfv.swift:110-117, Bfv+Encrypt.swift:75-140): words
    against the oracle at every level, two- and three-polynomial ciphertexts, extreme messages; decrypts to the sum."""
    degree = 256
    t = oracle.generate_primes([t_bits], True, degree)[0]
    q = oracle.generate_primes([62, 61, 62, 62], False, degree)
    ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
    client = BfvClient(oracle, ref, seed=t_bits)
    rng = np.random.default_rng(400 + t_bits)
    for level in (None, 2, 1):
        moduli = ref.ciphertext_context(level).moduli
        for polys in (2, 3):
            ct = _uniform(rng, (5, polys), moduli, degree)
            messages = rng.integers(0, t, size=(5, degree), dtype=np.uint64)
            messages[0, :] = t - 1
            messages[1, :] = 0
            messages[2, ::2] = 1
            for subtract in (False, True):
                got = heamd.to_host(ours.add_plain_(heamd.to_device(ct), heamd.to_device(messages), polys, subtract,
                                                    moduli_count=level))
                assert np.array_equal(got, ref.plaintext_translate(ct, messages, polys, subtract, moduli_count=level)), (
                    level, polys, subtract)
    if t_bits < 60:  # (a 60-bit t leaves no room for noise under these moduli's product at level 1)
        m1 = [int(v) for v in rng.integers(0, t, size=degree)]
        m2 = [int(v) for v in rng.integers(0, t, size=degree)]
        fresh = heamd.to_device(client.encrypt(m1)[None])
        ours.add_plain_(fresh, heamd.to_device(np.array([m2], dtype=np.uint64)), 2)
        assert client.decrypt(heamd.to_host(fresh)[0]) == [(a + b) % t for a, b in zip(m1, m2)]
        ours.add_plain_(fresh, heamd.to_device(np.array([m1], dtype=np.uint64)), 2, subtract=True)
        assert client.decrypt(heamd.to_host(fresh)[0]) == m2
    with pytest.raises(heamd.HeError):
        ours.add_plain_(heamd.to_device(_uniform(rng, (1, 2), q[:1], degree)), heamd.to_device(np.zeros((1, degree), dtype=np.uint64)),
                        2, moduli_count=ours.L + 1)


def test_inner_product_ct_ct(oracle, small):
    ours, ref, client = small
    rng = random.Random(90)
    count = 3
    m1 = [[rng.randrange(ours.t) for _ in range(ours.degree)] for _ in range(count)]
    m2 = [[rng.randrange(ours.t) for _ in range(ours.degree)] for _ in range(count)]
    lhs = np.stack([client.encrypt(m) for m in m1])
    rhs = np.stack([client.encrypt(m) for m in m2])
    got = heamd.to_host(ours.inner_product(heamd.to_device(lhs), heamd.to_device(rhs)))
    assert np.array_equal(got, ref.inner_product(lhs, rhs))
    expected = [0] * ours.degree
    for a, b in zip(m1, m2):
        expected = [(x + y) % ours.t for x, y in zip(expected, negacyclic_multiply(a, b, ours.t))]
    assert client.decrypt(got) == expected


def test_inner_product_plain_pir_shape(oracle, config3):
    """BASELINE config 5's inner loop at reduced d0/d1 (N=8192, L=4): 32 ciphertexts x 8 columns."""
    ours, ref = config3
    rng = np.random.default_rng(100)
    moduli = ref.ciphertext_context().moduli
    count, columns = 32, 8
    cts = _uniform(rng, (count, 2), moduli, ours.degree)
    pts = _uniform(rng, (columns, count), moduli, ours.degree)
    got = heamd.to_host(ours.inner_product_plain(heamd.to_device(cts), heamd.to_device(pts), None, 2, columns))
    for col in (0, 7):
        assert np.array_equal(got[col], ref.inner_product_plain(cts, pts[col], None))


@pytest.mark.parametrize("queries", [2, 3, 4])
def test_inner_product_plain_queries_side_by_side(oracle, small, config3, queries):
    """poly_count = 2 x queries: the ciphertext vectors of several queries laid side by side ([count][query][2][L][N])
    share every plaintext word; query q's slice of the output is its own Bfv.innerProduct(ciphertexts:plaintexts:)
    (Bfv.swift:476-505) word for word -- the oracle's on the small ring (nil plaintexts included) and the single-query
    kernel's on BASELINE config 5's ring."""
    ours, ref, _ = small
    moduli = ref.ciphertext_context().moduli
    rng = np.random.default_rng(200 + queries)
    count, columns = 7, 5
    cts = _uniform(rng, (count, queries, 2), moduli, ours.degree)
    pts = _uniform(rng, (columns, count), moduli, ours.degree)
    present = rng.integers(0, 2, size=(columns, count), dtype=np.uint8)
    present[0, :] = 1
    got = heamd.to_host(ours.inner_product_plain(heamd.to_device(cts), heamd.to_device(pts), present, 2 * queries, columns))
    got = got.reshape(columns, queries, 2, ours.L, ours.degree)
    for q in range(queries):
        own = np.ascontiguousarray(cts[:, q])
        for col in range(columns):
            assert np.array_equal(got[col, q], ref.inner_product_plain(own, pts[col], present[col])), (q, col)
    big, big_ref = config3
    moduli = big_ref.ciphertext_context().moduli
    count, columns = 12, 6
    cts = _uniform(rng, (count, queries, 2), moduli, big.degree)
    pts = heamd.to_device(_uniform(rng, (columns, count), moduli, big.degree))
    got = heamd.to_host(big.inner_product_plain(heamd.to_device(cts), pts, None, 2 * queries, columns))
    got = got.reshape(columns, queries, 2, big.L, big.degree)
    for q in range(queries):
        single = heamd.to_host(big.inner_product_plain(heamd.to_device(np.ascontiguousarray(cts[:, q])), pts, None, 2, columns))
        assert np.array_equal(got[:, q], single.reshape(columns, 2, big.L, big.degree)), q


@pytest.mark.parametrize("bits", [[50, 55], [55, 40, 62, 48], [45] * 6, [60, 61, 62, 55, 50, 45, 40, 58, 59]])
def test_mod_switch_down_to_single(oracle, bits):
    """Ciphertext.modSwitchDownToSingle (Bfv.swift:163-171) in one kernel: word for word the chain of modSwitchDown steps
    (the oracle's and the step entry point's), from every level of contexts with 1 to 8 ciphertext moduli; words at 0
    and q - 1 included."""
    degree = 64
    t = oracle.generate_primes([17], True, degree)[0]
    q = oracle.generate_primes(bits, False, degree)
    ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
    rng = np.random.default_rng(len(bits))
    for level in range(ours.L, 0, -1):
        moduli = q[:level]
        ct = _uniform(rng, (3, 2), moduli, degree)
        ct[0, :, :, 0] = 0
        ct[0, :, :, 1] = (np.array(moduli, dtype=np.uint64) - np.uint64(1))[None, :]
        expected = ct
        for step in range(level, 1, -1):
            expected = ref.mod_switch_down(expected, poly_count=2, moduli_count=step)
        got = heamd.to_host(ours.mod_switch_down_to_single(heamd.to_device(ct), 2, moduli_count=level))
        assert np.array_equal(got, expected.reshape(got.shape)), level


def test_single_modulus_context(oracle):
    """One coefficient modulus: no key-switching modulus (Context.swift:102-107); ct x ct still works."""
    degree = 32
    t = oracle.generate_primes([13], True, degree)[0]
    q = oracle.generate_primes([50], False, degree)
    ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
    assert ours.L == 1
    rng = np.random.default_rng(110)
    lhs, rhs = _uniform(rng, (2, 2), q, degree), _uniform(rng, (2, 2), q, degree)
    got = heamd.to_host(ours.mul(heamd.to_device(lhs), heamd.to_device(rhs)))
    assert np.array_equal(got, ref.mul(lhs, rhs))
    with pytest.raises(heamd.HeError) as err:
        ours.relinearize(heamd.to_device(got), heamd.to_device(np.zeros((1, 2, 2, degree), dtype=np.uint64)))
    assert err.value.name == "missingRelinearizationKey"


def test_bfv_uint32_context_matches_oracle(oracle):
    """Context<Bfv<UInt32>>: gamma = 2^30 - 20405, mTilde = 2^16, 29-bit Bsk primes; every scheme
    operation word-exact against the oracle built with the same word type, plus the decrypt checks."""
    degree = 64
    t = oracle.generate_primes([10], True, degree, word_bits=32)[0]
    q = oracle.generate_primes([27, 28, 28, 29], False, degree, word_bits=32)
    ours = heamd.BfvContext(degree, t, q, word_bits=32)
    ref = oracle.BfvContext(degree, t, q, word_bits=32)
    client = BfvClient(oracle, ref, seed=97)
    assert ours.bsk_moduli() == ref.rns_tool().bsk and all(b < (1 << 29) for b in ours.bsk_moduli())
    rng = np.random.default_rng(98)
    moduli = q[:-1]
    for level in (ours.L, ours.L - 1):
        x = _uniform(rng, (4,), moduli[:level], degree)
        tool = ref.rns_tool(level)
        lifted = heamd.to_host(ours.lift_q_to_qbsk(heamd.to_device(x), level))
        assert np.array_equal(lifted, np.stack([tool.lift_q_to_qbsk(p) for p in x]))
        y = _uniform(rng, (4,), ref.qbsk_context(level).moduli, degree)
        floored = heamd.to_host(ours.floor_qbsk_to_q(heamd.to_device(y), level))
        assert np.array_equal(floored, np.stack([tool.floor_qbsk_to_q(p) for p in y]))
        assert np.array_equal(heamd.to_host(ours.scale_and_round(heamd.to_device(x), 1, moduli_count=level)),
                              np.stack([tool.scale_and_round(p, 1) for p in x]))
    r = random.Random(99)
    m1 = [r.randrange(t) for _ in range(degree)]
    m2 = [r.randrange(t) for _ in range(degree)]
    ct1, ct2 = client.encrypt(m1)[None], client.encrypt(m2)[None]
    product = heamd.to_host(ours.mul(heamd.to_device(ct1), heamd.to_device(ct2)))
    assert np.array_equal(product, ref.mul(ct1, ct2))
    key = client.relinearization_key()
    relin = heamd.to_host(ours.relinearize(heamd.to_device(product), heamd.to_device(key)))
    assert np.array_equal(relin, ref.relinearize(product, key))
    assert client.decrypt(relin[0]) == negacyclic_multiply(m1, m2, t)
    with pytest.raises(heamd.HeError) as err:
        heamd.BfvContext(degree, t, oracle.generate_primes([31, 31], False, degree), word_bits=32)
    assert err.value.name == "invalidEncryptionParameters"


def test_bfv_uint32_packed_words_round_trip(oracle):
    """A Bfv<UInt32> caller holds [UInt32] arrays: packed ciphertexts are widened on the device, multiplied and
    relinearized by the scheme layer, narrowed back -- the packed result equals the oracle's words; odd word counts
    and the alignment / null checks of the bridge are covered too."""
    import torch

    degree = 64
    t = oracle.generate_primes([10], True, degree, word_bits=32)[0]
    q = oracle.generate_primes([27, 28, 28, 29], False, degree, word_bits=32)
    ours = heamd.BfvContext(degree, t, q, word_bits=32)
    ref = oracle.BfvContext(degree, t, q, word_bits=32)
    client = BfvClient(oracle, ref, seed=101)
    r = random.Random(102)
    m1 = [r.randrange(t) for _ in range(degree)]
    m2 = [r.randrange(t) for _ in range(degree)]
    ct1, ct2 = client.encrypt(m1)[None], client.encrypt(m2)[None]

    def packed(array):
        return torch.from_numpy(np.ascontiguousarray(array, dtype=np.uint32).view(np.int32)).cuda()

    wide1, wide2 = heamd.widen_u32(packed(ct1)), heamd.widen_u32(packed(ct2))
    assert np.array_equal(heamd.to_host(wide1), ct1)
    key = client.relinearization_key()
    relin = ours.relinearize(ours.mul(wide1, wide2), heamd.widen_u32(packed(key)))
    narrow = heamd.narrow_u64(relin)
    assert narrow.dtype == torch.int32 and tuple(narrow.shape) == tuple(relin.shape)
    expected = ref.relinearize(ref.mul(ct1, ct2), key)
    assert np.array_equal(narrow.cpu().numpy().view(np.uint32).astype(np.uint64), expected)
    assert client.decrypt(expected[0]) == negacyclic_multiply(m1, m2, t)
    # ragged tail (word count not a multiple of four) and a big slab
    rng = np.random.default_rng(103)
    for words in (1, 3, 4, 7, 1030, 1 << 20):
        x = rng.integers(0, 1 << 30, size=words, dtype=np.uint64)
        wide = heamd.widen_u32(packed(x))
        assert np.array_equal(heamd.to_host(wide), x)
        assert np.array_equal(heamd.narrow_u64(wide).cpu().numpy().view(np.uint32).astype(np.uint64), x)
    lib = heamd.load_library()
    assert lib.he_words_widen_u32_device(None, None, 0, None) == 0
    assert lib.he_words_widen_u32_device(None, None, 4, None) != 0
    misaligned = packed(np.arange(8))
    assert lib.he_words_widen_u32_device(misaligned.data_ptr() + 4, wide.data_ptr(), 4, None) != 0


@pytest.mark.parametrize("degree,bits", [(64, [61] * 9), (8192, [55] * 5), (4096, [34, 60, 47, 61, 40]), (64, [61, 61, 62, 61]),
                                         (64, [34, 33, 40])])
def test_one_word_quotient_reductions_at_their_limits(oracle, degree, bits):
    """The base conversions' dot products on the one-word-quotient Barrett (device_math.hpp reduce_product_sum_bounded,
    taken when every sum of the level stays below 2^(64 + bits(p) - 1) for moduli between 2^33 and 2^61): eight 61-bit
    ciphertext moduli (the largest sums that still qualify), BASELINE config 3's moduli, mixed widths from 34 bits up --
    and two sets that must fall back to the general reduction (a 62-bit and a 33-bit modulus) -- on the words that
    maximise every sum (all residues p - 1), on zeros and on uniform words, word for word against the oracle; then a
    product through the whole pipeline."""
    t = oracle.generate_primes([17], True, degree)[0]
    q = oracle.generate_primes(bits, False, degree)
    ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
    L = ours.L
    moduli = q[:-1]
    rng = np.random.default_rng(sum(bits))
    for level in sorted({L, max(1, L - 1), 1}):
        tool = ref.rns_tool(level)
        x = _uniform(rng, (4,), moduli[:level], degree)
        x[0] = 0
        x[1] = np.array(moduli[:level], dtype=np.uint64)[:, None] - np.uint64(1)
        x[2, :, ::2] = x[1, :, ::2]
        assert np.array_equal(heamd.to_host(ours.lift_q_to_qbsk(heamd.to_device(x), level)),
                              np.stack([tool.lift_q_to_qbsk(p) for p in x])), level
        qbsk = ref.qbsk_context(level).moduli
        y = _uniform(rng, (5,), qbsk, degree)
        y[0] = 0
        y[1] = np.array(qbsk, dtype=np.uint64)[:, None] - np.uint64(1)
        y[2, :level] = 0                      # x_Bsk maximal, the Q part zero and the other way round
        y[2, level:] = y[1, level:]
        y[3, :level] = y[1, :level]
        y[3, level:] = 0
        assert np.array_equal(heamd.to_host(ours.floor_qbsk_to_q(heamd.to_device(y), level)),
                              np.stack([tool.floor_qbsk_to_q(p) for p in y])), level
    lhs, rhs = _uniform(rng, (2, 2), moduli, degree), _uniform(rng, (2, 2), moduli, degree)
    lhs[0] = np.array(moduli, dtype=np.uint64)[None, :, None] - np.uint64(1)
    rhs[0] = lhs[0]
    assert np.array_equal(heamd.to_host(ours.mul(heamd.to_device(lhs), heamd.to_device(rhs))), ref.mul(lhs, rhs))


@pytest.mark.parametrize("bits", [[62, 62, 61, 62], [62] * 9, [61, 33, 62, 45, 62]])
def test_widest_moduli_match_oracle(oracle, bits):
    """The largest moduli the reference admits (2^62 - 1, MA/Modulus.swift:177-180) and the most rows the kernels are
    specialised for (8 ciphertext moduli): the carry-counting accumulators, the sign-select conditional subtract
    (m up to 2^63) and the merged BEHZ constants at their limits, word for word against the oracle on uniform words
    and on the extreme words 0 and q - 1."""
    degree = 64
    t = oracle.generate_primes([17], True, degree)[0]
    q = oracle.generate_primes(bits, False, degree)
    ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
    L = ours.L
    moduli = q[:-1]
    rng = np.random.default_rng(len(bits))
    x = _uniform(rng, (4,), moduli, degree)
    x[0] = 0
    x[1] = np.array(moduli, dtype=np.uint64)[:, None] - np.uint64(1)
    tool = ref.rns_tool(L)
    assert np.array_equal(heamd.to_host(ours.lift_q_to_qbsk(heamd.to_device(x))),
                          np.stack([tool.lift_q_to_qbsk(p) for p in x]))
    qbsk = ref.qbsk_context(L).moduli
    y = _uniform(rng, (4,), qbsk, degree)
    y[0] = 0
    y[1] = np.array(qbsk, dtype=np.uint64)[:, None] - np.uint64(1)
    assert np.array_equal(heamd.to_host(ours.floor_qbsk_to_q(heamd.to_device(y))),
                          np.stack([tool.floor_qbsk_to_q(p) for p in y]))
    assert np.array_equal(heamd.to_host(ours.scale_and_round(heamd.to_device(x))),
                          np.stack([tool.scale_and_round(p, 1) for p in x]))
    lhs, rhs = _uniform(rng, (3, 2), moduli, degree), _uniform(rng, (3, 2), moduli, degree)
    lhs[0] = np.array(moduli, dtype=np.uint64)[None, :, None] - np.uint64(1)
    rhs[0] = lhs[0]
    product = heamd.to_host(ours.mul(heamd.to_device(lhs), heamd.to_device(rhs)))
    assert np.array_equal(product, ref.mul(lhs, rhs))
    key = _uniform(rng, (L, 2), q, degree)
    key[0] = np.array(q, dtype=np.uint64)[None, :, None] - np.uint64(1)
    ct3 = _uniform(rng, (2, 3), moduli, degree)
    ct3[0] = np.array(moduli, dtype=np.uint64)[None, :, None] - np.uint64(1)
    assert np.array_equal(heamd.to_host(ours.relinearize(heamd.to_device(ct3), heamd.to_device(key))),
                          ref.relinearize(ct3, key))
    switched = heamd.to_host(ours.mod_switch_down(heamd.to_device(lhs), 2))
    assert np.array_equal(switched, np.stack([np.stack([ref.ciphertext_context().divide_and_round_q_last(p[None])[0]
                                                        for p in ct]) for ct in lhs]))


@pytest.mark.parametrize("bits", [[62, 62, 62], [62, 45, 61, 62]])
def test_inner_product_plain_reduction_cadence(oracle, bits):
    """Bfv.innerProduct(ciphertexts:plaintexts:) where the lazy sum must be reduced inside the loop: with 62-bit moduli
    at most 8 products fit below 2^127 (and 16 below the reference's own 2^128 bound), so 21 products cross several
    reductions; degree 256 takes the carry-counting kernel, words at q - 1 maximise every sum, `nil` plaintexts
    (Bfv.swift:486-489) shift the cadence column by column."""
    degree = 256
    t = oracle.generate_primes([17], True, degree)[0]
    q = oracle.generate_primes(bits, False, degree)
    ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
    moduli = q[:-1]
    rng = np.random.default_rng(len(bits))
    count, columns = 21, 5
    cts = _uniform(rng, (count, 2), moduli, degree)
    pts = _uniform(rng, (columns, count), moduli, degree)
    top = np.array(moduli, dtype=np.uint64)[:, None] - np.uint64(1)
    cts[:, :, :, : degree // 2] = top[None, None, :, :]
    pts[0] = top[None, :, :]
    pts[1, :, :, ::2] = top[None, :, :]
    present = np.ones((columns, count), dtype=np.uint8)
    present[2, ::3] = 0
    present[3, 5:] = 0
    present[4, :] = 0
    got = heamd.to_host(ours.inner_product_plain(heamd.to_device(cts), heamd.to_device(pts), present, 2, columns))
    for col in range(columns):
        assert np.array_equal(got[col], ref.inner_product_plain(cts, pts[col], present[col])), col
    unmasked = heamd.to_host(ours.inner_product_plain(heamd.to_device(cts), heamd.to_device(pts), None, 2, columns))
    for col in (0, 1, 4):
        assert np.array_equal(unmasked[col], ref.inner_product_plain(cts, pts[col], None)), col


@pytest.mark.parametrize("bits,polys", [([56, 56, 56], 2), ([56, 40, 55, 56], 2), ([56, 56, 56], 4), ([56, 56, 56], 6),
                                        ([56, 55, 56], 8), ([57, 56, 57], 8)])
def test_inner_product_plain_narrow_moduli(oracle, bits, polys):
    """Moduli below 2^56 take the accumulator without the middle column's carry counts (device_math.hpp NARROW), folded
    every 64 products: 150 products of words at q - 1 -- the largest cross terms a canonical operand can make -- cross
    the fold twice, for one query and for 2, 3 and 4 queries side by side; a 57-bit modulus keeps the full accumulator.
    Word-exact against the oracle's Bfv.innerProduct(ciphertexts:plaintexts:) per query."""
    degree = 256
    t = oracle.generate_primes([17], True, degree)[0]
    q = oracle.generate_primes(bits, False, degree)
    ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
    moduli = q[:-1]
    rng = np.random.default_rng(polys + len(bits))
    count, columns, queries = 150, 3, polys // 2
    cts = _uniform(rng, (count, queries, 2), moduli, degree)
    pts = _uniform(rng, (columns, count), moduli, degree)
    top = np.array(moduli, dtype=np.uint64)[:, None] - np.uint64(1)
    cts[:, :, :, :, : degree // 2] = top[None, None, None, :, :]
    pts[0] = top[None, :, :]
    pts[1, :, :, ::2] = top[None, :, :]
    present = np.ones((columns, count), dtype=np.uint8)
    present[2, ::3] = 0
    got = heamd.to_host(ours.inner_product_plain(heamd.to_device(cts), heamd.to_device(pts), present, polys, columns))
    got = got.reshape(columns, queries, 2, ours.L, degree)
    for query in range(queries):
        own = np.ascontiguousarray(cts[:, query])
        for col in range(columns):
            assert np.array_equal(got[col, query], ref.inner_product_plain(own, pts[col], present[col])), (query, col)


@pytest.mark.parametrize("degree,bits", [(4096, [55, 55, 55]), (16384, [55, 50, 55]), (4096, [61, 45, 62, 55]),
                                         (8192, [55, 61, 50, 55]), (16384, [61, 45, 62, 55]), (32768, [55, 50, 55]),
                                         (32768, [62, 55]), (8192, [29, 60, 60]), (8192, [40, 60, 60]), (4096, [60, 60, 60]),
                                         (8192, [28, 60, 60])])
def test_fused_transform_loads_other_degrees(oracle, degree, bits):
    """The transforms with a fused load stage (key-switching decomposition, plaintext lift, tensor product, key inner
    product) exist per tiled degree: N = 4096 and 16384 instantiations, headroom and mixed [Q, Bsk] bands, and moduli
    that force the exact butterflies; N = 32768 runs them as four interleaved sub-rows with the fused loads that pay there
    (Q band, tensor product, plaintext lift, key-switching decomposition -- round 5) and the key inner product unfused; the reference's 60-bit parameter sets (n_8192_logq_29_60_60 and its
    siblings, EncryptionParameters.swift:257-263) split their [Q, Bsk] records into runs of one butterfly class each -- the
    small modulus, the 60-bit ones on the fold butterflies, the auxiliary primes on the other fold form.  Word for word
    against the oracle."""
    t = oracle.generate_primes([17], True, degree)[0]
    q = oracle.generate_primes(bits, False, degree)
    ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
    moduli, L = q[:-1], len(q) - 1
    rng = np.random.default_rng(degree + len(bits))
    lhs, rhs = _uniform(rng, (1, 2), moduli, degree), _uniform(rng, (1, 2), moduli, degree)
    product = heamd.to_host(ours.mul(heamd.to_device(lhs), heamd.to_device(rhs)))
    assert np.array_equal(product, ref.mul(lhs, rhs))
    key = _uniform(rng, (L, 2), q, degree)
    relin = heamd.to_host(ours.relinearize(heamd.to_device(product), heamd.to_device(key)))
    assert np.array_equal(relin, ref.relinearize(product, key))
    values = rng.integers(0, t, size=(2, degree), dtype=np.uint64)
    lifted = heamd.to_host(ours.plaintext_to_eval(heamd.to_device(values)))
    assert np.array_equal(lifted, np.concatenate([ref.plaintext_to_eval(v) for v in values]).reshape(lifted.shape))


@pytest.mark.parametrize("bits,batch", [([55, 55, 55, 55], 9), ([55, 41, 50, 55], 5), ([61, 45, 62, 55], 3), ([50, 55], 140),
                                        ([55, 55], 7), ([55, 55, 54], 6)])
def test_interleaved_fused_loads_at_16384(oracle, bits, batch):
    """N = 16384: the transforms with a fused load stage run as two interleaved sub-rows of the 8192-point kernel (round 5;
    ntt_forward_interleaved<1, MODE, kSourceSpread | kSourceLift | kSourceRows>, ntt_inverse_interleaved<1, MODE, ...,
    kInverseFromTensor | kInverseFromKeyMac>) -- ct x ct, relinearize, applyGalois' decomposition without the automorphism and
    Plaintext.convertToEvalFormat on odd batches: limb-wise moduli, moduli of very different sizes (the decomposition reduces
    its source row first), 61 / 62-bit moduli (the [0, 8p) and the exact butterflies), a batch wide enough for the key
    switch's fused end on the other degrees, and moduli 2^b - d that all qualify for the shift-folded products (the first two
    55-bit primes of the degree; a 54-bit one beside them): every fused load on those products.  Word for word against the oracle (EncryptionParameters.swift:200-206 allows the
    degree)."""
    from conftest import host_threads

    degree = 16384
    t = oracle.generate_primes([17], True, degree)[0]
    q = oracle.generate_primes(bits, False, degree)
    ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
    moduli, L = q[:-1], len(q) - 1
    rng = np.random.default_rng(16384 + batch)
    lhs, rhs = _uniform(rng, (batch, 2), moduli, degree), _uniform(rng, (batch, 2), moduli, degree)
    for i, m in enumerate(moduli):
        lhs[0, :, i, :] = m - 1
        rhs[0, :, i, :] = m - 1
    product = heamd.to_host(ours.mul(heamd.to_device(lhs), heamd.to_device(rhs)))
    expected = ref.mul(lhs, rhs, threads=host_threads())
    assert np.array_equal(product, expected)
    key = _uniform(rng, (L, 2), q, degree)
    relin = heamd.to_host(ours.relinearize(heamd.to_device(product), heamd.to_device(key)))
    assert np.array_equal(relin, ref.relinearize(expected, key, threads=host_threads()))
    values = rng.integers(0, t, size=(3, degree), dtype=np.uint64)
    lifted = heamd.to_host(ours.plaintext_to_eval(heamd.to_device(values)))
    assert np.array_equal(lifted, np.concatenate([ref.plaintext_to_eval(v) for v in values]).reshape(lifted.shape))


@pytest.mark.parametrize("bits,count", [([62, 62, 62], 21), ([62, 45, 61, 62], 9), ([62] * 9, 17)])
def test_inner_product_ct_ct_reduction_cadence(oracle, bits, count):
    """Bfv.innerProduct(ct, ct) where the [Q, Bsk] accumulators must be reduced inside the loop (Bfv.swift:339-353):
    maxLazyProductAccumulationCount() / 2 is 8 with 62-bit moduli, so 9, 17 and 21 pairs cross one, two and two
    reductions; words at q - 1 maximise every partial sum."""
    degree = 64
    t = oracle.generate_primes([17], True, degree)[0]
    q = oracle.generate_primes(bits, False, degree)
    ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
    moduli = q[:-1]
    rng = np.random.default_rng(count)
    lhs, rhs = _uniform(rng, (count, 2), moduli, degree), _uniform(rng, (count, 2), moduli, degree)
    top = np.array(moduli, dtype=np.uint64)[:, None] - np.uint64(1)
    lhs[:, :, :, : degree // 2] = top[None, None, :, :]
    rhs[:, :, :, : degree // 4] = top[None, None, :, :]
    got = heamd.to_host(ours.inner_product(heamd.to_device(lhs), heamd.to_device(rhs)))
    assert np.array_equal(got, ref.inner_product(lhs, rhs))


@pytest.mark.parametrize("bits,count,items", [([62, 62, 62], 21, 3), ([62, 45, 61, 62], 9, 5), ([50, 50, 50], 4, 1)])
def test_inner_product_shared_left_vector(oracle, bits, count, items):
    """he_bfv_inner_product_shared_device (the PIR remaining-dimension step over all result groups at once): item i is
    Bfv.innerProduct(lhs, rhs[i]) word for word -- the oracle's and the per-item entry point's -- across the in-loop
    accumulator reductions of Bfv.swift:339-353 (words at q - 1 in the shared left vector maximise every partial sum)."""
    degree = 64
    t = oracle.generate_primes([17], True, degree)[0]
    q = oracle.generate_primes(bits, False, degree)
    ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
    moduli = q[:-1]
    rng = np.random.default_rng(100 * count + items)
    lhs, rhs = _uniform(rng, (count, 2), moduli, degree), _uniform(rng, (items, count, 2), moduli, degree)
    top = np.array(moduli, dtype=np.uint64)[:, None] - np.uint64(1)
    lhs[:, :, :, : degree // 2] = top[None, None, :, :]
    rhs[:, :, :, :, : degree // 4] = top[None, None, None, :, :]
    left = heamd.to_device(lhs)
    got = heamd.to_host(ours.inner_product_shared(left, heamd.to_device(rhs)))
    assert got.shape[0] == items
    for i in range(items):
        assert np.array_equal(got[i], ref.inner_product(lhs, rhs[i]))
        assert np.array_equal(got[i], heamd.to_host(ours.inner_product(left, heamd.to_device(rhs[i]))))


def test_inner_product_shared_config3_shape(oracle, config3):
    """The same on BASELINE config 3's ring (N=8192, L=4): 4 result groups of 6 ciphertexts against one query slice."""
    ours, ref = config3
    rng = np.random.default_rng(69)
    moduli = ref.ciphertext_context().moduli
    lhs, rhs = _uniform(rng, (6, 2), moduli, ours.degree), _uniform(rng, (4, 6, 2), moduli, ours.degree)
    got = heamd.to_host(ours.inner_product_shared(heamd.to_device(lhs), heamd.to_device(rhs)))
    for i in range(4):
        assert np.array_equal(got[i], ref.inner_product(lhs, rhs[i]))


def test_inner_product_ct_ct_config3_shape(oracle, config3):
    """Bfv.innerProduct(ct, ct) on BASELINE config 3's ring (N=8192, L=4), 12 pairs of uniform ciphertexts: every word
    equals the oracle's (the PIR second dimension at a realistic size)."""
    ours, ref = config3
    rng = np.random.default_rng(66)
    moduli = ref.ciphertext_context().moduli
    lhs, rhs = _uniform(rng, (12, 2), moduli, ours.degree), _uniform(rng, (12, 2), moduli, ours.degree)
    got = heamd.to_host(ours.inner_product(heamd.to_device(lhs), heamd.to_device(rhs)))
    assert np.array_equal(got, ref.inner_product(lhs, rhs))


def test_mul_and_relinearize_full_batch_properties(oracle, config3):
    """BASELINE config 3 at its full batch (1024 ciphertext pairs, generated on the device): every output word is
    canonical, EVERY product and every relinearized ciphertext equals the multi-threaded oracle's word for word (on a host
    with fewer than 8 threads: items sampled across the batch -- first, last, a middle one and the one straddling the odd
    workgroup tail), and the batch result does not depend on the batch it was computed in."""
    import torch

    ours, ref = config3
    moduli = ref.ciphertext_context().moduli
    batch, degree, L = 1024, ours.degree, ours.L
    gen = torch.Generator(device="cuda")
    gen.manual_seed(67)
    bound = torch.tensor(moduli, dtype=torch.int64, device="cuda").view(1, 1, L, 1)
    lhs = torch.randint(0, 1 << 62, (batch, 2, L, degree), dtype=torch.int64, device="cuda", generator=gen) % bound
    rhs = torch.randint(0, 1 << 62, (batch, 2, L, degree), dtype=torch.int64, device="cuda", generator=gen) % bound
    rng = np.random.default_rng(68)
    key = _uniform(rng, (L, 2), ref.key_switching_context().moduli, degree)
    key_dev = heamd.to_device(key)
    product = ours.mul(lhs, rhs)
    relin = ours.relinearize(product, key_dev)
    assert bool((product < bound).all()) and bool((relin < bound).all())
    from conftest import exhaustive_parity, host_threads

    if exhaustive_parity():
        # every product and every relinearized ciphertext against the multi-threaded oracle, 128 items (100 MiB) at a time
        compared = 0
        for first in range(0, batch, 128):
            part = slice(first, first + 128)
            expected_product = ref.mul(heamd.to_host(lhs[part].contiguous()), heamd.to_host(rhs[part].contiguous()),
                                       threads=host_threads())
            assert np.array_equal(heamd.to_host(product[part].contiguous()), expected_product), first
            assert np.array_equal(heamd.to_host(relin[part].contiguous()),
                                  ref.relinearize(expected_product, key, threads=host_threads())), first
            compared += expected_product.shape[0]
        assert compared == batch
        print(f"ct x ct + relinearize: {compared} of {batch} items compared with the oracle word for word")
    else:
        sample = [0, 511, 1022, 1023]
        host_lhs, host_rhs = heamd.to_host(lhs[sample].contiguous()), heamd.to_host(rhs[sample].contiguous())
        expected_product = ref.mul(host_lhs, host_rhs)
        assert np.array_equal(heamd.to_host(product[sample].contiguous()), expected_product)
        assert np.array_equal(heamd.to_host(relin[sample].contiguous()), ref.relinearize(expected_product, key))
        print(f"ct x ct + relinearize: {len(sample)} of {batch} items compared with the oracle (small host)")
    # an odd sub-batch (the ragged tail of the row-pair launches) gives the same words as the full batch
    sub = slice(513, 1020)
    again = ours.relinearize(ours.mul(lhs[sub].contiguous(), rhs[sub].contiguous()), key_dev)
    assert torch.equal(again, relin[sub])


@pytest.mark.parametrize("count", [7, 8, 23])
def test_uint32_extremes_and_inner_product_cadence(oracle, count):
    """Bfv<UInt32> in 4-byte arithmetic (rns_kernels.hip WordArith<uint32_t>): 64-bit sums of products below 2^60 and
    32-bit Shoup products.  Words at q - 1 and 0 -- the largest sums the base conversions, the tensor product and the
    ct . ct inner product (folded every 7 items: 7, 8 and 23 pairs sit on, just past and three times past the fold) can
    see -- word for word against the 32-bit oracle, with moduli right under 2^30."""
    degree = 64
    t = oracle.generate_primes([10], True, degree, word_bits=32)[0]
    q = oracle.generate_primes([30, 30, 30, 30], False, degree, word_bits=32)
    ours, ref = heamd.BfvContext32(degree, t, q), oracle.BfvContext(degree, t, q, word_bits=32)
    moduli = q[:-1]
    dev, host = heamd.to_device32, heamd.to_host32
    rng = np.random.default_rng(count)
    top = np.array(moduli, dtype=np.uint64)[:, None] - np.uint64(1)
    lhs, rhs = _uniform(rng, (count, 2), moduli, degree), _uniform(rng, (count, 2), moduli, degree)
    lhs[:, :, :, : degree // 2] = top[None, None, :, :]
    rhs[:, :, :, : degree // 4] = top[None, None, :, :]
    rhs[:, :, :, -4:] = 0
    assert np.array_equal(host(ours.inner_product(dev(lhs), dev(rhs))), ref.inner_product(lhs, rhs))
    product = host(ours.mul(dev(lhs[:3]), dev(rhs[:3])))
    assert np.array_equal(product, ref.mul(lhs[:3], rhs[:3]))
    key = _uniform(rng, (ours.L, 2), q, degree)
    key[:, :, :, : degree // 2] = (np.array(q, dtype=np.uint64)[:, None] - np.uint64(1))[None, None, :, :]
    assert np.array_equal(host(ours.relinearize(dev(product), dev(key))), ref.relinearize(product, key))
    tool = ref.rns_tool(ours.L)
    x = np.broadcast_to(top, (2, len(moduli), degree)).copy()
    assert np.array_equal(host(ours.lift_q_to_qbsk(dev(x), ours.L)), np.stack([tool.lift_q_to_qbsk(p) for p in x]))
    ext = ref.qbsk_context(ours.L).moduli
    y = np.broadcast_to(np.array(ext, dtype=np.uint64)[:, None] - np.uint64(1), (2, len(ext), degree)).copy()
    assert np.array_equal(host(ours.floor_qbsk_to_q(dev(y), ours.L)), np.stack([tool.floor_qbsk_to_q(p) for p in y]))


def test_lift_two_coefficients_per_lane(oracle):
    """liftQToQBsk on 8-byte slabs at the C3 ring (N=8192, five 55-bit moduli): the bounded kernel takes two coefficients
    per lane (one 16-byte access per row) when every row starts on a 16-byte boundary and one otherwise -- a slab that
    starts 8 bytes into its allocation gives the same words as the aligned one, and both equal the oracle's."""
    import torch

    degree = 8192
    q = oracle.generate_primes([55] * 5, False, degree)
    ours, ref = heamd.BfvContext(degree, 557057, q), oracle.BfvContext(degree, 557057, q)
    rng = np.random.default_rng(733)
    x = _uniform(rng, (3,), q[:-1], degree)
    expected = np.stack([ref.rns_tool().lift_q_to_qbsk(p) for p in x])
    aligned = heamd.to_device(x)
    assert aligned.data_ptr() % 16 == 0
    assert np.array_equal(heamd.to_host(ours.lift_q_to_qbsk(aligned)), expected)
    shifted = torch.empty(aligned.numel() + 1, dtype=aligned.dtype, device=aligned.device)[1:]
    shifted.copy_(aligned.reshape(-1))
    assert shifted.data_ptr() % 16 == 8
    assert np.array_equal(heamd.to_host(ours.lift_q_to_qbsk(shifted)), expected)


def test_uint32_base_conversions_four_words_per_lane(oracle):
    """Bfv<UInt32> lift / floor at n_4096_logq_27_28_28: the kernels take four 4-byte words per lane (one 16-byte access)
    when every row starts on a 16-byte boundary and one word per lane otherwise -- a slab that starts 4 bytes into its
    allocation gives the same words as the aligned one, and both equal the 32-bit oracle's."""
    import torch

    degree = 4096
    t = (1 << 16) + 1
    q = oracle.generate_primes([27, 28, 28], False, degree, word_bits=32)
    ours = heamd.BfvContext32(degree, t, q)
    ref = oracle.BfvContext(degree, t, q, word_bits=32)
    rng = np.random.default_rng(731)
    tool = ref.rns_tool(ours.L)
    x = _uniform(rng, (3,), q[:-1], degree)
    y = _uniform(rng, (3,), ref.qbsk_context(ours.L).moduli, degree)
    for words, run, expected in ((x, ours.lift_q_to_qbsk, np.stack([tool.lift_q_to_qbsk(p) for p in x])),
                                 (y, ours.floor_qbsk_to_q, np.stack([tool.floor_qbsk_to_q(p) for p in y]))):
        aligned = heamd.to_device32(words)
        assert aligned.data_ptr() % 16 == 0
        assert np.array_equal(heamd.to_host32(run(aligned)), expected)
        shifted = torch.empty(aligned.numel() + 1, dtype=aligned.dtype, device=aligned.device)[1:]
        shifted.copy_(aligned.reshape(-1))
        assert shifted.data_ptr() % 16 == 4
        assert np.array_equal(heamd.to_host32(run(shifted)), expected)


def test_bfv_uint32_packed_slabs_match_oracle(oracle):
    """Bfv<UInt32> on packed [UInt32] slabs (he_*_device_u32): no word is widened in memory.  Every scheme operation
    word for word against the 32-bit oracle -- lift / floor / scaleAndRound at two levels, ct x ct, relinearize,
    applyGalois, mod-switch, ct x pt, both inner products, plaintext <-> Eval -- and the decrypt checks; the 8-byte
    path on the same context gives the same words."""
    import torch

    degree = 64
    t = oracle.generate_primes([10], True, degree, word_bits=32)[0]
    q = oracle.generate_primes([27, 28, 28, 29], False, degree, word_bits=32)
    ours = heamd.BfvContext32(degree, t, q)
    ref = oracle.BfvContext(degree, t, q, word_bits=32)
    client = BfvClient(oracle, ref, seed=111)
    rng = np.random.default_rng(112)
    moduli = q[:-1]
    dev, host = heamd.to_device32, heamd.to_host32
    for level in (ours.L, ours.L - 1):
        x = _uniform(rng, (4,), moduli[:level], degree)
        tool = ref.rns_tool(level)
        assert np.array_equal(host(ours.lift_q_to_qbsk(dev(x), level)), np.stack([tool.lift_q_to_qbsk(p) for p in x]))
        y = _uniform(rng, (4,), ref.qbsk_context(level).moduli, degree)
        assert np.array_equal(host(ours.floor_qbsk_to_q(dev(y), level)), np.stack([tool.floor_qbsk_to_q(p) for p in y]))
        assert np.array_equal(host(ours.scale_and_round(dev(x), 1, moduli_count=level)),
                              np.stack([tool.scale_and_round(p, 1) for p in x]))
    r = random.Random(113)
    m1 = [r.randrange(t) for _ in range(degree)]
    m2 = [r.randrange(t) for _ in range(degree)]
    ct1 = np.stack([client.encrypt(m1), client.encrypt(m2)])
    ct2 = np.stack([client.encrypt(m2), client.encrypt(m2)])
    product = ours.mul(dev(ct1), dev(ct2))
    expected_product = ref.mul(ct1, ct2)
    assert np.array_equal(host(product), expected_product)
    key = client.relinearization_key()
    relin = ours.relinearize(product, dev(key))
    expected_relin = ref.relinearize(expected_product, key)
    assert np.array_equal(host(relin), expected_relin)
    assert client.decrypt(host(relin)[0]) == negacyclic_multiply(m1, m2, t)
    # the 8-byte entry points on zero-extended words agree
    wide = heamd.BfvContext.relinearize(ours, heamd.BfvContext.mul(ours, heamd.to_device(ct1), heamd.to_device(ct2)),
                                        heamd.to_device(key))
    assert np.array_equal(heamd.to_host(wide), expected_relin)
    # applyGalois
    element = 3
    galois_key = client.galois_key(element)
    rotated = ours.apply_galois(dev(ct1), element, dev(galois_key))
    assert np.array_equal(host(rotated), ref.apply_galois(ct1, element, galois_key))
    # mod switch, then a product below the top level
    lower = ours.mod_switch_down(relin, 2)
    expected_lower = ref.mod_switch_down(expected_relin, poly_count=2)
    assert np.array_equal(host(lower), expected_lower)
    assert client.decrypt(host(lower)[0], moduli_count=ours.L - 1) == negacyclic_multiply(m1, m2, t)
    assert np.array_equal(host(ours.mul(lower, lower, moduli_count=ours.L - 1)),
                          ref.mul(expected_lower, expected_lower, moduli_count=ours.L - 1))
    # ciphertext +- plaintext (Bfv.swift:110-117)
    messages = np.array([m2, [t - 1] * degree], dtype=np.uint64)
    for subtract in (False, True):
        assert np.array_equal(host(ours.add_plain_(dev(ct1), dev(messages), 2, subtract)),
                              ref.plaintext_translate(ct1, messages, 2, subtract))
    summed = host(ours.add_plain_(dev(ct1), dev(messages[:1]), 2))
    assert client.decrypt(summed[0]) == [(a + b) % t for a, b in zip(m1, m2)]
    # plaintexts and ct x pt
    pts = np.array([[r.randrange(t) for _ in range(degree)] for _ in range(3)], dtype=np.uint64)
    pt_eval = ours.plaintext_to_eval(dev(pts))
    expected_eval = ref.plaintext_to_eval(pts)
    assert np.array_equal(host(pt_eval), expected_eval)
    assert np.array_equal(host(ours.plaintext_to_coeff(pt_eval)), pts)
    qctx = ref.ciphertext_context()
    cts_eval = np.stack([np.stack([qctx.forward_ntt(p[None])[0] for p in client.encrypt(m)]) for m in (m1, m2, m1)])
    mask = np.array([[1, 0, 1], [1, 1, 1]], dtype=np.uint8)
    columns = np.stack([expected_eval, expected_eval[::-1].copy()])
    got = ours.inner_product_plain_resident(dev(cts_eval), dev(columns), torch.from_numpy(mask).cuda(), 2, 2)
    for c in range(2):
        assert np.array_equal(host(got)[c], ref.inner_product_plain(cts_eval, columns[c], mask[c]))
    scaled = dev(cts_eval[:1].copy())
    ours.mul_plain_(scaled, dev(expected_eval[:1].copy()), 2)
    assert np.array_equal(host(scaled)[0], ref.inner_product_plain(cts_eval[:1], expected_eval[:1], None))
    # ct . ct inner product
    assert np.array_equal(host(ours.inner_product(dev(ct1), dev(ct2))), ref.inner_product(ct1, ct2))
    # a Bfv<UInt64> context refuses 4-byte slabs
    wide_q = oracle.generate_primes([40, 40, 41], False, degree)
    wide_ctx = heamd.BfvContext(degree, oracle.generate_primes([17], True, degree)[0], wide_q)
    zeros = dev(np.zeros((1, 2 * wide_ctx.L + 1, degree), dtype=np.uint64))
    status = heamd.load_library().he_rns_lift_q_to_qbsk_device_u32(wide_ctx.h, wide_ctx.L, zeros.data_ptr(),
                                                                    zeros.data_ptr(), 1, None)
    assert status == 16  # HE_ERR_INVALID_ARGUMENT


@pytest.mark.parametrize("degree,bits,batch", [(4096, [29, 60, 60], 176), (4096, [60, 60, 60], 176), (8192, [55, 55, 60], 176),
                                               (4096, [50, 60, 50], 176), (4096, [55, 55, 55], 177)])
def test_key_switch_on_runs_of_butterfly_classes(oracle, degree, bits, batch):
    """relinearize and applyGalois on batches wide enough (more than two workgroup generations of rows) for the
    key-switching moduli to be launched as runs of one butterfly class each -- the reference's 60-bit parameter sets
    (29 | 60, 60: the 60-bit rows on the fold butterflies; the centred q_ks word exceeds the 29-bit modulus and is reduced
    as the key switch ends in the key-MAC transform's store), all-60-bit moduli (one fold run), 55, 55 | 60 (q_ks above
    both ciphertext moduli) and 50 | 60 | 50 (three classes) -- and an odd batch of the usual 55-bit moduli (the last
    polynomial goes one row per workgroup).  Word for word against the oracle (Bfv+Keys.swift:123-208, Bfv.swift:174-219)."""
    t = oracle.generate_primes([17], True, degree)[0]
    q = oracle.generate_primes(bits, False, degree)
    ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
    moduli, L = q[:-1], len(q) - 1
    rng = np.random.default_rng(degree + sum(bits))
    ct3 = _uniform(rng, (batch, 3), moduli, degree)
    key = _uniform(rng, (L, 2), q, degree)
    relin = heamd.to_host(ours.relinearize(heamd.to_device(ct3), heamd.to_device(key)))
    assert np.array_equal(relin, ref.relinearize(ct3, key, threads=16))
    element = 2 * degree - 1
    ct = _uniform(rng, (batch, 2), moduli, degree)
    rotated = heamd.to_host(ours.apply_galois(heamd.to_device(ct), element, heamd.to_device(key)))
    assert np.array_equal(rotated, ref.apply_galois(ct, element, key))


def test_pipelines_on_moduli_at_the_edge_of_the_shift_folded_products(oracle, monkeypatch):
    """ct x ct (row-fused and unfused) and relinearize with ciphertext moduli 2^b - d whose d is as large as kModeFoldLazy
    allows (tests/test_gpu_ntt.py _primes_below_power_of_two) -- the generated primes of every other test sit at the small end --
    and with one modulus just past the bound among them (that band then runs the limb-wise products): word for word."""
    from conftest import host_threads
    from test_gpu_ntt import _primes_below_power_of_two

    monkeypatch.setenv("HEAMD_BEHZ_FUSED_ABOVE", "256")  # (the row-fused kernels from 29 pairs on, as the batch below assumes)
    degree = 4096
    t = oracle.generate_primes([17], True, degree)[0]
    edge = [p for bits in (55, 54, 52, 50) for p in _primes_below_power_of_two(oracle, bits, degree, True)]
    past = _primes_below_power_of_two(oracle, 55, degree, False)
    for q in (edge, edge[:2] + past + edge[2:3]):
        ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
        L = ours.L
        moduli = ref.ciphertext_context(L).moduli
        rng = np.random.default_rng(len(q) + q[2] % 97)
        batch = 48  # 48 x 7 rows: wide enough for behz_kernels.hip
        lhs, rhs = _uniform(rng, (batch, 2), moduli, degree), _uniform(rng, (batch, 2), moduli, degree)
        for i, m in enumerate(moduli):
            lhs[1, :, i, :] = m - 1
            rhs[1, :, i, :] = m - 1
        product = ref.mul(lhs, rhs, L, threads=host_threads())
        assert np.array_equal(heamd.to_host(ours.mul(heamd.to_device(lhs), heamd.to_device(rhs), L)), product), q
        assert np.array_equal(heamd.to_host(ours.mul(heamd.to_device(lhs[:2].copy()), heamd.to_device(rhs[:2].copy()), L)), product[:2]), q
        key = _uniform(rng, (ours.L, 2), ref.key_switching_context().moduli, degree)
        got = heamd.to_host(ours.relinearize(heamd.to_device(product), heamd.to_device(key), L))
        assert np.array_equal(got, ref.relinearize(product, key, L, threads=host_threads())), q
