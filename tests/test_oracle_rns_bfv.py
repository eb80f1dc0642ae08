"""Pins the oracle's BEHZ toolbox and BFV scheme ops.

Mirrors Tests/HomomorphicEncryptionTests/RnsToolTests.swift, RnsBaseConverterTests.swift and the semantic
(decrypt) checks of Sources/_TestUtilities/HeApiTestUtils.swift:494-720,1223-1285.
"""
import random

import numpy as np
import pytest

from bfv_helpers import BfvClient, crt_compose, crt_decompose, galois_plain, negacyclic_multiply

MTILDE = 1 << 32


def _prod(xs):
    out = 1
    for x in xs:
        out *= x
    return out


def _poly_from_bigints(xs, moduli):
    return np.array([[x % m for x in xs] for m in moduli], dtype=np.uint64)


def _column(data, k):
    return [int(v) for v in data[:, k]]


def test_bsk_generation(oracle):
    # Bsk for N=8192, L=4
    ctx = oracle.PolyContext(8192, oracle.generate_primes([55] * 4, False, 8192))
    tool = oracle.RnsTool(ctx, 557057)
    assert tool.bsk == [1152921504606994433, 1152921504607191041, 1152921504607223809, 1152921504607338497,
                        1152921504607518721]
    assert tool.bsk == oracle.generate_primes([61] * 5, True, 8192)


def test_small_montgomery_reduce_known_answers(oracle, kats):
    # RnsToolTests.swift:118-166
    for c in kats["small_montgomery_reduce"]["cases"]:
        moduli = oracle.generate_primes(c["q_bits"], True)
        ctx = oracle.PolyContext(c["degree"], moduli)
        tool = oracle.RnsTool(ctx, 2)
        data = np.array(c["input_in_units_of_mtilde"], dtype=np.uint64) * np.uint64(MTILDE)
        got = tool.small_montgomery_reduce(data)
        assert np.array_equal(got, np.array(c["expected"], dtype=np.uint64)), c


@pytest.mark.parametrize("degree,bits", [(4, [20, 20]), (8, [30, 30, 30]), (16, [40, 40, 40, 40])])
def test_lift_q_to_qbsk_is_exact_centered_lift(oracle, degree, bits):
    # RnsToolTests.swift:168-208
    moduli = oracle.generate_primes(bits, True)
    ctx = oracle.PolyContext(degree, moduli)
    tool = oracle.RnsTool(ctx, 2)
    q = _prod(moduli)
    rng = random.Random(10)
    xs = [rng.randrange(q) for _ in range(degree)]
    out = tool.lift_q_to_qbsk(_poly_from_bigints(xs, moduli))
    qbsk_moduli = moduli + tool.bsk
    big = _prod(qbsk_moduli)
    for k, x in enumerate(xs):
        expected = big - (q - x) if x > q // 2 else x
        assert _column(out, k) == crt_decompose(expected, qbsk_moduli)


@pytest.mark.parametrize("degree,bits", [(32, [20, 20]), (16, [30, 30, 30]), (8, [40, 40, 40, 40])])
def test_convert_approximate_bsk_mtilde(oracle, degree, bits):
    # RnsToolTests.swift:66-116
    moduli = oracle.generate_primes(bits + [bits[-1]], True)
    t = min(moduli)
    moduli = [m for m in moduli if m != t]
    ctx = oracle.PolyContext(degree, moduli)
    tool = oracle.RnsTool(ctx, t)
    q = _prod(moduli)
    rng = random.Random(11)
    xs = [rng.randrange(q) for _ in range(degree)]
    out = tool.convert_approximate_bsk_mtilde(_poly_from_bigints(xs, moduli))
    out_moduli = tool.bsk + [MTILDE]
    base = _prod(out_moduli)
    for k, x in enumerate(xs):
        candidates = [crt_decompose(((x * (MTILDE % q)) % q + a * q) % base, out_moduli) for a in range(len(moduli))]
        assert _column(out, k) in candidates


@pytest.mark.parametrize("degree,bits", [(4, [20, 20]), (8, [30, 30, 30]), (16, [40, 40, 40, 40])])
def test_approximate_floor(oracle, degree, bits):
    # RnsToolTests.swift:210-260
    moduli = oracle.generate_primes(bits, True)
    ctx = oracle.PolyContext(degree, moduli)
    tool = oracle.RnsTool(ctx, 2)
    q, bsk = _prod(moduli), _prod(tool.bsk)
    qbsk_moduli = moduli + tool.bsk
    qbsk = q * bsk
    rng = random.Random(12)
    xs = [qbsk - 1, 1] + [rng.randrange(qbsk) for _ in range(degree - 2)]
    out = tool.approximate_floor(_poly_from_bigints(xs, qbsk_moduli))
    for k, x in enumerate(xs):
        candidates = []
        for a in range(len(moduli)):
            candidates.append(crt_decompose((x // q + a) % bsk, tool.bsk))
            candidates.append(crt_decompose((x // q + bsk - a) % bsk, tool.bsk))
        assert _column(out, k) in candidates


@pytest.mark.parametrize("degree,bits", [(4, [20, 20]), (8, [30, 30, 30])])
def test_convert_approximate_bsk_to_q_is_exact(oracle, degree, bits):
    # RnsToolTests.swift:262-305
    moduli = oracle.generate_primes(bits, True)
    ctx = oracle.PolyContext(degree, moduli)
    tool = oracle.RnsTool(ctx, 2)
    q, bsk_prod = _prod(moduli), _prod(tool.bsk)
    rng = random.Random(13)
    xs = [rng.randrange(q) for _ in range(degree)]
    out = tool.convert_approximate_bsk_to_q(_poly_from_bigints(xs, tool.bsk))
    for k, x in enumerate(xs):
        expected = q - ((bsk_prod - x) % q) if x > bsk_prod // 2 else x % q
        assert _column(out, k) == crt_decompose(expected, moduli)


def test_convert_approximate_bsk_to_q_below_the_top_level_uses_the_shared_msk_context(oracle):
    """RnsTool.swift:44-62, 240-250: one mSkContext (the TOP level's m_sk) serves every level's tool, so below the top
    level rnsConvertBtoMSk converts B to the top m_sk and inverseBModMSk is (B mod top m_sk)^-1 mod the level's m_sk.
    Restated here with Python integers, statement by statement of :402-450, and compared with the oracle's words."""
    degree = 16
    t = oracle.generate_primes([17], True, degree)[0]
    q = oracle.generate_primes([40, 40, 40, 41], False, degree)
    ctx = oracle.BfvContext(degree, t, q)
    top_msk = ctx.rns_tool(ctx.L).bsk[-1]
    for level in (ctx.L, ctx.L - 1, 1):
        tool = ctx.rns_tool(level)
        moduli = ctx.ciphertext_context(level).moduli
        bsk = tool.bsk
        b_moduli, m_sk = bsk[:-1], bsk[-1]
        assert (m_sk == top_msk) == (level == ctx.L)
        b = _prod(b_moduli)
        inverse_b = pow(b % top_msk, -1, m_sk)
        rng = random.Random(70 + level)
        rows = [[rng.randrange(m) for _ in range(degree)] for m in bsk]
        out = tool.convert_approximate_bsk_to_q(np.array(rows, dtype=np.uint64))
        for k in range(degree):
            products = [rows[i][k] * pow(b // b_moduli[i], -1, b_moduli[i]) % b_moduli[i] for i in range(level)]
            alpha = sum(products[i] * ((b // b_moduli[i]) % top_msk) for i in range(level)) % top_msk
            alpha = (alpha + m_sk - rows[level][k]) * inverse_b % m_sk
            for row, qi in enumerate(moduli):
                converted = sum(products[i] * ((b // b_moduli[i]) % qi) for i in range(level)) % qi
                adjust = (b % qi) * (m_sk - alpha) % qi if alpha > m_sk >> 1 else (qi - b % qi) % qi * alpha % qi
                assert int(out[row][k]) == (converted + adjust) % qi


def test_convert_approximate_set_valued(oracle):
    # RnsBaseConverterTests.swift:20-65
    degree = 8
    in_moduli = oracle.generate_primes([40, 40, 40], True)
    out_moduli = oracle.generate_primes([45, 45], True)
    in_ctx, out_ctx = oracle.PolyContext(degree, in_moduli), oracle.PolyContext(degree, out_moduli)
    q = _prod(in_moduli)
    rng = random.Random(14)
    xs = [rng.randrange(q) for _ in range(degree)]
    out = oracle.convert_approximate(in_ctx, out_ctx, _poly_from_bigints(xs, in_moduli))
    for k, x in enumerate(xs):
        candidates = [crt_decompose(x + a * q, out_moduli) for a in range(len(in_moduli))]
        assert _column(out, k) in candidates


def test_scale_and_round(oracle):
    # RnsToolTests.swift:20-64
    degree = 8
    moduli = oracle.generate_primes([20, 20, 20], True)
    t = oracle.generate_primes([15], True)[0]
    ctx = oracle.PolyContext(degree, moduli)
    tool = oracle.RnsTool(ctx, t)
    q, k_count = _prod(moduli), len(moduli)
    gamma = (1 << 62) - 40797
    delta = q // t
    v_bound = int(q / t * (0.5 - k_count / gamma) - (q % t) / 2.0)
    rng = random.Random(15)
    for _ in range(10):
        ms = [rng.randrange(t) for _ in range(degree)]
        cts = [(delta * m + rng.randrange(v_bound)) % q for m in ms]
        out = tool.scale_and_round(_poly_from_bigints(cts, moduli), 1)
        assert [int(v) for v in out] == ms


# ------------------------------------------------------------------ BFV semantic checks
@pytest.fixture(scope="module")
def small_bfv(oracle):
    degree = 64
    t = oracle.generate_primes([17], True, degree)[0]
    q = oracle.generate_primes([40, 40, 40, 41], False, degree)  # 3 ciphertext moduli + key-switching modulus
    ctx = oracle.BfvContext(degree, t, q)
    return ctx, BfvClient(oracle, ctx, seed=20)


def test_encrypt_decrypt_roundtrip(oracle, small_bfv):
    ctx, client = small_bfv
    rng = random.Random(21)
    message = [rng.randrange(ctx.t) for _ in range(ctx.degree)]
    ct = client.encrypt(message)
    assert client.decrypt(ct) == message
    assert client.decrypt_exact(ct) == message


def test_bfv_mul_and_relinearize_decrypt_to_product(oracle, small_bfv):
    # HeApiTestUtils.swift:494-556 (coefficient encoding: product is the negacyclic convolution mod t)
    ctx, client = small_bfv
    rng = random.Random(22)
    m1 = [rng.randrange(ctx.t) for _ in range(ctx.degree)]
    m2 = [rng.randrange(ctx.t) for _ in range(ctx.degree)]
    ct1, ct2 = client.encrypt(m1), client.encrypt(m2)
    product = ctx.mul(ct1[None], ct2[None])
    assert product.shape == (1, 3, ctx.L, ctx.degree)
    expected = negacyclic_multiply(m1, m2, ctx.t)
    assert client.decrypt(product[0]) == expected
    key = client.relinearization_key()
    relin = ctx.relinearize(product, key)
    assert relin.shape == (1, 2, ctx.L, ctx.degree)
    assert client.decrypt(relin[0]) == expected
    assert client.decrypt_exact(relin[0]) == expected


def test_bfv_apply_galois_decrypts_to_rotated_plaintext(oracle, small_bfv):
    # HeAPITests / HeApiTestUtils applyGalois checks: decrypt(applyGalois(ct, g)) = m(x^g) mod t (Bfv.swift:174-198)
    ctx, client = small_bfv
    rng = random.Random(26)
    message = [rng.randrange(ctx.t) for _ in range(ctx.degree)]
    ct = client.encrypt(message)
    for element in (3, 2 * ctx.degree - 1, 9):
        key = client.galois_key(element)
        rotated = ctx.apply_galois(ct[None], element, key)
        assert rotated.shape == (1, 2, ctx.L, ctx.degree)
        assert client.decrypt(rotated[0]) == galois_plain(message, element, ctx.t)
    # below top level (after one mod switch) with the same key
    lower = ctx.mod_switch_down(ct[None], poly_count=2)
    key = client.galois_key(3)
    rotated = ctx.apply_galois(lower, 3, key, moduli_count=ctx.L - 1)
    assert client.decrypt(rotated[0], moduli_count=ctx.L - 1) == galois_plain(message, 3, ctx.t)


def test_plaintext_eval_roundtrip_and_multiply(oracle, small_bfv):
    # Plaintext.convertToEvalFormat / convertToCoeffFormat (Plaintext.swift:149-191) and ct x pt
    # (HeApiTestUtils.swift:1223-1285): decrypt(ct * eval(pt)) = negacyclic product mod t
    ctx, client = small_bfv
    rng = random.Random(27)
    m1 = [rng.randrange(ctx.t) for _ in range(ctx.degree)]
    m2 = [rng.randrange(ctx.t) for _ in range(ctx.degree)]
    pt = np.array([m2, m1], dtype=np.uint64)
    for level in (ctx.L, ctx.L - 1):
        pt_eval = ctx.plaintext_to_eval(pt, moduli_count=level)
        assert pt_eval.shape == (2, level, ctx.degree)
        assert np.array_equal(ctx.plaintext_to_coeff(pt_eval, moduli_count=level), pt)
        moduli = ctx.ciphertext_context(level).moduli
        coeff = ctx.ciphertext_context(level).inverse_ntt(pt_eval)
        threshold = (ctx.t + 1) // 2
        for i, q in enumerate(moduli):
            expected_row = [v if v < threshold else v + q - ctx.t for v in m2]
            assert coeff[0, i].tolist() == expected_row
    ct = client.encrypt(m1)
    qctx = ctx.ciphertext_context()
    ct_eval = qctx.forward_ntt(ct)
    product = ctx.mul_plain(ct_eval[None], ctx.plaintext_to_eval(pt[:1]), poly_count=2)[0]
    assert client.decrypt(qctx.inverse_ntt(product)) == negacyclic_multiply(m1, m2, ctx.t)


def test_bfv_mod_switch_down_keeps_message(oracle, small_bfv):
    ctx, client = small_bfv
    rng = random.Random(23)
    message = [rng.randrange(ctx.t) for _ in range(ctx.degree)]
    ct = client.encrypt(message)
    lower = ctx.mod_switch_down(ct[None], poly_count=2)[0]
    assert lower.shape == (2, ctx.L - 1, ctx.degree)
    assert client.decrypt(lower, moduli_count=ctx.L - 1) == message


def test_bfv_inner_product_plain(oracle, small_bfv):
    # HeApiTestUtils.swift:560-720: sum_k ct_k * pt_k decrypts to sum of negacyclic products; nil plaintexts skipped
    ctx, client = small_bfv
    poly_ctx = ctx.ciphertext_context()
    rng = random.Random(24)
    count = 5
    present = [1, 1, 0, 1, 1]
    messages = [[rng.randrange(ctx.t) for _ in range(ctx.degree)] for _ in range(count)]
    plains = [[rng.randrange(ctx.t) for _ in range(ctx.degree)] for _ in range(count)]
    cts = np.stack([poly_ctx.forward_ntt(client.encrypt(m)) for m in messages])
    # Plaintext.convertToEvalFormat (Plaintext.swift:149-170): centered lift of t-residues to Q, then forward NTT
    def lift(p):
        thresh = (ctx.t + 1) // 2
        return poly_ctx.forward_ntt(np.array([[(v if v < thresh else q - (ctx.t - v)) for v in p]
                                              for q in poly_ctx.moduli], dtype=np.uint64))
    pts = np.stack([lift(p) for p in plains])
    out = ctx.inner_product_plain(cts, pts, present)
    expected = [0] * ctx.degree
    for keep, m, p in zip(present, messages, plains):
        if keep:
            prod = negacyclic_multiply(m, p, ctx.t)
            expected = [(a + b) % ctx.t for a, b in zip(expected, prod)]
    assert client.decrypt(poly_ctx.inverse_ntt(out)) == expected
    # single ct*pt (Bfv.mulAssign(ct, pt))
    single = ctx.mul_plain(cts[0][None], pts[0][None], poly_count=2)[0]
    assert client.decrypt(poly_ctx.inverse_ntt(single)) == negacyclic_multiply(messages[0], plains[0], ctx.t)


def test_bfv_inner_product_ct_ct(oracle, small_bfv):
    ctx, client = small_bfv
    rng = random.Random(25)
    count = 3
    m1 = [[rng.randrange(ctx.t) for _ in range(ctx.degree)] for _ in range(count)]
    m2 = [[rng.randrange(ctx.t) for _ in range(ctx.degree)] for _ in range(count)]
    lhs = np.stack([client.encrypt(m) for m in m1])
    rhs = np.stack([client.encrypt(m) for m in m2])
    out = ctx.inner_product(lhs, rhs)
    expected = [0] * ctx.degree
    for a, b in zip(m1, m2):
        expected = [(x + y) % ctx.t for x, y in zip(expected, negacyclic_multiply(a, b, ctx.t))]
    assert client.decrypt(out) == expected
    # one pair == mul
    assert np.array_equal(ctx.inner_product(lhs[:1], rhs[:1]), ctx.mul(lhs[:1], rhs[:1])[0])


def test_bfv_context_errors(oracle):
    degree = 64
    t = oracle.generate_primes([17], True, degree)[0]
    q = oracle.generate_primes([40, 40], False, degree)
    with pytest.raises(oracle.OracleError) as err:
        oracle.BfvContext(degree, t, q + [q[0] + 2])  # not prime / not NTT friendly
    assert err.value.name == "invalidEncryptionParameters"
    with pytest.raises(oracle.OracleError) as err:
        oracle.BfvContext(degree + 1, t, q)
    assert err.value.name == "invalidEncryptionParameters"
    with pytest.raises(oracle.OracleError) as err:
        oracle.BfvContext(degree, q[0], q)  # t must be < every q_i
    assert err.value.name == "invalidEncryptionParameters"


def test_pir_response_selects_the_queried_entry(oracle, small_bfv):
    """PirUtil.computeResponseForOneChunk (PirUtil.swift:408-486) on a 4 x 3 database: one-hot queries per dimension,
    the response (one modulus left) decrypts to the selected plaintext polynomial."""
    ctx, client = small_bfv
    rng = random.Random(28)
    dims = [4, 3]
    entries = [[rng.randrange(ctx.t) for _ in range(ctx.degree)] for _ in range(12)]
    database = ctx.plaintext_to_eval(np.array(entries, dtype=np.uint64))
    present = np.ones(12, dtype=np.uint8)
    present[5] = 0  # a nil plaintext (skipped by innerProduct, Bfv.swift:486-489)
    qctx = ctx.ciphertext_context()
    key = client.relinearization_key()
    one, zero = [1] + [0] * (ctx.degree - 1), [0] * ctx.degree
    for row, column in ((2, 1), (0, 2), (1, 1)):
        dim0 = np.stack([qctx.forward_ntt(client.encrypt(one if k == row else zero)) for k in range(dims[0])])
        rest = np.stack([client.encrypt(one if k == column else zero) for k in range(dims[1])])
        response = oracle.pir.compute_response_for_one_chunk(ctx, dims, dim0, rest, database, present, key)
        assert response.shape == (2, 1, ctx.degree)
        index = column * dims[0] + row
        expected = entries[index] if present[index] else zero
        assert client.decrypt(response, moduli_count=1) == expected
    # one-dimensional database: no ct x ct step, no key needed
    dim0 = np.stack([qctx.forward_ntt(client.encrypt(one if k == 3 else zero)) for k in range(4)])
    response = oracle.pir.compute_response_for_one_chunk(ctx, [4], dim0, None, database[:4], None, None)
    assert client.decrypt(response, moduli_count=1) == entries[3]


def _compressed_query(ctx, total, ones):
    """PirUtil.compressInputsForOneCiphertext (PirUtil.swift:357-377): inverse(2^ceilLog2(total)) at the chosen indices."""
    height = (total - 1).bit_length()
    inverse = pow(pow(2, height, ctx.t), -1, ctx.t)
    message = [0] * ctx.degree
    for index in ones:
        message[index] = inverse
    return message


@pytest.mark.parametrize("total,ones", [(8, [3]), (5, [0, 4]), (1, [0]), (6, [5])])
def test_pir_expand_produces_encrypted_selection_bits(oracle, small_bfv, total, ones):
    """PirUtil.expand (PirUtil.swift:313-355): output i decrypts to the constant polynomial [i in ones]."""
    ctx, client = small_bfv
    n = ctx.degree
    # MulPir's evaluation key holds the elements N/2^k + 1 (IndexPirParameter / MulPir.swift evaluationKeyConfig)
    keys = {(n >> k) + 1: client.galois_key((n >> k) + 1) for k in range(0, (total - 1).bit_length())}
    if not keys:
        keys = {n + 1: client.galois_key(n + 1)}
    query = client.encrypt(_compressed_query(ctx, total, ones))
    expanded = oracle.pir.expand(ctx, query[None], total, keys)
    assert expanded.shape == (total, 2, ctx.L, n)
    for index in range(total):
        want = [1 if index in ones else 0] + [0] * (n - 1)
        assert client.decrypt(expanded[index]) == want, index


@pytest.mark.parametrize("t_bits", [17, 41])  # t^2 below / above 2^64: both division branches of Bfv+Encrypt.swift:92-107
def test_plaintext_translate_is_the_rounded_scaling_and_decrypts_to_the_sum(oracle, t_bits):
    """Bfv.addAssignCoeff / subAssignCoeff(ciphertext, plaintext) (Bfv.swift:110-117, Bfv+Encrypt.swift:75-140) against
    arbitrary-precision integers: c0 +- (floor(Q/t) m + floor(((Q mod t) m + (t + 1) // 2) / t)) mod q_i."""
    degree = 32
    t = oracle.generate_primes([t_bits], True, degree)[0]
    q = oracle.generate_primes([50, 50, 50, 51], False, degree)
    ctx = oracle.BfvContext(degree, t, q)
    client = BfvClient(oracle, ctx, seed=t_bits)
    rng = random.Random(50 + t_bits)
    for level in (None, 2, 1):
        moduli = ctx.ciphertext_context(level).moduli
        big_q = _prod(moduli)
        m1 = [rng.randrange(t) for _ in range(degree)]
        m2 = [rng.choice((0, 1, t - 1, rng.randrange(t))) for _ in range(degree)]
        ct = np.stack([client.encrypt(m1, level), client.encrypt(m1, level)])
        pts = np.array([m2, m1], dtype=np.uint64)
        for subtract in (False, True):
            got = ctx.plaintext_translate(ct, pts, 2, subtract, moduli_count=level)
            for b in range(2):
                for i, qi in enumerate(moduli):
                    for k in range(degree):
                        m = int(pts[b, k])
                        term = ((big_q // t) * m + ((big_q % t) * m + (t + 1) // 2) // t) % qi
                        expected = (int(ct[b, 0, i, k]) + (-term if subtract else term)) % qi
                        assert int(got[b, 0, i, k]) == expected
                assert np.array_equal(got[b, 1], ct[b, 1])  # c1 untouched
            sign = -1 if subtract else 1
            assert client.decrypt(got[0], level) == [(a + sign * b) % t for a, b in zip(m1, m2)]
    # encrypt is a zero encryption translated by the message (Bfv+Encrypt.swift:66-73)
    zero = client.encrypt([0] * degree)
    assert client.decrypt(ctx.plaintext_translate(zero[None], np.array([m1], dtype=np.uint64))[0]) == m1


@pytest.fixture(scope="module")
def small_bfv32(oracle):
    """Context<Bfv<UInt32>>-shaped parameters: 27-28-bit moduli as in n_4096_logq_27_28_28
    (EncryptionParameters.swift:313-378), scaled down to N = 64."""
    degree = 64
    t = oracle.generate_primes([10], True, degree, word_bits=32)[0]
    q = oracle.generate_primes([27, 28, 28, 29], False, degree, word_bits=32)
    ctx = oracle.BfvContext(degree, t, q, word_bits=32)
    return ctx, BfvClient(oracle, ctx, seed=33)


def test_bfv_uint32_constants_and_decrypt(oracle, small_bfv32):
    """The UInt32 word type changes the BEHZ constants (gamma 2^30-20405, mTilde 2^16, 29-bit Bsk); the same
    semantic checks must hold (HeAPITests run every test for Bfv<UInt32> and Bfv<UInt64>)."""
    ctx, client = small_bfv32
    bsk = ctx.rns_tool().bsk
    assert len(bsk) == ctx.L + 1 and all((1 << 28) < b < (1 << 29) for b in bsk) and bsk == sorted(bsk)
    rng = random.Random(34)
    m1 = [rng.randrange(ctx.t) for _ in range(ctx.degree)]
    m2 = [rng.randrange(ctx.t) for _ in range(ctx.degree)]
    ct1, ct2 = client.encrypt(m1), client.encrypt(m2)
    assert client.decrypt(ct1) == m1 and client.decrypt_exact(ct1) == m1
    product = ctx.mul(ct1[None], ct2[None])
    expected = negacyclic_multiply(m1, m2, ctx.t)
    assert client.decrypt(product[0]) == expected
    relin = ctx.relinearize(product, client.relinearization_key())
    assert client.decrypt(relin[0]) == expected and client.decrypt_exact(relin[0]) == expected
    lower = ctx.mod_switch_down(relin, poly_count=2)
    assert client.decrypt(lower[0], moduli_count=ctx.L - 1) == expected
    with pytest.raises(oracle.OracleError):  # a 31-bit modulus does not fit UInt32's Modulus (max 2^30 - 1)
        oracle.BfvContext(ctx.degree, ctx.t, oracle.generate_primes([31, 31], False, ctx.degree), word_bits=32)
