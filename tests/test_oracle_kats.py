"""Pins the CPU oracle against every known-answer vector the reference's tests hold for the hot path.

Mirrors Tests/HomomorphicEncryptionTests/{NttTests,ScalarTests,PolyRqTests/*}.swift (file:line in the golden file).
"""
import random

import numpy as np
import pytest


def test_is_primitive_root_of_unity(oracle, kats):
    for c in kats["is_primitive_root_of_unity"]["cases"]:
        assert oracle.is_primitive_root_of_unity(c["root"], c["degree"], c["modulus"]) == c["expected"], c


def test_min_primitive_root_of_unity(oracle, kats):
    for c in kats["min_primitive_root_of_unity"]["cases"]:
        assert oracle.min_primitive_root_of_unity(c["modulus"], c["degree"]) == c["expected"], c


def test_ntt_known_answers(oracle, kats):
    for c in kats["ntt"]["cases"]:
        coeff = np.array(c["coeff"], dtype=np.uint64)
        evals = np.array(c["eval"], dtype=np.uint64)
        ctx = oracle.PolyContext(coeff.shape[1], c["moduli"])
        assert np.array_equal(ctx.forward_ntt(coeff), evals), c
        assert np.array_equal(ctx.inverse_ntt(evals), coeff), c


def test_ntt_delta_and_zero(oracle, kats):
    for c in kats["ntt"]["delta_cases"]:
        n = c["degree"]
        ctx = oracle.PolyContext(n, [c["modulus"]])
        zeros = np.zeros((1, n), dtype=np.uint64)
        one_hot = zeros.copy()
        one_hot[0, 0] = 1
        ones = np.ones((1, n), dtype=np.uint64)
        assert np.array_equal(ctx.forward_ntt(zeros), zeros)
        assert np.array_equal(ctx.inverse_ntt(zeros), zeros)
        assert np.array_equal(ctx.forward_ntt(one_hot), ones)
        assert np.array_equal(ctx.inverse_ntt(ones), one_hot)


def test_ntt_roundtrip_large_moduli(oracle):
    # NttTests.swift:193-206: N=256, moduli = generatePrimes([60, 62], preferringSmall: false, nttDegree: 256)
    degree = 256
    moduli = oracle.generate_primes([60, 62], False, degree)
    ctx = oracle.PolyContext(degree, moduli)
    rng = random.Random(1)
    coeff = np.array([[rng.randrange(q) for _ in range(degree)] for q in moduli], dtype=np.uint64)
    evals = ctx.forward_ntt(coeff)
    assert not np.array_equal(evals, coeff)
    for row, q in zip(evals, moduli):
        assert int(row.max()) < q
    assert np.array_equal(ctx.inverse_ntt(evals), coeff)


def _naive_negacyclic(x, y, p):
    n = len(x)
    out = [0] * n
    for i in range(n):
        acc = 0
        for j in range(i + 1):
            acc += x[j] * y[i - j]
        for j in range(i + 1, n):
            acc -= x[j] * y[n + i - j]
        out[i] = acc % p
    return out


def test_ntt_matches_naive_multiplication(oracle):
    # NttTests.swift:208-250: N=128, one 30-bit modulus
    degree = 128
    moduli = oracle.generate_primes([30], False, degree, word_bits=32)
    p = moduli[0]
    ctx = oracle.PolyContext(degree, moduli)
    rng = random.Random(2)
    x = [rng.randrange(p) for _ in range(degree)]
    y = [rng.randrange(p) for _ in range(degree)]
    xe = ctx.forward_ntt(np.array([x], dtype=np.uint64))
    ye = ctx.forward_ntt(np.array([y], dtype=np.uint64))
    prod = ctx.inverse_ntt(ctx.mul(xe, ye))
    assert [int(v) for v in prod[0]] == _naive_negacyclic(x, y, p)


def test_ntt_is_evaluation_at_odd_powers_of_min_root(oracle):
    """Independent definition check: out[bitrev(i)] = poly(psi^(2i+1)) with psi the minimal primitive 2N-th root."""
    degree = 16
    p = oracle.generate_primes([40], False, degree)[0]
    psi = oracle.min_primitive_root_of_unity(p, 2 * degree)
    rng = random.Random(3)
    x = [rng.randrange(p) for _ in range(degree)]
    ctx = oracle.PolyContext(degree, [p])
    got = [int(v) for v in ctx.forward_ntt(np.array([x], dtype=np.uint64))[0]]
    for i in range(degree):
        point = pow(psi, 2 * oracle.reverse_bits(i, 4) + 1, p)
        assert got[i] == sum(c * pow(point, k, p) for k, c in enumerate(x)) % p


def test_generate_primes(oracle, kats):
    g = kats["generate_primes"]
    for c in g["cases"]:
        got = oracle.generate_primes(c["bits"], c["preferring_small"], c["ntt_degree"], c["word_bits"])
        assert got == c["expected"], c
    for c in g["error_cases"]:
        with pytest.raises(oracle.OracleError) as err:
            oracle.generate_primes(c["bits"], c["preferring_small"], c["ntt_degree"], c["word_bits"])
        assert err.value.name == c["error"]
    assert sum(oracle.is_prime(v) for v in range(1, 1000)) == g["primes_below_1000"]


def test_barrett_and_shoup_against_bigint(oracle):
    # ScalarTests.swift:212-226,256-405 (randomised vs % and /)
    rng = random.Random(4)
    moduli = [2, 3, 5, 97, (1 << 32), (1 << 32) + 15, (1 << 55) - 311295, (1 << 61) - 1, (1 << 62) - 57]
    for p in moduli:
        for _ in range(200):
            x = rng.randrange(1 << 64)
            assert oracle.barrett_reduce_u64(p, x) == x % p
            w = rng.randrange(1 << 128)
            assert oracle.barrett_reduce_u128(p, w) == w % p
            a, b = rng.randrange(p), rng.randrange(p)
            assert oracle.barrett_reduce_product(p, a, b) == (a * b) % p
            c = rng.randrange(p)
            assert oracle.shoup_factor(c, p) == (c << 64) // p
            lazy = oracle.shoup_multiply_mod_lazy(c, p, x)
            assert lazy < 2 * p and lazy % p == (c * x) % p
            assert oracle.shoup_multiply_mod(c, p, x) == (c * x) % p


def test_inverse_mod(oracle):
    rng = random.Random(5)
    for p in [97, (1 << 55) - 311295, (1 << 61) - 1]:
        for _ in range(50):
            x = rng.randrange(1, p)
            assert (oracle.inverse_mod(x, p) * x) % p == 1
    with pytest.raises(oracle.OracleError):
        oracle.inverse_mod(6, 9)


def test_poly_ops_known_answers(oracle, kats):
    k = kats["poly_ops"]
    ctx = oracle.PolyContext(k["degree"], k["moduli"])
    x = np.array(k["x"], dtype=np.uint64)
    zero = np.zeros_like(x)
    assert np.array_equal(ctx.add(x, x), np.array(k["add_x_x"], dtype=np.uint64))
    assert np.array_equal(ctx.sub(zero, x), np.array(k["zero_minus_x"], dtype=np.uint64))
    assert np.array_equal(ctx.neg(x), np.array(k["neg_x"], dtype=np.uint64))
    y = np.array(k["mul_y"], dtype=np.uint64)
    assert np.array_equal(ctx.mul(x, y), np.array(k["mul_x_y"], dtype=np.uint64))
    s = k["scalar"]
    expected = np.array([[(int(v) * s) % q for v in row] for row, q in zip(k["x"], k["moduli"])], dtype=np.uint64)
    assert np.array_equal(ctx.mul_scalar(x, [s % q for q in k["moduli"]]), expected)


def test_divide_and_round_q_last_known_answers(oracle, kats):
    for c in kats["divide_and_round_q_last"]["cases"]:
        ctx = oracle.PolyContext(c["degree"], c["moduli"])
        got = ctx.divide_and_round_q_last(np.array(c["x"], dtype=np.uint64))
        assert np.array_equal(got[0], np.array(c["expected"], dtype=np.uint64)), c


def test_divide_and_round_q_last_matches_bigint(oracle):
    degree = 8
    moduli = oracle.generate_primes([40, 45, 50], False, degree)
    ctx = oracle.PolyContext(degree, moduli)
    q = moduli[0] * moduli[1] * moduli[2]
    q_last = moduli[-1]
    rng = random.Random(6)
    xs = [rng.randrange(q) for _ in range(degree)]
    data = np.array([[x % m for x in xs] for m in moduli], dtype=np.uint64)
    got = ctx.divide_and_round_q_last(data)[0]
    for k, x in enumerate(xs):
        rounded = (x + q_last // 2) // q_last
        for i, m in enumerate(moduli[:-1]):
            assert int(got[i, k]) == rounded % m


def test_poly_context_errors(oracle, kats):
    for c in kats["poly_context_errors"]["cases"]:
        with pytest.raises(oracle.OracleError) as err:
            oracle.PolyContext(c["degree"], c["moduli"])
        assert err.value.name == c["error"], c


def test_q_remainder(oracle, kats):
    for c in kats["q_remainder"]["cases"]:
        ctx = oracle.PolyContext(c["degree"], c["moduli"])
        assert ctx.q_remainder(c["dividing_by"]) == c["expected"], c


def test_max_lazy_product_accumulation_count(oracle, kats):
    for c in kats["max_lazy_product_accumulation_count"]["cases"]:
        ctx = oracle.PolyContext(c["degree"], c["moduli"])
        assert ctx.max_lazy_product_accumulation_count(c["word_bits"]) == c["expected"], c


def test_lazy_product_accumulation(oracle):
    degree = 8
    moduli = oracle.generate_primes([59, 60], False, degree)
    ctx = oracle.PolyContext(degree, moduli)
    rng = random.Random(7)
    acc = np.zeros((2, degree, 2), dtype=np.uint64)
    exact = [[0] * degree for _ in moduli]
    for _ in range(20):
        x = np.array([[rng.randrange(q) for _ in range(degree)] for q in moduli], dtype=np.uint64)
        y = np.array([[rng.randrange(q) for _ in range(degree)] for q in moduli], dtype=np.uint64)
        ctx.adding_lazy_product(x, y, acc)
        for i in range(2):
            for k in range(degree):
                exact[i][k] += int(x[i, k]) * int(y[i, k])
    got = ctx.reduce_accumulator(acc)
    for i, q in enumerate(moduli):
        assert [int(v) for v in got[i]] == [e % q for e in exact[i]]


def test_ntt_tables_layout(oracle):
    """Table order is part of the contract the HIP kernels consume (PolyRq+Ntt.swift:125-157)."""
    degree = 8
    p = oracle.generate_primes([30], False, degree)[0]
    ctx = oracle.PolyContext(degree, [p])
    psi = oracle.min_primitive_root_of_unity(p, 2 * degree)
    inv_psi = pow(psi, p - 2, p)
    t = ctx.ntt_tables(0)
    for i in range(degree):
        assert int(t["root_powers"][oracle.reverse_bits(i, 3)]) == pow(psi, i, p)
        assert int(t["root_factors"][i]) == (int(t["root_powers"][i]) << 64) // p
    inv_bitrev = [pow(inv_psi, oracle.reverse_bits(i, 3), p) for i in range(degree)]
    expected = [1] + inv_bitrev[4:8] + inv_bitrev[2:4] + inv_bitrev[1:2]
    # reordered = concat over m = N/2, N/4, ..., 1 of inversePowers[m..<2m], starting at index 1
    inv_powers = [0] * degree
    for i in range(degree):
        inv_powers[oracle.reverse_bits(i, 3)] = pow(inv_psi, i, p)
    expected = [1] + inv_powers[4:8] + inv_powers[2:4] + inv_powers[1:2]
    assert [int(v) for v in t["inv_root_powers"]] == expected
    assert t["inverse_degree"] == pow(degree, p - 2, p)
    assert t["inverse_degree_root"] == (t["inverse_degree"] * expected[degree - 1]) % p


def test_apply_galois_known_answers(oracle, kats):
    """GaloisTests.swift:20-86: the three KAT polynomials, Coeff and Eval forms, and the commuting property."""
    for case in kats["apply_galois"]["cases"]:
        degree, moduli = case["degree"], case["moduli"]
        ctx = oracle.PolyContext(degree, moduli)
        data = np.array(case["data"], dtype=np.uint64).reshape(1, len(moduli), degree)
        expected = np.array(case["expected"], dtype=np.uint64).reshape(1, len(moduli), degree)
        assert np.array_equal(ctx.apply_galois(data, case["element"]), expected)
        via_eval = ctx.inverse_ntt(ctx.apply_galois(ctx.forward_ntt(data), case["element"], eval_format=True))
        assert np.array_equal(via_eval, expected)
        for index in range(1, degree):
            element = 2 * index + 1
            assert np.array_equal(ctx.forward_ntt(ctx.apply_galois(data, element)),
                                  ctx.apply_galois(ctx.forward_ntt(data), element, eval_format=True))
        # swapping rows twice is the identity (GaloisElement.swappingRows = 2N - 1, Galois.swift)
        swap = 2 * degree - 1
        assert np.array_equal(ctx.apply_galois(ctx.apply_galois(data, swap), swap), data)


def test_multiply_power_of_x_is_negacyclic_shift(oracle):
    """PolyRq.multiplyPowerOfX (PolyRq.swift:398-422) restated literally must equal x^power mod (x^N + 1)."""
    degree, moduli = 16, [97, 193]
    ctx = oracle.PolyContext(degree, moduli)
    rng = np.random.default_rng(11)
    data = np.stack([rng.integers(0, q, size=(2, degree), dtype=np.uint64) for q in moduli], axis=1)
    for power in range(-3 * degree, 3 * degree + 1):
        expected = np.zeros_like(data)
        for r, q in enumerate(moduli):
            for i in range(degree):
                j = (i + power) % (2 * degree)
                column, negate = (j - degree, True) if j >= degree else (j, False)
                expected[:, r, column] = (q - data[:, r, i]) % q if negate else data[:, r, i]
        assert np.array_equal(ctx.multiply_power_of_x(data, power), expected), power


def test_coefficient_packing_known_answers(oracle, kats):
    """CoefficientPackingTests.swift:83-212."""
    for case in kats["coefficient_packing"]["bytes_to_coeffs"]:
        got = oracle.bytes_to_coefficients(case["bytes"], case["bits"], case["decode"], case["skip"])
        assert got.tolist() == case["expected"]
    for case in kats["coefficient_packing"]["coeffs_to_bytes"]:
        assert oracle.coefficients_to_bytes(case["coeffs"], case["bits"], case["skip"]).tolist() == case["expected"]


def test_coefficient_packing_roundtrips(oracle):
    """CoefficientPackingTests.swift:23-81 (bytesRoundtrip, coeffsRoundtrip) for every width."""
    rng = np.random.default_rng(5)
    for log2t in range(1, 61):
        data = rng.integers(0, 256, size=512, dtype=np.uint8)
        coeffs = oracle.bytes_to_coefficients(data, log2t, False)
        assert all(int(c) < (1 << log2t) + 1 for c in coeffs)
        assert oracle.coefficients_to_bytes(coeffs, log2t)[: len(data)].tolist() == data.tolist()
        values = rng.integers(0, (1 << log2t) + 1, size=512, dtype=np.uint64)
        packed = oracle.coefficients_to_bytes(values, log2t + 1)
        assert oracle.bytes_to_coefficients(packed, log2t + 1, True)[: len(values)].tolist() == values.tolist()


def test_poly_serialize_known_answers(oracle, kats):
    """PolyRq+SerializeTests.swift:39-103."""
    for case in kats["poly_serialize"]["roundtrip"]:
        moduli = case["moduli"]
        degree = len(case["poly"]) // len(moduli)
        ctx = oracle.PolyContext(degree, moduli)
        poly = np.array(case["poly"], dtype=np.uint64).reshape(1, len(moduli), degree)
        packed = ctx.serialize(poly, case["skip"])
        assert packed.shape[1] == ctx.serialization_byte_count(case["skip"])
        assert ctx.deserialize(packed, case["skip"]).ravel().tolist() == case["expected"]
    moduli = oracle.generate_primes([14, 16, 21, 22, 27], False, 1)
    ctx = oracle.PolyContext(32, moduli)
    rng = np.random.default_rng(6)
    poly = np.stack([rng.integers(0, q, size=(3, 32), dtype=np.uint64) for q in moduli], axis=1)
    assert np.array_equal(ctx.deserialize(ctx.serialize(poly)), poly)
    # a buffer serialized for one context does not fit a wider one (serializedBufferSizeMismatch, :21-36)
    narrow = oracle.PolyContext(32, oracle.generate_primes([5, 5, 5], False, 1, word_bits=32))
    wide = oracle.PolyContext(32, oracle.generate_primes([5, 5, 16], False, 1, word_bits=32))
    assert wide.serialization_byte_count() == 104
    packed = narrow.serialize(np.zeros((1, 3, 32), dtype=np.uint64))
    with pytest.raises(oracle.OracleError):
        wide.deserialize(packed)


def test_nist_ctr_drbg_vectors(oracle, kats):
    """NistCtrDrbgTests.swift:21-161."""
    block = kats["nist_ctr_drbg"]
    trace = block["state_trace"]
    drbg = oracle.CtrDrbg(bytes.fromhex(trace["entropy"]))
    assert [x.hex() for x in drbg.state()] == trace["after_init"]
    drbg.generate(64)
    assert [x.hex() for x in drbg.state()] == trace["after_first_generate"]
    drbg.generate(64)
    assert [x.hex() for x in drbg.state()] == trace["after_second_generate"]
    assert len(block["vectors"]) >= 10
    for vector in block["vectors"]:
        drbg = oracle.CtrDrbg(bytes.fromhex(vector["entropy"]))
        expected = bytes.fromhex(vector["returned_bits"])
        drbg.generate(len(expected))
        assert drbg.generate(len(expected)) == expected


def test_seeded_polynomial_is_the_buffered_stream_reduced(oracle):
    """PolyRq.randomizeUniform(using:) (PolyRq+Randomize.swift:56-75) over NistAes128Ctr (4096-byte refills):
    coefficient (i, k) = LE u128 at stream offset 16 (i N + k), reduced mod q_i."""
    degree, moduli = 512, oracle.generate_primes([55, 40, 20], False, 512)
    ctx = oracle.PolyContext(degree, moduli)
    seed = bytes(range(32))
    poly = ctx.random_from_seeds(np.frombuffer(seed, dtype=np.uint8))[0]
    drbg = oracle.CtrDrbg(seed)
    stream = b"".join(drbg.generate(4096) for _ in range(len(moduli) * degree * 16 // 4096))
    for i, q in enumerate(moduli):
        for k in (0, 1, 255, 256, degree - 1):
            offset = 16 * (i * degree + k)
            assert int(poly[i, k]) == int.from_bytes(stream[offset:offset + 16], "little") % q
