"""GPU parity tests of the "next" rows: Galois automorphisms, x^k multiplication, the
Galois key switch on ciphertexts and plaintext <-> Eval conversion.  Bit-exact against the CPU oracle, pinned by
the reference's KATs (Tests/HomomorphicEncryptionTests/PolyRqTests/GaloisTests.swift:20-86) and by decryption."""
import random

import numpy as np
import pytest

import heamd
from bfv_helpers import BfvClient, galois_plain, negacyclic_multiply

pytestmark = pytest.mark.gpu


def _uniform(rng, shape_prefix, moduli, degree):
    rows = [rng.integers(0, q, size=tuple(shape_prefix) + (degree,), dtype=np.uint64) for q in moduli]
    return np.ascontiguousarray(np.stack(rows, axis=len(shape_prefix)))


def test_apply_galois_known_answers(kats):
    for case in kats["apply_galois"]["cases"]:
        degree, moduli = case["degree"], case["moduli"]
        ctx = heamd.PolyContext(degree, moduli)
        data = np.array(case["data"], dtype=np.uint64).reshape(1, len(moduli), degree)
        expected = np.array(case["expected"], dtype=np.uint64).reshape(1, len(moduli), degree)
        got = heamd.to_host(ctx.apply_galois(heamd.to_device(data), case["element"]))
        assert np.array_equal(got, expected)
        evaluated = ctx.forward_ntt_(heamd.to_device(data))
        rotated = ctx.apply_galois(evaluated, case["element"], eval_format=True)
        assert np.array_equal(heamd.to_host(ctx.inverse_ntt_(rotated)), expected)


@pytest.mark.parametrize("degree,bits,batch", [(8, [20], 3), (64, [40, 41], 4), (4096, [55, 55], 3),
                                                (8192, [55, 55, 55, 55], 2), (16384, [55, 61, 45], 1)])
def test_apply_galois_matches_oracle(oracle, degree, bits, batch):
    moduli = oracle.generate_primes(bits, False, degree)
    ours, ref = heamd.PolyContext(degree, moduli), oracle.PolyContext(degree, moduli)
    rng = np.random.default_rng(degree + batch)
    slab = _uniform(rng, (batch,), moduli, degree)
    slab[0, :, 0] = 0
    slab[0, :, 1] = [m - 1 for m in moduli]
    elements = {3, 5, 2 * degree - 1, degree + 1, degree - 1} | {int(2 * rng.integers(1, degree) + 1) for _ in range(3)}
    for element in sorted(e for e in elements if 1 < e < 2 * degree):
        for eval_format in (False, True):
            got = heamd.to_host(ours.apply_galois(heamd.to_device(slab), element, eval_format=eval_format))
            assert np.array_equal(got, ref.apply_galois(slab, element, eval_format=eval_format)), (element, eval_format)
    # the two forms commute with the NTT (GaloisTests.swift:66-70)
    element = 2 * (degree // 4) + 1
    via_coeff = ours.forward_ntt_(ours.apply_galois(heamd.to_device(slab), element))
    via_eval = ours.apply_galois(ours.forward_ntt_(heamd.to_device(slab)), element, eval_format=True)
    assert np.array_equal(heamd.to_host(via_coeff), heamd.to_host(via_eval))


def test_apply_galois_rejects_bad_elements():
    ctx = heamd.PolyContext(16, [97])
    slab = heamd.to_device(np.zeros((1, 1, 16), dtype=np.uint64))
    for element in (0, 1, 2, 32, 33):
        with pytest.raises(heamd.HeError) as err:
            ctx.apply_galois(slab, element)
        assert err.value.name == "invalidArgument"


@pytest.mark.parametrize("degree,bits", [(16, [20, 21]), (4096, [55, 55]), (8192, [55, 40])])
def test_multiply_power_of_x_matches_oracle(oracle, degree, bits):
    moduli = oracle.generate_primes(bits, False, degree)
    ours, ref = heamd.PolyContext(degree, moduli), oracle.PolyContext(degree, moduli)
    rng = np.random.default_rng(degree)
    slab = _uniform(rng, (3,), moduli, degree)
    powers = [0, 1, -1, degree - 1, degree, degree + 1, 2 * degree - 1, 2 * degree, -degree, -2 * degree + 1,
              5 * degree + 3, -7 * degree - 2]
    for power in powers:
        got = heamd.to_host(ours.multiply_power_of_x(heamd.to_device(slab), power))
        assert np.array_equal(got, ref.multiply_power_of_x(slab, power)), power


@pytest.fixture(scope="module")
def small(oracle):
    degree = 64
    t = oracle.generate_primes([17], True, degree)[0]
    q = oracle.generate_primes([40, 40, 40, 41], False, degree)
    ref = oracle.BfvContext(degree, t, q)
    return heamd.BfvContext(degree, t, q), ref, BfvClient(oracle, ref, seed=50)


def test_bfv_apply_galois_matches_oracle_and_decrypts(oracle, small):
    ours, ref, client = small
    rng = random.Random(51)
    messages = [[rng.randrange(ref.t) for _ in range(ref.degree)] for _ in range(3)]
    cts = np.stack([client.encrypt(m) for m in messages])
    for element in (3, 2 * ref.degree - 1, 25):
        key = client.galois_key(element)
        got = heamd.to_host(ours.apply_galois(heamd.to_device(cts), element, heamd.to_device(key)))
        assert np.array_equal(got, ref.apply_galois(cts, element, key))
        for ct, message in zip(got, messages):
            assert client.decrypt(ct) == galois_plain(message, element, ref.t)
    lower = ref.mod_switch_down(cts, poly_count=2)
    key = client.galois_key(3)
    got = heamd.to_host(ours.apply_galois(heamd.to_device(lower), 3, heamd.to_device(key), moduli_count=ref.L - 1))
    assert np.array_equal(got, ref.apply_galois(lower, 3, key, moduli_count=ref.L - 1))
    assert client.decrypt(got[0], moduli_count=ref.L - 1) == galois_plain(messages[0], 3, ref.t)


def test_bfv_apply_galois_errors(small):
    ours, ref, _ = small
    ct = heamd.to_device(np.zeros((1, 2, ours.L, ours.degree), dtype=np.uint64))
    with pytest.raises(heamd.HeError) as err:
        ours.apply_galois(ct, 3, None)
    assert err.value.name == "missingGaloisKey"
    key = heamd.to_device(np.zeros((ours.L, 2, ours.L + 1, ours.degree), dtype=np.uint64))
    with pytest.raises(heamd.HeError) as err:
        ours.apply_galois(ct, 4, key)
    assert err.value.name == "invalidArgument"


def test_bfv_apply_galois_config3_shape(oracle):
    """N=8192, 4+1 55-bit moduli, uniform words (exact ciphertext words are what parity means here)."""
    degree = 8192
    q = oracle.generate_primes([55] * 5, False, degree)
    ours, ref = heamd.BfvContext(degree, 557057, q), oracle.BfvContext(degree, 557057, q)
    rng = np.random.default_rng(52)
    cts = _uniform(rng, (3, 2), q[:-1], degree)
    key = _uniform(rng, (ours.L, 2), q, degree)
    element = 2 * 1234 + 1
    workspace = None
    got = heamd.to_host(ours.apply_galois(heamd.to_device(cts), element, heamd.to_device(key), workspace=workspace))
    assert np.array_equal(got, ref.apply_galois(cts, element, key))


@pytest.mark.parametrize("degree,bits", [(4096, [50, 55, 45, 55]), (8192, [55, 41, 55]), (16384, [55, 55, 48, 55])])
def test_bfv_apply_galois_fused_path(oracle, degree, bits):
    """Degrees with a tiled transform take Bfv.applyGalois without a rotated copy (the automorphism rides the
    decomposition's loads and the last kernel's c0 term): every word equals the oracle's for elements that only flip
    signs (N + 1), reverse (2N - 1) and scatter (3, N/2 + 1), with ciphertext moduli above and below the key-switching
    ones; the in-place call (out = ct, which keeps the rotated copy) gives the same words."""
    import ctypes

    q = oracle.generate_primes(bits, False, degree)
    ours, ref = heamd.BfvContext(degree, 65537, q), oracle.BfvContext(degree, 65537, q)
    rng = np.random.default_rng(degree)
    cts = _uniform(rng, (3, 2), q[:-1], degree)
    cts[0, :, :, :5] = 0
    key = _uniform(rng, (ours.L, 2), q, degree)
    device_key = heamd.to_device(key)
    for element in (degree + 1, 2 * degree - 1, 3, degree // 2 + 1):
        expected = ref.apply_galois(cts, element, key)
        got = heamd.to_host(ours.apply_galois(heamd.to_device(cts), element, device_key))
        assert np.array_equal(got, expected), element
    in_place = heamd.to_device(cts)
    ptr = ctypes.c_void_p(in_place.data_ptr())
    assert heamd.load_library().he_bfv_apply_galois_device(ours.h, ours.L, ptr, 3, ctypes.c_void_p(device_key.data_ptr()), ptr,
                                                           3, None, 0, None) == 0
    assert np.array_equal(heamd.to_host(in_place), ref.apply_galois(cts, 3, key))


def test_plaintext_conversions_match_oracle(oracle, small):
    ours, ref, client = small
    rng = np.random.default_rng(53)
    pt = rng.integers(0, ref.t, size=(5, ref.degree), dtype=np.uint64)
    pt[0, :4] = [0, ref.t - 1, (ref.t + 1) // 2, (ref.t + 1) // 2 - 1]
    for level in (ref.L, ref.L - 1, 1):
        got = heamd.to_host(ours.plaintext_to_eval(heamd.to_device(pt), moduli_count=level))
        expected = ref.plaintext_to_eval(pt, moduli_count=level)
        assert np.array_equal(got, expected)
        back = heamd.to_host(ours.plaintext_to_coeff(heamd.to_device(expected), moduli_count=level))
        assert np.array_equal(back, pt)
    # ct x pt with the device-made Eval plaintext decrypts to the negacyclic product (HeApiTestUtils.swift:1223-1285)
    r = random.Random(54)
    m1 = [r.randrange(ref.t) for _ in range(ref.degree)]
    ct = client.encrypt(m1)
    qctx = ours.ciphertext_context()
    ct_eval = qctx.forward_ntt_(heamd.to_device(ct[None]))
    pt_eval = ours.plaintext_to_eval(heamd.to_device(pt[:1]))
    product = qctx.inverse_ntt_(ours.mul_plain_(ct_eval, pt_eval, poly_count=2))
    assert client.decrypt(heamd.to_host(product)[0]) == negacyclic_multiply(m1, [int(v) for v in pt[0]], ref.t)


def test_plaintext_to_eval_pir_shape(oracle):
    """Database preprocessing shape (MulPir.swift:507-556): N=8192, L=4, a slab of plaintexts."""
    degree = 8192
    q = oracle.generate_primes([55] * 5, False, degree)
    ours, ref = heamd.BfvContext(degree, 557057, q), oracle.BfvContext(degree, 557057, q)
    rng = np.random.default_rng(55)
    pt = rng.integers(0, 557057, size=(6, degree), dtype=np.uint64)
    got = heamd.to_host(ours.plaintext_to_eval(heamd.to_device(pt)))
    assert np.array_equal(got, ref.plaintext_to_eval(pt))
    assert np.array_equal(heamd.to_host(ours.plaintext_to_coeff(heamd.to_device(got))), pt)


@pytest.mark.parametrize("level", [None, 2, 1])
def test_scale_and_round_matches_oracle_and_decrypts(oracle, small, level):
    """_RnsTool.scaleAndRound (RnsTool.swift:272-302) on the device: word-exact vs the oracle on uniform inputs, and
    the decrypt path c0 + c1 s -> scaleAndRound recovers the message (Bfv+Decrypt.swift:29-40,188-204)."""
    ours, ref, client = small
    L = ref.L if level is None else level
    moduli = ref.ciphertext_context(L).moduli
    tool = ref.rns_tool(L)
    rng = np.random.default_rng(70 + L)
    x = _uniform(rng, (6,), moduli, ref.degree)
    x[0, :, 0] = 0
    x[0, :, 1] = [m - 1 for m in moduli]
    for scaling in (1, 2, ref.t - 1):
        got = heamd.to_host(ours.scale_and_round(heamd.to_device(x), scaling, moduli_count=L))
        expected = np.stack([tool.scale_and_round(p, scaling) for p in x])
        assert np.array_equal(got, expected), scaling
    r = random.Random(71)
    message = [r.randrange(ref.t) for _ in range(ref.degree)]
    ct = client.encrypt(message, moduli_count=L)
    qctx = ref.ciphertext_context(L)
    s = client._s_eval_for(qctx)
    ct_eval = qctx.forward_ntt(ct)
    dot = qctx.inverse_ntt(qctx.add(ct_eval[0], qctx.mul(ct_eval[1], s)))
    decrypted = heamd.to_host(ours.scale_and_round(heamd.to_device(dot[None]), 1, moduli_count=L))[0]
    assert [int(v) for v in decrypted] == message


def test_scale_and_round_config3_shape(oracle):
    degree = 8192
    q = oracle.generate_primes([55] * 5, False, degree)
    ours, ref = heamd.BfvContext(degree, 557057, q), oracle.BfvContext(degree, 557057, q)
    rng = np.random.default_rng(72)
    x = _uniform(rng, (3,), q[:-1], degree)
    tool = ref.rns_tool()
    got = heamd.to_host(ours.scale_and_round(heamd.to_device(x), 1))
    assert np.array_equal(got, np.stack([tool.scale_and_round(p, 1) for p in x]))
    with pytest.raises(heamd.HeError) as err:
        ours.scale_and_round(heamd.to_device(x), 557057)
    assert err.value.name == "invalidArgument"


def test_serialize_known_answers(kats):
    """PolyRq+SerializeTests.swift:57-103 and the CoefficientPacking KATs (:148-212) through one-row contexts."""
    import torch

    for case in kats["poly_serialize"]["roundtrip"]:
        moduli = case["moduli"]
        degree = len(case["poly"]) // len(moduli)
        ctx = heamd.PolyContext(degree, moduli)
        poly = np.array(case["poly"], dtype=np.uint64).reshape(1, len(moduli), degree)
        packed = ctx.serialize(heamd.to_device(poly), case["skip"])
        back = heamd.to_host(ctx.deserialize(packed, case["skip"]))
        assert back.ravel().tolist() == case["expected"]
    # coefficientsToBytes KATs with a power-of-two "modulus" row: ceilLog2(2^bits) = bits
    for case in kats["coefficient_packing"]["coeffs_to_bytes"]:
        count = len(case["coeffs"])
        if count & (count - 1):
            continue  # a PolyContext needs a power-of-two degree; the other KATs are covered by the oracle comparison
        ctx = heamd.PolyContext(count, [1 << case["bits"]])
        packed = ctx.serialize(heamd.to_device(np.array(case["coeffs"], dtype=np.uint64).reshape(1, 1, count)),
                               case["skip"])
        assert packed.cpu().numpy().ravel().tolist() == case["expected"]
    assert torch.cuda.is_available()


@pytest.mark.parametrize("degree,bits", [(32, [14, 16, 21, 22, 27]), (8192, [55, 55, 55, 55]), (4096, [27, 28, 28]),
                                         (64, [62, 33, 8]), (128, [62, 33, 8]), (256, [9, 17, 40, 62]),
                                         (1024, [62, 61, 50])])
def test_serialize_matches_oracle(oracle, degree, bits):
    import torch

    moduli = oracle.generate_primes(bits, False, 1)
    ours, ref = heamd.PolyContext(degree, moduli), oracle.PolyContext(degree, moduli)
    rng = np.random.default_rng(degree + len(bits))
    slab = _uniform(rng, (3,), moduli, degree)
    slab[0, :, 0] = 0
    slab[0, :, -1] = [m - 1 for m in moduli]
    for skip in (0, 1, 5):
        assert ours.serialization_byte_count(skip) == ref.serialization_byte_count(skip)
        packed = ours.serialize(heamd.to_device(slab), skip)
        expected = ref.serialize(slab, skip)
        assert np.array_equal(packed.cpu().numpy(), expected)
        back = heamd.to_host(ours.deserialize(torch.from_numpy(expected).cuda(), skip))
        assert np.array_equal(back, ref.deserialize(expected, skip))
        if skip == 0:
            assert np.array_equal(back, slab)


@pytest.mark.parametrize("misalignment", [8, 1])
def test_serialize_into_misaligned_buffers(oracle, misalignment):
    """The 16-byte tile kernels need aligned buffers; records at 8-byte and at odd addresses take the word and the byte
    kernels and must give the same bytes (PolyRq+Serialize.swift:69-99 knows nothing of alignment)."""
    import ctypes

    import torch

    degree, moduli = 256, oracle.generate_primes([55, 40], False, 1)
    ours, ref = heamd.PolyContext(degree, moduli), oracle.PolyContext(degree, moduli)
    slab = _uniform(np.random.default_rng(5), (2,), moduli, degree)
    expected = ref.serialize(slab, 0)
    per_poly = ours.serialization_byte_count(0)
    lib = heamd.load_library()
    device_slab = heamd.to_device(slab)
    buffer = torch.zeros(2 * per_poly + 64, dtype=torch.uint8, device="cuda")
    view = buffer[misalignment: misalignment + 2 * per_poly]
    assert lib.he_poly_serialize_device(ours.h, ctypes.c_void_p(device_slab.data_ptr()), 2, 0,
                                        ctypes.c_void_p(view.data_ptr()), None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(view.cpu().numpy().reshape(2, per_poly), expected)
    back = torch.zeros_like(device_slab)
    assert lib.he_poly_deserialize_device(ours.h, ctypes.c_void_p(view.data_ptr()), per_poly, 2, 0,
                                          ctypes.c_void_p(back.data_ptr()), None) == 0
    torch.cuda.synchronize()
    assert np.array_equal(heamd.to_host(back), slab)


def test_deserialize_rejects_short_records(oracle):
    import torch

    narrow = heamd.PolyContext(32, oracle.generate_primes([5, 5, 5], False, 1, word_bits=32))
    wide = heamd.PolyContext(32, oracle.generate_primes([5, 5, 16], False, 1, word_bits=32))
    assert wide.serialization_byte_count() == 104  # PolyRq+SerializeTests.swift:21-36
    packed = narrow.serialize(heamd.to_device(np.zeros((1, 3, 32), dtype=np.uint64)))
    with pytest.raises(heamd.HeError) as err:
        wide.deserialize(packed)
    assert err.value.name == "serializedBufferSizeMismatch"
    with pytest.raises(heamd.HeError) as err:
        narrow.serialize(heamd.to_device(np.zeros((1, 3, 32), dtype=np.uint64)), skip_lsbs=5)
    assert err.value.name == "invalidCoefficientPacking"
    assert isinstance(packed, torch.Tensor)


@pytest.mark.parametrize("degree,bits,batch", [(512, [55, 40, 20], 5), (16, [20, 21], 3), (8192, [55, 55, 55, 55], 3),
                                                (4096, [27, 28, 28], 9), (64, [30, 31, 33], 70)])
def test_seeded_polynomials_match_oracle(oracle, degree, bits, batch):
    """`a` of a seeded ciphertext (SerializedCiphertext.swift:53-58): AES-128 CTR_DRBG stream, 4096-byte refills,
    128 bits per coefficient reduced mod q_i -- word-exact against the oracle (itself pinned by the NIST vectors)."""
    import torch

    moduli = oracle.generate_primes(bits, False, degree)
    ours, ref = heamd.PolyContext(degree, moduli), oracle.PolyContext(degree, moduli)
    rng = np.random.default_rng(degree + batch)
    seeds = rng.integers(0, 256, size=(batch, 32), dtype=np.uint8)
    seeds[0] = 0
    seeds[1] = 255
    got = heamd.to_host(ours.random_from_seeds(torch.from_numpy(seeds).cuda()))
    assert np.array_equal(got, ref.random_from_seeds(seeds))


def test_seeded_ciphertext_roundtrip(oracle, small):
    """Seeded wire format end to end on the device: serialize poly0, keep the seed; the server deserializes poly0,
    regenerates a from the seed and inverse-transforms it -- the rebuilt ciphertext decrypts to the message."""
    import torch

    ours, ref, client = small
    qctx_ref = ref.ciphertext_context()
    qctx = ours.ciphertext_context()
    r = random.Random(80)
    message = [r.randrange(ref.t) for _ in range(ref.degree)]
    seed = np.array([r.randrange(256) for _ in range(32)], dtype=np.uint8)
    # encryptZero with a = random(seed) in Eval form (Bfv+Encrypt.swift:150-181)
    a = qctx_ref.random_from_seeds(seed)[0]
    c0 = qctx_ref.inverse_ntt(qctx_ref.mul(a, client._s_eval_for(qctx_ref)))
    c0 = qctx_ref.neg(qctx_ref.add(c0, client._error(qctx_ref)))
    zero_ct = np.stack([c0, qctx_ref.inverse_ntt(a)])
    plain = client.encrypt(message)  # only used for its plaintext translation: subtract its own zero part
    ct = zero_ct.copy()
    q = 1
    for m in qctx_ref.moduli:
        q *= m
    t = ref.t
    for i, qi in enumerate(qctx_ref.moduli):
        for k, m in enumerate(message):
            adjust = ((q % t) * m + (t + 1) // 2) // t
            ct[0, i, k] = (int(ct[0, i, k]) + ((q // t) % qi) * m + adjust) % qi
    assert client.decrypt(ct) == message and plain.shape == ct.shape
    wire_poly0 = qctx.serialize(heamd.to_device(ct[:1]))
    rebuilt0 = qctx.deserialize(wire_poly0)
    rebuilt1 = qctx.inverse_ntt_(qctx.random_from_seeds(torch.from_numpy(seed[None]).cuda()))
    rebuilt = np.stack([heamd.to_host(rebuilt0)[0], heamd.to_host(rebuilt1)[0]])
    assert np.array_equal(rebuilt, ct)
    assert client.decrypt(rebuilt) == message


def test_single_modulus_context_edge_cases(oracle):
    """A context with one coefficient modulus has no key-switching modulus (Context.swift:102-107): the conversions
    still work, the key-switching entry points report the missing key."""
    degree = 64
    t = oracle.generate_primes([12], True, degree)[0]
    q = oracle.generate_primes([40], False, degree)
    ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
    assert ours.L == ref.L == 1
    rng = np.random.default_rng(95)
    pt = rng.integers(0, t, size=(2, degree), dtype=np.uint64)
    got = heamd.to_host(ours.plaintext_to_eval(heamd.to_device(pt)))
    assert np.array_equal(got, ref.plaintext_to_eval(pt))
    assert np.array_equal(heamd.to_host(ours.plaintext_to_coeff(heamd.to_device(got))), pt)
    x = _uniform(rng, (2,), q, degree)
    tool = ref.rns_tool()
    assert np.array_equal(heamd.to_host(ours.scale_and_round(heamd.to_device(x), 1)),
                          np.stack([tool.scale_and_round(p, 1) for p in x]))
    ct = heamd.to_device(np.zeros((1, 2, 1, degree), dtype=np.uint64))
    key = heamd.to_device(np.zeros((1, 2, 2, degree), dtype=np.uint64))
    with pytest.raises(heamd.HeError) as err:
        ours.apply_galois(ct, 3, key)
    assert err.value.name == "missingGaloisKey"


def test_zero_batches_are_no_ops(oracle):
    import torch

    degree = 64
    moduli = oracle.generate_primes([40, 41], False, degree)
    ctx = heamd.PolyContext(degree, moduli)
    empty = torch.empty((0, 2, degree), dtype=torch.int64, device="cuda")
    assert ctx.apply_galois(empty, 3).shape == (0, 2, degree)
    assert ctx.multiply_power_of_x(empty, 5).shape == (0, 2, degree)
    assert ctx.serialize(empty).shape == (0, ctx.serialization_byte_count())
    assert ctx.random_from_seeds(torch.empty((0, 32), dtype=torch.uint8, device="cuda")).shape == (0, 2, degree)
    assert ctx.deserialize(torch.empty((0, ctx.serialization_byte_count()), dtype=torch.uint8, device="cuda")).shape == (0, 2, degree)
