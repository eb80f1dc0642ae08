"""GPU parity of the device-resident PIR response against the oracle's composition of the pinned
primitives, plus the semantic check: the response decrypts to the queried database entry."""
import random

import numpy as np
import pytest

import heamd
from bfv_helpers import BfvClient

pytestmark = pytest.mark.gpu


def _uniform(rng, shape_prefix, moduli, degree):
    rows = [rng.integers(0, q, size=tuple(shape_prefix) + (degree,), dtype=np.uint64) for q in moduli]
    return np.ascontiguousarray(np.stack(rows, axis=len(shape_prefix)))


@pytest.fixture(scope="module")
def small(oracle):
    degree = 64
    t = oracle.generate_primes([17], True, degree)[0]
    q = oracle.generate_primes([40, 40, 40, 41], False, degree)
    ref = oracle.BfvContext(degree, t, q)
    return heamd.BfvContext(degree, t, q), ref, BfvClient(oracle, ref, seed=60)


@pytest.mark.parametrize("dims", [[4, 3], [4], [2, 2, 2], [5, 1]])
def test_pir_response_matches_oracle_and_decrypts(oracle, small, dims):
    ours, ref, client = small
    rng = random.Random(61 + len(dims))
    total = int(np.prod(dims))
    entries = [[rng.randrange(ref.t) for _ in range(ref.degree)] for _ in range(total)]
    database = ref.plaintext_to_eval(np.array(entries, dtype=np.uint64))
    present = np.ones(total, dtype=np.uint8)
    if total > 5:
        present[5] = 0
    qctx = ref.ciphertext_context()
    key = client.relinearization_key()
    one, zero = [1] + [0] * (ref.degree - 1), [0] * ref.degree
    selection = [rng.randrange(d) for d in dims]
    dim0 = np.stack([qctx.forward_ntt(client.encrypt(one if k == selection[0] else zero)) for k in range(dims[0])])
    rest_list = [client.encrypt(one if k == selection[i] else zero) for i in range(1, len(dims)) for k in range(dims[i])]
    rest = np.stack(rest_list) if rest_list else None
    expected = oracle.pir.compute_response_for_one_chunk(ref, dims, dim0, rest, database, present, key)
    got = heamd.to_host(ours.pir_compute_response_chunk(
        dims, heamd.to_device(dim0), None if rest is None else heamd.to_device(rest), heamd.to_device(database),
        present, heamd.to_device(key)))
    assert np.array_equal(got, expected)
    # column-major flattening: index = sum_i selection[i] * prod(dims[:i])
    index, stride = 0, 1
    for sel, d in zip(selection, dims):
        index += sel * stride
        stride *= d
    want = entries[index] if present[index] else zero
    assert client.decrypt(got, moduli_count=1) == want


def test_pir_response_rejects_mismatched_dimensions(small):
    ours, ref, _ = small
    L, n = ours.L, ours.degree
    dim0 = heamd.to_device(np.zeros((2, 2, L, n), dtype=np.uint64))
    rest = heamd.to_device(np.zeros((2, 2, L, n), dtype=np.uint64))
    database = heamd.to_device(np.zeros((6, L, n), dtype=np.uint64))
    key = heamd.to_device(np.zeros((L, 2, L + 1, n), dtype=np.uint64))
    with pytest.raises(heamd.HeError) as err:  # 3 columns but 2 remaining query ciphertexts (PirUtil.swift:422)
        ours.pir_compute_response_chunk([2, 3], dim0, rest[:2], database, None, key)
    assert err.value.name == "invalidArgument"


def test_pir_response_config_shape(oracle):
    """N=8192, L=4 (BASELINE configs[4] ring), a 16 x 8 chunk of uniform words: exact words vs the oracle."""
    degree = 8192
    q = oracle.generate_primes([55] * 5, False, degree)
    ours, ref = heamd.BfvContext(degree, 557057, q), oracle.BfvContext(degree, 557057, q)
    rng = np.random.default_rng(62)
    dims = [16, 8]
    moduli = q[:-1]
    dim0 = _uniform(rng, (dims[0], 2), moduli, degree)
    rest = _uniform(rng, (dims[1], 2), moduli, degree)
    database = _uniform(rng, (dims[0] * dims[1],), moduli, degree)
    key = _uniform(rng, (ours.L, 2), q, degree)
    expected = oracle.pir.compute_response_for_one_chunk(ref, dims, dim0, rest, database, None, key)
    got = heamd.to_host(ours.pir_compute_response_chunk(dims, heamd.to_device(dim0), heamd.to_device(rest),
                                                        heamd.to_device(database), None, heamd.to_device(key)))
    assert np.array_equal(got, expected)


def _compressed_query(ctx, total, ones):
    """PirUtil.compressInputsForOneCiphertext (PirUtil.swift:357-377)."""
    height = (total - 1).bit_length()
    inverse = pow(pow(2, height, ctx.t), -1, ctx.t)
    message = [0] * ctx.degree
    for index in ones:
        message[index] = inverse
    return message


@pytest.mark.parametrize("total,ones,key_shifts", [(8, [3], None), (5, [0, 4], None), (1, [0], None), (6, [5], None),
                                                   (8, [6], [2]), (13, [12, 1], None)])
def test_pir_expand_matches_oracle_and_decrypts(oracle, small, total, ones, key_shifts):
    """PirUtil.expand (PirUtil.swift:196-355): same words and order as the oracle's recursion; output i decrypts to
    the constant [i in ones].  key_shifts=[2] keeps only the element N/4 + 1, which the first two levels reach by
    repeated application (PirUtil.swift:221-231)."""
    ours, ref, client = small
    n = ref.degree
    shifts = key_shifts if key_shifts is not None else list(range(0, max((total - 1).bit_length(), 1)))
    keys = {(n >> k) + 1: client.galois_key((n >> k) + 1) for k in shifts}
    query = client.encrypt(_compressed_query(ref, total, ones))
    expected = oracle.pir.expand(ref, query[None], total, keys)
    device_keys = {e: heamd.to_device(k) for e, k in keys.items()}
    got = heamd.to_host(ours.pir_expand(heamd.to_device(query[None]), total, device_keys))
    assert np.array_equal(got, expected)
    if key_shifts is None:
        for index in range(total):
            assert client.decrypt(got[index]) == [1 if index in ones else 0] + [0] * (n - 1), index


def test_pir_expand_two_query_ciphertexts(oracle, small):
    """More outputs than one ciphertext can carry: the second ciphertext expands the remainder (PirUtil.swift:327-333)."""
    ours, ref, client = small
    n = ref.degree
    total = n + 3
    keys = {(n >> k) + 1: client.galois_key((n >> k) + 1) for k in range(0, (n - 1).bit_length())}
    first = client.encrypt(_compressed_query(ref, n, [7]))
    second = client.encrypt(_compressed_query(ref, 3, [2]))
    queries = np.stack([first, second])
    expected = oracle.pir.expand(ref, queries, total, keys)
    got = heamd.to_host(ours.pir_expand(heamd.to_device(queries), total, {e: heamd.to_device(k) for e, k in keys.items()}))
    assert np.array_equal(got, expected)
    assert client.decrypt(got[7])[0] == 1 and client.decrypt(got[8])[0] == 0
    assert client.decrypt(got[n + 2])[0] == 1 and client.decrypt(got[n])[0] == 0


def test_pir_expand_errors(small):
    ours, ref, client = small
    ct = heamd.to_device(np.zeros((1, 2, ours.L, ours.degree), dtype=np.uint64))
    with pytest.raises(heamd.HeError) as err:
        ours.pir_expand(ct, 4, {})
    assert err.value.name == "missingGaloisKey"
    with pytest.raises(heamd.HeError) as err:
        ours.pir_expand(ct, ours.degree + 1, {})
    assert err.value.name == "invalidArgument"


def test_pir_one_dimension_single_modulus(oracle):
    """dimension_count == 1 and L == 1: no ct x ct step, no key, no mod-switch (PirUtil.swift:448-485 degenerate)."""
    degree = 64
    t = oracle.generate_primes([12], True, degree)[0]
    q = oracle.generate_primes([45], False, degree)
    ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
    rng = np.random.default_rng(96)
    dim0 = _uniform(rng, (3, 2), q, degree)
    database = _uniform(rng, (3,), q, degree)
    expected = oracle.pir.compute_response_for_one_chunk(ref, [3], dim0, None, database, None, None)
    got = heamd.to_host(ours.pir_compute_response_chunk([3], heamd.to_device(dim0), None, heamd.to_device(database)))
    assert np.array_equal(got, expected)
