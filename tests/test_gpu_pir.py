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


@pytest.mark.parametrize("degree,bits", [(8192, [55] * 5), (4096, [55] * 3), (8192, [29, 60, 60])])
def test_pir_response_config_shape(oracle, degree, bits):
    """N=8192, L=4 (BASELINE configs[4] ring), a 16 x 8 chunk of uniform words: exact words vs the oracle -- on the degrees whose
    first remaining dimension reads the dim-0 results in Eval form (pir_api.cpp results_in_eval: out-of-place inverse transform,
    carry-counting sums), N = 4096 and one of the reference's 60-bit parameter sets beside it."""
    q = oracle.generate_primes(bits, False, degree)
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


@pytest.mark.parametrize("total,key_shifts", [(6, None), (13, None), (8, [2]), (4, [2]), (5, [2]), (16, [1, 3])])
def test_pir_expand_fused_levels(oracle, total, key_shifts):
    """PirUtil.expand on a ring with a tiled transform (N=4096): the children of a level leave its (last) key switch
    directly (no rotated copy, no separate step kernel).  With key_shifts=[2] only the element N/4 + 1 has a key: the
    first levels reach their element by applying it 4 and 2 times (PirUtil.swift:221-231) -- all but the last application
    are plain Galois key switches, the last one forms the children with the level's parents (and, for total = 4, writes
    them to their output slots); key_shifts=[1, 3] mixes levels with and without a key of their own.  Uniform words,
    word-exact against the oracle's recursion; the same for two queries with different keys in one call."""
    degree = 4096
    q = oracle.generate_primes([50, 45, 55], False, degree)
    ours, ref = heamd.BfvContext(degree, 65537, q), oracle.BfvContext(degree, 65537, q)
    rng = np.random.default_rng(100 + total)
    shifts = key_shifts if key_shifts is not None else list(range(0, (total - 1).bit_length()))
    queries = _uniform(rng, (2, 1, 2), q[:-1], degree)
    keys = [{(degree >> k) + 1: _uniform(rng, (ours.L, 2), q, degree) for k in shifts} for _ in range(2)]
    expected = [oracle.pir.expand(ref, queries[i], total, keys[i]) for i in range(2)]
    device_keys = [{e: heamd.to_device(k) for e, k in keys[i].items()} for i in range(2)]
    got = heamd.to_host(ours.pir_expand(heamd.to_device(queries[0]), total, device_keys[0]))
    assert np.array_equal(got, expected[0])
    both = heamd.to_host(ours.pir_expand_batch(heamd.to_device(queries), total, device_keys))
    assert np.array_equal(both[0], expected[0]) and np.array_equal(both[1], expected[1])


@pytest.mark.parametrize("total,key_shifts", [(400, None), (512, [0, 2, 4, 5, 6, 7, 8])])
def test_pir_expand_wide_levels(oracle, total, key_shifts):
    """Expansions wide enough for every run of equal keys in their last level to exceed two workgroup generations of
    key-switching rows (N=4096, L=2, two queries under different keys in one call: 200 and 256 parents each): there the
    children leave the key-MAC transform's store (ntt_kernels.hip kInverseFromKeyMacFinish with the expand end) -- per run
    of equal keys, straight to their output slots, ragged totals (400) and levels reached by two applications (no keys for the
    elements N/2 + 1 and N/8 + 1; below 2^7 + 1 an element does not compose to the next one, PirUtil.swift:222-231) included.  Word for word against the oracle's recursion."""
    degree = 4096
    q = oracle.generate_primes([50, 45, 55], False, degree)
    ours, ref = heamd.BfvContext(degree, 65537, q), oracle.BfvContext(degree, 65537, q)
    rng = np.random.default_rng(300 + total)
    shifts = key_shifts if key_shifts is not None else list(range(0, (total - 1).bit_length()))
    queries = _uniform(rng, (2, 1, 2), q[:-1], degree)
    keys = [{(degree >> k) + 1: _uniform(rng, (ours.L, 2), q, degree) for k in shifts} for _ in range(2)]
    device_keys = [{e: heamd.to_device(k) for e, k in keys[i].items()} for i in range(2)]
    both = heamd.to_host(ours.pir_expand_batch(heamd.to_device(queries), total, device_keys))
    for i in range(2):
        assert np.array_equal(both[i], oracle.pir.expand(ref, queries[i], total, keys[i])), i


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
    # an evaluation key outside the 2^k + 1 ladder: the reference's expandCiphertextForOneStep traps on
    # precondition(currElement == targetElement) (PirUtil.swift:222-231); here the call fails, it does not return another
    # expansion.  N = 64: the first level targets 65; 7 <= 65 applied 2^(6 - 2) = 16 times is 7^16 mod 128 = 1, not 65.
    key = heamd.to_device(np.zeros((ours.L, 2, ours.L + 1, ours.degree), dtype=np.uint64))
    assert pow(7, 16, 2 * ours.degree) != ours.degree + 1
    with pytest.raises(heamd.HeError) as err:
        ours.pir_expand(ct, 4, {7: key})
    assert err.value.name == "missingGaloisKey"
    with pytest.raises(heamd.HeError) as err:  # element 1 (the identity) can reach nothing
        ours.pir_expand(ct, 4, {1: key})
    assert err.value.name == "missingGaloisKey"


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


# ---------------------------------------------------------------------------------------------------------------------
# column shards (PirUtil.swift:427-445: columns are grouped over tasks; here: over GPUs), chunk loop (:533-563)
@pytest.mark.parametrize("shards", [2, 3])
def test_column_sharded_response_equals_unsharded(oracle, small, shards):
    """The dim-0 inner products of each column shard (what one GPU of a column-sharded deployment computes, sharding by
    heamd.sharding.shard_bounds), concatenated as the RCCL all-gather leaves them, followed by the remaining dimensions,
    is word for word the unsharded chunk response -- including a ragged and a masked shard."""
    import torch

    from heamd import sharding

    ours, ref, client = small
    rng = np.random.default_rng(63 + shards)
    dims = [4, 5]
    moduli = ref.ciphertext_context().moduli
    dim0 = _uniform(rng, (dims[0], 2), moduli, ours.degree)
    rest = _uniform(rng, (dims[1], 2), moduli, ours.degree)
    database = _uniform(rng, (dims[0] * dims[1],), moduli, ours.degree)
    present = np.ones(dims[0] * dims[1], dtype=np.uint8)
    present[[2, 9, 17]] = 0
    key = client.relinearization_key()
    dim0_dev, rest_dev, key_dev = heamd.to_device(dim0), heamd.to_device(rest), heamd.to_device(key)
    whole = heamd.to_host(ours.pir_compute_response_chunk(dims, dim0_dev, rest_dev, heamd.to_device(database), present,
                                                          key_dev))
    assert np.array_equal(whole, oracle.pir.compute_response_for_one_chunk(ref, dims, dim0, rest, database, present, key))
    columns = database.reshape(dims[1], dims[0], ours.L, ours.degree)
    masks = present.reshape(dims[1], dims[0])
    parts = []
    for rank in range(shards):
        begin, end = sharding.shard_bounds(dims[1], shards, rank)
        if end == begin:
            continue
        mask_dev = torch.from_numpy(masks[begin:end].copy()).cuda()
        parts.append(ours.pir_dim0_columns(dim0_dev, heamd.to_device(columns[begin:end].copy()), mask_dev))
    gathered = torch.cat(parts, dim=0).contiguous()
    unsharded = ours.pir_dim0_columns(dim0_dev, heamd.to_device(columns.copy()), torch.from_numpy(masks.copy()).cuda())
    assert torch.equal(gathered, unsharded)
    got = heamd.to_host(ours.pir_remaining_dimensions(dims, gathered, rest_dev, key_dev))
    assert np.array_equal(got, whole)


@pytest.mark.parametrize("piece,cache", [(None, False), ("1", True), ("2", True), ("2", False)])
def test_multi_chunk_response_matches_oracle(oracle, small, monkeypatch, piece, cache):
    """PirUtil.computeResponse's chunk loop (PirUtil.swift:533-563): three chunks of one database answered with one
    query, each chunk word for word the oracle's computeResponseForOneChunk; nil plaintexts in the second chunk.
    piece: the chunks answered in pieces of that many (a ragged last one), each piece's remaining dimensions on a lane of
    the context beside the next piece's dim-0 pass (pir_api.cpp overlap_piece_chunks; at this size only when forced);
    cache: scratch from the library's block cache (he_set_scratch_cache) instead of the HIP pool."""
    import torch

    if piece is not None:
        monkeypatch.setenv("HEAMD_PIR_PIECE_CHUNKS", piece)
    heamd.set_scratch_cache(2**64 - 1 if cache else 0)
    ours, ref, client = small
    rng = random.Random(71)
    dims, chunks = [4, 3], 3
    per_chunk = 12
    entries = [[rng.randrange(ref.t) for _ in range(ref.degree)] for _ in range(per_chunk * chunks)]
    database = ref.plaintext_to_eval(np.array(entries, dtype=np.uint64)).reshape(chunks, per_chunk, ref.L, ref.degree)
    present = np.ones((chunks, per_chunk), dtype=np.uint8)
    present[1, [0, 7]] = 0
    qctx = ref.ciphertext_context()
    key = client.relinearization_key()
    one, zero = [1] + [0] * (ref.degree - 1), [0] * ref.degree
    selection = [2, 1]
    dim0 = np.stack([qctx.forward_ntt(client.encrypt(one if k == selection[0] else zero)) for k in range(dims[0])])
    rest = np.stack([client.encrypt(one if k == selection[1] else zero) for k in range(dims[1])])
    got = heamd.to_host(ours.pir_compute_response(dims, heamd.to_device(dim0), heamd.to_device(rest),
                                                  heamd.to_device(database), chunks,
                                                  present_device=torch.from_numpy(present).cuda(),
                                                  relinearization_key=heamd.to_device(key)))
    index = selection[0] + dims[0] * selection[1]
    for chunk in range(chunks):
        expected = oracle.pir.compute_response_for_one_chunk(ref, dims, dim0, rest, database[chunk], present[chunk], key)
        assert np.array_equal(got[chunk], expected), chunk
        want = entries[chunk * per_chunk + index] if present[chunk, index] else zero
        assert client.decrypt(got[chunk], moduli_count=1) == want
    heamd.set_scratch_cache(0)


@pytest.mark.parametrize("queries,piece", [(2, None), (3, None), (4, None), (3, "1"), (4, "2")])
def test_queries_share_one_pass_over_the_database(oracle, small, monkeypatch, queries, piece):
    """he_pir_compute_response_queries_device: the dim-0 inner products of several queries stream the database once;
    every query's responses are the oracle's computeResponseForOneChunk for that query alone (different selections,
    different clients' relinearization keys), nil plaintexts included, and decrypt to the selected entries.
    piece: as in test_multi_chunk_response_matches_oracle (three chunks then)."""
    import torch

    ours, ref, client = small
    rng = random.Random(300 + queries)
    dims, chunks, per_chunk = [4, 3], 2 if piece is None else 3, 12
    if piece is not None:
        monkeypatch.setenv("HEAMD_PIR_PIECE_CHUNKS", piece)
        heamd.set_scratch_cache()
    entries = [[rng.randrange(ref.t) for _ in range(ref.degree)] for _ in range(per_chunk * chunks)]
    database = ref.plaintext_to_eval(np.array(entries, dtype=np.uint64)).reshape(chunks, per_chunk, ref.L, ref.degree)
    present = np.ones((chunks, per_chunk), dtype=np.uint8)
    present[1, [3, 10]] = 0
    qctx = ref.ciphertext_context()
    key = client.relinearization_key()
    one, zero = [1] + [0] * (ref.degree - 1), [0] * ref.degree
    selections = [(q % dims[0], (q + 1) % dims[1]) for q in range(queries)]
    dim0 = np.stack([np.stack([qctx.forward_ntt(client.encrypt(one if k == sel[0] else zero)) for sel in selections])
                     for k in range(dims[0])])                                     # [d0][queries][2][L][N]
    rest = np.stack([np.stack([client.encrypt(one if k == sel[1] else zero) for k in range(dims[1])])
                     for sel in selections])                                       # [queries][d1][2][L][N]
    device_key = heamd.to_device(key)
    got = heamd.to_host(ours.pir_compute_response_queries(dims, heamd.to_device(dim0), heamd.to_device(rest),
                                                          heamd.to_device(database), chunks, [device_key] * queries,
                                                          present_device=torch.from_numpy(present).cuda()))
    for q, sel in enumerate(selections):
        own_dim0 = np.ascontiguousarray(dim0[:, q])
        index = sel[0] + dims[0] * sel[1]
        for chunk in range(chunks):
            expected = oracle.pir.compute_response_for_one_chunk(ref, dims, own_dim0, rest[q], database[chunk],
                                                                 present[chunk], key)
            assert np.array_equal(got[q, chunk], expected), (q, chunk)
            want = entries[chunk * per_chunk + index] if present[chunk, index] else zero
            assert client.decrypt(got[q, chunk], moduli_count=1) == want
    heamd.set_scratch_cache(0)


def test_queries_over_a_one_dimensional_database(oracle, small):
    """One dimension: no remaining query, no relinearization key (PirUtil.swift:448 loop is empty); three queries in one
    call each equal the single-query chunk response; five queries are refused."""
    ours, ref, client = small
    rng = random.Random(311)
    dims, chunks = [5], 2
    entries = [[rng.randrange(ref.t) for _ in range(ref.degree)] for _ in range(5 * chunks)]
    database = ref.plaintext_to_eval(np.array(entries, dtype=np.uint64)).reshape(chunks, 5, ref.L, ref.degree)
    qctx = ref.ciphertext_context()
    one, zero = [1] + [0] * (ref.degree - 1), [0] * ref.degree
    picks = [4, 0, 2]
    dim0 = np.stack([np.stack([qctx.forward_ntt(client.encrypt(one if k == pick else zero)) for pick in picks])
                     for k in range(dims[0])])
    device_db = heamd.to_device(database)
    got = heamd.to_host(ours.pir_compute_response_queries(dims, heamd.to_device(dim0), None, device_db, chunks, None))
    for q, pick in enumerate(picks):
        single = heamd.to_host(ours.pir_compute_response(dims, heamd.to_device(np.ascontiguousarray(dim0[:, q])), None,
                                                         device_db, chunks))
        assert np.array_equal(got[q], single)
        for chunk in range(chunks):
            assert client.decrypt(got[q, chunk], moduli_count=1) == entries[chunk * 5 + pick]
    five = heamd.to_device(np.zeros((dims[0], 5, 2, ref.L, ref.degree), dtype=np.uint64))
    with pytest.raises(heamd.HeError) as err:
        ours.pir_compute_response_queries(dims, five, None, device_db, chunks, None)
    assert err.value.name == "invalidArgument"


@pytest.mark.parametrize("indices", [1, 3, 6])
def test_whole_query_response(oracle, small, indices):
    """he_pir_compute_response_to_query_device = PirUtil.computeResponse with one database (PirUtil.swift:490-568): the
    query ciphertext expands into sum(dimensions) selection ciphertexts per index, each index's first dimensions[0] go to
    Eval, every chunk is answered.  Word for word the composition of the oracle's expand and chunk responses, for 1, 3
    and 6 indices in one Query (6: a group of four sharing the database pass, then two), and the responses decrypt to
    the selected entries."""
    import torch

    ours, ref, client = small
    n = ref.degree
    rng = random.Random(400 + indices)
    dims, chunks, per_chunk = [4, 3], 2, 12
    expanded_count = sum(dims)
    total = expanded_count * indices
    entries = [[rng.randrange(ref.t) for _ in range(n)] for _ in range(per_chunk * chunks)]
    database = ref.plaintext_to_eval(np.array(entries, dtype=np.uint64)).reshape(chunks, per_chunk, ref.L, n)
    present = np.ones((chunks, per_chunk), dtype=np.uint8)
    present[0, 5] = 0
    selections = [(rng.randrange(dims[0]), rng.randrange(dims[1])) for _ in range(indices)]
    ones = []
    for i, (a, b) in enumerate(selections):
        ones += [i * expanded_count + a, i * expanded_count + dims[0] + b]
    query = client.encrypt(_compressed_query(ref, total, ones))[None]
    shifts = range(0, max((total - 1).bit_length(), 1))
    galois = {(n >> k) + 1: client.galois_key((n >> k) + 1) for k in shifts}
    relin = client.relinearization_key()
    got = heamd.to_host(ours.pir_compute_response_to_query(
        dims, heamd.to_device(query), indices, {e: heamd.to_device(k) for e, k in galois.items()}, heamd.to_device(relin),
        heamd.to_device(database), chunks, present_devices=torch.from_numpy(present).cuda()))
    expanded = oracle.pir.expand(ref, query, total, galois)
    qctx = ref.ciphertext_context()
    zero = [0] * n
    for i, (a, b) in enumerate(selections):
        mine = expanded[i * expanded_count: (i + 1) * expanded_count]
        dim0 = np.stack([qctx.forward_ntt(ct) for ct in mine[: dims[0]]])
        index = a + dims[0] * b
        for chunk in range(chunks):
            expected = oracle.pir.compute_response_for_one_chunk(ref, dims, dim0, mine[dims[0]:], database[chunk],
                                                                 present[chunk], relin)
            assert np.array_equal(got[i, chunk], expected), (i, chunk)
            want = entries[chunk * per_chunk + index] if present[chunk, index] else zero
            assert client.decrypt(got[i, chunk], moduli_count=1) == want


def test_whole_query_with_a_database_per_index(oracle, small):
    """databases.count >= query.indicesCount (PirUtil.swift:498-500, :518): index i is answered from databases[i]; two
    databases for three indices is the reference's invalidBatchSize."""
    import torch

    ours, ref, client = small
    n = ref.degree
    rng = random.Random(500)
    dims, chunks, per_chunk, indices = [3, 2], 1, 6, 3
    expanded_count = sum(dims)
    total = expanded_count * indices
    tables = [[[rng.randrange(ref.t) for _ in range(n)] for _ in range(per_chunk)] for _ in range(indices)]
    databases = [ref.plaintext_to_eval(np.array(t, dtype=np.uint64)).reshape(chunks, per_chunk, ref.L, n) for t in tables]
    selections = [(rng.randrange(dims[0]), rng.randrange(dims[1])) for _ in range(indices)]
    ones = []
    for i, (a, b) in enumerate(selections):
        ones += [i * expanded_count + a, i * expanded_count + dims[0] + b]
    query = client.encrypt(_compressed_query(ref, total, ones))[None]
    galois = {(n >> k) + 1: heamd.to_device(client.galois_key((n >> k) + 1)) for k in range((total - 1).bit_length())}
    relin = heamd.to_device(client.relinearization_key())
    device_dbs = [heamd.to_device(d) for d in databases]
    got = heamd.to_host(ours.pir_compute_response_to_query(dims, heamd.to_device(query), indices, galois, relin, device_dbs,
                                                           chunks))
    for i, (a, b) in enumerate(selections):
        assert client.decrypt(got[i, 0], moduli_count=1) == tables[i][a + dims[0] * b], i
    with pytest.raises(heamd.HeError) as err:
        ours.pir_compute_response_to_query(dims, heamd.to_device(query), indices, galois, relin, device_dbs[:2], chunks)
    assert err.value.name == "invalidArgument"
    assert torch.cuda.is_available()


def test_pir_on_a_uint32_parameter_set(oracle):
    """The reference's PIR parameter sets with 27/28-bit moduli are Bfv<UInt32> (EncryptionParameters.swift:313-345):
    the PIR entry points take such a context on 8-byte slabs (zero-extended words; the constants -- 29-bit Bsk primes,
    gamma, mTilde = 2^16 -- are the UInt32 ones).  N=4096: expansion and a two-chunk response word for word against
    the 32-bit oracle, for one query and for three sharing the database pass."""
    degree = 4096
    q = [(1 << 27) - 40959, (1 << 28) - 65535, (1 << 28) - 73727]  # n_4096_logq_27_28_28
    t = (1 << 16) + 1
    ours = heamd.BfvContext(degree, t, q, word_bits=32)
    ref = oracle.BfvContext(degree, t, q, word_bits=32)
    rng = np.random.default_rng(99)
    moduli = q[:-1]
    total = 6
    query = _uniform(rng, (1, 2), moduli, degree)
    keys = {(degree >> k) + 1: _uniform(rng, (ours.L, 2), q, degree) for k in range(3)}
    expanded = heamd.to_host(ours.pir_expand(heamd.to_device(query), total, {e: heamd.to_device(k) for e, k in keys.items()}))
    assert np.array_equal(expanded, oracle.pir.expand(ref, query, total, keys))
    dims, chunks, queries = [4, 3], 2, 3
    database = _uniform(rng, (chunks, 12), moduli, degree)
    dim0 = _uniform(rng, (dims[0], queries, 2), moduli, degree)
    rest = _uniform(rng, (queries, dims[1], 2), moduli, degree)
    relin = [_uniform(rng, (ours.L, 2), q, degree) for _ in range(queries)]
    got = heamd.to_host(ours.pir_compute_response_queries(dims, heamd.to_device(dim0), heamd.to_device(rest),
                                                          heamd.to_device(database), chunks,
                                                          [heamd.to_device(k) for k in relin]))
    for query_index in range(queries):
        own = np.ascontiguousarray(dim0[:, query_index])
        for chunk in range(chunks):
            expected = oracle.pir.compute_response_for_one_chunk(ref, dims, own, rest[query_index], database[chunk], None,
                                                                 relin[query_index])
            assert np.array_equal(got[query_index, chunk], expected), (query_index, chunk)
    single = heamd.to_host(ours.pir_compute_response(dims, heamd.to_device(np.ascontiguousarray(dim0[:, 0])),
                                                     heamd.to_device(rest[0]), heamd.to_device(database), chunks,
                                                     relinearization_key=heamd.to_device(relin[0])))
    assert np.array_equal(single, got[0])


@pytest.mark.parametrize("dims", [[5], [4, 3], [3, 2, 2]])
def test_pir_response_on_packed_uint32_slabs(oracle, dims):
    """he_pir_compute_response_device_u32: query, database, key and responses in UInt32 words (Bfv<UInt32>, the
    n_4096_logq_27_28_28 parameter set; 1-, 2- and 3-dimensional databases, three chunks, nil plaintexts): every chunk
    response word for word the 32-bit oracle's."""
    import torch

    degree = 4096
    q = [(1 << 27) - 40959, (1 << 28) - 65535, (1 << 28) - 73727]
    t = (1 << 16) + 1
    ours, ref = heamd.BfvContext32(degree, t, q), oracle.BfvContext(degree, t, q, word_bits=32)
    rng = np.random.default_rng(len(dims))
    moduli = q[:-1]
    per_chunk, chunks, rest_count = int(np.prod(dims)), 3, sum(dims[1:])
    database = _uniform(rng, (chunks, per_chunk), moduli, degree)
    present = np.ones((chunks, per_chunk), dtype=np.uint8)
    present[1, [0, per_chunk - 1]] = 0
    dim0 = _uniform(rng, (dims[0], 2), moduli, degree)
    rest = _uniform(rng, (max(rest_count, 1), 2), moduli, degree)[:rest_count]
    key = _uniform(rng, (ours.L, 2), q, degree)
    dev = heamd.to_device32
    got = heamd.to_host32(ours.pir_compute_response(dims, dev(dim0), dev(rest) if rest_count else None, dev(database), chunks,
                                                    present_device=torch.from_numpy(present).cuda(),
                                                    relinearization_key=dev(key) if rest_count else None))
    for chunk in range(chunks):
        expected = oracle.pir.compute_response_for_one_chunk(ref, dims, dim0, rest, database[chunk], present[chunk],
                                                             key if rest_count else None)
        assert np.array_equal(got[chunk], expected), chunk


@pytest.mark.parametrize("dims,queries", [([4, 3], 2), ([3, 2, 2], 4), ([5], 3)])
def test_queries_share_one_pass_on_packed_uint32_slabs(oracle, dims, queries):
    """he_pir_compute_response_queries_device_u32: several queries over one 4-byte database in one call, each with its own
    relinearization key and a nil-plaintext mask: every response word for word the 32-bit oracle's single-query one."""
    import torch

    degree = 4096
    q = [(1 << 27) - 40959, (1 << 28) - 65535, (1 << 28) - 73727]
    t = (1 << 16) + 1
    ours, ref = heamd.BfvContext32(degree, t, q), oracle.BfvContext(degree, t, q, word_bits=32)
    rng = np.random.default_rng(10 * len(dims) + queries)
    moduli = q[:-1]
    per_chunk, chunks, rest_count = int(np.prod(dims)), 2, sum(dims[1:])
    database = _uniform(rng, (chunks, per_chunk), moduli, degree)
    present = np.ones((chunks, per_chunk), dtype=np.uint8)
    present[0, 1] = 0
    dim0 = _uniform(rng, (dims[0], queries, 2), moduli, degree)
    rest = _uniform(rng, (queries, max(rest_count, 1), 2), moduli, degree)[:, :rest_count]
    keys = [_uniform(rng, (ours.L, 2), q, degree) for _ in range(queries)]
    dev = heamd.to_device32
    got = heamd.to_host32(ours.pir_compute_response_queries(
        dims, dev(dim0), dev(np.ascontiguousarray(rest)) if rest_count else None, dev(database), chunks,
        [dev(k) for k in keys] if rest_count else None, present_device=torch.from_numpy(present).cuda()))
    for query in range(queries):
        own = np.ascontiguousarray(dim0[:, query])
        for chunk in range(chunks):
            expected = oracle.pir.compute_response_for_one_chunk(ref, dims, own, rest[query], database[chunk], present[chunk],
                                                                 keys[query] if rest_count else None)
            assert np.array_equal(got[query, chunk], expected), (query, chunk)
    with pytest.raises(heamd.HeError):
        five = dev(_uniform(rng, (dims[0], 5, 2), moduli, degree))
        ours.pir_compute_response_queries(dims, five, None if not rest_count else dev(np.zeros((5, rest_count, 2, ours.L, degree), dtype=np.uint64)),
                                          dev(database), chunks, [dev(keys[0])] * 5 if rest_count else None)


def test_whole_query_on_packed_uint32_slabs(oracle):
    """he_pir_compute_response_to_query_device_u32 (n_4096_logq_27_28_28, five indices -- a group of four that shares the
    pass over the database and one on its own --, 4 x 3 database, two chunks): the composition of the 32-bit oracle's
    expand and chunk responses, word for word."""
    degree = 4096
    q = [(1 << 27) - 40959, (1 << 28) - 65535, (1 << 28) - 73727]
    t = (1 << 16) + 1
    ours, ref = heamd.BfvContext32(degree, t, q), oracle.BfvContext(degree, t, q, word_bits=32)
    rng = np.random.default_rng(1234)
    moduli = q[:-1]
    dims, chunks, indices = [4, 3], 2, 5
    expanded_count = sum(dims)
    total = expanded_count * indices
    query = _uniform(rng, (1, 2), moduli, degree)
    galois = {(degree >> k) + 1: _uniform(rng, (ours.L, 2), q, degree) for k in range((total - 1).bit_length())}
    relin = _uniform(rng, (ours.L, 2), q, degree)
    database = _uniform(rng, (chunks, 12), moduli, degree)
    dev = heamd.to_device32
    got = heamd.to_host32(ours.pir_compute_response_to_query(
        dims, dev(query), indices, {e: heamd.to_device(k) for e, k in galois.items()}, dev(relin), dev(database), chunks))
    expanded = oracle.pir.expand(ref, query, total, galois)
    qctx = ref.ciphertext_context()
    for i in range(indices):
        mine = expanded[i * expanded_count: (i + 1) * expanded_count]
        dim0 = np.stack([qctx.forward_ntt(ct) for ct in mine[: dims[0]]])
        for chunk in range(chunks):
            expected = oracle.pir.compute_response_for_one_chunk(ref, dims, dim0, mine[dims[0]:], database[chunk], None, relin)
            assert np.array_equal(got[i, chunk], expected), (i, chunk)


def test_queries_share_one_pass_config_shape(oracle):
    """The same on BASELINE config 5's ring (N=8192, L=4; the LDS-tiled kernel), 3 queries with their own keys over two
    8 x 4 chunks of uniform words: each query's responses equal the single-query entry point's word for word."""
    degree = 8192
    q = oracle.generate_primes([55] * 5, False, degree)
    ours = heamd.BfvContext(degree, 557057, q)
    rng = np.random.default_rng(77)
    dims, chunks, queries = [8, 4], 2, 3
    moduli = q[:-1]
    dim0 = heamd.to_device(_uniform(rng, (dims[0], queries, 2), moduli, degree))
    rest = heamd.to_device(_uniform(rng, (queries, dims[1], 2), moduli, degree))
    database = heamd.to_device(_uniform(rng, (chunks, dims[0] * dims[1]), moduli, degree))
    keys = [heamd.to_device(_uniform(rng, (ours.L, 2), q, degree)) for _ in range(queries)]
    got = ours.pir_compute_response_queries(dims, dim0, rest, database, chunks, keys)
    for query in range(queries):
        single = ours.pir_compute_response(dims, dim0[:, query].contiguous(), rest[query], database, chunks,
                                           relinearization_key=keys[query])
        assert bool((got[query] == single).all()), query


def _unpack_rows(packed, moduli, degree, count):
    """Host restatement of the packed layout (kernels.hpp PackedLayout): per row a little-endian stream of N fields of
    bits(q) bits, rows in whole 8-byte words."""
    widths = [int(m).bit_length() for m in moduli]
    words = sum(degree // 64 * w for w in widths)
    out = np.zeros((count, len(moduli), degree), dtype=np.uint64)
    data = [int(v) for v in packed[: count * words]]
    for p in range(count):
        at = p * words
        for r, w in enumerate(widths):
            row_words = degree // 64 * w
            stream = 0
            for j in range(row_words):
                stream |= data[at + j] << (64 * j)
            mask = (1 << w) - 1
            for i in range(degree):
                out[p, r, i] = (stream >> (i * w)) & mask
            at += row_words
    return out


@pytest.mark.parametrize("bits", [[55, 55, 55], [40, 56, 33, 55], [62, 61, 62], [27, 28, 29]])
def test_packed_database_matches_plain(oracle, bits):
    """The packed plaintext layout (bits(q) bits per word, he_bfv_pack_plaintexts_device): the packed words unpack to the
    plaintexts on the host, and the inner products and the whole chunk loop over the packed database equal the ones over
    8-byte words word for word -- masks included, moduli of mixed widths, 62-bit moduli (full accumulator)."""
    import torch

    degree = 256
    t = oracle.generate_primes([17], True, degree)[0]
    q = oracle.generate_primes(bits, False, degree)
    ours = heamd.BfvContext(degree, t, q)
    moduli = q[:-1]
    rng = np.random.default_rng(len(bits))
    dims, chunks = [5, 3], 2
    database = _uniform(rng, (chunks, dims[0] * dims[1]), moduli, degree)
    database[0, 0] = np.array(moduli, dtype=np.uint64)[:, None] - np.uint64(1)
    database[0, 1] = 0
    device_db = heamd.to_device(database)
    packed = ours.pack_plaintexts(device_db)
    assert packed.numel() == chunks * 15 * ours.packed_plaintext_words() + 1
    assert np.array_equal(_unpack_rows(heamd.to_host(packed), moduli, degree, chunks * 15), database.reshape(-1, len(moduli), degree))
    present = np.ones((chunks, 15), dtype=np.uint8)
    present[1, [2, 14]] = 0
    mask = torch.from_numpy(present).cuda()
    dim0 = heamd.to_device(_uniform(rng, (dims[0], 2), moduli, degree))
    rest = heamd.to_device(_uniform(rng, (dims[1], 2), moduli, degree))
    key = heamd.to_device(_uniform(rng, (ours.L, 2), q, degree))
    # the inner product alone: the first chunk's columns ([column][d0] plaintexts)
    plain = ours.inner_product_plain_resident(dim0, device_db[0], mask[0], 2, dims[1])
    from_packed = ours.inner_product_plain_packed(dim0, packed, mask[0], 2, dims[1])
    assert bool((plain == from_packed).all())
    want = ours.pir_compute_response(dims, dim0, rest, device_db, chunks, present_device=mask, relinearization_key=key)
    got = ours.pir_compute_response_packed(dims, dim0, rest, packed, chunks, present_device=mask, relinearization_key=key)
    assert bool((want == got).all())


def test_packed_database_config_shape(oracle):
    """BASELINE config 5's ring (N=8192, L=4, 55-bit moduli: 56 320 bytes per row instead of 65 536): 16 x 4 database,
    packed and plain inner products agree on every word, and one column equals the oracle's."""
    degree = 8192
    q = oracle.generate_primes([55] * 5, False, degree)
    ours, ref = heamd.BfvContext(degree, 557057, q), oracle.BfvContext(degree, 557057, q)
    moduli = q[:-1]
    assert ours.packed_plaintext_words() == 4 * 8192 * 55 // 64
    rng = np.random.default_rng(88)
    count, columns = 16, 4
    cts = _uniform(rng, (count, 2), moduli, degree)
    pts = _uniform(rng, (columns, count), moduli, degree)
    device_cts, device_pts = heamd.to_device(cts), heamd.to_device(pts)
    packed = ours.pack_plaintexts(device_pts)
    got = ours.inner_product_plain_packed(device_cts, packed, None, 2, columns)
    assert bool((got == ours.inner_product_plain_resident(device_cts, device_pts, None, 2, columns)).all())
    assert np.array_equal(heamd.to_host(got)[3], ref.inner_product_plain(cts, pts[3], None))


def test_column_shard_at_the_benchmark_row_count(oracle):
    """BASELINE configs[4]'s ring and row count (N=8192, L=4, d0 = 1024 query ciphertexts), two columns: the sampled
    output words equal the oracle's lazy inner product, every word is canonical, and the two-column launch equals two
    one-column launches (a column's result does not depend on its shard)."""
    import torch

    degree, d0, columns = 8192, 1024, 2
    q = oracle.generate_primes([55] * 5, False, degree)
    ours, ref = heamd.BfvContext(degree, 557057, q), oracle.BfvContext(degree, 557057, q)
    moduli = q[:-1]
    gen = torch.Generator(device="cuda")
    gen.manual_seed(64)
    bound = torch.tensor(moduli, dtype=torch.int64, device="cuda").view(1, 1, len(moduli), 1)
    cts = torch.randint(0, 1 << 62, (d0, 2, len(moduli), degree), dtype=torch.int64, device="cuda", generator=gen) % bound
    pts = torch.randint(0, 1 << 62, (columns, d0, len(moduli), degree), dtype=torch.int64, device="cuda",
                        generator=gen) % bound.view(1, 1, len(moduli), 1)
    both = ours.inner_product_plain_resident(cts, pts, None, 2, columns)
    for c in range(columns):
        assert torch.equal(both[c], ours.inner_product_plain_resident(cts, pts[c:c + 1].contiguous(), None, 2, 1)[0])
    assert bool((both < bound).all())
    # sampled words against Python integers: out[c][poly][r][k] = sum_j cts[j][poly][r][k] * pts[c][j][r][k] mod q_r
    host_cts, host_pts, host_out = heamd.to_host(cts), heamd.to_host(pts), heamd.to_host(both)
    rng = random.Random(65)
    for _ in range(40):
        c, poly, r, k = rng.randrange(columns), rng.randrange(2), rng.randrange(len(moduli)), rng.randrange(degree)
        want = sum(int(host_cts[j, poly, r, k]) * int(host_pts[c, j, r, k]) for j in range(d0)) % moduli[r]
        assert int(host_out[c, poly, r, k]) == want


@pytest.mark.parametrize("total", [8, 13, 1])
def test_pir_expand_batch_of_queries_from_two_clients(oracle, small, total):
    """he_pir_expand_batch_device: four queries of one shape expanded level by level together -- queries 0, 1 under one
    client's Galois keys, 2, 3 under another's -- each query's outputs word for word the oracle's expand under its own
    keys (ragged totals put leaves on two levels, which exercises the per-query strides of the leaf / parent moves)."""
    ours, ref, client = small
    other = BfvClient(oracle, ref, seed=161)
    n = ref.degree
    shifts = list(range(0, max((total - 1).bit_length(), 1)))
    clients = [client, client, other, other]
    key_sets = {id(c): {(n >> k) + 1: c.galois_key((n >> k) + 1) for k in shifts} for c in (client, other)}
    device_sets = {cid: {e: heamd.to_device(k) for e, k in keys.items()} for cid, keys in key_sets.items()}
    ones = [[0], [total - 1], [total // 2], [0, total - 1] if total > 1 else [0]]
    queries = np.stack([c.encrypt(_compressed_query(ref, total, o))[None] for c, o in zip(clients, ones)])
    got = heamd.to_host(ours.pir_expand_batch(heamd.to_device(queries), total, [device_sets[id(c)] for c in clients]))
    assert got.shape == (4, total, 2, ref.L, n)
    for q, (c, o) in enumerate(zip(clients, ones)):
        expected = oracle.pir.expand(ref, queries[q], total, key_sets[id(c)])
        assert np.array_equal(got[q], expected), q
        for index in range(total):
            assert c.decrypt(got[q, index]) == [1 if index in o else 0] + [0] * (n - 1), (q, index)


def test_new_pir_entry_points_edge_cases(small):
    """Empty and malformed calls of the column-shard / chunk-loop / batch-expand entry points: nothing to do is HE_OK,
    a shape the reference would trap on (PirUtil.swift:420-422, :325-326) or a null operand is invalidArgument."""
    import ctypes

    import torch

    ours, ref, _ = small
    lib = heamd.load_library()
    L, n = ours.L, ours.degree
    dims = (ctypes.c_uint32 * 2)(2, 3)
    dim0 = heamd.to_device(np.zeros((2, 2, L, n), dtype=np.uint64))
    rest = heamd.to_device(np.zeros((3, 2, L, n), dtype=np.uint64))
    database = heamd.to_device(np.zeros((2, 6, L, n), dtype=np.uint64))
    key = heamd.to_device(np.zeros((L, 2, L + 1, n), dtype=np.uint64))
    out = torch.zeros((2, 2, 1, n), dtype=torch.int64, device="cuda")
    p = lambda t: ctypes.c_void_p(t.data_ptr())  # noqa: E731
    # zero columns / zero chunks / zero queries: nothing to do
    assert lib.he_pir_dim0_columns_device(ours.h, p(dim0), 2, p(database), None, 0, p(out), None) == 0
    assert lib.he_pir_compute_response_device(ours.h, dims, 2, p(dim0), p(rest), 3, p(database), None, 0, p(key), p(out),
                                              None) == 0
    assert lib.he_pir_expand_batch_device(ours.h, p(dim0), 0, 1, 4, None, None, 0, p(out), None) == 0
    # null operands, mismatched query, empty dimensions
    assert lib.he_pir_dim0_columns_device(ours.h, None, 2, p(database), None, 3, p(out), None) == 16
    assert lib.he_pir_compute_response_device(ours.h, dims, 2, p(dim0), p(rest), 2, p(database), None, 2, p(key), p(out),
                                              None) == 16  # 3 columns, 2 remaining query ciphertexts
    assert lib.he_pir_remaining_dimensions_device(ours.h, dims, 0, p(database), p(rest), 3, p(key), p(out), None) == 16
    assert lib.he_pir_expand_batch_device(ours.h, p(dim0), 2, 1, n + 1, None, None, 0, p(out), None) == 16  # > N outputs
    # the chunk loop answers chunk by chunk what the one-chunk call answers (all-zero operands: zero response)
    got = ours.pir_compute_response([2, 3], dim0, rest, database, 2, relinearization_key=key)
    torch.cuda.synchronize()
    assert got.shape == (2, 2, 1, n) and int(got.abs().sum()) == 0


def test_column_shard_at_the_benchmark_size(oracle):
    """The per-GPU shard of BASELINE configs[4] at the size bench.py --workload c5 times it: 1024 query ciphertexts x 128
    database columns (34 GB of Eval plaintexts, device-generated) through he_pir_dim0_columns_device -- the lazy inner
    products (Bfv.swift:476-505) and their inverse transforms (PirUtil.swift:428-446).  EVERY column equals the oracle's
    dim-0 step word for word (columns checked side by side on the host's threads; on a host with fewer than 8 threads:
    eight columns spread over the shard -- the first, the last, and ones whose plaintexts start beyond 4 GiB, 268 MB per
    column: from column 16 on, and beyond 32 GiB, column 120 on); every output word is canonical."""
    import torch

    degree, d0, columns = 8192, 1024, 128
    q = oracle.generate_primes([55] * 5, False, degree)
    ours, ref = heamd.BfvContext(degree, 557057, q), oracle.BfvContext(degree, 557057, q)
    moduli = q[:-1]
    gen = torch.Generator(device="cuda")
    gen.manual_seed(640)
    bound = torch.tensor(moduli, dtype=torch.int64, device="cuda").view(1, 1, len(moduli), 1)
    query = torch.randint(0, 1 << 62, (d0, 2, len(moduli), degree), dtype=torch.int64, device="cuda", generator=gen) % bound
    database = torch.empty((columns, d0, len(moduli), degree), dtype=torch.int64, device="cuda")
    for c in range(columns):  # column by column: randint's 268 MB temporaries instead of 34 GB ones
        database[c] = torch.randint(0, 1 << 62, (d0, len(moduli), degree), dtype=torch.int64, device="cuda",
                                    generator=gen) % bound.view(1, len(moduli), 1)
    assert database.numel() * 8 == columns * d0 * len(moduli) * degree * 8 > 34 * 10**9
    out = ours.pir_dim0_columns(query, database)
    assert out.shape == (columns, 2, len(moduli), degree)
    assert bool((out < bound).all()) and bool((out >= 0).all())
    from concurrent.futures import ThreadPoolExecutor

    from conftest import exhaustive_parity, host_threads

    host_query = heamd.to_host(query)
    host_out = heamd.to_host(out)

    def column_matches(c):  # the oracle's C calls release the GIL: columns are checked side by side
        want = oracle.pir.dim0_columns(ref, host_query, heamd.to_host(database[c:c + 1]))
        return np.array_equal(host_out[c], want[0])

    # every column (on a host with fewer than 8 threads: eight spread over the shard, on both sides of 4 GiB and 32 GiB)
    sample = list(range(columns)) if exhaustive_parity() else [0, 1, 15, 16, 17, 64, 120, 127]
    with ThreadPoolExecutor(max_workers=max(1, min(32, host_threads()))) as pool:  # 268 MB of plaintexts per column in flight
        verdicts = list(pool.map(column_matches, sample))
    assert all(verdicts), [c for c, ok in zip(sample, verdicts) if not ok]
    print(f"dim-0 columns: {len(sample)} of {columns} columns ({len(sample) * d0} of {columns * d0} ct x pt products) "
          "compared with the oracle word for word")


def _device_group_chunk_loop(oracle, ours, ref, client, group, members, rng, qctx, host_key, key, one, zero, chunks):
    import torch

    dims = [4, 5]
    d0, columns = dims
    entries = [[rng.randrange(ref.t) for _ in range(ref.degree)] for _ in range(chunks * d0 * columns)]
    database = ref.plaintext_to_eval(np.array(entries, dtype=np.uint64)).reshape(chunks, d0 * columns, ref.L, ref.degree)
    present = np.ones((chunks, d0 * columns), dtype=np.uint8)
    present[1, [0, 13]] = 0
    selection = (1, 3)
    dim0 = np.stack([qctx.forward_ntt(client.encrypt(one if k == selection[0] else zero)) for k in range(d0)])
    rest = np.stack([client.encrypt(one if k == selection[1] else zero) for k in range(columns)])
    dim0_device, rest_device = heamd.to_device(dim0), heamd.to_device(rest)
    database_device = heamd.to_device(database).view(chunks * columns, d0, ref.L, ref.degree)
    present_device = torch.from_numpy(present).cuda().view(chunks * columns, d0)
    shards, masks = [], []
    for m in range(members):
        begin, end = group.bounds(chunks * columns, m)
        shards.append(database_device[begin:end].contiguous() if end > begin else None)
        masks.append(present_device[begin:end].contiguous() if end > begin else None)
    looped = heamd.to_host(group.pir_compute_response(dims, dim0_device, rest_device, shards, chunks, present_shards=masks,
                                                      relinearization_key=key))
    single = heamd.to_host(ours.pir_compute_response(dims, dim0_device, rest_device, heamd.to_device(database), chunks,
                                                     present_device=torch.from_numpy(present).cuda(),
                                                     relinearization_key=key))
    assert np.array_equal(looped, single)
    for chunk in range(chunks):
        expected = oracle.pir.compute_response_for_one_chunk(ref, dims, dim0, rest, database[chunk], present[chunk], host_key)
        assert np.array_equal(looped[chunk], expected), chunk


@pytest.mark.parametrize("members,stage_all", [(1, False), (2, False), (2, True), (3, True)])
def test_device_group_equals_single_device(oracle, small, members, stage_all):
    """he_device_group (include/he_amd.h "Device groups"): the columns of a chunk split over the members of a group -- the
    one GPU listed `members` times, with stage_all every member but the first through the copies of a remote device (query
    in, finished columns out) -- give word for word the single-device dim-0 results and the oracle's chunk response; 5
    columns over 2 / 3 members is a ragged split, 2 columns over 3 an empty shard; nil plaintexts per shard."""
    import torch

    ours, ref, client = small
    group = heamd.DeviceGroup([0] * members, ref.degree, ref.t, ref.coefficient_moduli, stage_all=stage_all)
    assert len(group) == members
    rng = random.Random(500 + members)
    qctx = ref.ciphertext_context()
    host_key = client.relinearization_key()
    key = heamd.to_device(host_key)
    one, zero = [1] + [0] * (ref.degree - 1), [0] * ref.degree
    for dims in ([4, 5], [3, 2]):
        d0, columns = dims
        entries = [[rng.randrange(ref.t) for _ in range(ref.degree)] for _ in range(d0 * columns)]
        database = ref.plaintext_to_eval(np.array(entries, dtype=np.uint64))  # plaintext k of column c at c * d0 + k
        present = np.ones(d0 * columns, dtype=np.uint8)
        present[[1, d0 * columns - 2]] = 0
        selection = (d0 - 1, columns - 1)
        dim0 = np.stack([qctx.forward_ntt(client.encrypt(one if k == selection[0] else zero)) for k in range(d0)])
        rest = np.stack([client.encrypt(one if k == selection[1] else zero) for k in range(columns)])
        dim0_device, rest_device = heamd.to_device(dim0), heamd.to_device(rest)
        database_device = heamd.to_device(database).view(columns, d0, ref.L, ref.degree)
        present_device = torch.from_numpy(present).cuda().view(columns, d0)
        shards, masks = [], []
        for m in range(members):
            begin, end = group.bounds(columns, m)
            assert (begin, end) == heamd.shard_bounds(columns, members, m)
            shards.append(database_device[begin:end].contiguous() if end > begin else None)
            masks.append(present_device[begin:end].contiguous() if end > begin else None)
        whole = ours.pir_dim0_columns(dim0_device, database_device, present_device=present_device)
        split = group.pir_dim0_columns(dim0_device, shards, columns, present_shards=masks)
        torch.cuda.synchronize()
        assert torch.equal(whole, split), dims
        got = heamd.to_host(group.pir_compute_response_chunk(dims, dim0_device, rest_device, shards, present_shards=masks,
                                                             relinearization_key=key))
        expected = oracle.pir.compute_response_for_one_chunk(ref, dims, dim0, rest,
                                                             database.reshape(d0 * columns, ref.L, ref.degree), present,
                                                             host_key)
        assert np.array_equal(got, expected), dims
        index = selection[0] + d0 * selection[1]
        assert client.decrypt(got, moduli_count=1) == (entries[index] if present[index] else zero)
    # the chunk loop over the group: 3 chunks of 4 x 5, their 15 columns one column range sharded over the members; every chunk's
    # response is the single-device chunk loop's and the oracle's.  Then 2 chunks per member: every share is a whole number of
    # chunks, which the members answer from start to finish on their own (device_group.cpp response_by_whole_chunks); with
    # three members the first case is of that kind too.
    for chunks in (3, 2 * members):
        _device_group_chunk_loop(oracle, ours, ref, client, group, members, rng, qctx, host_key, key, one, zero, chunks)
    # a batch of polynomials that lives sharded: every member transforms its own share in place
    batch = 7
    slab = _uniform(np.random.default_rng(9), (batch,), ref.coefficient_moduli[:-1], ref.degree)
    pieces = []
    for m in range(members):
        begin, end = group.bounds(batch, m)
        pieces.append(heamd.to_device(slab[begin:end]) if end > begin else None)
    group.forward_ntt_(pieces, batch)
    group.synchronize()
    forward = np.concatenate([heamd.to_host(p) for p in pieces if p is not None])
    assert np.array_equal(forward, qctx.forward_ntt(slab))
    group.inverse_ntt_(pieces, batch)
    group.synchronize()
    assert np.array_equal(np.concatenate([heamd.to_host(p) for p in pieces if p is not None]), slab)
    with pytest.raises(heamd.HeError):
        heamd.DeviceGroup([], ref.degree, ref.t, ref.coefficient_moduli)
    with pytest.raises(heamd.HeError):
        heamd.DeviceGroup([99], ref.degree, ref.t, ref.coefficient_moduli)


def test_device_group_at_the_benchmark_ring(oracle):
    """The group path on the production kernels (N = 8192, L = 4: the replica-set inner product, the tiled inverse transform, the
    row-fused ct x ct of the remaining dimension): 3 members on the one GPU, every member but the first through the staging
    copies, 16 columns of 32 rows (a ragged 6 / 5 / 5 split) -- the gathered columns equal the single-device call's word for
    word, the chunk response the single-device chunk response."""
    import torch

    degree, d0, columns = 8192, 32, 16
    q = oracle.generate_primes([55] * 5, False, degree)
    t = oracle.generate_primes([20], True, degree)[0]
    ours = heamd.BfvContext(degree, t, q)
    group = heamd.DeviceGroup([0, 0, 0], degree, t, q, stage_all=True)
    rng = np.random.default_rng(77)
    moduli = q[:-1]
    dim0 = heamd.to_device(_uniform(rng, (d0, 2), moduli, degree))
    rest = heamd.to_device(_uniform(rng, (columns, 2), moduli, degree))
    key = heamd.to_device(_uniform(rng, (ours.L, 2), q, degree))
    database = heamd.to_device(_uniform(rng, (columns, d0), moduli, degree))
    present = (torch.arange(columns * d0, device="cuda") % 7 != 3).to(torch.uint8).view(columns, d0)
    shards, masks = [], []
    for m in range(3):
        begin, end = group.bounds(columns, m)
        shards.append(database[begin:end].contiguous())
        masks.append(present[begin:end].contiguous())
    assert [s.shape[0] for s in shards] == [6, 5, 5]
    whole = ours.pir_dim0_columns(dim0, database, present_device=present)
    split = group.pir_dim0_columns(dim0, shards, columns, present_shards=masks)
    torch.cuda.synchronize()
    assert torch.equal(whole, split)
    single = ours.pir_compute_response_chunk([d0, columns], dim0, rest, database.view(columns * d0, ours.L, degree),
                                             present.view(-1).cpu().numpy(), key)
    grouped = group.pir_compute_response_chunk([d0, columns], dim0, rest, shards, present_shards=masks, relinearization_key=key)
    torch.cuda.synchronize()
    assert torch.equal(single, grouped)
