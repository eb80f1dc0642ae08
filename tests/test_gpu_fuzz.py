"""Seeded sweep over parameter shapes no hand-written case names: random moduli sizes (40..62 bits, mixed), 1..5
ciphertext moduli, degrees with and without tiled transforms -- NTT, mod-switch, ct x ct, relinearize, the Galois key
switch (fused and composed paths), the ct x pt inner product for one and several queries, masked, from 8-byte and from
packed plaintexts -- word for word against the oracle.  Catches mode-selection mistakes (headroom / [0, 8p) / exact butterflies, mixed [Q, Bsk] bands, fused
loads) that depend on how the moduli happen to line up."""
import os
import random

import numpy as np
import pytest

import heamd

pytestmark = pytest.mark.gpu

SIZES = [40, 45, 50, 54, 55, 58, 60, 61, 62]


def _uniform(rng, prefix, moduli, degree):
    rows = [rng.integers(0, q, size=tuple(prefix) + (degree,), dtype=np.uint64) for q in moduli]
    return np.ascontiguousarray(np.stack(rows, axis=len(prefix)))


# HEAMD_FUZZ_SEEDS="1,2,3,..." widens the sweep for a one-off hunt
SEEDS = [int(v) for v in os.environ.get("HEAMD_FUZZ_SEEDS", "11,23,47").split(",")]


@pytest.mark.parametrize("seed", SEEDS)
def test_random_parameter_shapes(oracle, seed):
    rnd = random.Random(seed)
    for trial in range(6):
        degree = rnd.choice([256, 4096, 8192, 16384])
        L = rnd.randint(1, 5)
        bits = [rnd.choice(SIZES) for _ in range(L + 1)]
        q = oracle.generate_primes(bits, False, degree)
        t = oracle.generate_primes([17], True, degree)[0]
        ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
        rng = np.random.default_rng(seed * 100 + trial)
        moduli = q[:-1]
        label = (degree, bits)
        x = _uniform(rng, (3,), moduli, degree)
        pc, rc = heamd.PolyContext(degree, moduli), oracle.PolyContext(degree, moduli)
        assert np.array_equal(heamd.to_host(pc.forward_ntt_(heamd.to_device(x))), rc.forward_ntt(x)), label
        assert np.array_equal(heamd.to_host(pc.inverse_ntt_(heamd.to_device(x))), rc.inverse_ntt(x)), label
        if L >= 2:
            assert np.array_equal(heamd.to_host(pc.divide_and_round_q_last(heamd.to_device(x))),
                                  rc.divide_and_round_q_last(x)), label
        lhs, rhs = _uniform(rng, (2, 2), moduli, degree), _uniform(rng, (2, 2), moduli, degree)
        product = heamd.to_host(ours.mul(heamd.to_device(lhs), heamd.to_device(rhs)))
        assert np.array_equal(product, ref.mul(lhs, rhs)), label
        key = _uniform(rng, (L, 2), q, degree)
        relin = heamd.to_host(ours.relinearize(heamd.to_device(product), heamd.to_device(key)))
        assert np.array_equal(relin, ref.relinearize(product, key)), label
        # Bfv.applyGalois: sign-only, reversing and scattering elements (fused path on the tiled degrees)
        element = rnd.choice([degree + 1, 2 * degree - 1, 3, degree // 2 + 1, 2 * rnd.randrange(1, degree) + 1])
        cts = _uniform(rng, (2, 2), moduli, degree)
        rotated = heamd.to_host(ours.apply_galois(heamd.to_device(cts), element, heamd.to_device(key)))
        assert np.array_equal(rotated, ref.apply_galois(cts, element, key)), (label, element)
        # Bfv.addAssignCoeff / subAssignCoeff(ciphertext, plaintext)
        messages = rng.integers(0, t, size=(2, degree), dtype=np.uint64)
        subtract = bool(rnd.getrandbits(1))
        translated = heamd.to_host(ours.add_plain_(heamd.to_device(cts), heamd.to_device(messages), 2, subtract))
        assert np.array_equal(translated, ref.plaintext_translate(cts, messages, 2, subtract)), (label, subtract)
        # Bfv.innerProduct(ciphertexts:plaintexts:) with nil plaintexts, for 1..4 queries side by side
        queries = rnd.choice([1, 2, 3, 4])
        count, columns = rnd.randint(1, 70), rnd.randint(1, 9)
        vector = _uniform(rng, (count, queries, 2), moduli, degree)
        plaintexts = _uniform(rng, (columns, count), moduli, degree)
        present = rng.integers(0, 4, size=(columns, count), dtype=np.uint8).clip(0, 1)
        device_pts = heamd.to_device(plaintexts)
        got = heamd.to_host(ours.inner_product_plain(heamd.to_device(vector), device_pts, present, 2 * queries, columns))
        got = got.reshape(columns, queries, 2, L, degree)
        query, column = rnd.randrange(queries), rnd.randrange(columns)
        own = np.ascontiguousarray(vector[:, query])
        expected = ref.inner_product_plain(own, plaintexts[column], present[column])
        assert np.array_equal(got[column, query], expected), (label, queries, count, columns)
        if queries == 1 and all(b <= 61 for b in bits[:-1]):  # the packed path wants sums of two products below 2^127
            import torch

            packed = ours.pack_plaintexts(device_pts)
            from_packed = ours.inner_product_plain_packed(heamd.to_device(own), packed, torch.from_numpy(present).cuda(), 2,
                                                          columns)
            assert np.array_equal(heamd.to_host(from_packed)[column], expected), (label, "packed")


@pytest.mark.parametrize("seed", SEEDS)
def test_random_row_fused_products(oracle, monkeypatch, seed):
    """ct x ct through behz_kernels.hip (batches from half a workgroup generation of [Q, Bsk] rows up, N = 4096 / 8192) on random
    moduli mixes -- which rows form a run of one butterfly class (limb-wise | fold - | fold + | [0, 8p) | exact for the whole
    record), where the Q rows end inside or at the edge of a run -- random levels and ragged batches; every word of every
    product against the multi-threaded oracle, extremes of every residue included."""
    from conftest import host_threads

    monkeypatch.setenv("HEAMD_BEHZ_FUSED_ABOVE", "256")  # (production takes these kernels from 1152 workgroups on)
    rnd = random.Random(7000 + seed)
    for trial in range(3):
        degree = rnd.choice([4096, 8192])
        top = rnd.randint(1, 4)
        bits = [rnd.choice(SIZES) for _ in range(top + 1)]
        if rnd.random() < 0.3:
            bits[rnd.randrange(top)] = rnd.choice([29, 33])  # one small modulus (there are few such primes: at most one)
        q = oracle.generate_primes(bits, False, degree)
        t = oracle.generate_primes([rnd.choice([17, 20])], True, degree)[0]
        ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
        L = rnd.randint(1, top)
        rows = 2 * L + 1
        batch = 256 // rows + 1 + rnd.randint(0, 9)  # just past the threshold of the fused kernel
        moduli = ref.ciphertext_context(L).moduli
        rng = np.random.default_rng(seed * 1000 + trial)
        lhs, rhs = _uniform(rng, (batch, 2), moduli, degree), _uniform(rng, (batch, 2), moduli, degree)
        for i, m in enumerate(moduli):
            lhs[-1, :, i, :] = m - 1
            rhs[-1, :, i, :] = m - 1
            lhs[0, 0, i, ::2] = 0
        got = heamd.to_host(ours.mul(heamd.to_device(lhs), heamd.to_device(rhs), L))
        assert np.array_equal(got, ref.mul(lhs, rhs, L, threads=host_threads())), (degree, bits, L, batch)


@pytest.mark.parametrize("seed", SEEDS)
def test_random_expansions(oracle, seed):
    """PirUtil.expand over random output counts and random subsets of Galois keys (levels with their own key take the
    fused path, the others reach their element by repeated application; leaves at every depth, doubled or not), one and
    two queries per call, on a ring with a tiled transform: the same words and order as the oracle's recursion."""
    rnd = random.Random(seed)
    degree = 4096
    q = oracle.generate_primes([rnd.choice(SIZES), rnd.choice(SIZES), 55], False, degree)
    ours, ref = heamd.BfvContext(degree, 65537, q), oracle.BfvContext(degree, 65537, q)
    rng = np.random.default_rng(seed)
    for trial in range(3):
        total = rnd.randint(1, 40)
        height = max((total - 1).bit_length(), 1)
        shifts = {height - 1} | {k for k in range(height) if rnd.random() < 0.6}  # the deepest level's element always
        queries = _uniform(rng, (2, 1, 2), q[:-1], degree)
        keys = [{(degree >> k) + 1: _uniform(rng, (ours.L, 2), q, degree) for k in shifts} for _ in range(2)]
        device_keys = [{e: heamd.to_device(k) for e, k in keys[i].items()} for i in range(2)]
        expected = [oracle.pir.expand(ref, queries[i], total, keys[i]) for i in range(2)]
        got = heamd.to_host(ours.pir_expand(heamd.to_device(queries[0]), total, device_keys[0]))
        assert np.array_equal(got, expected[0]), (total, sorted(shifts))
        both = heamd.to_host(ours.pir_expand_batch(heamd.to_device(queries), total, device_keys))
        assert np.array_equal(both[0], expected[0]) and np.array_equal(both[1], expected[1]), (total, sorted(shifts))


@pytest.mark.parametrize("seed", SEEDS)
def test_random_pir_shapes(oracle, seed):
    """PIR responses over random database shapes (1-3 dimensions, 1-3 chunks, nil plaintexts), for one query and for 2-4
    queries sharing the database pass, from 8-byte and from packed plaintexts: every response equals the oracle's
    computeResponseForOneChunk for that query and chunk."""
    import torch

    rnd = random.Random(seed)
    degree = 256
    L = rnd.randint(2, 3)
    q = oracle.generate_primes([rnd.choice(SIZES[:7]) for _ in range(L + 1)], False, degree)
    t = oracle.generate_primes([17], True, degree)[0]
    ours, ref = heamd.BfvContext(degree, t, q), oracle.BfvContext(degree, t, q)
    rng = np.random.default_rng(seed)
    moduli = q[:-1]
    for trial in range(3):
        # the reference asks for columns == 1 or columns == remaining query count (PirUtil.swift:422): one or two
        # dimensions, or [d0, 2, 2]
        dims = [rnd.randint(1, 6)] + rnd.choice([[], [rnd.randint(1, 4)], [rnd.randint(1, 4)], [2, 2]])
        per_chunk, chunks, queries = int(np.prod(dims)), rnd.randint(1, 3), rnd.randint(1, 4)
        rest_count = sum(dims[1:])
        database = _uniform(rng, (chunks, per_chunk), moduli, degree)
        present = rng.integers(0, 5, size=(chunks, per_chunk), dtype=np.uint8).clip(0, 1)
        dim0 = _uniform(rng, (dims[0], queries, 2), moduli, degree)
        rest = _uniform(rng, (queries, max(rest_count, 1), 2), moduli, degree)[:, :rest_count]
        keys = [_uniform(rng, (ours.L, 2), q, degree) for _ in range(queries)]
        device_db, mask = heamd.to_device(database), torch.from_numpy(present).cuda()
        device_keys = [heamd.to_device(k) for k in keys] if rest_count else None
        label = (dims, chunks, queries)
        got = heamd.to_host(ours.pir_compute_response_queries(
            dims, heamd.to_device(dim0), heamd.to_device(np.ascontiguousarray(rest)) if rest_count else None, device_db,
            chunks, device_keys, present_device=mask))
        query = rnd.randrange(queries)
        own_dim0 = np.ascontiguousarray(dim0[:, query])
        own_rest = np.ascontiguousarray(rest[query])
        for chunk in range(chunks):
            expected = oracle.pir.compute_response_for_one_chunk(ref, dims, own_dim0, own_rest, database[chunk],
                                                                 present[chunk], keys[query] if rest_count else None)
            assert np.array_equal(got[query, chunk], expected), (label, query, chunk)
        single_args = (dims, heamd.to_device(own_dim0), heamd.to_device(own_rest) if rest_count else None)
        single_key = heamd.to_device(keys[query]) if rest_count else None
        single = ours.pir_compute_response(*single_args, device_db, chunks, present_device=mask, relinearization_key=single_key)
        assert np.array_equal(heamd.to_host(single), got[query]), label
        packed = ours.pir_compute_response_packed(*single_args, ours.pack_plaintexts(device_db), chunks, present_device=mask,
                                                  relinearization_key=single_key)
        assert bool((packed == single).all()), label


@pytest.mark.parametrize("seed", SEEDS)
def test_random_wire_formats(oracle, seed):
    """PolyRq.serialize / deserialize over random field widths (8..62-bit moduli, skipLSBs 0..3), degrees on both sides
    of the tiled kernels' threshold and batch sizes; seeded polynomials over random batches: byte for byte / word for
    word against the oracle."""
    import torch

    rnd = random.Random(seed)
    for trial in range(4):
        degree = rnd.choice([32, 64, 128, 256, 1024, 4096])
        bits = [rnd.randint(8, 62) for _ in range(rnd.randint(1, 4))]
        moduli = oracle.generate_primes(bits, False, 1)
        ours, ref = heamd.PolyContext(degree, moduli), oracle.PolyContext(degree, moduli)
        rng = np.random.default_rng(seed * 10 + trial)
        batch = rnd.randint(1, 5)
        slab = _uniform(rng, (batch,), moduli, degree)
        skip = rnd.randint(0, min(3, min(bits) - 2))
        label = (degree, bits, skip, batch)
        packed = ours.serialize(heamd.to_device(slab), skip)
        expected = ref.serialize(slab, skip)
        assert np.array_equal(packed.cpu().numpy(), expected), label
        back = heamd.to_host(ours.deserialize(torch.from_numpy(expected).cuda(), skip))
        assert np.array_equal(back, ref.deserialize(expected, skip)), label
    degree = rnd.choice([64, 512, 4096])
    moduli = oracle.generate_primes([rnd.randint(20, 62) for _ in range(rnd.randint(1, 3))], False, degree)
    ours, ref = heamd.PolyContext(degree, moduli), oracle.PolyContext(degree, moduli)
    seeds = np.random.default_rng(seed).integers(0, 256, size=(rnd.randint(1, 20), 32), dtype=np.uint8)
    got = heamd.to_host(ours.random_from_seeds(torch.from_numpy(seeds).cuda()))
    assert np.array_equal(got, ref.random_from_seeds(seeds)), (degree, len(seeds))


@pytest.mark.parametrize("degree,bits", [(8192, [30, 28, 29, 30]), (16384, [27, 30, 30]), (4096, [30, 20, 30, 24, 30])])
def test_uint32_fused_transforms_at_every_tiled_degree(oracle, degree, bits):
    """The 4-byte transforms with fused loads (word32_kernels.hip Source32: key-switching decomposition into the forward
    transform, tensor product and key inner product into the inverse one) at each degree that has a tiled 4-byte kernel,
    with moduli of mixed sizes (a decomposed row is reduced where its modulus is the larger one) and an odd batch: ct x ct,
    relinearize and the Galois key switch word for word against the 32-bit oracle."""
    dev, host = heamd.to_device32, heamd.to_host32
    q = oracle.generate_primes(bits, False, degree, word_bits=32)
    t = oracle.generate_primes([degree.bit_length() + 6], True, degree, word_bits=32)[0]
    ours, ref = heamd.BfvContext32(degree, t, q), oracle.BfvContext(degree, t, q, word_bits=32)
    rng = np.random.default_rng(degree + len(bits))
    moduli, L = q[:-1], len(q) - 1
    lhs, rhs = _uniform(rng, (3, 2), moduli, degree), _uniform(rng, (3, 2), moduli, degree)
    lhs[0] = np.array(moduli, dtype=np.uint64)[None, :, None] - np.uint64(1)
    rhs[1] = 0
    product = host(ours.mul(dev(lhs), dev(rhs)))
    assert np.array_equal(product, ref.mul(lhs, rhs))
    key = _uniform(rng, (L, 2), q, degree)
    assert np.array_equal(host(ours.relinearize(dev(product), dev(key))), ref.relinearize(product, key))
    element = 2 * 1234 + 1
    assert np.array_equal(host(ours.apply_galois(dev(lhs), element, dev(key))), ref.apply_galois(lhs, element, key))


@pytest.mark.parametrize("degree,bits,batch", [(4096, [27, 28, 28], 352), (8192, [30, 20, 30, 29], 264)])
def test_uint32_key_switch_ends_in_the_transform_store(oracle, degree, bits, batch):
    """Bfv<UInt32> relinearize and the (in-place) Galois key switch on batches beyond two workgroup generations of
    key-switching rows: there the key switch ends in the 4-byte key-MAC transform's store (word32_kernels.hip
    kSource32KeyMacFinish) -- the reference's n_4096_logq_27_28_28 moduli, and a set whose special modulus exceeds a 20-bit
    ciphertext modulus (the centred q_ks word is reduced first).  Word for word against the 32-bit oracle."""
    dev, host = heamd.to_device32, heamd.to_host32
    q = oracle.generate_primes(bits, False, degree, word_bits=32)
    t = oracle.generate_primes([degree.bit_length() + 4], True, degree, word_bits=32)[0]
    ours, ref = heamd.BfvContext32(degree, t, q), oracle.BfvContext(degree, t, q, word_bits=32)
    rng = np.random.default_rng(degree + batch)
    moduli, L = q[:-1], len(q) - 1
    ct3 = _uniform(rng, (batch, 3), moduli, degree)
    key = _uniform(rng, (L, 2), q, degree)
    assert np.array_equal(host(ours.relinearize(dev(ct3), dev(key))), ref.relinearize(ct3, key, threads=16))
    ct = _uniform(rng, (batch, 2), moduli, degree)
    element = 2 * degree - 1
    assert np.array_equal(host(ours.apply_galois(dev(ct), element, dev(key))), ref.apply_galois(ct, element, key))


@pytest.mark.parametrize("seed", SEEDS)
def test_random_uint32_shapes(oracle, seed):
    """Bfv<UInt32> on packed 4-byte slabs over random parameter shapes (17..30-bit moduli, 1..4 ciphertext moduli, degrees
    with and without tiled 4-byte transforms): base conversions, ct x ct, relinearize, the Galois key switch, mod-switch,
    both inner products -- word for word against the 32-bit oracle."""
    import torch

    rnd = random.Random(seed)
    dev, host = heamd.to_device32, heamd.to_host32
    for trial in range(4):
        degree = rnd.choice([64, 1024, 4096])
        L = rnd.randint(1, 4)
        log_2n = degree.bit_length()  # NTT-friendly primes are 1 mod 2N: leave room for a few of each size
        bits = [rnd.randint(max(17, log_2n + 7), 30) for _ in range(L + 1)]
        q = oracle.generate_primes(bits, False, degree, word_bits=32)
        t = oracle.generate_primes([log_2n + 4], True, degree, word_bits=32)[0]
        ours, ref = heamd.BfvContext32(degree, t, q), oracle.BfvContext(degree, t, q, word_bits=32)
        rng = np.random.default_rng(seed * 100 + trial)
        moduli = q[:-1]
        label = (degree, bits)
        tool = ref.rns_tool(L)
        x = _uniform(rng, (2,), moduli, degree)
        assert np.array_equal(host(ours.lift_q_to_qbsk(dev(x), L)), np.stack([tool.lift_q_to_qbsk(p) for p in x])), label
        y = _uniform(rng, (2,), ref.qbsk_context(L).moduli, degree)
        assert np.array_equal(host(ours.floor_qbsk_to_q(dev(y), L)), np.stack([tool.floor_qbsk_to_q(p) for p in y])), label
        assert np.array_equal(host(ours.scale_and_round(dev(x), 1)), np.stack([tool.scale_and_round(p, 1) for p in x])), label
        lhs, rhs = _uniform(rng, (2, 2), moduli, degree), _uniform(rng, (2, 2), moduli, degree)
        product = host(ours.mul(dev(lhs), dev(rhs)))
        assert np.array_equal(product, ref.mul(lhs, rhs)), label
        key = _uniform(rng, (L, 2), q, degree)
        relin = host(ours.relinearize(dev(product), dev(key)))
        assert np.array_equal(relin, ref.relinearize(product, key)), label
        element = 2 * rnd.randrange(1, degree) + 1
        assert np.array_equal(host(ours.apply_galois(dev(lhs), element, dev(key))), ref.apply_galois(lhs, element, key)), label
        if L >= 2:
            assert np.array_equal(host(ours.mod_switch_down(dev(lhs), 2)), ref.mod_switch_down(lhs, poly_count=2)), label
        messages = rng.integers(0, t, size=(2, degree), dtype=np.uint64)
        subtract = bool(rnd.getrandbits(1))
        assert np.array_equal(host(ours.add_plain_(dev(lhs), dev(messages), 2, subtract)),
                              ref.plaintext_translate(lhs, messages, 2, subtract)), (label, subtract)
        count, columns = rnd.randint(1, 30), rnd.randint(1, 6)
        if L >= 2:
            assert np.array_equal(host(ours.mod_switch_down(dev(lhs), 2)), ref.mod_switch_down(lhs, poly_count=2)), label
        count, columns = rnd.randint(1, 30), rnd.randint(1, 6)
        vector, other = _uniform(rng, (count, 2), moduli, degree), _uniform(rng, (count, 2), moduli, degree)
        assert np.array_equal(host(ours.inner_product(dev(vector), dev(other))), ref.inner_product(vector, other)), label
        plaintexts = _uniform(rng, (columns, count), moduli, degree)
        present = rng.integers(0, 4, size=(columns, count), dtype=np.uint8).clip(0, 1)
        queries = rnd.choice([1, 2, 3, 4])  # several queries' vectors side by side share the plaintext stream
        side_by_side = _uniform(rng, (count, queries, 2), moduli, degree)
        got = host(ours.inner_product_plain_resident(dev(side_by_side), dev(plaintexts), torch.from_numpy(present).cuda(),
                                                     2 * queries, columns)).reshape(columns, queries, 2, L, degree)
        column, query = rnd.randrange(columns), rnd.randrange(queries)
        own = np.ascontiguousarray(side_by_side[:, query])
        assert np.array_equal(got[column, query], ref.inner_product_plain(own, plaintexts[column], present[column])), label
        # the PIR chunk loop on the same packed slabs
        dims = [rnd.randint(1, 4)] + rnd.choice([[], [rnd.randint(1, 3)], [2, 2]])
        per_chunk, chunks, rest_count = int(np.prod(dims)), rnd.randint(1, 2), sum(dims[1:])
        database = _uniform(rng, (chunks, per_chunk), moduli, degree)
        mask = rng.integers(0, 5, size=(chunks, per_chunk), dtype=np.uint8).clip(0, 1)
        dim0 = _uniform(rng, (dims[0], 2), moduli, degree)
        rest = _uniform(rng, (max(rest_count, 1), 2), moduli, degree)[:rest_count]
        response = host(ours.pir_compute_response(dims, dev(dim0), dev(rest) if rest_count else None, dev(database), chunks,
                                                  present_device=torch.from_numpy(mask).cuda(),
                                                  relinearization_key=dev(key) if rest_count else None))
        for chunk in range(chunks):
            expected = oracle.pir.compute_response_for_one_chunk(ref, dims, dim0, rest, database[chunk], mask[chunk],
                                                                 key if rest_count else None)
            assert np.array_equal(response[chunk], expected), (label, dims, chunk)
