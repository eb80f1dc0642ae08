"""GPU parity tests of the element-wise PolyRq kernels, divideAndRoundQLast and the lazy accumulators.

Mirrors Tests/HomomorphicEncryptionTests/PolyRqTests/PolyRqTests.swift:45-176.  Bit-exact.
"""
import numpy as np
import pytest

import heamd

pytestmark = pytest.mark.gpu


def _rand_slab(rng, batch, moduli, degree):
    rows = [rng.integers(0, q, size=(batch, degree), dtype=np.uint64) for q in moduli]
    return np.ascontiguousarray(np.stack(rows, axis=1))


def test_poly_ops_known_answers(kats):
    k = kats["poly_ops"]
    ctx = heamd.PolyContext(k["degree"], k["moduli"])
    x = np.array(k["x"], dtype=np.uint64)
    dev = heamd.to_device
    assert np.array_equal(heamd.to_host(ctx.add_(dev(x), dev(x))), np.array(k["add_x_x"], dtype=np.uint64))
    assert np.array_equal(heamd.to_host(ctx.sub_(dev(np.zeros_like(x)), dev(x))),
                          np.array(k["zero_minus_x"], dtype=np.uint64))
    assert np.array_equal(heamd.to_host(ctx.neg_(dev(x))), np.array(k["neg_x"], dtype=np.uint64))
    y = np.array(k["mul_y"], dtype=np.uint64)
    assert np.array_equal(heamd.to_host(ctx.mul_(dev(x), dev(y))), np.array(k["mul_x_y"], dtype=np.uint64))
    s = k["scalar"]
    expected = np.array([[(int(v) * s) % q for v in row] for row, q in zip(k["x"], k["moduli"])], dtype=np.uint64)
    assert np.array_equal(heamd.to_host(ctx.mul_scalar_(dev(x), [s % q for q in k["moduli"]])), expected)


@pytest.mark.parametrize("degree,bits,batch", [(2, [20], 1), (16, [55, 52, 62, 58], 7), (4096, [55, 55], 3),
                                               (8192, [61, 62, 33, 55, 40], 2)])
def test_elementwise_matches_oracle(oracle, degree, bits, batch):
    moduli = oracle.generate_primes(bits, False, degree)
    ours = heamd.PolyContext(degree, moduli)
    ref = oracle.PolyContext(degree, moduli)
    rng = np.random.default_rng(degree + batch)
    x, y = _rand_slab(rng, batch, moduli, degree), _rand_slab(rng, batch, moduli, degree)
    x[0, 0, :2] = 0
    y[0, 0, :2] = [0, moduli[0] - 1]
    x[0, -1, -2:] = moduli[-1] - 1
    y[0, -1, -2:] = moduli[-1] - 1
    dev = heamd.to_device
    assert np.array_equal(heamd.to_host(ours.add_(dev(x), dev(y))), ref.add(x, y))
    assert np.array_equal(heamd.to_host(ours.sub_(dev(x), dev(y))), ref.sub(x, y))
    assert np.array_equal(heamd.to_host(ours.neg_(dev(x))), ref.neg(x))
    assert np.array_equal(heamd.to_host(ours.mul_(dev(x), dev(y))), ref.mul(x, y))
    scalars = [int(rng.integers(0, q)) for q in moduli]
    assert np.array_equal(heamd.to_host(ours.mul_scalar_(dev(x), scalars)), ref.mul_scalar(x, scalars))


def test_divide_and_round_q_last_known_answers(kats):
    for c in kats["divide_and_round_q_last"]["cases"]:
        ctx = heamd.PolyContext(c["degree"], c["moduli"])
        x = np.array(c["x"], dtype=np.uint64)
        got = heamd.to_host(ctx.divide_and_round_q_last(heamd.to_device(x)))
        assert np.array_equal(got[0], np.array(c["expected"], dtype=np.uint64)), c
        assert np.array_equal(ctx.divide_and_round_q_last_host(x)[0], np.array(c["expected"], dtype=np.uint64))


@pytest.mark.parametrize("degree,bits,batch", [(8, [40, 45, 50], 3), (4096, [55, 55], 4), (8192, [33, 61, 62, 55], 3),
                                               (16384, [55] * 6, 2)])
def test_divide_and_round_q_last_matches_oracle(oracle, degree, bits, batch):
    moduli = oracle.generate_primes(bits, False, degree)
    ours = heamd.PolyContext(degree, moduli)
    ref = oracle.PolyContext(degree, moduli)
    rng = np.random.default_rng(degree)
    x = _rand_slab(rng, batch, moduli, degree)
    x[0, -1, :4] = [0, moduli[-1] - 1, moduli[-1] // 2, moduli[-1] // 2 + 1]
    got = heamd.to_host(ours.divide_and_round_q_last(heamd.to_device(x)))
    assert np.array_equal(got, ref.divide_and_round_q_last(x))


def test_divide_and_round_needs_a_next_context(oracle):
    ctx = heamd.PolyContext(8, oracle.generate_primes([30], False, 8))
    with pytest.raises(heamd.HeError) as err:
        ctx.divide_and_round_q_last(heamd.to_device(np.zeros((1, 1, 8), dtype=np.uint64)))
    assert err.value.name == "invalidPolyContext"


def test_mod_switch_full_size_property(oracle):
    """BASELINE config 4 shape (N=16384, 6 -> 5 moduli) on 512 polynomials: a sample matches the oracle and the
    result is canonical; scaling the input by q_last leaves x / q_last exactly (round(q_last*y / q_last) == y)."""
    import torch

    degree, batch = 16384, 512
    moduli = oracle.generate_primes([55] * 6, False, degree)
    ours = heamd.PolyContext(degree, moduli)
    ref = oracle.PolyContext(degree, moduli)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(4)
    bound = torch.tensor(moduli, dtype=torch.int64, device="cuda").view(1, len(moduli), 1)
    x = torch.randint(0, 1 << 62, (batch, len(moduli), degree), dtype=torch.int64, device="cuda", generator=gen) % bound
    out = ours.divide_and_round_q_last(x)
    assert int((out >= bound[:, :-1]).sum()) == 0 and int((out < 0).sum()) == 0
    sample = [0, 5, 255, 511]
    assert np.array_equal(heamd.to_host(out[sample]), ref.divide_and_round_q_last(heamd.to_host(x[sample])))
    # x := q_last * y (in RNS: last residue 0, others y_i * q_last mod q_i)  ==>  divideAndRound(x) == y
    y = x[:, :-1, :].contiguous()
    scaled = torch.zeros_like(x)
    scaled[:, :-1, :] = y
    sub = heamd.PolyContext(degree, moduli[:-1])
    tmp = scaled[:, :-1, :].contiguous()
    sub.mul_scalar_(tmp, [moduli[-1] % q for q in moduli[:-1]])
    scaled[:, :-1, :] = tmp
    assert torch.equal(ours.divide_and_round_q_last(scaled), y)


def test_lazy_product_accumulation(oracle):
    degree = 4096
    moduli = oracle.generate_primes([59, 60], False, degree)
    ours = heamd.PolyContext(degree, moduli)
    ref = oracle.PolyContext(degree, moduli)
    rng = np.random.default_rng(7)
    import torch

    acc_dev = torch.zeros((2, degree, 2), dtype=torch.int64, device="cuda")
    acc_ref = np.zeros((2, degree, 2), dtype=np.uint64)
    for _ in range(5):
        x, y = _rand_slab(rng, 1, moduli, degree)[0], _rand_slab(rng, 1, moduli, degree)[0]
        ours.adding_lazy_product_(heamd.to_device(x), heamd.to_device(y), acc_dev)
        ref.adding_lazy_product(x, y, acc_ref)
    assert np.array_equal(heamd.to_host(acc_dev), acc_ref)
    assert np.array_equal(heamd.to_host(ours.reduce_accumulator(acc_dev)), ref.reduce_accumulator(acc_ref))
    # a full-range 128-bit accumulator (wrapping arithmetic + double-word Barrett on arbitrary input)
    wild = rng.integers(0, 1 << 63, size=(2, degree, 2), dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    assert np.array_equal(heamd.to_host(ours.reduce_accumulator(heamd.to_device(wild))), ref.reduce_accumulator(wild))


def test_mod_switch_at_the_benchmark_size(oracle):
    """BASELINE configs[3] at the size bench.py times it: N=16384, 6 -> 5 moduli, 8192 polynomials (6 GiB in, 5 GiB out,
    device-generated).  EVERY polynomial equals the oracle's divideAndRoundQLast (PolyRq.swift:365-393) word for word
    (multi-threaded oracle, 512 polynomials at a time; on a host with fewer than 8 threads: polynomials sampled over the
    whole slab -- the first, the last, and several whose input AND output byte offsets lie beyond 4 GiB, 786 432 B per
    input polynomial: from index 5462 on; 655 360 B per output polynomial: from index 6554 on), and every output word is
    canonical."""
    import torch

    degree, batch = 16384, 8192
    moduli = oracle.generate_primes([55] * 6, False, degree)
    ours = heamd.PolyContext(degree, moduli)
    ref = oracle.PolyContext(degree, moduli)
    gen = torch.Generator(device="cuda")
    gen.manual_seed(404)
    bound = torch.tensor(moduli, dtype=torch.int64, device="cuda").view(1, len(moduli), 1)
    x = torch.randint(0, 1 << 62, (batch, len(moduli), degree), dtype=torch.int64, device="cuda", generator=gen) % bound
    assert x.numel() * 8 == 6 << 30
    out = ours.divide_and_round_q_last(x)
    assert out.shape == (batch, len(moduli) - 1, degree)
    for part in range(0, batch, 1024):  # canonical everywhere (in slices: the comparison's temporaries stay small)
        piece = out[part:part + 1024]
        assert int((piece >= bound[:, :-1]).sum()) == 0 and int((piece < 0).sum()) == 0
    from conftest import exhaustive_parity, host_threads

    if exhaustive_parity():
        compared = 0
        for first in range(0, batch, 512):  # 384 MiB in, 320 MiB out per slice
            want = ref.divide_and_round_q_last(heamd.to_host(x[first:first + 512]), threads=host_threads())
            assert np.array_equal(heamd.to_host(out[first:first + 512]), want), first
            compared += want.shape[0]
        assert compared == batch
        print(f"divideAndRoundQLast: {compared} of {batch} polynomials compared with the oracle word for word")
    else:
        sample = [0, 1, 2047, 4096, 5461, 5462, 6553, 6554, 7000, 8190, 8191]
        assert sample[5] * len(moduli) * degree * 8 > 1 << 32 and sample[7] * (len(moduli) - 1) * degree * 8 > 1 << 32
        assert np.array_equal(heamd.to_host(out[sample]), ref.divide_and_round_q_last(heamd.to_host(x[sample])))
        print(f"divideAndRoundQLast: {len(sample)} of {batch} polynomials compared with the oracle (small host)")
