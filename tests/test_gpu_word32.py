"""PolyRq<UInt32> on the device (polynomial layer): 4-byte-word slabs, bit-exact against the oracle
(whose canonical words do not depend on the word type) and the reference's small NTT KATs."""
import numpy as np
import pytest

import heamd

pytestmark = pytest.mark.gpu


def _to_device32(array):
    import torch

    return torch.from_numpy(np.ascontiguousarray(array, dtype=np.uint32).view(np.int32)).cuda()


def _to_host32(tensor):
    return tensor.cpu().numpy().view(np.uint32).astype(np.uint64)


def _slab(rng, batch, moduli, degree):
    return np.stack([rng.integers(0, q, size=(batch, degree), dtype=np.uint64) for q in moduli], axis=1).copy()


def test_ntt_known_answers_u32(kats):
    checked = 0
    for case in kats["ntt"]["cases"]:  # NttTests.swift:72-191
        if max(case["moduli"]) > (1 << 30) - 1:
            continue
        coeff = np.array(case["coeff"], dtype=np.uint64)[None]
        evaluated = np.array(case["eval"], dtype=np.uint64)[None]
        ctx = heamd.PolyContext(coeff.shape[2], case["moduli"])
        assert np.array_equal(_to_host32(ctx.forward_ntt_u32_(_to_device32(coeff))), evaluated)
        assert np.array_equal(_to_host32(ctx.inverse_ntt_u32_(_to_device32(evaluated))), coeff)
        checked += 1
    assert checked >= 3


@pytest.mark.parametrize("degree,bits,batch", [(2, [20], 3), (64, [30, 29], 4), (1024, [27, 28, 28], 3),
                                                (4096, [27, 28, 28], 7),   # n_4096_logq_27_28_28 (EncryptionParameters.swift:313-378)
                                                (8192, [30, 30, 29], 3), (16384, [30, 29], 2), (32768, [30], 1)])
def test_ntt_u32_matches_oracle(oracle, degree, bits, batch):
    moduli = oracle.generate_primes(bits, False, degree, word_bits=32)
    ours, ref = heamd.PolyContext(degree, moduli), oracle.PolyContext(degree, moduli)
    rng = np.random.default_rng(degree + batch)
    slab = _slab(rng, batch, moduli, degree)
    slab[0, :, 0] = 0
    slab[0, :, -1] = [m - 1 for m in moduli]
    assert np.array_equal(_to_host32(ours.forward_ntt_u32_(_to_device32(slab))), ref.forward_ntt(slab))
    assert np.array_equal(_to_host32(ours.inverse_ntt_u32_(_to_device32(slab))), ref.inverse_ntt(slab))
    # and the 4-byte path agrees with the 8-byte path of the same context
    assert np.array_equal(heamd.to_host(ours.forward_ntt_(heamd.to_device(slab))), ref.forward_ntt(slab))


def test_poly_ops_u32_match_oracle(oracle):
    degree = 4096
    moduli = oracle.generate_primes([27, 28, 28], False, degree, word_bits=32)
    ours, ref = heamd.PolyContext(degree, moduli), oracle.PolyContext(degree, moduli)
    rng = np.random.default_rng(9)
    x, y = _slab(rng, 3, moduli, degree), _slab(rng, 3, moduli, degree)
    for op, fn in (("add", ref.add), ("sub", ref.sub), ("mul", ref.mul)):
        got = _to_host32(ours.elementwise_u32_(op, _to_device32(x), _to_device32(y)))
        assert np.array_equal(got, fn(x, y)), op
    assert np.array_equal(_to_host32(ours.elementwise_u32_("neg", _to_device32(x))), ref.neg(x))
    scalars = [int(rng.integers(0, q)) for q in moduli]
    assert np.array_equal(_to_host32(ours.mul_scalar_u32_(_to_device32(x), scalars)), ref.mul_scalar(x, scalars))
    assert np.array_equal(_to_host32(ours.divide_and_round_q_last_u32(_to_device32(x))), ref.divide_and_round_q_last(x))


def test_u32_rejects_wide_moduli(oracle):
    moduli = oracle.generate_primes([31], False, 64)
    ctx = heamd.PolyContext(64, moduli)
    with pytest.raises(heamd.HeError) as err:
        ctx.forward_ntt_u32_(_to_device32(np.zeros((1, 1, 64))))
    assert err.value.name == "invalidModulus"
