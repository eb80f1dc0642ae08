"""The limb-wise Shoup products and the shift-folded products of csrc/device_math.hpp on their own
(tests/device_probe/arith_probe.hip runs them one lane per operand pair): the congruence and the range each one states, against
Python integers -- for ANY 64-bit operand, which is what the butterflies rely on (the forward transform never folds its words,
the limb-wise inverse multiplies x - y as a signed word without adding a bound first)."""
import ctypes
import os
import random

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

PROBE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "device_probe", "libarith_probe.so")


@pytest.fixture(scope="module")
def probe():
    assert os.path.exists(PROBE), "tests/device_probe/libarith_probe.so is built by __graft_entry__.build()"
    lib = ctypes.CDLL(PROBE)
    lib.arith_probe_split_product.restype = ctypes.c_int
    lib.arith_probe_split_product.argtypes = [ctypes.c_int, ctypes.c_uint64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t,
                                              ctypes.c_void_p]
    lib.arith_probe_fold_product.restype = ctypes.c_int
    lib.arith_probe_fold_product.argtypes = lib.arith_probe_split_product.argtypes
    return lib


def _run(lib, kind, p, operands, constants):
    a = np.array(operands, dtype=np.uint64)
    c = np.array(constants, dtype=np.uint64)
    out = np.zeros(len(operands), dtype=np.uint64)
    entry = lib.arith_probe_split_product if kind < 4 else lib.arith_probe_fold_product
    status = entry(kind, p, a.ctypes.data, c.ctypes.data, len(operands), out.ctypes.data)
    assert status == 0, status
    return [int(v) for v in out]


def _operands(rng, count):
    edges = [0, 1, 2, (1 << 31) - 1, 1 << 31, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, (1 << 63) - 1, 1 << 63, (1 << 63) + 1,
             (1 << 64) - 1, (1 << 64) - 2, (1 << 64) - (1 << 32), (1 << 64) - (1 << 32) - 1, 0xFFFFFFFF00000000, 0x7FFFFFFFFFFFFFFF,
             0x80000000FFFFFFFF, 0x7FFFFFFF00000000, 0x00000000FFFFFFFF]
    return edges + [rng.getrandbits(64) for _ in range(count - len(edges))]


def _constants(rng, p, count):
    edges = [0, 1, 2, p - 1, p - 2, p >> 1, (p >> 1) + 1, (1 << 31) % p, (1 << 32) % p, ((1 << 32) - 1) % p]
    return edges + [rng.randrange(p) for _ in range(count - len(edges))]


MODULI = [(1 << 40) + 15, 1099511922689, 281474976694273, 18014398509309953, 36028797018652673, 36028797018914815,
          (1 << 55) - 55]


@pytest.mark.parametrize("p", MODULI)
def test_split_products_any_word(probe, p):
    """split_mul_add (unsigned word y): y w - Q 2p in [0, 8p), congruent to y w.  split_mul_signed (signed word d, the
    constant's second word in signed limbs): d w + bias - ... in (0, 6p), congruent to d w -- both for every 64-bit pattern,
    limb edges and sign edges included, constants at the ends of [0, p); moduli across [2^40, 2^55) (the products use nothing
    of a modulus but its size and oddness, so the list is not restricted to primes or NTT moduli)."""
    assert (1 << 40) <= p < (1 << 55) and p % 2 == 1
    rng = random.Random(p)
    count = 1 << 14
    operands = _operands(rng, count)
    rng.shuffle(operands)
    constants = _constants(rng, p, count)
    # every edge word against every edge constant, then the shuffled lists pair by pair
    edge_words, edge_constants = _operands(rng, 20), _constants(rng, p, 10)
    operands = [y for y in edge_words for _ in edge_constants] + operands
    constants = [w for _ in edge_words for w in edge_constants] + constants
    for kind in (0, 1):
        got = _run(probe, kind, p, operands, constants)
        for y, w, r in zip(operands, constants, got):
            value = y if kind == 0 else (y - (1 << 64) if y >> 63 else y)
            assert (r - value * w) % p == 0, (kind, hex(y), w, r)
            assert (0 <= r < 8 * p) if kind == 0 else (0 < r < 6 * p), (kind, hex(y), w, r // p)
    # wave-uniform constants (the constant's words in scalar registers): one constant per launch
    for w in constants[:12] + constants[-4:]:
        for kind in (2, 3):
            got = _run(probe, kind, p, operands[:1024], [w] * 1024)
            for y, r in zip(operands[:1024], got):
                value = y if kind == 2 else (y - (1 << 64) if y >> 63 else y)
                assert (r - value * w) % p == 0, (kind, hex(y), w, r)
                assert (0 <= r < 8 * p) if kind == 2 else (0 < r < 6 * p), (kind, hex(y), w, r // p)


# p = 2^b - d at both ends of d < 2^(b-32) for b = 41, 47, 52, 55 (kModeFoldLazy: the bound since round 6, 2^(b-33) before --
# both kept), 56 and 60 (kModeFoldMinus: d < 2^(b-33))
FOLD_MINUS = [(1 << 41) - 1, (1 << 41) - 255, (1 << 41) - 511, (1 << 47) - 8191, (1 << 47) - 16383, (1 << 47) - 32767,
              (1 << 52) - 245759, (1 << 52) - 524287, (1 << 52) - 1048575, (1 << 55) - 55, (1 << 55) - 4087807,
              (1 << 55) - 4194303, (1 << 55) - 8388607, (1 << 56) - 27, (1 << 56) - 8388607, (1 << 60) - 93,
              (1 << 60) - 134217727]
# p = 2^60 + e, e < 2^24 (the BEHZ auxiliary primes' form)
FOLD_PLUS = [(1 << 60) + 33, (1 << 60) + 1, (1 << 60) + (1 << 24) - 1, (1 << 60) + 8380417]


@pytest.mark.parametrize("p", FOLD_MINUS + FOLD_PLUS)
def test_folded_products_any_word(probe, p):
    """fold_mul (csrc/device_math.hpp): the product by a constant folded by a shift at 2^(b+2) (p = 2^b - d) or 2^62 (p = 2^60 + e):
    congruent to y w and below 6p for EVERY 64-bit y -- what lets kModeFoldLazy's butterflies run without a conditional
    subtract (tests/test_fold_product_bounds.py holds the same claims on Python integers); constants in vector registers and
    wave-uniform."""
    plus = p > (1 << 60)
    rng = random.Random(p)
    count = 1 << 13
    edge_words, edge_constants = _operands(rng, 20), _constants(rng, p, 10)
    operands = [y for y in edge_words for _ in edge_constants] + _operands(rng, count)
    constants = [w for _ in edge_words for w in edge_constants] + _constants(rng, p, count)
    got = _run(probe, 6 if plus else 4, p, operands, constants)
    for y, w, r in zip(operands, constants, got):
        assert (r - y * w) % p == 0, (hex(y), w, r)
        assert 0 <= r < 6 * p, (hex(y), w, r / p)
    for w in constants[:12] + constants[-4:]:
        got = _run(probe, 7 if plus else 5, p, operands[:1024], [w] * 1024)
        for y, r in zip(operands[:1024], got):
            assert (r - y * w) % p == 0, (hex(y), w, r)
            assert 0 <= r < 6 * p, (hex(y), w, r / p)
