"""The limb-wise Shoup products of csrc/device_math.hpp on their own (tests/device_probe/arith_probe.hip runs them one lane
per operand pair): the congruence and the range each one states, against Python integers -- for ANY 64-bit operand, which is
what the butterflies rely on (the forward transform never folds its words, the inverse one multiplies x - y as a signed
word without adding a bound first)."""
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
    return lib


def _run(lib, kind, p, operands, constants):
    a = np.array(operands, dtype=np.uint64)
    c = np.array(constants, dtype=np.uint64)
    out = np.zeros(len(operands), dtype=np.uint64)
    status = lib.arith_probe_split_product(kind, p, a.ctypes.data, c.ctypes.data, len(operands), out.ctypes.data)
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
