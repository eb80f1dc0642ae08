"""The arithmetic claims behind the shift-folded butterfly products (csrc/device_math.hpp fold_mul as kModeFoldLazy uses it,
csrc/ntt_common.hpp), restated limb by limb on Python integers and held at the corners the proof names:

    p = 2^b - d, 41 <= b <= 55, d < 2^(b-32);  y any 64-bit word = b0 + b1 2^32;  constants w < p, wt = w 2^32 mod p;
    V = b0 w + b1 wt  (formed in two 64-bit columns, one carry),  F = 2^(b+2) = 4d (mod p),
    r = (V mod F) + (V >> (b+2)) 4d  =  w y (mod p),   0 <= r < 6p,   every intermediate inside its register.

The forward transform then never subtracts p: a word gains < 6p per stage (the kernels budget 8p), < (1 + 8 log2 N) p < 2^9 p."""
import random

MASK32, MASK64 = (1 << 32) - 1, (1 << 64) - 1


def fold_mul(y, w, wt, p):
    """The instruction sequence of fold_mul<UNIFORM, false>; asserts every register-width claim on the way."""
    bits = p.bit_length()
    d = (1 << bits) - p
    multiplier, shift = 4 * d, bits - 30
    assert multiplier <= MASK32 and 0 < shift < 32
    mask = (1 << shift) - 1
    b0, b1, w0, w1, t0, t1 = y & MASK32, y >> 32, w & MASK32, w >> 32, wt & MASK32, wt >> 32
    a = b0 * w0                          # v_mad_u64_u32 a = b0 w0 + 0: fits 64 bits
    assert a <= MASK64
    t_full = b1 * t0 + a                 # v_mad_u64_u32 t = b1 t0 + a, carry out
    t, carry = t_full & MASK64, t_full >> 64
    assert carry in (0, 1)
    u = b0 * w1 + ((t >> 32) | (carry << 32))  # the 2^32 column on top of (hi32(t), carry)
    u = b1 * t1 + u
    assert u <= MASK64, "the 2^32 column cannot carry"
    assert (u << 32) + (t & MASK32) == b0 * w + b1 * wt  # V exactly
    high = (u >> shift) & MASK32         # v_alignbit_b32(hi32(u), lo32(u), shift)
    assert u >> shift <= MASK32, "V >> (b+2) is a 32-bit word"
    kept = (u & MASK32) & mask
    r = high * multiplier + (((kept << 32) | (t & MASK32)))
    assert r <= MASK64
    return r


def corner_moduli():
    """(b, d) at the edges of eligibility (poly_context.cpp: 41 <= bits <= 55, d < 2^(bits-32)); p need not be prime for the
    congruence and the bounds to hold."""
    for b in (41, 42, 47, 50, 54, 55):
        top = 1 << (b - 32)
        for d in {1, 3, top // 4 + 1, top // 2 - 1, top // 2 + 1, top - 1}:
            if 0 < d < top:
                yield b, d


def test_folded_product_is_congruent_and_below_6p():
    rnd = random.Random(5)
    for b, d in corner_moduli():
        p = (1 << b) - d
        words = [0, 1, p - 1, p, MASK64, MASK64 - 1, 1 << 63, (1 << 32) - 1, 1 << 32, (MASK32 << 32)]
        words += [rnd.getrandbits(64) for _ in range(200)]
        constants = [1, 2, p - 1, p - 2, (p - 1) // 2, MASK32, 1 << 32] + [rnd.randrange(1, p) for _ in range(40)]
        worst = 0
        for w in constants:
            w %= p
            wt = (w << 32) % p
            for y in words:
                r = fold_mul(y, w, wt, p)
                assert r % p == (w * y) % p, (b, d, w, y)
                worst = max(worst, r)
        assert worst < 6 * p, (b, d, worst / p)
        # the bound itself: F - 1 + (2^31 - 1) 4d < 6p  <=>  d (2^33 + 2) < 2^(b+1), true for every d < 2^(b-32)
        assert (1 << (b + 2)) - 1 + ((1 << 31) - 1) * 4 * d < 6 * p


def test_forward_schedule_never_leaves_its_ceiling():
    """(x, y) -> (x + r, x + 8p - r) with r < 6p: after s stages every word is below (1 + 8 s) p; 15 stages of a 55-bit modulus
    stay below 2^9 p < 2^64, what the final fp32-estimated quotient (ntt_common.hpp LazyReducer: x < 2^10 p) is given."""
    for b, d in corner_moduli():
        p = (1 << b) - d
        bound = p  # canonical input (the fused decomposition hands over words below 2p: one more p)
        for stage in range(15):
            assert bound + 8 * p - 0 <= MASK64  # x + 8p - r with r >= 0
            bound = bound + 8 * p  # max(x + r, x + 8p - r)
        assert bound + p < (1 << 9) * p <= MASK64
