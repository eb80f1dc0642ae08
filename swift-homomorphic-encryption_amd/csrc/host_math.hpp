// host_math.hpp -- host-side (setup-time) modular arithmetic of the engine.
//
// Reproduces the *values* the reference computes once per context: primes, minimal primitive roots, inverses,
// Barrett / Shoup constants.  These determine the twiddle tables and the Bsk base, hence every output word.
// Citations are file:line relative to /root/reference/Sources/.
#pragma once

#include <cstdint>
#include <vector>

namespace heamd {

using u64 = uint64_t;
using u128 = unsigned __int128;

constexpr u64 kMaxModulus = (u64(1) << 62) - 1;   // ModularArithmetic/Modulus.swift:177-180
constexpr u64 kMTilde = u64(1) << 32;             // ModularArithmetic/Scalar.swift:523-525
constexpr u64 kGamma = (u64(1) << 62) - 40797;    // ModularArithmetic/Scalar.swift:517-519

inline bool is_power_of_two(u64 x) { return x != 0 && (x & (x - 1)) == 0; }
inline int floor_log2(u64 x) { return 63 - __builtin_clzll(x); }
inline int bit_length(u64 x) { return x == 0 ? 0 : 64 - __builtin_clzll(x); }

inline u64 mul_mod(u64 a, u64 b, u64 p) { return static_cast<u64>((static_cast<u128>(a) * b) % p); }
inline u64 add_mod(u64 a, u64 b, u64 p) {
    u64 s = a + b;
    return s >= p ? s - p : s;
}
inline u64 neg_mod(u64 a, u64 p) { return a == 0 ? 0 : p - a; }

inline u64 pow_mod(u64 base, u64 exponent, u64 p) {
    if (p == 1) return 0;
    u64 result = 1;
    base %= p;
    for (; exponent; exponent >>= 1) {
        if (exponent & 1) result = mul_mod(result, base, p);
        base = mul_mod(base, base, p);
    }
    return result;
}

// HomomorphicEncryption/Scalar.swift:162-202 -- deterministic Miller-Rabin with the first twelve primes.
inline bool is_prime(u64 n) {
    constexpr u64 kBases[] = {2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37};
    if (n < 2) return false;
    for (u64 b : kBases) {
        if (n == b) return true;
        if (n % b == 0) return false;
    }
    u64 d = n - 1;
    int r = 0;
    while ((d & 1) == 0) {
        d >>= 1;
        ++r;
    }
    for (u64 b : kBases) {
        u64 x = pow_mod(b, d, n);
        if (x == 1 || x == n - 1) continue;
        bool composite = true;
        for (int i = 0; i < r && composite; ++i) {  // r squarings, as Scalar.swift:192-197 does
            x = mul_mod(x, x, n);
            composite = (x != n - 1);
        }
        if (composite) return false;
    }
    return true;
}

// HomomorphicEncryption/PolyRq/PolyRq+Ntt.swift:24-27
inline bool is_ntt_modulus(u64 p, u64 degree) { return is_power_of_two(degree) && p != 1 && p % (2 * degree) == 1; }

// HomomorphicEncryption/Scalar.swift:113-154.  Returns false when not enough primes exist (notEnoughPrimes).
inline bool generate_primes(const std::vector<int>& bit_counts, bool preferring_small, u64 ntt_degree,
                            std::vector<u64>& out) {
    out.clear();
    const u128 step = static_cast<u128>(2) * ntt_degree;
    for (int bits : bit_counts) {
        if (bits < 1 || bits > 64) return false;
        const u128 lower = static_cast<u128>(1) << (bits - 1);
        const u128 upper = bits == 64 ? ((static_cast<u128>(1) << 64) - 1) : (static_cast<u128>(1) << bits);
        if (!preferring_small && upper < step) return false;
        u128 candidate = preferring_small ? lower + 1 : upper - step + 1;
        bool found = false;
        while (candidate >= lower && candidate < upper) {
            const u64 c = static_cast<u64>(candidate);
            bool taken = false;
            for (u64 q : out) taken = taken || q == c;
            if (!taken && is_prime(c) && is_ntt_modulus(c, ntt_degree)) {
                out.push_back(c);
                found = true;
                break;
            }
            if (preferring_small) {
                candidate += step;
            } else if (candidate >= step) {
                candidate -= step;
            } else {
                break;
            }
        }
        if (!found) return false;
    }
    return true;
}

// HomomorphicEncryption/Scalar.swift:76-96: value^-1 mod modulus (modulus need not be prime); false = notInvertible.
inline bool inverse_mod(u64 value, u64 modulus, u64& out) {
    if (value == 0 || modulus == 0) return false;
    __int128 r0 = modulus, r1 = value % modulus, t0 = 0, t1 = 1;
    while (r1 != 0) {
        __int128 q = r0 / r1;
        __int128 tmp = r0 - q * r1;
        r0 = r1;
        r1 = tmp;
        tmp = t0 - q * t1;
        t0 = t1;
        t1 = tmp;
    }
    if (r0 != 1) return false;
    if (t0 < 0) t0 += modulus;
    out = static_cast<u64>(t0);
    return true;
}

// ModularArithmetic/Scalar.swift:238-254
inline uint32_t reverse_bits(uint32_t x, int bit_count) {
    uint32_t r = 0;
    for (int i = 0; i < bit_count; ++i) r |= ((x >> i) & 1u) << (bit_count - 1 - i);
    return r;
}

// HomomorphicEncryption/PolyRq/PolyRq+Ntt.swift:87-105: the smallest primitive degree'th root of unity mod p
// (degree a power of two dividing p-1).  0 if none.
inline u64 min_primitive_root_of_unity(u64 p, u64 degree) {
    if (!is_power_of_two(degree) || degree < 2 || (p - 1) % degree != 0) return 0;
    u64 generator = 0;
    for (u64 c = 2; c < p; ++c) {
        u64 candidate = pow_mod(c, (p - 1) / degree, p);
        if (pow_mod(candidate, degree / 2, p) == p - 1) {  // order exactly `degree`
            generator = candidate;
            break;
        }
    }
    if (!generator) return 0;
    // every primitive root is an odd power of `generator`
    const u64 square = mul_mod(generator, generator, p);
    u64 smallest = generator, current = generator;
    for (u64 i = 1; i < degree / 2; ++i) {
        current = mul_mod(current, square, p);
        if (current < smallest) smallest = current;
    }
    return smallest;
}

// Shoup constant floor(c * 2^64 / p) (HomomorphicEncryption/Modulus.swift:92-103)
inline u64 shoup_factor(u64 multiplicand, u64 p) { return static_cast<u64>((static_cast<u128>(multiplicand) << 64) / p); }

// Limb-wise Shoup constants of c mod p (device_math.hpp split_mul_add): c 2^32 mod p, and the two 31-bit quotient
// factors floor(c 2^32 / 2p), floor((c 2^32 mod p) 2^32 / 2p) packed low | high
inline u64 split_shifted(u64 c, u64 p) { return static_cast<u64>((static_cast<u128>(c) << 32) % p); }
inline u64 split_factors(u64 c, u64 p) {
    const u64 shifted = split_shifted(c, p);
    const u64 f = static_cast<u64>((static_cast<u128>(c) << 32) / (static_cast<u128>(p) * 2));
    const u64 ft = static_cast<u64>((static_cast<u128>(shifted) << 32) / (static_cast<u128>(p) * 2));
    return f | (ft << 32);
}

// Q mod m for Q = prod(moduli) (HomomorphicEncryption/PolyRq/PolyContext.swift:184-191)
inline u64 product_mod(const u64* moduli, size_t count, u64 m) {
    u64 prod = 1 % m;
    for (size_t i = 0; i < count; ++i) prod = mul_mod(prod, moduli[i] % m, m);
    return prod;
}

}  // namespace heamd
