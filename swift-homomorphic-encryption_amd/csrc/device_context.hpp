// device_context.hpp -- the device-resident, immutable image of a PolyContext that kernels receive by value.
#pragma once

#include <stdint.h>

#include "device_math.hpp"

namespace heamd {

// Per-modulus constants (one per RNS row).  Loaded through the scalar cache: every lane of a workgroup reads the
// same entry.  Sources: HomomorphicEncryption/Modulus.swift:24-45 (Barrett), PolyRq/PolyRq+Ntt.swift:159-168 (N^-1).
// DeviceModulus::has_ntt: with kNttPlainInverseDegree inv_degree is N^-1 itself (the inverse transform divides by N
// exactly, ntt_common.hpp divide_by_degree); with kNttScaledInverseDegree it carries another factor (t N^-1 in BEHZ's
// dropExtendedBase) and the last stage takes the Shoup product.
constexpr uint32_t kNttPlainInverseDegree = 1, kNttScaledInverseDegree = 2;
struct DeviceModulus {
    uint64_t p;
    uint64_t barrett64;        // floor(2^64 / p)
    uint64_t barrett128_lo;    // floor(2^128 / p), low word
    uint64_t barrett128_hi;    //                   high word
    uint64_t product_factor;   // floor(2^(bits(p)+62) / p)
    uint32_t product_shift;    // bits(p) - 2
    uint32_t has_ntt;          // 0: not an NTT modulus for this degree; kNttPlainInverseDegree; kNttScaledInverseDegree
    uint64_t inv_degree;       // N^-1 mod p                  (+ Shoup factor)
    uint64_t inv_degree_shoup;
    uint64_t inv_degree_root;  // N^-1 * psi^(-N/2) mod p     (+ Shoup factor)
    uint64_t inv_degree_root_shoup;
    uint64_t two64_mod_p;        // 2^64 mod p (+ Shoup factor): folds the high word of a 128-bit sum
    uint64_t two64_mod_p_shoup;
    // limb-wise Shoup constants of the two N^-1 factors (device_math.hpp split_mul_add): c 2^32 mod p and
    // floor(c 2^32 / 2p) | floor((c 2^32 mod p) 2^32 / 2p) << 32
    uint64_t inv_degree_split;
    uint64_t inv_degree_factors;
    uint64_t inv_degree_root_split;
    uint64_t inv_degree_root_factors;
    // one-word-quotient Barrett for exact sums below 2^(64 + wide_shift) (device_math.hpp reduce_product_sum_bounded):
    // wide_shift = bits(p) - 1, wide_factor = floor(2^(64 + wide_shift) / p); wide_shift = 0 where it does not apply
    // (p below 2^33, above 2^61, or a power of two)
    uint64_t wide_factor;
    uint32_t wide_shift;
    // != 0 (= b - 31) for a modulus just below a power of two, p = 2^b - delta, 41 <= b <= 55, delta < 2^(b - 32) -- what
    // generatePrimes(preferringSmall: false) returns, i.e. every parameter set of the reference: the flag that selects the
    // shift-folded products of ntt_common.hpp kModeFoldLazy (2^(b+2) = 4 delta mod p; products below 6p for any 64-bit word,
    // poly_context.cpp).  The value itself (rounds 3-4: the shift that read quotient factors off a constant) is not read on
    // the device.
    uint32_t split_shift;
};

struct DeviceContext {
    const DeviceModulus* moduli;   // [L]
    const U64x2* forward_twiddles; // [L][N]  (w, floor(w 2^64 / p)), bit-reversed order   PolyRq+Ntt.swift:125-143
    const U64x2* inverse_twiddles; // [L][N]  stage-major re-ordered                       PolyRq+Ntt.swift:146-157
    const U64x2* inverse_q_last;   // [L][L]  row k: (q_{k}^-1 mod q_i, Shoup) for i < k   PolyContext.swift:108-111
    // the same twiddles as limb-wise Shoup constants (device_math.hpp split_mul_add), same order:
    // pairs (w, w 2^32 mod p), factors floor(w 2^32 / 2p) | floor((w 2^32 mod p) 2^32 / 2p) << 32
    const U64x2* forward_split_pairs;      // [L][N]
    const U64x2* inverse_split_pairs;      // [L][N]
    const U64x2* inverse_split_pairs_signed;  // [L][N] the same with w 2^32 mod p in signed limbs (kModeSplitSigned)
    const uint64_t* forward_split_factors; // [L][N]
    const uint64_t* inverse_split_factors; // [L][N]
    // The same tables with every stage's block lane-major (ntt_common.hpp Twiddles::lanes) for the pass structures of the
    // tiled kernels with 8 words per lane at N = 4096 and N = 8192; nullptr at other degrees.  `_lanes`: the partition with
    // the partial pass on the low bits (N = 8192: 12-10 | 9-7 | 6-4 | 3-1 | 0; N = 4096: four full passes) -- every forward
    // kernel and the fused inverse ones; `_lanes_top`: 2-0 | 5-3 | 8-6 | 11-9 | 12, the plain-slab inverse at N = 8192.
    const U64x2* forward_split_pairs_lanes;
    const U64x2* inverse_split_pairs_lanes;
    const U64x2* inverse_split_pairs_signed_lanes;
    const U64x2* inverse_split_pairs_lanes_top;
    const uint64_t* forward_split_factors_lanes;
    const uint64_t* inverse_split_factors_lanes;
    const uint64_t* inverse_split_factors_lanes_top;
    uint32_t degree;
    uint32_t log_degree;
    uint32_t moduli_count;         // active moduli (a prefix of the context's list)
    uint32_t moduli_stride;        // moduli of the full context = row stride of inverse_q_last
    uint32_t approx_ok;            // 1 when every modulus is < 2^61 (lazy range [0, 8p) fits 64 bits)
    uint32_t headroom_ok;          // 1 when every modulus is in [2^40, 2^55): the NTT runs the fold-free split butterflies
    uint32_t headroom_prefix;      // how many leading moduli are in that range (the Q part of a [Q, Bsk] context)
    uint32_t shift_prefix;         // how many leading moduli also have DeviceModulus::split_shift != 0
    // bit i: modulus i takes the fold butterflies (ntt_common.hpp kModeFoldMinus / kModeFoldPlus): 2^b - d with
    // 56 <= b <= 60 and d < 2^(b-33), or 2^60 + e with e < 2^24
    uint64_t fold_minus_mask, fold_plus_mask;
    uint32_t scaled_inverse_degree;  // 1 when `moduli` is a table whose N^-1 constants carry another factor
                                     // (kNttScaledInverseDegree): the inverse transform must not divide by N exactly
};

// Image of a context whose moduli all fit UInt32 (<= 2^30 - 1), for slabs of 4-byte words: the same per-modulus
// constants, twiddles as (w, floor(w 2^32 / p)) pairs (the high halves of the 64-bit Shoup factors).
struct alignas(8) U32x2 {
    uint32_t x, y;
};
struct DeviceContext32 {
    const DeviceModulus* moduli;
    const U32x2* forward_twiddles;  // [L][N]
    const U32x2* inverse_twiddles;  // [L][N]
    const U64x2* inverse_q_last;    // [L][L]
    uint32_t degree, log_degree, moduli_count, moduli_stride;
};

}  // namespace heamd
