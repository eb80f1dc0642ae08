// device_math.hpp -- 64-bit modular arithmetic on CDNA4 vector ALUs.
//
// gfx950 has no 64-bit integer multiplier: every 64x64 product is built from v_mad_u64_u32 / v_mul_hi_u32
// (32x32 -> 64).  The routines below are written so that hipcc folds the partial-product additions into
// v_mad_u64_u32's 64-bit addend and so that the *count* of 32-bit multiplies is minimal -- the NTT is bound by the
// integer-multiply issue rate, not by HBM (DESIGN.md section "Rooflines").
//
// Exactness: every public entry point of the library returns canonical residues, so kernels are free to use lazy
// ranges internally as long as the last step lands in [0, p).  Reference semantics being reproduced:
//   MultiplyConstantModulus.multiplyModLazy / multiplyMod   ModularArithmetic/Modulus.swift:401-415
//   ReduceModulus.reduce(T) / reduce(T.DoubleWidth) / reduceProduct   ModularArithmetic/Modulus.swift:258-360
//   subtractIfExceeds / addMod / subtractMod / negateMod    ModularArithmetic/Scalar.swift:146-193
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>


namespace heamd {

struct alignas(16) U64x2 {
    uint64_t x, y;
};

// 16-byte loads / stores of words that are touched once per launch (polynomial slabs streamed through a kernel):
// non-temporal, so that they pass the caches without displacing what other workgroups re-read (constant tables,
// shared operand tiles).
typedef unsigned long long StreamWords __attribute__((ext_vector_type(2)));
__device__ __forceinline__ U64x2 stream_load(const U64x2* p) {
    const StreamWords v = __builtin_nontemporal_load(reinterpret_cast<const StreamWords*>(p));
    return U64x2{v.x, v.y};
}
__device__ __forceinline__ void stream_store(U64x2* p, U64x2 value) {
    const StreamWords v = {value.x, value.y};
    __builtin_nontemporal_store(v, reinterpret_cast<StreamWords*>(p));
}

// the 8-byte forms (one word per lane: the per-coefficient BEHZ kernels)
__device__ __forceinline__ uint64_t stream_load(const uint64_t* p) {
    return __builtin_nontemporal_load(p);
}
__device__ __forceinline__ void stream_store(uint64_t* p, uint64_t value) {
    __builtin_nontemporal_store(value, p);
}
// 4-byte slabs (PolyRq<UInt32> / Bfv<UInt32>): the word is widened in the register, never in memory
__device__ __forceinline__ uint64_t stream_load(const uint32_t* p) {
    return __builtin_nontemporal_load(p);
}
__device__ __forceinline__ void stream_store(uint32_t* p, uint64_t value) {
    __builtin_nontemporal_store(static_cast<uint32_t>(value), p);
}

__device__ __forceinline__ uint32_t lo32(uint64_t v) { return static_cast<uint32_t>(v); }
__device__ __forceinline__ uint32_t hi32(uint64_t v) { return static_cast<uint32_t>(v >> 32); }

// 32x32 + 64 -> 64 (selected as one v_mad_u64_u32)
__device__ __forceinline__ uint64_t mad32(uint32_t a, uint32_t b, uint64_t c) {
    return static_cast<uint64_t>(a) * b + c;
}
__device__ __forceinline__ uint64_t mul32(uint32_t a, uint32_t b) { return static_cast<uint64_t>(a) * b; }
// High half of a 32x32 product.  v_mul_hi_u32 issues at about half the rate of v_mad_u64_u32 on gfx950
// (profiles/r01_microbench_instruction_rates.txt: 20.4 vs 36.3 T lane-ops/s), so take the high register of a full
// multiply-add instead; asm because hipcc would narrow `(uint64_t(a) * b) >> 32` back to v_mul_hi_u32.
__device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) {
    uint64_t d, carry;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(carry) : "v"(a), "v"(b));
    return static_cast<uint32_t>(d >> 32);
}
// asm on purpose: with a plain `a * b` hipcc recognises the schoolbook pattern below, rebuilds a 64/128-bit multiply
// and re-expands it with redundant (even multiply-by-zero) v_mad_u64_u32.
__device__ __forceinline__ uint32_t mullo32(uint32_t a, uint32_t b) {
    uint32_t d;
    asm("v_mul_lo_u32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// Optimisation barrier (emits nothing): hides how a value was derived, so that hipcc does not re-associate our
// partial products with the additions that produced the operand and re-expand them as generic 64x64 multiplies.
__device__ __forceinline__ uint64_t opaque(uint64_t v) {
    asm("" : "+v"(v));
    return v;
}
// 32-bit flavour: keeps a word-sized add a word-sized add (hipcc otherwise widens `acc + (h << 32)` to a 64-bit add of
// a freshly built {0, h} pair: two moves and a v_lshl_add_u64 for one v_add_u32)
__device__ __forceinline__ uint32_t opaque32(uint32_t v) {
    asm("" : "+v"(v));
    return v;
}
__device__ __forceinline__ uint64_t pack64(uint32_t lo, uint32_t hi) { return (static_cast<uint64_t>(hi) << 32) | lo; }

// Exact high word of a 64x64 product: 4 multiplies.
__device__ __forceinline__ uint64_t mulhi64(uint64_t a, uint64_t b) {
    const uint32_t a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
    const uint32_t p00_hi = mulhi32(a0, b0);
    const uint64_t p01 = mad32(a0, b1, p00_hi);
    const uint64_t p10 = mad32(a1, b0, lo32(p01));
    return mad32(a1, b1, static_cast<uint64_t>(hi32(p01)) + hi32(p10));
}

// High word of a 64x64 product, allowed to be low by at most 2: 3 multiplies.
// Keeps a1*b1 + hi32(a0*b1) + hi32(a1*b0); drops the low-column carries (< 3).
__device__ __forceinline__ uint64_t mulhi64_approx(uint64_t a, uint64_t b) {
    const uint32_t a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
    const uint32_t p01_hi = mulhi32(a0, b1);
    const uint32_t p10_hi = mulhi32(a1, b0);
    return mad32(a1, b1, static_cast<uint64_t>(p01_hi) + p10_hi);
}

// Low word of a*b + c*d (mod 2^64): 6 multiplies.
__device__ __forceinline__ uint64_t mullo64_sum2(uint64_t a, uint64_t b, uint64_t c, uint64_t d) {
    const uint32_t a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
    const uint32_t c0 = lo32(c), c1 = hi32(c), d0 = lo32(d), d1 = hi32(d);
    const uint64_t acc = mad32(c0, d0, mul32(a0, b0));
    const uint32_t high = hi32(acc) + mullo32(a0, b1) + mullo32(a1, b0) + mullo32(c0, d1) + mullo32(c1, d0);
    return pack64(lo32(acc), high);
}

// x >= m ? x - m : x   for x < 2m, m <= 2^63 (every modulus multiple used here: p <= 2^62 - 1, 2p, and 4p for p < 2^61).
// The wrapped difference d = x + (2^64 - m) lies in [0, m) when x >= m and in [2^64 - m, 2^64) otherwise, and m <= 2^63
// puts exactly the second range above 2^63: the top bit of d selects.  One 64-bit add, one 32-bit sign test and the
// select -- no borrow chain, no VCC hazard, no copy of m's high word into a VGPR (4 instructions against 7).
// (asm for the add: hipcc otherwise turns it back into the subtract-with-borrow pair and a 64-bit signed compare.)
// UNIFORM: neg_m = 2^64 - m is wave-uniform and is read from an SGPR pair.
template <bool UNIFORM = false>
__device__ __forceinline__ uint64_t csub63(uint64_t x, uint64_t neg_m) {
    uint64_t d;
    if constexpr (UNIFORM) {
        asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(d) : "v"(x), "s"(neg_m));
    } else {
        asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(d) : "v"(x), "v"(neg_m));
    }
    return static_cast<int32_t>(hi32(d)) < 0 ? x : d;
}
__device__ __forceinline__ uint64_t csub(uint64_t x, uint64_t m) { return csub63<false>(x, 0 - m); }
__device__ __forceinline__ uint64_t csub_uniform(uint64_t x, uint64_t m) { return csub63<true>(x, 0 - m); }

// Shoup multiplication by the constant w (wf = floor(w * 2^64 / p)), `neg_p` = 2^64 - p.
//   lazy  : result in [0, 2p), exact quotient estimate (4 + 6 multiplies)
//   lazy4 : result in [0, 4p), quotient estimate low by <= 2 (3 + 6 multiplies); needs 4p < 2^64
__device__ __forceinline__ uint64_t shoup_lazy(uint64_t x, uint64_t w, uint64_t wf, uint64_t neg_p) {
    x = opaque(x);
    return mullo64_sum2(x, w, opaque(mulhi64(x, wf)), neg_p);
}

// ---- straight-line Shoup products (asm: hipcc keeps re-deriving a generic 64x64 multiply from the C form) -----------
// Low 64 bits of  addend + x w + q neg  (neg = 2^64 - p or 2^64 - 2p), as two multiply-add chains: the 2^0 column
// (a0 w0 + q0 n0, 64 bits, on top of the addend -- the forward butterfly's x + w y leaves the multiplier's addend
// port for free) and the 2^32 column (a0 w1 + a1 w0 + q0 n1 + q1 n0, of which only the low word matters).  Six
// multiply-adds and one add; no v_mul_hi_u32, no carry chains, no VCC.  UNIFORM: the twiddle words are wave-uniform
// and stay in SGPRs (first forward / last inverse pass); the reduction constant always does -- every instruction
// reads at most one SGPR.
template <bool UNIFORM, bool ADD>
__device__ __forceinline__ uint64_t shoup_low64(uint64_t addend, uint64_t x, uint64_t w, uint64_t q, uint64_t neg) {
    const uint32_t a0 = lo32(x), a1 = hi32(x), q0 = lo32(q), q1 = hi32(q), w0 = lo32(w), w1 = hi32(w);
    const uint32_t n0 = lo32(neg), n1 = hi32(neg);
    uint64_t acc, high, carry;
#define HEAMD_LOW64_TAIL                           \
    "v_mad_u64_u32 %0, %2, %7, %9, %0\n\t"          \
    "v_mad_u64_u32 %1, %2, %3, %6, 0\n\t"           \
    "v_mad_u64_u32 %1, %2, %4, %5, %1\n\t"          \
    "v_mad_u64_u32 %1, %2, %7, %10, %1\n\t"         \
    "v_mad_u64_u32 %1, %2, %8, %9, %1"
    if constexpr (UNIFORM && ADD) {
        asm("v_mad_u64_u32 %0, %2, %3, %5, %11\n\t" HEAMD_LOW64_TAIL
            : "=&v"(acc), "=&v"(high), "=&s"(carry)
            : "v"(a0), "v"(a1), "s"(w0), "s"(w1), "v"(q0), "v"(q1), "s"(n0), "s"(n1), "v"(addend));
    } else if constexpr (UNIFORM) {
        asm("v_mad_u64_u32 %0, %2, %3, %5, 0\n\t" HEAMD_LOW64_TAIL
            : "=&v"(acc), "=&v"(high), "=&s"(carry)
            : "v"(a0), "v"(a1), "s"(w0), "s"(w1), "v"(q0), "v"(q1), "s"(n0), "s"(n1));
    } else if constexpr (ADD) {
        asm("v_mad_u64_u32 %0, %2, %3, %5, %11\n\t" HEAMD_LOW64_TAIL
            : "=&v"(acc), "=&v"(high), "=&s"(carry)
            : "v"(a0), "v"(a1), "v"(w0), "v"(w1), "v"(q0), "v"(q1), "s"(n0), "s"(n1), "v"(addend));
    } else {
        asm("v_mad_u64_u32 %0, %2, %3, %5, 0\n\t" HEAMD_LOW64_TAIL
            : "=&v"(acc), "=&v"(high), "=&s"(carry)
            : "v"(a0), "v"(a1), "v"(w0), "v"(w1), "v"(q0), "v"(q1), "s"(n0), "s"(n1));
    }
#undef HEAMD_LOW64_TAIL
    return pack64(lo32(acc), opaque32(hi32(acc) + lo32(high)));
}

// Quotient estimate floor(x f / 2^64) from three multiply-adds (the a0 b0 partial product is dropped):
//   CARRY = false  needs x < 2^63 and f < 2^63 (headroom mode multiplies by f = wf >> 1): a0 b1 + a1 b0 <
//                  2^63 + 2^63 - 2^33 never leaves 64 bits; estimate low by <= 1;
//   CARRY = true   any x, f: the 65th bit of the cross column waits in an SGPR pair and re-enters as bit 32 of the
//                  estimate; low by <= 2.
template <bool UNIFORM, bool CARRY>
__device__ __forceinline__ uint64_t shoup_quotient(uint64_t x, uint64_t f) {
    const uint32_t a0 = lo32(x), a1 = hi32(x), b0 = lo32(f), b1 = hi32(f);
    uint64_t cross, q, carry;
    if constexpr (CARRY) {
        uint32_t carried;
        if constexpr (UNIFORM) {
            asm("v_mad_u64_u32 %0, %3, %4, %7, 0\n\t"
                "v_mad_u64_u32 %0, %3, %5, %6, %0\n\t"
                "v_cndmask_b32 %2, 0, 1, %3\n\t"
                "v_lshrrev_b64 %0, 32, %0\n\t"
                "v_mad_u64_u32 %1, %3, %5, %7, %0"
                : "=&v"(cross), "=&v"(q), "=&v"(carried), "=&s"(carry)
                : "v"(a0), "v"(a1), "s"(b0), "s"(b1));
        } else {
            asm("v_mad_u64_u32 %0, %3, %4, %7, 0\n\t"
                "v_mad_u64_u32 %0, %3, %5, %6, %0\n\t"
                "v_cndmask_b32 %2, 0, 1, %3\n\t"
                "v_lshrrev_b64 %0, 32, %0\n\t"
                "v_mad_u64_u32 %1, %3, %5, %7, %0"
                : "=&v"(cross), "=&v"(q), "=&v"(carried), "=&s"(carry)
                : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
        }
        return pack64(lo32(q), opaque32(hi32(q) + carried));
    } else {
        if constexpr (UNIFORM) {
            asm("v_mad_u64_u32 %0, %2, %3, %6, 0\n\t"
                "v_mad_u64_u32 %0, %2, %4, %5, %0\n\t"
                "v_lshrrev_b64 %0, 32, %0\n\t"
                "v_mad_u64_u32 %1, %2, %4, %6, %0"
                : "=&v"(cross), "=&v"(q), "=&s"(carry)
                : "v"(a0), "v"(a1), "s"(b0), "s"(b1));
        } else {
            asm("v_mad_u64_u32 %0, %2, %3, %6, 0\n\t"
                "v_mad_u64_u32 %0, %2, %4, %5, %0\n\t"
                "v_lshrrev_b64 %0, 32, %0\n\t"
                "v_mad_u64_u32 %1, %2, %4, %6, %0"
                : "=&v"(cross), "=&v"(q), "=&s"(carry)
                : "v"(a0), "v"(a1), "v"(b0), "v"(b1));
        }
        return q;
    }
}

// lazy4: any 64-bit x, p < 2^61; result x w - q p in [0, 4p) (q low by <= 2 -> < 3p)
__device__ __forceinline__ uint64_t shoup_lazy4(uint64_t x, uint64_t w, uint64_t wf, uint64_t neg_p) {
    return shoup_low64<false, false>(0, x, w, shoup_quotient<false, true>(x, wf), neg_p);
}
__device__ __forceinline__ uint64_t shoup_lazy4_uniform(uint64_t x, uint64_t w, uint64_t wf, uint64_t neg_p) {
    return shoup_low64<true, false>(0, x, w, shoup_quotient<true, true>(x, wf), neg_p);
}
// addend + (x w - q p) mod 2^64
template <bool UNIFORM>
__device__ __forceinline__ uint64_t shoup_lazy4_fma(uint64_t addend, uint64_t x, uint64_t w, uint64_t wf,
                                                    uint64_t neg_p) {
    return shoup_low64<UNIFORM, true>(addend, x, w, shoup_quotient<UNIFORM, true>(x, wf), neg_p);
}
// headroom: x < 2^63, wf_half = floor(w 2^63 / p) = wf >> 1, neg_2p = 2^64 - 2p; result x w - q 2p in [0, 5p)
// (q low by <= 1, and wf_half costs < 3/2 more)
__device__ __forceinline__ uint64_t shoup_headroom(uint64_t x, uint64_t w, uint64_t wf_half, uint64_t neg_2p) {
    return shoup_low64<false, false>(0, x, w, shoup_quotient<false, false>(x, wf_half), neg_2p);
}
__device__ __forceinline__ uint64_t shoup_headroom_uniform(uint64_t x, uint64_t w, uint64_t wf_half, uint64_t neg_2p) {
    return shoup_low64<true, false>(0, x, w, shoup_quotient<true, false>(x, wf_half), neg_2p);
}
// addend + (x w - q 2p) mod 2^64
template <bool UNIFORM>
__device__ __forceinline__ uint64_t shoup_headroom_fma(uint64_t addend, uint64_t x, uint64_t w, uint64_t wf_half,
                                                       uint64_t neg_2p) {
    return shoup_low64<UNIFORM, true>(addend, x, w, shoup_quotient<UNIFORM, false>(x, wf_half), neg_2p);
}

// ---- limb-wise Shoup product ("split" butterflies): 8 multiply-adds, no shift, no carry ------------------------------
// For y = b0 + b1 2^32 and a constant w with wt = w 2^32 mod p:  w y = b0 w + b1 wt (mod p), a sum below 2^33 p, so its
// quotient by 2p is a 32-bit word: Q = hi32(b0 f + b1 ft) with the 31-bit factors f = floor(w 2^32 / 2p),
// ft = floor(wt 2^32 / 2p) -- Shoup's estimate applied per limb; the two products < 2^63 cannot leave 64 bits and Q
// is at most 3 below the true quotient, hence
//     addend + b0 w + b1 wt - Q 2p   in   addend + [0, 8p)        for ANY 64-bit y.
// Its low 64 bits are two column chains of three multiply-adds (2^0 column on top of the addend, 2^32 column of
// which only the low word matters) and one 32-bit add.  Against the 64-bit Shoup product above: 8 multiplies instead
// of 9, no 64-bit shift, no bound on the multiplicand.  Constants per twiddle: (w, wt) and (f, ft) = 24 bytes.
// neg_2p = 2^64 - 2p.  UNIFORM: the constant's words are wave-uniform and stay in SGPRs (one SGPR per instruction).
template <bool UNIFORM, bool ADD>
__device__ __forceinline__ uint64_t split_mul_add(uint64_t addend, uint64_t y, uint64_t w, uint64_t wt, uint64_t f_pair,
                                                 uint64_t neg_2p) {
    const uint32_t b0 = lo32(y), b1 = hi32(y);
    const uint32_t w0 = lo32(w), w1 = hi32(w), t0 = lo32(wt), t1 = hi32(wt), f = lo32(f_pair), ft = hi32(f_pair);
    const uint32_t n0 = lo32(neg_2p), n1 = hi32(neg_2p);
    uint64_t q, c0, c1, carry;
#define HEAMD_SPLIT_BODY(FIRST)                       \
    "v_mad_u64_u32 %0, %3, %4, %10, 0\n\t"            \
    FIRST                                             \
    "v_mad_u64_u32 %2, %3, %4, %8, 0\n\t"             \
    "v_mad_u64_u32 %0, %3, %5, %11, %0\n\t"           \
    "v_mad_u64_u32 %1, %3, %5, %7, %1\n\t"            \
    "v_mad_u64_u32 %2, %3, %5, %9, %2"
    // operands: 0 q, 1 c0, 2 c1, 3 carry | 4 b0, 5 b1, 6 w0, 7 t0, 8 w1, 9 t1, 10 f, 11 ft, 12 addend
    if constexpr (UNIFORM && ADD) {
        asm(HEAMD_SPLIT_BODY("v_mad_u64_u32 %1, %3, %4, %6, %12\n\t")
            : "=&v"(q), "=&v"(c0), "=&v"(c1), "=&s"(carry)
            : "v"(b0), "v"(b1), "s"(w0), "s"(t0), "s"(w1), "s"(t1), "s"(f), "s"(ft), "v"(addend));
    } else if constexpr (UNIFORM) {
        asm(HEAMD_SPLIT_BODY("v_mad_u64_u32 %1, %3, %4, %6, 0\n\t")
            : "=&v"(q), "=&v"(c0), "=&v"(c1), "=&s"(carry)
            : "v"(b0), "v"(b1), "s"(w0), "s"(t0), "s"(w1), "s"(t1), "s"(f), "s"(ft));
    } else if constexpr (ADD) {
        asm(HEAMD_SPLIT_BODY("v_mad_u64_u32 %1, %3, %4, %6, %12\n\t")
            : "=&v"(q), "=&v"(c0), "=&v"(c1), "=&s"(carry)
            : "v"(b0), "v"(b1), "v"(w0), "v"(t0), "v"(w1), "v"(t1), "v"(f), "v"(ft), "v"(addend));
    } else {
        asm(HEAMD_SPLIT_BODY("v_mad_u64_u32 %1, %3, %4, %6, 0\n\t")
            : "=&v"(q), "=&v"(c0), "=&v"(c1), "=&s"(carry)
            : "v"(b0), "v"(b1), "v"(w0), "v"(t0), "v"(w1), "v"(t1), "v"(f), "v"(ft));
    }
#undef HEAMD_SPLIT_BODY
    uint64_t carry2;
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\t"
        "v_mad_u64_u32 %1, %2, %3, %5, %1"
        : "+v"(c0), "+v"(c1), "=&s"(carry2)
        : "v"(hi32(q)), "s"(n0), "s"(n1));
    return pack64(lo32(c0), opaque32(hi32(c0) + lo32(c1)));
}

// The same product for a SIGNED multiplicand d = b0 + b1 2^32 (b1 signed, any 64-bit d): the inverse butterfly's x - y
// as it leaves the subtraction, without the bound that would keep it non-negative (one 64-bit addition per butterfly).
// The constant's second word comes in signed-limb form, wt = t0s + t1' 2^32 with t0s = the low word read as signed and
// t1' = hi32(wt) + (t0 >> 31) (DeviceContext::inverse_split_pairs_signed holds it that way, poly_context.cpp), so that
// V = b0 w + b1 wt = d w (mod p) is formed by signed multiply-adds on b1 (v_mad_i64_i32) -- V in (-2^31 p, 3 2^31 p).
// Its quotient by 2p, Q = floor((b0 f + b1 ft) / 2^32) in [-2^30, 2^31 + 2^30), fits no 32-bit word of either signedness;
// Q' = Q + 2^30 does (the estimate's chain starts at 2^62 -- the inline constant 2.0 read as a 64-bit operand), and
//     V - Q 2p = V + Q' (2^64 - 2p) + 2^31 p   (mod 2^64),
// so the chains are the unsigned ones with the 2^0 column started at  bias = beta + 2^31 p (mod 2^64).  The estimate:
// b0 w / 2p - b0 f / 2^32 in [0, 1), b1 (wt / 2p - ft / 2^32) in (-1/2, 1/2), Q at most one above E - 1: V - Q 2p lies in
// (-p, 5p); beta = p brings it to (0, 6p), inside the [0, 8p) the unsigned form promises.  `bias` = p + 2^31 p (mod 2^64).
// 8 multiply-adds and one add like split_mul_add.  UNIFORM: the constant's words are in SGPRs and take the one scalar
// operand an instruction may read, so the bias comes in vector registers.
template <bool UNIFORM>
__device__ __forceinline__ uint64_t split_mul_signed(uint64_t d, uint64_t w, uint64_t wt_signed_limbs, uint64_t f_pair,
                                                    uint64_t neg_2p, uint64_t bias) {
    const uint32_t b0 = lo32(d), b1 = hi32(d);
    const uint32_t w0 = lo32(w), w1 = hi32(w), t0 = lo32(wt_signed_limbs), t1 = hi32(wt_signed_limbs);
    const uint32_t f = lo32(f_pair), ft = hi32(f_pair);
    const uint32_t n0 = lo32(neg_2p), n1 = hi32(neg_2p);
    uint64_t q, c0, c1, carry;
    // two blocks: the quotient's chain first (its words are free again once the high word is taken), then the columns
    if constexpr (UNIFORM) {
        asm("v_mad_u64_u32 %0, %1, %2, %4, 2.0\n\t"
            "v_mad_i64_i32 %0, %1, %3, %5, %0"
            : "=&v"(q), "=&s"(carry) : "v"(b0), "v"(b1), "s"(f), "s"(ft));
        asm("v_mad_u64_u32 %0, %2, %3, %5, %9\n\t"
            "v_mad_u64_u32 %1, %2, %3, %7, 0\n\t"
            "v_mad_i64_i32 %0, %2, %4, %6, %0\n\t"
            "v_mad_i64_i32 %1, %2, %4, %8, %1"
            : "=&v"(c0), "=&v"(c1), "=&s"(carry)
            : "v"(b0), "v"(b1), "s"(w0), "s"(t0), "s"(w1), "s"(t1), "v"(bias));
    } else {
        asm("v_mad_u64_u32 %0, %1, %2, %4, 2.0\n\t"
            "v_mad_i64_i32 %0, %1, %3, %5, %0"
            : "=&v"(q), "=&s"(carry) : "v"(b0), "v"(b1), "v"(f), "v"(ft));
        asm("v_mad_u64_u32 %0, %2, %3, %5, %9\n\t"
            "v_mad_u64_u32 %1, %2, %3, %7, 0\n\t"
            "v_mad_i64_i32 %0, %2, %4, %6, %0\n\t"
            "v_mad_i64_i32 %1, %2, %4, %8, %1"
            : "=&v"(c0), "=&v"(c1), "=&s"(carry)
            : "v"(b0), "v"(b1), "v"(w0), "v"(t0), "v"(w1), "v"(t1), "s"(bias));
    }
    uint64_t carry2;
    asm("v_mad_u64_u32 %0, %2, %3, %4, %0\n\t"
        "v_mad_u64_u32 %1, %2, %3, %5, %1"
        : "+v"(c0), "+v"(c1), "=&s"(carry2)
        : "v"(hi32(q)), "s"(n0), "s"(n1));
    return pack64(lo32(c0), opaque32(hi32(c0) + lo32(c1)));
}

// ---- product by a constant for moduli next to a power of two ("fold" butterflies): 5 multiply-adds -------------------
// For p = 2^b - d (the largest b-bit primes, what generatePrimes(preferringSmall: false) returns) or p = 2^60 + e (the
// BEHZ auxiliary primes, RnsTool.swift:28-66) the high part of a product folds back by a SHIFT: with y = b0 + b1 2^32 and
// the constant's (w, wt = w 2^32 mod p),  V = b0 w + b1 wt = w y (mod p)  is below 2^(b+33); cut at F = 2^(b+2):
//     minus form:  F = 4d (mod p):   r = (V mod F) + (V >> (b+2)) 4d
//     plus form:   F = -4e (mod p):  r = ((V + e) mod F) + 2^60 - ((V + e) >> 62) 4e      (e + 2^60 = p)
// r = w y (mod p), 0 <= r < F + 2^31 |4d| < 6p for ANY 64-bit y, no quotient estimate and no table of factors.  V is
// formed exactly: its low column b0 w0 + b1 t0 may carry (the carry-out of the second multiply-add, one add-with-carry to
// put it in place), its 2^32 column cannot.  5 multiply-adds, one shift, one mask, three simple operations (five in the
// plus form) against 9 multiply-adds and two 64-bit shifts for the Shoup product with its estimated quotient: 2.21 against
// 1.87 T butterflies/s with one conditional subtract each (bench_tools/butterfly_probe.hip).  The constants are scalar
// work on p (FoldConstants); eligibility is decided on the host (poly_context.cpp).
struct FoldConstants {
    uint32_t multiplier;  // 4d, or 2^32 - 4e
    uint32_t shift, mask; // V >> (b + 2) = U >> shift for U = V >> 32; mask = 2^shift - 1
    uint32_t high_bias;   // plus form: the high word of 2^60
    uint64_t addend;      // plus form: e
};
template <bool PLUS>
__device__ __forceinline__ FoldConstants fold_constants(uint64_t p) {
    FoldConstants c{};
    if constexpr (PLUS) {
        const uint64_t e = p - (uint64_t(1) << 60);
        c.multiplier = 0u - static_cast<uint32_t>(4 * e);
        c.shift = 30;
        c.high_bias = 1u << 28;
        c.addend = e;
    } else {
        const int bits = 64 - __builtin_clzll(p);
        c.multiplier = static_cast<uint32_t>(4 * ((uint64_t(1) << bits) - p));
        c.shift = static_cast<uint32_t>(bits - 30);
        c.high_bias = 0;
        c.addend = 0;
    }
    c.mask = (1u << c.shift) - 1u;
    return c;
}
// UNIFORM: the constant's words are wave-uniform (SGPRs); the addend then travels in a VGPR pair (one scalar operand per
// instruction).
template <bool UNIFORM, bool PLUS>
__device__ __forceinline__ uint64_t fold_mul(uint64_t y, uint64_t w, uint64_t wt, const FoldConstants& fc) {
    const uint32_t b0 = lo32(y), b1 = hi32(y), w0 = lo32(w), w1 = hi32(w), t0 = lo32(wt), t1 = hi32(wt);
    uint64_t a, t, carry0, carry;
    if constexpr (UNIFORM) {
        if constexpr (PLUS) {
            const uint64_t addend = opaque(fc.addend);
            asm("v_mad_u64_u32 %0, %2, %4, %6, %8\n\t"
                "v_mad_u64_u32 %1, %3, %5, %7, %0"
                : "=&v"(a), "=&v"(t), "=&s"(carry0), "=&s"(carry)
                : "v"(b0), "v"(b1), "s"(w0), "s"(t0), "v"(addend));
        } else {
            asm("v_mad_u64_u32 %0, %2, %4, %6, 0\n\t"
                "v_mad_u64_u32 %1, %3, %5, %7, %0"
                : "=&v"(a), "=&v"(t), "=&s"(carry0), "=&s"(carry)
                : "v"(b0), "v"(b1), "s"(w0), "s"(t0));
        }
    } else if constexpr (PLUS) {
        asm("v_mad_u64_u32 %0, %2, %4, %6, %8\n\t"
            "v_mad_u64_u32 %1, %3, %5, %7, %0"
            : "=&v"(a), "=&v"(t), "=&s"(carry0), "=&s"(carry)
            : "v"(b0), "v"(b1), "v"(w0), "v"(t0), "s"(fc.addend));
    } else {
        asm("v_mad_u64_u32 %0, %2, %4, %6, 0\n\t"
            "v_mad_u64_u32 %1, %3, %5, %7, %0"
            : "=&v"(a), "=&v"(t), "=&s"(carry0), "=&s"(carry)
            : "v"(b0), "v"(b1), "v"(w0), "v"(t0));
    }
    uint32_t carry_bit;
    asm("v_addc_co_u32 %0, %1, 0, 0, %1" : "=v"(carry_bit), "+s"(carry));
    uint64_t u = pack64(hi32(t), carry_bit), carry2;
    if constexpr (UNIFORM) {
        asm("v_mad_u64_u32 %0, %1, %2, %4, %0\n\t"
            "v_mad_u64_u32 %0, %1, %3, %5, %0"
            : "+v"(u), "=&s"(carry2)
            : "v"(b0), "v"(b1), "s"(w1), "s"(t1));
    } else {
        asm("v_mad_u64_u32 %0, %1, %2, %4, %0\n\t"
            "v_mad_u64_u32 %0, %1, %3, %5, %0"
            : "+v"(u), "=&s"(carry2)
            : "v"(b0), "v"(b1), "v"(w1), "v"(t1));
    }
    const uint32_t high = __builtin_amdgcn_alignbit(hi32(u), lo32(u), fc.shift);
    uint32_t kept = lo32(u) & fc.mask;
    if constexpr (PLUS) kept = kept + fc.high_bias - high;
    uint64_t r = pack64(lo32(t), kept), carry3;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(r), "=&s"(carry3) : "v"(high), "s"(fc.multiplier));
    return r;
}

__device__ __forceinline__ uint64_t shoup_mul(uint64_t x, uint64_t w, uint64_t wf, uint64_t p) {
    return csub63(shoup_lazy(x, w, wf, 0 - p), 0 - p);  // lazy < 2p < 2^63
}

// Shoup multiplication by a wave-uniform constant, canonical result: x < 2^63, w < p <= 2^62 - 1, wf the usual
// floor(w 2^64 / p).  Uses the halved factor so that every partial-product column fits the 64-bit addend of a
// v_mad_u64_u32 (a0 b1 + a1 b0 + hi(a0 b0) < 2^64 for a < 2^63, b < 2^63): q = floor(x (wf >> 1) / 2^64) exactly,
// x w - q 2p in [0, 3p).  The constants stay in SGPRs (each instruction reads at most one).
// _lazy: the product before its two folds, in [0, 3p)
__device__ __forceinline__ uint64_t shoup_mul_uniform_lazy(uint64_t x, uint64_t w, uint64_t wf, uint64_t p) {
    const uint64_t wf_half = wf >> 1, neg_2p = 0 - 2 * p;
    const uint32_t a0 = lo32(x), a1 = hi32(x), b0 = lo32(wf_half), b1 = hi32(wf_half);
    uint64_t low, cross, q, carry;
    asm("v_mad_u64_u32 %0, %3, %4, %6, 0\n\t"
        "v_lshrrev_b64 %0, 32, %0\n\t"
        "v_mad_u64_u32 %1, %3, %4, %7, %0\n\t"
        "v_mad_u64_u32 %1, %3, %5, %6, %1\n\t"
        "v_lshrrev_b64 %1, 32, %1\n\t"
        "v_mad_u64_u32 %2, %3, %5, %7, %1"
        : "=&v"(low), "=&v"(cross), "=&v"(q), "=&s"(carry)
        : "v"(a0), "v"(a1), "s"(b0), "s"(b1));
    return shoup_low64<true, false>(0, x, w, q, neg_2p);
}
__device__ __forceinline__ uint64_t shoup_mul_uniform(uint64_t x, uint64_t w, uint64_t wf, uint64_t p) {
    return csub_uniform(csub_uniform(shoup_mul_uniform_lazy(x, w, wf, p), 2 * p), p);
}

// operands canonical, p <= 2^62 - 1: every intermediate is below 2p < 2^63
__device__ __forceinline__ uint64_t add_mod(uint64_t a, uint64_t b, uint64_t p) { return csub63(a + b, 0 - p); }
__device__ __forceinline__ uint64_t sub_mod(uint64_t a, uint64_t b, uint64_t p) { return csub63(a + p - b, 0 - p); }
__device__ __forceinline__ uint64_t neg_mod(uint64_t a, uint64_t p) { return csub63(p - a, 0 - p); }
// the same with a wave-uniform modulus (a kernel-argument table entry)
__device__ __forceinline__ uint64_t add_mod_uniform(uint64_t a, uint64_t b, uint64_t p) {
    return csub63<true>(a + b, 0 - p);
}
__device__ __forceinline__ uint64_t sub_mod_uniform(uint64_t a, uint64_t b, uint64_t p) {
    return csub63<true>(a + p - b, 0 - p);
}
__device__ __forceinline__ uint64_t neg_mod_uniform(uint64_t a, uint64_t p) { return csub63<true>(p - a, 0 - p); }

// Single-word Barrett: x mod p for any 64-bit x; factor = floor(2^64 / p).  (Modulus.swift:258-263)
__device__ __forceinline__ uint64_t barrett_reduce64(uint64_t x, uint64_t p, uint64_t factor) {
    const uint64_t q = mulhi64(x, factor);
    return csub63(x - q * p, 0 - p);
}

struct U128 {
    uint64_t lo, hi;
};
__device__ __forceinline__ U128 mul_wide(uint64_t a, uint64_t b) {
    const uint32_t a0 = lo32(a), a1 = hi32(a), b0 = lo32(b), b1 = hi32(b);
    const uint64_t p00 = mul32(a0, b0);
    const uint64_t p01 = mad32(a0, b1, hi32(p00));
    const uint64_t p10 = mad32(a1, b0, lo32(p01));
    U128 r;
    r.lo = pack64(lo32(p00), lo32(p10));
    r.hi = mad32(a1, b1, static_cast<uint64_t>(hi32(p01)) + hi32(p10));
    return r;
}
__device__ __forceinline__ void add128(U128& acc, U128 v) {  // wrapping
    const uint64_t lo = acc.lo + v.lo;
    acc.hi += v.hi + (lo < acc.lo ? 1 : 0);
    acc.lo = lo;
}
__device__ __forceinline__ void mac128(U128& acc, uint64_t a, uint64_t b) { add128(acc, mul_wide(a, b)); }

// ---- sums of 64x64 products without a carry chain per term ---------------------------------------------------
// sum = t + c 2^32 + h 2^64 + (t_carry + c_carry 2^32) 2^64: every partial-product column rides the 64-bit addend of
// its own v_mad_u64_u32 and the carry-outs are counted (7 instructions per product instead of ~18 for a 128-bit
// multiply-add).  Exact for any operands while h does not wrap (65 536 products of 62-bit operands).
struct ProductSum {
    uint64_t t, c, h;
    uint32_t t_carry, c_carry;
};
__device__ __forceinline__ ProductSum product_sum_zero() { return ProductSum{0, 0, 0, 0, 0}; }
// b in VGPRs
__device__ __forceinline__ void product_sum_add(ProductSum& s, uint64_t a, uint64_t b) {
    uint64_t carry;
    asm("v_mad_u64_u32 %0, %5, %6, %8, %0\n\t"
        "v_addc_co_u32 %3, %5, 0, %3, %5\n\t"
        "v_mad_u64_u32 %1, %5, %6, %9, %1\n\t"
        "v_addc_co_u32 %4, %5, 0, %4, %5\n\t"
        "v_mad_u64_u32 %1, %5, %7, %8, %1\n\t"
        "v_addc_co_u32 %4, %5, 0, %4, %5\n\t"
        "v_mad_u64_u32 %2, %5, %7, %9, %2"
        : "+v"(s.t), "+v"(s.c), "+v"(s.h), "+v"(s.t_carry), "+v"(s.c_carry), "=&s"(carry)
        : "v"(lo32(a)), "v"(hi32(a)), "v"(lo32(b)), "v"(hi32(b)));
}
// Two or three sums that take products with the SAME b, instruction by instruction side by side: inside one sum every
// instruction waits for the carry its predecessor leaves in an SGPR pair, and a wavefront issues in order; neighbours
// with their own carry registers fill the gaps.
// NARROW: both operands are below 2^56 (canonical residues of moduli below 2^56), so a cross term a0 b1 + a1 b0 is below
// 2^57 and the middle column takes 127 products before it can wrap: its two carry counts are dropped (5 instructions
// per product instead of 7); the caller folds the sum at least every kNarrowProductSumCadence products.
constexpr uint64_t kNarrowProductSumCadence = 64;
template <bool NARROW>
__device__ __forceinline__ void product_sum_add_one(ProductSum& s, uint64_t a, uint64_t b) {
    if constexpr (!NARROW) {
        product_sum_add(s, a, b);
    } else {
        uint64_t carry;
        asm("v_mad_u64_u32 %0, %4, %5, %7, %0\n\t"
            "v_addc_co_u32 %3, %4, 0, %3, %4\n\t"
            "v_mad_u64_u32 %1, %4, %5, %8, %1\n\t"
            "v_mad_u64_u32 %1, %4, %6, %7, %1\n\t"
            "v_mad_u64_u32 %2, %4, %6, %8, %2"
            : "+v"(s.t), "+v"(s.c), "+v"(s.h), "+v"(s.t_carry), "=&s"(carry)
            : "v"(lo32(a)), "v"(hi32(a)), "v"(lo32(b)), "v"(hi32(b)));
    }
}
template <bool NARROW>
__device__ __forceinline__ void product_sum_add_pair(ProductSum& s0, ProductSum& s1, uint64_t a0, uint64_t a1, uint64_t b) {
    uint64_t carry0, carry1;
    if constexpr (!NARROW) {
        asm("v_mad_u64_u32 %0, %10, %12, %16, %0\n\t"
            "v_mad_u64_u32 %5, %11, %14, %16, %5\n\t"
            "v_addc_co_u32 %3, %10, 0, %3, %10\n\t"
            "v_addc_co_u32 %8, %11, 0, %8, %11\n\t"
            "v_mad_u64_u32 %1, %10, %12, %17, %1\n\t"
            "v_mad_u64_u32 %6, %11, %14, %17, %6\n\t"
            "v_addc_co_u32 %4, %10, 0, %4, %10\n\t"
            "v_addc_co_u32 %9, %11, 0, %9, %11\n\t"
            "v_mad_u64_u32 %1, %10, %13, %16, %1\n\t"
            "v_mad_u64_u32 %6, %11, %15, %16, %6\n\t"
            "v_addc_co_u32 %4, %10, 0, %4, %10\n\t"
            "v_addc_co_u32 %9, %11, 0, %9, %11\n\t"
            "v_mad_u64_u32 %2, %10, %13, %17, %2\n\t"
            "v_mad_u64_u32 %7, %11, %15, %17, %7"
            : "+v"(s0.t), "+v"(s0.c), "+v"(s0.h), "+v"(s0.t_carry), "+v"(s0.c_carry),  // 0-4
              "+v"(s1.t), "+v"(s1.c), "+v"(s1.h), "+v"(s1.t_carry), "+v"(s1.c_carry),  // 5-9
              "=&s"(carry0), "=&s"(carry1)                                              // 10, 11
            : "v"(lo32(a0)), "v"(hi32(a0)), "v"(lo32(a1)), "v"(hi32(a1)), "v"(lo32(b)), "v"(hi32(b)));  // 12-17
    } else {
        asm("v_mad_u64_u32 %0, %8, %10, %14, %0\n\t"
            "v_mad_u64_u32 %4, %9, %12, %14, %4\n\t"
            "v_addc_co_u32 %3, %8, 0, %3, %8\n\t"
            "v_addc_co_u32 %7, %9, 0, %7, %9\n\t"
            "v_mad_u64_u32 %1, %8, %10, %15, %1\n\t"
            "v_mad_u64_u32 %5, %9, %12, %15, %5\n\t"
            "v_mad_u64_u32 %1, %8, %11, %14, %1\n\t"
            "v_mad_u64_u32 %5, %9, %13, %14, %5\n\t"
            "v_mad_u64_u32 %2, %8, %11, %15, %2\n\t"
            "v_mad_u64_u32 %6, %9, %13, %15, %6"
            : "+v"(s0.t), "+v"(s0.c), "+v"(s0.h), "+v"(s0.t_carry),  // 0-3
              "+v"(s1.t), "+v"(s1.c), "+v"(s1.h), "+v"(s1.t_carry),  // 4-7
              "=&s"(carry0), "=&s"(carry1)                            // 8, 9
            : "v"(lo32(a0)), "v"(hi32(a0)), "v"(lo32(a1)), "v"(hi32(a1)), "v"(lo32(b)), "v"(hi32(b)));  // 10-15
    }
}
template <bool NARROW>
__device__ __forceinline__ void product_sum_add_triple(ProductSum& s0, ProductSum& s1, ProductSum& s2, uint64_t a0,
                                                       uint64_t a1, uint64_t a2, uint64_t b) {
    uint64_t carry0, carry1, carry2;
    if constexpr (!NARROW) {
        asm("v_mad_u64_u32 %0, %15, %18, %24, %0\n\t"
            "v_mad_u64_u32 %5, %16, %20, %24, %5\n\t"
            "v_mad_u64_u32 %10, %17, %22, %24, %10\n\t"
            "v_addc_co_u32 %3, %15, 0, %3, %15\n\t"
            "v_addc_co_u32 %8, %16, 0, %8, %16\n\t"
            "v_addc_co_u32 %13, %17, 0, %13, %17\n\t"
            "v_mad_u64_u32 %1, %15, %18, %25, %1\n\t"
            "v_mad_u64_u32 %6, %16, %20, %25, %6\n\t"
            "v_mad_u64_u32 %11, %17, %22, %25, %11\n\t"
            "v_addc_co_u32 %4, %15, 0, %4, %15\n\t"
            "v_addc_co_u32 %9, %16, 0, %9, %16\n\t"
            "v_addc_co_u32 %14, %17, 0, %14, %17\n\t"
            "v_mad_u64_u32 %1, %15, %19, %24, %1\n\t"
            "v_mad_u64_u32 %6, %16, %21, %24, %6\n\t"
            "v_mad_u64_u32 %11, %17, %23, %24, %11\n\t"
            "v_addc_co_u32 %4, %15, 0, %4, %15\n\t"
            "v_addc_co_u32 %9, %16, 0, %9, %16\n\t"
            "v_addc_co_u32 %14, %17, 0, %14, %17\n\t"
            "v_mad_u64_u32 %2, %15, %19, %25, %2\n\t"
            "v_mad_u64_u32 %7, %16, %21, %25, %7\n\t"
            "v_mad_u64_u32 %12, %17, %23, %25, %12"
            : "+v"(s0.t), "+v"(s0.c), "+v"(s0.h), "+v"(s0.t_carry), "+v"(s0.c_carry),  // 0-4
              "+v"(s1.t), "+v"(s1.c), "+v"(s1.h), "+v"(s1.t_carry), "+v"(s1.c_carry),  // 5-9
              "+v"(s2.t), "+v"(s2.c), "+v"(s2.h), "+v"(s2.t_carry), "+v"(s2.c_carry),  // 10-14
              "=&s"(carry0), "=&s"(carry1), "=&s"(carry2)                               // 15-17
            : "v"(lo32(a0)), "v"(hi32(a0)), "v"(lo32(a1)), "v"(hi32(a1)), "v"(lo32(a2)), "v"(hi32(a2)),  // 18-23
              "v"(lo32(b)), "v"(hi32(b)));                                                              // 24, 25
    } else {
        asm("v_mad_u64_u32 %0, %12, %15, %21, %0\n\t"
            "v_mad_u64_u32 %4, %13, %17, %21, %4\n\t"
            "v_mad_u64_u32 %8, %14, %19, %21, %8\n\t"
            "v_addc_co_u32 %3, %12, 0, %3, %12\n\t"
            "v_addc_co_u32 %7, %13, 0, %7, %13\n\t"
            "v_addc_co_u32 %11, %14, 0, %11, %14\n\t"
            "v_mad_u64_u32 %1, %12, %15, %22, %1\n\t"
            "v_mad_u64_u32 %5, %13, %17, %22, %5\n\t"
            "v_mad_u64_u32 %9, %14, %19, %22, %9\n\t"
            "v_mad_u64_u32 %1, %12, %16, %21, %1\n\t"
            "v_mad_u64_u32 %5, %13, %18, %21, %5\n\t"
            "v_mad_u64_u32 %9, %14, %20, %21, %9\n\t"
            "v_mad_u64_u32 %2, %12, %16, %22, %2\n\t"
            "v_mad_u64_u32 %6, %13, %18, %22, %6\n\t"
            "v_mad_u64_u32 %10, %14, %20, %22, %10"
            : "+v"(s0.t), "+v"(s0.c), "+v"(s0.h), "+v"(s0.t_carry),  // 0-3
              "+v"(s1.t), "+v"(s1.c), "+v"(s1.h), "+v"(s1.t_carry),  // 4-7
              "+v"(s2.t), "+v"(s2.c), "+v"(s2.h), "+v"(s2.t_carry),  // 8-11
              "=&s"(carry0), "=&s"(carry1), "=&s"(carry2)             // 12-14
            : "v"(lo32(a0)), "v"(hi32(a0)), "v"(lo32(a1)), "v"(hi32(a1)), "v"(lo32(a2)), "v"(hi32(a2)),  // 15-20
              "v"(lo32(b)), "v"(hi32(b)));                                                              // 21, 22
    }
}
// POLYS sums against one b, side by side in groups of three and two
template <int POLYS, bool NARROW>
__device__ __forceinline__ void product_sum_add_all(ProductSum (&s)[POLYS], const uint64_t (&a)[POLYS], uint64_t b) {
    constexpr int kTriples = POLYS % 3 == 0 ? POLYS / 3 : POLYS % 3 == 2 ? POLYS / 3 : POLYS >= 4 ? POLYS / 3 - 1 : 0;
    constexpr int kPairs = (POLYS - 3 * kTriples) / 2;
#pragma unroll
    for (int g = 0; g < kTriples; ++g)
        product_sum_add_triple<NARROW>(s[3 * g], s[3 * g + 1], s[3 * g + 2], a[3 * g], a[3 * g + 1], a[3 * g + 2], b);
#pragma unroll
    for (int g = 0; g < kPairs; ++g) {
        constexpr int base = 3 * kTriples;
        product_sum_add_pair<NARROW>(s[base + 2 * g], s[base + 2 * g + 1], a[base + 2 * g], a[base + 2 * g + 1], b);
    }
    if constexpr (3 * kTriples + 2 * kPairs < POLYS) product_sum_add_one<NARROW>(s[POLYS - 1], a[POLYS - 1], b);
}
// b wave-uniform (a table constant): stays in SGPRs
__device__ __forceinline__ void product_sum_add_uniform(ProductSum& s, uint64_t a, uint64_t b) {
    uint64_t carry;
    asm("v_mad_u64_u32 %0, %5, %6, %8, %0\n\t"
        "v_addc_co_u32 %3, %5, 0, %3, %5\n\t"
        "v_mad_u64_u32 %1, %5, %6, %9, %1\n\t"
        "v_addc_co_u32 %4, %5, 0, %4, %5\n\t"
        "v_mad_u64_u32 %1, %5, %7, %8, %1\n\t"
        "v_addc_co_u32 %4, %5, 0, %4, %5\n\t"
        "v_mad_u64_u32 %2, %5, %7, %9, %2"
        : "+v"(s.t), "+v"(s.c), "+v"(s.h), "+v"(s.t_carry), "+v"(s.c_carry), "=&s"(carry)
        : "v"(lo32(a)), "v"(hi32(a)), "s"(lo32(b)), "s"(hi32(b)));
}
// A sum that starts with its first product (b wave-uniform): no zeroed accumulator to add onto and only the one carry
// the first product can produce (the cross column).
__device__ __forceinline__ ProductSum product_sum_first_uniform(uint64_t a, uint64_t b) {
    ProductSum s;
    uint64_t carry;
    asm("v_mad_u64_u32 %0, %4, %5, %7, 0\n\t"
        "v_mad_u64_u32 %1, %4, %5, %8, 0\n\t"
        "v_mad_u64_u32 %1, %4, %6, %7, %1\n\t"
        "v_addc_co_u32 %3, %4, 0, 0, %4\n\t"
        "v_mad_u64_u32 %2, %4, %6, %8, 0"
        : "=&v"(s.t), "=&v"(s.c), "=&v"(s.h), "=&v"(s.c_carry), "=&s"(carry)
        : "v"(lo32(a)), "v"(hi32(a)), "s"(lo32(b)), "s"(hi32(b)));
    s.t_carry = 0;
    return s;
}

// Short sums whose cross column cannot wrap -- terms (hi32(a_max) + hi32(b_max) + 2) <= 2^32, e.g. four residues of
// 55-bit moduli times 61-bit table constants -- skip its carry counts: five instructions per product instead of seven,
// four for the first one (the callers check the bound on the host: RnsToolDevice::wide_reduce_ok).
__device__ __forceinline__ void product_sum_add_uniform_short(ProductSum& s, uint64_t a, uint64_t b) {
    uint64_t carry;
    asm("v_mad_u64_u32 %0, %4, %5, %7, %0\n\t"
        "v_addc_co_u32 %3, %4, 0, %3, %4\n\t"
        "v_mad_u64_u32 %1, %4, %5, %8, %1\n\t"
        "v_mad_u64_u32 %1, %4, %6, %7, %1\n\t"
        "v_mad_u64_u32 %2, %4, %6, %8, %2"
        : "+v"(s.t), "+v"(s.c), "+v"(s.h), "+v"(s.t_carry), "=&s"(carry)
        : "v"(lo32(a)), "v"(hi32(a)), "s"(lo32(b)), "s"(hi32(b)));
}
__device__ __forceinline__ ProductSum product_sum_first_uniform_short(uint64_t a, uint64_t b) {
    ProductSum s;
    uint64_t carry;
    asm("v_mad_u64_u32 %0, %3, %4, %6, 0\n\t"
        "v_mad_u64_u32 %1, %3, %4, %7, 0\n\t"
        "v_mad_u64_u32 %1, %3, %5, %6, %1\n\t"
        "v_mad_u64_u32 %2, %3, %5, %7, 0"
        : "=&v"(s.t), "=&v"(s.c), "=&v"(s.h), "=&s"(carry)
        : "v"(lo32(a)), "v"(hi32(a)), "s"(lo32(b)), "s"(hi32(b)));
    s.t_carry = 0;
    s.c_carry = 0;
    return s;
}

// the same with b in VGPRs
__device__ __forceinline__ ProductSum product_sum_first(uint64_t a, uint64_t b) {
    ProductSum s;
    uint64_t carry;
    asm("v_mad_u64_u32 %0, %4, %5, %7, 0\n\t"
        "v_mad_u64_u32 %1, %4, %5, %8, 0\n\t"
        "v_mad_u64_u32 %1, %4, %6, %7, %1\n\t"
        "v_addc_co_u32 %3, %4, 0, 0, %4\n\t"
        "v_mad_u64_u32 %2, %4, %6, %8, 0"
        : "=&v"(s.t), "=&v"(s.c), "=&v"(s.h), "=&v"(s.c_carry), "=&s"(carry)
        : "v"(lo32(a)), "v"(hi32(a)), "v"(lo32(b)), "v"(hi32(b)));
    s.t_carry = 0;
    return s;
}

// the sum mod 2^128: lo = t + (c << 32), hi = h + t_carry + (c >> 32) + (c_carry << 32) + carry(lo), as one five-
// instruction carry chain through an SGPR pair (the C form costs twice that in 64-bit adds, compares and moves)
__device__ __forceinline__ U128 product_sum_value(const ProductSum& s) {
    uint32_t lo_high, hi_low, hi_high;
    uint64_t carry;
    asm("v_add_co_u32 %0, %3, %4, %5\n\t"          // t_hi + c_lo
        "v_addc_co_u32 %1, %3, %6, %7, %3\n\t"     // h_lo + c_hi + carry
        "v_addc_co_u32 %2, %3, %8, %9, %3\n\t"     // h_hi + c_carry + carry
        "v_add_co_u32 %1, %3, %1, %10\n\t"         // + t_carry
        "v_addc_co_u32 %2, %3, 0, %2, %3"
        : "=&v"(lo_high), "=&v"(hi_low), "=&v"(hi_high), "=&s"(carry)
        : "v"(hi32(s.t)), "v"(lo32(s.c)), "v"(lo32(s.h)), "v"(hi32(s.c)), "v"(hi32(s.h)), "v"(s.c_carry),
          "v"(s.t_carry));
    U128 r;
    r.lo = pack64(lo32(s.t), lo_high);
    r.hi = pack64(hi_low, hi_high);
    return r;
}

// x mod p for any 64-bit x with wave-uniform p and factor = floor(2^64 / p).  For p >= 2^32 the factor is a single
// 32-bit word and the quotient estimate is two multiply-adds; smaller moduli take the general path.
// _lazy: before the final fold, in [0, 2p)
__device__ __forceinline__ uint64_t barrett_reduce64_uniform_lazy(uint64_t x, uint64_t p, uint64_t factor) {
    if (hi32(factor) != 0) return barrett_reduce64(x, p, factor);  // uniform branch: p < 2^32
    uint64_t t, qp, carry, carry2;
    uint32_t q_high;
    asm("v_mad_u64_u32 %0, %1, %2, %4, 0\n\t"       // x0 f
        "v_lshrrev_b64 %0, 32, %0\n\t"
        "v_mad_u64_u32 %0, %1, %3, %4, %0\n\t"      // x1 f + hi32(x0 f): its high word is q = floor(x f / 2^64)
        "v_lshrrev_b64 %0, 32, %0"
        : "=&v"(t), "=&s"(carry)
        : "v"(lo32(x)), "v"(hi32(x)), "s"(lo32(factor)));
    const uint32_t q = lo32(t);
    asm("v_mul_lo_u32 %1, %3, %5\n\t"               // q p1
        "v_mad_u64_u32 %0, %2, %3, %4, 0"             // q p0
        : "=&v"(qp), "=&v"(q_high), "=&s"(carry2)
        : "v"(q), "s"(lo32(p)), "s"(hi32(p)));
    const uint64_t q_times_p = pack64(lo32(qp), hi32(qp) + q_high);
    return x - q_times_p;
}
__device__ __forceinline__ uint64_t barrett_reduce64_uniform(uint64_t x, uint64_t p, uint64_t factor) {
    return csub63<true>(barrett_reduce64_uniform_lazy(x, p, factor), 0 - p);
}

// Canonical residue of a ProductSum whose value is < 2^127 (at most 8 products of operands < 2^62):
// (hi 2^64 + lo) mod p = ((hi (2^64 mod p)) mod p + lo mod p) mod p.
template <typename Modulus>
__device__ __forceinline__ uint64_t reduce_product_sum(const ProductSum& s, const Modulus& m) {
    const U128 v = product_sum_value(s);
    const uint64_t high = shoup_mul_uniform(v.hi, m.two64_mod_p, m.two64_mod_p_shoup, m.p);
    const uint64_t low = barrett_reduce64_uniform(v.lo, m.p, m.barrett64);
    return add_mod_uniform(high, low, m.p);
}
// The same residue class, unfolded: a value in [0, 5p) (needs 5p < 2^64; callers that feed it to another exact product
// or fold a whole sum at once skip four conditional subtracts).
template <typename Modulus>
__device__ __forceinline__ uint64_t reduce_product_sum_lazy(const ProductSum& s, const Modulus& m) {
    const U128 v = product_sum_value(s);
    return shoup_mul_uniform_lazy(v.hi, m.two64_mod_p, m.two64_mod_p_shoup, m.p) +
           barrett_reduce64_uniform_lazy(v.lo, m.p, m.barrett64);
}

// The same residues for a sum KNOWN to be below 2^(64 + sh), sh = m.wide_shift = bits(p) - 1 (2^33 < p < 2^61, hence
// 32 <= sh <= 60) -- the dot products of the base conversions, whose terms are residues times table constants.  Barrett
// with a one-word quotient: x = T >> sh fits a word, mu = floor(2^(64+sh) / p) < 2^64, floor(x mu / 2^64) is at most 2
// below floor(T / p); the high word of x mu is taken from three of its four partial products (at most 2 lower), so
// T - q p lies in [0, 5p) -- two alignbits, three multiply-adds for q, three for the low word of T + q (2^64 - p):
// 24 issue slots against 45 for the split at 2^64 above (_lazy) and 39 against 64 for the canonical residue.
template <typename Modulus>
__device__ __forceinline__ uint64_t reduce_product_sum_bounded_lazy(const ProductSum& s, const Modulus& m) {
    const U128 v = product_sum_value(s);
    const uint32_t shift = m.wide_shift - 32u;  // wave-uniform, 0..28
    const uint32_t x0 = __builtin_amdgcn_alignbit(lo32(v.hi), hi32(v.lo), shift);
    const uint32_t x1 = __builtin_amdgcn_alignbit(hi32(v.hi), lo32(v.hi), shift);
    const uint64_t q = shoup_quotient<true, true>(pack64(x0, x1), m.wide_factor);
    const uint64_t neg_p = 0 - m.p;
    uint64_t c0, c1, carry;
    asm("v_mad_u64_u32 %0, %2, %3, %5, %7\n\t"
        "v_mad_u64_u32 %1, %2, %3, %6, 0\n\t"
        "v_mad_u64_u32 %1, %2, %4, %5, %1"
        : "=&v"(c0), "=&v"(c1), "=&s"(carry)
        : "v"(lo32(q)), "v"(hi32(q)), "s"(lo32(neg_p)), "s"(hi32(neg_p)), "v"(v.lo));
    return pack64(lo32(c0), opaque32(hi32(c0) + lo32(c1)));
}
template <typename Modulus>
__device__ __forceinline__ uint64_t reduce_product_sum_bounded(const ProductSum& s, const Modulus& m) {
    return csub_uniform(csub_uniform(csub_uniform(reduce_product_sum_bounded_lazy(s, m), 4 * m.p), 2 * m.p), m.p);
}

// Barrett on a product x*y < p^2 (Modulus.swift:349-360): factor = floor(2^(bits(p)+62)/p), shift = bits(p)-2.
__device__ __forceinline__ uint64_t barrett_mul(uint64_t x, uint64_t y, uint64_t p, uint64_t factor, int shift) {
    const U128 prod = mul_wide(x, y);
    // shift in [0, 60]; p >= 2 => bits(p) >= 2
    const uint64_t shifted = shift == 0 ? prod.lo : ((prod.lo >> shift) | (prod.hi << (64 - shift)));
    const uint64_t q = mulhi64(shifted, factor);
    return csub63(prod.lo - q * p, 0 - p);
}

// Double-word Barrett: x mod p for any 128-bit x; factor = floor(2^128 / p) as (lo, hi).  (Modulus.swift:319-325)
// qHat = high 128 bits of x * factor; only its low word matters because z = x - qHat*p is taken mod 2^64 after
// noting z < 2p < 2^64.
__device__ __forceinline__ uint64_t barrett_reduce128(U128 x, uint64_t p, uint64_t f_lo, uint64_t f_hi) {
    // x * f = x.lo*f_lo + (x.lo*f_hi + x.hi*f_lo) * 2^64 + x.hi*f_hi * 2^128 ; we need bits [128, 192)
    const uint64_t ll_hi = mulhi64(x.lo, f_lo);
    const U128 lh = mul_wide(x.lo, f_hi);
    const U128 hl = mul_wide(x.hi, f_lo);
    // middle column (bits 64..127): ll_hi + lh.lo + hl.lo -> carry into bits 128+
    uint64_t mid = ll_hi + lh.lo;
    uint64_t carry = mid < ll_hi ? 1 : 0;
    const uint64_t mid2 = mid + hl.lo;
    carry += mid2 < mid ? 1 : 0;
    const uint64_t q_lo = lh.hi + hl.hi + carry + x.hi * f_hi;  // low word of bits [128, 192), wrapping
    return csub63(x.lo - q_lo * p, 0 - p);
}

}  // namespace heamd
