// ntt_common.hpp -- device helpers shared by the NTT kernels: register/lane <-> element mapping, padded LDS tile,
// global <-> register movement, Harvey/Shoup butterfly passes.  See ntt_kernels.hip for the design notes.
#pragma once

#include <hip/hip_runtime.h>

#include "device_context.hpp"
#include "device_math.hpp"

namespace heamd {
namespace ntt {

// ---- LDS tile addressing: word index -> padded word index (all accesses are 8-byte ds_read/write_b64) ----------
// +1 word per 8 words de-conflicts the stride-8/16 reads of the last pass; +8 words per 256 de-conflicts the
// middle pass whose lanes are 256 words apart (bank math in DESIGN.md).
// lds_slot is linear over the bits of idx (a sum of weighted bits, no carries), hence
// lds_slot(a | b) == lds_slot(a) + lds_slot(b) for disjoint a, b: the lane part is computed once per pass and the
// register part is a compile-time immediate offset of the ds_read/ds_write.
__host__ __device__ constexpr uint32_t lds_slot(uint32_t idx) { return idx + (idx >> 3) + ((idx >> 8) << 3); }
constexpr uint32_t lds_words(uint32_t n) { return n + (n >> 3) + ((n >> 8) << 3) + 8; }

// The padding above serves the two transposes next to the low passes; the layouts of the upper passes read and write
// runs of consecutive words, which it puts two to a bank (bench_tools/lds_bank_model.py: 8 / 4 cycles per store / load
// instruction against 4 / 2; SQ_LDS_BANK_CONFLICT = 43 % of SQ_LDS_IDX_ACTIVE on the N = 8192 kernels).  A transpose
// only has to suit ITS two layouts, so the headline kernel (N = 8192, 8 words per lane) pads per transpose:
//   scheme 1  passes on bits 10.. <-> 7..   runs of consecutive words on both sides: no padding
//   scheme 2  passes on bits 7..  <-> 4..   +16 words per 128: the lane bit that jumps 128 words lands 16 banks away
//   scheme 3  passes on bits 4..  <-> 1..   +2 words per 16
//   scheme 0  everything else (the rule above)
// -- every load and store of schemes 1-3 at its ideal cycle count.  All schemes keep a wave's 512-word block in the
// same 592 slots (the padding acts inside the block only): between two wave-private transposes nothing orders one
// wave's stores against its neighbours' loads, so the blocks must not move.  Each scheme is still a sum of per-bit
// weights: lds_slot(a | b) == lds_slot(a) + lds_slot(b) for disjoint a, b.
template <int SCHEME>
__host__ __device__ constexpr uint32_t lds_slot_scheme(uint32_t idx) {
    if constexpr (SCHEME == 0) {
        return lds_slot(idx);
    } else {
        const uint32_t block = (idx >> 9) * 592u, within = idx & 511u;
        if constexpr (SCHEME == 1) return block + within;
        if constexpr (SCHEME == 2) return block + within + ((within >> 7) << 4);
        return block + within + ((within >> 4) << 1);
    }
}
static_assert(lds_slot(512) == 592 && lds_slot_scheme<3>(511) < 592, "all schemes share the 592-slot wave blocks");
// the scheme of the transpose between the passes on bits [LO_A, ..) and [LO_B, ..)
template <int LOGN, int LOGE, int LO_A, int LO_B>
constexpr int transpose_scheme() {
    constexpr int low = LO_A < LO_B ? LO_A : LO_B;
    if constexpr (LOGN == 13 && LOGE == 3) return low == 7 ? 1 : low == 4 ? 2 : low == 1 ? 3 : 0;
    return 0;
}

// element index held in register r of lane tid during a pass over element bits [LO, LO + W)
template <int LOGN, int LOGE, int LO, int W>
__host__ __device__ constexpr uint32_t element_index(uint32_t r, uint32_t tid) {
    if constexpr (W == LOGE) {
        return ((tid >> LO) << (LO + LOGE)) | (r << LO) | (tid & ((1u << LO) - 1u));
    } else {
        static_assert(LO == 0, "a partial pass sits on the low bits");
        // The lane keeps 2^W contiguous words per run; its X = LOGE - W extra register bits sit just below the
        // wave-id bits, so that a wave owns the SAME elements as in the preceding full pass (wave = top bits of the
        // index in both) and the transpose between them never leaves the wave.
        constexpr int X = LOGE - W;
        constexpr int WB = (LOGN - LOGE) > 6 ? (LOGN - LOGE) - 6 : 0;  // wave-id bits of the lane index
        const uint32_t wave = tid >> 6, lane = tid & 63u;
        return (wave << (LOGN - WB)) | ((r >> W) << (LOGN - WB - X)) | (lane << W) | (r & ((1u << W) - 1u));
    }
}
// element_index(r, tid) == lane_part(tid) | register_part(r) with disjoint bit fields
template <int LOGN, int LOGE, int LO, int W>
__host__ __device__ constexpr uint32_t register_part(uint32_t r) {
    return element_index<LOGN, LOGE, LO, W>(r, 0);
}
template <int LOGN, int LOGE, int LO, int W>
__device__ __forceinline__ uint32_t lane_part(uint32_t tid) {
    return element_index<LOGN, LOGE, LO, W>(0, tid);
}

// Transposes after the first pass stay inside one wave (see element_index): the LDS queue of a wave is in order, so
// a full drain of its own DS operations is all the synchronisation needed -- no workgroup barrier, no waiting for
// sibling waves that the SIMD arbitration has let fall behind.
__device__ __forceinline__ void wave_private_lds_fence() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// A pass over bits [LO, LO+W) keeps "wave id = top bits of the element index" when its lane index has at least
// as many bits above LO as there are wave-id bits.  A transpose between two such passes never leaves the wave.
template <int LOGN, int LOGE, int LO>
constexpr bool kWaveOwnsTopBits = ((LOGN - LOGE) - LO) >= (((LOGN - LOGE) > 6) ? (LOGN - LOGE) - 6 : 0);

template <int LOGN, int LOGE, int LO_FROM, int LO_TO>
__device__ __forceinline__ void lds_transpose_fence() {
    if constexpr (kWaveOwnsTopBits<LOGN, LOGE, LO_FROM> && kWaveOwnsTopBits<LOGN, LOGE, LO_TO>) {
        wave_private_lds_fence();
    } else {
        __syncthreads();
    }
}

// Twiddle tables never change while a context lives: read them through the constant address space, so that a
// wave-uniform address becomes a scalar-cache load (s_load_dwordx4) wherever it sits relative to barriers.
__device__ __forceinline__ U64x2 load_twiddle(const U64x2* entry) {
    using ConstWord = const __attribute__((address_space(4))) uint64_t;
    ConstWord* const words = (ConstWord*)(entry);
    return U64x2{words[0], words[1]};
}
__device__ __forceinline__ uint64_t load_twiddle_word(const uint64_t* entry) {
    using ConstWord = const __attribute__((address_space(4))) uint64_t;
    return *(ConstWord*)(entry);
}

// Butterfly arithmetic modes (chosen per launch from the moduli the launch covers):
//   kModeExact   any p <= 2^62 - 1 : exact Shoup quotient, products in [0, 2p), values in [0, 4p)
//   kModeApprox  p < 2^61          : 3-multiply quotient estimate, products in [0, 4p), values in [0, 8p)
//   kModeSplit   2^40 <= p < 2^55  : limb-wise Shoup products (split_mul_add, device_math.hpp: 8 multiply-adds, any
//                                    64-bit multiplicand, products in [0, 8p)) and the spare top bits instead of a
//                                    conditional subtract per butterfly: the forward transform never folds (a word
//                                    gains at most 8p per stage: < (1 + 8 log2 N) p <= 113 p < 2^62), the inverse
//                                    brings its sums back under 2p once their bound reaches 2^9 p (one round for
//                                    N <= 8192); one float-estimated quotient makes outputs canonical.
//   kModeSplitSigned  inverse transforms only: the same schedule with the differences x - y multiplied as SIGNED words
//                    (split_mul_signed: no bound added to keep them non-negative; a second inverse table holds
//                    w 2^32 mod p in signed limbs for that) -- one 64-bit addition less per butterfly.  Used where it
//                    measures faster: every limb-wise inverse kernel but the plain-slab one at N = 8192
//                    (profiles/r04t_inverse_forms_ab.txt).
//   kModeFoldLazy  the same SCHEDULE (spare top bits, no conditional subtract per butterfly) for moduli just below a power
//                    of two, p = 2^b - d with d < 2^(b-32), 41 <= b <= 55 (DeviceModulus::split_shift != 0: what
//                    generatePrimes(preferringSmall: false) returns, i.e. every parameter set of the reference), with the
//                    product FOLDED BY A SHIFT at 2^(b+2) = 4d (mod p) instead of reduced by an estimated quotient
//                    (device_math.hpp fold_mul, the fold modes' product): 5 multiply-adds instead of 8, no factor table -- a
//                    gathered twiddle is its 16 bytes (w, w 2^32 mod p), 4 registers instead of 6, so three of them are kept
//                    in flight (kTwiddlesAhead) -- products below 2^(b+2) + 2^33 d < 6p for ANY 64-bit multiplicand, hence the
//                    same bounds as kModeSplit's [0, 8p).  Round 5: forward -3.4 %, inverse -9 % at N = 8192, -3 % / -5.5 % at
//                    N = 4096, relinearize +4 % (profiles/r05ad_fold_lazy_butterflies_ab.txt).  (Rounds 3-4 had a kModeSplitShift in this
//                    slot: limb-wise products whose quotient factors were read off the constants by a shift.)
//   kModeFoldMinus / kModeFoldPlus   2^55 < p < 2^60.2 next to a power of two -- p = 2^b - d, 56 <= b <= 60 (the largest
//                    b-bit primes: the reference's 60-bit parameter sets) or p = 2^60 + e (the BEHZ auxiliary primes): the
//                    product folds back by a shift (device_math.hpp fold_mul: 5 multiply-adds, products in [0, 6p), no
//                    factor table), values in [0, 14p) with one conditional subtract per butterfly like the [0, 8p)
//                    schedule it replaces where the moduli allow it (DeviceContext::fold_minus_mask / fold_plus_mask).
constexpr int kModeExact = 0, kModeApprox = 1, kModeSplit = 3, kModeFoldLazy = 4, kModeFoldMinus = 5, kModeFoldPlus = 6,
              kModeSplitSigned = 7;
constexpr bool is_split(int mode) { return mode == kModeSplit || mode == kModeFoldLazy || mode == kModeSplitSigned; }
constexpr bool is_fold(int mode) { return mode == kModeFoldMinus || mode == kModeFoldPlus; }
template <int MODE>
__device__ __forceinline__ FoldConstants mode_fold_constants(uint64_t p) {
    if constexpr (is_fold(MODE) || MODE == kModeFoldLazy) return fold_constants<MODE == kModeFoldPlus>(p);
    else return FoldConstants{};
}

using BufferResource = __amdgpu_buffer_rsrc_t;
typedef unsigned int Dwordx2 __attribute__((ext_vector_type(2)));
typedef unsigned int Dwordx4 __attribute__((ext_vector_type(4)));

// Buffer descriptor over `bytes` at a wave-uniform address: loads and stores through it take one 32-bit lane offset
// plus a scalar offset -- no 64-bit address arithmetic on the vector ALU.
__device__ __forceinline__ BufferResource make_resource(const void* base, uint32_t bytes) {
    // `base` must be wave-uniform (kernel arguments and blockIdx only): the descriptor then lives in SGPRs
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, static_cast<int>(bytes), 0x00020000);
}

// The same for an address and a size the compiler cannot PROVE wave-uniform (run-time term counts and row indices that went
// through vector registers on their way): both are pinned to scalar registers explicitly -- a descriptor in vector
// registers makes every load a "waterfall" loop over its distinct values.
__device__ __forceinline__ BufferResource make_uniform_resource(const void* base, uint32_t bytes) {
    const uint64_t address = reinterpret_cast<uint64_t>(base);
    const uint64_t pinned = pack64(__builtin_amdgcn_readfirstlane(lo32(address)), __builtin_amdgcn_readfirstlane(hi32(address)));
    return make_resource(reinterpret_cast<const void*>(pinned), __builtin_amdgcn_readfirstlane(bytes));
}

// The tables one residue row's butterflies read.  Exact / approx: `pairs` = (w, floor(w 2^64 / p)).
// Split: `pairs` = (w, w 2^32 mod p), `factors` = floor(w 2^32 / 2p) | floor((w 2^32 mod p) 2^32 / 2p) << 32.
template <int MODE>
struct Twiddles {
    const U64x2* pairs;
    const uint64_t* factors;
    BufferResource pair_resource, factor_resource;  // split mode gathers
    // Lane-major stage blocks (DeviceContext::*_split_pairs_lanes): within the block of a stage whose lanes hold 2^g twiddles
    // each (g = the pass's register bits above the stage's bit), entry o sits at (o mod 2^g) (m / 2^g) + (o >> g) -- the k-th
    // twiddle of all lanes of a wave is then one contiguous run (whole cache lines per request) instead of every 2^g-th entry
    // of a run 2^g times as long, fetched 2^g times over unless the vector L1 still holds it.
    bool lanes;
    // `skip`: the table as seen from entry `skip` on (the sub-transforms of an interleaved row index the tail of the
    // inverse table, ntt_kernels.hip ntt_inverse_interleaved)
    __device__ __forceinline__ Twiddles(const DeviceContext& ctx, bool inverse, uint32_t modulus_index, int log_degree,
                                        uint32_t skip = 0, int lane_major = 0) {
        const size_t at = (static_cast<size_t>(modulus_index) << log_degree) + skip;
        lanes = lane_major != 0;
        if constexpr (is_split(MODE) || is_fold(MODE)) {  // (w, w 2^32 mod p); the fold butterflies use no factors
            pairs = (inverse ? (MODE == kModeSplitSigned ? ctx.inverse_split_pairs_signed : ctx.inverse_split_pairs)
                             : ctx.forward_split_pairs) + at;
            factors = (inverse ? ctx.inverse_split_factors : ctx.forward_split_factors) + at;
            // (lane_major is a constant at every call site: 1 = the pass partition with the partial pass on the low bits,
            // 2 = on the top bit -- the plain-slab inverse at N = 8192)
            if (lane_major == 1) {
                pairs = (inverse ? (MODE == kModeSplitSigned ? ctx.inverse_split_pairs_signed_lanes : ctx.inverse_split_pairs_lanes)
                                 : ctx.forward_split_pairs_lanes) + at;
                factors = (inverse ? ctx.inverse_split_factors_lanes : ctx.forward_split_factors_lanes) + at;
            } else if (lane_major == 2) {
                pairs = ctx.inverse_split_pairs_lanes_top + at;
                factors = ctx.inverse_split_factors_lanes_top + at;
            }
            pair_resource = make_resource(pairs, (16u << log_degree) - 16u * skip);
            factor_resource = make_resource(factors, (8u << log_degree) - 8u * skip);
        } else {
            pairs = (inverse ? ctx.inverse_twiddles : ctx.forward_twiddles) + at;
            factors = nullptr;
        }
    }
};

// one twiddle in registers
struct TwiddleWords {
    uint64_t w, second, factors;  // second = Shoup factor (exact / approx) or w 2^32 mod p (split)
};
template <int MODE, bool UNIFORM>
__device__ __forceinline__ TwiddleWords fetch_twiddle(const Twiddles<MODE>& tw, uint32_t lane_index, uint32_t fixed_index) {
    TwiddleWords t;
    if constexpr (UNIFORM) {  // lane_index is wave-uniform (held in an SGPR): scalar-cache loads
        const U64x2 pair = load_twiddle(tw.pairs + fixed_index + lane_index);
        t.w = pair.x;
        t.second = pair.y;
        t.factors = 0;
        if constexpr (MODE == kModeSplit || MODE == kModeSplitSigned) t.factors = load_twiddle_word(tw.factors + fixed_index + lane_index);
    } else if constexpr (MODE == kModeSplit || MODE == kModeSplitSigned) {
        const Dwordx4 pair = __builtin_amdgcn_raw_buffer_load_b128(tw.pair_resource, lane_index << 4, fixed_index << 4, 0);
        const Dwordx2 factors = __builtin_amdgcn_raw_buffer_load_b64(tw.factor_resource, lane_index << 3, fixed_index << 3, 0);
        t.w = pack64(pair.x, pair.y);
        t.second = pack64(pair.z, pair.w);
        t.factors = pack64(factors.x, factors.y);
    } else if constexpr (is_fold(MODE) || MODE == kModeFoldLazy) {
        const Dwordx4 pair = __builtin_amdgcn_raw_buffer_load_b128(tw.pair_resource, lane_index << 4, fixed_index << 4, 0);
        t.w = pack64(pair.x, pair.y);
        t.second = pack64(pair.z, pair.w);
        t.factors = 0;
    } else {
        const U64x2 pair = load_twiddle(tw.pairs + fixed_index + lane_index);
        t.w = pair.x;
        t.second = pair.y;
        t.factors = 0;
    }
    return t;
}

// How many twiddles of a pass are in flight ahead of the butterflies (forward_pass / inverse_pass): every one held costs
// its registers under the 64-register cap of the 8-words-per-lane kernels -- 6 for the limb-wise products, where two ahead lose
// 2-12 % to scratch (profiles/r03q_ntt_twiddles_ahead.txt); 4 for the shift-folded ones, which keep three: inverse -2 % / -2 % and
// forward 0 / -0.7 % at N = 8192 for the second and the third (profiles/r05ad_fold_lazy_butterflies_ab.txt,
// r05ap_twiddles_in_flight_ab.txt).
template <int MODE>
constexpr int kTwiddlesAhead = MODE == kModeFoldLazy ? 3 : 1;
// ... and of the row groups of three and four (behz_kernels.hip: one workgroup per CU at 128 registers per lane -- there are
// registers for deeper requests, and with 4 wavefronts per SIMD less else to hide a gather's latency)
constexpr int kWideGroupTwiddlesAhead = 1;
template <int MODE, int ROWS>
constexpr int kGroupTwiddlesAhead = ROWS >= 3 ? kWideGroupTwiddlesAhead : kTwiddlesAhead<MODE>;

template <int MODE>
struct Lazy {
    static constexpr bool kSplit = is_split(MODE);
    // products < p << this (split: [0, 8p); with shifted factors [0, 12p))
    // (the fold modes keep their own fixed ranges -- products below 6p, forward words below 14p, inverse words below 6p --
    // in forward_butterfly / inverse_butterfly; the two constants below are not used for them)
    static constexpr int kProductLog = MODE == kModeExact ? 1 : MODE == kModeApprox ? 2 : 3;
    // cap on stage inputs of the inverse transform, as a shift of p (split: sums of two stay below 2^9 p < 2^64)
    static constexpr int kInverseCapLog = MODE == kModeExact ? 1 : MODE == kModeApprox ? 2 : is_fold(MODE) ? 3 : 8;
    // `reduction` = 2^64 - p (exact / approx) or 2^64 - 2p (split)
    template <bool UNIFORM = false>
    __device__ static __forceinline__ uint64_t mul(uint64_t x, const TwiddleWords& w, uint64_t reduction) {
        if constexpr (MODE == kModeFoldLazy) {
            return fold_mul<UNIFORM, false>(x, w.w, w.second, fold_constants<false>((0 - reduction) >> 1));
        } else if constexpr (kSplit) {
            return split_mul_add<UNIFORM, false>(0, x, w.w, w.second, w.factors, reduction);
        } else if constexpr (MODE == kModeApprox && UNIFORM) {
            return shoup_lazy4_uniform(x, w.w, w.second, reduction);
        } else if constexpr (MODE == kModeApprox) {
            return shoup_lazy4(x, w.w, w.second, reduction);
        } else {
            return shoup_lazy(x, w.w, w.second, reduction);
        }
    }
    __device__ static __forceinline__ uint64_t reduction_constant(uint64_t p) {
        if constexpr (kSplit) {
            return 0 - 2 * p;
        } else if constexpr (MODE == kModeApprox || is_fold(MODE)) {
            return 0 - p;  // asm multiply: the uniform constant is read from SGPRs (or copied once when needed in VGPRs)
        } else {
            return opaque(0 - p);  // keep in VGPRs: a uniform multiplicand triggers a poor 64-bit expansion
        }
    }
};

// x < 2^10 p, 2^40 <= p < 2^55  ->  x mod p, with the quotient estimated in fp32 from the high words:
// e = (x >> 32) * (2^32 / p) * (1 - 2^-18).  The dropped low word costs < 2^32 / p <= 2^-8, the down-bias
// < 2^10 * 2^-18 and fp32 rounding (four roundings, each 2^-24 relative, all dominated by the bias) keeps
// e <= x / p, so q = floor(e) is floor(x / p) or one less: x - q p lies in [0, 2p) and one conditional subtract
// finishes.  x - q p is the low 64 bits of x + q (2^64 - p): one multiply-add on top of x, one low product.
// x 2^-LOGN mod p without a Shoup product, for the N^-1 factor of the inverse transform's last stage: every NTT modulus
// is 1 mod 2N, so p = P 2^LOGN + 1 and, with m = 2^LOGN - (x mod 2^LOGN) in [1, 2^LOGN], x + m p is a multiple of
// 2^LOGN:  (x + m p) / 2^LOGN = floor(x / 2^LOGN) + 1 + m P  =  x 2^-LOGN (mod p),  below x / 2^LOGN + p + 1.
// m = (~x & (2^LOGN - 1)) + 1, hence the value is floor(x / 2^LOGN) + (~x & mask) P + (P + 1): one 64-bit shift, one bit
// operation, one multiply-add by the low word of P, one 24-bit multiply-add by its high word (P < 2^50, mask < 2^15) and
// one add; x < 2^LOGN p makes the result < 2p and one conditional subtract canonical (8 + 5 issue slots against 28 for
// the Shoup product and its float-estimated reduction).
template <int LOGN>
__device__ __forceinline__ uint64_t divide_by_degree(uint64_t x, uint64_t p) {
    static_assert(LOGN >= 10 && LOGN <= 15, "x < 2^9 p must land below 2p; the mask must fit 24 bits");
    const uint64_t P = (p - 1) >> LOGN;
    const uint32_t low_bits = ~lo32(x) & ((1u << LOGN) - 1u);
    const uint64_t shifted = x >> LOGN;
    uint64_t acc, carry;
    uint32_t high_word;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=&v"(acc), "=&s"(carry) : "v"(low_bits), "s"(lo32(P)), "v"(shifted));
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(high_word) : "v"(low_bits), "s"(hi32(P)), "v"(hi32(acc)));
    uint64_t lazy;
    asm("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(lazy) : "v"(pack64(lo32(acc), high_word)), "s"(P + 1));
    return csub63<true>(lazy, 0 - p);
}

struct LazyReducer {
    uint64_t neg_p;
    float scale;
    // the modulus is wave-uniform: the quotient scale is computed once per wave and parked in an SGPR
    __device__ __forceinline__ explicit LazyReducer(uint64_t modulus)
        : neg_p(0 - modulus),
          scale(__builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(
              int, (4294967296.0f * (1.0f - 1.0f / 262144.0f)) / static_cast<float>(modulus))))) {}
    // [0, 2p)
    __device__ __forceinline__ uint64_t lazy(uint64_t x) const {
        float high;  // asm: hipcc otherwise converts through its generic 64-bit path (7 instructions)
        asm("v_cvt_f32_u32 %0, %1" : "=v"(high) : "v"(hi32(x)));
        const uint32_t q = static_cast<uint32_t>(high * scale);
        // high word: q * hi32(neg_p) taken as signed 24-bit factors (q < 2^10; hi32(2^64 - p) = -(hi32(p - 1) + 1) >= -2^23
        // for p < 2^55) -- one full-rate multiply-add instead of a 32-bit multiply and an add
        uint64_t low, carry;
        uint32_t high_word;
        asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=&v"(low), "=&s"(carry) : "v"(q), "s"(lo32(neg_p)), "v"(x));
        asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(high_word) : "v"(q), "s"(hi32(neg_p)), "v"(hi32(low)));
        return pack64(lo32(low), high_word);
    }
    __device__ __forceinline__ uint64_t operator()(uint64_t x) const { return csub63<true>(lazy(x), neg_p); }
};

// ---- register passes ---------------------------------------------------------------------------------------------
// A pass runs W radix-2 stages on the 2^LOGE words a lane holds of each of ROWS residue rows of ONE modulus: every
// twiddle is fetched once and serves the butterflies of all rows (two rows per workgroup halve the gathers, scalar
// loads and address arithmetic per row).  The 2^W - 1 twiddles of a pass are walked in butterfly order and requested
// one step ahead -- the gather of twiddle k + 1 is in flight while the butterflies of twiddle k run; the first one is
// requested by the caller BEFORE the LDS exchange that precedes the pass (pass_first_twiddle).
// Twiddles per lane: forward stage j of a pass (stride 2^(W-1-j) registers) has 2^(LOGE-W+j) of them, inverse stage j
// (stride 2^j) has 2^(LOGE-1-j); they are walked stage by stage.
template <int LOGE, int W, bool INVERSE>
constexpr int pass_twiddle_count() {
    int count = 0;
    for (int j = 0; j < W; ++j) count += INVERSE ? (1 << (LOGE - 1 - j)) : (1 << (LOGE - W + j));
    return count;
}
// twiddles of the first `stages` stages of the walk
template <int LOGE, int W, bool INVERSE>
constexpr int pass_twiddle_prefix(int stages) {
    int count = 0;
    for (int j = 0; j < stages; ++j) count += INVERSE ? (1 << (LOGE - 1 - j)) : (1 << (LOGE - W + j));
    return count;
}
template <int LOGE, int W, bool INVERSE>
struct PassWalk {
    static constexpr int kCount = pass_twiddle_count<LOGE, W, INVERSE>();
    struct Table {
        int stage[kCount > 0 ? kCount : 1];
        int index[kCount > 0 ? kCount : 1];
    };
    static constexpr Table make() {
        Table t{};
        int k = 0;
        for (int j = 0; j < W; ++j) {
            const int here = INVERSE ? (1 << (LOGE - 1 - j)) : (1 << (LOGE - W + j));
            for (int i = 0; i < here; ++i, ++k) {
                t.stage[k] = j;
                t.index[k] = i;
            }
        }
        return t;
    }
    static constexpr Table kTable = make();
};
template <int LOGE, int W, bool INVERSE>
constexpr int pass_stage_of(int k) { return PassWalk<LOGE, W, INVERSE>::kTable.stage[k]; }
template <int LOGE, int W, bool INVERSE>
constexpr int pass_index_in_stage(int k) { return PassWalk<LOGE, W, INVERSE>::kTable.index[k]; }

template <int LOGN, int LOGE, int LO, int W, bool UNIFORM_TWIDDLES>
constexpr bool stage_is_uniform(int b) {
    // When the six in-wave lane bits all sit at or below b the twiddle index is the same for the whole wave: read it
    // through the scalar cache into SGPRs.
    return UNIFORM_TWIDDLES || (element_index<LOGN, LOGE, LO, W>(0, 63u) >> (b + 1)) == 0;
}

// k-th twiddle of a forward pass (stages from the top bit of the pass down)
template <int LOGN, int LOGE, int LO, int W, int MODE, bool UNIFORM_TWIDDLES>
__device__ __forceinline__ TwiddleWords forward_twiddle(const Twiddles<MODE>& tw, uint32_t lane_elements, int k) {
    const int j = pass_stage_of<LOGE, W, false>(k), idx = pass_index_in_stage<LOGE, W, false>(k);
    const int b = LO + W - 1 - j;      // element bit paired by this stage
    const int s = LOGN - 1 - b;        // global stage number; m = 2^s groups
    const int stride = 1 << (b - LO);  // register distance of a pair
    // twiddle index = 2^s + (element index >> (b+1)); lane and register parts are disjoint bit fields
    uint32_t fixed = (1u << s) + (register_part<LOGN, LOGE, LO, W>(idx * 2 * stride) >> (b + 1));
    uint32_t lane_twiddle = lane_elements >> (b + 1);
    const int g = LO + W - 1 - b;  // register bits of the pass above the stage's bit: 2^g twiddles per lane
    if (tw.lanes && g > 0) {
        // (its low g bits say which of the lane's twiddles; a partial pass's extra register bits lie above them)
        const uint32_t within = register_part<LOGN, LOGE, LO, W>(idx * 2 * stride) >> (b + 1);
        fixed = (1u << s) + (within & ((1u << g) - 1u)) * ((1u << s) >> g) + (within >> g);
        lane_twiddle = lane_elements >> (b + 1 + g);
    }
    if (stage_is_uniform<LOGN, LOGE, LO, W, UNIFORM_TWIDDLES>(b))
        return fetch_twiddle<MODE, true>(tw, __builtin_amdgcn_readfirstlane(lane_twiddle), fixed);
    return fetch_twiddle<MODE, false>(tw, lane_twiddle, fixed);
}

// One forward butterfly (x, y) -> (x + w y, x - w y + B) in the lazy ranges of MODE (forward_pass).  `fold`: exact /
// approx bring x under half_bound first (not needed on canonical input); the split modes never fold.
template <int MODE>
__device__ __forceinline__ void forward_butterfly(uint64_t& first, uint64_t& second, const TwiddleWords& w, bool uniform,
                                                  uint64_t neg_p, uint64_t half_bound, bool fold,
                                                  const FoldConstants& fc = FoldConstants{}) {
    uint64_t x = first;
    const uint64_t y = second;
    if constexpr (is_fold(MODE)) {
        // words below 14p: x under 8p first, the product below 6p; x + r < 14p and x + 6p - r in (0, 14p); p = 0 - neg_p
        constexpr bool PLUS = MODE == kModeFoldPlus;
        const uint64_t p = 0 - neg_p;
        if (fold) x = csub_uniform(x, 8 * p);
        const uint64_t r = uniform ? fold_mul<true, PLUS>(y, w.w, w.second, fc) : fold_mul<false, PLUS>(y, w.w, w.second, fc);
        first = x + r;
        second = x + 6 * p - r;
        return;
    }
    if constexpr (MODE == kModeFoldLazy) {
        // the product folded at 2^(b+2) (below 6p for any 64-bit y), no quotient, no factors; nothing ever brought back
        const uint64_t r = uniform ? fold_mul<true, false>(y, w.w, w.second, fc) : fold_mul<false, false>(y, w.w, w.second, fc);
        first = x + r;
        second = x + half_bound - r;
        return;
    }
    if (!is_split(MODE) && fold) x = csub_uniform(x, half_bound);
    if constexpr (is_split(MODE) || MODE == kModeApprox) {
        // x + w y leaves the multiplier's addend port; x - w y + B = (2x + B) - (x + w y)
        uint64_t sum;
        if constexpr (is_split(MODE)) {
            sum = uniform ? split_mul_add<true, true>(x, y, w.w, w.second, w.factors, neg_p)
                          : split_mul_add<false, true>(x, y, w.w, w.second, w.factors, neg_p);
        } else {
            sum = uniform ? shoup_lazy4_fma<true>(x, y, w.w, w.second, neg_p)
                          : shoup_lazy4_fma<false>(x, y, w.w, w.second, neg_p);
        }
        first = sum;
        second = ((x << 1) + half_bound) - sum;
    } else {
        const uint64_t t = uniform ? Lazy<MODE>::template mul<true>(y, w, neg_p) : Lazy<MODE>::template mul<false>(y, w, neg_p);
        first = x + t;
        second = x + half_bound - t;
    }
}

// STAGES: only the first STAGES stages of the layout's W (the one-stage pass of a schedule whose partial pass sits on
// the TOP bits runs in the layout of a full top pass); `first` is then still twiddle 0.
template <int LOGN, int LOGE, int LO, int W, int MODE, bool UNIFORM_TWIDDLES, int ROWS, int STAGES = W>
__device__ __forceinline__ void forward_pass(uint64_t (&v)[ROWS][1 << LOGE], uint32_t tid, const Twiddles<MODE>& tw,
                                             uint64_t p, bool first_stage_canonical, TwiddleWords first) {
    constexpr int COUNT = pass_twiddle_prefix<LOGE, W, false>(STAGES);
    const uint64_t neg_p = Lazy<MODE>::reduction_constant(p);
    const uint64_t half_bound = p << Lazy<MODE>::kProductLog;  // Harvey: fold x into [0, half_bound) first
    // split modes never fold: a word gains at most p << kProductLog per stage; the final reducer takes x < 2^10 p and
    // p < 2^55 keeps 2^9 p inside 64 bits
    static_assert(!is_split(MODE) || 1 + (LOGN << Lazy<MODE>::kProductLog) <= 511, "split mode: growth must stay below 2^9 p");
    const uint32_t lane_elements = lane_part<LOGN, LOGE, LO, W>(tid);
    const FoldConstants fc = mode_fold_constants<MODE>(p);
    constexpr int AHEAD = kGroupTwiddlesAhead<MODE, ROWS>;
    TwiddleWords pending[AHEAD];
    pending[0] = first;
#pragma unroll
    for (int a = 1; a < AHEAD; ++a)
        if (a < COUNT) pending[a] = forward_twiddle<LOGN, LOGE, LO, W, MODE, UNIFORM_TWIDDLES>(tw, lane_elements, a);
#pragma unroll
    for (int k = 0; k < COUNT; ++k) {
        const int j = pass_stage_of<LOGE, W, false>(k), idx = pass_index_in_stage<LOGE, W, false>(k);
        const int b = LO + W - 1 - j;
        const int stride = 1 << (b - LO);
        const int base = idx * 2 * stride;
        const bool uniform = stage_is_uniform<LOGN, LOGE, LO, W, UNIFORM_TWIDDLES>(b);
        const TwiddleWords w = pending[0];
#pragma unroll
        for (int a = 1; a < AHEAD; ++a) pending[a - 1] = pending[a];
        if (k + AHEAD < COUNT)
            pending[AHEAD - 1] = forward_twiddle<LOGN, LOGE, LO, W, MODE, UNIFORM_TWIDDLES>(tw, lane_elements, k + AHEAD);
        if (k + 1 < COUNT) __builtin_amdgcn_sched_barrier(0);  // the request stays ahead of the butterflies below
#pragma unroll
        for (int row = 0; row < ROWS; ++row) {
#pragma unroll
            for (int o = 0; o < stride; ++o)
                forward_butterfly<MODE>(v[row][base + o], v[row][base + o + stride], w, uniform, neg_p, half_bound,
                                        !(first_stage_canonical && j == 0), fc);
        }
        if (k + 1 < COUNT) __builtin_amdgcn_sched_barrier(0);
    }
}
// the first twiddle of a forward pass: request it before the LDS exchange that feeds the pass
template <int LOGN, int LOGE, int LO, int W, int MODE, bool UNIFORM_TWIDDLES>
__device__ __forceinline__ TwiddleWords forward_first_twiddle(const Twiddles<MODE>& tw, uint32_t tid) {
    return forward_twiddle<LOGN, LOGE, LO, W, MODE, UNIFORM_TWIDDLES>(tw, lane_part<LOGN, LOGE, LO, W>(tid), 0);
}

// Inverse transform, split mode: bound (as a shift of p) on the words ENTERING the stage on element bit b.  Canonical
// input; a product is < 8p; a sum doubles the bound; when a stage's sums would pass the cap they are reduced under 2p.
template <int MODE>
constexpr int inverse_in_shift(int b) {
    constexpr int H = Lazy<MODE>::kInverseCapLog, K = Lazy<MODE>::kProductLog;
    if constexpr (!is_split(MODE)) {
        return b == 0 ? 0 : (b + K - 1 < H ? b + K - 1 : H);
    } else {
        int shift = 0;
        for (int stage = 0; stage < b; ++stage) {
            const int sums = (shift + 1 > H) ? 1 : shift + 1;
            shift = sums > K ? sums : K;
        }
        return shift;
    }
}

// k-th twiddle of an inverse pass (stages from the low bit of the pass up); the very last stage of the transform
// (bit LOGN-1) multiplies by the two N^-1 constants instead
template <int LOGN, int LOGE, int LO, int W, int MODE, bool UNIFORM_TWIDDLES>
__device__ __forceinline__ TwiddleWords inverse_twiddle(const Twiddles<MODE>& tw, uint32_t lane_elements, int k) {
    constexpr uint32_t N = 1u << LOGN;
    const int j = pass_stage_of<LOGE, W, true>(k), idx = pass_index_in_stage<LOGE, W, true>(k);
    const int b = LO + j;
    if (b == LOGN - 1) return TwiddleWords{0, 0, 0};
    const int stride = 1 << (b - LO);
    const uint32_t m = N >> (b + 1);
    uint32_t fixed = (N - 2 * m + 1) + (register_part<LOGN, LOGE, LO, W>(idx * 2 * stride) >> (b + 1));
    uint32_t lane_twiddle = lane_elements >> (b + 1);
    const int g = LO + W - 1 - b;  // as forward_twiddle: the stage's block is lane-major where tw.lanes
    if (tw.lanes && g > 0) {
        const uint32_t within = register_part<LOGN, LOGE, LO, W>(idx * 2 * stride) >> (b + 1);
        fixed = (N - 2 * m + 1) + (within & ((1u << g) - 1u)) * (m >> g) + (within >> g);
        lane_twiddle = lane_elements >> (b + 1 + g);
    }
    if (stage_is_uniform<LOGN, LOGE, LO, W, UNIFORM_TWIDDLES>(b))
        return fetch_twiddle<MODE, true>(tw, __builtin_amdgcn_readfirstlane(lane_twiddle), fixed);
    return fetch_twiddle<MODE, false>(tw, lane_twiddle, fixed);
}

// beta + 2^31 p (mod 2^64) with beta = p: what split_mul_signed starts its 2^0 column at
__device__ __forceinline__ uint64_t split_signed_bias(uint64_t p) { return p + (p << 31); }
// a wave-uniform 64-bit value in vector registers, made where it is written (not hoisted, not merged)
__device__ __forceinline__ uint64_t vector_copy(uint64_t uniform_value) {
    uint64_t out;
    asm volatile("v_mov_b64 %0, %1" : "=v"(out) : "s"(uniform_value));
    return out;
}

// ---- inverse pass over element bits [LO, LO+W): stages run from the low bit up; the very last stage of the
// transform (bit LOGN-1) folds in N^-1 and N^-1 psi^(-N/2) and produces canonical words --------------------------
// SCALED: mod.inv_degree carries a factor besides N^-1 (DeviceModulus::has_ntt == kNttScaledInverseDegree): the last
// stage's sums take the Shoup product; otherwise they are divided by N exactly (divide_by_degree).
// One inverse butterfly below the last stage: (x, y) -> (x + y, (x - y + bound) w) with inputs below `bound` = p <<
// in_shift; `fold` brings the sum back under the cap (inverse_in_shift).  Split mode: (x + y, (x - y) w) with the
// difference multiplied as a signed word in kModeSplitSigned (`bias`: split_signed_bias(p), in vector registers beside a
// wave-uniform twiddle).
template <int MODE>
__device__ __forceinline__ void inverse_butterfly(uint64_t& first, uint64_t& second, const TwiddleWords& w, bool uniform,
                                                  uint64_t p, uint64_t neg_p, uint64_t bound, bool fold,
                                                  const FoldConstants& fc = FoldConstants{}, uint64_t bias = 0) {
    const uint64_t x = first, y = second;
    if constexpr (is_fold(MODE)) {
        // words below 6p (`fold`: not on canonical input): the sum back under 6p, x + 6p - y in (0, 12p), the product below 6p
        constexpr bool PLUS = MODE == kModeFoldPlus;
        const uint64_t sum = x + y;
        first = fold ? csub_uniform(sum, 6 * p) : sum;
        const uint64_t diff = x + 6 * p - y;
        second = uniform ? fold_mul<true, PLUS>(diff, w.w, w.second, fc) : fold_mul<false, PLUS>(diff, w.w, w.second, fc);
        return;
    }
    uint64_t sum = x + y;
    if constexpr (MODE == kModeSplitSigned) {
        // x - y goes to the product as a signed word (device_math.hpp split_mul_signed: the bound that would keep it
        // non-negative is one 64-bit addition per butterfly; inputs are below 2^63, so the difference cannot wrap); the
        // product comes back in (0, 6p).  The product first: its operands (x - y, the twiddle) die in it, and the sum can
        // then take x's registers.
        second = uniform ? split_mul_signed<true>(x - y, w.w, w.second, w.factors, neg_p, bias)
                         : split_mul_signed<false>(x - y, w.w, w.second, w.factors, neg_p, bias);
        if (fold) sum = LazyReducer(p).lazy(sum);
        first = sum;
        return;
    }
    const uint64_t diff = x + bound - y;
    if (fold) {
        if constexpr (is_split(MODE)) {
            sum = LazyReducer(p).lazy(sum);
        } else {
            sum = csub_uniform(sum, bound);
        }
    }
    first = sum;
    second = uniform ? Lazy<MODE>::template mul<true>(diff, w, neg_p) : Lazy<MODE>::template mul<false>(diff, w, neg_p);
}

// PRIOR: stages of the same transform that ran before the ones on bits [0, LOGN) (the cross stages of an interleaved
// row, ntt_kernels.hip): they only move the lazy bounds.  LOGD: log2 of the transform's degree (N^-1 = 2^-LOGD).
// FIRST_STAGE: the pass skips the first FIRST_STAGE stages of the layout's W (the top pass of a schedule whose partial
// pass sits on the top bits: the lower bits of its layout were paired by the pass before); `first` is the first twiddle
// of the stages it does run (inverse_first_twiddle with the same FIRST_STAGE).
// AHEAD / `head`: the pass keeps AHEAD twiddles in flight ahead of its butterflies; the caller hands it the first AHEAD of
// them (requested wherever it suits the caller: the transform's first pass asks for them before the rows are unpacked).
template <int LOGN, int LOGE, int LO, int W, int MODE, bool UNIFORM_TWIDDLES, int ROWS, bool SCALED = false, int PRIOR = 0,
          int LOGD = LOGN, int FIRST_STAGE = 0, int AHEAD = kTwiddlesAhead<MODE>>
__device__ __forceinline__ void inverse_pass(uint64_t (&v)[ROWS][1 << LOGE], uint32_t tid, const Twiddles<MODE>& tw,
                                             const DeviceModulus& mod, bool first_stage_canonical,
                                             const TwiddleWords (&head)[AHEAD]) {
    constexpr int COUNT = pass_twiddle_count<LOGE, W, true>(), BEGIN = pass_twiddle_prefix<LOGE, W, true>(FIRST_STAGE);
    const uint64_t p = mod.p;
    const uint64_t neg_p = Lazy<MODE>::reduction_constant(p);
    // Words entering the stage on element bit b live in [0, p << in_shift(b)): canonical input for b = 0; after
    // that sums double the bound and products are < p << K (K = Lazy::kProductLog); exact / approx: once the bound
    // reaches the cap H every sum is folded back under p << H; split: see inverse_in_shift.
    constexpr int H = Lazy<MODE>::kInverseCapLog;
    const uint32_t lane_elements = lane_part<LOGN, LOGE, LO, W>(tid);
    const FoldConstants fc = mode_fold_constants<MODE>(p);
    TwiddleWords pending[AHEAD];
#pragma unroll
    for (int a = 0; a < AHEAD; ++a) pending[a] = head[a];
#pragma unroll
    for (int k = BEGIN; k < COUNT; ++k) {
        const int j = pass_stage_of<LOGE, W, true>(k), idx = pass_index_in_stage<LOGE, W, true>(k);
        const int b = LO + j;
        const int stride = 1 << (b - LO);
        const int base = idx * 2 * stride;
        const bool last_stage = (b == LOGN - 1);
        const bool canonical_in = first_stage_canonical && j == 0;
        const int in_shift = canonical_in ? 0 : inverse_in_shift<MODE>(b + PRIOR);
        // (fold modes: every stage but one on canonical input folds its sums, and the last stage's difference is x + 6p - y)
        const uint64_t bound = is_fold(MODE) ? (canonical_in ? p : 6 * p) : p << in_shift;
        const bool fold = is_fold(MODE) ? !canonical_in : in_shift + 1 > H;  // x + y may reach 2 * bound: allowed while that stays under the cap
        const bool uniform = stage_is_uniform<LOGN, LOGE, LO, W, UNIFORM_TWIDDLES>(b);
        // (without the request one twiddle ahead the kernel fits its 64 registers with nothing spilled -- and runs 4 %
        // slower; without it but with the per-transpose LDS rules, still 1 % slower: profiles/r02ze_lds_schemes.txt)
        const TwiddleWords w = pending[0];
#pragma unroll
        for (int a = 1; a < AHEAD; ++a) pending[a - 1] = pending[a];
        if (k + AHEAD < COUNT)
            pending[AHEAD - 1] = inverse_twiddle<LOGN, LOGE, LO, W, MODE, UNIFORM_TWIDDLES>(tw, lane_elements, k + AHEAD);
        if (k + 1 < COUNT) __builtin_amdgcn_sched_barrier(0);
        // split mode: the signed product's bias (device_math.hpp split_mul_signed) -- a scalar operand beside gathered
        // twiddles; beside wave-uniform ones (their words take the instruction's one scalar operand) a vector copy made
        // here, once per twiddle, so that it is not carried through the gathered stages
        uint64_t bias = split_signed_bias(p);
        if constexpr (MODE == kModeSplitSigned) {
            if (uniform && !last_stage) bias = vector_copy(bias);
        }
#pragma unroll
        for (int row = 0; row < ROWS; ++row) {
#pragma unroll
            for (int o = 0; o < stride; ++o) {
                if (!last_stage) {
                    inverse_butterfly<MODE>(v[row][base + o], v[row][base + o + stride], w, uniform, p, neg_p, bound, fold, fc,
                                            bias);
                    continue;
                }
                const uint64_t x = v[row][base + o];
                const uint64_t y = v[row][base + o + stride];
                const uint64_t sum = x + y;
                const uint64_t diff = x + bound - y;
                {
                    if constexpr (is_split(MODE)) {
                        const LazyReducer reduce(p);
                        if constexpr (SCALED)
                            v[row][base + o] = reduce(split_mul_add<true, false>(0, sum, mod.inv_degree, mod.inv_degree_split,
                                                                                 mod.inv_degree_factors, neg_p));
                        else
                            v[row][base + o] = divide_by_degree<LOGD>(sum, p);  // sum < 2^9 p (inverse_in_shift)
                        v[row][base + o + stride] = reduce(split_mul_add<true, false>(
                            0, diff, mod.inv_degree_root, mod.inv_degree_root_split, mod.inv_degree_root_factors, neg_p));
                    } else {
                        v[row][base + o] = SCALED ? shoup_mul(sum, mod.inv_degree, mod.inv_degree_shoup, p)
                                                  : divide_by_degree<LOGD>(sum, p);  // sum < 8p
                        v[row][base + o + stride] = shoup_mul(diff, mod.inv_degree_root, mod.inv_degree_root_shoup, p);
                    }
                }
            }
        }
        if (k + 1 < COUNT) __builtin_amdgcn_sched_barrier(0);
    }
}
template <int LOGN, int LOGE, int LO, int W, int MODE, bool UNIFORM_TWIDDLES, int FIRST_STAGE = 0>
__device__ __forceinline__ TwiddleWords inverse_first_twiddle(const Twiddles<MODE>& tw, uint32_t tid) {
    return inverse_twiddle<LOGN, LOGE, LO, W, MODE, UNIFORM_TWIDDLES>(tw, lane_part<LOGN, LOGE, LO, W>(tid),
                                                                      pass_twiddle_prefix<LOGE, W, true>(FIRST_STAGE));
}
// the first AHEAD twiddles of an inverse pass
template <int LOGN, int LOGE, int LO, int W, int MODE, bool UNIFORM_TWIDDLES, int AHEAD, int FIRST_STAGE = 0>
__device__ __forceinline__ void inverse_first_twiddles(TwiddleWords (&head)[AHEAD], const Twiddles<MODE>& tw, uint32_t tid) {
    constexpr int COUNT = pass_twiddle_count<LOGE, W, true>(), BEGIN = pass_twiddle_prefix<LOGE, W, true>(FIRST_STAGE);
    const uint32_t lane_elements = lane_part<LOGN, LOGE, LO, W>(tid);
#pragma unroll
    for (int a = 0; a < AHEAD; ++a)
        head[a] = BEGIN + a < COUNT ? inverse_twiddle<LOGN, LOGE, LO, W, MODE, UNIFORM_TWIDDLES>(tw, lane_elements, BEGIN + a)
                                    : TwiddleWords{0, 0, 0};
}

template <int LOGN, int LOGE, int LO, int W, int SCHEME = 0>
__device__ __forceinline__ void lds_store(const uint64_t (&v)[1 << LOGE], uint32_t tid, uint64_t* lds) {
    uint64_t* const base = lds + lds_slot_scheme<SCHEME>(lane_part<LOGN, LOGE, LO, W>(tid));
#pragma unroll
    for (int r = 0; r < (1 << LOGE); ++r) base[lds_slot_scheme<SCHEME>(register_part<LOGN, LOGE, LO, W>(r))] = v[r];
}
template <int LOGN, int LOGE, int LO, int W, int SCHEME = 0>
__device__ __forceinline__ void lds_load(uint64_t (&v)[1 << LOGE], uint32_t tid, const uint64_t* lds) {
    // (tried: volatile reads to keep single ds_read_b64 instead of merged ds_read2_b64 -- 1.5 % slower, r01d notes)
    const uint64_t* const base = lds + lds_slot_scheme<SCHEME>(lane_part<LOGN, LOGE, LO, W>(tid));
#pragma unroll
    for (int r = 0; r < (1 << LOGE); ++r) v[r] = base[lds_slot_scheme<SCHEME>(register_part<LOGN, LOGE, LO, W>(r))];
}

// Global <-> registers through the row's buffer descriptor: one 32-bit lane offset per pass, the register part is a
// scalar offset.  For a pass on the low bits each lane owns runs of 2^W contiguous words: move them 16 B at a time.
// For the top pass consecutive lanes own consecutive words (8 B each, 512 B per wave instruction).
// Row data is touched once per launch: loaded and stored non-temporally it does not displace the twiddle tables from
// the vector L1 / L2 that every workgroup gathers from -- at N = 8192: forward 0.566 -> 0.525 ms, inverse 0.622 ->
// 0.586 ms per launch (profiles/r02e_ntt_ab_nt_policy.txt); N = 16384: 0.399 -> 0.376 ms.  At N = 4096 (half the
// table, 512-lane workgroups, four of them per CU) it is the other way round: 0.326 -> 0.391 ms, so those rows keep the
// default policy (profiles/r02j_ntt_policy_by_degree.txt).  aux bit 1 = nt on gfx940+.
template <int LOGN>
constexpr int row_policy() {
    return LOGN >= 13 ? 2 : 0;
}
// loads may take another policy than stores (production: the same; profiles/r02k_row_load_policy.txt)
template <int LOGN>
constexpr int row_load_policy() {
    return row_policy<LOGN>();
}
// POLICY: row_policy<LOGN>() for rows nobody else reads; 0 (cached) for source rows that several workgroups of a replica set
// read (ntt_kernels.hip locate_replica: the others are meant to hit in L2)
template <int LOGN, int LOGE, int LO, int W, int POLICY = row_load_policy<LOGN>()>
__device__ __forceinline__ void global_load(uint64_t (&v)[1 << LOGE], uint32_t tid, BufferResource row) {
    const uint32_t lane_bytes = lane_part<LOGN, LOGE, LO, W>(tid) << 3;
    if constexpr (LO == 0 && W >= 1) {
#pragma unroll
        for (int r = 0; r < (1 << LOGE); r += 2) {
            const Dwordx4 pair =
                __builtin_amdgcn_raw_buffer_load_b128(row, lane_bytes, register_part<LOGN, LOGE, LO, W>(r) << 3, POLICY);
            v[r] = pack64(pair.x, pair.y);
            v[r + 1] = pack64(pair.z, pair.w);
        }
    } else {
#pragma unroll
        for (int r = 0; r < (1 << LOGE); ++r) {
            const Dwordx2 word =
                __builtin_amdgcn_raw_buffer_load_b64(row, lane_bytes, register_part<LOGN, LOGE, LO, W>(r) << 3, POLICY);
            v[r] = pack64(word.x, word.y);
        }
    }
}
template <int LOGN, int LOGE, int LO, int W>
__device__ __forceinline__ void global_store(const uint64_t (&v)[1 << LOGE], uint32_t tid, BufferResource row) {
    const uint32_t lane_bytes = lane_part<LOGN, LOGE, LO, W>(tid) << 3;
    if constexpr (LO == 0 && W >= 1) {
#pragma unroll
        for (int r = 0; r < (1 << LOGE); r += 2) {
            const Dwordx4 pair = {lo32(v[r]), hi32(v[r]), lo32(v[r + 1]), hi32(v[r + 1])};
            __builtin_amdgcn_raw_buffer_store_b128(pair, row, lane_bytes, register_part<LOGN, LOGE, LO, W>(r) << 3, row_policy<LOGN>());
        }
    } else {
#pragma unroll
        for (int r = 0; r < (1 << LOGE); ++r) {
            const Dwordx2 word = {lo32(v[r]), hi32(v[r])};
            __builtin_amdgcn_raw_buffer_store_b64(word, row, lane_bytes, register_part<LOGN, LOGE, LO, W>(r) << 3, row_policy<LOGN>());
        }
    }
}

// Store of the last forward pass (bits [0, W), W >= 2) in full cache lines.  A lane ends the transform with runs of 2^W
// contiguous words (32 or 64 B): stored as they lie, every 16-byte store instruction puts a quarter or a half of 64
// different lines (N = 4096: 0.331 ms per launch against 0.289 ms with whole lines, N = 16384: 0.358 against 0.326,
// profiles/r02zd_store_probe.txt).  The wave's block of 64 2^W words (one per run index) goes through the wave's own
// slice of the LDS tile instead -- the transpose into this pass never left the wave, so the slice is free -- and comes
// back as 16-byte chunks in lane order: instruction j stores chunks [64 j, 64 j + 64) of the block, 1 KiB of whole
// lines.  Chunk j of lane l sits at chunk position l C + (j ^ s(l)), C = 2^(W-1) chunks per lane: the XOR spreads both
// the writes (lane stride 16 C bytes) and the reads over all banks.
template <int LOGN, int LOGE, int LO_PREVIOUS, int W>
constexpr bool kStagedStore = W >= 2 && W <= 3 && kWaveOwnsTopBits<LOGN, LOGE, LO_PREVIOUS> && kWaveOwnsTopBits<LOGN, LOGE, 0>;

template <int LOGN, int LOGE, int W>
__device__ __forceinline__ void global_store_staged(uint64_t (&v)[1 << LOGE], uint32_t tid, BufferResource row, uint64_t* lds) {
    constexpr int C = 1 << (W - 1);          // 16-byte chunks per lane and run
    constexpr int RUNS = 1 << (LOGE - W);    // runs per lane (the extra register bits of a partial pass)
    const uint32_t lane = tid & 63u;
    const uint32_t swizzle = (lane >> (4 - (W - 1))) & (C - 1);
#pragma unroll
    for (int run = 0; run < RUNS; ++run) {
        // first element of the wave's block for this run: the lane part of lane 0 of this wave, plus the run's register part
        const uint32_t first = lane_part<LOGN, LOGE, 0, W>(tid & ~63u) | register_part<LOGN, LOGE, 0, W>(run << W);
        char* const block = reinterpret_cast<char*>(lds + lds_slot(first));
#pragma unroll
        for (int j = 0; j < C; ++j)
            *reinterpret_cast<U64x2*>(block + ((lane * C + (j ^ swizzle)) << 4)) = U64x2{v[(run << W) + 2 * j], v[(run << W) + 2 * j + 1]};
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const uint32_t chunk = j * 64 + lane, owner = chunk / C, slot = chunk % C;
            const uint32_t owner_swizzle = (owner >> (4 - (W - 1))) & (C - 1);
            const U64x2 pair = *reinterpret_cast<const U64x2*>(block + ((owner * C + (slot ^ owner_swizzle)) << 4));
            const Dwordx4 words = {lo32(pair.x), hi32(pair.x), lo32(pair.y), hi32(pair.y)};
            __builtin_amdgcn_raw_buffer_store_b128(words, row, (first << 3) + (lane << 4), j << 10, row_policy<LOGN>());
        }
    }
}

// The mirror image for the first inverse pass: the wave's block is loaded as 16-byte chunks in lane order (whole
// lines per instruction), parked in the wave's own slice of the tile (nothing else has touched the tile yet, and every
// wave stores only into its own slice) and picked up as runs of 2^W contiguous words per lane.
template <int LOGN, int LOGE, int W>
constexpr bool kStagedLoad = W >= 2 && W <= 3 && kWaveOwnsTopBits<LOGN, LOGE, 0>;

// in two halves, so that a workgroup that walks over rows can have the next row's chunks in flight while it transforms
// the current one: the request (registers only) and the pick-up through the wave's slice of the tile
template <int LOGN, int LOGE, int W>
struct StagedChunks {
    Dwordx4 words[1 << (LOGE - W)][1 << (W - 1)];
};
template <int LOGN, int LOGE, int W, int POLICY = row_load_policy<LOGN>()>
__device__ __forceinline__ void global_load_staged_request(StagedChunks<LOGN, LOGE, W>& chunks, uint32_t tid, BufferResource row) {
    constexpr int C = 1 << (W - 1);
    constexpr int RUNS = 1 << (LOGE - W);
    const uint32_t lane = tid & 63u;
#pragma unroll
    for (int run = 0; run < RUNS; ++run) {
        const uint32_t first = lane_part<LOGN, LOGE, 0, W>(tid & ~63u) | register_part<LOGN, LOGE, 0, W>(run << W);
#pragma unroll
        for (int j = 0; j < C; ++j)
            chunks.words[run][j] = __builtin_amdgcn_raw_buffer_load_b128(row, (first << 3) + (lane << 4), j << 10, POLICY);
    }
}
template <int LOGN, int LOGE, int W>
__device__ __forceinline__ void global_load_staged_unpack(uint64_t (&v)[1 << LOGE], const StagedChunks<LOGN, LOGE, W>& chunks,
                                                          uint32_t tid, uint64_t* lds) {
    constexpr int C = 1 << (W - 1);
    constexpr int RUNS = 1 << (LOGE - W);
    const uint32_t lane = tid & 63u;
    const uint32_t swizzle = (lane >> (4 - (W - 1))) & (C - 1);
#pragma unroll
    for (int run = 0; run < RUNS; ++run) {
        const uint32_t first = lane_part<LOGN, LOGE, 0, W>(tid & ~63u) | register_part<LOGN, LOGE, 0, W>(run << W);
        char* const block = reinterpret_cast<char*>(lds + lds_slot(first));
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const uint32_t chunk = j * 64 + lane, owner = chunk / C, slot = chunk % C;
            const uint32_t owner_swizzle = (owner >> (4 - (W - 1))) & (C - 1);
            const Dwordx4 w = chunks.words[run][j];
            *reinterpret_cast<U64x2*>(block + ((owner * C + (slot ^ owner_swizzle)) << 4)) =
                U64x2{pack64(w.x, w.y), pack64(w.z, w.w)};
        }
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const U64x2 pair = *reinterpret_cast<const U64x2*>(block + ((lane * C + (j ^ swizzle)) << 4));
            v[(run << W) + 2 * j] = pair.x;
            v[(run << W) + 2 * j + 1] = pair.y;
        }
    }
}
template <int LOGN, int LOGE, int W, int POLICY = row_load_policy<LOGN>()>
__device__ __forceinline__ void global_load_staged(uint64_t (&v)[1 << LOGE], uint32_t tid, BufferResource row, uint64_t* lds) {
    StagedChunks<LOGN, LOGE, W> chunks;
    global_load_staged_request<LOGN, LOGE, W, POLICY>(chunks, tid, row);
    global_load_staged_unpack<LOGN, LOGE, W>(v, chunks, tid, lds);
}

template <int MODE>
__device__ __forceinline__ uint64_t canonicalize(uint64_t x, uint64_t p) {
    static_assert(MODE == kModeExact || MODE == kModeApprox || is_fold(MODE), "split outputs go through LazyReducer");
    if constexpr (is_fold(MODE)) x = csub_uniform(x, 8 * p);  // below 14p
    if constexpr (MODE == kModeApprox || is_fold(MODE)) x = csub_uniform(x, 4 * p);
    x = csub_uniform(x, 2 * p);
    return csub_uniform(x, p);
}

template <int MODE, int ROWS, int N>
__device__ __forceinline__ void canonicalize_all(uint64_t (&v)[ROWS][N], uint64_t p) {
#pragma unroll
    for (int row = 0; row < ROWS; ++row) {
        if constexpr (is_split(MODE)) {
            const LazyReducer reduce(p);
#pragma unroll
            for (int r = 0; r < N; ++r) v[row][r] = reduce(v[row][r]);
        } else {
#pragma unroll
            for (int r = 0; r < N; ++r) v[row][r] = canonicalize<MODE>(v[row][r], p);
        }
    }
}

// Pass schedule: P = ceil(LOGN / LOGE) passes; the partial pass (R = LOGN - (P-1) LOGE bits) sits on the low bits,
// i.e. it is the LAST forward pass and the FIRST inverse pass.
template <int LOGN, int LOGE>
struct Schedule {
    static constexpr int P = (LOGN + LOGE - 1) / LOGE;
    static constexpr int R = LOGN - (P - 1) * LOGE;
};
// TOP = true: the partial pass sits on the TOP bits instead -- the FIRST forward pass and the LAST inverse pass, run in
// the layout of a full top pass -- and the full passes cover bits [0, LOGE), [LOGE, 2 LOGE), ...  For N = 8192 with 8
// words per lane the passes are then bit 12 | 11-9 | 8-6 | 5-3 | 2-0 instead of 12-10 | 9-7 | 6-4 | 3-1 | 0: the twiddles
// of three passes instead of two are wave-uniform (scalar loads), a lane gathers 14 twiddles per transform instead of
// 18, and the one-stage pass -- one butterfly per row and twiddle -- has a single uniform twiddle instead of four
// gathered ones.  LOW: the width of the pass on the low bits (the layout rows are loaded / stored in).
template <int LOGN, int LOGE, bool TOP>
struct PassOrder {
    using S = Schedule<LOGN, LOGE>;
    static constexpr bool kTop = TOP && S::R < LOGE && S::P >= 3;
    static constexpr int LOW = kTop ? LOGE : S::R;
    // bit offset of full pass k (k = 1 .. P-1, counted from the top) and of the layout the rows leave the last one in
    static constexpr int lo(int k) { return kTop ? LOGN - S::R - k * LOGE : LOGN - (k + 1) * LOGE; }
};


}  // namespace ntt
}  // namespace heamd
