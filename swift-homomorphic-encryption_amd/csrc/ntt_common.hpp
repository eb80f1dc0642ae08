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

// Butterfly arithmetic modes (chosen per launch from the moduli the launch covers):
//   kModeExact    any p <= 2^62 - 1 : exact Shoup quotient, products in [0, 2p), values in [0, 4p)
//   kModeApprox   p < 2^61          : 3-multiply quotient estimate, products in [0, 4p), values in [0, 8p)
//   kModeHeadroom 2^40 <= p < 2^55  : shoup_headroom (products in [0, 8p)), and the spare top bits absorb the growth
//                                     instead of a conditional subtract per butterfly: the forward transform never
//                                     folds (a word gains at most 8p per stage: < (1 + 8 log2 N) p <= 113 p < 2^62),
//                                     the inverse folds only once sums could pass 2^7 p (its multiplicand x + B - y
//                                     must stay < 2^63); one float-estimated quotient brings forward outputs back
//                                     to [0, p).
//   kModeHeadroomHalved             : kModeHeadroom reading the context's pre-halved Shoup factors
constexpr int kModeExact = 0, kModeApprox = 1, kModeHeadroom = 2, kModeHeadroomHalved = 3;
constexpr bool is_headroom(int mode) { return mode == kModeHeadroom || mode == kModeHeadroomHalved; }

template <int MODE>
struct Lazy {
    static constexpr bool kHeadroom = is_headroom(MODE);
    static constexpr int kProductLog = MODE == kModeExact ? 1 : MODE == kModeApprox ? 2 : 3;  // products < p << this
    // cap on stage inputs of the inverse transform, as a shift of p
    static constexpr int kInverseCapLog = MODE == kModeExact ? 1 : MODE == kModeApprox ? 2 : 7;
    // twiddle as the butterflies want it: headroom mode multiplies by floor(w 2^63 / p) = wf >> 1
    __device__ static __forceinline__ U64x2 prepare(U64x2 w) {
        if constexpr (MODE == kModeHeadroom) w.y >>= 1;  // kModeHeadroomHalved: the table already holds wf >> 1
        return w;
    }
    // `reduction` = 2^64 - p (exact / approx, in VGPRs) or 2^64 - 2p (headroom, uniform)
    template <bool UNIFORM = false>
    __device__ static __forceinline__ uint64_t mul(uint64_t x, U64x2 w, uint64_t reduction) {
        if constexpr (kHeadroom && UNIFORM) {
            return shoup_headroom_uniform(x, w.x, w.y, reduction);
        } else if constexpr (kHeadroom) {
            return shoup_headroom(x, w.x, w.y, reduction);
        } else if constexpr (MODE == kModeApprox && UNIFORM) {
            return shoup_lazy4_uniform(x, w.x, w.y, reduction);
        } else if constexpr (MODE == kModeApprox) {
            return shoup_lazy4(x, w.x, w.y, reduction);
        } else {
            return shoup_lazy(x, w.x, w.y, reduction);
        }
    }
    __device__ static __forceinline__ uint64_t reduction_constant(uint64_t p) {
        if constexpr (kHeadroom) {
            return 0 - 2 * p;
        } else if constexpr (MODE == kModeApprox) {
            return 0 - p;  // asm multiply: the uniform constant is read from SGPRs (or copied once when needed in VGPRs)
        } else {
            return opaque(0 - p);  // keep in VGPRs: a uniform multiplicand triggers a poor 64-bit expansion
        }
    }
};

// ---- forward pass over element bits [LO, LO+W): stages run from the top bit down --------------------------------
template <int LOGN, int LOGE, int LO, int W, int MODE, bool UNIFORM_TWIDDLES, int ABLATE = 0>
__device__ __forceinline__ void forward_pass(uint64_t (&v)[1 << LOGE], uint32_t tid, const U64x2* __restrict__ tw,
                                             uint64_t p, bool first_stage_canonical) {
    constexpr int E = 1 << LOGE;
    const uint64_t neg_p = Lazy<MODE>::reduction_constant(p);
    const uint64_t half_bound = p << Lazy<MODE>::kProductLog;  // Harvey: fold x into [0, half_bound) first
    static_assert(!is_headroom(MODE) || 1 + 8 * LOGN <= 127, "headroom mode: growth must stay below 2^7 p");
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const int b = LO + W - 1 - j;         // element bit paired by this stage
        const int s = LOGN - 1 - b;           // global stage number; m = 2^s groups
        const int stride = 1 << (b - LO);     // register distance of a pair
        // twiddle index = 2^s + (element index >> (b+1)); lane and register parts are disjoint, so the lane part is
        // one address per stage and the register part an immediate offset.  When the six in-wave lane bits all sit
        // at or below b the index is the same for the whole wave: read it through the scalar cache into SGPRs.
        const bool uniform = UNIFORM_TWIDDLES || (element_index<LOGN, LOGE, LO, W>(0, 63u) >> (b + 1)) == 0;
        uint32_t lane_twiddle = lane_part<LOGN, LOGE, LO, W>(tid) >> (b + 1);
        if (uniform) lane_twiddle = __builtin_amdgcn_readfirstlane(lane_twiddle);
        const U64x2* const tw_stage = tw + (1u << s) + lane_twiddle;
#pragma unroll
        for (int base = 0; base < E; base += 2 * stride) {
            // ABLATE bit 0 (measurement only, wrong results): one wave-uniform twiddle instead of the gather
            const U64x2 w = Lazy<MODE>::prepare(
                load_twiddle((ABLATE & 1) ? tw + (1u << s) : tw_stage + (register_part<LOGN, LOGE, LO, W>(base) >> (b + 1))));
#pragma unroll
            for (int o = 0; o < stride; ++o) {
                uint64_t x = v[base + o];
                const uint64_t y = v[base + o + stride];
                if (!is_headroom(MODE) && !(first_stage_canonical && j == 0) && !(ABLATE & 8)) x = csub_uniform(x, half_bound);
                if constexpr (is_headroom(MODE) || MODE == kModeApprox) {
                    // x + w y leaves the multiplier's addend port; x - w y + B = (2x + B) - (x + w y)
                    uint64_t sum;
                    if constexpr (is_headroom(MODE)) {
                        sum = (uniform && !(ABLATE & 1)) ? shoup_headroom_fma<true>(x, y, w.x, w.y, neg_p)
                                                         : shoup_headroom_fma<false>(x, y, w.x, w.y, neg_p);
                    } else {
                        sum = (uniform && !(ABLATE & 1)) ? shoup_lazy4_fma<true>(x, y, w.x, w.y, neg_p)
                                                         : shoup_lazy4_fma<false>(x, y, w.x, w.y, neg_p);
                    }
                    v[base + o] = sum;
                    v[base + o + stride] = ((x << 1) + half_bound) - sum;
                    continue;
                }
                uint64_t t;
                if (uniform && !(ABLATE & 1)) {
                    t = Lazy<MODE>::template mul<true>(y, w, neg_p);
                } else {
                    t = Lazy<MODE>::template mul<false>(y, w, neg_p);
                }
                v[base + o] = x + t;
                v[base + o + stride] = x + half_bound - t;
            }
        }
    }
}

// ---- inverse pass over element bits [LO, LO+W): stages run from the low bit up; the very last stage of the
// transform (bit LOGN-1) folds in N^-1 and N^-1 psi^(-N/2) and produces canonical words --------------------------
template <int LOGN, int LOGE, int LO, int W, int MODE, bool UNIFORM_TWIDDLES = false>
__device__ __forceinline__ void inverse_pass(uint64_t (&v)[1 << LOGE], uint32_t tid, const U64x2* __restrict__ tw,
                                             const DeviceModulus& mod, bool first_stage_canonical) {
    constexpr int E = 1 << LOGE;
    constexpr uint32_t N = 1u << LOGN;
    const uint64_t p = mod.p;
    const uint64_t neg_p = Lazy<MODE>::reduction_constant(p);
    // Words entering the stage on element bit b live in [0, p << in_shift(b)): canonical input for b = 0; after
    // that sums double the bound and products are < p << K (K = Lazy::kProductLog), so in_shift(b) = min(b + K - 1, H);
    // once it reaches the cap H every sum is folded back under p << H.
    constexpr int H = Lazy<MODE>::kInverseCapLog, K = Lazy<MODE>::kProductLog;
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const int b = LO + j;
        const int stride = 1 << (b - LO);
        const uint32_t m = N >> (b + 1);
        const bool last_stage = (b == LOGN - 1);
        const bool canonical_in = first_stage_canonical && j == 0;
        const int in_shift = canonical_in ? 0 : (b + K - 1 < H ? b + K - 1 : H);
        const uint64_t bound = p << in_shift;
        const bool fold = in_shift + 1 > H;  // x + y may reach 2 * bound: allowed while that stays under the cap
        const bool uniform = UNIFORM_TWIDDLES || (element_index<LOGN, LOGE, LO, W>(0, 63u) >> (b + 1)) == 0;
        uint32_t lane_twiddle = lane_part<LOGN, LOGE, LO, W>(tid) >> (b + 1);
        if (uniform) lane_twiddle = __builtin_amdgcn_readfirstlane(lane_twiddle);
        const U64x2* const tw_stage = tw + (N - 2 * m + 1) + lane_twiddle;
#pragma unroll
        for (int base = 0; base < E; base += 2 * stride) {
            U64x2 w = {0, 0};
            if (!last_stage)
                w = Lazy<MODE>::prepare(load_twiddle(tw_stage + (register_part<LOGN, LOGE, LO, W>(base) >> (b + 1))));
#pragma unroll
            for (int o = 0; o < stride; ++o) {
                const uint64_t x = v[base + o];
                const uint64_t y = v[base + o + stride];
                uint64_t sum = x + y;
                const uint64_t diff = x + bound - y;
                if (last_stage) {
                    v[base + o] = shoup_mul(sum, mod.inv_degree, mod.inv_degree_shoup, p);
                    v[base + o + stride] = shoup_mul(diff, mod.inv_degree_root, mod.inv_degree_root_shoup, p);
                } else {
                    if (fold) sum = csub_uniform(sum, bound);
                    v[base + o] = sum;
                    if (uniform) {
                        v[base + o + stride] = Lazy<MODE>::template mul<true>(diff, w, neg_p);
                    } else {
                        v[base + o + stride] = Lazy<MODE>::template mul<false>(diff, w, neg_p);
                    }
                }
            }
        }
    }
}

template <int LOGN, int LOGE, int LO, int W>
__device__ __forceinline__ void lds_store(const uint64_t (&v)[1 << LOGE], uint32_t tid, uint64_t* lds) {
    uint64_t* const base = lds + lds_slot(lane_part<LOGN, LOGE, LO, W>(tid));
#pragma unroll
    for (int r = 0; r < (1 << LOGE); ++r) base[lds_slot(register_part<LOGN, LOGE, LO, W>(r))] = v[r];
}
template <int LOGN, int LOGE, int LO, int W>
__device__ __forceinline__ void lds_load(uint64_t (&v)[1 << LOGE], uint32_t tid, const uint64_t* lds) {
    // (tried: volatile reads to keep single ds_read_b64 instead of merged ds_read2_b64 -- 1.5 % slower, r01d notes)
    const uint64_t* const base = lds + lds_slot(lane_part<LOGN, LOGE, LO, W>(tid));
#pragma unroll
    for (int r = 0; r < (1 << LOGE); ++r) v[r] = base[lds_slot(register_part<LOGN, LOGE, LO, W>(r))];
}

// Global <-> registers.  For a pass on the low bits each lane owns runs of 2^W contiguous words: move them 16 B at
// a time.  For the top pass consecutive lanes own consecutive words (8 B each, 512 B per wave instruction).
template <int LOGN, int LOGE, int LO, int W>
__device__ __forceinline__ void global_load(uint64_t (&v)[1 << LOGE], uint32_t tid, const uint64_t* __restrict__ x) {
    if constexpr (LO == 0 && W >= 1) {
#pragma unroll
        for (int r = 0; r < (1 << LOGE); r += 2) {
            const U64x2 pair = *reinterpret_cast<const U64x2*>(x + register_part<LOGN, LOGE, LO, W>(r) +
                                                               lane_part<LOGN, LOGE, LO, W>(tid));
            v[r] = pair.x;
            v[r + 1] = pair.y;
        }
    } else {
#pragma unroll
        for (int r = 0; r < (1 << LOGE); ++r)
            v[r] = (x + register_part<LOGN, LOGE, LO, W>(r))[lane_part<LOGN, LOGE, LO, W>(tid)];
    }
}
template <int LOGN, int LOGE, int LO, int W>
__device__ __forceinline__ void global_store(const uint64_t (&v)[1 << LOGE], uint32_t tid, uint64_t* __restrict__ x) {
    if constexpr (LO == 0 && W >= 1) {
#pragma unroll
        for (int r = 0; r < (1 << LOGE); r += 2) {
            U64x2 pair;
            pair.x = v[r];
            pair.y = v[r + 1];
            *reinterpret_cast<U64x2*>(x + register_part<LOGN, LOGE, LO, W>(r) + lane_part<LOGN, LOGE, LO, W>(tid)) = pair;
        }
    } else {
#pragma unroll
        for (int r = 0; r < (1 << LOGE); ++r)
            (x + register_part<LOGN, LOGE, LO, W>(r))[lane_part<LOGN, LOGE, LO, W>(tid)] = v[r];
    }
}

template <int MODE>
__device__ __forceinline__ const U64x2* twiddle_table(const DeviceContext& ctx, bool inverse) {
    if constexpr (MODE == kModeHeadroomHalved) {
        return inverse ? ctx.inverse_twiddles_half : ctx.forward_twiddles_half;
    } else {
        return inverse ? ctx.inverse_twiddles : ctx.forward_twiddles;
    }
}

template <int MODE>
__device__ __forceinline__ uint64_t canonicalize(uint64_t x, uint64_t p) {
    static_assert(MODE == kModeExact || MODE == kModeApprox, "headroom outputs go through HeadroomReducer");
    if constexpr (MODE == kModeApprox) x = csub_uniform(x, 4 * p);
    x = csub_uniform(x, 2 * p);
    return csub_uniform(x, p);
}

// x < 64 p, 2^40 <= p < 2^55  ->  x mod p, with the quotient estimated in fp32 from the high words:
// e = (x >> 32) * (2^32 / p) * (1 - 2^-18).  The dropped low word costs < 2^32 / p <= 2^-8, the down-bias
// < 64 * 2^-18 and fp32 rounding (three roundings, each 2^-24 relative, all dominated by the bias) keeps
// e <= x / p, so q = floor(e) is floor(x / p) or one less and one conditional subtract finishes.
struct HeadroomReducer {
    uint64_t p;
    float scale;
    __device__ __forceinline__ explicit HeadroomReducer(uint64_t modulus)
        : p(modulus), scale((4294967296.0f * (1.0f - 1.0f / 262144.0f)) / static_cast<float>(modulus)) {}
    __device__ __forceinline__ uint64_t operator()(uint64_t x) const {
        float high;  // asm: hipcc otherwise converts through its generic 64-bit path (7 instructions)
        asm("v_cvt_f32_u32 %0, %1" : "=v"(high) : "v"(hi32(x)));
        const uint32_t q = static_cast<uint32_t>(high * scale);
        const uint64_t qp = mad32(q, static_cast<uint32_t>(p),
                                  static_cast<uint64_t>(mullo32(q, static_cast<uint32_t>(p >> 32))) << 32);
        return csub63<true>(x - qp, 0 - p);
    }
};

template <int MODE, int N>
__device__ __forceinline__ void canonicalize_all(uint64_t (&v)[N], uint64_t p) {
    if constexpr (is_headroom(MODE)) {
        const HeadroomReducer reduce(p);
#pragma unroll
        for (int r = 0; r < N; ++r) v[r] = reduce(v[r]);
    } else {
#pragma unroll
        for (int r = 0; r < N; ++r) v[r] = canonicalize<MODE>(v[r], p);
    }
}

// Pass schedule: P = ceil(LOGN / LOGE) passes; the partial pass (R = LOGN - (P-1) LOGE bits) sits on the low bits,
// i.e. it is the LAST forward pass and the FIRST inverse pass.
template <int LOGN, int LOGE>
struct Schedule {
    static constexpr int P = (LOGN + LOGE - 1) / LOGE;
    static constexpr int R = LOGN - (P - 1) * LOGE;
};


}  // namespace ntt
}  // namespace heamd
