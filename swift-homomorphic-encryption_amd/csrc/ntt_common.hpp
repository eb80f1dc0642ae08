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

// Butterfly arithmetic modes (chosen per launch from the moduli the launch covers):
//   kModeExact    any p <= 2^62 - 1 : exact Shoup quotient, products in [0, 2p), values in [0, 4p)
//   kModeApprox   p < 2^61          : 3-multiply quotient estimate, products in [0, 4p), values in [0, 8p)
//   kModeHeadroom 2^40 <= p < 2^55  : shoup_headroom (products in [0, 8p)), and the spare top bits absorb the growth
//                                     instead of a conditional subtract per butterfly: the forward transform never
//                                     folds (a word gains at most 8p per stage: < (1 + 8 log2 N) p <= 113 p < 2^62),
//                                     the inverse folds only once sums could pass 2^6 p (its multiplicand x + B - y
//                                     must stay < 2^62); one float-estimated quotient brings forward outputs back
//                                     to [0, p).
constexpr int kModeExact = 0, kModeApprox = 1, kModeHeadroom = 2;

template <int MODE>
struct Lazy {
    static constexpr int kProductLog = MODE == kModeExact ? 1 : MODE == kModeApprox ? 2 : 3;  // products < p << this
    // cap on stage inputs of the inverse transform, as a shift of p
    static constexpr int kInverseCapLog = MODE == kModeExact ? 1 : MODE == kModeApprox ? 2 : 6;
    // twiddle as the butterflies want it: headroom mode multiplies by floor(w 2^63 / p) = wf >> 1
    __device__ static __forceinline__ U64x2 prepare(U64x2 w) {
        if constexpr (MODE == kModeHeadroom) w.y >>= 1;
        return w;
    }
    // `reduction` = 2^64 - p (exact / approx, in VGPRs) or 2^64 - 2p (headroom, uniform)
    __device__ static __forceinline__ uint64_t mul(uint64_t x, U64x2 w, uint64_t reduction) {
        if constexpr (MODE == kModeHeadroom) {
            return shoup_headroom(x, w.x, w.y, reduction);
        } else if constexpr (MODE == kModeApprox) {
            return shoup_lazy4(x, w.x, w.y, reduction);
        } else {
            return shoup_lazy(x, w.x, w.y, reduction);
        }
    }
    __device__ static __forceinline__ uint64_t reduction_constant(uint64_t p) {
        if constexpr (MODE == kModeHeadroom) {
            return 0 - 2 * p;
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
    static_assert(MODE != kModeHeadroom || 1 + 8 * LOGN <= 127, "headroom mode: growth must stay below 2^7 p");
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const int b = LO + W - 1 - j;         // element bit paired by this stage
        const int s = LOGN - 1 - b;           // global stage number; m = 2^s groups
        const int stride = 1 << (b - LO);     // register distance of a pair
#pragma unroll
        for (int base = 0; base < E; base += 2 * stride) {
            // twiddle index = 2^s + (element index >> (b+1)); lane and register parts are disjoint, so the lane part
            // is one address per stage and the register part an immediate offset
            const U64x2* const tw_stage = tw + (1u << s) + (lane_part<LOGN, LOGE, LO, W>(tid) >> (b + 1));
            // ABLATE bit 0 (measurement only, wrong results): one wave-uniform twiddle instead of the gather
            const U64x2 w = Lazy<MODE>::prepare(
                (ABLATE & 1) ? tw[(1u << s)] : tw_stage[register_part<LOGN, LOGE, LO, W>(base) >> (b + 1)]);
#pragma unroll
            for (int o = 0; o < stride; ++o) {
                uint64_t x = v[base + o];
                const uint64_t y = v[base + o + stride];
                if (MODE != kModeHeadroom && !(first_stage_canonical && j == 0) && !(ABLATE & 8)) x = csub(x, half_bound);
                const uint64_t t = Lazy<MODE>::mul(y, w, neg_p);
                v[base + o] = x + t;
                v[base + o + stride] = x + half_bound - t;
            }
        }
    }
    (void)UNIFORM_TWIDDLES;
}

// ---- inverse pass over element bits [LO, LO+W): stages run from the low bit up; the very last stage of the
// transform (bit LOGN-1) folds in N^-1 and N^-1 psi^(-N/2) and produces canonical words --------------------------
template <int LOGN, int LOGE, int LO, int W, int MODE>
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
#pragma unroll
        for (int base = 0; base < E; base += 2 * stride) {
            U64x2 w = {0, 0};
            if (!last_stage)
                w = Lazy<MODE>::prepare((tw + (N - 2 * m + 1) + (lane_part<LOGN, LOGE, LO, W>(tid) >> (b + 1)))
                                            [register_part<LOGN, LOGE, LO, W>(base) >> (b + 1)]);
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
                    if (fold) sum = csub(sum, bound);
                    v[base + o] = sum;
                    v[base + o + stride] = Lazy<MODE>::mul(diff, w, neg_p);
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
__device__ __forceinline__ uint64_t canonicalize(uint64_t x, uint64_t p) {
    static_assert(MODE == kModeExact || MODE == kModeApprox, "headroom outputs go through HeadroomReducer");
    if constexpr (MODE == kModeApprox) x = csub(x, 4 * p);
    x = csub(x, 2 * p);
    return csub(x, p);
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
        const uint32_t q = static_cast<uint32_t>(static_cast<float>(static_cast<uint32_t>(x >> 32)) * scale);
        const uint64_t qp = mad32(q, static_cast<uint32_t>(p),
                                  static_cast<uint64_t>(mullo32(q, static_cast<uint32_t>(p >> 32))) << 32);
        return csub(x - qp, p);
    }
};

template <int MODE, int N>
__device__ __forceinline__ void canonicalize_all(uint64_t (&v)[N], uint64_t p) {
    if constexpr (MODE == kModeHeadroom) {
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
