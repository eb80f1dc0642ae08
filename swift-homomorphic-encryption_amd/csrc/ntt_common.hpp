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

template <bool APPROX>
struct Lazy {
    // values live in [0, BOUND * p)
    static constexpr int kBound = APPROX ? 8 : 4;
    __device__ static __forceinline__ uint64_t mul(uint64_t x, U64x2 w, uint64_t neg_p) {
        if constexpr (APPROX) {
            return shoup_lazy4(x, w.x, w.y, neg_p);
        } else {
            return shoup_lazy(x, w.x, w.y, neg_p);
        }
    }
};

// ---- forward pass over element bits [LO, LO+W): stages run from the top bit down --------------------------------
template <int LOGN, int LOGE, int LO, int W, bool APPROX, bool UNIFORM_TWIDDLES, int ABLATE = 0>
__device__ __forceinline__ void forward_pass(uint64_t (&v)[1 << LOGE], uint32_t tid, const U64x2* __restrict__ tw,
                                             uint64_t p, bool first_stage_canonical) {
    constexpr int E = 1 << LOGE;
    const uint64_t neg_p = opaque(0 - p);  // keep in VGPRs: a uniform multiplicand triggers a poor 64-bit expansion
    const uint64_t half_bound = (APPROX ? 4 : 2) * p;  // Harvey: fold x into [0, half_bound) before the butterfly
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
            const U64x2 w = (ABLATE & 1) ? tw[(1u << s)] : tw_stage[register_part<LOGN, LOGE, LO, W>(base) >> (b + 1)];
#pragma unroll
            for (int o = 0; o < stride; ++o) {
                uint64_t x = v[base + o];
                const uint64_t y = v[base + o + stride];
                if (!(first_stage_canonical && j == 0) && !(ABLATE & 8)) x = csub(x, half_bound);
                const uint64_t t = Lazy<APPROX>::mul(y, w, neg_p);
                v[base + o] = x + t;
                v[base + o + stride] = x + half_bound - t;
            }
        }
    }
    (void)UNIFORM_TWIDDLES;
}

// ---- inverse pass over element bits [LO, LO+W): stages run from the low bit up; the very last stage of the
// transform (bit LOGN-1) folds in N^-1 and N^-1 psi^(-N/2) and produces canonical words --------------------------
template <int LOGN, int LOGE, int LO, int W, bool APPROX>
__device__ __forceinline__ void inverse_pass(uint64_t (&v)[1 << LOGE], uint32_t tid, const U64x2* __restrict__ tw,
                                             const DeviceModulus& mod, bool first_stage_canonical) {
    constexpr int E = 1 << LOGE;
    constexpr uint32_t N = 1u << LOGN;
    const uint64_t p = mod.p;
    const uint64_t neg_p = opaque(0 - p);
    const uint64_t bound = (APPROX ? 4 : 2) * p;  // inputs/outputs of a stage live in [0, bound)
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const int b = LO + j;
        const int stride = 1 << (b - LO);
        const uint32_t m = N >> (b + 1);
        const bool last_stage = (b == LOGN - 1);
#pragma unroll
        for (int base = 0; base < E; base += 2 * stride) {
            U64x2 w = {0, 0};
            if (!last_stage)
                w = (tw + (N - 2 * m + 1) + (lane_part<LOGN, LOGE, LO, W>(tid) >> (b + 1)))
                    [register_part<LOGN, LOGE, LO, W>(base) >> (b + 1)];
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
                    if (!(first_stage_canonical && j == 0)) sum = csub(sum, bound);
                    v[base + o] = sum;
                    v[base + o + stride] = Lazy<APPROX>::mul(diff, w, neg_p);
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

template <bool APPROX>
__device__ __forceinline__ uint64_t canonicalize(uint64_t x, uint64_t p) {
    if constexpr (APPROX) x = csub(x, 4 * p);
    x = csub(x, 2 * p);
    return csub(x, p);
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
