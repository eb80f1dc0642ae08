// ntt_pipelined.hip -- persistent, software-pipelined NTT kernels (the production path for N = 4096 .. 16384).
//
// Same arithmetic and data mapping as the tiled kernels in ntt_kernels.hip (one residue row per workgroup, three
// register passes, two LDS transposes), restructured after measuring where the tiled kernel's time went
// (profiles/r01_ntt_ablation.txt: compute 0.57 ms + exposed global latency 0.39 ms + exposed twiddle-gather latency
// 0.29 ms were *adding up*, because all workgroups run phase-aligned):
//   * persistent workgroups: grid = 2 x CUs (a multiple of the moduli count), each workgroup walks rows
//     blockIdx.x, + gridDim.x, ...; a workgroup therefore stays on ONE modulus (constants and pass-0 scalar twiddles
//     are loop invariant);
//   * every long-latency load is issued one phase ahead of its use: the gathered twiddles of the next pass while the
//     current pass computes, the next row's coefficients while the last pass computes;
//   * the second wave of workgroups on a CU starts half a row late, so the two resident workgroups alternate between
//     memory and ALU phases instead of marching in lock step.
#include <hip/hip_runtime.h>

#include "device_context.hpp"
#include "device_math.hpp"
#include "kernels.hpp"
#include "ntt_common.hpp"

namespace heamd {

namespace {

using namespace ntt;

// ---- twiddles of one pass, held in registers --------------------------------------------------------------------
// forward pass over bits [LO, LO+W): local stage j pairs bit b = LO+W-1-j and has E >> (W-j) distinct twiddles
template <int LOGE, int W>
__host__ __device__ constexpr int forward_stage_offset(int j) {
    int offset = 0;
    for (int i = 0; i < j; ++i) offset += (1 << LOGE) >> (W - i);
    return offset;
}
// inverse pass: local stage j pairs bit b = LO+j and has E >> (j+1) distinct twiddles
template <int LOGE, int W>
__host__ __device__ constexpr int inverse_stage_offset(int j) {
    int offset = 0;
    for (int i = 0; i < j; ++i) offset += (1 << LOGE) >> (i + 1);
    return offset;
}
template <int LOGE, int W>
constexpr int kPassTwiddles = (1 << LOGE) - (1 << (LOGE - W));
template <int LOGE, int W>
using TwiddleRegs = U64x2[kPassTwiddles<LOGE, W>];

template <int LOGN, int LOGE, int LO, int W, int J0 = 0, int J1 = W>
__device__ __forceinline__ void load_forward_twiddles(TwiddleRegs<LOGE, W>& t, uint32_t tid,
                                                      const U64x2* __restrict__ tw) {
    constexpr int E = 1 << LOGE;
#pragma unroll
    for (int j = J0; j < J1; ++j) {
        const int b = LO + W - 1 - j, s = LOGN - 1 - b, stride = 1 << (b - LO);
#pragma unroll
        for (int g = 0; g < E / (2 * stride); ++g) {
            t[forward_stage_offset<LOGE, W>(j) + g] =
                (tw + (1u << s) + (lane_part<LOGN, LOGE, LO, W>(tid) >> (b + 1)))
                    [register_part<LOGN, LOGE, LO, W>(g * 2 * stride) >> (b + 1)];
        }
    }
}

template <int LOGN, int LOGE, int LO, int W, int J0 = 0, int J1 = W>
__device__ __forceinline__ void load_inverse_twiddles(TwiddleRegs<LOGE, W>& t, uint32_t tid,
                                                      const U64x2* __restrict__ tw) {
    constexpr int E = 1 << LOGE;
    constexpr uint32_t N = 1u << LOGN;
#pragma unroll
    for (int j = J0; j < J1; ++j) {
        const int b = LO + j, stride = 1 << j;
        const uint32_t m = N >> (b + 1);
#pragma unroll
        for (int g = 0; g < E / (2 * stride); ++g) {
            t[inverse_stage_offset<LOGE, W>(j) + g] =
                (tw + (N - 2 * m + 1) + (lane_part<LOGN, LOGE, LO, W>(tid) >> (b + 1)))
                    [register_part<LOGN, LOGE, LO, W>(g * 2 * stride) >> (b + 1)];
        }
    }
}

// local stages [J0, J1) of a forward pass with register twiddles
template <int LOGN, int LOGE, int LO, int W, bool APPROX, int J0, int J1>
__device__ __forceinline__ void forward_stages(uint64_t (&v)[1 << LOGE], const TwiddleRegs<LOGE, W>& t,
                                               uint64_t p, uint64_t neg_p) {
    constexpr int E = 1 << LOGE;
    const uint64_t half_bound = (APPROX ? 4 : 2) * p;
#pragma unroll
    for (int j = J0; j < J1; ++j) {
        const int stride = 1 << (W - 1 - j);
#pragma unroll
        for (int g = 0; g < E / (2 * stride); ++g) {
            const U64x2 w = t[forward_stage_offset<LOGE, W>(j) + g];
#pragma unroll
            for (int o = 0; o < stride; ++o) {
                const int lo = g * 2 * stride + o, hi = lo + stride;
                const uint64_t x = csub(v[lo], half_bound);
                const uint64_t tt = Lazy<APPROX>::mul(v[hi], w, neg_p);
                v[lo] = x + tt;
                v[hi] = x + half_bound - tt;
            }
        }
    }
}

// local stages [J0, J1) of an inverse pass with register twiddles (never the transform's last stage)
template <int LOGN, int LOGE, int LO, int W, bool APPROX, int J0, int J1>
__device__ __forceinline__ void inverse_stages(uint64_t (&v)[1 << LOGE], const TwiddleRegs<LOGE, W>& t,
                                               uint64_t p, uint64_t neg_p, bool first_stage_canonical) {
    constexpr int E = 1 << LOGE;
    static_assert(LO + W < LOGN, "the pass holding the last stage uses inverse_pass (uniform twiddles)");
    const uint64_t bound = (APPROX ? 4 : 2) * p;
#pragma unroll
    for (int j = J0; j < J1; ++j) {
        const int stride = 1 << j;
#pragma unroll
        for (int g = 0; g < E / (2 * stride); ++g) {
            const U64x2 w = t[inverse_stage_offset<LOGE, W>(j) + g];
#pragma unroll
            for (int o = 0; o < stride; ++o) {
                const int lo = g * 2 * stride + o, hi = lo + stride;
                const uint64_t x = v[lo], y = v[hi];
                uint64_t sum = x + y;
                if (!(first_stage_canonical && j == 0)) sum = csub(sum, bound);
                v[lo] = sum;
                v[hi] = Lazy<APPROX>::mul(x + bound - y, w, neg_p);
            }
        }
    }
}

// Half a row period of idle time for the second wave of workgroups on each CU (about 4 us at 2 GHz).
__device__ __forceinline__ void stagger_second_wave(uint32_t first_wave_blocks) {
    if (blockIdx.x >= first_wave_blocks) {
#pragma unroll 1
        for (int i = 0; i < 2; ++i) __builtin_amdgcn_s_sleep(127);
    }
}

enum : int { kFlagPrefetchRow = 1, kFlagStagger = 2 };

// A persistent workgroup stays on one modulus, so its twiddle gathers are loop invariant and hipcc hoists ALL of
// them out of the row loop (236 VGPRs of twiddles -> spills).  Laundering the (wave-uniform) table pointer once per
// iteration keeps each gather inside the iteration, where its registers die after the pass that uses them.
// The same goes for the ~200 loop-invariant LDS / global / twiddle addresses derived from the lane id: recomputing
// them costs a few VALU ops, hoisting them costs a spill each.
__device__ __forceinline__ uint32_t per_iteration(uint32_t lane_id) {
    asm volatile("" : "+v"(lane_id));
    return lane_id;
}
template <typename T>
__device__ __forceinline__ const T* per_iteration(const T* pointer) {
    asm volatile("" : "+v"(pointer));
    // back to a scalar register pair: the table base is wave-uniform (scalar loads for pass 0, saddr for the gathers)
    const uint64_t bits = reinterpret_cast<uint64_t>(pointer);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(bits));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(bits >> 32));
    return reinterpret_cast<const T*>((static_cast<uint64_t>(hi) << 32) | lo);
}

template <int LOGN, int LOGT, bool APPROX, int FLAGS>
__global__ void __launch_bounds__(1 << LOGT, 2)
    ntt_forward_pipelined(uint64_t* __restrict__ slab, const DeviceContext ctx, uint32_t mod_base,
                          uint32_t mod_period, size_t rows, uint32_t first_wave_blocks) {
    constexpr int LOGE = LOGN - LOGT;
    constexpr int E = 1 << LOGE;
    using S = Schedule<LOGN, LOGE>;
    static_assert(S::P == 3, "the pipelined kernel is written for three passes");
    constexpr int LO0 = LOGN - LOGE, LO1 = LOGN - 2 * LOGE, R = S::R;
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
    const uint32_t lane_id = threadIdx.x;
    uint32_t tid = lane_id;
    size_t row = blockIdx.x;
    if (row >= rows) return;
    const uint32_t mi = mod_base + static_cast<uint32_t>(row % mod_period);  // invariant: gridDim.x % mod_period == 0
    const DeviceModulus mod = ctx.moduli[mi];
    const U64x2* tw = ctx.forward_twiddles + (static_cast<size_t>(mi) << LOGN);
    const uint64_t p = mod.p;
    const uint64_t neg_p = opaque(0 - p);
    if constexpr (FLAGS & kFlagStagger) stagger_second_wave(first_wave_blocks);

    uint64_t cur[E], nxt[E];
    global_load<LOGN, LOGE, LO0, LOGE>(cur, tid, slab + (row << LOGN));
    const U64x2* const tw_table = tw;
    for (;;) {
        const size_t next_row = row + gridDim.x;
        const bool has_next = next_row < rows;
        const U64x2* const tw_gather = per_iteration(tw_table);
        tid = per_iteration(lane_id) & ((1u << LOGT) - 1u);
        // gathered twiddles of the middle pass: issued now, consumed after pass 0 and the first transpose
        U64x2 t_mid[kPassTwiddles<LOGE, LOGE>];
        load_forward_twiddles<LOGN, LOGE, LO1, LOGE, 0, LOGE - 1>(t_mid, tid, tw_gather);  // stages 0..LOGE-2: E/2-1 pairs
        forward_pass<LOGN, LOGE, LO0, LOGE, APPROX, true>(cur, tid, tw_gather, p, true);
        lds_store<LOGN, LOGE, LO0, LOGE>(cur, tid, lds);
        __syncthreads();
        lds_load<LOGN, LOGE, LO1, LOGE>(cur, tid, lds);
        load_forward_twiddles<LOGN, LOGE, LO1, LOGE, LOGE - 1, LOGE>(t_mid, tid, tw_gather);  // last stage: E/2 pairs
        forward_stages<LOGN, LOGE, LO1, LOGE, APPROX, 0, LOGE>(cur, t_mid, p, neg_p);
        lds_store<LOGN, LOGE, LO1, LOGE>(cur, tid, lds);
        lds_transpose_fence<LOGN, LOGE, LO1, 0>();  // stays inside the wave for every supported shape
        lds_load<LOGN, LOGE, 0, R>(cur, tid, lds);
        __syncthreads();  // the next iteration's first lds_store must not overtake these reads
        if constexpr (FLAGS & kFlagPrefetchRow) {
            if (has_next) global_load<LOGN, LOGE, LO0, LOGE>(nxt, tid, slab + (next_row << LOGN));
        }
        // the last pass gathers its twiddles stage by stage (28 pairs would not fit next to the prefetched row)
        forward_pass<LOGN, LOGE, 0, R, APPROX, false>(cur, tid, tw_gather, p, false);
#pragma unroll
        for (int r = 0; r < E; ++r) cur[r] = canonicalize<APPROX>(cur[r], p);
        global_store<LOGN, LOGE, 0, R>(cur, tid, slab + (row << LOGN));
        if (!has_next) break;
        if constexpr (FLAGS & kFlagPrefetchRow) {
#pragma unroll
            for (int r = 0; r < E; ++r) cur[r] = nxt[r];
        } else {
            global_load<LOGN, LOGE, LO0, LOGE>(cur, tid, slab + (next_row << LOGN));
        }
        row = next_row;
    }
}

template <int LOGN, int LOGT, bool APPROX, int FLAGS>
__global__ void __launch_bounds__(1 << LOGT, 2)
    ntt_inverse_pipelined(uint64_t* __restrict__ slab, const DeviceContext ctx, uint32_t mod_base,
                          uint32_t mod_period, size_t rows, uint32_t first_wave_blocks) {
    constexpr int LOGE = LOGN - LOGT;
    constexpr int E = 1 << LOGE;
    using S = Schedule<LOGN, LOGE>;
    static_assert(S::P == 3, "the pipelined kernel is written for three passes");
    constexpr int R = S::R, LO1 = R, LO2 = LOGN - LOGE;
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
    const uint32_t lane_id = threadIdx.x;
    uint32_t tid = lane_id;
    size_t row = blockIdx.x;
    if (row >= rows) return;
    const uint32_t mi = mod_base + static_cast<uint32_t>(row % mod_period);
    const DeviceModulus mod = ctx.moduli[mi];
    const U64x2* tw = ctx.inverse_twiddles + (static_cast<size_t>(mi) << LOGN);
    const uint64_t p = mod.p;
    const uint64_t neg_p = opaque(0 - p);
    if constexpr (FLAGS & kFlagStagger) stagger_second_wave(first_wave_blocks);

    uint64_t cur[E], nxt[E];
    U64x2 t_low[kPassTwiddles<LOGE, R>];
    load_inverse_twiddles<LOGN, LOGE, 0, R>(t_low, tid, tw);
    global_load<LOGN, LOGE, 0, R>(cur, tid, slab + (row << LOGN));
    const U64x2* const tw_table = tw;
    for (;;) {
        const size_t next_row = row + gridDim.x;
        const bool has_next = next_row < rows;
        const U64x2* const tw_gather = per_iteration(tw_table);
        tid = per_iteration(lane_id) & ((1u << LOGT) - 1u);
        inverse_stages<LOGN, LOGE, 0, R, APPROX, 0, 1>(cur, t_low, p, neg_p, true);
        // middle-pass twiddles: its first stage (E/2 pairs) is issued once the low pass's largest batch is consumed,
        // the remaining E/2-1 pairs after the transpose
        U64x2 t_mid[kPassTwiddles<LOGE, LOGE>];
        load_inverse_twiddles<LOGN, LOGE, LO1, LOGE, 0, 1>(t_mid, tid, tw_gather);
        inverse_stages<LOGN, LOGE, 0, R, APPROX, 1, R>(cur, t_low, p, neg_p, false);
        lds_store<LOGN, LOGE, 0, R>(cur, tid, lds);
        lds_transpose_fence<LOGN, LOGE, 0, LO1>();  // stays inside the wave for every supported shape
        lds_load<LOGN, LOGE, LO1, LOGE>(cur, tid, lds);
        load_inverse_twiddles<LOGN, LOGE, LO1, LOGE, 1, LOGE>(t_mid, tid, tw_gather);
        inverse_stages<LOGN, LOGE, LO1, LOGE, APPROX, 0, LOGE>(cur, t_mid, p, neg_p, false);
        lds_store<LOGN, LOGE, LO1, LOGE>(cur, tid, lds);
        __syncthreads();
        lds_load<LOGN, LOGE, LO2, LOGE>(cur, tid, lds);
        __syncthreads();
        // the top pass uses wave-uniform twiddles: registers are free to prefetch the next row and its low-pass
        // twiddles (the same ones every iteration; they just do not fit in registers across the middle pass)
        if (has_next) {
            if constexpr (FLAGS & kFlagPrefetchRow) {
                global_load<LOGN, LOGE, 0, R>(nxt, tid, slab + (next_row << LOGN));
            } else {
                load_inverse_twiddles<LOGN, LOGE, 0, R>(t_low, tid, tw_gather);
            }
        }
        inverse_pass<LOGN, LOGE, LO2, LOGE, APPROX>(cur, tid, tw_gather, mod, false);
        global_store<LOGN, LOGE, LO2, LOGE>(cur, tid, slab + (row << LOGN));
        if (!has_next) break;
        if constexpr (FLAGS & kFlagPrefetchRow) {
            load_inverse_twiddles<LOGN, LOGE, 0, R>(t_low, tid, tw_gather);
#pragma unroll
            for (int r = 0; r < E; ++r) cur[r] = nxt[r];
        } else {
            global_load<LOGN, LOGE, 0, R>(cur, tid, slab + (next_row << LOGN));
        }
        row = next_row;
    }
}

int device_cu_count() {
    static int cached = 0;
    if (cached == 0) {
        int device = 0, count = 0;
        if (hipGetDevice(&device) == hipSuccess &&
            hipDeviceGetAttribute(&count, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && count > 0) {
            cached = count;
        } else {
            cached = 256;
        }
    }
    return cached;
}

template <int LOGN, int LOGT, int FLAGS>
hipError_t launch_pipelined_flags(bool inverse, bool approx, uint64_t* slab, const DeviceContext& ctx,
                                  uint32_t mod_base, uint32_t mod_period, size_t rows, hipStream_t stream) {
    constexpr size_t lds_bytes = lds_words(1u << LOGN) * sizeof(uint64_t);
    using Kernel = void (*)(uint64_t*, const DeviceContext, uint32_t, uint32_t, size_t, uint32_t);
    Kernel kernel;
    if (inverse) {
        kernel = approx ? ntt_inverse_pipelined<LOGN, LOGT, true, FLAGS> : ntt_inverse_pipelined<LOGN, LOGT, false, FLAGS>;
    } else {
        kernel = approx ? ntt_forward_pipelined<LOGN, LOGT, true, FLAGS> : ntt_forward_pipelined<LOGN, LOGT, false, FLAGS>;
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
    if (e != hipSuccess) return e;
    // resident workgroups per CU: limited by the LDS tile (160 KiB per CU)
    const size_t per_cu = (160 * 1024) / lds_bytes > 0 ? (160 * 1024) / lds_bytes : 1;
    const size_t cus = static_cast<size_t>(device_cu_count());
    size_t first_wave = cus / mod_period * mod_period;  // keep every wave of workgroups a multiple of the period
    if (first_wave == 0) first_wave = mod_period;
    size_t grid = first_wave * (per_cu > 4 ? 4 : per_cu);
    const size_t rows_rounded = (rows + mod_period - 1) / mod_period * mod_period;
    if (grid > rows_rounded) grid = rows_rounded;
    hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(grid)), dim3(1u << LOGT), lds_bytes, stream, slab, ctx,
                       mod_base, mod_period, rows, static_cast<uint32_t>(first_wave));
    return hipGetLastError();
}

template <int LOGN, int LOGT>
hipError_t launch_pipelined_size(bool inverse, bool approx, int flags, uint64_t* slab, const DeviceContext& ctx,
                                 uint32_t mod_base, uint32_t mod_period, size_t rows, hipStream_t stream) {
    switch (flags & 3) {
        case 0: return launch_pipelined_flags<LOGN, LOGT, 0>(inverse, approx, slab, ctx, mod_base, mod_period, rows, stream);
        case 1: return launch_pipelined_flags<LOGN, LOGT, 1>(inverse, approx, slab, ctx, mod_base, mod_period, rows, stream);
        case 2: return launch_pipelined_flags<LOGN, LOGT, 2>(inverse, approx, slab, ctx, mod_base, mod_period, rows, stream);
        default: return launch_pipelined_flags<LOGN, LOGT, 3>(inverse, approx, slab, ctx, mod_base, mod_period, rows, stream);
    }
}

}  // namespace

bool ntt_pipelined_supports(uint32_t log_degree) { return log_degree == 12 || log_degree == 13 || log_degree == 14; }

hipError_t launch_ntt_pipelined(bool inverse, bool approx, int flags, uint64_t* slab, const DeviceContext& ctx,
                                uint32_t mod_base, uint32_t mod_period, size_t rows, hipStream_t stream) {
    switch (ctx.log_degree) {
        case 12: return launch_pipelined_size<12, 8>(inverse, approx, flags, slab, ctx, mod_base, mod_period, rows, stream);
        case 13: return launch_pipelined_size<13, 8>(inverse, approx, flags, slab, ctx, mod_base, mod_period, rows, stream);
        case 14: return launch_pipelined_size<14, 9>(inverse, approx, flags, slab, ctx, mod_base, mod_period, rows, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace heamd
