// ntt_stream.hip -- persistent forward NTT whose next row streams into LDS while the current row finishes.
//
// The tiled kernel of ntt_kernels.hip is register-bound at 8 waves per SIMD and LDS-bound at two rows per CU, so a
// row's global-load latency, its store drain and the workgroup launch cannot be covered by a third resident row
// (profiles/r01h_ntt_ablation.txt: 0.43 ms of multiplier work inside a 0.61 ms launch).  This kernel keeps the same
// arithmetic, passes and LDS tile, and changes only how rows arrive:
//   * persistent workgroups (2 per CU) walk rows blockIdx.x, + gridDim.x, ...;
//   * after the LAST transpose of row k a wave no longer needs its private slice of the LDS tile, so it fires the
//     LDS-DMA loads (global_load_lds_dwordx4: no VGPRs, 1 KiB per wave instruction) of ITS chunk of row k + 1 into
//     that slice, then computes the last pass of row k and stores it;
//   * the top of row k + 1 waits for the DMA only (counted vmcnt: the stores issued after it may still be in
//     flight), crosses a barrier and picks the first pass's words out of the LDS image.
// Measured (profiles/r01h_ntt_stream.txt): 0.654 ms against the tiled kernel's 0.616 ms at N = 8192, L = 4, 4096
// polynomials -- the two extra barriers, the LDS image read and the lock step of equal-cost rows (a staggered start of
// the second workgroup per CU did not help) cost more than the hidden load latency returns.  Kept as variant 11 for
// the record and for parity tests; launch_ntt does not select it.
// hipcc does not count asm memory operations, so every load/store that is in flight across the DMA (the last pass's
// twiddle gathers, the row stores) is asm as well and waited for by hand; see cdna_hip_programming.md section 5.
#include <hip/hip_runtime.h>

#include "device_context.hpp"
#include "device_math.hpp"
#include "kernels.hpp"
#include "ntt_common.hpp"

namespace heamd {

namespace {

using namespace ntt;

typedef uint32_t Word4 __attribute__((ext_vector_type(4)));

// loop-invariant lane arithmetic is cheaper to redo per row than to keep in registers (64-VGPR budget)
__device__ __forceinline__ uint32_t per_row(uint32_t lane_id) {
    asm volatile("" : "+v"(lane_id));
    return lane_id;
}

// pins a wave-uniform pointer to a scalar register pair (asm "s" operands do not get one by themselves when hipcc has
// parked the value in vector registers)
template <typename T>
__device__ __forceinline__ T* uniform_pointer(T* pointer) {
    const uint64_t bits = reinterpret_cast<uint64_t>(pointer);
    const uint32_t lo = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(bits));
    const uint32_t hi = __builtin_amdgcn_readfirstlane(static_cast<uint32_t>(bits >> 32));
    return reinterpret_cast<T*>((static_cast<uint64_t>(hi) << 32) | lo);
}

template <int LOGN, int LOGT>
struct StreamShape {
    static constexpr int LOGE = LOGN - LOGT;
    static constexpr int WB = LOGT > 6 ? LOGT - 6 : 0;       // wave-id bits
    static constexpr int LOG_CHUNK = LOGN - WB;             // a wave's private slice of the row, in words
    static constexpr int CHUNK = 1 << LOG_CHUNK;
    static constexpr int DMA_PER_WAVE = CHUNK / 128;        // 64 lanes x 16 B = 128 words per instruction
    static constexpr int STORES = (1 << LOGE) / 2;          // 16 B row stores per lane
};

// wave `chunk`'s slice of the padded tile starts at lds_slot(chunk * CHUNK); the DMA image is linear inside it
template <int LOGN, int LOGT>
__device__ __forceinline__ uint32_t image_slot(uint32_t element) {
    using Shape = StreamShape<LOGN, LOGT>;
    return lds_slot((element >> Shape::LOG_CHUNK) << Shape::LOG_CHUNK) + (element & (Shape::CHUNK - 1));
}

template <int LOGN, int LOGT>
__device__ __forceinline__ void prefetch_row(const uint64_t* row, uint32_t tid, uint64_t* lds) {
    using Shape = StreamShape<LOGN, LOGT>;
#ifdef HEAMD_STREAM_NO_DMA  // measurement only: the transform runs on whatever the tile holds
    return;
#endif
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63u;
    const uint32_t lds_base = static_cast<uint32_t>(reinterpret_cast<uintptr_t>(
                                  (__attribute__((address_space(3))) uint64_t*)(lds))) +
                              lds_slot(wave << Shape::LOG_CHUNK) * 8u;
    const uint64_t* source = uniform_pointer(row + (static_cast<size_t>(wave) << Shape::LOG_CHUNK));
    const uint32_t lane_bytes = lane * 16u;
#pragma unroll
    for (int k = 0; k < Shape::DMA_PER_WAVE; ++k) {
        uint32_t keep;
        asm volatile(
            "s_mov_b32 %0, m0\n\t"
            "s_mov_b32 m0, %3\n\t"
            "s_nop 4\n\t"  // the base may be fresh from v_readfirstlane: 5 wait states before a VMEM reads it
            "global_load_lds_dwordx4 %1, %2\n\t"
            "s_mov_b32 m0, %0"
            : "=&s"(keep)
            : "v"(lane_bytes), "s"(source + k * 128), "s"(lds_base + k * 1024u)
            : "memory");
    }
}

template <int LOGN, int LOGT, int MODE>
__global__ void __launch_bounds__(1 << LOGT, 8)
    ntt_forward_stream(uint64_t* __restrict__ slab, const DeviceContext ctx, uint32_t mod_base, uint32_t mod_period,
                       uint32_t rows) {
    using Shape = StreamShape<LOGN, LOGT>;
    constexpr int LOGE = Shape::LOGE;
    constexpr int E = 1 << LOGE;
    using S = Schedule<LOGN, LOGE>;
    static_assert(S::P >= 3 && S::P <= 5, "first pass, middle passes, last pass");
    static_assert(S::R == 1, "the hand-waited last pass is written for one stage");
    static_assert(is_headroom(MODE), "built for the headroom butterflies");
    constexpr int LO0 = LOGN - LOGE;
    constexpr int LO_LAST_FULL = LOGN - (S::P - 1) * LOGE;  // = R
    static_assert(kWaveOwnsTopBits<LOGN, LOGE, LO_LAST_FULL>, "the last transpose must stay inside the wave");
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
    const uint32_t lane_id = threadIdx.x;
    uint32_t row = blockIdx.x;
    if (row >= rows) return;

    prefetch_row<LOGN, LOGT>(slab + (static_cast<size_t>(row) << LOGN), lane_id, lds);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (;;) {
        const uint32_t next_row = row + gridDim.x;
        const bool has_next = next_row < rows;
        const uint32_t tid = per_row(lane_id);
        const uint32_t mi = mod_base + row % mod_period;
        // through the constant address space: with asm memory clobbers around, hipcc would otherwise fetch the
        // (immutable) modulus with a vector load into VGPRs
        const uint64_t p = *(const __attribute__((address_space(4))) uint64_t*)(&ctx.moduli[mi].p);
        const U64x2* __restrict__ tw = twiddle_table<MODE>(ctx, false) + (static_cast<size_t>(mi) << LOGN);
        uint64_t* __restrict__ x = slab + (static_cast<size_t>(row) << LOGN);
        uint64_t v[E];

        // every wave's DMA has landed (each waited for its own before arriving here)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < E; ++r) v[r] = lds[image_slot<LOGN, LOGT>(tid + (static_cast<uint32_t>(r) << LO0))];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();  // the image is consumed: the tile is free for the transposes
        asm volatile("" ::: "memory");

        forward_pass<LOGN, LOGE, LO0, LOGE, MODE, true>(v, tid, tw, p, true);
        lds_store<LOGN, LOGE, LO0, LOGE>(v, tid, lds);
        __syncthreads();
        {
            constexpr int LO1 = LOGN - 2 * LOGE;
            lds_load<LOGN, LOGE, LO1, LOGE>(v, tid, lds);
            forward_pass<LOGN, LOGE, LO1, LOGE, MODE, false>(v, tid, tw, p, false);
            lds_store<LOGN, LOGE, LO1, LOGE>(v, tid, lds);
            lds_transpose_fence<LOGN, LOGE, LO1, (S::P >= 4 ? LOGN - 3 * LOGE : 0)>();
        }
        if constexpr (S::P >= 4) {
            constexpr int LO2 = LOGN - 3 * LOGE;
            lds_load<LOGN, LOGE, LO2, LOGE>(v, tid, lds);
            forward_pass<LOGN, LOGE, LO2, LOGE, MODE, false>(v, tid, tw, p, false);
            lds_store<LOGN, LOGE, LO2, LOGE>(v, tid, lds);
            lds_transpose_fence<LOGN, LOGE, LO2, (S::P >= 5 ? LOGN - 4 * LOGE : 0)>();
        }
        if constexpr (S::P >= 5) {
            constexpr int LO3 = LOGN - 4 * LOGE;
            lds_load<LOGN, LOGE, LO3, LOGE>(v, tid, lds);
            forward_pass<LOGN, LOGE, LO3, LOGE, MODE, false>(v, tid, tw, p, false);
            lds_store<LOGN, LOGE, LO3, LOGE>(v, tid, lds);
            lds_transpose_fence<LOGN, LOGE, LO3, 0>();
        }

        // last pass (one stage on bit 0): its E/2 twiddles are gathered by hand, BEFORE the DMA, so that waiting for
        // them (vmcnt counts in order) does not wait for the next row
        const uint32_t lane_element = lane_part<LOGN, LOGE, 0, 1>(tid);
        const uint32_t twiddle_bytes = (lane_element >> 1) * 16u;
        Word4 t[E / 2];
#pragma unroll
        for (int g = 0; g < E / 2; ++g) {
            const U64x2* const entry =
                uniform_pointer(tw + (1u << (LOGN - 1)) + (register_part<LOGN, LOGE, 0, 1>(2 * g) >> 1));
            asm volatile("s_nop 4\n\tglobal_load_dwordx4 %0, %1, %2" : "=v"(t[g]) : "v"(twiddle_bytes), "s"(entry) : "memory");
        }
        lds_load<LOGN, LOGE, 0, 1>(v, tid, lds);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's slice of the tile is free from here on
        // One unconditional wait statement: a second one on another branch makes hipcc copy the twiddle registers
        // ahead of it, i.e. before the data has landed.  The last trip therefore fetches a dummy (its own row again).
        prefetch_row<LOGN, LOGT>(slab + (static_cast<size_t>(has_next ? next_row : row) << LOGN), tid, lds);
        asm volatile("s_waitcnt vmcnt(%4)"
                     : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3])
                     : "n"(Shape::DMA_PER_WAVE)
                     : "memory");
        static_assert(E / 2 == 4, "the wait statements name four twiddle registers");
        const uint64_t neg_2p = Lazy<MODE>::reduction_constant(p);
        const uint64_t cap = p << Lazy<MODE>::kProductLog;
#pragma unroll
        for (int g = 0; g < E / 2; ++g) {
            const uint64_t w = pack64(t[g].x, t[g].y);
            uint64_t wf = pack64(t[g].z, t[g].w);
            if constexpr (MODE == kModeHeadroom) wf >>= 1;
            const uint64_t a = v[2 * g], b = v[2 * g + 1];
            const uint64_t sum = shoup_headroom_fma<false>(a, b, w, wf, neg_2p);
            v[2 * g] = sum;
            v[2 * g + 1] = ((a << 1) + cap) - sum;
        }
        canonicalize_all<MODE>(v, p);
        const uint32_t store_bytes = lane_element * 8u;
#pragma unroll
        for (int g = 0; g < E / 2; ++g) {
            Word4 words;
            words.x = lo32(v[2 * g]);
            words.y = hi32(v[2 * g]);
            words.z = lo32(v[2 * g + 1]);
            words.w = hi32(v[2 * g + 1]);
#ifdef HEAMD_STREAM_NO_STORE  // measurement only
            if (words.x == 0x12345u && words.w == 0x6789u)
#endif
            asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2\n\ts_nop 1"
                         :
                         : "v"(store_bytes), "v"(words), "s"(uniform_pointer(x + register_part<LOGN, LOGE, 0, 1>(2 * g)))
                         : "memory");
        }
        if (!has_next) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the dummy DMA must not outlive this workgroup's LDS
            break;
        }
        // the DMA (older than the stores) has landed once at most the stores are outstanding
        asm volatile("s_waitcnt vmcnt(%0)" : : "n"(Shape::STORES) : "memory");
        row = next_row;
    }
}

// ---- persistent kernel with REGISTER prefetch: 16 words per lane ---------------------------------------------------
// Two rows per CU is an LDS limit, so with 16 words per lane (512 lanes per row) only 4 waves per SIMD are resident
// and 128 VGPRs per lane are available where the arithmetic needs ~75: the spare 32 hold the NEXT row, loaded at the
// top of the current one -- a whole row period of latency tolerance, no asm, no counted waits.  Transposes and
// butterflies are the tiled kernel's (<LOGN, LOGT> = <13, 9>: passes of 4 + 4 + 4 + 1 stages, three transposes).
// Measured (profiles/r01i_ntt_prefetch.txt): 0.648 ms against 0.625 ms for the same shape launched one workgroup per
// row and 0.626 ms for the production kernel; a hashed start delay per workgroup changes nothing.  With the load
// latency provably hidden and no gain, the 17 % the "no global load" ablation returns is the halved HBM traffic, not
// latency.  Kept as variant 12 for the record and for parity tests.
template <int LOGN, int LOGT, int MODE>
__global__ void __launch_bounds__(1 << LOGT, 4)
    ntt_forward_prefetch(uint64_t* __restrict__ slab, const DeviceContext ctx, uint32_t mod_base, uint32_t mod_period,
                         uint32_t rows) {
    constexpr int LOGE = LOGN - LOGT;
    constexpr int E = 1 << LOGE;
    using S = Schedule<LOGN, LOGE>;
    static_assert(S::P == 4, "written out for four passes");
    constexpr int LO0 = LOGN - LOGE, LO1 = LOGN - 2 * LOGE, LO2 = LOGN - 3 * LOGE;
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
    const uint32_t lane_id = threadIdx.x;
    uint32_t row = blockIdx.x;
    if (row >= rows) return;
    uint64_t v[E], nxt[E];
    global_load<LOGN, LOGE, LO0, LOGE>(v, lane_id, slab + (static_cast<size_t>(row) << LOGN));
    for (;;) {
        const uint32_t next_row = row + gridDim.x;
        const bool has_next = next_row < rows;
        const uint32_t tid = per_row(lane_id);  // keeps the lane-derived addresses out of loop-invariant registers
        if (has_next) global_load<LOGN, LOGE, LO0, LOGE>(nxt, tid, slab + (static_cast<size_t>(next_row) << LOGN));
        const uint32_t mi = mod_base + row % mod_period;
        const DeviceModulus mod = ctx.moduli[mi];
        const U64x2* __restrict__ tw = twiddle_table<MODE>(ctx, false) + (static_cast<size_t>(mi) << LOGN);
        uint64_t* __restrict__ x = slab + (static_cast<size_t>(row) << LOGN);
        const uint64_t p = mod.p;
        forward_pass<LOGN, LOGE, LO0, LOGE, MODE, true>(v, tid, tw, p, true);
        lds_store<LOGN, LOGE, LO0, LOGE>(v, tid, lds);
        __syncthreads();
        lds_load<LOGN, LOGE, LO1, LOGE>(v, tid, lds);
        forward_pass<LOGN, LOGE, LO1, LOGE, MODE, false>(v, tid, tw, p, false);
        lds_store<LOGN, LOGE, LO1, LOGE>(v, tid, lds);
        lds_transpose_fence<LOGN, LOGE, LO1, LO2>();
        lds_load<LOGN, LOGE, LO2, LOGE>(v, tid, lds);
        forward_pass<LOGN, LOGE, LO2, LOGE, MODE, false>(v, tid, tw, p, false);
        lds_store<LOGN, LOGE, LO2, LOGE>(v, tid, lds);
        lds_transpose_fence<LOGN, LOGE, LO2, 0>();
        lds_load<LOGN, LOGE, 0, S::R>(v, tid, lds);
        forward_pass<LOGN, LOGE, 0, S::R, MODE, false>(v, tid, tw, p, false);
        canonicalize_all<MODE>(v, p);
        global_store<LOGN, LOGE, 0, S::R>(v, tid, x);
        if (!has_next) break;
        // the next row's first transpose crosses waves: every wave must be done with this row's tile first
        __syncthreads();
#pragma unroll
        for (int r = 0; r < E; ++r) v[r] = nxt[r];
        row = next_row;
    }
}

}  // namespace

bool ntt_stream_supports(const DeviceContext& ctx) {
    return ctx.log_degree == 13 && ctx.approx_ok != 0 && ctx.headroom_ok != 0;
}

hipError_t launch_ntt_forward_stream(uint64_t* slab, const DeviceContext& ctx, uint32_t mod_base,
                                     uint32_t mod_period, size_t rows, uint32_t workgroups, hipStream_t stream) {
    if (!ntt_stream_supports(ctx) || rows > 0xffffffffull) return hipErrorNotSupported;
    if (rows == 0) return hipSuccess;
    constexpr int LOGN = 13, LOGT = 10;
    constexpr size_t lds_bytes = ntt::lds_words(1u << LOGN) * sizeof(uint64_t);
    auto kernel = ctx.forward_twiddles_half != nullptr ? ntt_forward_stream<LOGN, LOGT, ntt::kModeHeadroomHalved>
                                                       : ntt_forward_stream<LOGN, LOGT, ntt::kModeHeadroom>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
    if (e != hipSuccess) return e;
    const unsigned grid = static_cast<unsigned>(rows < workgroups ? rows : workgroups);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(1u << LOGT), lds_bytes, stream, slab, ctx, mod_base, mod_period,
                       static_cast<uint32_t>(rows));
    return hipGetLastError();
}

hipError_t launch_ntt_forward_prefetch(uint64_t* slab, const DeviceContext& ctx, uint32_t mod_base,
                                       uint32_t mod_period, size_t rows, uint32_t workgroups, hipStream_t stream) {
    if (ctx.log_degree != 13 || ctx.approx_ok == 0 || rows > 0xffffffffull) return hipErrorNotSupported;
    if (rows == 0) return hipSuccess;
    constexpr int LOGN = 13, LOGT = 9;
    constexpr size_t lds_bytes = ntt::lds_words(1u << LOGN) * sizeof(uint64_t);
    auto kernel = ctx.headroom_ok == 0                  ? ntt_forward_prefetch<LOGN, LOGT, ntt::kModeApprox>
                  : ctx.forward_twiddles_half != nullptr ? ntt_forward_prefetch<LOGN, LOGT, ntt::kModeHeadroomHalved>
                                                         : ntt_forward_prefetch<LOGN, LOGT, ntt::kModeHeadroom>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
    if (e != hipSuccess) return e;
    const unsigned grid = static_cast<unsigned>(rows < workgroups ? rows : workgroups);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(1u << LOGT), lds_bytes, stream, slab, ctx, mod_base, mod_period,
                       static_cast<uint32_t>(rows));
    return hipGetLastError();
}

}  // namespace heamd
