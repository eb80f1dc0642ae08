// word32_kernels.hip -- PolyRq<UInt32> on the device (the polynomial layer): the same transforms
// and element-wise operations over 4-byte words, for contexts whose moduli fit UInt32 (<= 2^30 - 1,
// ModularArithmetic/Scalar.swift:498-511) -- e.g. the n_4096_logq_27_28_28 PIR parameter sets
// (EncryptionParameters.swift:313-378).  Results are the reference's canonical words, so the 64-bit oracle pins
// them unchanged; only the storage width and the arithmetic width differ.
//
//   NTT            PolyRq+Ntt.swift:237-319, 379-483 -- one workgroup per residue row, the row lives in LDS
//                  (4 N bytes), radix-2 stages with a barrier each; Harvey butterflies in [0, 4p) < 2^32 with
//                  32-bit Shoup constants floor(w 2^32 / p): 1 v_mul_hi_u32 + 2 v_mul_lo_u32 per butterfly.
//   + - neg * *s   PolyRq.swift:147-245, 299-309
//   divideAndRoundQLast   PolyRq.swift:365-393
#include <hip/hip_runtime.h>

#include "device_context.hpp"
#include "device_math.hpp"
#include "kernels.hpp"

namespace heamd {

namespace {

constexpr unsigned kThreads = 256;

inline unsigned grid_for(size_t work_items) {
    const size_t blocks = (work_items + kThreads - 1) / kThreads;
    const size_t cap = 256 * 8;
    return static_cast<unsigned>(blocks < cap ? (blocks ? blocks : 1) : cap);
}

__device__ __forceinline__ uint32_t csub32(uint32_t x, uint32_t m) { return x >= m ? x - m : x; }
// x w mod p in [0, 2p) for any 32-bit x: Shoup with wf = floor(w 2^32 / p)
__device__ __forceinline__ uint32_t shoup32_lazy(uint32_t x, uint32_t w, uint32_t wf, uint32_t p) {
    return x * w - __umulhi(x, wf) * p;
}

// +1 word per 32: de-conflicts the power-of-two strides of the late forward / early inverse stages
__device__ __forceinline__ uint32_t slot32(uint32_t idx) { return idx + (idx >> 5); }

template <bool INVERSE>
__global__ void __launch_bounds__(1024)
    ntt32_kernel(uint32_t* __restrict__ slab, const DeviceContext32 ctx, uint32_t mod_base, uint32_t mod_period) {
    extern __shared__ uint32_t tile[];
    const uint32_t n = ctx.degree, logn = ctx.log_degree;
    const size_t row = blockIdx.x;
    const uint32_t mi = mod_base + static_cast<uint32_t>(row % mod_period);
    const DeviceModulus mod = ctx.moduli[mi];
    const uint32_t p = static_cast<uint32_t>(mod.p), two_p = 2 * p;
    const U32x2* __restrict__ tw = (INVERSE ? ctx.inverse_twiddles : ctx.forward_twiddles) + static_cast<size_t>(mi) * n;
    uint32_t* x = slab + row * n;
    for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) tile[slot32(k)] = x[k];
    __syncthreads();
    if (!INVERSE) {
        for (uint32_t s = 0; s < logn; ++s) {  // Cooley-Tukey, natural -> bit-reversed (PolyRq+Ntt.swift:271-287)
            const uint32_t t = n >> (s + 1);
            for (uint32_t k = threadIdx.x; k < (n >> 1); k += blockDim.x) {
                const uint32_t i = k >> (logn - 1 - s), o = k & (t - 1);
                const uint32_t a = 2 * i * t + o;
                const U32x2 w = tw[(1u << s) + i];
                const uint32_t xv = csub32(tile[slot32(a)], two_p);
                const uint32_t tv = shoup32_lazy(tile[slot32(a + t)], w.x, w.y, p);
                tile[slot32(a)] = xv + tv;
                tile[slot32(a + t)] = xv + two_p - tv;
            }
            __syncthreads();
        }
        for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) x[k] = csub32(csub32(tile[slot32(k)], two_p), p);
    } else {
        const uint32_t inv_n = static_cast<uint32_t>(mod.inv_degree), inv_n_f = static_cast<uint32_t>(mod.inv_degree_shoup >> 32);
        const uint32_t inv_r = static_cast<uint32_t>(mod.inv_degree_root),
                       inv_r_f = static_cast<uint32_t>(mod.inv_degree_root_shoup >> 32);
        for (uint32_t b = 0; b < logn; ++b) {  // Gentleman-Sande, bit-reversed -> natural (PolyRq+Ntt.swift:359-421)
            const uint32_t t = 1u << b, m = n >> (b + 1);
            const bool last = (b + 1 == logn);
            for (uint32_t k = threadIdx.x; k < (n >> 1); k += blockDim.x) {
                const uint32_t i = k >> b, o = k & (t - 1);
                const uint32_t a = 2 * i * t + o;
                const uint32_t xv = tile[slot32(a)], yv = tile[slot32(a + t)];
                const uint32_t sum = xv + yv, diff = xv + two_p - yv;
                if (last) {  // N^-1 and N^-1 psi^(-N/2) folded into the last stage (:407-421)
                    tile[slot32(a)] = csub32(shoup32_lazy(sum, inv_n, inv_n_f, p), p);
                    tile[slot32(a + t)] = csub32(shoup32_lazy(diff, inv_r, inv_r_f, p), p);
                } else {
                    const U32x2 w = tw[(n - 2 * m + 1) + i];
                    tile[slot32(a)] = csub32(sum, two_p);
                    tile[slot32(a + t)] = shoup32_lazy(diff, w.x, w.y, p);
                }
            }
            __syncthreads();
        }
        for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) x[k] = tile[slot32(k)];
    }
}

// ---- element-wise: one lane = one word -------------------------------------------------------------------------
template <ElementwiseOp OP>
__global__ void __launch_bounds__(kThreads)
    elementwise32_kernel(uint32_t* __restrict__ lhs, const uint32_t* __restrict__ rhs, const uint64_t* __restrict__ scalars,
                         const DeviceContext32 ctx, size_t words) {
    for (size_t i = blockIdx.x * static_cast<size_t>(kThreads) + threadIdx.x; i < words;
         i += static_cast<size_t>(gridDim.x) * kThreads) {
        const uint32_t mi = static_cast<uint32_t>((i >> ctx.log_degree) % ctx.moduli_count);
        const DeviceModulus m = ctx.moduli[mi];
        const uint32_t p = static_cast<uint32_t>(m.p);
        const uint32_t a = lhs[i];
        uint32_t r = 0;
        if constexpr (OP == ElementwiseOp::Add) r = csub32(a + rhs[i], p);
        if constexpr (OP == ElementwiseOp::Sub) r = csub32(a + p - rhs[i], p);
        if constexpr (OP == ElementwiseOp::Neg) r = csub32(p - a, p);
        if constexpr (OP == ElementwiseOp::Mul)
            r = static_cast<uint32_t>(barrett_reduce64(static_cast<uint64_t>(a) * rhs[i], m.p, m.barrett64));
        if constexpr (OP == ElementwiseOp::MulScalar) {
            const uint32_t s = static_cast<uint32_t>(scalars[2 * mi]), sf = static_cast<uint32_t>(scalars[2 * mi + 1] >> 32);
            r = csub32(shoup32_lazy(a, s, sf, p), p);
        }
        lhs[i] = r;
    }
}

// divideAndRoundQLast on 4-byte words: in [polys][L][N] -> out [polys][L-1][N]
__global__ void __launch_bounds__(kThreads)
    divide_and_round_q_last32_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                     const DeviceContext32 ctx, uint32_t moduli_count, size_t polys) {
    const uint32_t logn = ctx.log_degree;
    const size_t n = size_t(1) << logn, total = polys << logn;
    const uint32_t last = moduli_count - 1;
    const uint64_t q_last = ctx.moduli[last].p, q_last_div2 = q_last >> 1;
    const U64x2* __restrict__ inverse_q_last = ctx.inverse_q_last + static_cast<size_t>(last) * ctx.moduli_stride;
    for (size_t idx = blockIdx.x * static_cast<size_t>(kThreads) + threadIdx.x; idx < total;
         idx += static_cast<size_t>(gridDim.x) * kThreads) {
        const size_t poly = idx >> logn, k = idx & (n - 1);
        const uint32_t* src = in + poly * moduli_count * n + k;
        uint32_t* dst = out + poly * last * n + k;
        const uint64_t r = add_mod(src[size_t(last) * n], q_last_div2, q_last);  // PolyRq.swift:373-379
        for (uint32_t row = 0; row < last; ++row) {
            const DeviceModulus m = ctx.moduli[row];
            const U64x2 inv = inverse_q_last[row];
            const uint64_t half_mod_qi = barrett_reduce64(q_last_div2, m.p, m.barrett64);
            const uint64_t t = barrett_reduce64(r, m.p, m.barrett64);
            const uint64_t v = sub_mod(add_mod(src[size_t(row) * n], half_mod_qi, m.p), t, m.p);
            dst[size_t(row) * n] = static_cast<uint32_t>(shoup_mul(v, inv.x, inv.y, m.p));
        }
    }
}

}  // namespace

hipError_t launch_ntt32(bool inverse, uint32_t* slab, const DeviceContext32& ctx, uint32_t mod_base, uint32_t mod_period,
                        size_t rows, hipStream_t stream) {
    if (rows == 0 || ctx.degree < 2) return hipSuccess;
    if (ctx.log_degree > 15) return hipErrorNotSupported;  // the row must fit the LDS (4 N bytes + padding)
    constexpr size_t kMaxRowsPerLaunch = size_t(1) << 30;
    for (size_t done = 0; done < rows;) {
        const size_t chunk = (kMaxRowsPerLaunch / mod_period) * mod_period;
        const size_t now = rows - done < chunk ? rows - done : chunk;
        const uint32_t n = ctx.degree;
        const unsigned threads = n / 2 < 64 ? 64 : (n / 2 > 1024 ? 1024 : n / 2);
        const size_t lds_bytes = (static_cast<size_t>(n) + (n >> 5) + 1) * sizeof(uint32_t);
        auto kernel = inverse ? ntt32_kernel<true> : ntt32_kernel<false>;
        if (lds_bytes > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(now)), dim3(threads), lds_bytes, stream,
                           slab + done * n, ctx, mod_base, mod_period);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        done += now;
    }
    return hipSuccess;
}

hipError_t launch_elementwise32(ElementwiseOp op, uint32_t* lhs, const uint32_t* rhs, const uint64_t* scalars,
                                const DeviceContext32& ctx, size_t rows, hipStream_t stream) {
    const size_t words = rows << ctx.log_degree;
    if (words == 0) return hipSuccess;
    const dim3 grid(grid_for(words)), block(kThreads);
    switch (op) {
        case ElementwiseOp::Add:
            hipLaunchKernelGGL(elementwise32_kernel<ElementwiseOp::Add>, grid, block, 0, stream, lhs, rhs, scalars, ctx, words);
            break;
        case ElementwiseOp::Sub:
            hipLaunchKernelGGL(elementwise32_kernel<ElementwiseOp::Sub>, grid, block, 0, stream, lhs, rhs, scalars, ctx, words);
            break;
        case ElementwiseOp::Neg:
            hipLaunchKernelGGL(elementwise32_kernel<ElementwiseOp::Neg>, grid, block, 0, stream, lhs, rhs, scalars, ctx, words);
            break;
        case ElementwiseOp::Mul:
            hipLaunchKernelGGL(elementwise32_kernel<ElementwiseOp::Mul>, grid, block, 0, stream, lhs, rhs, scalars, ctx, words);
            break;
        case ElementwiseOp::MulScalar:
            hipLaunchKernelGGL(elementwise32_kernel<ElementwiseOp::MulScalar>, grid, block, 0, stream, lhs, rhs, scalars, ctx,
                               words);
            break;
    }
    return hipGetLastError();
}

hipError_t launch_divide_and_round_q_last32(const uint32_t* in, uint32_t* out, const DeviceContext32& ctx,
                                            uint32_t moduli_count, size_t polys, hipStream_t stream) {
    const size_t total = polys << ctx.log_degree;
    if (total == 0 || moduli_count < 2) return hipSuccess;
    hipLaunchKernelGGL(divide_and_round_q_last32_kernel, dim3(grid_for(total)), dim3(kThreads), 0, stream, in, out, ctx,
                       moduli_count, polys);
    return hipGetLastError();
}

}  // namespace heamd
