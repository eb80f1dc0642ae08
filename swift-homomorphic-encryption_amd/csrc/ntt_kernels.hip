// ntt_kernels.hip -- batched negacyclic NTT over the (batch x moduli x N) slab, hand-written for gfx950.
//
// Computes exactly what _NttContext.forwardNtt / inverseNtt compute (reference
// Sources/HomomorphicEncryption/PolyRq/PolyRq+Ntt.swift:237-319, 379-483): Cooley-Tukey, natural order in ->
// bit-reversed order out (forward); Gentleman-Sande, bit-reversed in -> natural out with N^-1 folded into the last
// stage (inverse); all outputs canonical.  The lazy-reduction *schedule* is ours (only canonical words are
// observable).
//
// Mapping to the machine (DESIGN.md "NTT kernel"):
//   * one workgroup per residue row; a row of N = 2^LOGN words is held entirely in registers, E = N / T words per
//     lane (T = 2^LOGT lanes), and is touched in HBM exactly once in and once out;
//   * log2(N) radix-2 stages are grouped into passes of LOGE stages executed on registers; between passes the row
//     is transposed through a padded LDS tile (2 transposes for N = 8192: 13 = 5 + 5 + 3);
//   * pass 0 twiddles are wave-uniform (scalar loads); later passes gather (w, w') pairs from the L2-resident
//     per-modulus table (128 KiB per modulus per direction at N = 8192, shared by every workgroup of that modulus);
//   * butterflies are Harvey's with Shoup constants; with every modulus < 2^61 the quotient estimate uses 3 instead
//     of 4 32-bit multiplies and values live in [0, 8p).
#include <hip/hip_runtime.h>

#include "device_context.hpp"
#include "device_math.hpp"
#include "kernels.hpp"

namespace heamd {

namespace {

// ---- LDS tile addressing: word index -> padded word index (all accesses are 8-byte ds_read/write_b64) ----------
// +1 word per 8 words de-conflicts the stride-8/16 reads of the last pass; +8 words per 256 de-conflicts the
// middle pass whose lanes are 256 words apart (bank math in DESIGN.md).
__device__ __forceinline__ uint32_t lds_slot(uint32_t idx) { return idx + (idx >> 3) + ((idx >> 8) << 3); }
constexpr uint32_t lds_words(uint32_t n) { return n + (n >> 3) + ((n >> 8) << 3) + 8; }

// element index held in register r of lane tid during a pass over element bits [LO, LO + W)
template <int LOGN, int LOGE, int LO, int W>
__device__ __forceinline__ uint32_t element_index(uint32_t r, uint32_t tid) {
    if constexpr (W == LOGE) {
        return ((tid >> LO) << (LO + LOGE)) | (r << LO) | (tid & ((1u << LO) - 1u));
    } else {
        static_assert(LO == 0, "a partial pass sits on the low bits");
        constexpr int X = LOGE - W;  // extra register bits = the top X element bits
        return ((r >> W) << (LOGN - X)) | (tid << W) | (r & ((1u << W) - 1u));
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
template <int LOGN, int LOGE, int LO, int W, bool APPROX, bool UNIFORM_TWIDDLES>
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
            const uint32_t idx = element_index<LOGN, LOGE, LO, W>(base, tid);
            const U64x2 w = tw[(1u << s) + (idx >> (b + 1))];
#pragma unroll
            for (int o = 0; o < stride; ++o) {
                uint64_t x = v[base + o];
                const uint64_t y = v[base + o + stride];
                if (!(first_stage_canonical && j == 0)) x = csub(x, half_bound);
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
            const uint32_t idx = element_index<LOGN, LOGE, LO, W>(base, tid);
            U64x2 w = {0, 0};
            if (!last_stage) w = tw[(N - 2 * m + 1) + (idx >> (b + 1))];
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
#pragma unroll
    for (int r = 0; r < (1 << LOGE); ++r) lds[lds_slot(element_index<LOGN, LOGE, LO, W>(r, tid))] = v[r];
}
template <int LOGN, int LOGE, int LO, int W>
__device__ __forceinline__ void lds_load(uint64_t (&v)[1 << LOGE], uint32_t tid, const uint64_t* lds) {
#pragma unroll
    for (int r = 0; r < (1 << LOGE); ++r) v[r] = lds[lds_slot(element_index<LOGN, LOGE, LO, W>(r, tid))];
}

// Global <-> registers.  For a pass on the low bits each lane owns runs of 2^W contiguous words: move them 16 B at
// a time.  For the top pass consecutive lanes own consecutive words (8 B each, 512 B per wave instruction).
template <int LOGN, int LOGE, int LO, int W>
__device__ __forceinline__ void global_load(uint64_t (&v)[1 << LOGE], uint32_t tid, const uint64_t* __restrict__ x) {
    if constexpr (LO == 0 && W >= 1) {
#pragma unroll
        for (int r = 0; r < (1 << LOGE); r += 2) {
            const U64x2 pair = *reinterpret_cast<const U64x2*>(x + element_index<LOGN, LOGE, LO, W>(r, tid));
            v[r] = pair.x;
            v[r + 1] = pair.y;
        }
    } else {
#pragma unroll
        for (int r = 0; r < (1 << LOGE); ++r) v[r] = x[element_index<LOGN, LOGE, LO, W>(r, tid)];
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
            *reinterpret_cast<U64x2*>(x + element_index<LOGN, LOGE, LO, W>(r, tid)) = pair;
        }
    } else {
#pragma unroll
        for (int r = 0; r < (1 << LOGE); ++r) x[element_index<LOGN, LOGE, LO, W>(r, tid)] = v[r];
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

template <int LOGN, int LOGT, bool APPROX>
__global__ void __launch_bounds__(1 << LOGT, (LOGT >= 9 ? 4 : 2))
    ntt_forward_tiled(uint64_t* __restrict__ slab, const DeviceContext ctx, uint32_t mod_base, uint32_t mod_period) {
    constexpr int LOGE = LOGN - LOGT;
    constexpr int E = 1 << LOGE;
    using S = Schedule<LOGN, LOGE>;
    static_assert(S::P >= 1 && S::P <= 4, "unsupported pass count");
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
    const uint32_t tid = threadIdx.x;
    const size_t row = blockIdx.x;
    const uint32_t mi = mod_base + static_cast<uint32_t>(row % mod_period);
    const DeviceModulus mod = ctx.moduli[mi];
    const U64x2* __restrict__ tw = ctx.forward_twiddles + (static_cast<size_t>(mi) << LOGN);
    uint64_t* __restrict__ x = slab + (row << LOGN);
    const uint64_t p = mod.p;
    uint64_t v[E];

    if constexpr (S::P == 1) {
        global_load<LOGN, LOGE, 0, LOGN>(v, tid, x);
        forward_pass<LOGN, LOGE, 0, LOGN, APPROX, true>(v, tid, tw, p, true);
#pragma unroll
        for (int r = 0; r < E; ++r) v[r] = canonicalize<APPROX>(v[r], p);
        global_store<LOGN, LOGE, 0, LOGN>(v, tid, x);
    } else {
        constexpr int LO0 = LOGN - LOGE;
        global_load<LOGN, LOGE, LO0, LOGE>(v, tid, x);
        forward_pass<LOGN, LOGE, LO0, LOGE, APPROX, true>(v, tid, tw, p, true);
        lds_store<LOGN, LOGE, LO0, LOGE>(v, tid, lds);
        __syncthreads();
        if constexpr (S::P >= 3) {
            constexpr int LO1 = LOGN - 2 * LOGE;
            lds_load<LOGN, LOGE, LO1, LOGE>(v, tid, lds);
            forward_pass<LOGN, LOGE, LO1, LOGE, APPROX, false>(v, tid, tw, p, false);
            lds_store<LOGN, LOGE, LO1, LOGE>(v, tid, lds);
            __syncthreads();
        }
        if constexpr (S::P >= 4) {
            constexpr int LO2 = LOGN - 3 * LOGE;
            lds_load<LOGN, LOGE, LO2, LOGE>(v, tid, lds);
            forward_pass<LOGN, LOGE, LO2, LOGE, APPROX, false>(v, tid, tw, p, false);
            lds_store<LOGN, LOGE, LO2, LOGE>(v, tid, lds);
            __syncthreads();
        }
        lds_load<LOGN, LOGE, 0, S::R>(v, tid, lds);
        forward_pass<LOGN, LOGE, 0, S::R, APPROX, false>(v, tid, tw, p, false);
#pragma unroll
        for (int r = 0; r < E; ++r) v[r] = canonicalize<APPROX>(v[r], p);
        global_store<LOGN, LOGE, 0, S::R>(v, tid, x);
    }
}

template <int LOGN, int LOGT, bool APPROX>
__global__ void __launch_bounds__(1 << LOGT, (LOGT >= 9 ? 4 : 2))
    ntt_inverse_tiled(uint64_t* __restrict__ slab, const DeviceContext ctx, uint32_t mod_base, uint32_t mod_period) {
    constexpr int LOGE = LOGN - LOGT;
    constexpr int E = 1 << LOGE;
    using S = Schedule<LOGN, LOGE>;
    static_assert(S::P >= 1 && S::P <= 4, "unsupported pass count");
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
    const uint32_t tid = threadIdx.x;
    const size_t row = blockIdx.x;
    const uint32_t mi = mod_base + static_cast<uint32_t>(row % mod_period);
    const DeviceModulus mod = ctx.moduli[mi];
    const U64x2* __restrict__ tw = ctx.inverse_twiddles + (static_cast<size_t>(mi) << LOGN);
    uint64_t* __restrict__ x = slab + (row << LOGN);
    uint64_t v[E];

    if constexpr (S::P == 1) {
        global_load<LOGN, LOGE, 0, LOGN>(v, tid, x);
        inverse_pass<LOGN, LOGE, 0, LOGN, APPROX>(v, tid, tw, mod, true);
        global_store<LOGN, LOGE, 0, LOGN>(v, tid, x);
    } else {
        global_load<LOGN, LOGE, 0, S::R>(v, tid, x);
        inverse_pass<LOGN, LOGE, 0, S::R, APPROX>(v, tid, tw, mod, true);
        lds_store<LOGN, LOGE, 0, S::R>(v, tid, lds);
        __syncthreads();
        if constexpr (S::P >= 3) {
            constexpr int LO1 = S::R;
            lds_load<LOGN, LOGE, LO1, LOGE>(v, tid, lds);
            inverse_pass<LOGN, LOGE, LO1, LOGE, APPROX>(v, tid, tw, mod, false);
            lds_store<LOGN, LOGE, LO1, LOGE>(v, tid, lds);
            __syncthreads();
        }
        if constexpr (S::P >= 4) {
            constexpr int LO2 = S::R + LOGE;
            lds_load<LOGN, LOGE, LO2, LOGE>(v, tid, lds);
            inverse_pass<LOGN, LOGE, LO2, LOGE, APPROX>(v, tid, tw, mod, false);
            lds_store<LOGN, LOGE, LO2, LOGE>(v, tid, lds);
            __syncthreads();
        }
        constexpr int LOL = LOGN - LOGE;
        lds_load<LOGN, LOGE, LOL, LOGE>(v, tid, lds);
        inverse_pass<LOGN, LOGE, LOL, LOGE, APPROX>(v, tid, tw, mod, false);
        global_store<LOGN, LOGE, LOL, LOGE>(v, tid, x);
    }
}

// ---- any power-of-two degree: one workgroup per row, radix-2 stage loop over an LDS (or, for rows that do not
// fit, global-memory) buffer.  Exact Harvey butterflies in [0, 4p): valid for every modulus <= 2^62 - 1. ---------
template <bool USE_LDS>
__global__ void __launch_bounds__(256)
    ntt_forward_generic(uint64_t* __restrict__ slab, const DeviceContext ctx, uint32_t mod_base, uint32_t mod_period) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
    const uint32_t n = ctx.degree, logn = ctx.log_degree;
    const size_t row = blockIdx.x;
    const uint32_t mi = mod_base + static_cast<uint32_t>(row % mod_period);
    const DeviceModulus mod = ctx.moduli[mi];
    const U64x2* __restrict__ tw = ctx.forward_twiddles + static_cast<size_t>(mi) * n;
    uint64_t* x = slab + row * n;
    uint64_t* buf = USE_LDS ? lds : x;
    const uint64_t p = mod.p, two_p = 2 * p, neg_p = 0 - p;
    if (USE_LDS) {
        for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) buf[k] = x[k];
        __syncthreads();
    }
    for (uint32_t s = 0; s < logn; ++s) {
        const uint32_t t = n >> (s + 1);
        const bool last = (s + 1 == logn);
        for (uint32_t k = threadIdx.x; k < (n >> 1); k += blockDim.x) {
            const uint32_t i = k >> (logn - 1 - s), o = k & (t - 1);
            const uint32_t a = 2 * i * t + o;
            const U64x2 w = tw[(1u << s) + i];
            uint64_t xv = csub(buf[a], two_p);
            const uint64_t tv = shoup_lazy(buf[a + t], w.x, w.y, neg_p);
            uint64_t xo = xv + tv, yo = xv + two_p - tv;
            if (last) {
                xo = canonicalize<false>(xo, p);
                yo = canonicalize<false>(yo, p);
            }
            buf[a] = xo;
            buf[a + t] = yo;
        }
        __syncthreads();
    }
    if (USE_LDS) {
        for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) x[k] = buf[k];
    }
}

template <bool USE_LDS>
__global__ void __launch_bounds__(256)
    ntt_inverse_generic(uint64_t* __restrict__ slab, const DeviceContext ctx, uint32_t mod_base, uint32_t mod_period) {
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
    const uint32_t n = ctx.degree, logn = ctx.log_degree;
    const size_t row = blockIdx.x;
    const uint32_t mi = mod_base + static_cast<uint32_t>(row % mod_period);
    const DeviceModulus mod = ctx.moduli[mi];
    const U64x2* __restrict__ tw = ctx.inverse_twiddles + static_cast<size_t>(mi) * n;
    uint64_t* x = slab + row * n;
    uint64_t* buf = USE_LDS ? lds : x;
    const uint64_t p = mod.p, two_p = 2 * p, neg_p = 0 - p;
    if (USE_LDS) {
        for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) buf[k] = x[k];
        __syncthreads();
    }
    for (uint32_t b = 0; b < logn; ++b) {
        const uint32_t t = 1u << b, m = n >> (b + 1);
        const bool last = (b + 1 == logn);
        for (uint32_t k = threadIdx.x; k < (n >> 1); k += blockDim.x) {
            const uint32_t i = k >> b, o = k & (t - 1);
            const uint32_t a = 2 * i * t + o;
            const uint64_t xv = buf[a], yv = buf[a + t];
            const uint64_t sum = xv + yv, diff = xv + two_p - yv;
            if (last) {
                buf[a] = shoup_mul(sum, mod.inv_degree, mod.inv_degree_shoup, p);
                buf[a + t] = shoup_mul(diff, mod.inv_degree_root, mod.inv_degree_root_shoup, p);
            } else {
                const U64x2 w = tw[(n - 2 * m + 1) + i];
                buf[a] = csub(sum, two_p);
                buf[a + t] = shoup_lazy(diff, w.x, w.y, neg_p);
            }
        }
        __syncthreads();
    }
    if (USE_LDS) {
        for (uint32_t k = threadIdx.x; k < n; k += blockDim.x) x[k] = buf[k];
    }
}

template <int LOGN, int LOGT>
hipError_t launch_tiled(bool inverse, bool approx, uint64_t* slab, const DeviceContext& ctx, uint32_t mod_base,
                        uint32_t mod_period, size_t rows, hipStream_t stream) {
    constexpr int LOGE = LOGN - LOGT;
    constexpr size_t lds_bytes = (Schedule<LOGN, LOGE>::P > 1) ? lds_words(1u << LOGN) * sizeof(uint64_t) : 0;
    using Kernel = void (*)(uint64_t*, const DeviceContext, uint32_t, uint32_t);
    Kernel kernel;
    if (inverse) {
        kernel = approx ? ntt_inverse_tiled<LOGN, LOGT, true> : ntt_inverse_tiled<LOGN, LOGT, false>;
    } else {
        kernel = approx ? ntt_forward_tiled<LOGN, LOGT, true> : ntt_forward_tiled<LOGN, LOGT, false>;
    }
    if (lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(rows)), dim3(1u << LOGT), lds_bytes, stream, slab, ctx,
                       mod_base, mod_period);
    return hipGetLastError();
}

}  // namespace

const char* ntt_variant_name(uint32_t log_degree) {
    switch (log_degree) {
        case 12: return "tiled<4096,256thr,16/lane>";
        case 13: return "tiled<8192,256thr,32/lane>";
        case 14: return "tiled<16384,512thr,32/lane>";
        default: return "generic radix-2";
    }
}

hipError_t launch_ntt(bool inverse, uint64_t* slab, const DeviceContext& ctx, uint32_t mod_base, uint32_t mod_period,
                      size_t rows, hipStream_t stream, int force_variant) {
    if (rows == 0) return hipSuccess;
    // grid.x is limited to 2^31-1 workgroups; split very large batches
    constexpr size_t kMaxRowsPerLaunch = size_t(1) << 30;
    if (rows > kMaxRowsPerLaunch) {
        // keep the row -> modulus mapping intact: chunk must be a multiple of mod_period
        const size_t chunk = (kMaxRowsPerLaunch / mod_period) * mod_period;
        for (size_t done = 0; done < rows; done += chunk) {
            const size_t now = rows - done < chunk ? rows - done : chunk;
            hipError_t e = launch_ntt(inverse, slab + done * ctx.degree, ctx, mod_base, mod_period, now, stream,
                                      force_variant);
            if (e != hipSuccess) return e;
        }
        return hipSuccess;
    }
    const bool approx = ctx.approx_ok != 0 && force_variant != kNttVariantExact && force_variant != kNttVariantGeneric;
    if (force_variant == kNttVariantWide) {
        switch (ctx.log_degree) {
            case 12: return launch_tiled<12, 9>(inverse, approx, slab, ctx, mod_base, mod_period, rows, stream);
            case 13: return launch_tiled<13, 9>(inverse, approx, slab, ctx, mod_base, mod_period, rows, stream);
            case 14: return launch_tiled<14, 10>(inverse, approx, slab, ctx, mod_base, mod_period, rows, stream);
            default: break;
        }
    }
    if (force_variant != kNttVariantGeneric) {
        switch (ctx.log_degree) {
            case 12: return launch_tiled<12, 8>(inverse, approx, slab, ctx, mod_base, mod_period, rows, stream);
            case 13: return launch_tiled<13, 8>(inverse, approx, slab, ctx, mod_base, mod_period, rows, stream);
            case 14: return launch_tiled<14, 9>(inverse, approx, slab, ctx, mod_base, mod_period, rows, stream);
            default: break;
        }
    }
    const uint32_t n = ctx.degree;
    if (n < 2) return hipSuccess;  // degree 1: no stages (the reference loops over zero stages)
    const bool use_lds = n <= 4096;
    const unsigned threads = n / 2 < 256 ? (n / 2 < 64 ? 64 : n / 2) : 256;
    const size_t lds_bytes = use_lds ? n * sizeof(uint64_t) : 0;
    using Kernel = void (*)(uint64_t*, const DeviceContext, uint32_t, uint32_t);
    Kernel kernel = inverse ? (use_lds ? ntt_inverse_generic<true> : ntt_inverse_generic<false>)
                            : (use_lds ? ntt_forward_generic<true> : ntt_forward_generic<false>);
    hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(rows)), dim3(threads), lds_bytes, stream, slab, ctx,
                       mod_base, mod_period);
    return hipGetLastError();
}

}  // namespace heamd
