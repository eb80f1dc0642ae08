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
#include "ntt_common.hpp"
#include "placement.hpp"

namespace heamd {

namespace {

constexpr unsigned kThreads = 256;

constexpr size_t kGridCap = (size_t(1) << 31) - 1;
// One workgroup per kThreads work items, up to the grid limit: the kernels keep their grid-stride loops for what lies beyond it,
// but a lane that walks many items serialises its loads -- divideAndRoundQLast at N = 16384, L = 6 ran at 0.66 of 8 TB/s on
// 256 x 8 workgroups and at 0.79 with one item per lane (profiles/r06y_exact_grids.txt)
inline unsigned grid_for(size_t work_items) {
    const size_t blocks = (work_items + kThreads - 1) / kThreads;
    const size_t cap = kGridCap;
    return static_cast<unsigned>(blocks < cap ? (blocks ? blocks : 1) : cap);
}

__device__ __forceinline__ uint32_t csub32(uint32_t x, uint32_t m) { return x >= m ? x - m : x; }
// x w mod p in [0, 2p) for any 32-bit x: Shoup with wf = floor(w 2^32 / p)
__device__ __forceinline__ uint32_t shoup32_lazy(uint32_t x, uint32_t w, uint32_t wf, uint32_t p) {
    return x * w - __umulhi(x, wf) * p;
}
// The same product on the full-width multiplier for the register passes: q = high word of x wf, then the low word of
// x w + q (2^32 - p) -- three v_mad_u64_u32 (two issue slots each) against v_mul_hi_u32 (three to four), two
// v_mul_lo_u32 (two each) and a subtract.  UNIFORM: the twiddle is wave-uniform and read from SGPRs (one scalar operand
// per instruction); neg_p = 2^32 - p always is.
template <bool UNIFORM>
__device__ __forceinline__ uint32_t shoup32_lazy_mad(uint32_t x, uint32_t w, uint32_t wf, uint32_t neg_p) {
    uint64_t q, t, carry;
    if constexpr (UNIFORM) {
        asm("v_mad_u64_u32 %0, %2, %3, %4, 0\n\t"
            "v_mad_u64_u32 %1, %2, %3, %5, 0"
            : "=&v"(q), "=&v"(t), "=&s"(carry)
            : "v"(x), "s"(wf), "s"(w));
    } else {
        asm("v_mad_u64_u32 %0, %2, %3, %4, 0\n\t"
            "v_mad_u64_u32 %1, %2, %3, %5, 0"
            : "=&v"(q), "=&v"(t), "=&s"(carry)
            : "v"(x), "v"(wf), "v"(w));
    }
    asm("v_mad_u64_u32 %0, %1, %2, %3, %0" : "+v"(t), "=&s"(carry) : "v"(static_cast<uint32_t>(q >> 32)), "s"(neg_p));
    return static_cast<uint32_t>(t);
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

// ---- register-tiled transform for N = 4096 / 8192 / 16384: the 8-byte kernel's structure (ntt_kernels.hip) on 4-byte
// words -- one workgroup per row, 2^LOGE words per lane, passes of LOGE radix-2 stages on registers, LDS transposes
// in between (wave-private after the first), wave-uniform twiddles through the scalar cache.  The lane <-> element
// maps, the pass schedule and the fences are the shared helpers of ntt_common.hpp; only the arithmetic is narrower.
using ntt::element_index;
using ntt::lane_part;
using ntt::register_part;
using ntt::Schedule;

__device__ __forceinline__ U32x2 load_twiddle32(const U32x2* entry) {
    using ConstWord = const __attribute__((address_space(4))) uint32_t;
    ConstWord* const words = (ConstWord*)(entry);
    return U32x2{words[0], words[1]};
}

template <int LOGN, int LOGE, int LO, int W>
__device__ __forceinline__ void forward_pass32(uint32_t (&v)[1 << LOGE], uint32_t tid, const U32x2* __restrict__ tw,
                                               uint32_t p, bool first_stage_canonical) {
    constexpr int E = 1 << LOGE;
    const uint32_t two_p = 2 * p, neg_p = 0u - p;
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const int b = LO + W - 1 - j, s = LOGN - 1 - b, stride = 1 << (b - LO);
        const bool uniform = (element_index<LOGN, LOGE, LO, W>(0, 63u) >> (b + 1)) == 0;
        uint32_t lane_twiddle = lane_part<LOGN, LOGE, LO, W>(tid) >> (b + 1);
        if (uniform) lane_twiddle = __builtin_amdgcn_readfirstlane(lane_twiddle);
        const U32x2* const tw_stage = tw + (1u << s) + lane_twiddle;
#pragma unroll
        for (int base = 0; base < E; base += 2 * stride) {
            const U32x2 w = load_twiddle32(tw_stage + (register_part<LOGN, LOGE, LO, W>(base) >> (b + 1)));
#pragma unroll
            for (int o = 0; o < stride; ++o) {
                uint32_t x = v[base + o];
                if (!(first_stage_canonical && j == 0)) x = csub32(x, two_p);
                const uint32_t t = uniform ? shoup32_lazy_mad<true>(v[base + o + stride], w.x, w.y, neg_p)
                                           : shoup32_lazy_mad<false>(v[base + o + stride], w.x, w.y, neg_p);
                v[base + o] = x + t;
                v[base + o + stride] = x + two_p - t;
            }
        }
    }
}

template <int LOGN, int LOGE, int LO, int W>
__device__ __forceinline__ void inverse_pass32(uint32_t (&v)[1 << LOGE], uint32_t tid, const U32x2* __restrict__ tw,
                                               const DeviceModulus& mod, bool first_stage_canonical) {
    constexpr int E = 1 << LOGE;
    constexpr uint32_t N = 1u << LOGN;
    const uint32_t p = static_cast<uint32_t>(mod.p), two_p = 2 * p, neg_p = 0u - p;
#pragma unroll
    for (int j = 0; j < W; ++j) {
        const int b = LO + j, stride = 1 << (b - LO);
        const uint32_t m = N >> (b + 1);
        const bool last_stage = (b == LOGN - 1);
        const bool uniform = (element_index<LOGN, LOGE, LO, W>(0, 63u) >> (b + 1)) == 0;
        uint32_t lane_twiddle = lane_part<LOGN, LOGE, LO, W>(tid) >> (b + 1);
        if (uniform) lane_twiddle = __builtin_amdgcn_readfirstlane(lane_twiddle);
        const U32x2* const tw_stage = tw + (N - 2 * m + 1) + lane_twiddle;
#pragma unroll
        for (int base = 0; base < E; base += 2 * stride) {
            U32x2 w = {0, 0};
            if (!last_stage) w = load_twiddle32(tw_stage + (register_part<LOGN, LOGE, LO, W>(base) >> (b + 1)));
#pragma unroll
            for (int o = 0; o < stride; ++o) {
                const uint32_t x = v[base + o], y = v[base + o + stride];
                // inputs in [0, 2p) (canonical on the very first stage): sum < 4p, diff in (0, 4p)
                const uint32_t sum = x + y, diff = x + two_p - y;
                if (last_stage) {
                    v[base + o] = csub32(shoup32_lazy_mad<true>(sum, static_cast<uint32_t>(mod.inv_degree),
                                                                static_cast<uint32_t>(mod.inv_degree_shoup >> 32), neg_p), p);
                    v[base + o + stride] = csub32(shoup32_lazy_mad<true>(diff, static_cast<uint32_t>(mod.inv_degree_root),
                                                                         static_cast<uint32_t>(mod.inv_degree_root_shoup >> 32), neg_p), p);
                } else {
                    v[base + o] = (first_stage_canonical && j == 0) ? sum : csub32(sum, two_p);
                    v[base + o + stride] = uniform ? shoup32_lazy_mad<true>(diff, w.x, w.y, neg_p)
                                                   : shoup32_lazy_mad<false>(diff, w.x, w.y, neg_p);
                }
            }
        }
    }
}

__host__ __device__ constexpr uint32_t tile32_slot(uint32_t idx) { return idx + (idx >> 5); }
constexpr uint32_t tile32_words(uint32_t n) { return n + (n >> 5) + 8; }

template <int LOGN, int LOGE, int LO, int W>
__device__ __forceinline__ void tile32_store(const uint32_t (&v)[1 << LOGE], uint32_t tid, uint32_t* lds) {
    uint32_t* const base = lds + tile32_slot(lane_part<LOGN, LOGE, LO, W>(tid));
#pragma unroll
    for (int r = 0; r < (1 << LOGE); ++r) base[tile32_slot(register_part<LOGN, LOGE, LO, W>(r))] = v[r];
}
template <int LOGN, int LOGE, int LO, int W>
__device__ __forceinline__ void tile32_load(uint32_t (&v)[1 << LOGE], uint32_t tid, const uint32_t* lds) {
    const uint32_t* const base = lds + tile32_slot(lane_part<LOGN, LOGE, LO, W>(tid));
#pragma unroll
    for (int r = 0; r < (1 << LOGE); ++r) v[r] = base[tile32_slot(register_part<LOGN, LOGE, LO, W>(r))];
}
template <int LOGN, int LOGE, int LO, int W>
__device__ __forceinline__ void row32_load(uint32_t (&v)[1 << LOGE], uint32_t tid, const uint32_t* __restrict__ x) {
#pragma unroll
    for (int r = 0; r < (1 << LOGE); ++r)
        v[r] = x[register_part<LOGN, LOGE, LO, W>(r) + lane_part<LOGN, LOGE, LO, W>(tid)];
}
template <int LOGN, int LOGE, int LO, int W>
__device__ __forceinline__ void row32_store(const uint32_t (&v)[1 << LOGE], uint32_t tid, uint32_t* __restrict__ x) {
#pragma unroll
    for (int r = 0; r < (1 << LOGE); ++r)
        x[register_part<LOGN, LOGE, LO, W>(r) + lane_part<LOGN, LOGE, LO, W>(tid)] = v[r];
}

// Rows of the partial pass (bits [0, W)) in whole cache lines, as ntt_common.hpp global_store_staged / global_load_staged
// do for 8-byte words: a lane's run of 2^W words is 2^(W-2) 16-byte chunks; with more than one of them per lane a store
// as the words lie fills half of 64 lines per instruction.  The wave's block goes through its own slice of the tile and
// crosses the memory interface in lane order.
template <int LOGN, int LOGE, int W>
constexpr bool kStaged32 = W == 3 && W == LOGE && ntt::kWaveOwnsTopBits<LOGN, LOGE, 0>;

template <int LOGN, int LOGE, int W>
__device__ __forceinline__ void row32_store_staged(const uint32_t (&v)[1 << LOGE], uint32_t tid, uint32_t* __restrict__ x,
                                                   uint32_t* tile) {
    constexpr int C = 1 << (W - 2);
    const uint32_t lane = tid & 63u, swizzle = (lane >> (4 - (W - 2))) & (C - 1);
    const uint32_t first = lane_part<LOGN, LOGE, 0, W>(tid & ~63u);
    char* const block = reinterpret_cast<char*>(tile + tile32_slot(first));
#pragma unroll
    for (int j = 0; j < C; ++j)
        *reinterpret_cast<ntt::Dwordx4*>(block + ((lane * C + (j ^ swizzle)) << 4)) =
            ntt::Dwordx4{v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]};
#pragma unroll
    for (int j = 0; j < C; ++j) {
        const uint32_t chunk = j * 64 + lane, owner = chunk / C, slot = chunk % C;
        const uint32_t owner_swizzle = (owner >> (4 - (W - 2))) & (C - 1);
        *reinterpret_cast<ntt::Dwordx4*>(x + first + 4 * chunk) =
            *reinterpret_cast<const ntt::Dwordx4*>(block + ((owner * C + (slot ^ owner_swizzle)) << 4));
    }
}
template <int LOGN, int LOGE, int W>
__device__ __forceinline__ void row32_load_staged(uint32_t (&v)[1 << LOGE], uint32_t tid, const uint32_t* __restrict__ x,
                                                  uint32_t* tile) {
    constexpr int C = 1 << (W - 2);
    const uint32_t lane = tid & 63u, swizzle = (lane >> (4 - (W - 2))) & (C - 1);
    const uint32_t first = lane_part<LOGN, LOGE, 0, W>(tid & ~63u);
    char* const block = reinterpret_cast<char*>(tile + tile32_slot(first));
    ntt::Dwordx4 words[C];
#pragma unroll
    for (int j = 0; j < C; ++j) words[j] = *reinterpret_cast<const ntt::Dwordx4*>(x + first + 4 * (j * 64 + lane));
#pragma unroll
    for (int j = 0; j < C; ++j) {
        const uint32_t chunk = j * 64 + lane, owner = chunk / C, slot = chunk % C;
        const uint32_t owner_swizzle = (owner >> (4 - (W - 2))) & (C - 1);
        *reinterpret_cast<ntt::Dwordx4*>(block + ((owner * C + (slot ^ owner_swizzle)) << 4)) = words[j];
    }
#pragma unroll
    for (int j = 0; j < C; ++j) {
        const ntt::Dwordx4 mine = *reinterpret_cast<const ntt::Dwordx4*>(block + ((lane * C + (j ^ swizzle)) << 4));
        v[4 * j] = mine.x;
        v[4 * j + 1] = mine.y;
        v[4 * j + 2] = mine.z;
        v[4 * j + 3] = mine.w;
    }
}

// Row sources other than the slab itself, as in ntt_kernels.hip: the step that would otherwise write the slab for this
// kernel to read back is applied to the words while they are loaded.  All of them take mod_period rows per record with
// modulus mod_base + (row within the record); the workgroups that read the same source words form one replica set on one
// XCD (placement.hpp).
//   kSource32Spread  forward: row (poly, j, r) of a [polys][L][L+1][N] slab is the transform mod ks_modulus[r] of row j of
//                    polynomial `poly` at first + poly * stride + j * N, reduced mod r first when q_j > modulus r
//                    (Bfv+Keys.swift:165-179)
//   kSource32Tensor  inverse: row r of record (item, c), c in {0, 1, 2}, is a0 b0 | a0 b1 + a1 b0 | a1 b1 of row r of the
//                    four Eval polynomials of the item at first + (item * 4 + k) * mod_period * N
//                    (Bfv+Multiply.swift:80-82)
//   kSource32KeyMac  inverse: row r of record (poly, c), c in {0, 1}, is sum_j spread[poly][j][r] key[j][c][key_row(r)]
//                    mod ks_modulus[r] (Bfv+Keys.swift:180-202): products below 2^60, at most 15 of them in a 64-bit word
//   kSource32Rows    forward: rows [0, L) of record item * 4 + slot of a [records][mod_period][N] slab are rows of the
//                    source ciphertexts (liftQToQBsk leaves them equal to its input, RnsTool.swift:329-330): polynomial
//                    slot & 1 of item `item` of `first` (slots 0, 1) or `second` (slots 2, 3), items `stride` words
//                    apart; the other rows are read from the slab
//   kSource32KeyMacFinish  the same load over the band rows r < L, with the key switch's last step (drop the special
//                    modulus, add the update to the ciphertext: key_switch_finish_kernel, rns_kernels.hip) applied to the
//                    transform's canonical words before they are stored -- the 4-byte twin of ntt_kernels.hip
//                    kInverseFromKeyMacFinish: the q_ks rows come from an earlier launch of kSource32KeyMac over that band
constexpr int kSource32Slab = 0, kSource32Spread = 1, kSource32Tensor = 2, kSource32KeyMac = 3, kSource32Rows = 4,
              kSource32KeyMacFinish = 5;
constexpr bool is_key_mac32(int source) { return source == kSource32KeyMac || source == kSource32KeyMacFinish; }
struct Source32 {
    const uint32_t* first;
    const uint32_t* second;  // key MAC: the key
    size_t stride;           // spread: words between source polynomials
    uint32_t L, top_rows;    // spread / key MAC: source moduli; key MAC: rows per key polynomial
    // key MAC: the launch covers the band [band_offset, band_offset + band_rows) of every record's rows (band_rows = 0: all)
    uint32_t band_offset, band_rows;
    // kSource32KeyMacFinish: the ciphertexts the update is added to ([item] ct_stride words apart, polynomial c at c L N; the
    // first `added_polys` polynomials are added, the others replaced) and where the result goes ([item][2][L][N]); the
    // kernel's slab is the product slab whose q_ks rows are read
    const uint32_t* ct_base;
    size_t ct_stride;
    uint32_t* out;
    uint32_t added_polys;
};

template <int LOGN, int LOGT, bool INVERSE, int SOURCE = kSource32Slab>
__global__ void __launch_bounds__(1 << LOGT)
    ntt32_tiled_kernel(uint32_t* __restrict__ slab, const DeviceContext32 ctx, uint32_t mod_base, uint32_t mod_period,
                       const Source32 source) {
    constexpr int LOGE = LOGN - LOGT;
    constexpr int E = 1 << LOGE;
    using S = Schedule<LOGN, LOGE>;
    static_assert(S::P >= 2 && S::P <= 5, "unsupported pass count");
    static_assert(SOURCE == kSource32Slab || (SOURCE == kSource32Spread || SOURCE == kSource32Rows) == !INVERSE,
                  "spread and source rows feed the forward transform");
    extern __shared__ __attribute__((aligned(16))) uint32_t tile[];
    const uint32_t tid = threadIdx.x;
    size_t row = blockIdx.x;
    uint32_t within = 0, set = 0, replica = 0, group = 0;  // fused sources: row within the record, replica set, member, record group
    if constexpr (SOURCE == kSource32Slab || SOURCE == kSource32Rows) {
        set = static_cast<uint32_t>(row / mod_period);  // the record
        within = static_cast<uint32_t>(row - size_t(set) * mod_period);
    } else if constexpr (SOURCE == kSource32Spread) {
        locate_replica(blockIdx.x, gridDim.x / mod_period, mod_period, set, within);  // set = poly * L + j
        row = size_t(set) * mod_period + within;
    } else {
        constexpr uint32_t REPLICAS = SOURCE == kSource32Tensor ? 3 : 2;
        locate_replica(blockIdx.x, gridDim.x / REPLICAS, REPLICAS, set, replica);     // set = record group * band + r
        const uint32_t band = is_key_mac32(SOURCE) && source.band_rows != 0 ? source.band_rows : mod_period;
        group = set / band;
        within = (is_key_mac32(SOURCE) ? source.band_offset : 0u) + (set - group * band);
        row = (size_t(group) * REPLICAS + replica) * mod_period + within;
    }
    const uint32_t mi = mod_base + within;
    const DeviceModulus mod = ctx.moduli[mi];
    const uint32_t p = static_cast<uint32_t>(mod.p);
    const U32x2* __restrict__ tw = (INVERSE ? ctx.inverse_twiddles : ctx.forward_twiddles) + (static_cast<size_t>(mi) << LOGN);
    uint32_t* __restrict__ x = slab + (row << LOGN);
    uint32_t v[E];
    if constexpr (!INVERSE) {
        constexpr int LO0 = LOGN - LOGE;
        if constexpr (SOURCE == kSource32Spread) {
            const uint32_t poly = set / source.L, j = set - poly * source.L;
            row32_load<LOGN, LOGE, LO0, LOGE>(v, tid, source.first + size_t(poly) * source.stride + (size_t(j) << LOGN));
            if (ctx.moduli[j].p > mod.p) {  // wave-uniform
#pragma unroll
                for (int r = 0; r < E; ++r) v[r] = static_cast<uint32_t>(barrett_reduce64_uniform(v[r], mod.p, mod.barrett64));
            }
        } else if constexpr (SOURCE == kSource32Rows) {
            const uint32_t* from = x;
            if (within < source.L) {  // wave-uniform
                const size_t item = set >> 2, slot = set & 3;
                from = ((slot & 2) != 0 ? source.second : source.first) + item * source.stride +
                       (((slot & 1) * source.L + within) << LOGN);
            }
            row32_load<LOGN, LOGE, LO0, LOGE>(v, tid, from);
        } else {
            row32_load<LOGN, LOGE, LO0, LOGE>(v, tid, x);
        }
        forward_pass32<LOGN, LOGE, LO0, LOGE>(v, tid, tw, p, true);
        tile32_store<LOGN, LOGE, LO0, LOGE>(v, tid, tile);
        __syncthreads();
        if constexpr (S::P >= 3) {
            constexpr int LO1 = LOGN - 2 * LOGE;
            tile32_load<LOGN, LOGE, LO1, LOGE>(v, tid, tile);
            forward_pass32<LOGN, LOGE, LO1, LOGE>(v, tid, tw, p, false);
            tile32_store<LOGN, LOGE, LO1, LOGE>(v, tid, tile);
            ntt::lds_transpose_fence<LOGN, LOGE, LO1, (S::P >= 4 ? LOGN - 3 * LOGE : 0)>();
        }
        if constexpr (S::P >= 4) {
            constexpr int LO2 = LOGN - 3 * LOGE;
            tile32_load<LOGN, LOGE, LO2, LOGE>(v, tid, tile);
            forward_pass32<LOGN, LOGE, LO2, LOGE>(v, tid, tw, p, false);
            tile32_store<LOGN, LOGE, LO2, LOGE>(v, tid, tile);
            ntt::lds_transpose_fence<LOGN, LOGE, LO2, (S::P >= 5 ? LOGN - 4 * LOGE : 0)>();
        }
        if constexpr (S::P >= 5) {
            constexpr int LO3 = LOGN - 4 * LOGE;
            tile32_load<LOGN, LOGE, LO3, LOGE>(v, tid, tile);
            forward_pass32<LOGN, LOGE, LO3, LOGE>(v, tid, tw, p, false);
            tile32_store<LOGN, LOGE, LO3, LOGE>(v, tid, tile);
            ntt::lds_transpose_fence<LOGN, LOGE, LO3, 0>();
        }
        tile32_load<LOGN, LOGE, 0, S::R>(v, tid, tile);
        forward_pass32<LOGN, LOGE, 0, S::R>(v, tid, tw, p, false);
#pragma unroll
        for (int r = 0; r < E; ++r) v[r] = csub32(csub32(v[r], 2 * p), p);
        // (the transpose into the last pass stayed inside the wave for every schedule with three or more passes)
        if constexpr (kStaged32<LOGN, LOGE, S::R> && S::P >= 3 && ntt::kWaveOwnsTopBits<LOGN, LOGE, LOGN - (S::P - 1) * LOGE>) {
            row32_store_staged<LOGN, LOGE, S::R>(v, tid, x, tile);
        } else {
            row32_store<LOGN, LOGE, 0, S::R>(v, tid, x);
        }
    } else {
        if constexpr (SOURCE == kSource32Tensor) {
            const size_t poly_words = static_cast<size_t>(mod_period) << LOGN;
            const uint32_t* const base = source.first + size_t(group) * 4 * poly_words + (size_t(within) << LOGN);
            uint32_t a[E], b[E];
            if (replica != 1) {  // wave-uniform: a0 b0 or a1 b1
                row32_load<LOGN, LOGE, 0, S::R>(a, tid, base + (replica == 0 ? 0 : 1) * poly_words);
                row32_load<LOGN, LOGE, 0, S::R>(b, tid, base + (replica == 0 ? 2 : 3) * poly_words);
#pragma unroll
                for (int r = 0; r < E; ++r)
                    v[r] = static_cast<uint32_t>(barrett_reduce64_uniform(mul32(a[r], b[r]), mod.p, mod.barrett64));
            } else {           // a0 b1 + a1 b0: two products below 2^60 in one word, one reduction
                uint32_t c[E], d[E];
                row32_load<LOGN, LOGE, 0, S::R>(a, tid, base);
                row32_load<LOGN, LOGE, 0, S::R>(b, tid, base + 3 * poly_words);
                row32_load<LOGN, LOGE, 0, S::R>(c, tid, base + poly_words);
                row32_load<LOGN, LOGE, 0, S::R>(d, tid, base + 2 * poly_words);
#pragma unroll
                for (int r = 0; r < E; ++r)
                    v[r] = static_cast<uint32_t>(barrett_reduce64_uniform(mad32(c[r], d[r], mul32(a[r], b[r])), mod.p, mod.barrett64));
            }
        } else if constexpr (is_key_mac32(SOURCE)) {
            const uint32_t L = source.L, top_rows = source.top_rows;
            const uint32_t key_row = within == L ? top_rows - 1 : within;  // Bfv+Keys.swift:153
            const size_t poly = group;
            const uint32_t* spread_row = source.first + ((poly * L * mod_period + within) << LOGN);       // + j (L+1) N
            const uint32_t* key_rows = source.second + ((size_t(replica) * top_rows + key_row) << LOGN);  // + j 2 top_rows N
            uint64_t sum[E];
#pragma unroll
            for (int r = 0; r < E; ++r) sum[r] = 0;
            uint32_t a[E], b[E];
            row32_load<LOGN, LOGE, 0, S::R>(a, tid, spread_row);
            row32_load<LOGN, LOGE, 0, S::R>(b, tid, key_rows);
            for (uint32_t j = 0; j < L; ++j) {
                // the words of term j + 1 are requested before term j is accumulated (past the last term the request
                // repeats it: a load behind a branch would drain the queue)
                const uint32_t ahead = j + 1 < L ? j + 1 : j;
                uint32_t an[E], bn[E];
                row32_load<LOGN, LOGE, 0, S::R>(an, tid, spread_row + ((size_t(ahead) * mod_period) << LOGN));
                row32_load<LOGN, LOGE, 0, S::R>(bn, tid, key_rows + ((size_t(ahead) * 2 * top_rows) << LOGN));
#pragma unroll
                for (int r = 0; r < E; ++r) {
                    sum[r] = mad32(a[r], b[r], sum[r]);
                    a[r] = an[r];
                    b[r] = bn[r];
                }
            }
#pragma unroll
            for (int r = 0; r < E; ++r) v[r] = static_cast<uint32_t>(barrett_reduce64_uniform(sum[r], mod.p, mod.barrett64));
        } else if constexpr (kStaged32<LOGN, LOGE, S::R>) {
            row32_load_staged<LOGN, LOGE, S::R>(v, tid, x, tile);
        } else {
            row32_load<LOGN, LOGE, 0, S::R>(v, tid, x);
        }
        inverse_pass32<LOGN, LOGE, 0, S::R>(v, tid, tw, mod, true);
        tile32_store<LOGN, LOGE, 0, S::R>(v, tid, tile);
        if constexpr (S::P >= 3) {
            ntt::lds_transpose_fence<LOGN, LOGE, 0, S::R>();
            constexpr int LO1 = S::R;
            tile32_load<LOGN, LOGE, LO1, LOGE>(v, tid, tile);
            inverse_pass32<LOGN, LOGE, LO1, LOGE>(v, tid, tw, mod, false);
            tile32_store<LOGN, LOGE, LO1, LOGE>(v, tid, tile);
        }
        if constexpr (S::P >= 4) {
            ntt::lds_transpose_fence<LOGN, LOGE, S::R, S::R + LOGE>();
            constexpr int LO2 = S::R + LOGE;
            tile32_load<LOGN, LOGE, LO2, LOGE>(v, tid, tile);
            inverse_pass32<LOGN, LOGE, LO2, LOGE>(v, tid, tw, mod, false);
            tile32_store<LOGN, LOGE, LO2, LOGE>(v, tid, tile);
        }
        if constexpr (S::P >= 5) {
            ntt::lds_transpose_fence<LOGN, LOGE, S::R + LOGE, S::R + 2 * LOGE>();
            constexpr int LO3 = S::R + 2 * LOGE;
            tile32_load<LOGN, LOGE, LO3, LOGE>(v, tid, tile);
            inverse_pass32<LOGN, LOGE, LO3, LOGE>(v, tid, tw, mod, false);
            tile32_store<LOGN, LOGE, LO3, LOGE>(v, tid, tile);
        }
        __syncthreads();
        constexpr int LOL = LOGN - LOGE;
        tile32_load<LOGN, LOGE, LOL, LOGE>(v, tid, tile);
        inverse_pass32<LOGN, LOGE, LOL, LOGE>(v, tid, tw, mod, false);
        if constexpr (SOURCE == kSource32KeyMacFinish) {
            // with x_ks the q_ks word of the coefficient and c its centred representative: out = (x - c) q_ks^-1 mod q_r
            // (+ the ciphertext word); |c| is reduced mod q_r where q_ks / 2 is not below it (wave-uniform)
            const uint32_t L = source.L, r = within;
            const uint64_t q = mod.p, q_last = ctx.moduli[L].p, half = q_last >> 1;
            const U64x2 inverse_q_last = ctx.inverse_q_last[size_t(L) * ctx.moduli_stride + r];
            const size_t pc = size_t(group) * 2 + replica;  // polynomial * 2 + c
            const bool add = replica < source.added_polys, wide = half >= q;
            uint32_t last[E], added[E];
            row32_load<LOGN, LOGE, LOL, LOGE>(last, tid, slab + ((pc * (L + 1) + L) << LOGN));
            if (add) row32_load<LOGN, LOGE, LOL, LOGE>(added, tid, source.ct_base + size_t(group) * source.ct_stride + ((size_t(replica) * L + r) << LOGN));
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const uint64_t shifted = add_mod_uniform(last[e], half, q_last);
                const bool negative = shifted < half;
                uint64_t t = negative ? half - shifted : shifted - half;
                if (wide) t = barrett_reduce64_uniform(t, q, mod.barrett64);
                const uint64_t difference = csub_uniform(uint64_t(v[e]) + (negative ? t : q - t), q);
                const uint64_t update = shoup_mul_uniform(difference, inverse_q_last.x, inverse_q_last.y, q);
                v[e] = static_cast<uint32_t>(csub_uniform((add ? uint64_t(added[e]) : 0) + update, q));
            }
            row32_store<LOGN, LOGE, LOL, LOGE>(v, tid, source.out + ((pc * L + r) << LOGN));
        } else {
            row32_store<LOGN, LOGE, LOL, LOGE>(v, tid, x);
        }
    }
}

template <int LOGN, int LOGT, int SOURCE = kSource32Slab>
hipError_t launch_ntt32_tiled(bool inverse, uint32_t* slab, const DeviceContext32& ctx, uint32_t mod_base,
                              uint32_t mod_period, size_t rows, hipStream_t stream,
                              const Source32& source = Source32{nullptr, nullptr, 0, 0, 0, 0, 0, nullptr, 0, nullptr, 0}) {
    constexpr size_t lds_bytes = tile32_words(1u << LOGN) * sizeof(uint32_t);
    using Kernel = void (*)(uint32_t*, const DeviceContext32, uint32_t, uint32_t, const Source32);
    Kernel kernel;
    if constexpr (SOURCE == kSource32Slab) {
        kernel = inverse ? ntt32_tiled_kernel<LOGN, LOGT, true> : ntt32_tiled_kernel<LOGN, LOGT, false>;
    } else if constexpr (SOURCE == kSource32Spread || SOURCE == kSource32Rows) {
        if (inverse) return hipErrorInvalidValue;
        kernel = ntt32_tiled_kernel<LOGN, LOGT, false, SOURCE>;
    } else {
        if (!inverse) return hipErrorInvalidValue;
        kernel = ntt32_tiled_kernel<LOGN, LOGT, true, SOURCE>;
    }
    if (lds_bytes > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(rows)), dim3(1u << LOGT), lds_bytes, stream, slab, ctx,
                       mod_base, mod_period, source);
    return hipGetLastError();
}

// the fused-source launches: every record's rows in one launch (rows <= 2^30), tiled degrees only
template <int SOURCE>
hipError_t launch_ntt32_fused(bool inverse, uint32_t* slab, const DeviceContext32& ctx, uint32_t record_rows, size_t rows,
                              const Source32& source, hipStream_t stream) {
    if (rows == 0) return hipSuccess;
    if (rows > (size_t(1) << 30) || ctx.degree < 2) return hipErrorNotSupported;
    switch (ctx.log_degree) {
        case 12: return launch_ntt32_tiled<12, 9, SOURCE>(inverse, slab, ctx, 0, record_rows, rows, stream, source);
        case 13: return launch_ntt32_tiled<13, 10, SOURCE>(inverse, slab, ctx, 0, record_rows, rows, stream, source);
        case 14: return launch_ntt32_tiled<14, 10, SOURCE>(inverse, slab, ctx, 0, record_rows, rows, stream, source);
        default: return hipErrorNotSupported;
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

// Bfv+Keys.swift:165-179 into the forward transform's load: target polynomials of L rows, `stride` words apart ->
// spread [polys][L][L+1][N] (Eval).  hipErrorNotSupported where the degree has no tiled 4-byte transform.
hipError_t launch_ntt32_spread(const uint32_t* target, size_t stride, uint32_t L, size_t polys, uint32_t* spread,
                               const DeviceContext32& ks_ctx, hipStream_t stream) {
    if (ks_ctx.moduli_count < L + 1) return hipErrorInvalidValue;
    return launch_ntt32_fused<kSource32Spread>(false, spread, ks_ctx, L + 1, polys * L * (L + 1),
                                               Source32{target, nullptr, stride, L, 0, 0, 0, nullptr, 0, nullptr, 0}, stream);
}
// Forward transform of lifted [items][4][rows][N] records whose rows [0, L) were left unwritten by the lift: they are read
// from the ciphertext pairs (Bfv+Multiply.swift:51-57)
bool ntt32_lifted_forward_supported(const DeviceContext32& ctx) { return ctx.log_degree >= 12 && ctx.log_degree <= 14; }
hipError_t launch_ntt32_lifted_forward(uint32_t* lifted, const DeviceContext32& ctx, uint32_t record_rows, size_t items,
                                       const uint32_t* lhs, const uint32_t* rhs, size_t stride, uint32_t L,
                                       hipStream_t stream) {
    return launch_ntt32_fused<kSource32Rows>(false, lifted, ctx, record_rows, items * 4 * record_rows,
                                             Source32{lhs, rhs, stride, L, 0, 0, 0, nullptr, 0, nullptr, 0}, stream);
}
// Bfv+Multiply.swift:80-82 into the inverse transform's load: lifted [items][4][rows][N] (Eval) -> out [items][3][rows][N]
// (Coeff; the context carries t N^-1)
hipError_t launch_ntt32_tensor_inverse(const uint32_t* lifted, uint32_t* out, const DeviceContext32& ctx, uint32_t record_rows,
                                       size_t items, hipStream_t stream) {
    return launch_ntt32_fused<kSource32Tensor>(true, out, ctx, record_rows, items * 3 * record_rows,
                                               Source32{lifted, nullptr, 0, 0, 0, 0, 0, nullptr, 0, nullptr, 0}, stream);
}
// Bfv+Keys.swift:180-207 into the inverse transform's load: spread [polys][L][L+1][N], key [L][2][top_rows][N] ->
// out [polys][2][L+1][N] (Coeff)
hipError_t launch_ntt32_key_mac_inverse(const uint32_t* spread, const uint32_t* key, uint32_t* out,
                                        const DeviceContext32& ks_ctx, uint32_t L, uint32_t top_rows, size_t polys,
                                        hipStream_t stream) {
    if (L > 15) return hipErrorNotSupported;  // the sum of L products below 2^60 stays in 64 bits
    return launch_ntt32_fused<kSource32KeyMac>(true, out, ks_ctx, L + 1, polys * 2 * (L + 1),
                                               Source32{spread, key, 0, L, top_rows, 0, 0, nullptr, 0, nullptr, 0}, stream);
}
// The same with the key switch's last step in the store of the rows r < L (kSource32KeyMacFinish): first the q_ks row of
// every (polynomial, c) into `prod`, then the other rows, which read it: out [polys][2][L][N].  hipErrorNotSupported
// (nothing launched) where the degree has no tiled 4-byte transform.
hipError_t launch_ntt32_key_mac_inverse_finish(const uint32_t* spread, const uint32_t* key, uint32_t* prod,
                                               const uint32_t* ct_base, size_t ct_stride, uint32_t* out,
                                               const DeviceContext32& ks_ctx, uint32_t L, uint32_t top_rows, size_t polys,
                                               uint32_t added_polys, hipStream_t stream) {
    // (beyond two workgroup generations only -- 4 x 256 workgroups at N = 4096: a smaller batch is latency-bound and two
    // transforms in a row cost more than one transform and the small element-wise kernel)
    if (L == 0 || L > 15 || ks_ctx.moduli_count != L + 1 || ks_ctx.log_degree < 12 || ks_ctx.log_degree > 14 ||
        polys * 2 * (L + 1) <= 2048)
        return hipErrorNotSupported;
    if (polys == 0) return hipSuccess;
    hipError_t e = launch_ntt32_fused<kSource32KeyMac>(
        true, prod, ks_ctx, L + 1, polys * 2, Source32{spread, key, 0, L, top_rows, L, 1, nullptr, 0, nullptr, 0}, stream);
    if (e != hipSuccess) return e;
    return launch_ntt32_fused<kSource32KeyMacFinish>(
        true, prod, ks_ctx, L + 1, polys * 2 * L, Source32{spread, key, 0, L, top_rows, 0, L, ct_base, ct_stride, out, added_polys},
        stream);
}

hipError_t launch_ntt32(bool inverse, uint32_t* slab, const DeviceContext32& ctx, uint32_t mod_base, uint32_t mod_period,
                        size_t rows, hipStream_t stream) {
    if (rows == 0 || ctx.degree < 2) return hipSuccess;
    if (ctx.log_degree > 15) return hipErrorNotSupported;  // the row must fit the LDS (4 N bytes + padding)
    constexpr size_t kMaxRowsPerLaunch = size_t(1) << 30;
    for (size_t done = 0; done < rows;) {
        const size_t chunk = (kMaxRowsPerLaunch / mod_period) * mod_period;
        const size_t now = rows - done < chunk ? rows - done : chunk;
        const uint32_t n = ctx.degree;
        if (ctx.log_degree >= 12 && ctx.log_degree <= 14) {  // register-tiled kernels, 8 (16) words per lane
            hipError_t e = ctx.log_degree == 12   ? launch_ntt32_tiled<12, 9>(inverse, slab + done * n, ctx, mod_base, mod_period, now, stream)
                           : ctx.log_degree == 13 ? launch_ntt32_tiled<13, 10>(inverse, slab + done * n, ctx, mod_base, mod_period, now, stream)
                                                  : launch_ntt32_tiled<14, 10>(inverse, slab + done * n, ctx, mod_base, mod_period, now, stream);
            if (e != hipSuccess) return e;
            done += now;
            continue;
        }
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

// ---- word-size bridge between packed [UInt32] slabs and the 8-byte words of the Bfv<UInt32> scheme layer -------
// four words per lane: 16 B in / 32 B out (widen) or 32 B in / 16 B out (narrow); a narrowed word keeps its low half
// (scheme-layer results of a UInt32 context are < 2^30)
namespace {
typedef uint32_t Packed4 __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(kThreads)
    widen_kernel(const uint32_t* __restrict__ in, uint64_t* __restrict__ out, size_t words) {
    const size_t quads = words >> 2;
    for (size_t q = blockIdx.x * static_cast<size_t>(kThreads) + threadIdx.x; q < quads;
         q += static_cast<size_t>(gridDim.x) * kThreads) {
        const Packed4 x = reinterpret_cast<const Packed4*>(in)[q];
        reinterpret_cast<U64x2*>(out)[2 * q] = U64x2{x.x, x.y};
        reinterpret_cast<U64x2*>(out)[2 * q + 1] = U64x2{x.z, x.w};
    }
    if (blockIdx.x == 0 && threadIdx.x < (words & 3)) out[(quads << 2) + threadIdx.x] = in[(quads << 2) + threadIdx.x];
}
__global__ void __launch_bounds__(kThreads)
    narrow_kernel(const uint64_t* __restrict__ in, uint32_t* __restrict__ out, size_t words) {
    const size_t quads = words >> 2;
    for (size_t q = blockIdx.x * static_cast<size_t>(kThreads) + threadIdx.x; q < quads;
         q += static_cast<size_t>(gridDim.x) * kThreads) {
        const U64x2 a = reinterpret_cast<const U64x2*>(in)[2 * q], b = reinterpret_cast<const U64x2*>(in)[2 * q + 1];
        Packed4 y;
        y.x = static_cast<uint32_t>(a.x);
        y.y = static_cast<uint32_t>(a.y);
        y.z = static_cast<uint32_t>(b.x);
        y.w = static_cast<uint32_t>(b.y);
        reinterpret_cast<Packed4*>(out)[q] = y;
    }
    if (blockIdx.x == 0 && threadIdx.x < (words & 3))
        out[(quads << 2) + threadIdx.x] = static_cast<uint32_t>(in[(quads << 2) + threadIdx.x]);
}
// the reference point of the transforms' roofline figures: every word read once and written once, 8 bytes per lane (the
// NTT's access width).  NT = non-temporal like the transforms' row loads and stores, or the default cache policy.
template <bool NT>
__global__ void __launch_bounds__(kThreads)
    stream_copy_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, size_t words) {
    for (size_t i = blockIdx.x * static_cast<size_t>(kThreads) + threadIdx.x; i < words;
         i += static_cast<size_t>(gridDim.x) * kThreads) {
        if constexpr (NT) stream_store(out + i, stream_load(in + i));
        else out[i] = in[i];
    }
}
// `records` runs of record_words words, source runs src_stride words apart, destination runs dst_stride apart; four 16-byte
// words per lane (one per lane made 65 536 workgroups of the PIR tail's copy: 75 us for 2 x 134 MB)
constexpr int kCopyRecordVectors = 4;
__global__ void __launch_bounds__(kThreads)
    copy_records_kernel(const uint64_t* __restrict__ in, size_t src_stride, uint64_t* __restrict__ out, size_t dst_stride,
                        uint32_t blocks_per_record) {
    const size_t record = blockIdx.x / blocks_per_record;
    const size_t first = (blockIdx.x - record * blocks_per_record) * size_t(kThreads) * kCopyRecordVectors + threadIdx.x;
    const U64x2* src = reinterpret_cast<const U64x2*>(in + record * src_stride) + first;
    U64x2* dst = reinterpret_cast<U64x2*>(out + record * dst_stride) + first;
    U64x2 words[kCopyRecordVectors];
#pragma unroll
    for (int v = 0; v < kCopyRecordVectors; ++v) words[v] = src[v * kThreads];
#pragma unroll
    for (int v = 0; v < kCopyRecordVectors; ++v) dst[v * kThreads] = words[v];
}
}  // namespace

hipError_t launch_copy_records(const uint64_t* in, size_t src_stride, uint64_t* out, size_t dst_stride, size_t record_words,
                               size_t records, hipStream_t stream) {
    if (records == 0 || record_words == 0) return hipSuccess;
    constexpr size_t block_words = 2 * size_t(kThreads) * kCopyRecordVectors;
    const size_t blocks_per_record = record_words / block_words;
    if (record_words % block_words != 0 || (src_stride | dst_stride) % 2 != 0 || blocks_per_record * records >= (size_t(1) << 31) ||
        ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15u) != 0)
        return hipErrorInvalidValue;
    hipLaunchKernelGGL(copy_records_kernel, dim3(static_cast<unsigned>(blocks_per_record * records)), dim3(kThreads), 0, stream, in,
                       src_stride, out, dst_stride, static_cast<uint32_t>(blocks_per_record));
    return hipGetLastError();
}

hipError_t launch_stream_copy(const uint64_t* in, uint64_t* out, size_t words, bool non_temporal, hipStream_t stream) {
    if (words == 0) return hipSuccess;
    if (non_temporal)
        hipLaunchKernelGGL(stream_copy_kernel<true>, dim3(grid_for(words)), dim3(kThreads), 0, stream, in, out, words);
    else
        hipLaunchKernelGGL(stream_copy_kernel<false>, dim3(grid_for(words)), dim3(kThreads), 0, stream, in, out, words);
    return hipGetLastError();
}

hipError_t launch_widen_words(const uint32_t* in, uint64_t* out, size_t words, hipStream_t stream) {
    if (words == 0) return hipSuccess;
    hipLaunchKernelGGL(widen_kernel, dim3(grid_for((words >> 2) + 1)), dim3(kThreads), 0, stream, in, out, words);
    return hipGetLastError();
}
hipError_t launch_narrow_words(const uint64_t* in, uint32_t* out, size_t words, hipStream_t stream) {
    if (words == 0) return hipSuccess;
    hipLaunchKernelGGL(narrow_kernel, dim3(grid_for((words >> 2) + 1)), dim3(kThreads), 0, stream, in, out, words);
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
