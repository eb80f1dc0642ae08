// rns_kernels.hip -- BEHZ base conversions, the ct x ct tensor product and hybrid key switching for gfx950.
//
// Every kernel here is per-coefficient (or per-word) independent work over [.., rows, N] slabs: one lane per
// coefficient column, consecutive lanes on consecutive columns, each row access a coalesced 512-byte wave load.
// Reference semantics (Sources/HomomorphicEncryption/):
//   _RnsTool.liftQToQBsk            RnsTool.swift:313-368   (+ _RnsBaseConverter, RnsBaseConverter.swift:97-143)
//   _RnsTool.floorQBskToQ           RnsTool.swift:378-456
//   Bfv.multiplyWithoutScaling      Bfv/Bfv+Multiply.swift:63-85 (tensor product in Eval form over [Q, Bsk])
//   Bfv._computeKeySwitchingUpdate  Bfv/Bfv+Keys.swift:123-208
// All base-conversion sums are exact wrapping UInt128 sums followed by the double-word Barrett reduction, exactly as
// the reference does, so the overflow term "a" of the fast base conversion is reproduced bit for bit.
#include <hip/hip_runtime.h>

#include "bfv_context.hpp"
#include "device_math.hpp"
#include "rns_kernels.hpp"

namespace heamd {

namespace {

constexpr unsigned kThreads = 256;
constexpr int kMaxL = 16;  // compile-time specialisations for L = 1..kMaxL (16 x 55 bits = the N = 32768 security cap)

constexpr size_t kGridCap = (size_t(1) << 31) - 1;
// One workgroup per kThreads work items, up to the grid limit: the kernels keep their grid-stride loops for what lies beyond it,
// but a lane that walks many items serialises its loads -- divideAndRoundQLast at N = 16384, L = 6 ran at 0.66 of 8 TB/s on
// 256 x 8 workgroups and at 0.79 with one item per lane (profiles/r06y_exact_grids.txt)
inline unsigned grid_for(size_t work_items) {
    const size_t blocks = (work_items + kThreads - 1) / kThreads;
    const size_t cap = kGridCap;
    return static_cast<unsigned>(blocks < cap ? (blocks ? blocks : 1) : cap);
}

// one lane per work item, no striding (kernels that keep many table constants live)
inline unsigned exact_grid(size_t work_items) {
    const size_t blocks = (work_items + kThreads - 1) / kThreads;
    return static_cast<unsigned>(blocks ? blocks : 1);
}

__device__ __forceinline__ uint64_t reduce128(U128 x, const DeviceModulus& m) {
    return barrett_reduce128(x, m.p, m.barrett128_lo, m.barrett128_hi);
}
// x < 2^63 times a table constant (wave-uniform), canonical result
__device__ __forceinline__ uint64_t shoup_mul_pair(uint64_t x, U64x2 c, uint64_t p) {
    return shoup_mul_uniform(x, c.x, c.y, p);
}

// ---- arithmetic by slab word type ----------------------------------------------------------------------------------
// The kernels below are one body for Bfv<UInt64> and Bfv<UInt32>; what differs is how a sum of products and a product
// by a table constant are carried out.  8-byte words: 62-bit moduli, exact 128-bit sums on the carry-counting
// accumulator, 64-bit Shoup products.  4-byte words: every modulus (Bsk primes, gamma, t, mTilde included) is below 2^30
// (MA/Scalar.swift:498-511), so a product is below 2^60, sixteen of them fit a 64-bit word (the kernels sum at most
// nine), a sum is ONE multiply-add per term, its reduction a single-word Barrett, and a product by a constant a 32-bit
// Shoup product (the table's floor(w 2^64 / p) holds floor(w 2^32 / p) in its high word) -- a fifth of the instructions.
// 4-byte words: items between folds of a 64-bit sum that takes two products below 2^60 per item (2 x 7 + 1 < 16)
constexpr uint64_t kNarrowTerms = 7;
template <typename W>
struct WordArith {  // uint64_t
    using Sum = ProductSum;
    static constexpr uint64_t kSlack = 5;  // a lazily reduced sum lies in [0, kSlack p)
    // BOUNDED (below): the cross column of the sum cannot wrap either -- its carry counts are skipped
    template <bool BOUNDED = false>
    static __device__ __forceinline__ Sum first(uint64_t a, uint64_t b) {
        if constexpr (BOUNDED) return product_sum_first_uniform_short(a, b);
        else return product_sum_first_uniform(a, b);
    }
    template <bool BOUNDED = false>
    static __device__ __forceinline__ void add(Sum& s, uint64_t a, uint64_t b) {
        if constexpr (BOUNDED) product_sum_add_uniform_short(s, a, b);
        else product_sum_add_uniform(s, a, b);
    }
    static __device__ __forceinline__ void add_vector(Sum& s, uint64_t a, uint64_t b) { product_sum_add(s, a, b); }
    static __device__ __forceinline__ Sum zero() { return product_sum_zero(); }
    static __device__ __forceinline__ uint64_t low_word(const Sum& s) { return product_sum_value(s).lo; }
    // BOUNDED: the sum is known to lie below 2^(64 + m.wide_shift) (RnsToolDevice::wide_reduce_ok): the one-word-quotient
    // Barrett of device_math.hpp, 21 / 25 issue slots less per residue
    template <bool BOUNDED = false, typename Modulus>
    static __device__ __forceinline__ uint64_t reduce(const Sum& s, const Modulus& m) {
        if constexpr (BOUNDED) return reduce_product_sum_bounded(s, m);
        else return reduce_product_sum(s, m);
    }
    template <bool BOUNDED = false, typename Modulus>
    static __device__ __forceinline__ uint64_t reduce_lazy(const Sum& s, const Modulus& m) {
        if constexpr (BOUNDED) return reduce_product_sum_bounded_lazy(s, m);
        else return reduce_product_sum_lazy(s, m);
    }
    static __device__ __forceinline__ uint64_t shoup(uint64_t x, U64x2 c, uint64_t p) {
        return shoup_mul_uniform(x, c.x, c.y, p);
    }
    static __device__ __forceinline__ uint64_t shoup_lazy(uint64_t x, U64x2 c, uint64_t p) {
        return shoup_mul_uniform_lazy(x, c.x, c.y, p);
    }
    static __device__ __forceinline__ uint64_t mul(uint64_t a, uint64_t b, const DeviceModulus& m) {
        return barrett_mul(a, b, m.p, m.product_factor, static_cast<int>(m.product_shift));
    }
};
template <>
struct WordArith<uint32_t> {
    using Sum = uint64_t;
    static constexpr uint64_t kSlack = 1;  // sums are reduced to [0, p) at once
    template <bool BOUNDED = false>
    static __device__ __forceinline__ Sum first(uint64_t a, uint64_t b) { return mul32(lo32(a), lo32(b)); }
    template <bool BOUNDED = false>
    static __device__ __forceinline__ void add(Sum& s, uint64_t a, uint64_t b) { s = mad32(lo32(a), lo32(b), s); }
    static __device__ __forceinline__ void add_vector(Sum& s, uint64_t a, uint64_t b) { s = mad32(lo32(a), lo32(b), s); }
    static __device__ __forceinline__ Sum zero() { return 0; }
    static __device__ __forceinline__ uint64_t low_word(const Sum& s) { return s; }
    template <bool BOUNDED = false, typename Modulus>
    static __device__ __forceinline__ uint64_t reduce(const Sum& s, const Modulus& m) {
        return barrett_reduce64_uniform(s, m.p, m.barrett64);
    }
    template <bool BOUNDED = false, typename Modulus>
    static __device__ __forceinline__ uint64_t reduce_lazy(const Sum& s, const Modulus& m) { return reduce(s, m); }
    // x < 2^32, constant w < p < 2^31 (or p = 2^16 for mTilde): x w - floor(x floor(w 2^32 / p) / 2^32) p in [0, 2p)
    static __device__ __forceinline__ uint64_t shoup(uint64_t x, U64x2 c, uint64_t p) {
        const uint32_t xs = lo32(x), q = mulhi32(xs, hi32(c.y));
        const uint32_t r = xs * lo32(c.x) - q * lo32(p);
        return r >= lo32(p) ? r - lo32(p) : r;
    }
    static __device__ __forceinline__ uint64_t shoup_lazy(uint64_t x, U64x2 c, uint64_t p) { return shoup(x, c, p); }
    static __device__ __forceinline__ uint64_t mul(uint64_t a, uint64_t b, const DeviceModulus& m) {
        return barrett_reduce64_uniform(mul32(lo32(a), lo32(b)), m.p, m.barrett64);
    }
};

// V consecutive coefficients of a row per lane.  4-byte slabs: FOUR words per lane in lift and floor (one 16-byte access per
// lane and row instead of four 4-byte ones: lift 129.6 -> 102.9 us and floor 211.3 -> 187.7 us per 2048 products at
// n_4096_logq_27_28_28, profiles/r04n_word32_base_conversions_ab.txt; two words per lane: 112.8 / 199.0).  8-byte slabs: the
// kernels are bound by their multiply-adds (about 400 issue slots per coefficient at L = 4); the lift takes two words per
// lane (kLiftPairs below), the floor one.  Rows whose base is not 16-byte aligned fall back to one word per lane.
template <int V, typename W>
__device__ __forceinline__ void load_words(const W* p, uint64_t (&x)[V]) {
    if constexpr (V == 1) {
        x[0] = stream_load(p);
    } else if constexpr (V == 2 && sizeof(W) == 8) {
        const StreamWords pair = __builtin_nontemporal_load(reinterpret_cast<const StreamWords*>(p));
        x[0] = pair.x;
        x[1] = pair.y;
    } else if constexpr (V == 2) {
        const uint64_t pair = __builtin_nontemporal_load(reinterpret_cast<const uint64_t*>(p));
        x[0] = lo32(pair);
        x[1] = hi32(pair);
    } else {
        static_assert(V == 4 && sizeof(W) == 4, "four 4-byte words per lane");
        const StreamWords quad = __builtin_nontemporal_load(reinterpret_cast<const StreamWords*>(p));
        x[0] = lo32(quad.x);
        x[1] = hi32(quad.x);
        x[2] = lo32(quad.y);
        x[3] = hi32(quad.y);
    }
}
template <int V, typename W>
__device__ __forceinline__ void store_words(W* p, const uint64_t (&x)[V]) {
    if constexpr (V == 1) {
        stream_store(p, x[0]);
    } else if constexpr (V == 2 && sizeof(W) == 8) {
        const StreamWords pair = {x[0], x[1]};
        __builtin_nontemporal_store(pair, reinterpret_cast<StreamWords*>(p));
    } else if constexpr (V == 2) {
        __builtin_nontemporal_store(lo32(x[0]) | (x[1] << 32), reinterpret_cast<uint64_t*>(p));
    } else {
        static_assert(V == 4 && sizeof(W) == 4, "four 4-byte words per lane");
        const StreamWords quad = {lo32(x[0]) | (x[1] << 32), lo32(x[2]) | (x[3] << 32)};
        __builtin_nontemporal_store(quad, reinterpret_cast<StreamWords*>(p));
    }
}
// 8-byte slabs: the bounded lift takes two coefficients per lane (one 16-byte access per lane and row; scalar constants
// and addresses shared by both): 588.7 -> 560.7 us per 1024 products at N = 8192, L = 4; the floor gained under 1 % that way
// and stays at one (profiles/r04p_behz_two_coefficients_per_lane.txt)
constexpr bool kLiftPairs = true;
template <typename W>
inline bool quad_aligned(const W* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// ---- liftQToQBsk: in [polys][L][N] -> out [polys][2L+1][N] -------------------------------------------------------
// Polynomial p = item * polys_per_item + c is read at in + item * in_item_stride + c * L * N and written at
// out + item * out_item_stride + c * (2L+1) * N (strides in words), so one launch can fill a slot range of a larger
// per-item record.
struct LiftLayout {
    size_t polys_per_item, in_item_stride, out_item_stride;
    uint32_t store_input;  // 0: rows [0, L) of the output are left to the transform that follows (it reads the input itself)
    // 1: the Bsk rows may be left as the one-word-quotient reduction hands them over, in [0, 5p), for a consumer that takes such
    // words (the row-fused ct x ct kernel's fold butterflies: behz_kernels.hip) -- three conditional subtracts per word less
    uint32_t lazy_output;
    // a second operand in the same launch (ct x ct lifts both ciphertexts of every pair): polynomials [first_polys, polys) are
    // read from `second` with the same strides and written second_out_delta words from where the first operand's are; nullptr:
    // one operand
    const void* second = nullptr;
    size_t first_polys = 0;
    ptrdiff_t second_out_delta = 0;
};

// W: the slab's word type -- uint64_t (Bfv<UInt64>) or uint32_t (Bfv<UInt32>: every modulus <= 2^30 - 1).  Words are
// widened when loaded and narrowed when stored; the arithmetic is the same code for both.
template <int L, typename W, bool BOUNDED, int V = 1>
__global__ void __launch_bounds__(kThreads)
    lift_kernel(const W* __restrict__ in, W* __restrict__ out, const RnsToolDevice tool, size_t polys,
                const LiftLayout layout) {
    const uint32_t logn = tool.log_degree;
    const size_t n = size_t(1) << logn;
    const size_t total = polys << logn;
    using A = WordArith<W>;
    const uint64_t kMTildeValue = tool.mtilde;  // 2^32 (UInt64 contexts) or 2^16 (UInt32 contexts)
    // V coefficients per lane, no grid-stride loop: with a loop hipcc hoists every table constant out of it and
    // spills SGPRs into VGPR lanes
    for (size_t idx = (blockIdx.x * size_t(kThreads) + threadIdx.x) * V; idx < total; idx = total) {
        size_t poly = idx >> logn;
        const size_t k = idx & (n - 1);
        const W* operand = in;
        W* lifted_base = out;
        if (layout.second != nullptr && poly >= layout.first_polys) {  // (uniform: a workgroup's lanes share the polynomial)
            poly -= layout.first_polys;
            operand = static_cast<const W*>(layout.second);
            lifted_base = out + layout.second_out_delta;
        }
        const size_t item = poly / layout.polys_per_item, c = poly - item * layout.polys_per_item;
        const W* src = operand + item * layout.in_item_stride + c * L * n + k;
        W* dst = lifted_base + item * layout.out_item_stride + c * (2 * L + 1) * n + k;
        uint64_t y[V][L];
#pragma unroll
        for (int i = 0; i < L; ++i) {
            uint64_t x[V];
            load_words<V>(src + i * n, x);
            if (layout.store_input != 0) store_words<V>(dst + i * n, x);  // rows [0, L): the input itself (RnsTool.swift:329-330)
#pragma unroll
            for (int v = 0; v < V; ++v) y[v][i] = A::shoup(x[v], tool.lift_scale[i], tool.q_moduli[i].p);
        }
        // mTilde row first: r = -(x' * Q^-1) mod mTilde  (smallMontgomeryReduce, RnsTool.swift:343-348)
        // the (L+2)'th extended modulus is mTilde = 2^32 at the top level; a lower-level tool takes a prefix of
        // [Bsk..., mTilde] and finds a Bsk prime there (the reference's own behaviour, reproduced as is)
        const DeviceModulus last = tool.ext_moduli[L + 1];
        uint64_t r[V];
        bool below[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            if (sizeof(W) == 8 && last.p == kMTildeValue) {
                // mod mTilde = 2^32 only the low words of the terms count: one multiply-add each (uniform branch)
                uint64_t low = mul32(lo32(y[v][0]), lo32(tool.q_to_ext[(L + 1) * L + 0]));
#pragma unroll
                for (int i = 1; i < L; ++i) low = mad32(lo32(y[v][i]), lo32(tool.q_to_ext[(L + 1) * L + i]), low);
                r[v] = low & (kMTildeValue - 1);
            } else {
                typename A::Sum acc = A::first(y[v][0], tool.q_to_ext[(L + 1) * L + 0]);
#pragma unroll
                for (int i = 1; i < L; ++i) A::add(acc, y[v][i], tool.q_to_ext[(L + 1) * L + i]);
                r[v] = last.p == kMTildeValue ? (A::low_word(acc) & (kMTildeValue - 1)) : A::reduce(acc, last);
            }
            r[v] = A::shoup(r[v], tool.neg_inv_q_mod_mtilde, kMTildeValue);
            below[v] = r[v] < (kMTildeValue >> 1);
        }
#pragma unroll
        for (int j = 0; j <= L; ++j) {
            const DeviceModulus m = tool.ext_moduli[j];
            uint64_t lifted[V];
#pragma unroll
            for (int v = 0; v < V; ++v) {
                typename A::Sum sum = A::template first<BOUNDED>(y[v][0], tool.q_to_bsk_scaled[j * L + 0]);
#pragma unroll
                for (int i = 1; i < L; ++i) A::template add<BOUNDED>(sum, y[v][i], tool.q_to_bsk_scaled[j * L + i]);
                const uint64_t centered = below[v] ? r[v] : r[v] + m.p - kMTildeValue;  // RnsTool.swift:357-361
                // (x'_j + (Q mod Bsk_j) r) mTilde^-1 (RnsTool.swift:363-364) with mTilde^-1 already inside both constants
                const U64x2 scaled = tool.q_mod_bsk_scaled[j];
                uint64_t unfolded;
                if constexpr (BOUNDED && sizeof(W) == 8) {
                    // the r term is one more product of the same exact sum (through the multiply-add that counts its
                    // carries: both factors are 61-bit words; RnsToolDevice::wide_reduce_ok covers the longer sum): one
                    // reduction, below 5p, instead of a reduction and a Shoup product
                    A::template add<false>(sum, centered, scaled.x);
                    unfolded = A::template reduce_lazy<BOUNDED>(sum, m);
                } else {
                    // the two terms stay unfolded (< 5p and < 3p; the extended moduli are < 2^61) and the sum is folded once
                    unfolded = A::template reduce_lazy<BOUNDED>(sum, m) + A::shoup_lazy(centered, scaled, m.p);
                }
                if (BOUNDED && sizeof(W) == 8 && layout.lazy_output != 0) lifted[v] = unfolded;  // (wave-uniform branch)
                else lifted[v] = csub_uniform(csub_uniform(csub_uniform(unfolded, 4 * m.p), 2 * m.p), m.p);
            }
            store_words<V>(dst + (L + j) * n, lifted);
        }
    }
}

// ---- floorQBskToQ: in [polys][2L+1][N] -> out [polys][L][N] ------------------------------------------------------
template <int L, typename W, bool BOUNDED, int V = 1>
__global__ void __launch_bounds__(kThreads)
    floor_kernel(const W* __restrict__ in, W* __restrict__ out, const RnsToolDevice tool, size_t polys) {
    using A = WordArith<W>;
    const uint32_t logn = tool.log_degree;
    const size_t n = size_t(1) << logn;
    const size_t total = polys << logn;
    for (size_t idx = (blockIdx.x * size_t(kThreads) + threadIdx.x) * V; idx < total; idx = total) {
        const size_t poly = idx >> logn, k = idx & (n - 1);
        const W* src = in + poly * (2 * L + 1) * n + k;
        W* dst = out + poly * L * n + k;
        // approximateFloor (RnsTool.swift:378-398)
        uint64_t y[V][L];
#pragma unroll
        for (int i = 0; i < L; ++i) {
            uint64_t x[V];
            load_words<V>(src + i * n, x);
#pragma unroll
            for (int v = 0; v < V; ++v) y[v][i] = A::shoup(x[v], tool.inv_punctured_q[i], tool.q_moduli[i].p);
        }
        // (x_Bsk_j - conv_j) Q^-1 mod Bsk_j, and for j < L straight on to the Bsk -> Q converter's first product
        // z_j = f_j (B/Bsk_j)^-1 mod Bsk_j: two exact products mod Bsk_j = one by the product of the constants
        uint64_t z[V][L], f_msk[V];
#pragma unroll
        for (int j = 0; j <= L; ++j) {
            const DeviceModulus m = tool.ext_moduli[j];
            uint64_t x[V];
            load_words<V>(src + (L + j) * n, x);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                typename A::Sum sum = A::template first<BOUNDED>(y[v][0], tool.q_to_ext[j * L + 0]);
#pragma unroll
                for (int i = 1; i < L; ++i) A::template add<BOUNDED>(sum, y[v][i], tool.q_to_ext[j * L + i]);
                // x - conv with conv unfolded in [0, 5p): the difference stays below 6p < 2^63 (extended moduli < 2^63 / 6,
                // checked when the tool is built), which is all the next exact product needs
                const uint64_t difference = x[v] + A::kSlack * m.p - A::template reduce_lazy<BOUNDED>(sum, m);
                if (j < L) {
                    z[v][j] = A::shoup(difference, tool.floor_scale_b[j], m.p);
                } else {
                    f_msk[v] = A::shoup(difference, tool.inv_q_mod_bsk[j], m.p);
                }
            }
        }
        // convertApproximateBskToQ (RnsTool.swift:402-450)
        const DeviceModulus msk = tool.ext_moduli[L];
        uint64_t alpha[V];
        bool exceeds[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            typename A::Sum alpha_sum = A::template first<BOUNDED>(z[v][0], tool.b_to_msk[0]);
#pragma unroll
            for (int i = 1; i < L; ++i) A::template add<BOUNDED>(alpha_sum, z[v][i], tool.b_to_msk[i]);
            // the converter's output modulus is the top level's m_sk (RnsTool.swift:44-62, 240-250); below the top level
            // its canonical residue is then read as an integer mod THIS level's m_sk, as the reference does
            alpha[v] = tool.alpha_modulus_is_msk != 0 ? A::template reduce_lazy<BOUNDED>(alpha_sum, msk)  // < 5 m_sk
                                                      : A::template reduce<BOUNDED>(alpha_sum, tool.alpha_modulus[0]);
            alpha[v] = A::shoup(alpha[v] + msk.p - f_msk[v], tool.inv_b_mod_msk, msk.p);
            exceeds[v] = alpha[v] > (msk.p >> 1);
        }
#pragma unroll
        for (int row = 0; row < L; ++row) {
            const DeviceModulus m = tool.q_moduli[row];
            uint64_t floored[V];
#pragma unroll
            for (int v = 0; v < V; ++v) {
                typename A::Sum sum = A::template first<BOUNDED>(z[v][0], tool.b_to_q[row * L + 0]);
#pragma unroll
                for (int i = 1; i < L; ++i) A::template add<BOUNDED>(sum, z[v][i], tool.b_to_q[row * L + i]);
                // RnsTool.swift:436-446: + (m_sk - alpha) (B mod q) when alpha > m_sk/2, else + alpha (-B mod q)
                if (tool.floor_merge_ok != 0) {
                    // the correction is one more product of the same exact sum: one reduction instead of a Shoup product,
                    // a negation and a modular add (uniform branch; the sum stays below 2^127)
                    const U64x2 plus = tool.b_mod_q[row], minus = tool.neg_b_mod_q[row];
                    A::add_vector(sum, exceeds[v] ? msk.p - alpha[v] : alpha[v], exceeds[v] ? plus.x : minus.x);
                    floored[v] = A::template reduce<BOUNDED>(sum, m);
                } else {
                    const uint64_t converted = A::template reduce<BOUNDED>(sum, m);
                    // the second form is the negation of alpha (B mod q), so one product serves both
                    const uint64_t magnitude = A::shoup(exceeds[v] ? msk.p - alpha[v] : alpha[v], tool.b_mod_q[row], m.p);
                    const uint64_t adjust = exceeds[v] ? magnitude : neg_mod_uniform(magnitude, m.p);
                    floored[v] = add_mod_uniform(converted, adjust, m.p);
                }
            }
            store_words<V>(dst + row * n, floored);
        }
    }
}

// ---- scaleAndRound: in [polys][L][N] (Coeff over Q) -> out [polys][N] (mod t)        RnsTool.swift:272-302 ----------
template <int L, typename W>
__global__ void __launch_bounds__(kThreads)
    scale_and_round_kernel(const W* __restrict__ in, W* __restrict__ out, const RnsToolDevice tool,
                           const U64x2 final_scale, size_t polys) {
    using A = WordArith<W>;
    const uint32_t logn = tool.log_degree;
    const size_t n = size_t(1) << logn;
    const size_t total = polys << logn;
    for (size_t idx = blockIdx.x * size_t(kThreads) + threadIdx.x; idx < total; idx = total) {
        const size_t poly = idx >> logn, k = idx & (n - 1);
        const W* src = in + poly * L * n + k;
        uint64_t y[L];
#pragma unroll
        for (int i = 0; i < L; ++i)
            y[i] = A::shoup(stream_load(src + i * n), tool.scale_round_scale[i], tool.q_moduli[i].p);
        uint64_t converted[2];  // (gamma t x) converted to base [t, gamma], times -(Q^-1)          :279-282
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const DeviceModulus m = tool.t_gamma[j];
            typename A::Sum sum = A::first(y[0], tool.q_to_t_gamma[j * L + 0]);
#pragma unroll
            for (int i = 1; i < L; ++i) A::add(sum, y[i], tool.q_to_t_gamma[j * L + i]);
            converted[j] = A::shoup(A::reduce(sum, m), tool.neg_inv_q_mod_t_gamma[j], m.p);
        }
        const DeviceModulus t = tool.t_gamma[0];
        const uint64_t gamma = tool.t_gamma[1].p;
        const uint64_t mod_gamma = converted[1];
        // centred remainder mod gamma, taken mod t                                                :289-297
        const bool above = mod_gamma > (gamma >> 1);
        const uint64_t reduced = barrett_reduce64_uniform(above ? gamma - mod_gamma : mod_gamma, t.p, t.barrett64);
        const uint64_t s_gamma = above ? neg_mod_uniform(reduced, t.p) : reduced;
        stream_store(out + idx, A::shoup(sub_mod_uniform(converted[0], s_gamma, t.p), final_scale, t.p));  // :298-301
    }
}

// ---- plaintextTranslate: c0 +-= floor(Q/t) m + floor(((Q mod t) m + (t+1)/2) / t)      Bfv+Encrypt.swift:75-140 ----
// ct [batch][polys][L][N] Coeff (only c0 is touched), plaintexts [batch][N] with values < t.  The rounding term is the
// same for every residue row: its 128-bit dividend is divided by t with the double-word Barrett estimate (one below
// the quotient at worst) and one fix-up.
template <int L, typename W, bool SUBTRACT>
__global__ void __launch_bounds__(kThreads)
    plaintext_translate_kernel(W* __restrict__ ct, const W* __restrict__ plaintexts, const RnsToolDevice tool,
                               size_t ct_words, size_t batch) {
    using A = WordArith<W>;
    const uint32_t logn = tool.log_degree;
    const size_t n = size_t(1) << logn;
    const size_t total = batch << logn;
    const size_t idx = blockIdx.x * size_t(kThreads) + threadIdx.x;
    if (idx >= total) return;
    const size_t item = idx >> logn, k = idx & (n - 1);
    const uint64_t m = stream_load(plaintexts + idx);
    const DeviceModulus t = tool.t_gamma[0];
    U128 x = mul_wide(tool.q_mod_t, m);  // < t^2                                                   :92-107
    const uint64_t threshold = (t.p + 1) >> 1;  // RnsTool.swift:123-125
    x.lo += threshold;
    x.hi += x.lo < threshold ? 1 : 0;
    const uint64_t ll_hi = mulhi64(x.lo, t.barrett128_lo);
    const U128 lh = mul_wide(x.lo, t.barrett128_hi), hl = mul_wide(x.hi, t.barrett128_lo);
    const uint64_t mid = ll_hi + lh.lo, mid2 = mid + hl.lo;
    const uint64_t estimate = lh.hi + hl.hi + (mid < ll_hi ? 1 : 0) + (mid2 < mid ? 1 : 0) + x.hi * t.barrett128_hi;
    const uint64_t adjust = estimate + ((x.lo - estimate * t.p) >= t.p ? 1 : 0);  // the quotient is below t: one word
    W* c0 = ct + item * ct_words + k;
#pragma unroll
    for (int i = 0; i < L; ++i) {
        const uint64_t p = tool.q_moduli[i].p;
        const uint64_t round_q_times_mt = add_mod_uniform(A::shoup(m, tool.q_div_t[i], p), adjust, p);   // :117-120
        const uint64_t c = stream_load(c0 + i * n);
        stream_store(c0 + i * n, SUBTRACT ? sub_mod_uniform(c, round_q_times_mt, p) : add_mod_uniform(c, round_q_times_mt, p));
    }
}

// ---- tensor product: (a0, a1) x (b0, b1) -> (a0 b0, a0 b1 + a1 b0, a1 b1), word-wise in Eval form ----------------
// in: [items][4][rows][N] (a0, a1, b0, b1); out: [items][3][rows][N]
// blockIdx.y = item * rows + row: the modulus constants are wave-uniform
template <typename W>
__global__ void __launch_bounds__(kThreads)
    tensor_kernel(const W* __restrict__ in, W* __restrict__ out, const DeviceContext ctx) {
    const uint32_t logn = ctx.log_degree;
    const size_t n = size_t(1) << logn;
    const size_t k = blockIdx.x * size_t(kThreads) + threadIdx.x;
    if (k >= n) return;
    const size_t poly_words = size_t(ctx.moduli_count) << logn;
    const size_t item = blockIdx.y / ctx.moduli_count;
    const uint32_t row = blockIdx.y - static_cast<uint32_t>(item) * ctx.moduli_count;
    const DeviceModulus m = ctx.moduli[row];
    const W* src = in + item * 4 * poly_words + (size_t(row) << logn) + k;
    const uint64_t a0 = src[0], a1 = src[poly_words], b0 = src[2 * poly_words], b1 = src[3 * poly_words];
    using A = WordArith<W>;
    W* dst = out + item * 3 * poly_words + (size_t(row) << logn) + k;
    dst[0] = static_cast<W>(A::mul(a0, b0, m));
    if constexpr (sizeof(W) == 4) {  // two products below 2^60: one sum, one reduction (the same canonical word)
        typename A::Sum cross = A::first(a0, b1);
        A::add_vector(cross, a1, b0);
        dst[poly_words] = static_cast<W>(A::reduce(cross, m));
    } else {
        dst[poly_words] = static_cast<W>(add_mod_uniform(A::mul(a0, b1, m), A::mul(a1, b0, m), m.p));
    }
    dst[2 * poly_words] = static_cast<W>(A::mul(a1, b1, m));
}

// ---- lazy tensor accumulation for Bfv.innerProduct(ct, ct) (Bfv.swift:315-361) -----------------------------------
// in: [count][4][rows][N]; out: [3][rows][N] = reduce(sum_k tensor_k)
template <typename W>
__global__ void __launch_bounds__(kThreads)
    tensor_accumulate_kernel(const W* __restrict__ in, W* __restrict__ out, const DeviceContext ctx, size_t count,
                             uint64_t max_lazy) {
    const uint32_t logn = ctx.log_degree;
    const size_t poly_words = size_t(ctx.moduli_count) << logn;
    for (size_t w = blockIdx.x * size_t(kThreads) + threadIdx.x; w < poly_words; w += size_t(gridDim.x) * kThreads) {
        const DeviceModulus m = ctx.moduli[w >> logn];
        if constexpr (sizeof(W) == 4) {
            // products below 2^60: 64-bit sums, folded every kNarrowTerms items (the middle sum takes two per item)
            uint64_t s0 = 0, s1 = 0, s2 = 0, since32 = 0;
            for (size_t item = 0; item < count; ++item) {
                const W* src = in + item * 4 * poly_words + w;
                const uint32_t a0 = src[0], a1 = src[poly_words], b0 = src[2 * poly_words], b1 = src[3 * poly_words];
                s0 = mad32(a0, b0, s0);
                s1 = mad32(a0, b1, mad32(a1, b0, s1));
                s2 = mad32(a1, b1, s2);
                if (++since32 >= kNarrowTerms) {
                    since32 = 0;
                    s0 = barrett_reduce64_uniform(s0, m.p, m.barrett64);
                    s1 = barrett_reduce64_uniform(s1, m.p, m.barrett64);
                    s2 = barrett_reduce64_uniform(s2, m.p, m.barrett64);
                }
            }
            out[w] = static_cast<W>(barrett_reduce64_uniform(s0, m.p, m.barrett64));
            out[poly_words + w] = static_cast<W>(barrett_reduce64_uniform(s1, m.p, m.barrett64));
            out[2 * poly_words + w] = static_cast<W>(barrett_reduce64_uniform(s2, m.p, m.barrett64));
            continue;
        }
        U128 d0{0, 0}, d1{0, 0}, d2{0, 0};
        uint64_t since = 0;
        for (size_t item = 0; item < count; ++item) {
            const W* src = in + item * 4 * poly_words + w;
            const uint64_t a0 = src[0], a1 = src[poly_words], b0 = src[2 * poly_words], b1 = src[3 * poly_words];
            mac128(d0, a0, b0);
            mac128(d1, a0, b1);
            mac128(d1, a1, b0);
            mac128(d2, a1, b1);
            if (++since >= max_lazy) {  // reduceInPlace cadence, Bfv.swift:349-353
                since = 0;
                d0 = U128{reduce128(d0, m), 0};
                d1 = U128{reduce128(d1, m), 0};
                d2 = U128{reduce128(d2, m), 0};
            }
        }
        out[w] = static_cast<W>(reduce128(d0, m));
        out[poly_words + w] = static_cast<W>(reduce128(d1, m));
        out[2 * poly_words + w] = static_cast<W>(reduce128(d2, m));
    }
}

// The same for `items` inner products that share their left vector (the PIR remaining-dimension step of every
// result group of every chunk, PirUtil.swift:448-479): lhs [count][2][rows][N], rhs [items][count][2][rows][N] (both
// lifted, Eval) -> out [items][3][rows][N].  blockIdx.y = item.
template <typename W>
__global__ void __launch_bounds__(kThreads)
    tensor_accumulate_shared_kernel(const W* __restrict__ lhs, const W* __restrict__ rhs, W* __restrict__ out,
                                    const DeviceContext ctx, size_t count, uint64_t max_lazy) {
    const uint32_t logn = ctx.log_degree;
    const size_t poly_words = size_t(ctx.moduli_count) << logn;
    const size_t item = blockIdx.y;
    const W* __restrict__ right = rhs + item * count * 2 * poly_words;
    W* __restrict__ sum = out + item * 3 * poly_words;
    for (size_t w = blockIdx.x * size_t(kThreads) + threadIdx.x; w < poly_words; w += size_t(gridDim.x) * kThreads) {
        const DeviceModulus m = ctx.moduli[w >> logn];
        if constexpr (sizeof(W) == 4) {
            uint64_t s0 = 0, s1 = 0, s2 = 0, since32 = 0;
            for (size_t k = 0; k < count; ++k) {
                const W* a = lhs + k * 2 * poly_words + w;
                const W* b = right + k * 2 * poly_words + w;
                const uint32_t a0 = a[0], a1 = a[poly_words], b0 = b[0], b1 = b[poly_words];
                s0 = mad32(a0, b0, s0);
                s1 = mad32(a0, b1, mad32(a1, b0, s1));
                s2 = mad32(a1, b1, s2);
                if (++since32 >= kNarrowTerms) {
                    since32 = 0;
                    s0 = barrett_reduce64_uniform(s0, m.p, m.barrett64);
                    s1 = barrett_reduce64_uniform(s1, m.p, m.barrett64);
                    s2 = barrett_reduce64_uniform(s2, m.p, m.barrett64);
                }
            }
            sum[w] = static_cast<W>(barrett_reduce64_uniform(s0, m.p, m.barrett64));
            sum[poly_words + w] = static_cast<W>(barrett_reduce64_uniform(s1, m.p, m.barrett64));
            sum[2 * poly_words + w] = static_cast<W>(barrett_reduce64_uniform(s2, m.p, m.barrett64));
            continue;
        }
        U128 d0{0, 0}, d1{0, 0}, d2{0, 0};
        uint64_t since = 0;
        for (size_t k = 0; k < count; ++k) {
            const W* a = lhs + k * 2 * poly_words + w;
            const W* b = right + k * 2 * poly_words + w;
            const uint64_t a0 = a[0], a1 = a[poly_words], b0 = b[0], b1 = b[poly_words];
            mac128(d0, a0, b0);
            mac128(d1, a0, b1);
            mac128(d1, a1, b0);
            mac128(d2, a1, b1);
            if (++since >= max_lazy) {  // reduceInPlace cadence, Bfv.swift:349-353
                since = 0;
                d0 = U128{reduce128(d0, m), 0};
                d1 = U128{reduce128(d1, m), 0};
                d2 = U128{reduce128(d2, m), 0};
            }
        }
        sum[w] = static_cast<W>(reduce128(d0, m));
        sum[poly_words + w] = static_cast<W>(reduce128(d1, m));
        sum[2 * poly_words + w] = static_cast<W>(reduce128(d2, m));
    }
}

// The same sums on the carry-counting accumulator (device_math.hpp ProductSum: 7 instructions per product against ~18 for a
// 128-bit multiply-add -- at 64 terms the 128-bit form was bound by the multiplier, 177 us where the operands stream in 100),
// 8-byte words.  `cadence` terms at most between folds, chosen by the launcher so that the middle sum (two products per term)
// stays below 2^127 (reduce_product_sum's contract); the canonical result does not depend on it.
// rhs_q != nullptr: rows [0, q_rows) of the right-hand polynomials are read from there instead -- [items][count][2][q_rows][N],
// the Eval-form ciphertexts a caller already holds (bfv_api.cpp bfv_inner_product_shared_eval_rhs) -- and the lifted records'
// own first q_rows rows are not touched.
__global__ void __launch_bounds__(kThreads)
    tensor_accumulate_shared_sums_kernel(const uint64_t* __restrict__ lhs, const uint64_t* __restrict__ rhs,
                                         const uint64_t* __restrict__ rhs_q, uint32_t q_rows, uint64_t* __restrict__ out,
                                         const DeviceContext ctx, size_t count, uint64_t cadence) {
    const uint32_t logn = ctx.log_degree;
    const size_t poly_words = size_t(ctx.moduli_count) << logn;
    const size_t item = blockIdx.y;
    const size_t w = blockIdx.x * size_t(kThreads) + threadIdx.x;  // (the grid covers poly_words exactly: degree >= kThreads)
    const uint32_t row = static_cast<uint32_t>((blockIdx.x * size_t(kThreads)) >> logn);  // uniform
    const DeviceModulus m = ctx.moduli[row];
    const bool from_q = rhs_q != nullptr && row < q_rows;
    const size_t right_poly = from_q ? size_t(q_rows) << logn : poly_words;
    const uint64_t* __restrict__ a = lhs + w;
    const uint64_t* __restrict__ b = (from_q ? rhs_q : rhs) + item * count * 2 * right_poly + w;
    ProductSum d0 = product_sum_zero(), d1 = product_sum_zero(), d2 = product_sum_zero();
    // the operands of kAhead terms in flight.  (Two measured slower: 66 registers, 156 against 145 us for the 8 x 64 terms of the
    // PIR tail -- which streams 604 MB, so 100 us is the floor; the 128-bit sums took 177 us at 74 registers.)
    constexpr int kAhead = 1;
    uint64_t a0[kAhead], a1[kAhead], b0[kAhead], b1[kAhead];
    auto fetch = [&](int slot, size_t k) {
        const size_t term = k < count ? k : count - 1;  // (past the end: the last term again -- no branch around the loads)
        a0[slot] = a[term * 2 * poly_words];
        a1[slot] = a[term * 2 * poly_words + poly_words];
        b0[slot] = __builtin_nontemporal_load(b + term * 2 * right_poly);
        b1[slot] = __builtin_nontemporal_load(b + term * 2 * right_poly + right_poly);
    };
#pragma unroll
    for (int slot = 0; slot < kAhead; ++slot) fetch(slot, slot);
    uint64_t since = 0;
    for (size_t k = 0; k < count; k += kAhead) {
#pragma unroll
        for (int slot = 0; slot < kAhead; ++slot) {
            const uint64_t x0 = a0[slot], x1 = a1[slot], y0 = b0[slot], y1 = b1[slot];
            fetch(slot, k + slot + kAhead);
            if (k + slot < count) {  // (uniform)
                product_sum_add_pair<false>(d0, d1, x0, x1, y0);  // d0 += a0 b0, d1 += a1 b0
                product_sum_add_pair<false>(d1, d2, x0, x1, y1);  // d1 += a0 b1, d2 += a1 b1
                if (++since >= cadence) {
                    since = 0;
                    d0 = ProductSum{reduce_product_sum(d0, m), 0, 0, 0, 0};
                    d1 = ProductSum{reduce_product_sum(d1, m), 0, 0, 0, 0};
                    d2 = ProductSum{reduce_product_sum(d2, m), 0, 0, 0, 0};
                }
            }
        }
    }
    uint64_t* __restrict__ sum = out + item * 3 * poly_words + w;
    sum[0] = reduce_product_sum(d0, m);
    sum[poly_words] = reduce_product_sum(d1, m);
    sum[2 * poly_words] = reduce_product_sum(d2, m);
}

// ---- key switching, step 1: decompose-and-spread (Bfv+Keys.swift:165-172) ----------------------------------------
// target: row j of polynomial `poly` at  target_base + poly * target_stride + j * N   (Coeff, mod q_j)
// out: [polys][L][L+1][N]: word (poly, j, r, k) = target[j][k] mod ks_modulus[r]  (reduced only when q_j > modulus r)
template <typename W>
__global__ void __launch_bounds__(kThreads)
    key_switch_spread_kernel(const W* __restrict__ target_base, size_t target_stride, W* __restrict__ out,
                             const DeviceContext ks, uint32_t L, size_t polys) {
    const uint32_t logn = ks.log_degree;
    const size_t n = size_t(1) << logn;
    const size_t total = (polys * L) << logn;
    for (size_t idx = blockIdx.x * size_t(kThreads) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * kThreads) {
        const size_t k = idx & (n - 1);
        const size_t pj = idx >> logn;
        const size_t poly = pj / L, j = pj - poly * L;
        const uint64_t x = target_base[poly * target_stride + j * n + k];
        const uint64_t qj = ks.moduli[j].p;
        W* dst = out + (pj * (L + 1)) * n + k;
        for (uint32_t r = 0; r <= L; ++r) {
            const DeviceModulus m = ks.moduli[r];
            dst[r * n] = static_cast<W>(qj > m.p ? barrett_reduce64(x, m.p, m.barrett64) : x);
        }
    }
}

// ---- key switching, step 2: lazy inner product with the key (Bfv+Keys.swift:180-202) ------------------------------
// spread: [polys][L][L+1][N] (Eval); key: [L_top][2][L_top+1][N]; out: [polys][2][L+1][N]
//   out[poly][c][r][k] = ( sum_j spread[poly][j][r][k] * key[j][c][key_row(r)][k] ) mod ks_modulus[r]
// blockIdx.y = poly * (L+1) + r, so the modulus, the key row and every address base are wave-uniform (SGPRs) and the
// sums ride the carry-counting product accumulator (device_math.hpp ProductSum: L <= 8 products < 2^127).
template <typename W>
__global__ void __launch_bounds__(kThreads)
    key_switch_mac_kernel(const W* __restrict__ spread, const W* __restrict__ key, W* __restrict__ out,
                          const DeviceContext ks, uint32_t L, uint32_t top_rows) {
    const uint32_t logn = ks.log_degree;
    const size_t n = size_t(1) << logn;
    const size_t k = blockIdx.x * size_t(kThreads) + threadIdx.x;
    if (k >= n) return;
    const size_t pr = blockIdx.y;
    const size_t poly = pr / (L + 1);
    const uint32_t r = static_cast<uint32_t>(pr - poly * (L + 1));
    const uint32_t key_row = (r == L) ? top_rows - 1 : r;  // Bfv+Keys.swift:153
    const DeviceModulus m = ks.moduli[r];
    const W* __restrict__ x_row = spread + ((poly * L) * (L + 1) + r) * n + k;
    const W* __restrict__ key_row0 = key + size_t(key_row) * n + k;
    using A = WordArith<W>;
    typename A::Sum acc0 = A::zero(), acc1 = A::zero();
    for (uint32_t j = 0; j < L; ++j) {
        const uint64_t x = x_row[size_t(j) * (L + 1) * n];
        const W* key_j = key_row0 + size_t(j) * 2 * top_rows * n;
        A::add_vector(acc0, x, key_j[0]);
        A::add_vector(acc1, x, key_j[size_t(top_rows) * n]);
    }
    W* dst = out + (poly * 2 * (L + 1) + r) * n + k;
    dst[0] = static_cast<W>(A::reduce(acc0, m));
    dst[size_t(L + 1) * n] = static_cast<W>(A::reduce(acc1, m));
}

// ---- key switching, step 4: drop the special modulus and add into the ciphertext (Bfv.swift:216-217) --------------
// prod: [polys][2][L+1][N] Coeff over (q_0..q_{L-1}, q_ks); ct: poly c of item at ct_base + item*ct_stride + c*L*N;
// out: [polys][2][L][N] = ct + divideAndRoundQLast(prod)
// MODE kFinishGalois: the ciphertext is the one BEFORE the automorphism and its c0 term is read through it here
// (Bfv.swift:190-196: c0' = galois(c0) + update0, c1' = update1) -- coefficient k takes source coefficient
// i = k g^-1 mod 2N, negated when i >= N (PolyRq/Galois.swift:115-143).
// MODE kFinishExpand: on top of that, one step of PirUtil.expand (PirUtil.swift:204-236): with c' = applyGalois(ct),
// out [2 polys][2][L][N] holds the children ct + c' and (ct - c') x^shift, interleaved; the second one is written
// where its coefficient lands (k + shift mod 2N, negated past N) instead of being gathered by another kernel.  With
// `targets` the children are leaves of the expansion and go straight to their output slots (ExpandTargets).
// `own_base` (kFinishExpand): the ciphertext the children are formed WITH -- ct_base itself when the level's element has
// its own key; when the element is reached by applying a smaller one several times (PirUtil.swift:221-231), ct_base is
// the result of the applications before the last one (what this key switch rotates) and own_base the level's parent.
constexpr int kFinishPlain = 0, kFinishGalois = 1, kFinishExpand = 2;
template <typename W, int MODE>
__global__ void __launch_bounds__(kThreads)
    key_switch_finish_kernel(const W* __restrict__ prod, const W* __restrict__ ct_base, size_t ct_stride,
                             W* __restrict__ out, const DeviceContext ks, uint32_t L, size_t polys,
                             uint32_t added_polys, uint32_t galois_inverse, uint32_t expand_shift,
                             const ExpandTargets targets, const W* __restrict__ own_base) {
    const uint32_t logn = ks.log_degree;
    const size_t n = size_t(1) << logn;
    const size_t total = (polys * 2) << logn;
    const uint64_t q_last = ks.moduli[L].p, q_last_div2 = q_last >> 1;
    const U64x2* __restrict__ inverse_q_last = ks.inverse_q_last + size_t(L) * ks.moduli_stride;
    for (size_t idx = blockIdx.x * size_t(kThreads) + threadIdx.x; idx < total; idx += size_t(gridDim.x) * kThreads) {
        const size_t k = idx & (n - 1);
        const size_t pc = idx >> logn;  // poly * 2 + c
        const size_t poly = pc >> 1, c = pc & 1;
        const W* src = prod + pc * (L + 1) * n + k;
        const W* ct_poly = ct_base + poly * ct_stride + c * L * n;
        const W* ct = ct_poly + k;
        // the ciphertext(s) this polynomial's words go to
        size_t first_ct = MODE == kFinishExpand ? 2 * poly : poly, second_ct = 2 * poly + 1;
        [[maybe_unused]] bool first_doubled = false, second_doubled = false;
        if constexpr (MODE == kFinishExpand) {
            if (targets.table != nullptr) {
                const size_t group = poly / targets.group_size, parent = poly - group * targets.group_size;
                const uint32_t first = targets.table[4 * parent + 1], second = targets.table[4 * parent + 3];
                first_ct = group * targets.group_stride + (first >> 1);
                second_ct = group * targets.group_stride + (second >> 1);
                first_doubled = (first & 1u) != 0;
                second_doubled = (second & 1u) != 0;
            }
        }
        W* dst = out + (first_ct * 2 + c) * L * n + k;
        [[maybe_unused]] W* moved_dst = out + (second_ct * 2 + c) * L * n;
        // the automorphism's source coefficient and sign for this k (used for c0 only)
        [[maybe_unused]] uint32_t galois_source = 0;
        [[maybe_unused]] bool galois_negate = false;
        if constexpr (MODE != kFinishPlain) {
            const uint32_t doubled = (static_cast<uint32_t>(k) * galois_inverse) & static_cast<uint32_t>(2 * n - 1);
            galois_negate = doubled >= n;
            galois_source = doubled & static_cast<uint32_t>(n - 1);
        }
        // where coefficient k of (ct - c') x^shift lands, and whether it changes sign on the way
        [[maybe_unused]] size_t moved = 0;
        [[maybe_unused]] bool moved_negate = false;
        if constexpr (MODE == kFinishExpand) {
            const uint32_t doubled = (static_cast<uint32_t>(k) + expand_shift) & static_cast<uint32_t>(2 * n - 1);
            moved_negate = doubled >= n;
            moved = doubled & static_cast<uint32_t>(n - 1);
        }
        // divideAndRoundQLast by the centred representative of the special-modulus word (poly_kernels.hip has the
        // derivation): out_i = (x_i - c) q_ks^-1 mod q_i
        const uint64_t r = add_mod_uniform(stream_load(src + size_t(L) * n), q_last_div2, q_last);
        const bool negative = r < q_last_div2;
        const uint64_t magnitude = negative ? q_last_div2 - r : r - q_last_div2;
        for (uint32_t row = 0; row < L; ++row) {
            const DeviceModulus m = ks.moduli[row];
            const U64x2 inv = inverse_q_last[row];
            const uint64_t t = barrett_reduce64_uniform(magnitude, m.p, m.barrett64);
            const uint64_t x = stream_load(src + row * n);
            const uint64_t v = WordArith<W>::shoup(negative ? add_mod_uniform(x, t, m.p) : sub_mod_uniform(x, t, m.p), inv, m.p);
            if constexpr (MODE == kFinishPlain) {
                // relinearize adds the update to (c0, c1) (Bfv.swift:216-217); applyGalois adds it to c0 only and
                // replaces c1 (Bfv.swift:194-195)
                stream_store(dst + row * n, c < added_polys ? add_mod_uniform(stream_load(ct + row * n), v, m.p) : v);
            } else {
                uint64_t rotated = v;  // c' = applyGalois(ct), polynomial c, coefficient k
                if (c == 0) {
                    const uint64_t word = ct_poly[row * n + galois_source];
                    rotated = add_mod_uniform(galois_negate ? neg_mod_uniform(word, m.p) : word, v, m.p);
                }
                if constexpr (MODE == kFinishGalois) {
                    stream_store(dst + row * n, rotated);
                } else {
                    const uint64_t own = own_base[poly * ct_stride + c * L * n + row * n + k];
                    const uint64_t sum = add_mod_uniform(own, rotated, m.p);
                    stream_store(dst + row * n, first_doubled ? add_mod_uniform(sum, sum, m.p) : sum);
                    uint64_t difference = sub_mod_uniform(own, rotated, m.p);
                    difference = moved_negate ? neg_mod_uniform(difference, m.p) : difference;
                    stream_store(moved_dst + row * n + moved,
                                 second_doubled ? add_mod_uniform(difference, difference, m.p) : difference);
                }
            }
        }
    }
}

template <template <int> class Launcher, typename... Args>
hipError_t dispatch_L(uint32_t L, Args&&... args) {
    switch (L) {
        case 1: return Launcher<1>::run(args...);
        case 2: return Launcher<2>::run(args...);
        case 3: return Launcher<3>::run(args...);
        case 4: return Launcher<4>::run(args...);
        case 5: return Launcher<5>::run(args...);
        case 6: return Launcher<6>::run(args...);
        case 7: return Launcher<7>::run(args...);
        case 8: return Launcher<8>::run(args...);
        case 9: return Launcher<9>::run(args...);
        case 10: return Launcher<10>::run(args...);
        case 11: return Launcher<11>::run(args...);
        case 12: return Launcher<12>::run(args...);
        case 13: return Launcher<13>::run(args...);
        case 14: return Launcher<14>::run(args...);
        case 15: return Launcher<15>::run(args...);
        case 16: return Launcher<16>::run(args...);
        default: return hipErrorNotSupported;
    }
}
static_assert(kMaxL == 16, "dispatch_L covers 1..16");

template <int L>
struct LiftLauncher {
    template <typename W>
    static hipError_t run(const W* in, W* out, const RnsToolDevice& tool, size_t polys, const LiftLayout& layout,
                          hipStream_t s) {
        if (((polys << tool.log_degree) + kThreads - 1) / kThreads > 0x7fffffffull) return hipErrorInvalidValue;
        bool launched = false;
        if constexpr (sizeof(W) == 8) {
            if (tool.wide_reduce_ok != 0) {
                launched = true;
                bool pairs = false;
                if constexpr (kLiftPairs && L <= 6) {  // two coefficients per lane stay within 64 registers up to L = 6
                    pairs = quad_aligned(in) && quad_aligned(out) && layout.in_item_stride % 2 == 0 &&
                            layout.out_item_stride % 2 == 0 && tool.log_degree >= 1;
                    if (pairs)
                        hipLaunchKernelGGL((lift_kernel<L, W, true, 2>), dim3(exact_grid(polys << (tool.log_degree - 1))),
                                           dim3(kThreads), 0, s, in, out, tool, polys, layout);
                }
                if (!pairs)
                    hipLaunchKernelGGL((lift_kernel<L, W, true>), dim3(exact_grid(polys << tool.log_degree)), dim3(kThreads), 0,
                                       s, in, out, tool, polys, layout);
            }
        }
        if constexpr (sizeof(W) == 4 && L <= 8) {  // (beyond L = 8 four coefficients per lane no longer fit the register file)
            // four words per lane where every row starts on a 16-byte boundary (strides are multiples of N in practice)
            if (quad_aligned(in) && quad_aligned(out) && layout.in_item_stride % 4 == 0 && layout.out_item_stride % 4 == 0 &&
                tool.log_degree >= 2) {
                launched = true;
                hipLaunchKernelGGL((lift_kernel<L, W, false, 4>), dim3(exact_grid(polys << (tool.log_degree - 2))),
                                   dim3(kThreads), 0, s, in, out, tool, polys, layout);
            }
        }
        if (!launched)
            hipLaunchKernelGGL((lift_kernel<L, W, false>), dim3(exact_grid(polys << tool.log_degree)), dim3(kThreads), 0, s, in,
                               out, tool, polys, layout);
        return hipGetLastError();
    }
};
template <int L>
struct FloorLauncher {
    template <typename W>
    static hipError_t run(const W* in, W* out, const RnsToolDevice& tool, size_t polys, hipStream_t s) {
        if (((polys << tool.log_degree) + kThreads - 1) / kThreads > 0x7fffffffull) return hipErrorInvalidValue;
        bool launched = false;
        if constexpr (sizeof(W) == 8) {
            if (tool.wide_reduce_ok != 0) {
                launched = true;
                hipLaunchKernelGGL((floor_kernel<L, W, true>), dim3(exact_grid(polys << tool.log_degree)), dim3(kThreads), 0, s,
                                   in, out, tool, polys);
            }
        }
        if constexpr (sizeof(W) == 4 && L <= 8) {
            if (quad_aligned(in) && quad_aligned(out) && tool.log_degree >= 2) {
                launched = true;
                hipLaunchKernelGGL((floor_kernel<L, W, false, 4>), dim3(exact_grid(polys << (tool.log_degree - 2))),
                                   dim3(kThreads), 0, s, in, out, tool, polys);
            }
        }
        if (!launched)
            hipLaunchKernelGGL((floor_kernel<L, W, false>), dim3(exact_grid(polys << tool.log_degree)), dim3(kThreads), 0, s, in,
                               out, tool, polys);
        return hipGetLastError();
    }
};

template <int L>
struct PlaintextTranslateLauncher {
    template <typename W>
    static hipError_t run(W* ct, const W* plaintexts, const RnsToolDevice& tool, size_t ct_words, bool subtract, size_t batch,
                          hipStream_t s) {
        if (((batch << tool.log_degree) + kThreads - 1) / kThreads > 0x7fffffffull) return hipErrorInvalidValue;
        const dim3 grid(exact_grid(batch << tool.log_degree));
        if (subtract)
            hipLaunchKernelGGL((plaintext_translate_kernel<L, W, true>), grid, dim3(kThreads), 0, s, ct, plaintexts, tool,
                               ct_words, batch);
        else
            hipLaunchKernelGGL((plaintext_translate_kernel<L, W, false>), grid, dim3(kThreads), 0, s, ct, plaintexts, tool,
                               ct_words, batch);
        return hipGetLastError();
    }
};

template <int L>
struct ScaleAndRoundLauncher {
    template <typename W>
    static hipError_t run(const W* in, W* out, const RnsToolDevice& tool, U64x2 final_scale, size_t polys,
                          hipStream_t s) {
        if (((polys << tool.log_degree) + kThreads - 1) / kThreads > 0x7fffffffull) return hipErrorInvalidValue;
        hipLaunchKernelGGL((scale_and_round_kernel<L, W>), dim3(exact_grid(polys << tool.log_degree)), dim3(kThreads), 0,
                           s, in, out, tool, final_scale, polys);
        return hipGetLastError();
    }
};

}  // namespace

uint32_t rns_max_supported_moduli() { return kMaxL; }

template <typename W>
hipError_t launch_scale_and_round(const W* in, W* out, const RnsToolDevice& tool, U64x2 final_scale, size_t polys,
                                  hipStream_t stream) {
    if (polys == 0) return hipSuccess;
    return dispatch_L<ScaleAndRoundLauncher>(tool.L, in, out, tool, final_scale, polys, stream);
}

template <typename W>
hipError_t launch_plaintext_translate(W* ct, const W* plaintexts, const RnsToolDevice& tool, uint32_t poly_count,
                                      bool subtract, size_t batch, hipStream_t stream) {
    if (batch == 0) return hipSuccess;
    const size_t ct_words = (size_t(poly_count) * tool.L) << tool.log_degree;
    return dispatch_L<PlaintextTranslateLauncher>(tool.L, ct, plaintexts, tool, ct_words, subtract, batch, stream);
}

template <typename W>
hipError_t launch_lift_q_to_qbsk(const W* in, W* out, const RnsToolDevice& tool, size_t polys, hipStream_t stream) {
    if (polys == 0) return hipSuccess;
    const size_t n = size_t(1) << tool.log_degree;
    const LiftLayout layout{1, tool.L * n, (2 * size_t(tool.L) + 1) * n, 1, 0};
    return dispatch_L<LiftLauncher>(tool.L, in, out, tool, polys, layout, stream);
}

template <typename W>
hipError_t launch_lift_q_to_qbsk_strided(const W* in, W* out, const RnsToolDevice& tool, size_t items,
                                         size_t polys_per_item, size_t in_item_stride, size_t out_item_stride,
                                         size_t out_offset, hipStream_t stream, bool store_input, bool lazy_output) {
    if (items == 0 || polys_per_item == 0) return hipSuccess;
    const LiftLayout layout{polys_per_item, in_item_stride, out_item_stride, store_input ? 1u : 0u, lazy_output ? 1u : 0u};
    return dispatch_L<LiftLauncher>(tool.L, in, out + out_offset, tool, items * polys_per_item, layout, stream);
}

template <typename W>
hipError_t launch_lift_pair_q_to_qbsk_strided(const W* first, const W* second, W* out, const RnsToolDevice& tool, size_t items,
                                              size_t polys_per_item, size_t in_item_stride, size_t out_item_stride,
                                              size_t first_out_offset, size_t second_out_offset, hipStream_t stream,
                                              bool store_input, bool lazy_output) {
    if (items == 0 || polys_per_item == 0) return hipSuccess;
    LiftLayout layout{polys_per_item, in_item_stride, out_item_stride, store_input ? 1u : 0u, lazy_output ? 1u : 0u};
    layout.second = second;
    layout.first_polys = items * polys_per_item;
    layout.second_out_delta = static_cast<ptrdiff_t>(second_out_offset) - static_cast<ptrdiff_t>(first_out_offset);
    return dispatch_L<LiftLauncher>(tool.L, first, out + first_out_offset, tool, 2 * items * polys_per_item, layout, stream);
}

template <typename W>
hipError_t launch_floor_qbsk_to_q(const W* in, W* out, const RnsToolDevice& tool, size_t polys, hipStream_t stream) {
    if (polys == 0) return hipSuccess;
    return dispatch_L<FloorLauncher>(tool.L, in, out, tool, polys, stream);
}

template <typename W>
hipError_t launch_tensor(const W* in, W* out, const DeviceContext& qbsk, size_t items, hipStream_t stream) {
    if (items == 0) return hipSuccess;
    const size_t n = size_t(1) << qbsk.log_degree, rows = qbsk.moduli_count;
    const size_t max_items = 65535 / rows;  // grid.y carries (item, row)
    for (size_t done = 0; done < items; done += max_items) {
        const size_t now = items - done < max_items ? items - done : max_items;
        hipLaunchKernelGGL(tensor_kernel<W>, dim3(static_cast<unsigned>((n + kThreads - 1) / kThreads),
                                                  static_cast<unsigned>(now * rows)),
                           dim3(kThreads), 0, stream, in + done * 4 * rows * n, out + done * 3 * rows * n, qbsk);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

template <typename W>
hipError_t launch_tensor_accumulate(const W* in, W* out, const DeviceContext& qbsk, size_t count, uint64_t max_lazy,
                                    hipStream_t stream) {
    const size_t total = size_t(qbsk.moduli_count) << qbsk.log_degree;
    hipLaunchKernelGGL(tensor_accumulate_kernel<W>, dim3(grid_for(total)), dim3(kThreads), 0, stream, in, out, qbsk,
                       count, max_lazy);
    return hipGetLastError();
}

// the terms the carry-counting sums take between folds: the middle sum's 2 products per term below 2^127, starting from a
// folded residue (moduli: the context's, host copies); 0: a modulus too wide for them
uint64_t tensor_sums_cadence(const uint64_t* moduli, uint32_t count) {
    unsigned __int128 cadence = ~uint64_t(0);
    for (uint32_t i = 0; i < count; ++i) {
        const unsigned __int128 below = moduli[i] - 1;
        if (below == 0) continue;
        const unsigned __int128 limit = ((static_cast<unsigned __int128>(1) << 127) - moduli[i]) / (below * below) / 2;
        if (limit < cadence) cadence = limit;
    }
    return static_cast<uint64_t>(cadence);
}

hipError_t launch_tensor_accumulate_shared_sums(const uint64_t* lhs, const uint64_t* rhs, const uint64_t* rhs_q, uint32_t q_rows,
                                                uint64_t* out, const DeviceContext& qbsk, size_t count, size_t items,
                                                uint64_t cadence, hipStream_t stream) {
    if (items == 0) return hipSuccess;
    const size_t total = size_t(qbsk.moduli_count) << qbsk.log_degree;
    if (cadence == 0 || qbsk.degree < kThreads || total / kThreads >= (size_t(1) << 31)) return hipErrorNotSupported;
    const size_t right = rhs_q != nullptr ? size_t(q_rows) << qbsk.log_degree : 0;
    for (size_t done = 0; done < items; done += 65535) {  // grid.y carries the item
        const size_t now = items - done < 65535 ? items - done : 65535;
        hipLaunchKernelGGL(tensor_accumulate_shared_sums_kernel, dim3(static_cast<unsigned>(total / kThreads), static_cast<unsigned>(now)),
                           dim3(kThreads), 0, stream, lhs, rhs + done * count * 2 * total,
                           rhs_q != nullptr ? rhs_q + done * count * 2 * right : nullptr, q_rows, out + done * 3 * total, qbsk, count,
                           cadence);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

template <typename W>
hipError_t launch_tensor_accumulate_shared(const W* lhs, const W* rhs, W* out, const DeviceContext& qbsk, size_t count,
                                           size_t items, uint64_t max_lazy, hipStream_t stream) {
    if (items == 0) return hipSuccess;
    const size_t total = size_t(qbsk.moduli_count) << qbsk.log_degree;
    const size_t blocks = (total + kThreads - 1) / kThreads;
    for (size_t done = 0; done < items; done += 65535) {  // grid.y carries the item
        const size_t now = items - done < 65535 ? items - done : 65535;
        hipLaunchKernelGGL(tensor_accumulate_shared_kernel<W>, dim3(static_cast<unsigned>(blocks), static_cast<unsigned>(now)),
                           dim3(kThreads), 0, stream, lhs, rhs + done * count * 2 * total, out + done * 3 * total, qbsk,
                           count, max_lazy);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

template <typename W>
hipError_t launch_key_switch_spread(const W* target_base, size_t target_stride, W* out, const DeviceContext& ks,
                                    uint32_t L, size_t polys, hipStream_t stream) {
    if (polys == 0) return hipSuccess;
    hipLaunchKernelGGL(key_switch_spread_kernel<W>, dim3(grid_for((polys * L) << ks.log_degree)), dim3(kThreads), 0,
                       stream, target_base, target_stride, out, ks, L, polys);
    return hipGetLastError();
}

template <typename W>
hipError_t launch_key_switch_mac(const W* spread, const W* key, W* out, const DeviceContext& ks, uint32_t L,
                                 uint32_t top_rows, size_t polys, hipStream_t stream) {
    if (polys == 0) return hipSuccess;
    const size_t n = size_t(1) << ks.log_degree;
    if (L > kMaxL) return hipErrorInvalidValue;  // (ProductSum itself has headroom for 65 536 products)
    // grid.y carries (polynomial, modulus); it is limited to 65535, so long batches go out in slices
    const size_t rows_per_poly = L + 1, max_polys = 65535 / rows_per_poly;
    for (size_t done = 0; done < polys; done += max_polys) {
        const size_t now = polys - done < max_polys ? polys - done : max_polys;
        hipLaunchKernelGGL(key_switch_mac_kernel<W>,
                           dim3(static_cast<unsigned>((n + kThreads - 1) / kThreads), static_cast<unsigned>(now * rows_per_poly)),
                           dim3(kThreads), 0, stream, spread + done * L * rows_per_poly * n, key,
                           out + done * 2 * rows_per_poly * n, ks, L, top_rows);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

template <typename W>
hipError_t launch_key_switch_finish(const W* prod, const W* ct_base, size_t ct_stride, W* out, const DeviceContext& ks,
                                    uint32_t L, size_t polys, uint32_t added_polys, hipStream_t stream) {
    if (polys == 0) return hipSuccess;
    hipLaunchKernelGGL((key_switch_finish_kernel<W, kFinishPlain>), dim3(grid_for((polys * 2) << ks.log_degree)),
                       dim3(kThreads), 0, stream, prod, ct_base, ct_stride, out, ks, L, polys, added_polys, 0u, 0u,
                       ExpandTargets{nullptr, 1, 0}, ct_base);
    return hipGetLastError();
}

template <typename W>
hipError_t launch_galois_finish(const W* prod, const W* ct_base, size_t ct_stride, W* out, const DeviceContext& ks,
                                uint32_t L, size_t polys, uint32_t galois_inverse, uint32_t expand_shift,
                                const ExpandTargets& targets, hipStream_t stream, const W* own_base) {
    if (polys == 0) return hipSuccess;
    const dim3 grid(grid_for((polys * 2) << ks.log_degree));
    if (expand_shift != 0)
        hipLaunchKernelGGL((key_switch_finish_kernel<W, kFinishExpand>), grid, dim3(kThreads), 0, stream, prod, ct_base,
                           ct_stride, out, ks, L, polys, 1u, galois_inverse, expand_shift, targets,
                           own_base != nullptr ? own_base : ct_base);
    else
        hipLaunchKernelGGL((key_switch_finish_kernel<W, kFinishGalois>), grid, dim3(kThreads), 0, stream, prod, ct_base,
                           ct_stride, out, ks, L, polys, 1u, galois_inverse, 0u, ExpandTargets{nullptr, 1, 0}, ct_base);
    return hipGetLastError();
}

// the two slab word types of the library: Bfv<UInt64> and Bfv<UInt32>
#define HEAMD_INSTANTIATE_RNS(W)                                                                                          \
    template hipError_t launch_scale_and_round<W>(const W*, W*, const RnsToolDevice&, U64x2, size_t, hipStream_t);        \
    template hipError_t launch_plaintext_translate<W>(W*, const W*, const RnsToolDevice&, uint32_t, bool, size_t,         \
                                                      hipStream_t);                                                       \
    template hipError_t launch_lift_q_to_qbsk<W>(const W*, W*, const RnsToolDevice&, size_t, hipStream_t);                \
    template hipError_t launch_lift_q_to_qbsk_strided<W>(const W*, W*, const RnsToolDevice&, size_t, size_t, size_t,      \
                                                         size_t, size_t, hipStream_t, bool, bool);                        \
    template hipError_t launch_lift_pair_q_to_qbsk_strided<W>(const W*, const W*, W*, const RnsToolDevice&, size_t, size_t, \
                                                              size_t, size_t, size_t, size_t, hipStream_t, bool, bool);   \
    template hipError_t launch_floor_qbsk_to_q<W>(const W*, W*, const RnsToolDevice&, size_t, hipStream_t);               \
    template hipError_t launch_tensor<W>(const W*, W*, const DeviceContext&, size_t, hipStream_t);                        \
    template hipError_t launch_tensor_accumulate<W>(const W*, W*, const DeviceContext&, size_t, uint64_t, hipStream_t);   \
    template hipError_t launch_tensor_accumulate_shared<W>(const W*, const W*, W*, const DeviceContext&, size_t, size_t,  \
                                                           uint64_t, hipStream_t);                                        \
    template hipError_t launch_key_switch_spread<W>(const W*, size_t, W*, const DeviceContext&, uint32_t, size_t,         \
                                                    hipStream_t);                                                         \
    template hipError_t launch_key_switch_mac<W>(const W*, const W*, W*, const DeviceContext&, uint32_t, uint32_t,        \
                                                 size_t, hipStream_t);                                                    \
    template hipError_t launch_key_switch_finish<W>(const W*, const W*, size_t, W*, const DeviceContext&, uint32_t,       \
                                                    size_t, uint32_t, hipStream_t);                                       \
    template hipError_t launch_galois_finish<W>(const W*, const W*, size_t, W*, const DeviceContext&, uint32_t, size_t,   \
                                                uint32_t, uint32_t, const ExpandTargets&, hipStream_t, const W*);
HEAMD_INSTANTIATE_RNS(uint64_t)
HEAMD_INSTANTIATE_RNS(uint32_t)
#undef HEAMD_INSTANTIATE_RNS

}  // namespace heamd
