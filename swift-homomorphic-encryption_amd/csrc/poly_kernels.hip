// poly_kernels.hip -- element-wise PolyRq kernels, RNS modulus switching and lazy inner products for gfx950.
//
// All of these stream every word once (HBM-bound by design): lanes move 16 B each, consecutive lanes touch
// consecutive addresses, grids are sized to a few workgroups per CU and stride over the slab.
// Reference semantics (Sources/HomomorphicEncryption/):
//   PolyRq += -= prefix- *= (Eval) *= [T]      PolyRq/PolyRq.swift:147-174,184-204,232-245,299-309
//   PolyRq.divideAndRoundQLast                  PolyRq/PolyRq.swift:365-393
//   PolyRq.addingLazyProduct, Bfv.reduceToCiphertext, Bfv.innerProduct(ciphertexts:plaintexts:)
//                                               PolyRq/PolyRq.swift:210-225, Bfv/Bfv.swift:365-394,476-505
#include <hip/hip_runtime.h>

#include "device_context.hpp"
#include "device_math.hpp"
#include "kernels.hpp"
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

template <ElementwiseOp OP>
__device__ __forceinline__ uint64_t apply(uint64_t a, uint64_t b, const DeviceModulus& m, U64x2 scalar) {
    if constexpr (OP == ElementwiseOp::Add) return add_mod(a, b, m.p);
    if constexpr (OP == ElementwiseOp::Sub) return sub_mod(a, b, m.p);
    if constexpr (OP == ElementwiseOp::Neg) return neg_mod(a, m.p);
    if constexpr (OP == ElementwiseOp::Mul)
        return barrett_mul(a, b, m.p, m.product_factor, static_cast<int>(m.product_shift));
    if constexpr (OP == ElementwiseOp::MulScalar) return shoup_mul(a, scalar.x, scalar.y, m.p);
    return 0;
}

// One lane = one pair of consecutive words (degree >= 2), so a pair never straddles two rows.
template <ElementwiseOp OP>
__global__ void __launch_bounds__(kThreads)
    elementwise_kernel(uint64_t* __restrict__ lhs, const uint64_t* __restrict__ rhs, const DeviceContext ctx,
                       size_t pairs) {
    const uint32_t log_pairs_per_row = ctx.log_degree - 1;
    for (size_t i = blockIdx.x * static_cast<size_t>(kThreads) + threadIdx.x; i < pairs;
         i += static_cast<size_t>(gridDim.x) * kThreads) {
        const uint32_t mi = static_cast<uint32_t>((i >> log_pairs_per_row) % ctx.moduli_count);
        const DeviceModulus m = ctx.moduli[mi];
        U64x2 a = stream_load(reinterpret_cast<const U64x2*>(lhs) + i);
        U64x2 b = {0, 0};
        U64x2 scalar = {0, 0};
        if constexpr (OP == ElementwiseOp::MulScalar) {
            scalar = reinterpret_cast<const U64x2*>(rhs)[mi];
        } else if constexpr (OP != ElementwiseOp::Neg) {
            b = stream_load(reinterpret_cast<const U64x2*>(rhs) + i);
        }
        a.x = apply<OP>(a.x, b.x, m, scalar);
        a.y = apply<OP>(a.y, b.y, m, scalar);
        stream_store(reinterpret_cast<U64x2*>(lhs) + i, a);
    }
}

// degree 1 contexts (rows of a single word): scalar path
template <ElementwiseOp OP>
__global__ void __launch_bounds__(kThreads)
    elementwise_scalar_kernel(uint64_t* __restrict__ lhs, const uint64_t* __restrict__ rhs, const DeviceContext ctx,
                              size_t words) {
    for (size_t i = blockIdx.x * static_cast<size_t>(kThreads) + threadIdx.x; i < words;
         i += static_cast<size_t>(gridDim.x) * kThreads) {
        const uint32_t mi = static_cast<uint32_t>((i >> ctx.log_degree) % ctx.moduli_count);
        const DeviceModulus m = ctx.moduli[mi];
        U64x2 scalar = {0, 0};
        uint64_t b = 0;
        if constexpr (OP == ElementwiseOp::MulScalar) {
            scalar = reinterpret_cast<const U64x2*>(rhs)[mi];
        } else if constexpr (OP != ElementwiseOp::Neg) {
            b = rhs[i];
        }
        lhs[i] = apply<OP>(lhs[i], b, m, scalar);
    }
}

template <ElementwiseOp OP>
hipError_t launch_elementwise_op(uint64_t* lhs, const uint64_t* rhs, const DeviceContext& ctx, size_t rows,
                                 hipStream_t stream) {
    const size_t words = rows * ctx.degree;
    if (words == 0) return hipSuccess;
    if (ctx.degree >= 2) {
        const size_t pairs = words / 2;
        hipLaunchKernelGGL(elementwise_kernel<OP>, dim3(grid_for(pairs)), dim3(kThreads), 0, stream, lhs, rhs, ctx,
                           pairs);
    } else {
        hipLaunchKernelGGL(elementwise_scalar_kernel<OP>, dim3(grid_for(words)), dim3(kThreads), 0, stream, lhs, rhs,
                           ctx, words);
    }
    return hipGetLastError();
}

// ct [batch][polys][L][N] *= pt [batch][L][N]   (Bfv.mulAssign(EvalCiphertext, EvalPlaintext), Bfv.swift:120-129)
__global__ void __launch_bounds__(kThreads)
    mul_plain_kernel(uint64_t* __restrict__ ct, const uint64_t* __restrict__ pt, const DeviceContext ctx,
                     uint32_t poly_count, size_t pairs_per_poly, size_t batch) {
    const uint32_t log_pairs_per_row = ctx.log_degree - 1;
    const size_t total = pairs_per_poly * batch;
    for (size_t i = blockIdx.x * static_cast<size_t>(kThreads) + threadIdx.x; i < total;
         i += static_cast<size_t>(gridDim.x) * kThreads) {
        const size_t b = i / pairs_per_poly, k = i - b * pairs_per_poly;
        const uint32_t mi = static_cast<uint32_t>(k >> log_pairs_per_row);
        const DeviceModulus m = ctx.moduli[mi];
        const U64x2 y = stream_load(reinterpret_cast<const U64x2*>(pt) + i);
        for (uint32_t c = 0; c < poly_count; ++c) {
            U64x2* slot = reinterpret_cast<U64x2*>(ct) + (b * poly_count + c) * pairs_per_poly + k;
            U64x2 x = stream_load(slot);
            x.x = barrett_mul(x.x, y.x, m.p, m.product_factor, static_cast<int>(m.product_shift));
            x.y = barrett_mul(x.y, y.y, m.p, m.product_factor, static_cast<int>(m.product_shift));
            stream_store(slot, x);
        }
    }
}

// divideAndRoundQLast: one lane owns one coefficient pair (columns k, k+1) of one polynomial and walks the rows.
//   r      = (x_last + floor(q_last/2)) mod q_last
//   out_i  = ((x_i + (floor(q_last/2) mod q_i) - (r mod q_i)) mod q_i) * q_last^-1 mod q_i
__global__ void __launch_bounds__(kThreads)
    divide_and_round_q_last_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out,
                                   const DeviceContext ctx, uint32_t moduli_count, size_t polys) {
    const uint32_t pairs_per_row = ctx.degree >> 1;
    const size_t total = polys * pairs_per_row;
    const uint32_t last = moduli_count - 1;
    const uint64_t q_last = ctx.moduli[last].p;
    const uint64_t q_last_div2 = q_last >> 1;
    const U64x2* __restrict__ inverse_q_last = ctx.inverse_q_last + static_cast<size_t>(last) * ctx.moduli_stride;
    for (size_t i = blockIdx.x * static_cast<size_t>(kThreads) + threadIdx.x; i < total;
         i += static_cast<size_t>(gridDim.x) * kThreads) {
        const size_t poly = i / pairs_per_row;
        const uint32_t k = static_cast<uint32_t>(i - poly * pairs_per_row);
        const U64x2* src = reinterpret_cast<const U64x2*>(in) + poly * moduli_count * pairs_per_row + k;
        U64x2* dst = reinterpret_cast<U64x2*>(out) + poly * last * pairs_per_row + k;
        // r - floor(q_last/2) with r = (x_last + floor(q_last/2)) mod q_last is the centred representative c of x_last
        // (-q_last/2 <= c < q_last/2), so out_i = (x_i - c) q_last^-1 mod q_i: one reduction of |c| per row instead
        // of reducing r and floor(q_last/2) separately -- the same canonical word as the reference's three steps
        const U64x2 last_row = stream_load(src + static_cast<size_t>(last) * pairs_per_row);
        const uint64_t r0 = add_mod_uniform(last_row.x, q_last_div2, q_last);
        const uint64_t r1 = add_mod_uniform(last_row.y, q_last_div2, q_last);
        const bool negative0 = r0 < q_last_div2, negative1 = r1 < q_last_div2;
        const uint64_t magnitude0 = negative0 ? q_last_div2 - r0 : r0 - q_last_div2;
        const uint64_t magnitude1 = negative1 ? q_last_div2 - r1 : r1 - q_last_div2;
        for (uint32_t row = 0; row < last; ++row) {
            const DeviceModulus m = ctx.moduli[row];
            const U64x2 inv = inverse_q_last[row];
            U64x2 x = stream_load(src + static_cast<size_t>(row) * pairs_per_row);
            const uint64_t t0 = barrett_reduce64_uniform(magnitude0, m.p, m.barrett64);
            const uint64_t t1 = barrett_reduce64_uniform(magnitude1, m.p, m.barrett64);
            x.x = shoup_mul_uniform(negative0 ? add_mod_uniform(x.x, t0, m.p) : sub_mod_uniform(x.x, t0, m.p), inv.x,
                                    inv.y, m.p);
            x.y = shoup_mul_uniform(negative1 ? add_mod_uniform(x.y, t1, m.p) : sub_mod_uniform(x.y, t1, m.p), inv.x,
                                    inv.y, m.p);
            stream_store(dst + static_cast<size_t>(row) * pairs_per_row, x);
        }
    }
}

// The same with the number of rows known at compile time: all L loads of a coefficient pair are issued before the first one is
// used (the loop above has one in flight per lane, which at 8 waves per SIMD is half of what the memory system needs to stay
// busy: 0.70 of 8 TB/s at N = 16384, L = 6 against the copy's 0.78).
template <int L>
__global__ void __launch_bounds__(kThreads)
    divide_and_round_q_last_rows_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, const DeviceContext ctx,
                                        size_t polys) {
    constexpr int last = L - 1;
    const uint32_t pairs_per_row = ctx.degree >> 1;
    const size_t total = polys * pairs_per_row;
    const uint64_t q_last = ctx.moduli[last].p;
    const uint64_t q_last_div2 = q_last >> 1;
    const U64x2* __restrict__ inverse_q_last = ctx.inverse_q_last + static_cast<size_t>(last) * ctx.moduli_stride;
    for (size_t i = blockIdx.x * static_cast<size_t>(kThreads) + threadIdx.x; i < total;
         i += static_cast<size_t>(gridDim.x) * kThreads) {
        const size_t poly = i / pairs_per_row;
        const uint32_t k = static_cast<uint32_t>(i - poly * pairs_per_row);
        const U64x2* src = reinterpret_cast<const U64x2*>(in) + poly * L * pairs_per_row + k;
        U64x2* dst = reinterpret_cast<U64x2*>(out) + poly * last * pairs_per_row + k;
        U64x2 x[L];
#pragma unroll
        for (int row = 0; row < L; ++row) x[row] = stream_load(src + static_cast<size_t>(row) * pairs_per_row);
        const uint64_t r0 = add_mod_uniform(x[last].x, q_last_div2, q_last);
        const uint64_t r1 = add_mod_uniform(x[last].y, q_last_div2, q_last);
        const bool negative0 = r0 < q_last_div2, negative1 = r1 < q_last_div2;
        const uint64_t magnitude0 = negative0 ? q_last_div2 - r0 : r0 - q_last_div2;
        const uint64_t magnitude1 = negative1 ? q_last_div2 - r1 : r1 - q_last_div2;
#pragma unroll
        for (int row = 0; row < last; ++row) {
            const DeviceModulus m = ctx.moduli[row];
            const U64x2 inv = inverse_q_last[row];
            const uint64_t t0 = barrett_reduce64_uniform(magnitude0, m.p, m.barrett64);
            const uint64_t t1 = barrett_reduce64_uniform(magnitude1, m.p, m.barrett64);
            U64x2 y;
            y.x = shoup_mul_uniform(negative0 ? add_mod_uniform(x[row].x, t0, m.p) : sub_mod_uniform(x[row].x, t0, m.p), inv.x,
                                    inv.y, m.p);
            y.y = shoup_mul_uniform(negative1 ? add_mod_uniform(x[row].y, t1, m.p) : sub_mod_uniform(x[row].y, t1, m.p), inv.x,
                                    inv.y, m.p);
            stream_store(dst + static_cast<size_t>(row) * pairs_per_row, y);
        }
    }
}

// Ciphertext.modSwitchDownToSingle (Bfv.swift:163-171): divideAndRoundQLast from L moduli down to one, the L - 1 steps
// in registers -- in [polys][L][N] -> out [polys][1][N], each step the same words as the kernel above.
template <int L>
__global__ void __launch_bounds__(kThreads)
    mod_switch_down_to_single_kernel(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, const DeviceContext ctx,
                                     size_t polys) {
    const uint32_t pairs_per_row = ctx.degree >> 1;
    const size_t total = polys * pairs_per_row;
    for (size_t i = blockIdx.x * static_cast<size_t>(kThreads) + threadIdx.x; i < total;
         i += static_cast<size_t>(gridDim.x) * kThreads) {
        const size_t poly = i / pairs_per_row;
        const uint32_t k = static_cast<uint32_t>(i - poly * pairs_per_row);
        const U64x2* src = reinterpret_cast<const U64x2*>(in) + poly * L * pairs_per_row + k;
        U64x2 x[L];
#pragma unroll
        for (int row = 0; row < L; ++row) x[row] = stream_load(src + static_cast<size_t>(row) * pairs_per_row);
#pragma unroll
        for (int last = L - 1; last >= 1; --last) {
            const uint64_t q_last = ctx.moduli[last].p, q_last_div2 = q_last >> 1;
            const U64x2* __restrict__ inverse_q_last = ctx.inverse_q_last + static_cast<size_t>(last) * ctx.moduli_stride;
            const uint64_t r0 = add_mod_uniform(x[last].x, q_last_div2, q_last);
            const uint64_t r1 = add_mod_uniform(x[last].y, q_last_div2, q_last);
            const bool negative0 = r0 < q_last_div2, negative1 = r1 < q_last_div2;
            const uint64_t magnitude0 = negative0 ? q_last_div2 - r0 : r0 - q_last_div2;
            const uint64_t magnitude1 = negative1 ? q_last_div2 - r1 : r1 - q_last_div2;
#pragma unroll
            for (int row = 0; row < last; ++row) {
                const DeviceModulus m = ctx.moduli[row];
                const U64x2 inv = inverse_q_last[row];
                const uint64_t t0 = barrett_reduce64_uniform(magnitude0, m.p, m.barrett64);
                const uint64_t t1 = barrett_reduce64_uniform(magnitude1, m.p, m.barrett64);
                x[row].x = shoup_mul_uniform(negative0 ? add_mod_uniform(x[row].x, t0, m.p) : sub_mod_uniform(x[row].x, t0, m.p),
                                             inv.x, inv.y, m.p);
                x[row].y = shoup_mul_uniform(negative1 ? add_mod_uniform(x[row].y, t1, m.p) : sub_mod_uniform(x[row].y, t1, m.p),
                                             inv.x, inv.y, m.p);
            }
        }
        stream_store(reinterpret_cast<U64x2*>(out) + poly * pairs_per_row + k, x[0]);
    }
}

__global__ void __launch_bounds__(kThreads)
    adding_lazy_product_kernel(const uint64_t* __restrict__ lhs, const uint64_t* __restrict__ rhs,
                               uint64_t* __restrict__ acc, size_t words) {
    for (size_t i = blockIdx.x * static_cast<size_t>(kThreads) + threadIdx.x; i < words;
         i += static_cast<size_t>(gridDim.x) * kThreads) {
        U64x2 a = reinterpret_cast<U64x2*>(acc)[i];
        U128 sum = {a.x, a.y};
        mac128(sum, lhs[i], rhs[i]);
        a.x = sum.lo;
        a.y = sum.hi;
        reinterpret_cast<U64x2*>(acc)[i] = a;
    }
}

__global__ void __launch_bounds__(kThreads)
    reduce_accumulator_kernel(const uint64_t* __restrict__ acc, uint64_t* __restrict__ out, const DeviceContext ctx,
                              size_t words) {
    for (size_t i = blockIdx.x * static_cast<size_t>(kThreads) + threadIdx.x; i < words;
         i += static_cast<size_t>(gridDim.x) * kThreads) {
        const DeviceModulus m = ctx.moduli[(i >> ctx.log_degree) % ctx.moduli_count];
        const U64x2 a = reinterpret_cast<const U64x2*>(acc)[i];
        out[i] = barrett_reduce128(U128{a.x, a.y}, m.p, m.barrett128_lo, m.barrett128_hi);
    }
}

// Lazy inner product: each lane owns one coefficient pair of COLS output ciphertexts and streams the `count`
// (ciphertext, plaintext) pairs, accumulating in 128 bits (Bfv.swift:476-505).  The ciphertext words are shared by
// the COLS columns held in registers and, through L2/MALL, by the workgroups of neighbouring columns (blockIdx.x is
// the column group, so concurrently-resident workgroups read the same ciphertext tile).
template <int POLYS, int COLS, typename W>
__global__ void __launch_bounds__(kThreads)
    inner_product_plain_kernel(const W* __restrict__ cts, const W* __restrict__ pts,
                               const uint8_t* __restrict__ present, W* __restrict__ out,
                               const DeviceContext ctx, size_t count, size_t columns, uint64_t max_lazy) {
    // One lane = one word of COLS output columns (8-byte streams: the width the copy probe runs fastest at, and half
    // the accumulator registers of a 16-byte lane, so twice the waves hide the latency of the plaintext stream).
    const size_t words_per_poly = static_cast<size_t>(ctx.moduli_count) << ctx.log_degree;
    const size_t word = blockIdx.y * static_cast<size_t>(kThreads) + threadIdx.x;
    if (word >= words_per_poly) return;
    const size_t col0 = static_cast<size_t>(blockIdx.x) * COLS;
    const DeviceModulus m = ctx.moduli[word >> ctx.log_degree];
    U128 acc[COLS][POLYS];
#pragma unroll
    for (int c = 0; c < COLS; ++c)
#pragma unroll
        for (int q = 0; q < POLYS; ++q) acc[c][q] = U128{0, 0};
    uint64_t since_reduce[COLS];
    bool live[COLS];
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
        since_reduce[c] = 0;
        live[c] = col0 + c < columns;
    }
    // every stream is "uniform base + this lane's word": the uniform part stays in SGPRs.  A column past the end
    // re-reads the last real column (its products are never stored).
    const W* ct_base = cts + word;
    const W* pt_lane = pts + word;
    size_t pt_column[COLS];  // uniform word offsets of the columns' first plaintexts
#pragma unroll
    for (int c = 0; c < COLS; ++c) pt_column[c] = (live[c] ? col0 + c : columns - 1) * count * words_per_poly;

    // software pipeline: the loads of item j + 1 are in flight while item j is multiplied
    uint64_t x_next[POLYS], y_next[COLS];
    auto fetch = [&](size_t j) {
#pragma unroll
        for (int q = 0; q < POLYS; ++q) x_next[q] = ct_base[(j * POLYS + q) * words_per_poly];
#pragma unroll
        for (int c = 0; c < COLS; ++c) y_next[c] = pt_lane[pt_column[c] + j * words_per_poly];
    };
    if (count > 0) fetch(0);
    for (size_t j = 0; j < count; ++j) {
        uint64_t x[POLYS], y[COLS];
#pragma unroll
        for (int q = 0; q < POLYS; ++q) x[q] = x_next[q];
#pragma unroll
        for (int c = 0; c < COLS; ++c) y[c] = y_next[c];
        if (j + 1 < count) fetch(j + 1);
#pragma unroll
        for (int c = 0; c < COLS; ++c) {
            if (!live[c]) continue;
            if (present != nullptr && present[(col0 + c) * count + j] == 0) continue;  // nil plaintext, Bfv.swift:486-489
#pragma unroll
            for (int q = 0; q < POLYS; ++q) mac128(acc[c][q], x[q], y[c]);
            if (++since_reduce[c] >= max_lazy) {  // Bfv.swift:496-500 reduceInPlace cadence
                since_reduce[c] = 0;
#pragma unroll
                for (int q = 0; q < POLYS; ++q)
                    acc[c][q] = U128{barrett_reduce128(acc[c][q], m.p, m.barrett128_lo, m.barrett128_hi), 0};
            }
        }
    }
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
        if (!live[c]) continue;
#pragma unroll
        for (int q = 0; q < POLYS; ++q)
            out[((col0 + c) * POLYS + q) * words_per_poly + word] =
                static_cast<W>(barrett_reduce128(acc[c][q], m.p, m.barrett128_lo, m.barrett128_hi));
    }
}

// The lazy sums of the inner-product kernels by slab word type.  8-byte words: the carry-counting accumulator
// (device_math.hpp ProductSum), folded by reduce_product_sum.  4-byte words (Bfv<UInt32>: every modulus below 2^30): a
// product is below 2^60, so a sum is one multiply-add per term in a 64-bit word, folded at least every
// kWord32Cadence terms (15 x 2^60 + a folded residue stays below 2^64) by a single-word Barrett.
constexpr uint64_t kWord32Cadence = 15;
template <typename W, int POLYS, bool NARROW>
struct PlainSums {
    using Sum = ProductSum;
    static __device__ __forceinline__ Sum zero() { return product_sum_zero(); }
    static __device__ __forceinline__ void add_all(Sum (&s)[POLYS], const uint64_t (&x)[POLYS], uint64_t y) {
        product_sum_add_all<POLYS, NARROW>(s, x, y);
    }
    static __device__ __forceinline__ uint64_t reduce(const Sum& s, const DeviceModulus& m) { return reduce_product_sum(s, m); }
    static __device__ __forceinline__ Sum from_residue(uint64_t r) {
        Sum s = product_sum_zero();
        s.t = r;
        return s;
    }
};
template <int POLYS, bool NARROW>
struct PlainSums<uint32_t, POLYS, NARROW> {
    using Sum = uint64_t;
    static __device__ __forceinline__ Sum zero() { return 0; }
    static __device__ __forceinline__ void add_all(Sum (&s)[POLYS], const uint64_t (&x)[POLYS], uint64_t y) {
#pragma unroll
        for (int q = 0; q < POLYS; ++q) s[q] = mad32(lo32(x[q]), lo32(y), s[q]);
    }
    static __device__ __forceinline__ uint64_t reduce(const Sum& s, const DeviceModulus& m) {
        return barrett_reduce64_uniform(s, m.p, m.barrett64);
    }
    static __device__ __forceinline__ Sum from_residue(uint64_t r) { return r; }
};

// The same inner product on the carry-counting accumulator (device_math.hpp ProductSum: 4 multiply-adds and 3 carry
// counts per product against ~18 instructions for a 128-bit multiply-add) with the residue row -- hence the modulus
// -- wave-uniform: blockIdx.y covers kThreads words of ONE row (degree >= kThreads).  `cadence` products at most are
// summed between reductions, chosen by the launcher so that a sum stays below 2^127 (reduce_product_sum's contract)
// and never exceeds the reference's own lazy count (Bfv.swift:496-500); the canonical result does not depend on it.
// NARROW: every modulus is below 2^56 and `cadence` at most kNarrowProductSumCadence (device_math.hpp): the middle
// column's carry counts are not kept.
// MASKED: `present` is not null.  Every load is issued unconditionally (the item after the last one re-reads the last
// one), the mask byte of an item travels with its plaintext words one item ahead: the loads in flight at each point are
// then a fixed number and no wait drains the queue (a branch around a load makes the compiler wait for all of them).
// PACKED: `pts` holds packed residue rows (kernels.hpp PackedLayout, 8-byte slabs only): a lane fetches the two stream
// words its field may straddle and shifts the field out -- 6.875 bytes of database per word at 55 bits instead of 8.
template <int POLYS, int COLS, bool NARROW, bool MASKED, bool PACKED, typename W>
__global__ void __launch_bounds__(kThreads)
    inner_product_plain_rows_kernel(const W* __restrict__ cts, const W* __restrict__ pts,
                                    const uint8_t* __restrict__ present, W* __restrict__ out,
                                    const DeviceContext ctx, size_t count, size_t columns, uint64_t cadence,
                                    uint32_t column_groups, const PackedLayout packed) {
    static_assert(!PACKED || sizeof(W) == 8, "packed plaintexts are 8-byte slabs");
    const uint32_t logn = ctx.log_degree;
    const size_t words_per_poly = static_cast<size_t>(ctx.moduli_count) << logn;
    // the column groups of one word block stream the same ciphertext words: one replica set per word block, so that
    // they run on one XCD and all but the first read the ciphertexts from its L2 (placement.hpp)
    uint32_t word_block, column_group;
    locate_replica(blockIdx.x, static_cast<uint32_t>(words_per_poly / kThreads), column_groups, word_block, column_group);
    const size_t block_word = word_block * static_cast<size_t>(kThreads);
    const size_t word = block_word + threadIdx.x;
    const size_t col0 = static_cast<size_t>(column_group) * COLS;
    const DeviceModulus m = ctx.moduli[block_word >> logn];
    using Sums = PlainSums<W, POLYS, NARROW>;
    typename Sums::Sum acc[COLS][POLYS];
#pragma unroll
    for (int c = 0; c < COLS; ++c)
#pragma unroll
        for (int q = 0; q < POLYS; ++q) acc[c][q] = Sums::zero();
    uint64_t since_reduce[COLS];
    bool live[COLS];
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
        since_reduce[c] = 0;
        live[c] = col0 + c < columns;
    }
    const W* ct_base = cts + word;
    // one plaintext's words in `pts`, and this lane's place among them
    size_t pt_stride = words_per_poly;
    const W* pt_lane = pts + word;
    [[maybe_unused]] uint32_t field_shift = 0, first_word_lane = 0;
    // per wavefront and column: 64 stream words and the one a last field's second word index may name (never used)
    __shared__ uint64_t unpack_tile[PACKED ? kThreads / 64 : 1][PACKED ? COLS : 1][PACKED ? 66 : 1];
    [[maybe_unused]] uint64_t field_mask = ~uint64_t(0);
    if constexpr (PACKED) {
        // The 64 fields of a wavefront are exactly `width` stream words: lane l < width fetches word l of that span (the
        // others repeat the last one: no extra line, no divergent load) and every lane then takes the two words its field
        // may lie in from the wavefront's LDS slice -- the vector-memory path carries the packed bytes and nothing else.
        const uint32_t row = static_cast<uint32_t>(block_word >> logn), width = packed.width[row];
        const uint32_t lane = threadIdx.x & 63u;
        const uint32_t first_field = (static_cast<uint32_t>(word) & ((1u << logn) - 1u)) - lane;  // a multiple of 64
        const uint32_t bit = lane * width;
        pt_stride = packed.word_offset[packed.rows];
        pt_lane = pts + packed.word_offset[row] + size_t(first_field / 64) * width + (lane < width ? lane : width - 1);
        field_shift = bit & 63u;
        first_word_lane = bit >> 6;  // index of the field's first word in the wavefront's span (the one after it: + 1)
        field_mask = width == 64 ? ~uint64_t(0) : ((uint64_t(1) << width) - 1);
    }
    size_t pt_column[COLS], mask_column[COLS];
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
        const size_t column = live[c] ? col0 + c : columns - 1;  // a column past the end re-reads the last one
        pt_column[c] = column * count * pt_stride;
        mask_column[c] = column * count;
    }
    uint64_t x_next[POLYS], y_next[COLS];
    [[maybe_unused]] uint8_t mask_next[COLS];
    auto fetch = [&](size_t j) {
#pragma unroll
        for (int q = 0; q < POLYS; ++q) x_next[q] = ct_base[(j * POLYS + q) * words_per_poly];
#pragma unroll
        for (int c = 0; c < COLS; ++c) {  // the database is read once: streamed past the caches (non-temporal)
            // a packed span (64 fields = `width` words) does not end on a cache line: its neighbours re-read the line
            // it shares with them, which has to come from L2 then, not from HBM a second time
            if constexpr (PACKED) y_next[c] = pt_lane[pt_column[c] + j * pt_stride];
            else y_next[c] = __builtin_nontemporal_load(pt_lane + pt_column[c] + j * pt_stride);
            if constexpr (MASKED) mask_next[c] = present[mask_column[c] + j];
        }
    };
    if (count > 0) fetch(0);
    for (size_t j = 0; j < count; ++j) {
        uint64_t x[POLYS], y[COLS];
        [[maybe_unused]] uint8_t mask[COLS];
#pragma unroll
        for (int q = 0; q < POLYS; ++q) x[q] = x_next[q];
#pragma unroll
        for (int c = 0; c < COLS; ++c) {
            y[c] = y_next[c];
            if constexpr (PACKED) {
                // through the wavefront's slice of LDS: its `width` words in, every lane's two words out in one read;
                // the LDS serves a wavefront's instructions in order, the fences only pin the compiler
                uint64_t* span = unpack_tile[threadIdx.x >> 6][c];
                span[threadIdx.x & 63u] = y[c];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const uint64_t first = span[first_word_lane], second = span[first_word_lane + 1];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                // the bits(q) bits starting at bit field_shift of (first, second): two 32-bit funnel shifts over the
                // three double-words they touch
                const bool upper = field_shift >= 32;
                const uint32_t d0 = upper ? hi32(first) : lo32(first), d1 = upper ? lo32(second) : hi32(first),
                               d2 = upper ? hi32(second) : lo32(second);
                const uint32_t low = __builtin_amdgcn_alignbit(d1, d0, field_shift & 31u);
                const uint32_t high = __builtin_amdgcn_alignbit(d2, d1, field_shift & 31u);
                y[c] = pack64(low, high) & field_mask;
            }
            if constexpr (MASKED) mask[c] = mask_next[c];
        }
        fetch(j + 1 < count ? j + 1 : j);
#pragma unroll
        for (int c = 0; c < COLS; ++c) {
            bool active = live[c];
            if constexpr (MASKED) active = active && mask[c] != 0;  // nil plaintext, Bfv.swift:486-489
            if (!active) continue;
            Sums::add_all(acc[c], x, y[c]);
            if (++since_reduce[c] >= cadence) {
                since_reduce[c] = 0;
#pragma unroll
                for (int q = 0; q < POLYS; ++q) {
                    acc[c][q] = Sums::from_residue(Sums::reduce(acc[c][q], m));
                }
            }
        }
    }
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
        if (!live[c]) continue;
#pragma unroll
        for (int q = 0; q < POLYS; ++q)
            out[((col0 + c) * POLYS + q) * words_per_poly + word] = static_cast<W>(Sums::reduce(acc[c][q], m));
    }
}

// ---- wide ciphertexts: several queries' ciphertext vectors side by side over one database ---------------------------
// With POLYS >= 4 the single-lane register tile above runs out of registers before it runs out of memory pipe: its
// POLYS + COLS operand loads per POLYS x COLS products are 8-byte gathers through the texture path, and what bounds it
// is that path, not HBM.  Here the four wavefronts of a workgroup take the SAME 64 words and different columns: the
// ciphertext words of a chunk of 4 or 8 vector items are fetched once per workgroup into LDS (double-buffered, one
// barrier per chunk) and read from there by all four wavefronts; only the plaintext words -- the database, read once
// from HBM whatever the number of queries -- still come through the vector memory path.  Per product: 8 / POLYS bytes
// of database and 2 / COLS bytes of ciphertexts, against 8 / POLYS + 8 / COLS above.
constexpr int kTileWords = 64, kTileWavefronts = 4;

template <int POLYS, int COLS, bool NARROW, bool MASKED, typename W>
__global__ void __launch_bounds__(kTileWords * kTileWavefronts)
    inner_product_plain_tile_kernel(const W* __restrict__ cts, const W* __restrict__ pts,
                                    const uint8_t* __restrict__ present, W* __restrict__ out, const DeviceContext ctx,
                                    size_t count, size_t columns, uint64_t cadence, uint32_t column_groups) {
    constexpr int kChunk = POLYS <= 4 ? 8 : 4;                  // vector items per chunk: 24 or 32 rows of 512 bytes
    constexpr int kRows = kChunk * POLYS;                       // ciphertext rows of a chunk: (item, polynomial)
    constexpr int kRowsPerWave = kRows / kTileWavefronts;       // each wavefront fetches this many of them
    static_assert(kRows % kTileWavefronts == 0, "the wavefronts share the chunk's rows evenly");
    __shared__ uint64_t tile[2][kRows][kTileWords];
    const uint32_t logn = ctx.log_degree;
    const size_t words_per_poly = static_cast<size_t>(ctx.moduli_count) << logn;
    uint32_t word_block, column_group;
    locate_replica(blockIdx.x, static_cast<uint32_t>(words_per_poly / kTileWords), column_groups, word_block, column_group);
    const uint32_t lane = threadIdx.x & (kTileWords - 1);
    const uint32_t wave = __builtin_amdgcn_readfirstlane(threadIdx.x / kTileWords);
    const size_t block_word = word_block * static_cast<size_t>(kTileWords);
    const size_t word = block_word + lane;
    const size_t col0 = (static_cast<size_t>(column_group) * kTileWavefronts + wave) * COLS;
    const DeviceModulus m = ctx.moduli[block_word >> logn];
    using Sums = PlainSums<W, POLYS, NARROW>;
    typename Sums::Sum acc[COLS][POLYS];
#pragma unroll
    for (int c = 0; c < COLS; ++c)
#pragma unroll
        for (int q = 0; q < POLYS; ++q) acc[c][q] = Sums::zero();
    uint64_t since_reduce[COLS];
    bool live[COLS];
    size_t pt_column[COLS], mask_column[COLS];
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
        since_reduce[c] = 0;
        live[c] = col0 + c < columns;
        const size_t column = live[c] ? col0 + c : columns - 1;  // a column past the end re-reads the last one
        pt_column[c] = column * count * words_per_poly;
        mask_column[c] = column * count;
    }
    const W* ct_lane = cts + word;
    const W* pt_lane = pts + word;
    // Every load below is issued unconditionally (indices past the end are clamped to the last item and their words
    // never used): the number of loads in flight at each point is then fixed, and a wait names exactly the load it
    // needs instead of draining the queue.
    const size_t last = count - 1;
    // this wavefront's share of a chunk's ciphertext rows: row = (item in chunk) * POLYS + polynomial
    uint64_t staged[kRowsPerWave];
    auto fetch_chunk = [&](size_t first) {
#pragma unroll
        for (int i = 0; i < kRowsPerWave; ++i) {
            const size_t row = static_cast<size_t>(wave) * kRowsPerWave + i;
            const size_t item = first + row / POLYS < count ? first + row / POLYS : last;
            staged[i] = ct_lane[(item * POLYS + row % POLYS) * words_per_poly];
        }
    };
    auto store_chunk = [&](int buffer) {
#pragma unroll
        for (int i = 0; i < kRowsPerWave; ++i) tile[buffer][wave * kRowsPerWave + i][lane] = staged[i];
    };
    if (count == 0) {
#pragma unroll
        for (int c = 0; c < COLS; ++c) {
            if (!live[c]) continue;
#pragma unroll
            for (int q = 0; q < POLYS; ++q) out[((col0 + c) * POLYS + q) * words_per_poly + word] = 0;
        }
        return;
    }
    fetch_chunk(0);
    store_chunk(0);
    __syncthreads();
    // the plaintext words (and mask bytes) of the next kDepth vector items are in flight
    constexpr int kDepth = 4;
    static_assert(kChunk % kDepth == 0, "ring slots are compile-time indices inside a chunk");
    uint64_t y_ring[kDepth][COLS];
    [[maybe_unused]] uint8_t mask_ring[kDepth][COLS];
    auto fetch_plain = [&](int slot, size_t j) {
        const size_t item = j < count ? j : last;
#pragma unroll
        for (int c = 0; c < COLS; ++c) {  // the database is read once: streamed past the caches (non-temporal)
            y_ring[slot][c] = __builtin_nontemporal_load(pt_lane + pt_column[c] + item * words_per_poly);
            if constexpr (MASKED) mask_ring[slot][c] = present[mask_column[c] + item];
        }
    };
#pragma unroll
    for (int d = 0; d < kDepth; ++d) fetch_plain(d, d);
    int buffer = 0;
    for (size_t first = 0; first < count; first += kChunk, buffer ^= 1) {
        fetch_chunk(first + kChunk);  // in flight while this chunk is multiplied
#pragma unroll
        for (int k = 0; k < kChunk; ++k) {
            const size_t j = first + k;
            uint64_t y[COLS];
            [[maybe_unused]] uint8_t mask[COLS];
#pragma unroll
            for (int c = 0; c < COLS; ++c) {
                y[c] = y_ring[k % kDepth][c];
                if constexpr (MASKED) mask[c] = mask_ring[k % kDepth][c];
            }
            fetch_plain(k % kDepth, j + kDepth);
            uint64_t x[POLYS];
#pragma unroll
            for (int q = 0; q < POLYS; ++q) x[q] = tile[buffer][k * POLYS + q][lane];
#pragma unroll
            for (int c = 0; c < COLS; ++c) {
                bool active = live[c] && j < count;
                if constexpr (MASKED) active = active && mask[c] != 0;  // nil plaintext, Bfv.swift:486-489
                if (!active) continue;
                Sums::add_all(acc[c], x, y[c]);
                if (++since_reduce[c] >= cadence) {
                    since_reduce[c] = 0;
#pragma unroll
                    for (int q = 0; q < POLYS; ++q) {
                        acc[c][q] = Sums::from_residue(Sums::reduce(acc[c][q], m));
                    }
                }
            }
        }
        store_chunk(buffer ^ 1);  // that buffer was last read before the barrier that ended the previous chunk
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < COLS; ++c) {
        if (!live[c]) continue;
#pragma unroll
        for (int q = 0; q < POLYS; ++q)
            out[((col0 + c) * POLYS + q) * words_per_poly + word] = static_cast<W>(Sums::reduce(acc[c][q], m));
    }
}

}  // namespace

hipError_t launch_elementwise(ElementwiseOp op, uint64_t* lhs, const uint64_t* rhs, const DeviceContext& ctx,
                              size_t rows, hipStream_t stream) {
    switch (op) {
        case ElementwiseOp::Add: return launch_elementwise_op<ElementwiseOp::Add>(lhs, rhs, ctx, rows, stream);
        case ElementwiseOp::Sub: return launch_elementwise_op<ElementwiseOp::Sub>(lhs, rhs, ctx, rows, stream);
        case ElementwiseOp::Neg: return launch_elementwise_op<ElementwiseOp::Neg>(lhs, rhs, ctx, rows, stream);
        case ElementwiseOp::Mul: return launch_elementwise_op<ElementwiseOp::Mul>(lhs, rhs, ctx, rows, stream);
        case ElementwiseOp::MulScalar:
            return launch_elementwise_op<ElementwiseOp::MulScalar>(lhs, rhs, ctx, rows, stream);
    }
    return hipErrorInvalidValue;
}

hipError_t launch_mul_plain(uint64_t* ct, const uint64_t* pt, const DeviceContext& ctx, uint32_t poly_count,
                            size_t batch, hipStream_t stream) {
    if (batch == 0 || poly_count == 0) return hipSuccess;
    if (ctx.degree < 2) return hipErrorInvalidValue;
    const size_t pairs_per_poly = static_cast<size_t>(ctx.moduli_count) * (ctx.degree / 2);
    hipLaunchKernelGGL(mul_plain_kernel, dim3(grid_for(pairs_per_poly * batch)), dim3(kThreads), 0, stream, ct, pt,
                       ctx, poly_count, pairs_per_poly, batch);
    return hipGetLastError();
}

constexpr bool kDivideAndRoundRowsAtCompileTime = true;
hipError_t launch_divide_and_round_q_last(const uint64_t* in, uint64_t* out, const DeviceContext& ctx,
                                          uint32_t moduli_count, size_t polys, hipStream_t stream) {
    if (polys == 0) return hipSuccess;
    if (ctx.degree < 2 || moduli_count < 2) return hipErrorInvalidValue;
    const size_t total = polys * (ctx.degree / 2);
#define HEAMD_Q_LAST_CASE(L)                                                                                              \
    case L:                                                                                                               \
        hipLaunchKernelGGL(divide_and_round_q_last_rows_kernel<L>, dim3(grid_for(total)), dim3(kThreads), 0, stream, in, out, \
                           ctx, polys);                                                                                   \
        return hipGetLastError()
    if (kDivideAndRoundRowsAtCompileTime) {
        switch (moduli_count) {
            HEAMD_Q_LAST_CASE(2);
            HEAMD_Q_LAST_CASE(3);
            HEAMD_Q_LAST_CASE(4);
            HEAMD_Q_LAST_CASE(5);
            HEAMD_Q_LAST_CASE(6);
            HEAMD_Q_LAST_CASE(7);
            HEAMD_Q_LAST_CASE(8);
            default: break;  // longer chains: the rolled loop over the rows
        }
    }
#undef HEAMD_Q_LAST_CASE
    hipLaunchKernelGGL(divide_and_round_q_last_kernel, dim3(grid_for(total)), dim3(kThreads), 0, stream, in, out, ctx,
                       moduli_count, polys);
    return hipGetLastError();
}

hipError_t launch_mod_switch_down_to_single(const uint64_t* in, uint64_t* out, const DeviceContext& ctx,
                                            uint32_t moduli_count, size_t polys, hipStream_t stream) {
    const size_t total = polys * (ctx.degree >> 1);
    if (total == 0) return hipSuccess;
    if (ctx.degree < 2) return hipErrorNotSupported;
#define HEAMD_TO_SINGLE_CASE(L)                                                                                      \
    case L:                                                                                                          \
        hipLaunchKernelGGL(mod_switch_down_to_single_kernel<L>, dim3(grid_for(total)), dim3(kThreads), 0, stream, in, \
                           out, ctx, polys);                                                                         \
        return hipGetLastError()
    switch (moduli_count) {
        HEAMD_TO_SINGLE_CASE(2);
        HEAMD_TO_SINGLE_CASE(3);
        HEAMD_TO_SINGLE_CASE(4);
        HEAMD_TO_SINGLE_CASE(5);
        HEAMD_TO_SINGLE_CASE(6);
        HEAMD_TO_SINGLE_CASE(7);
        HEAMD_TO_SINGLE_CASE(8);
        default: return hipErrorNotSupported;  // the caller chains launch_divide_and_round_q_last
    }
#undef HEAMD_TO_SINGLE_CASE
}

hipError_t launch_adding_lazy_product(const uint64_t* lhs, const uint64_t* rhs, uint64_t* acc_lo_hi,
                                      const DeviceContext& ctx, hipStream_t stream) {
    const size_t words = static_cast<size_t>(ctx.moduli_count) * ctx.degree;
    hipLaunchKernelGGL(adding_lazy_product_kernel, dim3(grid_for(words)), dim3(kThreads), 0, stream, lhs, rhs,
                       acc_lo_hi, words);
    return hipGetLastError();
}

hipError_t launch_reduce_accumulator(const uint64_t* acc_lo_hi, uint64_t* out, const DeviceContext& ctx,
                                     hipStream_t stream) {
    const size_t words = static_cast<size_t>(ctx.moduli_count) * ctx.degree;
    hipLaunchKernelGGL(reduce_accumulator_kernel, dim3(grid_for(words)), dim3(kThreads), 0, stream, acc_lo_hi, out,
                       ctx, words);
    return hipGetLastError();
}

// A workgroup accumulates POLYS x COLS sums of one word block: each ciphertext word serves COLS columns and each
// plaintext word POLYS polynomials.  One query: 2 x 4 accumulators per lane; several queries' ciphertexts side by side
// (which then share every plaintext word they stream) take two columns per wavefront of the LDS-tiled kernel.
template <int POLYS, int COLS, bool NARROW, typename W>
hipError_t launch_inner_product_plain_polys(const W* cts, const W* pts, const uint8_t* present_device, W* out,
                                            const DeviceContext& ctx, size_t count, size_t columns, uint64_t max_lazy,
                                            uint64_t cadence, hipStream_t stream) {
    const size_t words_per_poly = static_cast<size_t>(ctx.moduli_count) * ctx.degree;
    const dim3 grid(static_cast<unsigned>((columns + COLS - 1) / COLS),
                    static_cast<unsigned>((words_per_poly + kThreads - 1) / kThreads));
    if constexpr (POLYS >= 4) {
        // several queries side by side: ciphertext words through LDS, four column sets per workgroup
        const size_t column_groups = (columns + COLS * kTileWavefronts - 1) / (COLS * kTileWavefronts);
        const size_t blocks = column_groups * (words_per_poly / kTileWords);
        if (ctx.degree >= kThreads && cadence != 0 && blocks < (size_t(1) << 31)) {
            const dim3 tile_grid(static_cast<unsigned>(blocks)), tile_block(kTileWords * kTileWavefronts);
            if (present_device != nullptr)
                hipLaunchKernelGGL((inner_product_plain_tile_kernel<POLYS, COLS, NARROW, true, W>), tile_grid, tile_block, 0,
                                   stream, cts, pts, present_device, out, ctx, count, columns, cadence,
                                   static_cast<uint32_t>(column_groups));
            else
                hipLaunchKernelGGL((inner_product_plain_tile_kernel<POLYS, COLS, NARROW, false, W>), tile_grid, tile_block, 0,
                                   stream, cts, pts, present_device, out, ctx, count, columns, cadence,
                                   static_cast<uint32_t>(column_groups));
            return hipGetLastError();
        }
    }
    if (ctx.degree >= kThreads && cadence != 0 && static_cast<size_t>(grid.x) * grid.y < (size_t(1) << 31)) {
        // one-dimensional grid: the kernel places the column groups of a word block on one XCD itself
        if (present_device != nullptr)
            hipLaunchKernelGGL((inner_product_plain_rows_kernel<POLYS, COLS, NARROW, true, false, W>),
                               dim3(grid.x * grid.y), dim3(kThreads), 0, stream, cts, pts, present_device, out, ctx, count,
                               columns, cadence, grid.x, PackedLayout{});
        else
            hipLaunchKernelGGL((inner_product_plain_rows_kernel<POLYS, COLS, NARROW, false, false, W>),
                               dim3(grid.x * grid.y), dim3(kThreads), 0, stream, cts, pts, present_device, out, ctx, count,
                               columns, cadence, grid.x, PackedLayout{});
        return hipGetLastError();
    }
    hipLaunchKernelGGL((inner_product_plain_kernel<POLYS, COLS, W>), grid, dim3(kThreads), 0, stream, cts, pts,
                       present_device, out, ctx, count, columns, max_lazy);
    return hipGetLastError();
}

template <typename W>
hipError_t launch_inner_product_plain(const W* cts, const W* pts, const uint8_t* present_device, W* out,
                                      const DeviceContext& ctx, uint32_t poly_count, size_t count, size_t columns,
                                      uint64_t max_lazy, uint64_t cadence, bool narrow_moduli, hipStream_t stream) {
    if (columns == 0) return hipSuccess;
    const bool narrow = narrow_moduli && cadence != 0;
    if (narrow && cadence > kNarrowProductSumCadence) cadence = kNarrowProductSumCadence;
    if (sizeof(W) == 4 && cadence > kWord32Cadence) cadence = kWord32Cadence;  // 64-bit sums of products below 2^60
#define HEAMD_INNER_PRODUCT_CASE(POLYS, COLS)                                                                              \
    case POLYS:                                                                                                            \
        return narrow ? launch_inner_product_plain_polys<POLYS, COLS, true>(cts, pts, present_device, out, ctx, count,     \
                                                                            columns, max_lazy, cadence, stream)            \
                      : launch_inner_product_plain_polys<POLYS, COLS, false>(cts, pts, present_device, out, ctx, count,    \
                                                                             columns, max_lazy, cadence, stream)
    switch (poly_count) {
        HEAMD_INNER_PRODUCT_CASE(1, 4);
        HEAMD_INNER_PRODUCT_CASE(2, 4);
        HEAMD_INNER_PRODUCT_CASE(3, 4);
        HEAMD_INNER_PRODUCT_CASE(4, 4);  // 16 accumulators per lane again (two columns: 36.0 M MAC/s, four: 37.9)
        HEAMD_INNER_PRODUCT_CASE(6, 2);  // two columns per wavefront: half the LDS reads per product (one column:
        HEAMD_INNER_PRODUCT_CASE(8, 2);  // 41.8 / 46.6 M MAC/s at 3 / 4 queries, two: 47.6 / 48.8, bench_tools/ab_queries.py)
        default: return hipErrorInvalidValue;
    }
#undef HEAMD_INNER_PRODUCT_CASE
}
template hipError_t launch_inner_product_plain<uint64_t>(const uint64_t*, const uint64_t*, const uint8_t*, uint64_t*,
                                                         const DeviceContext&, uint32_t, size_t, size_t, uint64_t,
                                                         uint64_t, bool, hipStream_t);
template hipError_t launch_inner_product_plain<uint32_t>(const uint32_t*, const uint32_t*, const uint8_t*, uint32_t*,
                                                         const DeviceContext&, uint32_t, size_t, size_t, uint64_t,
                                                         uint64_t, bool, hipStream_t);

namespace {
template <int POLYS, bool NARROW>
hipError_t launch_packed_polys(const uint64_t* cts, const uint64_t* packed_pts, const PackedLayout& layout,
                               const uint8_t* present_device, uint64_t* out, const DeviceContext& ctx, size_t count,
                               size_t columns, uint64_t cadence, hipStream_t stream) {
    constexpr int kCols = 4;
    const size_t words_per_poly = static_cast<size_t>(ctx.moduli_count) * ctx.degree;
    const size_t column_groups = (columns + kCols - 1) / kCols, blocks = column_groups * (words_per_poly / kThreads);
    if (blocks >= (size_t(1) << 31)) return hipErrorInvalidValue;
    const dim3 grid(static_cast<unsigned>(blocks));
    if (present_device != nullptr)
        hipLaunchKernelGGL((inner_product_plain_rows_kernel<POLYS, kCols, NARROW, true, true, uint64_t>), grid,
                           dim3(kThreads), 0, stream, cts, packed_pts, present_device, out, ctx, count, columns, cadence,
                           static_cast<uint32_t>(column_groups), layout);
    else
        hipLaunchKernelGGL((inner_product_plain_rows_kernel<POLYS, kCols, NARROW, false, true, uint64_t>), grid,
                           dim3(kThreads), 0, stream, cts, packed_pts, present_device, out, ctx, count, columns, cadence,
                           static_cast<uint32_t>(column_groups), layout);
    return hipGetLastError();
}
}  // namespace

hipError_t launch_inner_product_plain_packed(const uint64_t* cts, const uint64_t* packed_pts, const PackedLayout& layout,
                                             const uint8_t* present_device, uint64_t* out, const DeviceContext& ctx,
                                             uint32_t poly_count, size_t count, size_t columns, uint64_t cadence,
                                             bool narrow_moduli, hipStream_t stream) {
    if (columns == 0) return hipSuccess;
    if (ctx.degree < kThreads || cadence == 0 || layout.rows != ctx.moduli_count) return hipErrorInvalidValue;
    if (narrow_moduli && cadence > kNarrowProductSumCadence) cadence = kNarrowProductSumCadence;
#define HEAMD_PACKED_CASE(POLYS)                                                                                          \
    case POLYS:                                                                                                           \
        return narrow_moduli ? launch_packed_polys<POLYS, true>(cts, packed_pts, layout, present_device, out, ctx, count,  \
                                                                columns, cadence, stream)                                 \
                             : launch_packed_polys<POLYS, false>(cts, packed_pts, layout, present_device, out, ctx, count, \
                                                                 columns, cadence, stream)
    switch (poly_count) {
        HEAMD_PACKED_CASE(1);
        HEAMD_PACKED_CASE(2);
        HEAMD_PACKED_CASE(3);
        default: return hipErrorInvalidValue;
    }
#undef HEAMD_PACKED_CASE
}

// ct [batch][polys][L][N] *= pt [batch][L][N] on 4-byte words (one word per lane)
namespace {
__global__ void __launch_bounds__(kThreads)
    mul_plain_kernel32(uint32_t* __restrict__ ct, const uint32_t* __restrict__ pt, const DeviceContext ctx,
                       uint32_t poly_count, size_t words_per_poly, size_t batch) {
    const size_t total = words_per_poly * batch;
    for (size_t i = blockIdx.x * static_cast<size_t>(kThreads) + threadIdx.x; i < total;
         i += static_cast<size_t>(gridDim.x) * kThreads) {
        const size_t b = i / words_per_poly, k = i - b * words_per_poly;
        const DeviceModulus m = ctx.moduli[k >> ctx.log_degree];
        const uint64_t y = stream_load(pt + i);
        for (uint32_t c = 0; c < poly_count; ++c) {
            uint32_t* slot = ct + (b * poly_count + c) * words_per_poly + k;
            stream_store(slot, barrett_mul(stream_load(slot), y, m.p, m.product_factor, static_cast<int>(m.product_shift)));
        }
    }
}
}  // namespace

hipError_t launch_mul_plain32(uint32_t* ct, const uint32_t* pt, const DeviceContext& ctx, uint32_t poly_count, size_t batch,
                              hipStream_t stream) {
    const size_t words_per_poly = static_cast<size_t>(ctx.moduli_count) * ctx.degree;
    if (words_per_poly * batch == 0) return hipSuccess;
    hipLaunchKernelGGL(mul_plain_kernel32, dim3(grid_for(words_per_poly * batch)), dim3(kThreads), 0, stream, ct, pt, ctx,
                       poly_count, words_per_poly, batch);
    return hipGetLastError();
}

}  // namespace heamd
