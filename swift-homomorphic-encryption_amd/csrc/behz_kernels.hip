// behz_kernels.hip -- BEHZ multiplication row by row: multiplyWithoutScaling's transforms and tensor product and
// dropExtendedBase's inverse transforms (reference Sources/HomomorphicEncryption/Bfv/Bfv+Multiply.swift:18-85) in ONE kernel
// per [Q, Bsk] row band.
//
// Per (item, row r of the [Q, Bsk] records) the reference runs four forward transforms (row r of a0, a1, b0, b1:
// Bfv+Multiply.swift:51-57, 76-79), three dyadic products a0 b0 | a0 b1 + a1 b0 | a1 b1 (:80-82) and three inverse
// transforms scaled by t (:31-48) -- and none of them ever needs another row.  The unfused pipeline (ntt_kernels.hip:
// kSourceRows forward + kInverseFromTensor inverse) writes the four transformed rows to HBM and reads them back three
// times over; here one workgroup keeps them in registers:
//
//   load 4 rows (Q rows straight from the ciphertexts, Bsk rows from the lift's slab)            4 x 64 KiB in
//   forward_row<ROWS = 4>    one pass structure, every twiddle fetched ONCE for the four rows
//   tensor product           element-wise in the low-pass layout both transforms share -- no exchange in between
//   inverse_row<ROWS = 3>    every twiddle fetched once for the three rows, t N^-1 in the last stage
//   store 3 rows                                                                                  3 x 64 KiB out
//
// 7 row moves per 7 transforms instead of 14 + the tensor load's re-reads; two launches (Q band, Bsk band) instead of
// four.  One 1024-lane workgroup per CU at 128 registers per lane (4 rows x 8 words = 64 registers of row data); the rows
// go through two transposition tiles side by side, two rows per round (ntt_rows.hpp kWideGroupTiles).
#include <hip/hip_runtime.h>

#include <cstdlib>

#include <type_traits>

#include "device_context.hpp"
#include "device_math.hpp"
#include "kernels.hpp"
#include "ntt_common.hpp"
#include "ntt_rows.hpp"

namespace heamd {

namespace {

using namespace ntt;

// Where the four operand rows of (item, r) come from and where the three product rows go.
struct BehzRows {
    const uint64_t* lhs;     // [items][2][L][N] Coeff: (a0, a1), items ct_stride words apart
    const uint64_t* rhs;     // (b0, b1)
    size_t ct_stride;
    const uint64_t* lifted;  // [items][4][record_rows][N]: the Bsk rows r >= L of the lifted a0, a1, b0, b1 (Coeff)
    uint32_t L;              // ciphertext moduli = rows that are read from the ciphertexts themselves (RnsTool.swift:329-330)
};

constexpr int kBehzOperands = 4, kBehzProducts = 3;

// The transformed operand words as the tensor product takes them.  Where the products go through the one-word-quotient Barrett
// (reduce_product_sum_bounded_lazy: any sum below 2^(63 + bits(p))) the words stay in [0, 2p) -- the last conditional subtract
// of every word is not spent: the cross term of such words is below 8 p^2, which is inside the bound for every modulus these
// butterfly classes admit (limb-wise: p < 2^55; fold: p = 2^b - d with b <= 60, or 2^60 + e with e < 2^24, whose 8 p^2 is
// 2^123 (1 + 2^-35)); the [0, 8p) and exact classes hand over canonical words.
constexpr bool kBehzLazyOperands = true;
template <int MODE, int ROWS, int E>
__device__ __forceinline__ void reduce_operands(uint64_t (&v)[ROWS][E], uint64_t p) {
    if constexpr (!kLazyTransformInput<MODE> || !kBehzLazyOperands) {
        canonicalize_all<MODE>(v, p);
    } else {
#pragma unroll
        for (int row = 0; row < ROWS; ++row) {
#pragma unroll
            for (int r = 0; r < E; ++r) {
                if constexpr (is_split(MODE)) {
                    v[row][r] = LazyReducer(p).lazy(v[row][r]);  // below 2^10 p -> [0, 2p)
                } else {
                    static_assert(is_fold(MODE), "the lazy transform inputs are the limb-wise and the fold classes");
                    v[row][r] = csub_uniform(csub_uniform(csub_uniform(v[row][r], 8 * p), 4 * p), 2 * p);  // below 14p -> [0, 2p)
                }
            }
        }
    }
}

// a0 b0 | a0 b1 + a1 b0 | a1 b1, as the inverse transform of MODE takes them: in [0, 5p) for the limb-wise and fold
// butterflies (ntt_rows.hpp kLazyTransformInput; operands in [0, 2p), reduce_operands), canonical otherwise (canonical
// operands).  The cross term is one exact 128-bit sum and one reduction.
template <int MODE, int E>
__device__ __forceinline__ void tensor_rows(uint64_t (&v)[kBehzOperands][E], const DeviceModulus& mod) {
#pragma unroll
    for (int r = 0; r < E; ++r) {
        const uint64_t a0 = v[0][r], a1 = v[1][r], b0 = v[2][r], b1 = v[3][r];
        ProductSum cross = product_sum_first(a0, b1);
        product_sum_add(cross, a1, b0);
        if constexpr (kLazyTransformInput<MODE>) {
            // (these moduli are 41 .. 61 bits: wide_shift != 0, 2 p^2 inside the bounded reduction's range)
            v[0][r] = reduce_product_sum_bounded_lazy(product_sum_first(a0, b0), mod);
            v[1][r] = reduce_product_sum_bounded_lazy(cross, mod);
            v[2][r] = reduce_product_sum_bounded_lazy(product_sum_first(a1, b1), mod);
        } else {
            v[0][r] = barrett_mul(a0, b0, mod.p, mod.product_factor, static_cast<int>(mod.product_shift));
            v[1][r] = mod.wide_shift != 0 ? reduce_product_sum_bounded(cross, mod) : reduce_product_sum(cross, mod);  // wave-uniform
            v[2][r] = barrett_mul(a1, b1, mod.p, mod.product_factor, static_cast<int>(mod.product_shift));
        }
    }
}

// MODE_F / MODE_I: the butterfly classes of the band's forward and inverse transforms (the limb-wise inverse runs its signed
// form, ntt_common.hpp kModeSplitSigned).  `ctx`: the [Q, Bsk] context with t N^-1 as its inverse-degree constants
// (bfv_api.cpp: qbsk_moduli_scaled_by_t); `out`: [items][3][record_rows][N] Coeff.
template <int LOGN, int LOGT, int MODE_F, int MODE_I>
__global__ void __launch_bounds__(1 << LOGT, min_waves_per_simd(LOGN - LOGT, kBehzOperands))
    behz_rows_fused(uint64_t* __restrict__ out, const DeviceContext ctx, const RowMap map, const BehzRows src) {
    constexpr int LOGE = LOGN - LOGT, E = 1 << LOGE, LO0 = LOGN - LOGE;
    using S = Schedule<LOGN, LOGE>;
    static_assert(S::P >= 2 && S::P <= 5, "row groups go through the LDS tile");
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
    const uint32_t tid = threadIdx.x;
    uint32_t item, within;
    locate(map, blockIdx.x, item, within);
    item += map.record_base;
    const uint32_t r = map.band_offset + within, mi = map.mod_base + within;
    const size_t record_words = static_cast<size_t>(map.record_rows) << LOGN;
    uint64_t v[kBehzOperands][E];
    // ---- the four operand rows, each read once (non-temporal: nobody else reads them)
    if (r < src.L) {  // wave-uniform
#pragma unroll
        for (int k = 0; k < kBehzOperands; ++k) {
            const uint64_t* const row = ((k & 2) != 0 ? src.rhs : src.lhs) + size_t(item) * src.ct_stride +
                                        ((size_t(k & 1) * src.L + r) << LOGN);
            global_load<LOGN, LOGE, LO0, LOGE>(v[k], tid, make_uniform_resource(row, 8u << LOGN));
        }
    } else {
#pragma unroll
        for (int k = 0; k < kBehzOperands; ++k) {
            const uint64_t* const row = src.lifted + (size_t(item) * kBehzOperands + k) * record_words + (size_t(r) << LOGN);
            global_load<LOGN, LOGE, LO0, LOGE>(v[k], tid, make_uniform_resource(row, 8u << LOGN));
        }
    }
    const DeviceModulus mod = ctx.moduli[mi];
    {
        const Twiddles<MODE_F> tw(ctx, false, mi, LOGN, 0, kLaneMajorTwiddles<LOGN, LOGT, MODE_F, false, true>);
        forward_row<LOGN, LOGE, MODE_F, kBehzOperands, false, false>(v, tid, tw, mod.p, lds);
    }
    static_assert(kLazyTransformInput<MODE_F> == kLazyTransformInput<MODE_I>, "one butterfly class per band");
    reduce_operands<MODE_F>(v, mod.p);
    // ---- the tensor product where the words lie: forward_row leaves them in the layout of the pass on the low bits, which
    // is the layout inverse_row takes them in
    tensor_rows<MODE_I>(v, mod);
    uint64_t (&products)[kBehzProducts][E] = *reinterpret_cast<uint64_t (*)[kBehzProducts][E]>(&v[0]);
    {
        constexpr int INPUT_STAGES = kLazyTransformInput<MODE_I> ? kLazyInputStages : 0;
        const Twiddles<MODE_I> tw(ctx, true, mi, LOGN, 0, kLaneMajorTwiddles<LOGN, LOGT, MODE_I, true, false>);
        constexpr int HEAD = kGroupTwiddlesAhead<MODE_I, kBehzProducts>;
        TwiddleWords head[HEAD];
        inverse_row_head<LOGN, LOGE, MODE_I, false, HEAD>(head, tw, tid);
        inverse_row<LOGN, LOGE, MODE_I, kBehzProducts, true, INPUT_STAGES, LOGN, false, HEAD>(products, tid, tw, mod, lds, head);
    }
    const uint32_t store_lane = step_lane<MODE_I>(tid);
#pragma unroll
    for (int c = 0; c < kBehzProducts; ++c) {
        uint64_t* const row = out + (size_t(item) * kBehzProducts + c) * record_words + (size_t(r) << LOGN);
        global_store<LOGN, LOGE, LO0, LOGE>(products[c], store_lane, make_uniform_resource(row, 8u << LOGN));
    }
}

constexpr bool kBehzFoldLazy = true;
template <int LOGN, int LOGT, int MODE_F, int MODE_I>
hipError_t launch_band(uint64_t* out, const DeviceContext& ctx, const RowMap& map, size_t workgroups, const BehzRows& src,
                       hipStream_t stream) {
    constexpr size_t lds_bytes = kGroupTiles<kBehzOperands> * lds_words(1u << LOGN) * sizeof(uint64_t);
    auto kernel = behz_rows_fused<LOGN, LOGT, MODE_F, MODE_I>;
    if (hipError_t e = allow_dynamic_lds(kernel, lds_bytes); e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(workgroups)), dim3(1u << LOGT), lds_bytes, stream, out, ctx, map, src);
    return hipGetLastError();
}

// `mode`: kModeSplit, kModeApprox (a fold form is resolved from the band's moduli, as launch_forward_kernel does) or
// kModeExact
template <int LOGN, int LOGT>
hipError_t launch_band_in_mode(int mode, uint64_t* out, const DeviceContext& ctx, const RowMap& map, size_t workgroups,
                               const BehzRows& src, hipStream_t stream) {
    if (mode == kModeSplit) {
        // moduli 2^b - d (every parameter set of the reference): the products folded by a shift (ntt_common.hpp kModeFoldLazy)
        if (kBehzFoldLazy && map.mod_base + map.band_rows <= ctx.shift_prefix)
            return launch_band<LOGN, LOGT, kModeFoldLazy, kModeFoldLazy>(out, ctx, map, workgroups, src, stream);
        return launch_band<LOGN, LOGT, kModeSplit, kModeSplitSigned>(out, ctx, map, workgroups, src, stream);
    }
    if (mode == kModeApprox) {
        const int fold = ctx.forward_split_pairs != nullptr ? fold_mode(ctx, map.mod_base, map.band_rows) : 0;
        if (fold == kModeFoldMinus) return launch_band<LOGN, LOGT, kModeFoldMinus, kModeFoldMinus>(out, ctx, map, workgroups, src, stream);
        if (fold == kModeFoldPlus) return launch_band<LOGN, LOGT, kModeFoldPlus, kModeFoldPlus>(out, ctx, map, workgroups, src, stream);
        return launch_band<LOGN, LOGT, kModeApprox, kModeApprox>(out, ctx, map, workgroups, src, stream);
    }
    return launch_band<LOGN, LOGT, kModeExact, kModeExact>(out, ctx, map, workgroups, src, stream);
}

}  // namespace

// (item, row) workgroups from which the row-fused kernels are taken: a row-fused workgroup runs seven transforms in a row, alone on
// its CU -- below a few generations of them the unfused launches (one transform per workgroup, two per CU) finish sooner: ct x ct
// on 32 / 64 ciphertext pairs 172 -> 158 / 287 -> 268 us, equal at 128, the fused kernels ahead from 256 on
// (profiles/r06x_small_chains.txt).  The callers' parts of a batch (bfv_api.cpp kBehzFloorParts: at least 256 items) stay above it.
constexpr size_t kBehzRowsFusedAbove = 1152;
namespace {
// the degrees whose tiled transform holds a row in 8 words per lane (four rows = 64 registers)
bool rows_fused_shape(const DeviceContext& qbsk, uint32_t record_rows, uint32_t source_moduli, size_t items) {
    const bool tiled = qbsk.log_degree == 12 || qbsk.log_degree == 13;
    return tiled && source_moduli != 0 && source_moduli < record_rows && record_rows <= 64 && qbsk.moduli_count >= record_rows &&
           items != 0 && items * record_rows <= (size_t(1) << 30);
}
}  // namespace
// (the batch threshold is the CALLER's question -- a part of a batch that passed it is launched whatever its own size)
// HEAMD_BEHZ_FUSED_ABOVE=k overrides the threshold (the parity tests put small batches through the row-fused kernels with it)
bool behz_rows_fused_supported(const DeviceContext& qbsk, uint32_t record_rows, uint32_t source_moduli, size_t items) {
    size_t above = kBehzRowsFusedAbove;
    if (const char* forced = std::getenv("HEAMD_BEHZ_FUSED_ABOVE")) above = static_cast<size_t>(std::strtoull(forced, nullptr, 10));
    return rows_fused_shape(qbsk, record_rows, source_moduli, items) && items * record_rows > above;
}

constexpr bool kBehzLazyLiftedRows = true;
bool behz_lifted_rows_may_be_lazy(const DeviceContext& scaled_qbsk, uint32_t record_rows, uint32_t source_moduli) {
    if (!kBehzLazyLiftedRows || !kFoldButterflies || record_rows > 64 || source_moduli >= record_rows) return false;
    if (scaled_qbsk.log_degree != 12 && scaled_qbsk.log_degree != 13) return false;
    ntt::BandRun runs[ntt::kMaxBandRuns];
    if (ntt::band_runs(scaled_qbsk, record_rows, runs) <= 1) return false;  // one launch in one mode for every row
    const uint64_t lifted = ((record_rows == 64 ? ~uint64_t(0) : (uint64_t(1) << record_rows) - 1) >> source_moduli) << source_moduli;
    return (scaled_qbsk.fold_plus_mask & lifted) == lifted && scaled_qbsk.forward_split_pairs != nullptr;
}

hipError_t launch_behz_rows_fused(const uint64_t* lhs, const uint64_t* rhs, size_t ct_stride, const uint64_t* lifted,
                                  uint64_t* out, const DeviceContext& scaled_qbsk, uint32_t record_rows, uint32_t source_moduli,
                                  size_t items, hipStream_t stream, int part) {
    if (items == 0) return hipSuccess;
    if (!rows_fused_shape(scaled_qbsk, record_rows, source_moduli, items) || scaled_qbsk.scaled_inverse_degree == 0)
        return hipErrorNotSupported;
    const BehzRows src{lhs, rhs, ct_stride, lifted, source_moduli};
    auto launch = [&](int mode, uint32_t base, uint32_t band) {
        const RowMap map = make_row_map(base, band, record_rows, base);
        const size_t workgroups = items * band;
        return scaled_qbsk.log_degree == 12 ? launch_band_in_mode<12, 9>(mode, out, scaled_qbsk, map, workgroups, src, stream)
                                            : launch_band_in_mode<13, 10>(mode, out, scaled_qbsk, map, workgroups, src, stream);
    };
    ntt::BandRun runs[ntt::kMaxBandRuns];
    const int count = ntt::band_runs(scaled_qbsk, record_rows, runs);
    // `part`: kBehzAllRows, or the runs that read nothing of the lift's (every row below source_moduli: kBehzCiphertextRows) /
    // the others (kBehzLiftedRows) -- the first kind does not wait for the lift
    if (count <= 1) return part == kBehzCiphertextRows ? hipSuccess : launch(ntt::production_mode(scaled_qbsk), 0, record_rows);
    for (int k = 0; k < count; ++k) {
        const bool from_ciphertexts = runs[k].base + runs[k].rows <= source_moduli;
        if ((part == kBehzCiphertextRows && !from_ciphertexts) || (part == kBehzLiftedRows && from_ciphertexts)) continue;
        if (hipError_t e = launch(runs[k].mode, runs[k].base, runs[k].rows); e != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace heamd
