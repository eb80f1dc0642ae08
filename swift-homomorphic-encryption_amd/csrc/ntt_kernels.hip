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
//     lane (T = 2^LOGT lanes), and is touched in HBM exactly once in and once out (buffer loads / stores: one 32-bit
//     lane offset, scalar row offsets -- no 64-bit address arithmetic on the vector ALU);
//   * log2(N) radix-2 stages are grouped into passes of LOGE stages executed on registers; between passes the row
//     is transposed through a padded LDS tile (4 transposes for N = 8192 with 8 words per lane: 13 = 3+3+3+3+1, the
//     one-stage pass on bit 0 or -- the plain-slab inverse -- on bit 12: kTopPartialOrder below);
//   * twiddles of the first two passes are wave-uniform (scalar loads); later passes gather them from the L2-resident
//     per-modulus tables, shared by every workgroup of that modulus;
//   * butterflies: limb-wise Shoup products for the usual <= 55-bit moduli (ntt_common.hpp kModeSplit), products folded
//     by a shift for larger moduli next to a power of two (kModeFoldMinus / kModeFoldPlus: the 60-bit parameter sets, the
//     BEHZ auxiliary primes), Harvey butterflies with a 3- or 4-multiply quotient for the other moduli up to 2^61 / 2^62;
//   * N = 16384 / 32768: the row as 2 / 4 interleaved sub-rows of 8192 words through the same machinery
//     (ntt_forward_interleaved / ntt_inverse_interleaved below).
#include <hip/hip_runtime.h>

#include <type_traits>

#include "device_context.hpp"
#include "device_math.hpp"
#include "kernels.hpp"
#include "ntt_common.hpp"
#include "ntt_rows.hpp"
#include "placement.hpp"

namespace heamd {

namespace {

using namespace ntt;

// Row sources of the forward transform other than the slab itself: the step that would otherwise write the slab (and
// this kernel read it back) is applied to the words as they are loaded.
//   kSourceSpread  the key-switching decomposition (Bfv+Keys.swift:165-179): output row (poly, j, r) of a
//                  [polys][L][L+1][N] slab is the transform mod ks_modulus[r] of row j of polynomial `poly`, read
//                  straight from the ciphertext and reduced mod r first when q_j > modulus r; with galois_inverse
//                  set, of row j of the polynomial's image under the automorphism (Bfv.applyGalois key-switches
//                  galois(c1), Bfv.swift:190-196): the row is read in order and permuted through the LDS tile;
//   kSourceLift    Plaintext.convertToEvalFormat (Plaintext.swift:149-170): output row (poly, r) of a [polys][L][N]
//                  slab is the transform mod q_r of the centred lift of plaintext `poly` ([N] values < t):
//                  x < (t + 1) / 2 ? x : x + (q_r - t).
//   kSourceRows    the Q band of BEHZ's [Q, Bsk] records (Bfv+Multiply.swift:51-57): liftQToQBsk leaves rows [0, L) of
//                  a lifted polynomial equal to the input (RnsTool.swift:329-330), so row (record, r) of the band is
//                  the transform of row r of the record's source polynomial, read where the ciphertext lies instead
//                  of from a copy the lift would write: record = item * 4 + slot takes polynomial slot & 1 of item
//                  `item` of `base` (slots 0, 1) or `second` (slots 2, 3), items `stride` words apart; without a
//                  second operand, record r takes the polynomial at base + r * stride.
constexpr int kSourceSlab = 0, kSourceSpread = 1, kSourceLift = 2, kSourceRows = 3;
struct SpreadSource {
    const uint64_t* base;  // row j of polynomial `poly` at base + poly * stride + j * N
    size_t stride;
    uint32_t L;            // source rows per polynomial (1 for a plaintext)
    uint64_t plaintext_modulus;
    // kSourceSpread only: g^-1 mod 2N when the source polynomial is to be taken through f(x) -> f(x^g) first
    // (PolyRq/Galois.swift:115-143), 0 otherwise
    uint32_t galois_inverse;
    const uint64_t* second;  // kSourceRows only: the operand of slots 2, 3 (nullptr: one operand, consecutive polynomials)
};

template <int LOGN, int LOGT, int MODE, int SPREAD = kSourceSlab, int ROWS = 1>
__global__ void __launch_bounds__(1 << LOGT, min_waves_per_simd(LOGN - LOGT, ROWS))
    ntt_forward_tiled(uint64_t* __restrict__ slab, const DeviceContext ctx, const RowMap map, const SpreadSource spread) {
    constexpr int LOGE = LOGN - LOGT;
    constexpr int E = 1 << LOGE;
    using S = Schedule<LOGN, LOGE>;
    static_assert(S::P >= 1 && S::P <= 5, "unsupported pass count");
    static_assert(ROWS == 1 || S::P >= 2, "row groups go through the LDS tile");
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
    const uint32_t tid = threadIdx.x;
    uint32_t record, within;
    size_t rows[ROWS];
    if constexpr (SPREAD != kSourceSlab && SPREAD != kSourceRows) {
        // the band_rows output rows of a record are transforms of one source row: one replica set per group of ROWS
        // consecutive records (a workgroup transforms the same band row of each of them)
        // (a launch may cover a band of the records' rows -- the moduli of one butterfly class, launch_ntt_spread)
        uint32_t group;
        locate_replica(blockIdx.x, gridDim.x / map.band_rows, map.band_rows, group, within);
        record = map.record_base + group * ROWS;
        const size_t record_rows = map.record_rows == 0 ? map.band_rows : map.record_rows;
#pragma unroll
        for (int k = 0; k < ROWS; ++k) rows[k] = size_t(record + k) * record_rows + map.band_offset + within;
    } else {
        locate_rows<ROWS>(map, blockIdx.x, rows, record, within);
    }
    const uint32_t mi = map.mod_base + within;
    const DeviceModulus mod = ctx.moduli[mi];
    const Twiddles<MODE> tw(ctx, false, mi, LOGN, 0, kLaneMajorTwiddles<LOGN, LOGT, MODE, false, true>);
    const uint64_t p = mod.p;
    uint64_t v[ROWS][E];

    if constexpr (S::P == 1) {
        const BufferResource x = make_resource(slab + (rows[0] << LOGN), 8u << LOGN);
        global_load<LOGN, LOGE, 0, LOGN>(v[0], tid, x);
        forward_pass<LOGN, LOGE, 0, LOGN, MODE, true, 1>(v, tid, tw, p, true,
                                                        forward_first_twiddle<LOGN, LOGE, 0, LOGN, MODE, true>(tw, tid));
        canonicalize_all<MODE>(v, p);
        global_store<LOGN, LOGE, 0, LOGN>(v[0], tid, x);
    } else {
        constexpr int LO0 = LOGN - LOGE;
        if constexpr (SPREAD == kSourceRows) {
#pragma unroll
            for (int k = 0; k < ROWS; ++k) {
                const size_t rec = record + k;
                const uint64_t* source;
                if (spread.second == nullptr) {
                    source = spread.base + rec * spread.stride + (static_cast<size_t>(within) << LOGN);
                } else {
                    const size_t item = rec >> 2, slot = rec & 3;
                    source = ((slot & 2) != 0 ? spread.second : spread.base) + item * spread.stride +
                             (((slot & 1) * spread.L + within) << LOGN);
                }
                global_load<LOGN, LOGE, LO0, LOGE>(v[k], tid, make_resource(source, 8u << LOGN));
            }
        } else if constexpr (SPREAD != kSourceSlab) {
            bool reduce[ROWS];  // uniform: the source row is canonical mod a larger modulus than this row's
#pragma unroll
            for (int k = 0; k < ROWS; ++k) {
                const size_t poly = (record + k) / spread.L, j = (record + k) - poly * spread.L;  // record = poly * L + j
                global_load<LOGN, LOGE, LO0, LOGE, 0>(  // cached: the other rows of this record read the same words
                    v[k], tid, make_resource(spread.base + poly * spread.stride + (j << LOGN), 8u << LOGN));
                // the split butterflies take any 64-bit multiplicand and have room for an addend below 2p, so they
                // transform a residue below 2p as it is
                reduce[k] = SPREAD == kSourceSpread && ctx.moduli[j].p > p && !(is_split(MODE) && ctx.moduli[j].p < 2 * p);
                if constexpr (SPREAD == kSourceSpread) {
                    if (spread.galois_inverse != 0) {
                        // output coefficient e takes source coefficient i = e g^-1 mod 2N, negated mod q_j when i >= N.
                        // The row sits in the tile in order; consecutive lanes read words an odd stride apart, which
                        // spreads over all banks.
                        const uint64_t source_modulus = ctx.moduli[j].p;
                        if (k > 0) __syncthreads();
#pragma unroll
                        for (int r = 0; r < E; ++r) lds[element_index<LOGN, LOGE, LO0, LOGE>(r, tid)] = v[k][r];
                        __syncthreads();
#pragma unroll
                        for (int r = 0; r < E; ++r) {
                            const uint32_t doubled = (element_index<LOGN, LOGE, LO0, LOGE>(r, tid) * spread.galois_inverse) &
                                                     ((2u << LOGN) - 1u);
                            const uint64_t x = lds[doubled & ((1u << LOGN) - 1u)];
                            v[k][r] = (doubled >> LOGN) != 0 && x != 0 ? source_modulus - x : x;
                        }
                        if (k + 1 == ROWS) __syncthreads();  // the transform's exchanges reuse the tile
                    }
                }
            }
            if constexpr (SPREAD == kSourceLift) {
                const uint64_t threshold = (spread.plaintext_modulus + 1) >> 1, increment = p - spread.plaintext_modulus;
#pragma unroll
                for (int k = 0; k < ROWS; ++k)
#pragma unroll
                    for (int r = 0; r < E; ++r) v[k][r] = v[k][r] < threshold ? v[k][r] : v[k][r] + increment;
            } else {
                bool any = false;
#pragma unroll
                for (int k = 0; k < ROWS; ++k) any |= reduce[k];
                if (any) {  // rare (moduli of very different sizes): kept out of the straight-line path
#pragma unroll
                    for (int k = 0; k < ROWS; ++k)
#pragma unroll
                        for (int r = 0; r < E; ++r)
                            v[k][r] = reduce[k] ? barrett_reduce64_uniform(v[k][r], p, mod.barrett64) : v[k][r];
                }
            }
        } else {
#pragma unroll
            for (int k = 0; k < ROWS; ++k)
                global_load<LOGN, LOGE, LO0, LOGE>(v[k], tid, make_resource(slab + (rows[k] << LOGN), 8u << LOGN));
        }
        using O = PassOrder<LOGN, LOGE, kTopPartialOrder<LOGN, LOGE, false, SPREAD == kSourceSlab>>;
        // (N = 4096, two rows per workgroup on the shift-folded products: one register short with a carried lane index)
        constexpr bool LATE = LOGN == 12 && ROWS == 2 && MODE == kModeFoldLazy;
        forward_row<LOGN, LOGE, MODE, ROWS, true, O::kTop, LATE>(v, tid, tw, p, lds);
        constexpr int LO_BEFORE_LOW = O::kTop ? O::lo(S::P - 2) : LOGN - (S::P - 1) * LOGE;
#pragma unroll
        for (int k = 0; k < ROWS; ++k) {
            const BufferResource out = make_resource(slab + (rows[k] << LOGN), 8u << LOGN);
            if constexpr (kStagedStore<LOGN, LOGE, LO_BEFORE_LOW, O::LOW>) {
                global_store_staged<LOGN, LOGE, O::LOW>(v[k], tid, out, lds);
            } else {
                global_store<LOGN, LOGE, 0, O::LOW>(v[k], tid, out);
            }
        }
    }
}

// TENSOR: the BEHZ tensor product (Bfv+Multiply.swift:80-82) fused into the load.  The launch covers records
// (item, c), c in {0, 1, 2}, of `record_rows` rows; the words of row r of record (item, c) are computed from the four
// Eval polynomials (a0, a1, b0, b1) of the item at tensor_source + (item * 4 + k) * record_rows * N + r * N as
// a0 b0 | a0 b1 + a1 b0 | a1 b1 while they are loaded, instead of by a kernel that writes them for this one to read.
// KEYMAC: the lazy inner product with the key-switching key (Bfv+Keys.swift:180-202) fused into the load.  Records
// are (polynomial, c), c in {0, 1}, of record_rows = L + 1 rows; word k of row r is
// sum_j spread[poly][j][r][k] * key[j][c][key_row(r)][k] mod ks_modulus[r], accumulated in the carry-counting form.
// kInverseFromSlabScaled: a plain slab whose context carries t N^-1 (dropExtendedBase without the fused tensor load)
// kInverseFromKeyMacFinish: the same load, restricted to the band rows r < L, with the last step of the key switch --
// drop the special modulus (divideAndRoundQLast by the centred representative of the q_ks word, Bfv+Keys.swift:203-207)
// and add the update to the ciphertext (Bfv.swift:216-217) -- applied to the transform's canonical words before they are
// stored: the q_ks row of every (polynomial, c) comes from an earlier launch of the plain kInverseFromKeyMac kernel over
// that one band row, the rows r < L of the product are never written, and the separate finish kernel (26 row moves per
// ciphertext) is gone.
constexpr int kInverseFromSlab = 0, kInverseFromTensor = 1, kInverseFromKeyMac = 2, kInverseFromSlabScaled = 3,
              kInverseFromKeyMacFinish = 4;
constexpr bool is_key_mac(int source) { return source == kInverseFromKeyMac || source == kInverseFromKeyMacFinish; }
constexpr bool kKeyMacBoundedReduce = true;
// Where the limb-wise inverse transform multiplies its differences as signed words (ntt_common.hpp kModeSplitSigned): the
// plain-slab transform at N = 8192 (partial pass on the top bits) is 2.5 % faster in the unsigned form, N = 4096 1.8 % and the
// interleaved rows of N = 16384 3.3 % faster in the signed one, the key-MAC transforms 1.9 % (profiles/r04t_inverse_forms_ab.txt).
template <int LOGN, int SOURCE>
constexpr bool kSignedInverse = !(LOGN == 13 && (SOURCE == kInverseFromSlab || SOURCE == kInverseFromSlabScaled));
// Which two rows a key-MAC workgroup takes where the register file holds two: the same key column of two consecutive
// polynomials, the other column in a sibling workgroup of the same XCD.  The counters read 1.7 x the spread slab for this
// kernel (profiles/r03z_pmc_traffic_per_kernel.txt), which suggested pairing the two COLUMNS of one polynomial instead
// (every spread word fetched once).  Measured both ways and dropped: with per-word sums (8-byte loads at a 16-byte lane
// stride, twice the load instructions) relinearize 727 -> 671 k/s (profiles/r04j_keymac_loads_and_column_words_ab.txt,
// the variant lived under bench_tools/variants/ until round 4, commit 006a646); with the two columns summed one after the other through the same 16-byte
// loads 723 -> 708 k/s
// (profiles/r04c_keymac_columns_in_turn_ab.txt).  The slab's re-reads come out of L2 / the memory-side cache; they are not
// what binds the kernel.
// Byte offset of argument INDEX of a kernel in its kernel-argument segment (arguments in order, each at its natural
// alignment), from the kernel's own type; the number of arguments.
template <size_t INDEX, typename... Args>
constexpr size_t kernarg_offset(void (*)(Args...)) {
    constexpr size_t sizes[] = {sizeof(Args)...}, aligns[] = {alignof(Args)...};
    size_t at = 0;
    for (size_t k = 0; k <= INDEX; ++k) {
        at = (at + aligns[k] - 1) / aligns[k] * aligns[k];
        if (k < INDEX) at += sizes[k];
    }
    return at;
}
template <typename... Args>
constexpr size_t kernarg_count(void (*)(Args...)) { return sizeof...(Args); }

struct InverseSource {
    const uint64_t* first;   // tensor: the lifted polynomials; key MAC: the spread slab
    const uint64_t* second;  // key MAC: the key
    uint32_t L, top_rows;    // key MAC: source moduli, rows per key polynomial
    // kInverseFromKeyMacFinish: the ciphertexts the update is added to ([item] ct_stride words apart, polynomial c at
    // c L N; the first `added_polys` polynomials are added, the others replaced), and where the result goes
    // ([item][2][L][N]); the slab argument of the kernel is the product slab whose q_ks rows are read
    const uint64_t* ct_base;
    size_t ct_stride;
    uint64_t* out;
    uint32_t added_polys;
    // ... the Galois key switch's end (Bfv.swift:190-196): galois_inverse = g^-1 mod 2N != 0: `ct_base` holds the
    // ciphertexts BEFORE the automorphism, polynomial 0 of the result is galois(c0) + update0 -- coefficient k takes source
    // coefficient k g^-1 mod 2N, negated past N (PolyRq/Galois.swift:115-143) -- and polynomial 1 is update1;
    // expand_shift != 0 on top of that: one step of PirUtil.expand (PirUtil.swift:204-236), out = the children
    // own + c' and (own - c') x^expand_shift of every polynomial (rns_kernels.hpp ExpandTargets says where they go; own_base:
    // the ciphertexts the children are formed with).  poly_base: the launch's first polynomial in ct_base / own_base / out
    // (the spread and product slabs of a run of equal keys are passed from their own start).
    uint32_t galois_inverse, expand_shift;
    const uint64_t* own_base;
    const uint32_t* targets_table;
    size_t targets_group_size, targets_group_stride;
    size_t poly_base;
};

// How many twiddles of the first (gather-heavy) pass are requested before the rows are loaded and then kept in flight
// through that pass (inverse_row_head).  1: the first one only, requested once the rows are in (the schedule of rounds 2-3, and
// of every limb-wise kernel: their twiddles are 6 registers each).  The plain-slab transform on the shift-folded products
// (4 registers per twiddle, 62 of its 64 in use) takes three: -2.9 % at N = 8192 (two: -2.0 %; four spill: +7.7 %;
// profiles/r05ao_inverse_head_twiddles_ab.txt) -- with no gathers at all the kernel would be 12 % faster (r05an).
template <int LOGN, int LOGE, int MODE, int SOURCE, int ROWS>
constexpr int kInverseHeadTwiddles = (MODE == kModeFoldLazy && SOURCE == 0) ? 3 : 1;

template <int LOGN, int LOGT, int MODE, int SOURCE = kInverseFromSlab, int ROWS = 1>
__global__ void __launch_bounds__(1 << LOGT, min_waves_per_simd(LOGN - LOGT, ROWS))
    ntt_inverse_tiled(uint64_t* __restrict__ slab, const DeviceContext ctx, const RowMap map,
                      const InverseSource source_spec) {
    constexpr bool TENSOR = SOURCE == kInverseFromTensor;
    constexpr bool SCALED = SOURCE == kInverseFromTensor || SOURCE == kInverseFromSlabScaled;
    constexpr bool FROM_SLAB = SOURCE == kInverseFromSlab || SOURCE == kInverseFromSlabScaled;
    constexpr bool KEYMAC = is_key_mac(SOURCE), FINISH = SOURCE == kInverseFromKeyMacFinish;
    constexpr int INPUT_STAGES = (TENSOR || KEYMAC) && kLazyTransformInput<MODE> ? kLazyInputStages : 0;
    constexpr bool LATE = KEYMAC && ROWS == 2;  // (ntt_rows.hpp step_lane)
    static_assert(ROWS == 1 || SOURCE != kInverseFromSlabScaled, "scaled plain slabs go one row per workgroup");
    const uint64_t* __restrict__ tensor_source = source_spec.first;
    constexpr int LOGE = LOGN - LOGT;
    constexpr int E = 1 << LOGE;
    using S = Schedule<LOGN, LOGE>;
    using O = PassOrder<LOGN, LOGE, kTopPartialOrder<LOGN, LOGE, true, FROM_SLAB>>;
    constexpr int LOW = O::LOW;  // the width of the pass on the low bits: the layout the rows are loaded in
    static_assert(S::P >= 1 && S::P <= 5, "unsupported pass count");
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
    const uint32_t tid = threadIdx.x;
    uint32_t record, within;
    size_t rows[ROWS];
    if constexpr (!FROM_SLAB) {
        // records (item, c) of one item read the same source rows: one replica set per (group of ROWS consecutive
        // items, band row); the workgroup transforms record (item + k, c) for k < ROWS
        constexpr uint32_t REPLICAS = TENSOR ? 3 : 2;
        uint32_t set, c, group;
        locate_replica(blockIdx.x, gridDim.x / REPLICAS, REPLICAS, set, c);
        locate(map, set, group, within);
        record = (map.record_base + group * ROWS) * REPLICAS + c;  // of the first item; record_base counts items here
#pragma unroll
        for (int k = 0; k < ROWS; ++k)
            rows[k] = size_t(record + k * REPLICAS) * map.record_rows + map.band_offset + within;
    } else {
        locate_rows<ROWS>(map, blockIdx.x, rows, record, within);
    }
    const uint32_t mi = map.mod_base + within;
    const DeviceModulus mod = ctx.moduli[mi];
    const Twiddles<MODE> tw(ctx, true, mi, LOGN, 0, kLaneMajorTwiddles<LOGN, LOGT, MODE, true, FROM_SLAB>);
    uint64_t v[ROWS][E];

    if constexpr (S::P == 1) {
        const BufferResource x = make_resource(slab + (rows[0] << LOGN), 8u << LOGN);
        global_load<LOGN, LOGE, 0, LOGN>(v[0], tid, x);
        const TwiddleWords head[1] = {inverse_first_twiddle<LOGN, LOGE, 0, LOGN, MODE, false>(tw, tid)};
        inverse_pass<LOGN, LOGE, 0, LOGN, MODE, false, 1, SCALED, 0, LOGN, 0, 1>(v, tid, tw, mod, true, head);
        global_store<LOGN, LOGE, 0, LOGN>(v[0], tid, x);
    } else {
        // every transpose but the last one (into the top pass) stays inside a wave
        constexpr int HEAD = kInverseHeadTwiddles<LOGN, LOGE, MODE, SOURCE, ROWS>;
        TwiddleWords head[HEAD];
        if constexpr (HEAD > 1) inverse_row_head<LOGN, LOGE, MODE, O::kTop, HEAD>(head, tw, tid);
        if constexpr (TENSOR) {
            const size_t first_item = record / 3;
            const uint32_t c = static_cast<uint32_t>(record - first_item * 3);  // wave-uniform
            const size_t poly_words = static_cast<size_t>(map.record_rows) << LOGN;
            const uint64_t p = mod.p, factor = mod.product_factor;
            const int shift = static_cast<int>(mod.product_shift);
            const uint32_t lane_words = lane_part<LOGN, LOGE, 0, LOW>(tid);
#pragma unroll
            for (int k = 0; k < ROWS; ++k) {
                const uint64_t* const source = tensor_source + (first_item + k) * 4 * poly_words +
                                               (static_cast<size_t>(map.band_offset + within) << LOGN);
#pragma unroll
                for (int r = 0; r < E; r += 2) {
                    const size_t at = register_part<LOGN, LOGE, 0, LOW>(r) + lane_words;
                    if (c != 1) {
                        const U64x2 a = *reinterpret_cast<const U64x2*>(source + (c == 0 ? 0 : 1) * poly_words + at);
                        const U64x2 b = *reinterpret_cast<const U64x2*>(source + (c == 0 ? 2 : 3) * poly_words + at);
                        if constexpr (kLazyTransformInput<MODE>) {
                            // (these moduli are 41 .. 61 bits: wide_shift != 0, p^2 inside the bounded reduction's range)
                            v[k][r] = reduce_product_sum_bounded_lazy(product_sum_first(a.x, b.x), mod);
                            v[k][r + 1] = reduce_product_sum_bounded_lazy(product_sum_first(a.y, b.y), mod);
                        } else {
                            v[k][r] = barrett_mul(a.x, b.x, p, factor, shift);
                            v[k][r + 1] = barrett_mul(a.y, b.y, p, factor, shift);
                        }
                    } else {
                        const U64x2 a0 = *reinterpret_cast<const U64x2*>(source + at);
                        const U64x2 a1 = *reinterpret_cast<const U64x2*>(source + poly_words + at);
                        const U64x2 b0 = *reinterpret_cast<const U64x2*>(source + 2 * poly_words + at);
                        const U64x2 b1 = *reinterpret_cast<const U64x2*>(source + 3 * poly_words + at);
                        // a0 b1 + a1 b0 as one exact 128-bit sum and one reduction (two products < 2^125); below 2^61 the
                        // sum 2 p^2 is inside the one-word-quotient Barrett's bound 2^(64 + wide_shift)
                        ProductSum cross0 = product_sum_first(a0.x, b1.x), cross1 = product_sum_first(a0.y, b1.y);
                        product_sum_add(cross0, a1.x, b0.x);
                        product_sum_add(cross1, a1.y, b0.y);
                        if constexpr (kLazyTransformInput<MODE>) {
                            v[k][r] = reduce_product_sum_bounded_lazy(cross0, mod);
                            v[k][r + 1] = reduce_product_sum_bounded_lazy(cross1, mod);
                        } else if (mod.wide_shift != 0) {  // wave-uniform
                            v[k][r] = reduce_product_sum_bounded(cross0, mod);
                            v[k][r + 1] = reduce_product_sum_bounded(cross1, mod);
                        } else {
                            v[k][r] = reduce_product_sum(cross0, mod);
                            v[k][r + 1] = reduce_product_sum(cross1, mod);
                        }
                    }
                }
            }
        } else if constexpr (KEYMAC) {
            const size_t first_poly = record >> 1, c = record & 1;  // record = poly * 2 + c
            const uint32_t L = source_spec.L, top_rows = source_spec.top_rows;
            const uint32_t r = map.band_offset + within;
            const uint32_t key_row = (r == L) ? top_rows - 1 : r;  // Bfv+Keys.swift:153
            // (flat 64-bit addresses: the same loads through scalar buffer descriptors with a scalar offset per term --
            // no address arithmetic on the vector ALU -- measured 1.4 % slower, profiles/r04j_keymac_loads_and_column_words_ab.txt)
            const uint32_t lane_words = lane_part<LOGN, LOGE, 0, LOW>(tid);
#pragma unroll
            for (int k = 0; k < ROWS; ++k) {
#pragma unroll
                for (int q = 0; q < E; q += 2) {
                    const uint32_t at = register_part<LOGN, LOGE, 0, LOW>(q);
                    // the words of term j + 1 are requested before term j is accumulated (the count L is a run-time
                    // value: the loop is not unrolled, and without the request ahead every term would wait out its own
                    // L2 round trip); past the last term the request repeats it -- a load behind a branch would drain
                    // the queue
                    const uint64_t* const flat_spread =
                        source_spec.first + (((first_poly + k) * L * (L + 1) + r) << LOGN) + lane_words + at;  // + j (L+1) N
                    const uint64_t* const flat_key =
                        source_spec.second + ((c * top_rows + key_row) << LOGN) + lane_words + at;          // + j 2 top_rows N
                    U64x2 xs = *reinterpret_cast<const U64x2*>(flat_spread);
                    U64x2 ks = *reinterpret_cast<const U64x2*>(flat_key);
                    ProductSum acc0 = product_sum_zero(), acc1 = product_sum_zero();
                    for (uint32_t j = 0; j < L; ++j) {
                        const uint32_t ahead = j + 1 < L ? j + 1 : j;
                        const U64x2 xn = *reinterpret_cast<const U64x2*>(flat_spread + ((size_t(ahead) * (L + 1)) << LOGN));
                        const U64x2 kn = *reinterpret_cast<const U64x2*>(flat_key + ((size_t(ahead) * 2 * top_rows) << LOGN));
                        // (limb-wise schedule = every modulus below 2^55: canonical operands keep the middle column of a
                        // sum from wrapping for 127 products, so its carry counts are not kept -- 5 instructions per
                        // product instead of 7; L <= 64 terms stay inside kNarrowProductSumCadence)
                        product_sum_add_one<is_split(MODE)>(acc0, xs.x, ks.x);
                        product_sum_add_one<is_split(MODE)>(acc1, xs.y, ks.y);
                        xs = xn;
                        ks = kn;
                    }
                    // (the one-word-quotient Barrett where the sum allows it: L products of canonical words stay below
                    // L p^2 < 2^(64 + wide_shift) = 2^(63 + bits(p)) whenever L p < 2^63)
                    if (kKeyMacBoundedReduce && mod.wide_shift != 0 && L <= 8 && uint64_t(L) * mod.p < (uint64_t(1) << 63)) {  // wave-uniform
                        if constexpr (kLazyTransformInput<MODE>) {
                            v[k][q] = reduce_product_sum_bounded_lazy(acc0, mod);
                            v[k][q + 1] = reduce_product_sum_bounded_lazy(acc1, mod);
                        } else {
                            v[k][q] = reduce_product_sum_bounded(acc0, mod);
                            v[k][q + 1] = reduce_product_sum_bounded(acc1, mod);
                        }
                    } else {
                        v[k][q] = reduce_product_sum(acc0, mod);
                        v[k][q + 1] = reduce_product_sum(acc1, mod);
                    }
                }
            }
        } else {
            [[maybe_unused]] const uint64_t p = mod.p;
            // (plain slabs may be transformed OUT OF PLACE: rows read from source_spec.first, laid out like the slab, when it is set)
            const uint64_t* load_base = slab;
            if constexpr (SOURCE == kInverseFromSlab) load_base = tensor_source != nullptr ? tensor_source : slab;
#pragma unroll
            for (int k = 0; k < ROWS; ++k) {
                const BufferResource in = make_resource(load_base + (rows[k] << LOGN), 8u << LOGN);
                if constexpr (kStagedLoad<LOGN, LOGE, LOW>) {
                    global_load_staged<LOGN, LOGE, LOW>(v[k], tid, in, lds);
                } else {
                    global_load<LOGN, LOGE, 0, LOW>(v[k], tid, in);
                }
            }
        }
        if constexpr (HEAD == 1) inverse_row_head<LOGN, LOGE, MODE, O::kTop, HEAD>(head, tw, tid);
        constexpr int LOL = LOGN - LOGE;
        if constexpr (FINISH) {
            // key_switch_finish_kernel's arithmetic (rns_kernels.hip) on each row as its last pass completes: with x_ks the
            // q_ks word of the coefficient and c its centred representative, out = (x - c) q_ks^-1 mod q_r (+ the ciphertext
            // word).  |c| <= q_ks / 2 needs a reduction mod q_r only where q_r is the smaller one (`wide`, wave-uniform: the
            // 29-bit modulus next to the 60-bit ones of the reference's n_8192_logq_29_60_60); the word loop exists in both
            // forms.
            // The end's parameters are read from the kernel-argument segment WHERE THEY ARE USED (scalar loads off the segment's
            // base), not as kernel arguments: those are all loaded in the entry block, and this kernel then parks them in
            // vector-register lanes across the whole transform -- two of its 64 registers, paid for in scratch.
            struct KernelArguments {
                uint64_t* slab;
                DeviceContext ctx;
                RowMap map;
                InverseSource source_spec;
            };
            // The AMDGPU kernel-argument segment lays the arguments out in order, each at its natural alignment -- the rule
            // of a C++ struct with the same members.  kernarg_offset derives the offset of argument 3 from THIS kernel's own
            // parameter list: adding, removing or reordering a parameter without touching the mirror fails here, not at run
            // time with garbage in out / ct_base / poly_base.
            using ThisKernel = decltype(&ntt_inverse_tiled<LOGN, LOGT, MODE, SOURCE, ROWS>);
            static_assert(kernarg_offset<3>(static_cast<ThisKernel>(nullptr)) == offsetof(KernelArguments, source_spec) &&
                              kernarg_count(static_cast<ThisKernel>(nullptr)) == 4,
                          "KernelArguments must mirror ntt_inverse_tiled's parameter list");
            static_assert(std::is_trivially_copyable<InverseSource>::value && std::is_standard_layout<KernelArguments>::value,
                          "kernel arguments are copied byte for byte into the segment");
            using ConstSpec = const __attribute__((address_space(4))) InverseSource;
            using ConstByte = const __attribute__((address_space(4))) char;
            ConstSpec* const end_spec = (ConstSpec*)((ConstByte*)__builtin_amdgcn_kernarg_segment_ptr() +
                                                     offsetof(KernelArguments, source_spec));
            const uint32_t L = source_spec.L, r = map.band_offset + within;
            const uint64_t p = mod.p, q_last = ctx.moduli[L].p, half = q_last >> 1;
            const U64x2 inverse_q_last = load_twiddle(ctx.inverse_q_last + size_t(L) * ctx.moduli_stride + r);
            const bool wide = half >= p;
            constexpr int kEndPlain = 0, kEndGalois = 1, kEndExpand = 2;
            auto finish = [&](int k, uint64_t (&row)[E]) {
                const uint32_t lane = step_lane<MODE, LATE>(tid);
                const uint32_t lane_words = lane_part<LOGN, LOGE, LOL, LOGE>(lane), lane_bytes = lane_words << 3;
                constexpr int CHUNK = 4;  // words in flight: the q_ks and ciphertext words of a chunk are requested together
                auto word = [](Dwordx2 w) { return pack64(w.x, w.y); };
                const size_t pc = record + 2 * k;  // polynomial * 2 + c (the rows are the same column of consecutive polynomials)
                const size_t poly = end_spec->poly_base + (pc >> 1);  // in the ciphertext / output slabs
                const uint32_t c = static_cast<uint32_t>(pc & 1);
                const BufferResource last_row = make_resource(slab + ((pc * (L + 1) + L) << LOGN), 8u << LOGN);
                const uint32_t galois_inverse = end_spec->galois_inverse, expand_shift = end_spec->expand_shift;
                // polynomial c of the ciphertext: added as it lies (relinearize), or -- c0 under an automorphism -- gathered
                const bool gathered = galois_inverse != 0 && c == 0;
                const bool add = galois_inverse != 0 ? gathered : c < end_spec->added_polys;  // wave-uniform
                const BufferResource added_row =
                    add ? make_resource(end_spec->ct_base + poly * end_spec->ct_stride + ((size_t(c) * L + r) << LOGN), 8u << LOGN)
                        : last_row;  // (without an addend the q_ks row is read twice)
                auto words_of_row = [&](auto reduce_magnitude, auto end_tag) {
                    constexpr int END = decltype(end_tag)::value;
                    // where the row goes: polynomial c of ciphertext `poly`, or -- an expand step -- of its two children.  The
                    // descriptors are built inside the end that uses them: held across all three they cost scalar registers
                    // that this kernel parks in vector-register lanes
                    size_t first_ct = poly, second_ct = 0;
                    [[maybe_unused]] bool first_doubled = false, second_doubled = false;
                    if constexpr (END == kEndExpand) {
                        first_ct = 2 * poly;
                        second_ct = 2 * poly + 1;
                        if (end_spec->targets_table != nullptr) {
                            using ConstWord = const __attribute__((address_space(4))) uint32_t;
                            const size_t group = poly / end_spec->targets_group_size, parent = poly - group * end_spec->targets_group_size;
                            const uint32_t first = *(ConstWord*)(end_spec->targets_table + 4 * parent + 1);
                            const uint32_t second = *(ConstWord*)(end_spec->targets_table + 4 * parent + 3);
                            first_ct = group * end_spec->targets_group_stride + (first >> 1);
                            second_ct = group * end_spec->targets_group_stride + (second >> 1);
                            first_doubled = (first & 1u) != 0;
                            second_doubled = (second & 1u) != 0;
                        }
                    }
                    const BufferResource out_row = make_uniform_resource(end_spec->out + (((first_ct * 2 + c) * L + r) << LOGN), 8u << LOGN);
                    [[maybe_unused]] BufferResource moved_row = out_row, own_row = last_row;
                    if constexpr (END == kEndExpand) {
                        moved_row = make_uniform_resource(end_spec->out + (((second_ct * 2 + c) * L + r) << LOGN), 8u << LOGN);
                        own_row = make_resource(end_spec->own_base + poly * end_spec->ct_stride + ((size_t(c) * L + r) << LOGN), 8u << LOGN);
                    }
#pragma unroll
                    for (int base = 0; base < E; base += CHUNK) {
                        uint64_t last[CHUNK], added[CHUNK];
                        [[maybe_unused]] uint64_t own[CHUNK];
                        [[maybe_unused]] bool negate[CHUNK];
#pragma unroll
                        for (int e = 0; e < CHUNK; ++e) {
                            const uint32_t at = register_part<LOGN, LOGE, LOL, LOGE>(base + e) << 3;
                            last[e] = word(__builtin_amdgcn_raw_buffer_load_b64(last_row, lane_bytes, at, row_load_policy<LOGN>()));
                            if constexpr (END == kEndPlain) {
                                added[e] = word(__builtin_amdgcn_raw_buffer_load_b64(added_row, lane_bytes, at, row_load_policy<LOGN>()));
                            } else {
                                // coefficient idx of galois(c0) = -+ coefficient idx g^-1 mod 2N of c0 (row c = 1: not used)
                                const uint32_t idx = register_part<LOGN, LOGE, LOL, LOGE>(base + e) | lane_words;
                                const uint32_t doubled = (idx * galois_inverse) & ((2u << LOGN) - 1u);
                                negate[e] = (doubled >> LOGN) != 0;
                                added[e] = word(__builtin_amdgcn_raw_buffer_load_b64(added_row, (doubled & ((1u << LOGN) - 1u)) << 3, 0, 0));
                                if constexpr (END == kEndExpand)
                                    own[e] = word(__builtin_amdgcn_raw_buffer_load_b64(own_row, lane_bytes, at, row_load_policy<LOGN>()));
                            }
                        }
#pragma unroll
                        for (int e = 0; e < CHUNK; ++e) {
                            const uint32_t at = register_part<LOGN, LOGE, LOL, LOGE>(base + e) << 3;
                            // straight-line: x - c = x + (c < 0 ? |c| : p - |c|) mod p, an absent addend is zero
                            const uint64_t shifted = add_mod_uniform(last[e], half, q_last);
                            const bool negative = shifted < half;
                            uint64_t t = negative ? half - shifted : shifted - half;
                            if constexpr (decltype(reduce_magnitude)::value) t = barrett_reduce64_uniform(t, p, mod.barrett64);
                            const uint64_t difference = csub_uniform(row[base + e] + (negative ? t : p - t), p);
                            const uint64_t update = shoup_mul_uniform(difference, inverse_q_last.x, inverse_q_last.y, p);
                            uint64_t addend = add ? added[e] : 0;
                            if constexpr (END != kEndPlain) addend = negate[e] && addend != 0 ? p - addend : addend;
                            const uint64_t result = csub_uniform(addend + update, p);  // c' (or ct + update)
                            if constexpr (END != kEndExpand) {
                                const Dwordx2 words = {lo32(result), hi32(result)};
                                __builtin_amdgcn_raw_buffer_store_b64(words, out_row, lane_bytes, at, row_policy<LOGN>());
                            } else {
                                // children own + c' and (own - c') x^shift: the second one lands at idx + shift mod 2N,
                                // negated past N
                                uint64_t sum = csub_uniform(own[e] + result, p);
                                if (first_doubled) sum = csub_uniform(sum + sum, p);
                                const Dwordx2 first_words = {lo32(sum), hi32(sum)};
                                __builtin_amdgcn_raw_buffer_store_b64(first_words, out_row, lane_bytes, at, row_policy<LOGN>());
                                const uint32_t idx = register_part<LOGN, LOGE, LOL, LOGE>(base + e) | lane_words;
                                const uint32_t landed = (idx + expand_shift) & ((2u << LOGN) - 1u);
                                uint64_t moved = csub_uniform(own[e] + (p - result), p);
                                moved = (landed >> LOGN) != 0 && moved != 0 ? p - moved : moved;
                                if (second_doubled) moved = csub_uniform(moved + moved, p);
                                const Dwordx2 moved_words = {lo32(moved), hi32(moved)};
                                __builtin_amdgcn_raw_buffer_store_b64(moved_words, moved_row, (landed & ((1u << LOGN) - 1u)) << 3, 0,
                                                                      row_policy<LOGN>());
                            }
                        }
                    }
                };
                auto with_end = [&](auto reduce_magnitude) {
                    if (expand_shift != 0) words_of_row(reduce_magnitude, std::integral_constant<int, kEndExpand>{});
                    else if (galois_inverse != 0) words_of_row(reduce_magnitude, std::integral_constant<int, kEndGalois>{});
                    else words_of_row(reduce_magnitude, std::integral_constant<int, kEndPlain>{});
                };
                if (wide) with_end(std::true_type{});
                else with_end(std::false_type{});
            };
            inverse_row<LOGN, LOGE, MODE, ROWS, SCALED, INPUT_STAGES, LOGN, O::kTop, HEAD, LATE>(v, tid, tw, mod, lds, head, finish);
        } else {
            inverse_row<LOGN, LOGE, MODE, ROWS, SCALED, INPUT_STAGES, LOGN, O::kTop, HEAD, LATE>(v, tid, tw, mod, lds, head);
            const uint32_t store_lane = step_lane<MODE, LATE>(tid);
#pragma unroll
            for (int k = 0; k < ROWS; ++k)
                global_store<LOGN, LOGE, LOL, LOGE>(v[k], store_lane, make_resource(slab + (rows[k] << LOGN), 8u << LOGN));
        }
    }
}

// ---- interleaved rows: N = 2^(13 + LOGS) as 2^LOGS sub-rows of 8192 words --------------------------------------------
// A row of 16384 (32768) words fills the CU's LDS as one tile.  Taken as its 2 (4) sub-rows "element index mod 2^LOGS"
// it is the row group of the N = 8192 kernel: sub-row h holds the words idx = i 2^LOGS + h, and every stage on an element
// bit b >= LOGS pairs words of ONE sub-row (i and i + 2^(b - LOGS)) under a twiddle that only depends on idx >> (b + 1)
// = i >> (b - LOGS + 1) -- the same for all sub-rows, and the very index the 8192-point transform of sub-row words uses
// in its own stage: W[2^s + group] of the degree's table, s < 13, is read exactly as the smaller transform would read
// its own table.  So the first 13 forward stages are forward_row<13> over the sub-rows (one 76 KB tile in turn, every
// twiddle fetched once for all sub-rows, two workgroups per CU at N = 16384), and the remaining LOGS stages pair words
// of different sub-rows held in the same register of the same lane (cross stages, gathered twiddles).  The inverse
// mirrors it: cross stages first, then inverse_row<13> against the tail of the inverse table (which lists its stages
// from the low bit up, ntt_common.hpp inverse_twiddle: block m at N - 2m + 1, so the sub-transform's index plus
// N - 8192).  In the top-pass layout a lane's words i = r 1024 + tid of all sub-rows are 16 (32) contiguous bytes of
// the row: the forward load and the inverse store move whole lines with 16-byte accesses.
constexpr int kSubLogN = 13, kSubLogT = 10, kSubLogE = kSubLogN - kSubLogT;
// The sub-rows' 8192-point transforms read lane-major stage blocks like the N = 8192 kernels (ntt_rows.hpp kLaneMajorTwiddles,
// built by PolyContext::upload inside the degree's own tables): with the plain tables a wave's gather of one twiddle touches
// every 2^g-th entry of a run, up to 64 cache lines for 1 KiB -- made-up twiddles showed the gathers cost the inverse 19 % and
// the forward transform 8 % at N = 16384 (profiles/r06j_n16384_what_binds.txt).
template <int MODE>
constexpr int kInterleavedLaneMajor = (is_split(MODE) || is_fold(MODE)) ? 1 : 0;

// forward cross stage c (global stage 13 + c) pairs sub-row bit LOGS - 1 - c; twiddle 2^(13+c) + (idx >> (LOGS - c))
// Twiddles in flight: a cross stage's twiddles serve ONE butterfly per pair of sub-rows each (N = 16384: 8 gathers per lane, one
// butterfly between two of them), so with one request ahead every gather's L2 round trip is waited out in full -- the shift-folded
// products (4 registers per twiddle) keep kTwiddlesAhead of them in flight, the limb-wise ones (6) one.
// (measured at N = 16384, profiles/r06n_n16384_cross_stage_prefetch_ab.txt: two or three limb-wise twiddles in flight spill
// -- 16 / 40 B -- and lose 1 / 8 %; the first one requested before the row's loads: +-0)
constexpr int kCrossAheadSplit = 1;
constexpr bool kCrossEarlySplit = false;
template <int MODE>
constexpr int kCrossAhead = MODE == kModeFoldLazy ? 3 : (is_split(MODE) ? kCrossAheadSplit : 1);
// the inverse carries the finished sub-rows through its cross stages, so its third twiddle in flight costs it 20 B of scratch:
// two measure 4-5 % faster at N = 16384 and keep none (profiles/r06v_interleaved_fold_inverse_ab.txt)
template <int MODE>
constexpr int kCrossAheadInverse = MODE == kModeFoldLazy ? 2 : kCrossAhead<MODE>;
template <int LOGS, int MODE>
__device__ __forceinline__ void forward_cross_stages(uint64_t (&v)[1 << LOGS][1 << kSubLogE], uint32_t tid,
                                                     const Twiddles<MODE>& tw, uint64_t p) {
    constexpr int E = 1 << kSubLogE, R = Schedule<kSubLogN, kSubLogE>::R, AHEAD = kCrossAhead<MODE>;
    static_assert(!is_split(MODE) || 1 + ((kSubLogN + LOGS) << Lazy<MODE>::kProductLog) <= 511, "growth stays below 2^9 p");
    const uint64_t neg_p = Lazy<MODE>::reduction_constant(p);
    const uint64_t half_bound = p << Lazy<MODE>::kProductLog;
    const FoldConstants fc = mode_fold_constants<MODE>(p);
    const uint32_t lane_words = lane_part<kSubLogN, kSubLogE, 0, R>(tid);
#pragma unroll
    for (int c = 0; c < LOGS; ++c) {
        const int count = E << c;  // twiddles of the stage per lane: one per (word, upper sub-row bits)
        auto request = [&](int k) {
            const int r = k >> c, upper = k & ((1 << c) - 1);
            const uint32_t fixed = (1u << (kSubLogN + c)) + (register_part<kSubLogN, kSubLogE, 0, R>(r) << c) + upper;
            return fetch_twiddle<MODE, false>(tw, lane_words << c, fixed);
        };
        TwiddleWords ring[AHEAD];
#pragma unroll
        for (int a = 0; a < AHEAD; ++a) ring[a] = request(a < count ? a : count - 1);
#pragma unroll
        for (int k = 0; k < count; ++k) {
            const TwiddleWords w = ring[k % AHEAD];
            if (k + AHEAD < count) {
                ring[k % AHEAD] = request(k + AHEAD);
                __builtin_amdgcn_sched_barrier(0);
            }
            const int r = k >> c, upper = k & ((1 << c) - 1);
            const int span = 1 << (LOGS - 1 - c);  // sub-row distance of a pair
#pragma unroll
            for (int low = 0; low < span; ++low) {
                const int h = (upper << (LOGS - c)) | low;
                forward_butterfly<MODE>(v[h][r], v[h + span][r], w, false, neg_p, half_bound, true, fc);
            }
            if (k + AHEAD < count) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// inverse cross stage c pairs sub-row bit c (element bit c): m = N >> (c + 1) groups, twiddle (N - 2m + 1) + (idx >> (c + 1))
// PRIOR: the words enter in the lazy range of a row whose first PRIOR stages already ran (the fused loads' [0, 5p) words)
// `head`: the first kTwiddlesAhead twiddles of stage 0, requested by the caller BEFORE it loads the row (inverse_cross_head): the
// transform opens with this stage, and its gathers then travel beside the row's loads instead of starting behind them.
template <int LOGS, int MODE>
__device__ __forceinline__ TwiddleWords inverse_cross_twiddle(const Twiddles<MODE>& tw, uint32_t lane_words, int c, int k) {
    constexpr int R = Schedule<kSubLogN, kSubLogE>::R;
    constexpr uint32_t N = 1u << (kSubLogN + LOGS);
    const uint32_t m = N >> (c + 1);
    const int upper_bits = LOGS - 1 - c;  // sub-row bits above the paired one
    const int r = k >> upper_bits, upper = k & ((1 << upper_bits) - 1);
    const uint32_t fixed = (N - 2 * m + 1) + (register_part<kSubLogN, kSubLogE, 0, R>(r) << upper_bits) + upper;
    return fetch_twiddle<MODE, false>(tw, lane_words << upper_bits, fixed);
}
template <int LOGS, int MODE>
__device__ __forceinline__ void inverse_cross_head(TwiddleWords (&head)[kCrossAheadInverse<MODE>], uint32_t tid, const Twiddles<MODE>& tw) {
    constexpr int R = Schedule<kSubLogN, kSubLogE>::R;
    const uint32_t lane_words = lane_part<kSubLogN, kSubLogE, 0, R>(tid);
#pragma unroll
    for (int a = 0; a < kCrossAheadInverse<MODE>; ++a) head[a] = inverse_cross_twiddle<LOGS, MODE>(tw, lane_words, 0, a);
}
template <int LOGS, int MODE, int PRIOR = 0>
__device__ __forceinline__ void inverse_cross_stages(uint64_t (&v)[1 << LOGS][1 << kSubLogE], uint32_t tid,
                                                     const Twiddles<MODE>& tw, uint64_t p,
                                                     const TwiddleWords (&head)[kCrossAheadInverse<MODE>]) {
    constexpr int E = 1 << kSubLogE, R = Schedule<kSubLogN, kSubLogE>::R, H = Lazy<MODE>::kInverseCapLog, AHEAD = kCrossAheadInverse<MODE>;
    static_assert(AHEAD <= E, "the head twiddles are all of stage 0");
    const uint64_t neg_p = Lazy<MODE>::reduction_constant(p);
    const FoldConstants fc = mode_fold_constants<MODE>(p);
    const uint32_t lane_words = lane_part<kSubLogN, kSubLogE, 0, R>(tid);
#pragma unroll
    for (int c = 0; c < LOGS; ++c) {
        const int in_shift = inverse_in_shift<MODE>(c + PRIOR);
        const uint64_t bound = p << in_shift;
        const bool fold = in_shift + 1 > H;
        const int upper_bits = LOGS - 1 - c;  // sub-row bits above the paired one
        const int count = E << upper_bits;
        TwiddleWords ring[AHEAD];
#pragma unroll
        for (int a = 0; a < AHEAD; ++a)
            ring[a] = c == 0 ? head[a] : inverse_cross_twiddle<LOGS, MODE>(tw, lane_words, c, a < count ? a : count - 1);
#pragma unroll
        for (int k = 0; k < count; ++k) {
            const TwiddleWords w = ring[k % AHEAD];
            if (k + AHEAD < count) {
                ring[k % AHEAD] = inverse_cross_twiddle<LOGS, MODE>(tw, lane_words, c, k + AHEAD);
                __builtin_amdgcn_sched_barrier(0);
            }
            const int upper = k & ((1 << upper_bits) - 1), r = k >> upper_bits;
            const int span = 1 << c;
#pragma unroll
            for (int low = 0; low < span; ++low) {
                const int h = (upper << (c + 1)) | low;
                inverse_butterfly<MODE>(v[h][r], v[h + span][r], w, false, p, neg_p, bound, fold, fc, split_signed_bias(p));
            }
            if (k + AHEAD < count) __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// Words i = r 1024 + tid (top-pass layout of the sub-rows) of all sub-rows: 8 2^LOGS contiguous bytes of the row per lane.
template <int LOGS, bool STORE, int POLICY = row_policy<kSubLogN + LOGS>()>
__device__ __forceinline__ void interleaved_top_words(uint64_t (&v)[1 << LOGS][1 << kSubLogE], uint32_t tid, BufferResource row) {
    constexpr int ROWS = 1 << LOGS;
    const uint32_t lane_bytes = tid << (3 + LOGS);
#pragma unroll
    for (int r = 0; r < (1 << kSubLogE); ++r) {
#pragma unroll
        for (int h = 0; h < ROWS; h += 2) {
            const uint32_t fixed = (static_cast<uint32_t>(r) << (kSubLogT + 3 + LOGS)) + h * 8;
            if constexpr (STORE) {
                const Dwordx4 words = {lo32(v[h][r]), hi32(v[h][r]), lo32(v[h + 1][r]), hi32(v[h + 1][r])};
                __builtin_amdgcn_raw_buffer_store_b128(words, row, lane_bytes, fixed, POLICY);
            } else {
                const Dwordx4 words = __builtin_amdgcn_raw_buffer_load_b128(row, lane_bytes, fixed, POLICY);
                v[h][r] = pack64(words.x, words.y);
                v[h + 1][r] = pack64(words.z, words.w);
            }
        }
    }
}
// Words of the low-pass layout of the sub-rows (element_index<13, 3, 0, 1>: runs of 2 words of a sub-row per lane, i.e.
// runs of 2^(LOGS + 1) words of the row) of all sub-rows: 16-byte accesses as the words lie.
template <int LOGS, bool STORE>
__device__ __forceinline__ void interleaved_low_words(uint64_t (&v)[1 << LOGS][1 << kSubLogE], uint32_t tid, BufferResource row) {
    constexpr int ROWS = 1 << LOGS, R = Schedule<kSubLogN, kSubLogE>::R, POLICY = row_policy<kSubLogN + LOGS>();
    const uint32_t lane_bytes = lane_part<kSubLogN, kSubLogE, 0, R>(tid) << (3 + LOGS);
#pragma unroll
    for (int r = 0; r < (1 << kSubLogE); ++r) {
#pragma unroll
        for (int h = 0; h < ROWS; h += 2) {
            const uint32_t fixed = (register_part<kSubLogN, kSubLogE, 0, R>(r) << (3 + LOGS)) + h * 8;
            if constexpr (STORE) {
                const Dwordx4 words = {lo32(v[h][r]), hi32(v[h][r]), lo32(v[h + 1][r]), hi32(v[h + 1][r])};
                __builtin_amdgcn_raw_buffer_store_b128(words, row, lane_bytes, fixed, POLICY);
            } else {
                const Dwordx4 words = __builtin_amdgcn_raw_buffer_load_b128(row, lane_bytes, fixed, POLICY);
                v[h][r] = pack64(words.x, words.y);
                v[h + 1][r] = pack64(words.z, words.w);
            }
        }
    }
}

// The same words in whole cache lines (ntt_common.hpp global_store_staged / global_load_staged): a lane's run is 2^(LOGS+1)
// contiguous words of the row, i.e. C = 2^LOGS 16-byte chunks; the wave's block of 64 C chunks per run goes through the
// wave's own 592-slot slice of the tile (free on both occasions: the exchange next to the low pass never leaves the
// wave) and crosses the memory interface in lane order, 1 KiB per instruction.
constexpr bool kInterleavedStaged = true;
template <int LOGS, bool STORE>
__device__ __forceinline__ void interleaved_low_words_staged(uint64_t (&v)[1 << LOGS][1 << kSubLogE], uint32_t tid,
                                                             BufferResource row, uint64_t* lds) {
    constexpr int ROWS = 1 << LOGS, C = ROWS, R = Schedule<kSubLogN, kSubLogE>::R, POLICY = row_policy<kSubLogN + LOGS>();
    static_assert(R == 1 && C * 64 * 16 <= 592 * 8, "runs of two sub-row words; the block fits the wave's slice");
    const uint32_t lane = tid & 63u;
    const uint32_t swizzle = (lane >> (4 - LOGS)) & (C - 1);
    char* const block = reinterpret_cast<char*>(lds + (tid >> 6) * 592u);
#pragma unroll
    for (int run = 0; run < (1 << (kSubLogE - R)); ++run) {
        const uint32_t first = (lane_part<kSubLogN, kSubLogE, 0, R>(tid & ~63u) | register_part<kSubLogN, kSubLogE, 0, R>(run << R))
                               << LOGS;  // row word the wave's block starts at
        if constexpr (STORE) {
#pragma unroll
            for (int j = 0; j < C; ++j) {
                const int b = j / (ROWS / 2), h = (j % (ROWS / 2)) * 2;
                *reinterpret_cast<U64x2*>(block + ((lane * C + (j ^ swizzle)) << 4)) = U64x2{v[h][(run << R) + b], v[h + 1][(run << R) + b]};
            }
        }
        U64x2 picked[C];
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const uint32_t chunk = j * 64 + lane, owner = chunk / C, slot = chunk % C;
            const uint32_t owner_swizzle = (owner >> (4 - LOGS)) & (C - 1);
            char* const at = block + ((owner * C + (slot ^ owner_swizzle)) << 4);
            if constexpr (STORE) {
                const U64x2 pair = *reinterpret_cast<const U64x2*>(at);
                const Dwordx4 words = {lo32(pair.x), hi32(pair.x), lo32(pair.y), hi32(pair.y)};
                __builtin_amdgcn_raw_buffer_store_b128(words, row, (first << 3) + (lane << 4), j << 10, POLICY);
            } else {
                const Dwordx4 words = __builtin_amdgcn_raw_buffer_load_b128(row, (first << 3) + (lane << 4), j << 10, POLICY);
                picked[j] = U64x2{pack64(words.x, words.y), pack64(words.z, words.w)};
            }
        }
        if constexpr (!STORE) {
#pragma unroll
            for (int j = 0; j < C; ++j) {
                const uint32_t chunk = j * 64 + lane, owner = chunk / C, slot = chunk % C;
                const uint32_t owner_swizzle = (owner >> (4 - LOGS)) & (C - 1);
                *reinterpret_cast<U64x2*>(block + ((owner * C + (slot ^ owner_swizzle)) << 4)) = picked[j];
            }
#pragma unroll
            for (int j = 0; j < C; ++j) {
                const int b = j / (ROWS / 2), h = (j % (ROWS / 2)) * 2;
                const U64x2 pair = *reinterpret_cast<const U64x2*>(block + ((lane * C + (j ^ swizzle)) << 4));
                v[h][(run << R) + b] = pair.x;
                v[h + 1][(run << R) + b] = pair.y;
            }
        }
    }
}

// SPREAD: the row sources of ntt_forward_tiled (kSourceSpread without the automorphism, kSourceLift, kSourceRows) -- the step
// that would otherwise write the slab is applied to the words as they are loaded, in the top-pass layout of the sub-rows
// (a lane's words of all sub-rows are contiguous bytes of the SOURCE row too).
template <int LOGS, int MODE, int SPREAD = kSourceSlab>
__global__ void __launch_bounds__(1 << kSubLogT, min_waves_per_simd(kSubLogE, 1 << LOGS))
    ntt_forward_interleaved(uint64_t* __restrict__ slab, const DeviceContext ctx, const RowMap map, const SpreadSource spread) {
    constexpr int ROWS = 1 << LOGS, LOGD = kSubLogN + LOGS, E = 1 << kSubLogE;
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
    const uint32_t tid = threadIdx.x;
    uint32_t record, within;
    size_t rows[1];
    if constexpr (SPREAD == kSourceSpread || SPREAD == kSourceLift) {
        // the band_rows output rows of a record are transforms of one source row: one replica set per record
        uint32_t group;
        locate_replica(blockIdx.x, gridDim.x / map.band_rows, map.band_rows, group, within);
        record = map.record_base + group;
        rows[0] = size_t(record) * (map.record_rows == 0 ? map.band_rows : map.record_rows) + map.band_offset + within;
    } else {
        locate_rows<1>(map, blockIdx.x, rows, record, within);
    }
    const uint32_t mi = map.mod_base + within;
    const DeviceModulus mod = ctx.moduli[mi];
    const uint64_t p = mod.p;
    const Twiddles<MODE> tw(ctx, false, mi, LOGD, 0, kInterleavedLaneMajor<MODE>);  // (the cross stages' block is not permuted)
    const BufferResource row = make_resource(slab + (rows[0] << LOGD), 8u << LOGD);
    uint64_t v[ROWS][E];
    if constexpr (SPREAD == kSourceRows) {
        const uint64_t* source;
        if (spread.second == nullptr) {
            source = spread.base + size_t(record) * spread.stride + (static_cast<size_t>(within) << LOGD);
        } else {
            const size_t item = record >> 2, slot = record & 3;
            source = ((slot & 2) != 0 ? spread.second : spread.base) + item * spread.stride + (((slot & 1) * spread.L + within) << LOGD);
        }
        interleaved_top_words<LOGS, false>(v, tid, make_uniform_resource(source, 8u << LOGD));
    } else if constexpr (SPREAD != kSourceSlab) {
        const size_t poly = record / spread.L, j = record - poly * spread.L;  // record = poly * L + j
        // cached: the other rows of this record read the same words
        interleaved_top_words<LOGS, false, 0>(v, tid, make_uniform_resource(spread.base + poly * spread.stride + (j << LOGD), 8u << LOGD));
        if constexpr (SPREAD == kSourceLift) {
            const uint64_t threshold = (spread.plaintext_modulus + 1) >> 1, increment = p - spread.plaintext_modulus;
#pragma unroll
            for (int h = 0; h < ROWS; ++h)
#pragma unroll
                for (int r = 0; r < E; ++r) v[h][r] = v[h][r] < threshold ? v[h][r] : v[h][r] + increment;
        } else {
            // rare (moduli of very different sizes; ntt_forward_tiled): the source row is canonical mod a larger modulus
            const bool reduce = ctx.moduli[j].p > p && !(is_split(MODE) && ctx.moduli[j].p < 2 * p);  // wave-uniform
            if (reduce) {
#pragma unroll
                for (int h = 0; h < ROWS; ++h)
#pragma unroll
                    for (int r = 0; r < E; ++r) v[h][r] = reduce ? barrett_reduce64_uniform(v[h][r], p, mod.barrett64) : v[h][r];
            }
        }
    } else {
        interleaved_top_words<LOGS, false>(v, tid, row);
    }
    forward_row<kSubLogN, kSubLogE, MODE, ROWS, false>(v, tid, tw, p, lds);
    forward_cross_stages<LOGS, MODE>(v, tid, tw, p);
    canonicalize_all<MODE>(v, p);
    // (the store's lane addresses are derived from an opaque copy of the lane index: the compiler otherwise computes them
    // at the top of the kernel and carries them through every pass in scratch)
    if constexpr (kInterleavedStaged) interleaved_low_words_staged<LOGS, true>(v, opaque32(tid), row, lds);
    else interleaved_low_words<LOGS, true>(v, tid, row);
}

// SOURCE: kInverseFromSlab, or the fused loads of ntt_inverse_tiled -- kInverseFromTensor (the BEHZ tensor product) and
// kInverseFromKeyMac (the lazy inner product with the key-switching key) -- formed in the low-pass layout of the sub-rows: the
// words (element i, sub-rows h, h + 1) are 16 contiguous bytes of every source row.
template <int LOGS, int MODE, bool SCALED, int SOURCE = kInverseFromSlab>
__global__ void __launch_bounds__(1 << kSubLogT, min_waves_per_simd(kSubLogE, 1 << LOGS))
    ntt_inverse_interleaved(uint64_t* __restrict__ slab, const DeviceContext ctx, const RowMap map, const InverseSource source_spec) {
    constexpr int ROWS = 1 << LOGS, LOGD = kSubLogN + LOGS, E = 1 << kSubLogE, R = Schedule<kSubLogN, kSubLogE>::R;
    constexpr bool TENSOR = SOURCE == kInverseFromTensor, KEYMAC = SOURCE == kInverseFromKeyMac;
    static_assert(SOURCE == kInverseFromSlab || TENSOR || KEYMAC, "the key switch's end stays with the tiled kernel");
    static_assert(!TENSOR || SCALED, "the fused tensor load belongs to dropExtendedBase (t N^-1)");
    constexpr int INPUT_STAGES = (TENSOR || KEYMAC) && kLazyTransformInput<MODE> ? kLazyInputStages : 0;
    extern __shared__ __attribute__((aligned(16))) uint64_t lds[];
    const uint32_t tid = threadIdx.x;
    uint32_t record, within;
    size_t rows[1];
    if constexpr (TENSOR || KEYMAC) {
        constexpr uint32_t REPLICAS = TENSOR ? 3 : 2;  // records (item, c) of one item read the same source rows
        uint32_t set, c, group;
        locate_replica(blockIdx.x, gridDim.x / REPLICAS, REPLICAS, set, c);
        locate(map, set, group, within);
        record = (map.record_base + group) * REPLICAS + c;
        rows[0] = size_t(record) * map.record_rows + map.band_offset + within;
    } else {
        locate_rows<1>(map, blockIdx.x, rows, record, within);
    }
    const uint32_t mi = map.mod_base + within;
    const DeviceModulus mod = ctx.moduli[mi];
    const Twiddles<MODE> cross(ctx, true, mi, LOGD);
    const Twiddles<MODE> tail(ctx, true, mi, LOGD, (1u << LOGD) - (1u << kSubLogN), kInterleavedLaneMajor<MODE>);
    const BufferResource row = make_resource(slab + (rows[0] << LOGD), 8u << LOGD);
    // (before the row's loads where the registers allow it: the plain slab on the shift-folded products -- a fused load needs
    // them for its operands, a limb-wise twiddle is 6 registers)
    constexpr bool EARLY_HEAD = SOURCE == kInverseFromSlab && (MODE == kModeFoldLazy || kCrossEarlySplit);
    TwiddleWords cross_head[kCrossAheadInverse<MODE>];
    if constexpr (EARLY_HEAD) inverse_cross_head<LOGS, MODE>(cross_head, tid, cross);
    uint64_t v[ROWS][E];
    if constexpr (TENSOR) {
        const size_t item = record / 3;
        const uint32_t c = static_cast<uint32_t>(record - item * 3);  // wave-uniform
        const size_t poly_words = static_cast<size_t>(map.record_rows) << LOGD;
        const uint64_t* const source = source_spec.first + item * 4 * poly_words + (static_cast<size_t>(map.band_offset + within) << LOGD);
        const uint32_t lane_words = lane_part<kSubLogN, kSubLogE, 0, R>(tid) << LOGS;
        auto reduce = [&](const ProductSum& sum) {
            if constexpr (kLazyTransformInput<MODE>) return reduce_product_sum_bounded_lazy(sum, mod);
            else return mod.wide_shift != 0 ? reduce_product_sum_bounded(sum, mod) : reduce_product_sum(sum, mod);  // wave-uniform
        };
#pragma unroll
        for (int r = 0; r < E; ++r) {
#pragma unroll
            for (int h = 0; h < ROWS; h += 2) {
                const size_t at = (register_part<kSubLogN, kSubLogE, 0, R>(r) << LOGS) + lane_words + h;
                if (c != 1) {
                    const U64x2 a = *reinterpret_cast<const U64x2*>(source + (c == 0 ? 0 : 1) * poly_words + at);
                    const U64x2 b = *reinterpret_cast<const U64x2*>(source + (c == 0 ? 2 : 3) * poly_words + at);
                    if constexpr (kLazyTransformInput<MODE>) {
                        v[h][r] = reduce_product_sum_bounded_lazy(product_sum_first(a.x, b.x), mod);
                        v[h + 1][r] = reduce_product_sum_bounded_lazy(product_sum_first(a.y, b.y), mod);
                    } else {
                        v[h][r] = barrett_mul(a.x, b.x, mod.p, mod.product_factor, static_cast<int>(mod.product_shift));
                        v[h + 1][r] = barrett_mul(a.y, b.y, mod.p, mod.product_factor, static_cast<int>(mod.product_shift));
                    }
                } else {
                    const U64x2 a0 = *reinterpret_cast<const U64x2*>(source + at);
                    const U64x2 a1 = *reinterpret_cast<const U64x2*>(source + poly_words + at);
                    const U64x2 b0 = *reinterpret_cast<const U64x2*>(source + 2 * poly_words + at);
                    const U64x2 b1 = *reinterpret_cast<const U64x2*>(source + 3 * poly_words + at);
                    ProductSum cross0 = product_sum_first(a0.x, b1.x), cross1 = product_sum_first(a0.y, b1.y);
                    product_sum_add(cross0, a1.x, b0.x);
                    product_sum_add(cross1, a1.y, b0.y);
                    v[h][r] = reduce(cross0);
                    v[h + 1][r] = reduce(cross1);
                }
            }
        }
    } else if constexpr (KEYMAC) {
        const size_t poly = record >> 1, c = record & 1;  // record = poly * 2 + c
        const uint32_t L = source_spec.L, top_rows = source_spec.top_rows;
        const uint32_t band_row = map.band_offset + within;
        const uint32_t key_row = (band_row == L) ? top_rows - 1 : band_row;  // Bfv+Keys.swift:153
        const uint32_t lane_words = lane_part<kSubLogN, kSubLogE, 0, R>(tid) << LOGS;
        const bool bounded = kKeyMacBoundedReduce && mod.wide_shift != 0 && L <= 8 && uint64_t(L) * mod.p < (uint64_t(1) << 63);  // wave-uniform
#pragma unroll
        for (int r = 0; r < E; ++r) {
#pragma unroll
            for (int h = 0; h < ROWS; h += 2) {
                const uint32_t at = (register_part<kSubLogN, kSubLogE, 0, R>(r) << LOGS) + h;
                // the words of term j + 1 are requested before term j is accumulated (ntt_inverse_tiled)
                const uint64_t* const flat_spread = source_spec.first + ((poly * L * (L + 1) + band_row) << LOGD) + lane_words + at;
                const uint64_t* const flat_key = source_spec.second + ((c * top_rows + key_row) << LOGD) + lane_words + at;
                U64x2 xs = *reinterpret_cast<const U64x2*>(flat_spread);
                U64x2 ks = *reinterpret_cast<const U64x2*>(flat_key);
                ProductSum acc0 = product_sum_zero(), acc1 = product_sum_zero();
                for (uint32_t j = 0; j < L; ++j) {
                    const uint32_t ahead = j + 1 < L ? j + 1 : j;
                    const U64x2 xn = *reinterpret_cast<const U64x2*>(flat_spread + ((size_t(ahead) * (L + 1)) << LOGD));
                    const U64x2 kn = *reinterpret_cast<const U64x2*>(flat_key + ((size_t(ahead) * 2 * top_rows) << LOGD));
                    product_sum_add_one<is_split(MODE)>(acc0, xs.x, ks.x);
                    product_sum_add_one<is_split(MODE)>(acc1, xs.y, ks.y);
                    xs = xn;
                    ks = kn;
                }
                if (bounded) {
                    if constexpr (kLazyTransformInput<MODE>) {
                        v[h][r] = reduce_product_sum_bounded_lazy(acc0, mod);
                        v[h + 1][r] = reduce_product_sum_bounded_lazy(acc1, mod);
                    } else {
                        v[h][r] = reduce_product_sum_bounded(acc0, mod);
                        v[h + 1][r] = reduce_product_sum_bounded(acc1, mod);
                    }
                } else {
                    v[h][r] = reduce_product_sum(acc0, mod);
                    v[h + 1][r] = reduce_product_sum(acc1, mod);
                }
            }
        }
    } else {
        if constexpr (kInterleavedStaged) interleaved_low_words_staged<LOGS, false>(v, tid, row, lds);
        else interleaved_low_words<LOGS, false>(v, tid, row);
    }
    if constexpr (!EARLY_HEAD) inverse_cross_head<LOGS, MODE>(cross_head, tid, cross);
    inverse_cross_stages<LOGS, MODE, INPUT_STAGES>(v, tid, cross, mod.p, cross_head);
    TwiddleWords head[1];
    inverse_row_head<kSubLogN, kSubLogE, MODE, false, 1>(head, tail, tid);
    // (two sub-rows on the shift-folded products: every step re-derives the lane index, as the key-MAC transforms -- step_lane)
    constexpr bool LATE = ROWS == 2 && MODE == kModeFoldLazy;
    inverse_row<kSubLogN, kSubLogE, MODE, ROWS, SCALED, LOGS + INPUT_STAGES, LOGD, false, 1, LATE>(v, tid, tail, mod, lds, head);
    interleaved_top_words<LOGS, true>(v, LATE ? late_lane(tid) : tid, row);
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
                xo = canonicalize<kModeExact>(xo, p);
                yo = canonicalize<kModeExact>(yo, p);
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

// Row pairs: where the register file allows it (8 words per lane), a workgroup transforms the same band row of two
// consecutive records -- one modulus, every twiddle fetched once for both.
// Where the shift-folded products (ntt_common.hpp kModeFoldLazy) replace the limb-wise Shoup ones for launches whose moduli
// are all of the form 2^b - d (DeviceContext::shift_prefix): every kernel of the 8-words-per-lane shapes at N = 4096 / 8192 --
// plain slabs and the fused loads, both directions -- and the Q band of the row-fused ct x ct kernel (behz_kernels.hip); since
// round 6 the interleaved sub-rows of N = 16384 / 32768 too, where every modulus of the launch qualifies (at N = 16384 only the
// first two of the standard 55-bit primes do: with 4-register twiddles the cross stages keep three in flight and the inverse
// requests its first ones before the row's loads -- forward -4 %, inverse -3 % on two such moduli,
// profiles/r06p_fold_interleaved_eligible_moduli_ab.txt; round 5's "the same either way" was measured on a context that did
// not qualify).
template <int LOGN, int LOGT>
constexpr bool kFoldLazyForward = (LOGN == 12 && LOGT == 9) || (LOGN == 13 && LOGT == 10);
template <int LOGN, int LOGT, int SOURCE>
constexpr bool kFoldLazyInverse = (LOGN == 12 && LOGT == 9) || (LOGN == 13 && LOGT == 10);
constexpr bool kFoldLazyInterleaved = true;
// ... and the fold butterflies of the plus form (ntt_common.hpp kModeFoldPlus) for the BEHZ primes' rows of N = 16384 / 32768: the
// Bsk bands of ct x ct ran on the [0, 8p) butterflies there (0.38 / 0.27 of 8 TB/s forward / inverse at N = 16384 against the Q
// band's 0.54 / 0.35: profiles/r06q_fold_plus_interleaved.txt)
constexpr bool kFoldPlusInterleaved = true;
// every modulus of a launch is of the form 2^b - d (DeviceContext::shift_prefix)
inline bool fold_lazy_band(const DeviceContext& ctx, const RowMap& map) {
    return map.band_rows != 0 && map.mod_base + map.band_rows <= ctx.shift_prefix;
}

constexpr int kRowGroup = 2;
// ... unless the launch is small: below a few rows per workgroup slot (256 CUs x 2 workgroups of 64 registers) a pair only
// doubles the launch's latency -- the levels of a query's expansion and key switches on a few ciphertexts are chains of such
// launches.  Measured (profiles/r06w_small_launches.txt): one query's expansion to 320 ciphertexts 1.078 -> 0.898 ms,
// relinearize on 8 / 64 ciphertexts 67.7 -> 45.5 / 133 -> 114 us; from 8192 rows up the pairs win (the key MAC of 1024
// ciphertexts -1.3 %, the headline transform's 16384 rows -6 % without them).
constexpr size_t kUngroupedBelowRows = 4096;
template <int LOGN, int LOGT>
constexpr int kRowsPerWorkgroup = (LOGN - LOGT <= 3 && Schedule<LOGN, LOGN - LOGT>::P >= 2) ? kRowGroup : 1;

// Rows per workgroup of the fused-load inverse kernels.  The tensor load gains nothing from a second row (the three
// workgroups of a replica set already gather the same twiddles side by side; profiles/r02i_c3_fused_row_groups.txt,
// r03j_c3_fused_row_groups.txt).  The key MAC does since its load is a rolled loop over the terms with the next term
// requested ahead -- two rows then cost 20 bytes of scratch per lane instead of the 104 of the unrolled
// carry-counting form that lost 6 % in round 2: relinearize 638 -> 659 k/s.  (The fused FORWARD loads -- spread, lift --
// are plain row loads and run two rows per workgroup: relinearize +7 %, convertToEvalFormat +21 %.)
constexpr int kTensorRowGroup = 1, kKeyMacRowGroup = 2;
template <int LOGN, int LOGT>
constexpr int kTensorRows = kRowsPerWorkgroup<LOGN, LOGT> > 1 ? kTensorRowGroup : 1;
template <int LOGN, int LOGT>
constexpr int kKeyMacRows = kRowsPerWorkgroup<LOGN, LOGT> > 1 ? kKeyMacRowGroup : 1;

// Interleaved launches (ntt_forward_interleaved / ntt_inverse_interleaved): one workgroup per row of 2^(13 + LOGS) words.
// N = 16384 takes them for plain slabs instead of the streamed rows (profiles/r03p_ntt_interleaved.txt); N = 32768 has no
// other tiled kernel.  Round 5: the fused loads at N = 16384 (key-switching decomposition, plaintext lift, the Q band of the
// lifted records into the forward transform; tensor product and key inner product into the inverse one) take the same form
// -- two workgroups per CU at 64 registers instead of the 16-words-per-lane tile at one (kInterleavedFusedLoads).
constexpr bool kInterleaved16384 = true;
constexpr bool kInterleavedFusedLoads = true;
// The largest degree whose pipelines take the fused loads (2^15: as four interleaved sub-rows, round 5; 14: N = 32768 runs its
// pipelines unfused over the plain interleaved transforms, rounds 3-4)
constexpr uint32_t kMaxFusedLoadLogDegree = 15;
// ... of which, at N = 32768, the key switch's two are decided on their own measurements (128 pairs, L = 6,
// profiles/r05p_fused_loads_32768_ab.txt): the decomposition into the forward transform gains (relinearize 50.5 -> 62.4 k/s),
// the key inner product into the inverse one -- a rolled loop over the terms in a 128-register workgroup that is alone on its
// CU -- loses (40.5 k/s with both, 34.9 k/s with it alone) and stays a kernel of its own in front of the plain transforms
constexpr bool kFusedSpreadAt32768 = true, kFusedKeyMacAt32768 = false;
// N = 16384: the key inner product as a kernel of its own in front of the plain interleaved inverse transforms, then the finish
// kernel: 163.3 k relinearize/s at L = 6 against 157.4 k with the q_ks row's MAC fused into the interleaved inverse and the other
// rows on the 16-words-per-lane tile with the fused end (profiles/r05s_keymac_16384_ab.txt) -- a fused load pays where its kernel
// has a second workgroup on the CU to run under it (N = 4096 / 8192), not where the workgroup is alone
constexpr bool kFusedKeyMacAt16384 = false;
// (the key MAC's rows r < L stay on the 16-words-per-lane tile, whose store carries the key switch's end: interleaved sub-rows
// plus the separate finish kernel measured 134.9 k relinearize/s at N = 16384, L = 6 against 157.8 k this way and 143.7 k with
// every fused load on the tile -- profiles/r05f_fused_loads_16384_ab.txt; the q_ks row and the other fused loads are interleaved)
constexpr bool kInterleavedKeyMacAt16384 = false;
// (four sub-rows -- N = 32768, one workgroup per CU at 128 registers -- go through two tiles side by side like the row groups
// of behz_kernels.hip: ntt_rows.hpp kGroupTiles)
template <int LOGS>
constexpr size_t kInterleavedLdsBytes = kGroupTiles<(1 << LOGS)> * lds_words(1u << kSubLogN) * sizeof(uint64_t);
template <int LOGS, int SPREAD>
hipError_t launch_interleaved_forward(int mode, uint64_t* slab, const DeviceContext& ctx, const RowMap& map, size_t rows,
                                      const SpreadSource& spread, hipStream_t stream) {
    auto kernel = mode == kModeSplit    ? ntt_forward_interleaved<LOGS, kModeSplit, SPREAD>
                  : mode == kModeApprox ? ntt_forward_interleaved<LOGS, kModeApprox, SPREAD>
                                        : ntt_forward_interleaved<LOGS, kModeExact, SPREAD>;
    if constexpr (kFoldLazyInterleaved) {
        if (mode == kModeSplit && fold_lazy_band(ctx, map)) kernel = ntt_forward_interleaved<LOGS, kModeFoldLazy, SPREAD>;
    }
    if constexpr (kFoldPlusInterleaved && SPREAD == kSourceSlab) {
        // the BEHZ primes 2^60 + e (the Bsk band of a lifted record, transformed in place) on the fold butterflies, as at N = 8192
        if (mode == kModeApprox && ctx.forward_split_pairs != nullptr && fold_mode(ctx, map.mod_base, map.band_rows) == kModeFoldPlus)
            kernel = ntt_forward_interleaved<LOGS, kModeFoldPlus, SPREAD>;
    }
    if (hipError_t e = allow_dynamic_lds(kernel, kInterleavedLdsBytes<LOGS>); e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(rows)), dim3(1u << kSubLogT), kInterleavedLdsBytes<LOGS>, stream, slab, ctx, map,
                       spread);
    return hipGetLastError();
}
// the limb-wise butterflies of the interleaved inverse: the signed difference (ntt_common.hpp kModeSplitSigned; N = 16384 -3.9 %,
// profiles/r04t_inverse_forms_ab.txt)
constexpr int kInterleavedInverseSplit = kModeSplitSigned;
template <int LOGS, int SOURCE>
hipError_t launch_interleaved_inverse(int mode, uint64_t* slab, const DeviceContext& ctx, const RowMap& map, size_t rows,
                                      const InverseSource& source_spec, hipStream_t stream) {
    using Kernel = void (*)(uint64_t*, const DeviceContext, const RowMap, const InverseSource);
    Kernel kernel;
    if constexpr (SOURCE == kInverseFromTensor) {
        if (ctx.scaled_inverse_degree == 0) return hipErrorInvalidValue;  // the fused tensor load belongs to dropExtendedBase
        kernel = mode == kModeSplit    ? ntt_inverse_interleaved<LOGS, kInterleavedInverseSplit, true, SOURCE>
                 : mode == kModeApprox ? ntt_inverse_interleaved<LOGS, kModeApprox, true, SOURCE>
                                       : ntt_inverse_interleaved<LOGS, kModeExact, true, SOURCE>;
    } else if constexpr (SOURCE == kInverseFromKeyMac) {
        if (ctx.scaled_inverse_degree != 0) return hipErrorInvalidValue;  // the key-switching contexts are never scaled
        kernel = mode == kModeSplit    ? ntt_inverse_interleaved<LOGS, kInterleavedInverseSplit, false, SOURCE>
                 : mode == kModeApprox ? ntt_inverse_interleaved<LOGS, kModeApprox, false, SOURCE>
                                       : ntt_inverse_interleaved<LOGS, kModeExact, false, SOURCE>;
    } else if (ctx.scaled_inverse_degree != 0) {
        kernel = mode == kModeSplit    ? ntt_inverse_interleaved<LOGS, kInterleavedInverseSplit, true, SOURCE>
                 : mode == kModeApprox ? ntt_inverse_interleaved<LOGS, kModeApprox, true, SOURCE>
                                       : ntt_inverse_interleaved<LOGS, kModeExact, true, SOURCE>;
    } else {
        kernel = mode == kModeSplit    ? ntt_inverse_interleaved<LOGS, kInterleavedInverseSplit, false, SOURCE>
                 : mode == kModeApprox ? ntt_inverse_interleaved<LOGS, kModeApprox, false, SOURCE>
                                       : ntt_inverse_interleaved<LOGS, kModeExact, false, SOURCE>;
    }
    if constexpr (kFoldLazyInterleaved) {
        if (mode == kModeSplit && fold_lazy_band(ctx, map)) {
            constexpr bool ALWAYS_SCALED = SOURCE == kInverseFromTensor, NEVER_SCALED = SOURCE == kInverseFromKeyMac;
            if constexpr (ALWAYS_SCALED) kernel = ntt_inverse_interleaved<LOGS, kModeFoldLazy, true, SOURCE>;
            else if constexpr (NEVER_SCALED) kernel = ntt_inverse_interleaved<LOGS, kModeFoldLazy, false, SOURCE>;
            else kernel = ctx.scaled_inverse_degree != 0 ? ntt_inverse_interleaved<LOGS, kModeFoldLazy, true, SOURCE>
                                                         : ntt_inverse_interleaved<LOGS, kModeFoldLazy, false, SOURCE>;
        }
    }
    if constexpr (kFoldPlusInterleaved && (SOURCE == kInverseFromSlab || SOURCE == kInverseFromTensor)) {
        if (mode == kModeApprox && ctx.forward_split_pairs != nullptr && fold_mode(ctx, map.mod_base, map.band_rows) == kModeFoldPlus) {
            if constexpr (SOURCE == kInverseFromTensor) kernel = ntt_inverse_interleaved<LOGS, kModeFoldPlus, true, SOURCE>;
            else kernel = ctx.scaled_inverse_degree != 0 ? ntt_inverse_interleaved<LOGS, kModeFoldPlus, true, SOURCE>
                                                         : ntt_inverse_interleaved<LOGS, kModeFoldPlus, false, SOURCE>;
        }
    }
    if (hipError_t e = allow_dynamic_lds(kernel, kInterleavedLdsBytes<LOGS>); e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(rows)), dim3(1u << kSubLogT), kInterleavedLdsBytes<LOGS>, stream, slab, ctx, map,
                       source_spec);
    return hipGetLastError();
}
template <int LOGS>
hipError_t launch_interleaved(bool inverse, int mode, uint64_t* slab, const DeviceContext& ctx, const RowMap& map, size_t rows,
                              hipStream_t stream) {
    if (!inverse) return launch_interleaved_forward<LOGS, kSourceSlab>(mode, slab, ctx, map, rows, SpreadSource{nullptr, 0, 0, 0, 0}, stream);
    return launch_interleaved_inverse<LOGS, kInverseFromSlab>(mode, slab, ctx, map, rows,
                                                              InverseSource{nullptr, nullptr, 0, 0, nullptr, 0, nullptr, 0}, stream);
}

template <int LOGN, int LOGT, int SPREAD, int ROWS>
hipError_t launch_forward_kernel(int mode, uint64_t* slab, const DeviceContext& ctx, const RowMap& map, size_t workgroups,
                                 const SpreadSource& spread, hipStream_t stream) {
    constexpr int LOGE = LOGN - LOGT;
    constexpr size_t lds_bytes = (Schedule<LOGN, LOGE>::P > 1) ? lds_words(1u << LOGN) * sizeof(uint64_t) : 0;
    auto kernel = mode == kModeSplit    ? ntt_forward_tiled<LOGN, LOGT, kModeSplit, SPREAD, ROWS>
                  : mode == kModeApprox ? ntt_forward_tiled<LOGN, LOGT, kModeApprox, SPREAD, ROWS>
                                        : ntt_forward_tiled<LOGN, LOGT, kModeExact, SPREAD, ROWS>;
    if constexpr (kFoldLazyForward<LOGN, LOGT>) {
        // every modulus of the launch is just below a power of two: products folded by a shift, no quotient, no factors
        if (mode == kModeSplit && fold_lazy_band(ctx, map)) kernel = ntt_forward_tiled<LOGN, LOGT, kModeFoldLazy, SPREAD, ROWS>;
    }
    if constexpr (kFoldShape<LOGN, LOGT> && (SPREAD == kSourceSlab || SPREAD == kSourceSpread)) {
        if (mode == kModeApprox && ctx.forward_split_pairs != nullptr) {
            const int fold = fold_mode(ctx, map.mod_base, map.band_rows);
            if (fold == kModeFoldMinus) kernel = ntt_forward_tiled<LOGN, LOGT, kModeFoldMinus, SPREAD, ROWS>;
            if (fold == kModeFoldPlus) kernel = ntt_forward_tiled<LOGN, LOGT, kModeFoldPlus, SPREAD, ROWS>;
        }
    }
    if (hipError_t e = allow_dynamic_lds(kernel, lds_bytes); e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(workgroups)), dim3(1u << LOGT), lds_bytes, stream, slab, ctx, map,
                       spread);
    return hipGetLastError();
}

template <int LOGN, int LOGT, int SPREAD>
hipError_t launch_forward_tiled(int mode, uint64_t* slab, const DeviceContext& ctx, uint32_t mod_base,
                                uint32_t mod_period, size_t rows, const SpreadSource& spread, hipStream_t stream,
                                uint32_t row_period = 0, uint32_t row_offset = 0) {
    constexpr bool INTERLEAVED = LOGN == 14 && LOGT == 10 && SPREAD != kSourceSlab && kInterleavedFusedLoads;
    if constexpr (INTERLEAVED && SPREAD != kSourceSpread) {
        // as two interleaved sub-rows (the tile kernel of this source is not instantiated)
        return launch_interleaved_forward<1, SPREAD>(mode, slab, ctx, make_row_map(mod_base, mod_period, row_period, row_offset), rows,
                                                     spread, stream);
    } else {
        if constexpr (INTERLEAVED) {
            // (the automorphism of a Galois key switch permutes whole rows through the tile: the tiled kernel below)
            if (spread.galois_inverse == 0)
                return launch_interleaved_forward<1, SPREAD>(mode, slab, ctx, make_row_map(mod_base, mod_period, row_period, row_offset),
                                                             rows, spread, stream);
        }
        size_t paired_records = 0;  // records covered by the launch of row groups
        constexpr int GROUP = kRowsPerWorkgroup<LOGN, LOGT>;
        if constexpr (GROUP > 1) {
            paired_records = rows <= kUngroupedBelowRows ? 0 : (rows / mod_period) / GROUP * GROUP;
            if (paired_records != 0) {
                hipError_t e = launch_forward_kernel<LOGN, LOGT, SPREAD, GROUP>(
                    mode, slab, ctx, make_row_map(mod_base, mod_period, row_period, row_offset),
                    paired_records / GROUP * mod_period, spread, stream);
                if (e != hipSuccess) return e;
            }
        }
        const size_t rest = rows - paired_records * mod_period;
        if (rest == 0) return hipSuccess;
        return launch_forward_kernel<LOGN, LOGT, SPREAD, 1>(
            mode, slab, ctx, make_row_map(mod_base, mod_period, row_period, row_offset, static_cast<uint32_t>(paired_records)),
            rest, spread, stream);
    }
}

template <int LOGN, int LOGT, int SOURCE, int ROWS>
hipError_t launch_inverse_kernel(int mode, uint64_t* slab, const DeviceContext& ctx, const RowMap& map, size_t workgroups,
                                 const InverseSource& source_spec, hipStream_t stream) {
    constexpr int LOGE = LOGN - LOGT;
    constexpr size_t lds_bytes = (Schedule<LOGN, LOGE>::P > 1) ? lds_words(1u << LOGN) * sizeof(uint64_t) : 0;
    // (the limb-wise inverse in its signed form wherever that measured faster: ntt_common.hpp kModeSplitSigned)
    constexpr int SPLIT = kSignedInverse<LOGN, SOURCE> ? kModeSplitSigned : kModeSplit;
    auto kernel = mode == kModeSplit    ? ntt_inverse_tiled<LOGN, LOGT, SPLIT, SOURCE, ROWS>
                  : mode == kModeApprox ? ntt_inverse_tiled<LOGN, LOGT, kModeApprox, SOURCE, ROWS>
                                        : ntt_inverse_tiled<LOGN, LOGT, kModeExact, SOURCE, ROWS>;
    if constexpr (kFoldLazyInverse<LOGN, LOGT, SOURCE>) {
        if (mode == kModeSplit && fold_lazy_band(ctx, map)) kernel = ntt_inverse_tiled<LOGN, LOGT, kModeFoldLazy, SOURCE, ROWS>;
    }
    if constexpr (kFoldShape<LOGN, LOGT>) {
        if (mode == kModeApprox && ctx.forward_split_pairs != nullptr) {
            const int fold = fold_mode(ctx, map.mod_base, map.band_rows);
            if (fold == kModeFoldMinus) kernel = ntt_inverse_tiled<LOGN, LOGT, kModeFoldMinus, SOURCE, ROWS>;
            if (fold == kModeFoldPlus) kernel = ntt_inverse_tiled<LOGN, LOGT, kModeFoldPlus, SOURCE, ROWS>;
        }
    }
    if (hipError_t e = allow_dynamic_lds(kernel, lds_bytes); e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(static_cast<unsigned>(workgroups)), dim3(1u << LOGT), lds_bytes, stream, slab, ctx, map,
                       source_spec);
    return hipGetLastError();
}

// One fused inverse source over records (item, c): groups of consecutive ITEMS share a workgroup (same c, same band row); the
// odd items at the end go one per workgroup.  record_base counts items for these kernels.
template <int LOGN, int LOGT, int SOURCE>
hipError_t launch_fused_inverse(int mode, uint64_t* slab, const DeviceContext& ctx, uint32_t mod_base, uint32_t mod_period,
                                size_t rows, uint32_t row_period, uint32_t row_offset, const InverseSource& source_spec,
                                hipStream_t stream) {
    constexpr bool TENSOR = SOURCE == kInverseFromTensor;
    constexpr size_t REPLICAS = TENSOR ? 3 : 2;
    constexpr int GROUP = TENSOR ? kTensorRows<LOGN, LOGT> : kKeyMacRows<LOGN, LOGT>;
    const size_t items = rows / mod_period / REPLICAS;
    const size_t grouped = (GROUP > 1 && rows > kUngroupedBelowRows) ? items / GROUP * GROUP : 0;
    auto map_from = [&](size_t first_item) {
        return make_row_map(mod_base, mod_period, row_period, row_offset, static_cast<uint32_t>(first_item));
    };
    if constexpr (GROUP > 1) {
        if (grouped != 0) {
            hipError_t e = launch_inverse_kernel<LOGN, LOGT, SOURCE, GROUP>(mode, slab, ctx, map_from(0),
                                                                            grouped / GROUP * REPLICAS * mod_period, source_spec, stream);
            if (e != hipSuccess) return e;
        }
    }
    if (items == grouped) return hipSuccess;
    return launch_inverse_kernel<LOGN, LOGT, SOURCE, 1>(mode, slab, ctx, map_from(grouped), (items - grouped) * REPLICAS * mod_period,
                                                        source_spec, stream);
}

template <int LOGN, int LOGT>
hipError_t launch_tiled(bool inverse, int mode, uint64_t* slab, const DeviceContext& ctx, uint32_t mod_base,
                        uint32_t mod_period, size_t rows, hipStream_t stream, uint32_t row_period = 0,
                        uint32_t row_offset = 0, int source = kInverseFromSlab,
                        const InverseSource& source_spec = InverseSource{nullptr, nullptr, 0, 0, nullptr, 0, nullptr, 0}) {
    if constexpr (LOGN == 14 && LOGT == 10 && kInterleaved16384) {
        if (source == kInverseFromSlab || !inverse)
            return launch_interleaved<1>(inverse, mode, slab, ctx, make_row_map(mod_base, mod_period, row_period, row_offset),
                                         rows, stream);
    }
    if (!inverse) {
        return launch_forward_tiled<LOGN, LOGT, kSourceSlab>(mode, slab, ctx, mod_base, mod_period, rows,
                                                             SpreadSource{nullptr, 0, 0, 0, 0}, stream, row_period,
                                                             row_offset);
    }
    constexpr int LOGE = LOGN - LOGT;
    if (source != kInverseFromSlab && (row_period == 0 || Schedule<LOGN, LOGE>::P == 1)) return hipErrorInvalidValue;
    const RowMap map = make_row_map(mod_base, mod_period, row_period, row_offset);
    // The fused loads and the scaled contexts of dropExtendedBase exist for the PRODUCTION shape of each degree only (8 words
    // per lane; N = 16384: 16): the 16- and 32-words-per-lane shapes are test variants of the plain transform
    // (kNttVariantWide, public contexts only) -- instantiating every source for them was a hundred kernels
    // nobody could launch, the exact-mode ones among them with their register tile in scratch.
    constexpr bool FUSED_SHAPE = (LOGN == 12 && LOGT == 9) || (LOGN == 13 && LOGT == 10) || (LOGN == 14 && LOGT == 10);
    if constexpr (!FUSED_SHAPE) {
        if (source != kInverseFromSlab || ctx.scaled_inverse_degree != 0) return hipErrorInvalidValue;
    } else {
        if (ctx.scaled_inverse_degree != 0) {
            if (is_key_mac(source)) return hipErrorInvalidValue;  // the key-switching contexts are never scaled
            if (source == kInverseFromSlab)
                return launch_inverse_kernel<LOGN, LOGT, kInverseFromSlabScaled, 1>(mode, slab, ctx, map, rows, source_spec, stream);
        } else if (source == kInverseFromTensor) {
            return hipErrorInvalidValue;  // the fused tensor load belongs to dropExtendedBase (t N^-1)
        }
    }
    constexpr int GROUP = kRowsPerWorkgroup<LOGN, LOGT>;
    if constexpr (LOGN == 14 && LOGT == 10 && kInterleavedFusedLoads) {
        // tensor product / key inner product into the interleaved inverse: one record (item, c) per workgroup; the key switch's
        // fused end (kInverseFromKeyMacFinish) is not offered at this degree (ntt_key_mac_finish_supported)
        if (source == kInverseFromTensor)
            return launch_interleaved_inverse<1, kInverseFromTensor>(mode, slab, ctx, map, rows, source_spec, stream);
        if constexpr (kFusedKeyMacAt16384) {
            if (source == kInverseFromKeyMac)
                return launch_interleaved_inverse<1, kInverseFromKeyMac>(mode, slab, ctx, map, rows, source_spec, stream);
        }
    }
    if constexpr (FUSED_SHAPE) {
        // (N = 16384 with the interleaved fused loads: only the key switch's fused end is left to the tile -- the others are
        // not instantiated for it)
        constexpr bool ONLY_FINISH = LOGN == 14 && kInterleavedFusedLoads;
        if constexpr (LOGN != 14 || kFusedKeyMacAt16384) {
            if (source == kInverseFromKeyMacFinish)
                return launch_fused_inverse<LOGN, LOGT, kInverseFromKeyMacFinish>(mode, slab, ctx, mod_base, mod_period, rows,
                                                                                  row_period, row_offset, source_spec, stream);
        }
        if constexpr (!ONLY_FINISH) {
            if (source == kInverseFromTensor)
                return launch_fused_inverse<LOGN, LOGT, kInverseFromTensor>(mode, slab, ctx, mod_base, mod_period, rows, row_period,
                                                                            row_offset, source_spec, stream);
            if (source == kInverseFromKeyMac)
                return launch_fused_inverse<LOGN, LOGT, kInverseFromKeyMac>(mode, slab, ctx, mod_base, mod_period, rows, row_period,
                                                                            row_offset, source_spec, stream);
        } else if (source != kInverseFromSlab) {
            return hipErrorInvalidValue;
        }
    }
    size_t paired_records = 0;
    if constexpr (GROUP > 1) {
        paired_records = rows <= kUngroupedBelowRows ? 0 : (rows / mod_period) / GROUP * GROUP;
        if (paired_records != 0) {
            hipError_t e = launch_inverse_kernel<LOGN, LOGT, kInverseFromSlab, GROUP>(
                mode, slab, ctx, map, paired_records / GROUP * mod_period, source_spec, stream);
            if (e != hipSuccess) return e;
        }
    }
    const size_t rest = rows - paired_records * mod_period;
    if (rest == 0) return hipSuccess;
    return launch_inverse_kernel<LOGN, LOGT, kInverseFromSlab, 1>(
        mode, slab, ctx, make_row_map(mod_base, mod_period, row_period, row_offset, static_cast<uint32_t>(paired_records)), rest,
        source_spec, stream);
}

}  // namespace

hipError_t launch_ntt_spread(const uint64_t* source, size_t poly_stride, uint32_t source_moduli, size_t polys,
                             uint64_t* spread, const DeviceContext& ks_ctx, uint32_t galois_inverse, hipStream_t stream) {
    const uint32_t period = source_moduli + 1;
    const size_t rows = polys * source_moduli * period;
    if (rows == 0) return hipSuccess;
    if (rows > (size_t(1) << 30) || ks_ctx.moduli_count < period) return hipErrorInvalidValue;
    const SpreadSource src{source, poly_stride, source_moduli, 0, galois_inverse};
    // One launch per run of key-switching moduli of one butterfly class (band_runs: e.g. 29 | 60, 60 for the reference's
    // n_8192_logq_29_60_60, whose 60-bit rows then take the fold butterflies); the usual contexts -- every modulus in
    // [2^40, 2^55) -- are one run, and a batch of one workgroup generation takes one launch either way.
    auto launch = [&](int mode, uint32_t base, uint32_t band) {
        const size_t band_total = polys * source_moduli * band;
        const uint32_t row_period = band == period ? 0 : period;
        switch (ks_ctx.log_degree) {
            case 12: return launch_forward_tiled<12, 9, kSourceSpread>(mode, spread, ks_ctx, base, band, band_total, src, stream, row_period, base);
            case 13: return launch_forward_tiled<13, 10, kSourceSpread>(mode, spread, ks_ctx, base, band, band_total, src, stream, row_period, base);
            case 14: return launch_forward_tiled<14, 10, kSourceSpread>(mode, spread, ks_ctx, base, band, band_total, src, stream, row_period, base);
            case 15:  // four interleaved sub-rows (the automorphism of a Galois key switch: the caller's unfused path)
                if (galois_inverse != 0 || kMaxFusedLoadLogDegree < 15 || !kFusedSpreadAt32768) return hipErrorNotSupported;
                return launch_interleaved_forward<2, kSourceSpread>(mode, spread, ks_ctx, make_row_map(base, band, row_period, base),
                                                                    band_total, src, stream);
            default: return hipErrorNotSupported;  // caller falls back to spread kernel + launch_ntt
        }
    };
    BandRun runs[kMaxBandRuns];
    const int count = (ks_ctx.log_degree == 12 || ks_ctx.log_degree == 13) && rows > kOneGeneration ? band_runs(ks_ctx, period, runs) : 0;
    if (count <= 1) return launch(production_mode(ks_ctx), 0, period);
    for (int k = 0; k < count; ++k)
        if (hipError_t e = launch(runs[k].mode, runs[k].base, runs[k].rows); e != hipSuccess) return e;
    return hipSuccess;
}

hipError_t launch_ntt_lift(const uint64_t* plaintexts, uint64_t plaintext_modulus, size_t polys, uint64_t* out,
                           const DeviceContext& ctx, hipStream_t stream) {
    const uint32_t period = ctx.moduli_count;
    const size_t rows = polys * period;
    if (rows == 0) return hipSuccess;
    if (rows > (size_t(1) << 30)) return hipErrorInvalidValue;
    const SpreadSource src{plaintexts, size_t(1) << ctx.log_degree, 1, plaintext_modulus, 0};
    const int mode = production_mode(ctx);
    switch (ctx.log_degree) {
        case 12: return launch_forward_tiled<12, 9, kSourceLift>(mode, out, ctx, 0, period, rows, src, stream);
        case 13: return launch_forward_tiled<13, 10, kSourceLift>(mode, out, ctx, 0, period, rows, src, stream);
        case 14: return launch_forward_tiled<14, 10, kSourceLift>(mode, out, ctx, 0, period, rows, src, stream);
        case 15:
            if (kMaxFusedLoadLogDegree < 15) return hipErrorNotSupported;
            return launch_interleaved_forward<2, kSourceLift>(mode, out, ctx, make_row_map(0, period, 0, 0), rows, src, stream);
        default: return hipErrorNotSupported;  // caller falls back to the lift kernel + launch_ntt
    }
}

const char* ntt_variant_name(uint32_t log_degree) {
    switch (log_degree) {
        case 12: return "ntt_tiled<4096, 512 lanes x 8 words>";
        case 13: return "ntt_tiled<8192, 1024 lanes x 8 words>";
        case 14: return "ntt_interleaved<16384 = 2 x 8192, 1024 lanes x 8 words x 2 sub-rows>";
        case 15: return "ntt_interleaved<32768 = 4 x 8192, 1024 lanes x 8 words x 4 sub-rows>";
        default: return "generic radix-2";
    }
}

hipError_t launch_ntt_band(bool inverse, uint64_t* slab, const DeviceContext& ctx, uint32_t mod_base, uint32_t band_rows,
                           uint32_t record_rows, uint32_t band_offset, size_t records, int mode, hipStream_t stream,
                           int source = kInverseFromSlab,
                           const InverseSource& source_spec = InverseSource{nullptr, nullptr, 0, 0, nullptr, 0, nullptr, 0}) {
    const size_t rows = records * band_rows;
    if (rows == 0) return hipSuccess;
    if (rows > (size_t(1) << 30)) return hipErrorInvalidValue;
    switch (ctx.log_degree) {
        case 12: return launch_tiled<12, 9>(inverse, mode, slab, ctx, mod_base, band_rows, rows, stream, record_rows, band_offset, source, source_spec);
        case 13: return launch_tiled<13, 10>(inverse, mode, slab, ctx, mod_base, band_rows, rows, stream, record_rows, band_offset, source, source_spec);
        case 14: return launch_tiled<14, 10>(inverse, mode, slab, ctx, mod_base, band_rows, rows, stream, record_rows, band_offset, source, source_spec);
        case 15: {
            const RowMap map = make_row_map(mod_base, band_rows, record_rows, band_offset);
            if (kMaxFusedLoadLogDegree < 15 && inverse && source != kInverseFromSlab) return hipErrorNotSupported;
            if (inverse && source == kInverseFromTensor) return launch_interleaved_inverse<2, kInverseFromTensor>(mode, slab, ctx, map, rows, source_spec, stream);
            if constexpr (kFusedKeyMacAt32768) {  // (not instantiated otherwise)
                if (inverse && source == kInverseFromKeyMac) return launch_interleaved_inverse<2, kInverseFromKeyMac>(mode, slab, ctx, map, rows, source_spec, stream);
            }
            if (inverse && source != kInverseFromSlab) return hipErrorNotSupported;  // the key switch's fused end: not at this degree
            return launch_interleaved<2>(inverse, mode, slab, ctx, map, rows, stream);
        }
        default: return hipErrorNotSupported;
    }
}

// [Q, Bsk] records (BEHZ): the first `headroom_prefix` moduli (the ciphertext moduli, when they are the usual <= 55-bit
// primes) take the fold-free split butterflies, the 61-bit Bsk primes the [0, 8p) ones -- two launches over row bands of
// the same slab.  Falls back to one launch when the context has no such prefix or no tiled kernel.
hipError_t launch_ntt_mixed(bool inverse, uint64_t* slab, const DeviceContext& ctx, uint32_t record_rows, size_t records,
                            hipStream_t stream) {
    const bool tiled = ctx.log_degree >= 12 && ctx.log_degree <= kMaxFusedLoadLogDegree;
    BandRun runs[kMaxBandRuns];
    const int count = tiled ? band_runs(ctx, record_rows, runs) : 0;
    if (count <= 1 || records * record_rows > (size_t(1) << 30) || records * record_rows <= kOneGeneration)
        return launch_ntt(inverse, slab, ctx, 0, record_rows, records * record_rows, stream);
    for (int k = 0; k < count; ++k) {
        hipError_t e = launch_ntt_band(inverse, slab, ctx, runs[k].base, runs[k].rows, record_rows, runs[k].base, records,
                                       runs[k].mode, stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// The rows [first, first + count) of every record only (row r uses modulus r), one launch per run of moduli of one butterfly
// class inside that range: e.g. the Bsk rows of lifted records whose Q rows already hold their transform.
hipError_t launch_ntt_record_band(bool inverse, uint64_t* slab, const DeviceContext& ctx, uint32_t record_rows, uint32_t first,
                                  uint32_t count, size_t records, hipStream_t stream) {
    if (records == 0 || count == 0) return hipSuccess;
    const bool tiled = ctx.log_degree >= 12 && ctx.log_degree <= kMaxFusedLoadLogDegree;
    if (!tiled || first + count > record_rows || records * record_rows > (size_t(1) << 30)) return hipErrorNotSupported;
    BandRun runs[kMaxBandRuns];
    const int run_count = band_runs(ctx, record_rows, runs);
    if (run_count <= 1) return launch_ntt_band(inverse, slab, ctx, first, count, record_rows, first, records, production_mode(ctx), stream);
    for (int k = 0; k < run_count; ++k) {
        const uint32_t begin = runs[k].base > first ? runs[k].base : first;
        const uint32_t end = runs[k].base + runs[k].rows < first + count ? runs[k].base + runs[k].rows : first + count;
        if (begin >= end) continue;
        hipError_t e = launch_ntt_band(inverse, slab, ctx, begin, end - begin, record_rows, begin, records, runs[k].mode, stream);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// Forward NTT of lifted [Q, Bsk] records whose Q rows are still where the ciphertexts lie (the lift was told not to copy
// them, rns_kernels launch_lift_q_to_qbsk_strided): the Q band is read from the source polynomials (kSourceRows) and
// written into the slab, the Bsk band is transformed in place.  hipErrorNotSupported when the context has no fold-free
// prefix of exactly source_moduli rows or no tiled kernel -- the caller then lifts with the copy and runs
// launch_ntt_mixed.
bool ntt_lifted_forward_supported(const DeviceContext& ctx, uint32_t record_rows, uint32_t source_moduli, size_t records) {
    const bool tiled = ctx.log_degree >= 12 && ctx.log_degree <= kMaxFusedLoadLogDegree;
    return tiled && source_moduli != 0 && source_moduli < record_rows && ctx.headroom_prefix == source_moduli &&
           ctx.approx_ok != 0 && ctx.forward_split_pairs != nullptr && records * record_rows > kOneGeneration &&
           records * record_rows <= (size_t(1) << 30);
}
hipError_t launch_ntt_lifted_forward(uint64_t* slab, const DeviceContext& ctx, uint32_t record_rows, size_t records,
                                     const uint64_t* base, const uint64_t* second, size_t stride, uint32_t source_moduli,
                                     hipStream_t stream) {
    if (records == 0) return hipSuccess;
    if (!ntt_lifted_forward_supported(ctx, record_rows, source_moduli, records)) return hipErrorNotSupported;
    const SpreadSource src{base, stride, source_moduli, 0, 0, second};
    const size_t rows = records * source_moduli;
    hipError_t e;
    switch (ctx.log_degree) {
        case 12: e = launch_forward_tiled<12, 9, kSourceRows>(kModeSplit, slab, ctx, 0, source_moduli, rows, src, stream, record_rows, 0); break;
        case 13: e = launch_forward_tiled<13, 10, kSourceRows>(kModeSplit, slab, ctx, 0, source_moduli, rows, src, stream, record_rows, 0); break;
        case 14: e = launch_forward_tiled<14, 10, kSourceRows>(kModeSplit, slab, ctx, 0, source_moduli, rows, src, stream, record_rows, 0); break;
        case 15: e = launch_interleaved_forward<2, kSourceRows>(kModeSplit, slab, ctx, make_row_map(0, source_moduli, record_rows, 0), rows, src, stream); break;
        default: return hipErrorNotSupported;  // (ntt_lifted_forward_supported says so first; a new degree must be named here)
    }
    if (e != hipSuccess) return e;
    return launch_ntt_band(false, slab, ctx, source_moduli, record_rows - source_moduli, record_rows, source_moduli, records,
                           kModeApprox, stream);
}

// Tensor product + inverse NTT of BEHZ multiplication in one kernel per row band: lifted [items][4][record_rows][N]
// (Eval) -> out [items][3][record_rows][N] (Coeff).  hipErrorNotSupported where no tiled kernel exists (the caller
// then runs launch_tensor + launch_ntt_mixed).
hipError_t launch_ntt_tensor_inverse(const uint64_t* lifted, uint64_t* out, const DeviceContext& ctx, uint32_t record_rows,
                                     size_t items, hipStream_t stream) {
    const bool tiled = ctx.log_degree >= 12 && ctx.log_degree <= kMaxFusedLoadLogDegree;
    const size_t records = items * 3;
    if (!tiled || records * record_rows > (size_t(1) << 30)) return hipErrorNotSupported;
    if (records == 0) return hipSuccess;
    const InverseSource spec{lifted, nullptr, 0, 0, nullptr, 0, nullptr, 0};
    BandRun runs[kMaxBandRuns];
    // (a product of a few ciphertexts is one workgroup generation whichever butterflies it takes: one launch over all rows in
    // the mode every modulus of the context accepts, like launch_ntt_mixed -- 13.6 + 14.5 -> 15 us on one ciphertext pair,
    // profiles/r06x_small_chains.txt)
    const int count = records * record_rows <= kOneGeneration ? 0 : band_runs(ctx, record_rows, runs);
    if (count <= 1)
        return launch_ntt_band(true, out, ctx, 0, record_rows, record_rows, 0, records, production_mode(ctx), stream,
                               kInverseFromTensor, spec);
    for (int k = 0; k < count; ++k) {
        hipError_t e = launch_ntt_band(true, out, ctx, runs[k].base, runs[k].rows, record_rows, runs[k].base, records,
                                       runs[k].mode, stream, kInverseFromTensor, spec);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

// Key-switching inner product + inverse NTT in one kernel (Bfv+Keys.swift:180-207): spread [polys][L][L+1][N] (Eval),
// key [top][2][top_rows][N] -> out [polys][2][L+1][N] (Coeff).  hipErrorNotSupported where no tiled kernel exists.
// The rows [first, first + count) of every (polynomial, c) record of the key-switching product, one launch per run of
// moduli of one butterfly class inside that range (band_runs over the whole key-switching context, clipped).
hipError_t launch_key_mac_runs(uint64_t* prod, const DeviceContext& ks, uint32_t first, uint32_t count, uint32_t L,
                               size_t records, int source, const InverseSource& spec, hipStream_t stream) {
    BandRun runs[kMaxBandRuns];
    const bool fold_shape = ks.log_degree == 12 || ks.log_degree == 13;
    const int run_count = fold_shape && records * count > kOneGeneration ? band_runs(ks, L + 1, runs) : 0;
    if (run_count <= 1)
        return launch_ntt_band(true, prod, ks, first, count, L + 1, first, records, production_mode(ks), stream, source, spec);
    for (int k = 0; k < run_count; ++k) {
        const uint32_t begin = runs[k].base > first ? runs[k].base : first;
        const uint32_t end = runs[k].base + runs[k].rows < first + count ? runs[k].base + runs[k].rows : first + count;
        if (begin >= end) continue;
        hipError_t e = launch_ntt_band(true, prod, ks, begin, end - begin, L + 1, begin, records, runs[k].mode, stream, source, spec);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_ntt_key_mac_inverse(const uint64_t* spread, const uint64_t* key, uint64_t* out, const DeviceContext& ks,
                                      uint32_t L, uint32_t top_rows, size_t polys, hipStream_t stream) {
    const bool tiled = ks.log_degree >= 12 && ks.log_degree <= (kFusedKeyMacAt32768 ? kMaxFusedLoadLogDegree : 14u) &&
                       (kFusedKeyMacAt16384 || ks.log_degree != 14);
    const size_t records = polys * 2;
    if (!tiled || L > 64 || records * (L + 1) > (size_t(1) << 30)) return hipErrorNotSupported;
    if (records == 0) return hipSuccess;
    return launch_key_mac_runs(out, ks, 0, L + 1, L, records, kInverseFromKeyMac,
                               InverseSource{spread, key, L, top_rows, nullptr, 0, nullptr, 0}, stream);
}

// The same with the key switch's last step applied as the rows r < L are stored (kInverseFromKeyMacFinish): first the q_ks
// row of every (polynomial, c) into `prod` (the only rows of it that are written), then the rows r < L, which read that
// row, drop the special modulus and add the update to the first `added_polys` polynomials of the ciphertext at
// ct_base + polynomial * ct_stride: out [polys][2][L][N].  hipErrorNotSupported (nothing launched) where the degree has
// no kernel that pairs the key columns; the caller then runs launch_ntt_key_mac_inverse + launch_key_switch_finish.
bool ntt_key_mac_finish_supported(const DeviceContext& ks, uint32_t L, size_t polys) {
    const bool tiled = ks.log_degree >= 12 && ks.log_degree <= 14;  // the degrees with a fused key-MAC transform
    // (a batch of up to two workgroup generations is latency-bound: there two transforms in a row cost more than one
    // transform and the small element-wise kernel -- single-query expansions lost 11-29 % with the fused end at every
    // level, and the Galois key switch crosses over between 96 and 128 ciphertexts at N = 8192, L = 4:
    // profiles/r04m_galois_fused_end_ab.txt)
    // (N = 16384: the key MAC runs as interleaved sub-rows, two workgroups per CU, and ends in the separate finish kernel --
    // kInterleavedKeyMacAt16384, measured against the 16-words-per-lane tile with the fused end)
    if (ks.log_degree == 14 && ((kInterleavedFusedLoads && kInterleavedKeyMacAt16384) || !kFusedKeyMacAt16384)) return false;
    return tiled && L >= 1 && L < 64 && ks.moduli_count == L + 1 && polys * 2 * (L + 1) > 2 * kOneGeneration &&
           polys * 2 * (L + 1) <= (size_t(1) << 30);
}
hipError_t launch_ntt_key_mac_inverse_finish(const uint64_t* spread, const uint64_t* key, uint64_t* prod,
                                             const KeySwitchEnd& end, const DeviceContext& ks, uint32_t L, uint32_t top_rows,
                                             size_t polys, hipStream_t stream) {
    if (!ntt_key_mac_finish_supported(ks, L, polys)) return hipErrorNotSupported;
    if (polys == 0) return hipSuccess;
    const InverseSource spec{spread, key, L, top_rows, end.ct_base, end.ct_stride, end.out, end.added_polys,
                             end.galois_inverse, end.expand_shift, end.own_base != nullptr ? end.own_base : end.ct_base,
                             end.targets_table, end.targets_group_size == 0 ? 1 : end.targets_group_size,
                             end.targets_group_stride, end.poly_base};
    hipError_t e = launch_key_mac_runs(prod, ks, L, 1, L, polys * 2, kInverseFromKeyMac, spec, stream);
    if (e != hipSuccess) return e;
    return launch_key_mac_runs(prod, ks, 0, L, L, polys * 2, kInverseFromKeyMacFinish, spec, stream);
}

// Inverse NTT of `rows` rows read from `source` and written to the same places of `slab` (the tiled 8-words-per-lane kernels of
// N = 4096 / 8192 load their rows through source_spec.first when it is set); hipErrorNotSupported (nothing launched) elsewhere.
hipError_t launch_ntt_inverse_out_of_place(const uint64_t* source, uint64_t* slab, const DeviceContext& ctx, uint32_t mod_base,
                                           uint32_t mod_period, size_t rows, hipStream_t stream) {
    if (rows == 0) return hipSuccess;
    if (source == nullptr || ctx.scaled_inverse_degree != 0 || ctx.approx_ok == 0 || rows > (size_t(1) << 30)) return hipErrorNotSupported;
    const InverseSource spec{source, nullptr, 0, 0, nullptr, 0, nullptr, 0};
    switch (ctx.log_degree) {
        case 12: return launch_tiled<12, 9>(true, production_mode(ctx), slab, ctx, mod_base, mod_period, rows, stream, 0, 0, kInverseFromSlab, spec);
        case 13: return launch_tiled<13, 10>(true, production_mode(ctx), slab, ctx, mod_base, mod_period, rows, stream, 0, 0, kInverseFromSlab, spec);
        default: return hipErrorNotSupported;
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
    switch (force_variant) {
        case kNttVariantAuto: case kNttVariantExact: case kNttVariantGeneric: case kNttVariantWide:
        case kNttVariantApprox: break;
        default: return hipErrorInvalidValue;
    }
    // kNttVariantExact / kNttVariantApprox pin the butterfly schedule (tests); every variant computes the same transform
    const int mode = (ctx.approx_ok == 0 || force_variant == kNttVariantExact || force_variant == kNttVariantGeneric)
                         ? kModeExact
                         : (force_variant == kNttVariantApprox ? kModeApprox : production_mode(ctx));
    // production choice (auto): 8 words per lane wherever a workgroup of <= 1024 lanes allows it.  Two rows fit the
    // CU's LDS at a time, so the waves per CU -- what hides the global/LDS latencies -- are set by the lanes per row:
    // 32 waves per CU with 8 words per lane against 16 with 16 words.
    if (force_variant == kNttVariantAuto || force_variant == kNttVariantExact || force_variant == kNttVariantApprox) {
        switch (ctx.log_degree) {
            case 12: return launch_tiled<12, 9>(inverse, mode, slab, ctx, mod_base, mod_period, rows, stream);
            case 13: return launch_tiled<13, 10>(inverse, mode, slab, ctx, mod_base, mod_period, rows, stream);
            case 14: return launch_tiled<14, 10>(inverse, mode, slab, ctx, mod_base, mod_period, rows, stream);
            case 15: return launch_interleaved<2>(inverse, mode, slab, ctx, make_row_map(mod_base, mod_period, 0, 0), rows, stream);
            default: break;
        }
    }
    if (force_variant == kNttVariantWide) {  // 16 words per lane
        switch (ctx.log_degree) {
            case 12: return launch_tiled<12, 8>(inverse, mode, slab, ctx, mod_base, mod_period, rows, stream);
            case 13: return launch_tiled<13, 9>(inverse, mode, slab, ctx, mod_base, mod_period, rows, stream);
            case 14: return launch_tiled<14, 10>(inverse, mode, slab, ctx, mod_base, mod_period, rows, stream);
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
