// ntt_rows.hpp -- the row-level machinery of the tiled transforms, shared by the translation units that build kernels from
// it (ntt_kernels.hip: the plain and fused-load transforms; behz_kernels.hip: the row-fused BEHZ multiplication): which rows
// a workgroup takes, the LDS exchange between passes, ROWS residue rows of one modulus registers to registers
// (forward_row / inverse_row), and the butterfly-class bookkeeping of the launchers.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

#include "device_context.hpp"
#include "device_math.hpp"
#include "ntt_common.hpp"

namespace heamd {
namespace ntt {

// Which row a workgroup transforms.  Workgroup b covers row (b / band_rows) * record_rows + band_offset + b % band_rows
// with modulus mod_base + b % band_rows: a launch covers a band of `band_rows` rows inside records of `record_rows`
// rows (record_rows = 0: the rows are consecutive and band_rows is the modulus period).  The quotient comes from one
// scalar multiply-high by band_magic = floor(2^32 / band_rows) + 1 and a fix-up (exact for b < 2^30), so that no
// vector instruction is spent on it.
struct RowMap {
    uint32_t mod_base, band_rows, band_magic, record_rows, band_offset;
    uint32_t record_base;  // records before the first one of this launch (the odd record after a launch of pairs)
};
__device__ __forceinline__ void locate(const RowMap& map, uint32_t block, uint32_t& record, uint32_t& within) {
    if (map.band_rows == 1) {
        record = block;
        within = 0;
        return;
    }
    uint32_t q = __umulhi(block, map.band_magic);
    int32_t r = static_cast<int32_t>(block - q * map.band_rows);
    if (r < 0) {
        q -= 1;
        r += static_cast<int32_t>(map.band_rows);
    }
    record = q;
    within = static_cast<uint32_t>(r);
}
inline RowMap make_row_map(uint32_t mod_base, uint32_t band_rows, uint32_t record_rows, uint32_t band_offset,
                    uint32_t record_base = 0) {
    return RowMap{mod_base, band_rows, static_cast<uint32_t>((uint64_t(1) << 32) / band_rows) + 1u, record_rows, band_offset,
                  record_base};
}
// A workgroup transforms ROWS rows of one modulus: the same band row of ROWS consecutive records.
template <int ROWS>
__device__ __forceinline__ void locate_rows(const RowMap& map, uint32_t block, size_t (&rows)[ROWS], uint32_t& record,
                                            uint32_t& within) {
    uint32_t group;
    locate(map, block, group, within);
    record = map.record_base + group * ROWS;
    const size_t stride = map.record_rows == 0 ? map.band_rows : map.record_rows;
    const size_t first = size_t(record) * stride + (map.record_rows == 0 ? 0 : map.band_offset) + within;
#pragma unroll
    for (int k = 0; k < ROWS; ++k) rows[k] = first + k * stride;
}

constexpr int min_waves_per_simd(int log_words_per_lane, int rows = 1) {
    const int words = rows << log_words_per_lane;  // 64-bit words of row data a lane holds
    return words <= 16 ? 8 : words <= 32 ? 4 : 2;
}

// ROWS rows of one modulus through ONE LDS tile, one after the other: a row's words leave in the layout of pass FROM
// and come back in the layout of pass TO.  Between two rows every reader of the first must be done before the
// second is written (the same fence the exchange itself needs: wave-private once a wave owns its slice of the row).
// PER_TRANSPOSE: the padding rule that suits this transpose's two layouts (ntt_common.hpp transpose_scheme) instead of
// the common one.  Every rule costs its own lane-base address words; the split-mode inverse kernel, which already
// spills three words, spills six with them and loses more (0.572 -> 0.585 ms) than the conflict-free transposes give,
// so it keeps the common rule; the forward kernels (0.510 -> 0.505 ms) and the [0, 8p) inverse (0.668 -> 0.659 ms) take
// the per-transpose rules (profiles/r02ze_lds_schemes.txt).
// Row groups of three and four (behz_kernels.hip) have the CU's LDS to themselves and go through kWideGroupTiles tiles side
// by side (2 x 76 KB of the CU's 160): that many rows per store / fence / load round, i.e. half the fences (workgroup
// barriers in two of the four transposes) and more LDS operations in flight per wave -- ct x ct +2.3 % over one tile
// (profiles/r05e_behz_tiles_and_lds_rules_ab.txt).
constexpr int kWideGroupTiles = 2;
template <int ROWS>
constexpr int kGroupTiles = ROWS >= 3 ? kWideGroupTiles : 1;
template <int LOGN, int LOGE, int LO_FROM, int W_FROM, int LO_TO, int W_TO, int ROWS, bool PER_TRANSPOSE = true>
__device__ __forceinline__ void exchange(uint64_t (&v)[ROWS][1 << LOGE], uint32_t tid, uint64_t* lds) {
    constexpr int TILES = kGroupTiles<ROWS>;
    constexpr uint32_t TILE_WORDS = lds_words(1u << LOGN);
    constexpr int SCHEME = PER_TRANSPOSE ? transpose_scheme<LOGN, LOGE, LO_FROM, LO_TO>() : 0;
#pragma unroll
    for (int row = 0; row < ROWS; row += TILES) {
        if (row > 0) lds_transpose_fence<LOGN, LOGE, LO_FROM, LO_TO>();
#pragma unroll
        for (int t = 0; t < TILES; ++t)
            if (row + t < ROWS) lds_store<LOGN, LOGE, LO_FROM, W_FROM, SCHEME>(v[row + t], tid, lds + t * TILE_WORDS);
        lds_transpose_fence<LOGN, LOGE, LO_FROM, LO_TO>();
#pragma unroll
        for (int t = 0; t < TILES; ++t)
            if (row + t < ROWS) lds_load<LOGN, LOGE, LO_TO, W_TO, SCHEME>(v[row + t], tid, lds + t * TILE_WORDS);
    }
}

// Which tiled kernels run the pass order with the partial pass on the top bits (ntt_common.hpp PassOrder): the plain-slab
// INVERSE transform at N = 8192 with 8 words per lane, -4.7 % (profiles/r03x_ntt_top_partial.txt).  The forward transform
// loses 2 % with it (its rows then leave as runs of 8 words through the staged store, and the per-transpose LDS rules
// were laid out for the other order); the fused inverse loads read their sources 16 bytes per lane in the low-pass
// layout, and with runs of 8 words per lane instead of 2 every load instruction takes a quarter of 64 lines
// (relinearize 675 -> 409 k/s) -- both keep the partial pass on the low bits.
template <int LOGN, int LOGE, bool INVERSE, bool FROM_SLAB>
constexpr bool kTopPartialOrder = LOGN == 13 && LOGE == 3 && INVERSE && FROM_SLAB;

// ROWS residue rows of one modulus, registers to registers: in -- the words of the top pass
// (element_index<LOGN, LOGE, LOGN - LOGE, LOGE>), out -- the canonical transforms in the layout of the last pass
// (element_index<LOGN, LOGE, 0, Schedule::R>).  The first twiddle of a pass is requested before the exchange that
// feeds the pass.
// CANONICAL = false: the words stay in the lazy range of MODE (more stages follow: ntt_forward_interleaved).
// One forward pass over element bits [LO_TO, LO_TO + LOGE) fed by the exchange out of the layout (LO_FROM, LOGE); its
// first twiddle is requested before the exchange.
// kLateLaneAddresses: every step derives its lane addresses (LDS slots, twiddle offsets) from an opaque copy of the lane
// index, i.e. next to where it uses them -- the compiler otherwise computes all of them at the top of the kernel and
// carries them through the passes, in scratch where the register file is full.
// Lane-major twiddle blocks (ntt_common.hpp Twiddles::lanes, built by PolyContext::upload for N = 4096 and N = 8192): which copy of
// the tables a tiled kernel with 8 words per lane reads -- 0: the plain tables (other shapes; the [0, 8p) / exact butterflies,
// whose tables have no lane-major copy), 1: the partition with the partial pass on the low bits (every forward kernel, the fused
// inverse ones, N = 4096), 2: on the top bit (the plain-slab inverse at N = 8192).  L2 read requests per launch of the headline
// pair: 24.2 -> 20.9 M forward, 31.9 -> 23.0 M inverse (profiles/r04x_lane_major_twiddles.txt).
template <int LOGN, int LOGT, int MODE, bool INVERSE, bool PLAIN>
constexpr int kLaneMajorTwiddles =
    !((LOGN == 13 && LOGT == 10) || (LOGN == 12 && LOGT == 9)) || !(is_split(MODE) || is_fold(MODE)) ? 0
    : (INVERSE && PLAIN && LOGN == 13)                                                              ? 2
                                                                                                    : 1;
constexpr bool kLateLaneAddresses = true;
// The lane index again, without a vector register between uses: wave base (the first lane's index -- one scalar register,
// computed once) + the lane's position in its wave (two instructions where it is needed).
__device__ __forceinline__ uint32_t late_lane(uint32_t lane) {
    uint32_t within;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(within));
    return static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(lane))) + within;
}
// Which form a step takes: the kernels of kModeSplitSigned re-derive the index (they sit at the 64-register cap with
// nothing to spare: no scratch in the plain-slab inverse at N = 4096 and in the key-MAC transform that way, N = 4096 inverse
// -2 ... -4 %); the others carry an opaque copy (the plain-slab inverse at N = 8192 is 2 % faster with it, and the forward
// kernels fit either way) -- profiles/r04t_inverse_forms_ab.txt.
// LATE: re-derive whatever the mode (the key-MAC transforms on the shift-folded products: their 64 registers hold two rows,
// four operand pairs of the inner product in flight and the sums -- a carried lane index and the addresses derived from it
// went to scratch, 20 B per lane at N = 8192; profiles/r06g_key_mac_scratch.txt).
template <int MODE, bool LATE = false>
__device__ __forceinline__ uint32_t step_lane(uint32_t lane) {
    if constexpr (!kLateLaneAddresses) return lane;
    else if constexpr (MODE == kModeSplitSigned || LATE) return late_lane(lane);
    else return opaque32(lane);
}
template <int LOGN, int LOGE, int LO_FROM, int LO_TO, int MODE, int ROWS, bool LATE = false>
__device__ __forceinline__ void forward_step(uint64_t (&v)[ROWS][1 << LOGE], uint32_t lane, const Twiddles<MODE>& tw, uint64_t p,
                                             uint64_t* lds) {
    const uint32_t tid = step_lane<MODE, LATE>(lane);
    const TwiddleWords first = forward_first_twiddle<LOGN, LOGE, LO_TO, LOGE, MODE, false>(tw, tid);
    exchange<LOGN, LOGE, LO_FROM, LOGE, LO_TO, LOGE, ROWS>(v, tid, lds);
    forward_pass<LOGN, LOGE, LO_TO, LOGE, MODE, false, ROWS>(v, tid, tw, p, false, first);
}

// TOP: the pass order with the partial pass on the top bits (ntt_common.hpp PassOrder); out -- the layout of the full pass
// on bits [0, LOGE).
template <int LOGN, int LOGE, int MODE, int ROWS, bool CANONICAL = true, bool TOP = false, bool LATE = false>
__device__ __forceinline__ void forward_row(uint64_t (&v)[ROWS][1 << LOGE], uint32_t tid, const Twiddles<MODE>& tw,
                                            uint64_t p, uint64_t* lds) {
    using S = Schedule<LOGN, LOGE>;
    constexpr int LO0 = LOGN - LOGE;
    if constexpr (PassOrder<LOGN, LOGE, TOP>::kTop) {
        using O = PassOrder<LOGN, LOGE, TOP>;
        // the partial pass in the layout of a full top pass: its stages pair the layout's top R bits
        forward_pass<LOGN, LOGE, LO0, LOGE, MODE, true, ROWS, S::R>(
            v, tid, tw, p, true, forward_first_twiddle<LOGN, LOGE, LO0, LOGE, MODE, true>(tw, tid));
        forward_step<LOGN, LOGE, LO0, O::lo(1), MODE, ROWS, LATE>(v, tid, tw, p, lds);
        if constexpr (S::P >= 3) forward_step<LOGN, LOGE, O::lo(1), O::lo(2), MODE, ROWS, LATE>(v, tid, tw, p, lds);
        if constexpr (S::P >= 4) forward_step<LOGN, LOGE, O::lo(2), O::lo(3), MODE, ROWS, LATE>(v, tid, tw, p, lds);
        if constexpr (S::P >= 5) forward_step<LOGN, LOGE, O::lo(3), O::lo(4), MODE, ROWS, LATE>(v, tid, tw, p, lds);
        static_assert(O::lo(S::P - 1) == 0, "the last full pass sits on bits [0, LOGE)");
        if constexpr (CANONICAL) canonicalize_all<MODE>(v, p);
        return;
    }
    forward_pass<LOGN, LOGE, LO0, LOGE, MODE, true, ROWS>(
        v, tid, tw, p, true, forward_first_twiddle<LOGN, LOGE, LO0, LOGE, MODE, true>(tw, tid));
    // (LATE: each step re-derives the lane index -- step_lane; this order otherwise takes the kernel's own copy)
    [[maybe_unused]] auto at_step = [&]() { return LATE ? late_lane(tid) : tid; };
    if constexpr (S::P >= 3) {
        constexpr int LO1 = LOGN - 2 * LOGE;
        const uint32_t lane = at_step();
        const TwiddleWords first = forward_first_twiddle<LOGN, LOGE, LO1, LOGE, MODE, false>(tw, lane);
        exchange<LOGN, LOGE, LO0, LOGE, LO1, LOGE, ROWS>(v, lane, lds);
        forward_pass<LOGN, LOGE, LO1, LOGE, MODE, false, ROWS>(v, lane, tw, p, false, first);
    }
    if constexpr (S::P >= 4) {
        constexpr int LO1 = LOGN - 2 * LOGE, LO2 = LOGN - 3 * LOGE;
        const uint32_t lane = at_step();
        const TwiddleWords first = forward_first_twiddle<LOGN, LOGE, LO2, LOGE, MODE, false>(tw, lane);
        exchange<LOGN, LOGE, LO1, LOGE, LO2, LOGE, ROWS>(v, lane, lds);
        forward_pass<LOGN, LOGE, LO2, LOGE, MODE, false, ROWS>(v, lane, tw, p, false, first);
    }
    if constexpr (S::P >= 5) {
        constexpr int LO2 = LOGN - 3 * LOGE, LO3 = LOGN - 4 * LOGE;
        const uint32_t lane = at_step();
        const TwiddleWords first = forward_first_twiddle<LOGN, LOGE, LO3, LOGE, MODE, false>(tw, lane);
        exchange<LOGN, LOGE, LO2, LOGE, LO3, LOGE, ROWS>(v, lane, lds);
        forward_pass<LOGN, LOGE, LO3, LOGE, MODE, false, ROWS>(v, lane, tw, p, false, first);
    }
    {
        constexpr int LO_PREVIOUS = LOGN - (S::P - 1) * LOGE;
        const uint32_t lane = at_step();
        const TwiddleWords first = forward_first_twiddle<LOGN, LOGE, 0, S::R, MODE, false>(tw, lane);
        exchange<LOGN, LOGE, LO_PREVIOUS, LOGE, 0, S::R, ROWS>(v, lane, lds);
        forward_pass<LOGN, LOGE, 0, S::R, MODE, false, ROWS>(v, lane, tw, p, false, first);
    }
    if constexpr (CANONICAL) canonicalize_all<MODE>(v, p);
}

// One inverse pass over element bits [LO_TO, LO_TO + LOGE) fed by the exchange out of the layout of pass
// (LO_FROM, W_FROM).  Its first twiddle is requested after the exchange: requesting it before (as the forward
// transform does, the gather then overlaps the LDS round trip) keeps six more registers live across the exchange and
// doubles the inverse kernel's spills to scratch -- 0.659 against 0.620 ms per launch (profiles/r02d_ntt_ab_inverse_variants.txt).
template <int MODE>
constexpr bool kInverseFirstTwiddleEarly = false;
constexpr bool kWideGroupFirstTwiddleEarly = false;
// the limb-wise inverse of a wide row group may take the per-transpose LDS padding rules (ntt_common.hpp transpose_scheme)
// that the 64-register kernels cannot afford (exchange<> PER_TRANSPOSE)
constexpr bool kWideGroupPerTransposeLds = false;
template <int LOGN, int LOGE, int LO_FROM, int W_FROM, int LO_TO, int MODE, bool UNIFORM, int ROWS, bool SCALED, int PRIOR = 0,
          int LOGD = LOGN, int FIRST_STAGE = 0, bool LATE = false>
__device__ __forceinline__ void inverse_step(uint64_t (&v)[ROWS][1 << LOGE], uint32_t lane, const Twiddles<MODE>& tw,
                                             const DeviceModulus& mod, uint64_t* lds) {
    const uint32_t tid = step_lane<MODE, LATE>(lane);
    // (row groups of three and four -- behz_kernels.hip, 128 registers per lane -- have the registers for requests before the
    // exchange and for more than one twiddle in flight: kWideGroupFirstTwiddleEarly, ntt_common.hpp kWideGroupTwiddlesAhead)
    constexpr int AHEAD = kGroupTwiddlesAhead<MODE, ROWS>;
    constexpr bool EARLY = ROWS >= 3 ? kWideGroupFirstTwiddleEarly : kInverseFirstTwiddleEarly<MODE>;
    TwiddleWords head[AHEAD];
    if constexpr (EARLY) inverse_first_twiddles<LOGN, LOGE, LO_TO, LOGE, MODE, UNIFORM, AHEAD, FIRST_STAGE>(head, tw, tid);
    exchange<LOGN, LOGE, LO_FROM, W_FROM, LO_TO, LOGE, ROWS, !is_split(MODE) || (ROWS >= 3 && kWideGroupPerTransposeLds)>(v, tid, lds);
    if constexpr (!EARLY) inverse_first_twiddles<LOGN, LOGE, LO_TO, LOGE, MODE, UNIFORM, AHEAD, FIRST_STAGE>(head, tw, tid);
    inverse_pass<LOGN, LOGE, LO_TO, LOGE, MODE, UNIFORM, ROWS, SCALED, PRIOR, LOGD, FIRST_STAGE, AHEAD>(v, tid, tw, mod, false, head);
}

// ROWS residue rows of the inverse transform, registers to registers: in -- the words of the low pass
// (element_index<LOGN, LOGE, 0, Schedule::R>), out -- canonical words in the layout of the top pass.
// PRIOR / LOGD: the rows are the sub-rows of an interleaved row of degree 2^LOGD whose first PRIOR stages already ran
// (ntt_inverse_interleaved); `tw` then indexes the tail of the degree's table.
// HEAD / `head`: the first HEAD twiddles of the first pass (inverse_row_head), which then keeps HEAD of them in flight:
// the caller requests them BEFORE the rows are loaded -- they travel with the row loads instead of starting a chain of L2
// round trips (one per twiddle of the gather-heavy first pass) behind them.
template <int LOGN, int LOGE, int MODE, bool TOP, int HEAD>
__device__ __forceinline__ void inverse_row_head(TwiddleWords (&head)[HEAD], const Twiddles<MODE>& tw, uint32_t tid) {
    constexpr int LOW = PassOrder<LOGN, LOGE, TOP>::LOW;
    inverse_first_twiddles<LOGN, LOGE, 0, LOW, MODE, false, HEAD>(head, tw, tid);
}
// The transform's last step (the exchange into the top pass and that pass) ROW BY ROW, each row handed to `finish` as soon
// as its canonical words exist: row 0 is brought into the top layout, row 1 is parked in the tile behind it -- its
// registers are free while row 0 runs its last pass and whatever `finish` does with it (loads of other operands, the
// store) -- and is picked up afterwards.  Same barriers as the exchange of both rows; the pass's wave-uniform twiddles are
// scalar loads and are simply read again for the second row.
struct NoFinish {};
template <int LOGN, int LOGE, int LO_FROM, int W_FROM, int MODE, int ROWS, bool SCALED, int PRIOR, int LOGD, bool LATE, typename Finish>
__device__ __forceinline__ void inverse_last_step_by_row(uint64_t (&v)[ROWS][1 << LOGE], uint32_t lane, const Twiddles<MODE>& tw,
                                                         const DeviceModulus& mod, uint64_t* lds, Finish& finish) {
    static_assert(ROWS <= 2, "one row in registers, one parked in the tile");
    const uint32_t tid = step_lane<MODE, LATE>(lane);
    constexpr int E = 1 << LOGE, LOL = LOGN - LOGE;
    constexpr int SCHEME = !is_split(MODE) ? transpose_scheme<LOGN, LOGE, LO_FROM, LOL>() : 0;  // as exchange<> in inverse_step
    lds_store<LOGN, LOGE, LO_FROM, W_FROM, SCHEME>(v[0], tid, lds);
    lds_transpose_fence<LOGN, LOGE, LO_FROM, LOL>();
    lds_load<LOGN, LOGE, LOL, LOGE, SCHEME>(v[0], tid, lds);
    if constexpr (ROWS == 2) {
        lds_transpose_fence<LOGN, LOGE, LO_FROM, LOL>();
        lds_store<LOGN, LOGE, LO_FROM, W_FROM, SCHEME>(v[ROWS - 1], tid, lds);
    }
    // (the rows one after the other as straight-line code, not as a loop to unroll: with the key switch's three ends in it the
    // body is past the optimizer's unrolling budget, and a rolled loop indexes the rows at run time -- i.e. out of scratch)
    auto one_row = [&](auto row_tag) {
        constexpr int k = decltype(row_tag)::value;
        if constexpr (k == 1) {
            lds_transpose_fence<LOGN, LOGE, LO_FROM, LOL>();
            lds_load<LOGN, LOGE, LOL, LOGE, SCHEME>(v[ROWS - 1], tid, lds);
        }
        uint64_t (&row)[1][E] = *reinterpret_cast<uint64_t (*)[1][E]>(&v[k]);
        const TwiddleWords head[1] = {inverse_first_twiddle<LOGN, LOGE, LOL, LOGE, MODE, true>(tw, tid)};
        inverse_pass<LOGN, LOGE, LOL, LOGE, MODE, true, 1, SCALED, PRIOR, LOGD, 0, 1>(row, tid, tw, mod, false, head);
        finish(k, row[0]);
    };
    one_row(std::integral_constant<int, 0>{});
    if constexpr (ROWS == 2) one_row(std::integral_constant<int, 1>{});
}

// Finish: NoFinish, or a callable (row index, the row's canonical words in the top layout) that takes over each row as it
// is completed (inverse_last_step_by_row; only with the partial pass on the low bits).
template <int LOGN, int LOGE, int MODE, int ROWS, bool SCALED, int PRIOR = 0, int LOGD = LOGN, bool TOP = false, int HEAD = 1,
          bool LATE = false, typename Finish = NoFinish>
__device__ __forceinline__ void inverse_row(uint64_t (&v)[ROWS][1 << LOGE], uint32_t tid, const Twiddles<MODE>& tw,
                                            const DeviceModulus& mod, uint64_t* lds, const TwiddleWords (&head)[HEAD],
                                            Finish finish = Finish{}) {
    using S = Schedule<LOGN, LOGE>;
    constexpr int R = S::R, LOL = LOGN - LOGE;
    constexpr bool BY_ROW = !std::is_same<Finish, NoFinish>::value;
    static_assert(!BY_ROW || !PassOrder<LOGN, LOGE, TOP>::kTop, "rows are finished one by one in the low-partial order only");
    if constexpr (PassOrder<LOGN, LOGE, TOP>::kTop) {
        // in -- the layout of the full pass on bits [0, LOGE); the full passes from the low bits up, then the partial
        // pass (the transform's last R stages) in the layout of a full top pass
        using O = PassOrder<LOGN, LOGE, TOP>;
        inverse_pass<LOGN, LOGE, 0, LOGE, MODE, false, ROWS, false, PRIOR, LOGD, 0, HEAD>(v, tid, tw, mod, PRIOR == 0, head);
        if constexpr (S::P >= 5)
            inverse_step<LOGN, LOGE, O::lo(4), LOGE, O::lo(3), MODE, false, ROWS, SCALED, PRIOR, LOGD, 0, LATE>(v, tid, tw, mod, lds);
        if constexpr (S::P >= 4)
            inverse_step<LOGN, LOGE, O::lo(3), LOGE, O::lo(2), MODE, false, ROWS, SCALED, PRIOR, LOGD, 0, LATE>(v, tid, tw, mod, lds);
        if constexpr (S::P >= 3)
            inverse_step<LOGN, LOGE, O::lo(2), LOGE, O::lo(1), MODE, false, ROWS, SCALED, PRIOR, LOGD, 0, LATE>(v, tid, tw, mod, lds);
        inverse_step<LOGN, LOGE, O::lo(1), LOGE, LOL, MODE, true, ROWS, SCALED, PRIOR, LOGD, LOGE - R, LATE>(v, tid, tw, mod, lds);
        return;
    }
    inverse_pass<LOGN, LOGE, 0, R, MODE, false, ROWS, false, PRIOR, LOGD, 0, HEAD>(v, tid, tw, mod, PRIOR == 0, head);
    if constexpr (S::P >= 3)
        inverse_step<LOGN, LOGE, 0, R, R, MODE, false, ROWS, SCALED, PRIOR, LOGD, 0, LATE>(v, tid, tw, mod, lds);
    if constexpr (S::P >= 4)
        inverse_step<LOGN, LOGE, R, LOGE, R + LOGE, MODE, false, ROWS, SCALED, PRIOR, LOGD, 0, LATE>(v, tid, tw, mod, lds);
    if constexpr (S::P >= 5)
        inverse_step<LOGN, LOGE, R + LOGE, LOGE, R + 2 * LOGE, MODE, false, ROWS, SCALED, PRIOR, LOGD, 0, LATE>(v, tid, tw, mod, lds);
    // into the top pass (uniform twiddles; its last stage folds in N^-1)
    if constexpr (BY_ROW) {
        if constexpr (S::P == 2) inverse_last_step_by_row<LOGN, LOGE, 0, R, MODE, ROWS, SCALED, PRIOR, LOGD, LATE>(v, tid, tw, mod, lds, finish);
        else inverse_last_step_by_row<LOGN, LOGE, LOL - LOGE, LOGE, MODE, ROWS, SCALED, PRIOR, LOGD, LATE>(v, tid, tw, mod, lds, finish);
    } else if constexpr (S::P == 2) {
        inverse_step<LOGN, LOGE, 0, R, LOL, MODE, true, ROWS, SCALED, PRIOR, LOGD, 0, LATE>(v, tid, tw, mod, lds);
    } else {
        inverse_step<LOGN, LOGE, LOL - LOGE, LOGE, LOL, MODE, true, ROWS, SCALED, PRIOR, LOGD, 0, LATE>(v, tid, tw, mod, lds);
    }
}

// The fused inverse loads (tensor product, key MAC) hand the transform the one-word-quotient Barrett's remainder as it
// comes, in [0, 5p), where the butterflies take lazy input: the limb-wise ones (any word below 2^63; the stage bounds are
// those of a row whose first kLazyInputStages stages already ran: inverse_in_shift(3) = 5 covers 5p) and the fold ones
// (words below 6p).  Three conditional subtracts less per word; the [0, 8p) / exact butterflies keep canonical input.
template <int MODE>
constexpr bool kLazyTransformInput = is_split(MODE) || is_fold(MODE);
constexpr int kLazyInputStages = 3;
template <typename Kernel>
inline hipError_t allow_dynamic_lds(Kernel kernel, size_t lds_bytes) {
    if (lds_bytes <= 48 * 1024) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                               static_cast<int>(lds_bytes));
}

// The fold butterflies (ntt_common.hpp kModeFoldMinus / kModeFoldPlus) in place of the [0, 8p) ones where every modulus of
// the launch allows them -- the 60-bit moduli of the reference's parameter sets, the 61-bit BEHZ auxiliary primes: the
// plain-slab transforms and the tensor-fused inverse of the production shapes at N = 4096 / 8192
// (profiles/r03w_ntt_fold_butterflies.txt).  0: not all of one form.
constexpr bool kFoldButterflies = true;
template <int LOGN, int LOGT>
constexpr bool kFoldShape = kFoldButterflies && ((LOGN == 12 && LOGT == 9) || (LOGN == 13 && LOGT == 10));
inline int fold_mode(const DeviceContext& ctx, uint32_t mod_base, uint32_t band_rows) {
    if (band_rows == 0 || mod_base + band_rows > 64) return 0;
    const uint64_t band = (band_rows == 64 ? ~uint64_t(0) : ((uint64_t(1) << band_rows) - 1)) << mod_base;
    if ((ctx.fold_minus_mask & band) == band) return kModeFoldMinus;
    if ((ctx.fold_plus_mask & band) == band) return kModeFoldPlus;
    return 0;
}


// the production butterfly schedule for a context (what kNttVariantAuto picks)
inline int production_mode(const DeviceContext& ctx) {
    if (ctx.approx_ok == 0) return kModeExact;
    if (ctx.headroom_ok == 0 || ctx.forward_split_pairs == nullptr) return kModeApprox;
    return kModeSplit;
}


// The rows of a record (row r uses modulus r) as runs of moduli of one butterfly class -- the fold-free limb-wise
// butterflies (the leading moduli in [2^40, 2^55)), the fold butterflies of either form, the [0, 8p) ones: one launch per
// run over that row band of every record.  BEHZ's [Q, Bsk] records with the usual 55-bit ciphertext moduli are two runs
// (Q | Bsk); with the reference's 60-bit parameter sets, e.g. 29 | 60, 60 | Bsk.  0 runs: the context takes one mode as a
// whole (a modulus above 2^61: exact butterflies) or has no tables for a split.
struct BandRun {
    uint32_t base, rows;
    int mode;  // what launch_ntt_band is given: kModeSplit or kModeApprox (a fold form is resolved from the band's moduli)
};
constexpr int kMaxBandRuns = 8;
inline int band_runs(const DeviceContext& ctx, uint32_t record_rows, BandRun (&runs)[kMaxBandRuns]) {
    if (ctx.approx_ok == 0 || ctx.forward_split_pairs == nullptr || record_rows == 0 || record_rows > 64) return 0;
    const uint32_t prefix = ctx.headroom_prefix < record_rows ? ctx.headroom_prefix : record_rows;
    auto row_class = [&](uint32_t r) {
        if (r < prefix) return 0;
        if (((ctx.fold_minus_mask >> r) & 1) != 0) return 1;
        if (((ctx.fold_plus_mask >> r) & 1) != 0) return 2;
        return 3;
    };
    int count = 0;
    for (uint32_t r = 0; r < record_rows;) {
        const int cls = row_class(r);
        uint32_t end = r + 1;
        while (end < record_rows && row_class(end) == cls) ++end;
        if (count == kMaxBandRuns) return 0;  // a context this fragmented takes one launch in the common mode
        runs[count++] = BandRun{r, end - r, cls == 0 ? kModeSplit : kModeApprox};
        r = end;
    }
    return count;
}

// a handful of rows (one ciphertext's worth: the tail of a PIR response) is one workgroup generation either way: one
// launch in the mode that serves every modulus costs one kernel latency instead of two
constexpr size_t kOneGeneration = 512;

}  // namespace ntt
}  // namespace heamd
