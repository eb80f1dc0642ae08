// kernels.hpp -- host-callable launchers of the HIP kernels (all asynchronous on `stream`).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "device_context.hpp"

namespace heamd {

// kernel schedule for launch_ntt (tests pin each against the oracle; production callers pass kNttVariantAuto).  Every
// variant computes the same canonical transform.
enum {
    kNttVariantAuto = 0,     // tiled kernel, 8 words per lane where a workgroup of <= 1024 lanes allows it
    kNttVariantExact = 1,    // same kernel, exact-quotient butterflies (what moduli >= 2^61 get)
    kNttVariantGeneric = 2,  // radix-2 stage loop (what degrees without a tiled kernel get)
    kNttVariantWide = 3,     // tiled kernel, 16 words per lane
    kNttVariantApprox = 10   // production kernel pinned to the [0, 8p) butterflies (what 56..61-bit moduli get)
};

// NTT of `rows` contiguous length-N rows.  Row r uses modulus index mod_base + (r % mod_period).
hipError_t launch_ntt(bool inverse, uint64_t* slab, const DeviceContext& ctx, uint32_t mod_base, uint32_t mod_period,
                      size_t rows, hipStream_t stream, int force_variant = kNttVariantAuto);
// Inverse NTT out of place: rows read from `source`, written to the same places of `slab` (N = 4096 / 8192 with every modulus
// below 2^61; hipErrorNotSupported, nothing launched, elsewhere)
hipError_t launch_ntt_inverse_out_of_place(const uint64_t* source, uint64_t* slab, const DeviceContext& ctx, uint32_t mod_base,
                                           uint32_t mod_period, size_t rows, hipStream_t stream);
// Key-switching decomposition fused into the forward NTT (Bfv+Keys.swift:165-179): spread [polys][L][L+1][N] row
// (poly, j, r) = NTT_{ks modulus r}( source row j of polynomial `poly`, reduced mod r when q_j > modulus r ), read
// from source + poly * poly_stride + j * N.  galois_inverse = g^-1 mod 2N: the source polynomial is first taken through
// f(x) -> f(x^g) (Coeff form, PolyRq/Galois.swift:115-143) as its rows are loaded; 0: as it is.
// hipErrorNotSupported for degrees without a tiled kernel (the caller then runs launch_key_switch_spread + launch_ntt).
hipError_t launch_ntt_spread(const uint64_t* source, size_t poly_stride, uint32_t source_moduli, size_t polys,
                             uint64_t* spread, const DeviceContext& ks_ctx, uint32_t galois_inverse, hipStream_t stream);
// Plaintext.convertToEvalFormat fused into the forward NTT (Plaintext.swift:149-170): out [polys][L][N] row (poly, r)
// = NTT_{q_r}(centred lift of plaintexts[poly][N] mod q_r).  hipErrorNotSupported for degrees without a tiled kernel.
hipError_t launch_ntt_lift(const uint64_t* plaintexts, uint64_t plaintext_modulus, size_t polys, uint64_t* out,
                           const DeviceContext& ctx, hipStream_t stream);
// NTT of `records` records of `record_rows` rows each (row r uses modulus r), the leading ctx.headroom_prefix rows on
// the fold-free butterflies and the rest on the [0, 8p) ones (the [Q, Bsk] slabs of BEHZ multiplication)
hipError_t launch_ntt_mixed(bool inverse, uint64_t* slab, const DeviceContext& ctx, uint32_t record_rows, size_t records,
                            hipStream_t stream);
// The rows [first, first + count) of every record only; hipErrorNotSupported (nothing launched) for degrees without a tiled kernel.
hipError_t launch_ntt_record_band(bool inverse, uint64_t* slab, const DeviceContext& ctx, uint32_t record_rows, uint32_t first,
                                  uint32_t count, size_t records, hipStream_t stream);
// Forward NTT of lifted [Q, Bsk] records [records][record_rows][N] whose first source_moduli rows were left unwritten by
// the lift: they are read from the source polynomials -- record = item * 4 + slot from polynomial slot & 1 of item
// `item` of `base` (slots 0, 1) or `second` (slots 2, 3), items `stride` words apart; second == nullptr: record r from
// base + r * stride.  hipErrorNotSupported (ntt_lifted_forward_supported false): lift with the copy, launch_ntt_mixed.
bool ntt_lifted_forward_supported(const DeviceContext& ctx, uint32_t record_rows, uint32_t source_moduli, size_t records);
hipError_t launch_ntt_lifted_forward(uint64_t* slab, const DeviceContext& ctx, uint32_t record_rows, size_t records,
                                     const uint64_t* base, const uint64_t* second, size_t stride, uint32_t source_moduli,
                                     hipStream_t stream);
hipError_t launch_ntt_tensor_inverse(const uint64_t* lifted, uint64_t* out, const DeviceContext& ctx, uint32_t record_rows,
                                     size_t items, hipStream_t stream);
// behz_kernels.hip -- BEHZ multiplication row by row (Bfv+Multiply.swift:18-85): per (item, [Q, Bsk] row) the four forward
// transforms, the tensor product and the three inverse transforms scaled by t in one workgroup, the transformed rows never
// leaving its registers.  lhs, rhs: [items][2][source_moduli][N] Coeff, items ct_stride words apart (their rows are the Q
// rows of the lifted polynomials); lifted: [items][4][record_rows][N] with the Bsk rows written by the lift (the Q rows are
// not read); scaled_qbsk: the [Q, Bsk] context with t N^-1 as inverse-degree constants; out: [items][3][record_rows][N]
// Coeff, what launch_floor_qbsk_to_q takes.  hipErrorNotSupported (nothing launched, behz_rows_fused_supported false):
// launch_ntt_lifted_forward + launch_ntt_tensor_inverse.
bool behz_rows_fused_supported(const DeviceContext& qbsk, uint32_t record_rows, uint32_t source_moduli, size_t items);
// whether the row bands that read the lift's rows take them in [0, 5p) (every such row on the fold butterflies of the plus form,
// whose first stage wants x < 8p and any y): the lift may then skip its three conditional subtracts per word
bool behz_lifted_rows_may_be_lazy(const DeviceContext& scaled_qbsk, uint32_t record_rows, uint32_t source_moduli);
constexpr int kBehzAllRows = 0, kBehzCiphertextRows = 1, kBehzLiftedRows = 2;
hipError_t launch_behz_rows_fused(const uint64_t* lhs, const uint64_t* rhs, size_t ct_stride, const uint64_t* lifted,
                                  uint64_t* out, const DeviceContext& scaled_qbsk, uint32_t record_rows, uint32_t source_moduli,
                                  size_t items, hipStream_t stream, int part = kBehzAllRows);
hipError_t launch_ntt_key_mac_inverse(const uint64_t* spread, const uint64_t* key, uint64_t* out, const DeviceContext& ks,
                                      uint32_t L, uint32_t top_rows, size_t polys, hipStream_t stream);
// launch_ntt_key_mac_inverse with the key switch's last step (drop the special modulus, add the update to the first
// `added_polys` polynomials of the ciphertext at ct_base + polynomial * ct_stride) applied as the rows are stored:
// out [polys][2][L][N]; of `prod` ([polys][2][L+1][N]) only the q_ks rows are written.  hipErrorNotSupported (nothing
// launched, ntt_key_mac_finish_supported false): launch_ntt_key_mac_inverse + launch_key_switch_finish.
// KeySwitchEnd says which end: relinearize (the update added to the first `added_polys` polynomials), the Galois key switch
// (galois_inverse = g^-1 mod 2N: ct_base holds the ciphertexts BEFORE the automorphism, c0 is read through it), or one step
// of PirUtil.expand on top of that (expand_shift != 0: out takes the two children of every polynomial; own_base: the
// ciphertexts the children are formed with, nullptr = ct_base; targets_*: rns_kernels.hpp ExpandTargets).  poly_base: the
// launch's first polynomial in ct_base / own_base / out (spread, key and prod are passed from a run's own start).
struct KeySwitchEnd {
    const uint64_t* ct_base;
    size_t ct_stride;
    uint64_t* out;
    uint32_t added_polys;
    uint32_t galois_inverse = 0, expand_shift = 0;
    const uint64_t* own_base = nullptr;
    const uint32_t* targets_table = nullptr;
    size_t targets_group_size = 1, targets_group_stride = 0;
    size_t poly_base = 0;
};
bool ntt_key_mac_finish_supported(const DeviceContext& ks, uint32_t L, size_t polys);
hipError_t launch_ntt_key_mac_inverse_finish(const uint64_t* spread, const uint64_t* key, uint64_t* prod,
                                             const KeySwitchEnd& end, const DeviceContext& ks, uint32_t L, uint32_t top_rows,
                                             size_t polys, hipStream_t stream);
const char* ntt_variant_name(uint32_t log_degree);

enum class ElementwiseOp : int { Add = 0, Sub = 1, Neg = 2, Mul = 3, MulScalar = 4 };
// lhs[k] = op(lhs[k], rhs[k]) over `rows` rows of [..][L][N]; rhs may be NULL for Neg.  For MulScalar `rhs` is a
// device array of L (scalar, shoup) pairs.
hipError_t launch_elementwise(ElementwiseOp op, uint64_t* lhs, const uint64_t* rhs, const DeviceContext& ctx,
                              size_t rows, hipStream_t stream);
// ct [batch][polys][L][N] *= pt [batch][L][N]
hipError_t launch_mul_plain(uint64_t* ct, const uint64_t* pt, const DeviceContext& ctx, uint32_t poly_count,
                            size_t batch, hipStream_t stream);
// divideAndRoundQLast with the first `moduli_count` moduli of ctx: in [polys][L][N] -> out [polys][L-1][N]
hipError_t launch_divide_and_round_q_last(const uint64_t* in, uint64_t* out, const DeviceContext& ctx,
                                          uint32_t moduli_count, size_t polys, hipStream_t stream);
// Ciphertext.modSwitchDownToSingle: in [polys][moduli_count][N] -> out [polys][1][N], the moduli_count - 1 steps of
// divideAndRoundQLast in one kernel (2..8 moduli; hipErrorNotSupported otherwise: chain the kernel above).
hipError_t launch_mod_switch_down_to_single(const uint64_t* in, uint64_t* out, const DeviceContext& ctx,
                                            uint32_t moduli_count, size_t polys, hipStream_t stream);
hipError_t launch_adding_lazy_product(const uint64_t* lhs, const uint64_t* rhs, uint64_t* acc_lo_hi,
                                      const DeviceContext& ctx, hipStream_t stream);
hipError_t launch_reduce_accumulator(const uint64_t* acc_lo_hi, uint64_t* out, const DeviceContext& ctx,
                                     hipStream_t stream);
// out[col][poly][L][N] = sum_k cts[k][poly][L][N] * pts[col][k][L][N]  (skip where present[col*count+k] == 0).
// max_lazy: the reference's reduction cadence (used as is by the 128-bit accumulator kernel for degree < 256);
// cadence: products between reductions of the carry-counting accumulator: <= max_lazy and sums below 2^127
// (0 = use the 128-bit kernel).  narrow_moduli: every modulus of ctx is below 2^56 (canonical operands then let the
// accumulator drop two of its three carry counts; the cadence is cut to kNarrowProductSumCadence).
// (W: uint64_t or uint32_t slabs)
template <typename W>
hipError_t launch_inner_product_plain(const W* cts, const W* pts, const uint8_t* present_device, W* out,
                                      const DeviceContext& ctx, uint32_t poly_count, size_t count, size_t columns,
                                      uint64_t max_lazy, uint64_t cadence, bool narrow_moduli, hipStream_t stream);

// ---- packed residue rows: the resident PIR database without the zero top bits of its words -------------------------
// A library-private layout (nothing in the reference's wire format): row r of a polynomial is a little-endian bit
// stream of N fields of width[r] = bits(q_r) bits -- coefficient i occupies stream bits [i width, (i + 1) width), stream
// word j is bits [64 j, 64 j + 64) -- and starts at 8-byte word word_offset[r] of the polynomial (degree a multiple of
// 64: whole words).  A 55-bit modulus stores 6.875 bytes per word instead of 8.
constexpr uint32_t kMaxPackedRows = 8;
struct PackedLayout {
    uint32_t rows;
    uint32_t width[kMaxPackedRows];
    uint32_t word_offset[kMaxPackedRows + 1];  // [rows] = 8-byte words per packed polynomial
};
// slab [polys][rows][N] -> packed [polys][word_offset[rows]] words
hipError_t launch_pack_rows(const uint64_t* slab, uint64_t* packed, const PackedLayout& layout, uint32_t log_degree,
                            size_t polys, hipStream_t stream);
// launch_inner_product_plain with the plaintexts packed: pts [columns][count][word_offset[rows]] words.  poly_count 1..3,
// degree >= 256.
hipError_t launch_inner_product_plain_packed(const uint64_t* cts, const uint64_t* packed_pts, const PackedLayout& layout,
                                             const uint8_t* present_device, uint64_t* out, const DeviceContext& ctx,
                                             uint32_t poly_count, size_t count, size_t columns, uint64_t cadence,
                                             bool narrow_moduli, hipStream_t stream);
// ct [batch][polys][L][N] *= pt [batch][L][N] on 4-byte words
hipError_t launch_mul_plain32(uint32_t* ct, const uint32_t* pt, const DeviceContext& ctx, uint32_t poly_count, size_t batch,
                              hipStream_t stream);


// word32_kernels.hip: PolyRq<UInt32> (4-byte words; every modulus <= 2^30 - 1)
hipError_t launch_ntt32(bool inverse, uint32_t* slab, const DeviceContext32& ctx, uint32_t mod_base, uint32_t mod_period,
                        size_t rows, hipStream_t stream);
// the same transforms with the step before them applied to the words as they are loaded (word32_kernels.hip Source32):
// the key-switching decomposition, the BEHZ tensor product, the inner product with the key-switching key.
// hipErrorNotSupported where the degree has no tiled 4-byte transform (the caller runs the unfused kernels).
hipError_t launch_ntt32_spread(const uint32_t* target, size_t stride, uint32_t L, size_t polys, uint32_t* spread,
                               const DeviceContext32& ks_ctx, hipStream_t stream);
bool ntt32_lifted_forward_supported(const DeviceContext32& ctx);
hipError_t launch_ntt32_lifted_forward(uint32_t* lifted, const DeviceContext32& ctx, uint32_t record_rows, size_t items,
                                       const uint32_t* lhs, const uint32_t* rhs, size_t stride, uint32_t L,
                                       hipStream_t stream);
hipError_t launch_ntt32_tensor_inverse(const uint32_t* lifted, uint32_t* out, const DeviceContext32& ctx, uint32_t record_rows,
                                       size_t items, hipStream_t stream);
hipError_t launch_ntt32_key_mac_inverse(const uint32_t* spread, const uint32_t* key, uint32_t* out,
                                        const DeviceContext32& ks_ctx, uint32_t L, uint32_t top_rows, size_t polys,
                                        hipStream_t stream);
// ... and the key switch's last step applied as the rows r < L are stored (launch_ntt_key_mac_inverse_finish's 4-byte twin)
hipError_t launch_ntt32_key_mac_inverse_finish(const uint32_t* spread, const uint32_t* key, uint32_t* prod,
                                               const uint32_t* ct_base, size_t ct_stride, uint32_t* out,
                                               const DeviceContext32& ks_ctx, uint32_t L, uint32_t top_rows, size_t polys,
                                               uint32_t added_polys, hipStream_t stream);
// scalars: device array of L (scalar, 64-bit Shoup factor) pairs, only for MulScalar
hipError_t launch_elementwise32(ElementwiseOp op, uint32_t* lhs, const uint32_t* rhs, const uint64_t* scalars,
                                const DeviceContext32& ctx, size_t rows, hipStream_t stream);
// packed [UInt32] words <-> the zero-extended 8-byte words of the Bfv<UInt32> scheme layer (16-byte aligned slabs)
hipError_t launch_widen_words(const uint32_t* in, uint64_t* out, size_t words, hipStream_t stream);
hipError_t launch_stream_copy(const uint64_t* in, uint64_t* out, size_t words, bool non_temporal, hipStream_t stream);
// `records` runs of record_words words from runs src_stride words apart to runs dst_stride apart (record_words a multiple of 2048,
// 16-byte aligned slabs, even strides; hipErrorInvalidValue otherwise)
hipError_t launch_copy_records(const uint64_t* in, size_t src_stride, uint64_t* out, size_t dst_stride, size_t record_words,
                               size_t records, hipStream_t stream);
hipError_t launch_narrow_words(const uint64_t* in, uint32_t* out, size_t words, hipStream_t stream);
hipError_t launch_divide_and_round_q_last32(const uint32_t* in, uint32_t* out, const DeviceContext32& ctx,
                                            uint32_t moduli_count, size_t polys, hipStream_t stream);

// seeded_kernels.hip: out[b] = PolyRq.random(context, NistAes128Ctr(seed: seeds[b])), seeds [batch][32] bytes
// scratch: seeded_uniform_scratch_bytes(ctx, batch) bytes (the per-chunk round keys of every seed's re-key chain)
size_t seeded_uniform_scratch_bytes(const DeviceContext& ctx, size_t batch);
hipError_t launch_seeded_uniform(const uint8_t* seeds, uint64_t* out, const DeviceContext& ctx, size_t batch,
                                 void* scratch, hipStream_t stream);

// wire format of a polynomial: per residue row, the serialized bit width and the byte offset of the row
constexpr uint32_t kMaxSerializedRows = 64;
struct SerializeLayout {
    uint32_t rows;
    uint32_t width[kMaxSerializedRows];            // ceilLog2(q_r) - skipLSBs
    uint64_t byte_offset[kMaxSerializedRows + 1];  // prefix sums of ceil(N width / 8); [rows] = bytes per polynomial
};
hipError_t launch_serialize(const uint64_t* slab, uint8_t* bytes, const SerializeLayout& layout, uint32_t log_degree,
                            uint32_t skip_lsbs, size_t batch, hipStream_t stream);
hipError_t launch_deserialize(const uint8_t* bytes, uint64_t* slab, const SerializeLayout& layout, uint32_t log_degree,
                              uint32_t skip_lsbs, size_t bytes_per_poly, size_t batch, hipStream_t stream);

// ---- galois_kernels.hip (in and out must not alias) ------------------------------------------------------------
// f(x) -> f(x^g) on Coeff rows; `inverse_element` = g^-1 mod 2N
// (W: uint64_t or uint32_t slabs)
template <typename W>
hipError_t launch_galois_coeff(const W* in, W* out, const DeviceContext& ctx, uint32_t inverse_element, size_t rows,
                               hipStream_t stream);
// f(x) -> f(x^g) on Eval (bit-reversed) rows
hipError_t launch_galois_eval(const uint64_t* in, uint64_t* out, const DeviceContext& ctx, uint32_t element,
                              size_t rows, hipStream_t stream);
// f(x) x^shift mod (x^N + 1), 0 <= shift < 2N
// PirUtil.expand, one level: children of `batch` parents interleaved into next (c1 + parent, (parent - c1) x^-shift')
hipError_t launch_expand_step(const uint64_t* parents, const uint64_t* c1, uint64_t* next, const DeviceContext& ctx,
                              uint32_t shift, size_t batch, hipStream_t stream);
// dst[table[2k+1] >> 1] = src[table[2k]] (doubled mod q when table[2k+1] & 1), whole ciphertexts [2][L][N]
// for `queries` expansions of one shape: query q moves src + q * src_stride -> dst + q * dst_stride (in ciphertexts)
hipError_t launch_expand_move(const uint64_t* src, uint64_t* dst, const uint32_t* table, const DeviceContext& ctx,
                              size_t count, size_t queries, size_t src_stride, size_t dst_stride, hipStream_t stream);
hipError_t launch_multiply_power_of_x(const uint64_t* in, uint64_t* out, const DeviceContext& ctx, uint32_t shift,
                                      size_t rows, hipStream_t stream);
// plaintext [batch][N] mod t -> centered lift into every row of [batch][L][N] (L = ctx.moduli_count)
template <typename W>
hipError_t launch_plaintext_lift(const W* plaintext, W* out, const DeviceContext& ctx, uint64_t t, size_t batch,
                                 hipStream_t stream);
// rows [words] over q0, Coeff form: undo the centered lift in place
template <typename W>
hipError_t launch_plaintext_unlift(W* rows, uint64_t q0, uint64_t t, size_t words, hipStream_t stream);
// residue row 0 of every polynomial: [batch][L][N] -> [batch][N]
template <typename W>
hipError_t launch_first_rows(const W* in, W* out, const DeviceContext& ctx, size_t batch, hipStream_t stream);

}  // namespace heamd
