// rns_kernels.hpp -- launchers of the BEHZ / tensor / key-switching kernels (asynchronous on `stream`).
// W is the slab's word type: uint64_t (Bfv<UInt64>) or uint32_t (Bfv<UInt32>, every modulus <= 2^30 - 1); both are
// instantiated in rns_kernels.hip.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "bfv_context.hpp"
#include "device_context.hpp"

namespace heamd {

uint32_t rns_max_supported_moduli();
// in [polys][L][N] -> out [polys][2L+1][N]
template <typename W>
hipError_t launch_lift_q_to_qbsk(const W* in, W* out, const RnsToolDevice& tool, size_t polys, hipStream_t stream);
// Strided form: item i's `polys_per_item` polynomials are read at in + i*in_item_stride (+ c*L*N) and written at
// out + out_offset + i*out_item_stride (+ c*(2L+1)*N); strides and offset in words.  store_input = false leaves rows
// [0, L) of every output polynomial (the copy of the input) unwritten: the transform that follows reads the input itself
// (kernels.hpp launch_ntt_lifted_forward).  lazy_output = true allows the Bsk rows to be left in [0, 5p) (8-byte slabs with the
// one-word-quotient reduction; other shapes store canonical words all the same): only for a consumer that takes such words
// (kernels.hpp behz_lifted_rows_may_be_lazy).
template <typename W>
hipError_t launch_lift_q_to_qbsk_strided(const W* in, W* out, const RnsToolDevice& tool, size_t items,
                                         size_t polys_per_item, size_t in_item_stride, size_t out_item_stride,
                                         size_t out_offset, hipStream_t stream, bool store_input = true, bool lazy_output = false);
// Two operands in ONE launch (the two ciphertexts of every ct x ct pair): `first` as above at out + first_out_offset, `second`
// with the same strides at out + second_out_offset.
template <typename W>
hipError_t launch_lift_pair_q_to_qbsk_strided(const W* first, const W* second, W* out, const RnsToolDevice& tool, size_t items,
                                              size_t polys_per_item, size_t in_item_stride, size_t out_item_stride,
                                              size_t first_out_offset, size_t second_out_offset, hipStream_t stream,
                                              bool store_input = true, bool lazy_output = false);
// in [polys][2L+1][N] -> out [polys][L][N]
template <typename W>
hipError_t launch_floor_qbsk_to_q(const W* in, W* out, const RnsToolDevice& tool, size_t polys, hipStream_t stream);
// in [items][4][rows][N] -> out [items][3][rows][N]
template <typename W>
hipError_t launch_tensor(const W* in, W* out, const DeviceContext& qbsk, size_t items, hipStream_t stream);
// in [count][4][rows][N] -> out [3][rows][N]
template <typename W>
hipError_t launch_tensor_accumulate(const W* in, W* out, const DeviceContext& qbsk, size_t count, uint64_t max_lazy,
                                    hipStream_t stream);
// The shared-lhs sums below on the carry-counting accumulator (8-byte words, degree >= 256): `cadence` terms between folds from
// tensor_sums_cadence (0: hipErrorNotSupported, nothing launched).  rhs_q != nullptr: rows [0, q_rows) of the right-hand
// polynomials come from rhs_q [items][count][2][q_rows][N] instead of the lifted records.
uint64_t tensor_sums_cadence(const uint64_t* moduli, uint32_t count);
hipError_t launch_tensor_accumulate_shared_sums(const uint64_t* lhs, const uint64_t* rhs, const uint64_t* rhs_q, uint32_t q_rows,
                                                uint64_t* out, const DeviceContext& qbsk, size_t count, size_t items,
                                                uint64_t cadence, hipStream_t stream);
// `items` sums that share lhs: lhs [count][2][rows][N], rhs [items][count][2][rows][N] -> out [items][3][rows][N]
template <typename W>
hipError_t launch_tensor_accumulate_shared(const W* lhs, const W* rhs, W* out, const DeviceContext& qbsk, size_t count,
                                           size_t items, uint64_t max_lazy, hipStream_t stream);
template <typename W>
hipError_t launch_key_switch_spread(const W* target_base, size_t target_stride, W* out, const DeviceContext& ks,
                                    uint32_t L, size_t polys, hipStream_t stream);
template <typename W>
hipError_t launch_key_switch_mac(const W* spread, const W* key, W* out, const DeviceContext& ks, uint32_t L,
                                 uint32_t top_rows, size_t polys, hipStream_t stream);
// scaleAndRound: in [polys][L][N] -> out [polys][N]; final_scale = Shoup pair of (gamma^-1 scalingFactor) mod t
template <typename W>
hipError_t launch_scale_and_round(const W* in, W* out, const RnsToolDevice& tool, U64x2 final_scale, size_t polys,
                                  hipStream_t stream);
// Bfv+Encrypt.swift:75-140 plaintextTranslate: ct [batch][poly_count][L][N] (c0 only) +-= plaintexts [batch][N] (< t)
template <typename W>
hipError_t launch_plaintext_translate(W* ct, const W* plaintexts, const RnsToolDevice& tool, uint32_t poly_count,
                                      bool subtract, size_t batch, hipStream_t stream);
// out[poly][c] = (c < added_polys ? ct[poly][c] : 0) + divideAndRoundQLast(prod[poly][c])
template <typename W>
hipError_t launch_key_switch_finish(const W* prod, const W* ct_base, size_t ct_stride, W* out, const DeviceContext& ks,
                                    uint32_t L, size_t polys, uint32_t added_polys, hipStream_t stream);
// Where the children of an expand step go when they are leaves of the expansion (PirUtil.swift:262-299): parent p of
// group g (= one query's parents, group_size of them) has children 2p and 2p + 1 in its level; table entries
// (node, slot << 1 | doubled) of that level in node order send child i to ciphertext g * group_stride + slot of `out`,
// added to itself when `doubled`.  table = nullptr: children 2 poly, 2 poly + 1 of `out`.
struct ExpandTargets {
    const uint32_t* table;
    size_t group_size, group_stride;
};
// The same for Bfv.applyGalois with `ct` the ciphertext BEFORE the automorphism (Bfv.swift:190-196):
// c' = (galois(ct.c0) + update0, update1), galois_inverse = g^-1 mod 2N.  expand_shift = 0: out [polys][2][L][N] = c';
// otherwise one PirUtil.expand step (PirUtil.swift:204-236): out [2 polys][2][L][N] = the children
// ct + c' and (ct - c') x^expand_shift, interleaved.  out must not alias ct.  own_base (expand steps only; nullptr =
// ct_base): the ciphertexts the children are formed with when they differ from the ones the key switch rotates -- a level
// whose element is reached by repeated application (PirUtil.swift:221-231): children own + c' and (own - c') x^shift.
template <typename W>
hipError_t launch_galois_finish(const W* prod, const W* ct_base, size_t ct_stride, W* out, const DeviceContext& ks,
                                uint32_t L, size_t polys, uint32_t galois_inverse, uint32_t expand_shift,
                                const ExpandTargets& targets, hipStream_t stream, const W* own_base = nullptr);

}  // namespace heamd
