// kernels.hpp -- host-callable launchers of the HIP kernels (all asynchronous on `stream`).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "device_context.hpp"

namespace heamd {

enum { kNttVariantAuto = 0, kNttVariantExact = 1, kNttVariantGeneric = 2, kNttVariantWide = 3 };

// NTT of `rows` contiguous length-N rows.  Row r uses modulus index mod_base + (r % mod_period).
hipError_t launch_ntt(bool inverse, uint64_t* slab, const DeviceContext& ctx, uint32_t mod_base, uint32_t mod_period,
                      size_t rows, hipStream_t stream, int force_variant = kNttVariantAuto);
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
hipError_t launch_adding_lazy_product(const uint64_t* lhs, const uint64_t* rhs, uint64_t* acc_lo_hi,
                                      const DeviceContext& ctx, hipStream_t stream);
hipError_t launch_reduce_accumulator(const uint64_t* acc_lo_hi, uint64_t* out, const DeviceContext& ctx,
                                     hipStream_t stream);
// out[col][poly][L][N] = sum_k cts[k][poly][L][N] * pts[col][k][L][N]  (skip where present[col*count+k] == 0)
hipError_t launch_inner_product_plain(const uint64_t* cts, const uint64_t* pts, const uint8_t* present_device,
                                      uint64_t* out, const DeviceContext& ctx, uint32_t poly_count, size_t count,
                                      size_t columns, uint64_t max_lazy, hipStream_t stream);

}  // namespace heamd
