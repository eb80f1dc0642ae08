// rns_kernels.hpp -- launchers of the BEHZ / tensor / key-switching kernels (asynchronous on `stream`).
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "bfv_context.hpp"
#include "device_context.hpp"

namespace heamd {

uint32_t rns_max_supported_moduli();
// in [polys][L][N] -> out [polys][2L+1][N]
hipError_t launch_lift_q_to_qbsk(const uint64_t* in, uint64_t* out, const RnsToolDevice& tool, size_t polys,
                                 hipStream_t stream);
// Strided form: item i's `polys_per_item` polynomials are read at in + i*in_item_stride (+ c*L*N) and written at
// out + out_offset + i*out_item_stride (+ c*(2L+1)*N); strides and offset in words.
hipError_t launch_lift_q_to_qbsk_strided(const uint64_t* in, uint64_t* out, const RnsToolDevice& tool, size_t items,
                                         size_t polys_per_item, size_t in_item_stride, size_t out_item_stride,
                                         size_t out_offset, hipStream_t stream);
// in [polys][2L+1][N] -> out [polys][L][N]
hipError_t launch_floor_qbsk_to_q(const uint64_t* in, uint64_t* out, const RnsToolDevice& tool, size_t polys,
                                  hipStream_t stream);
// in [items][4][rows][N] -> out [items][3][rows][N]
hipError_t launch_tensor(const uint64_t* in, uint64_t* out, const DeviceContext& qbsk, size_t items,
                         hipStream_t stream);
// in [count][4][rows][N] -> out [3][rows][N]
hipError_t launch_tensor_accumulate(const uint64_t* in, uint64_t* out, const DeviceContext& qbsk, size_t count,
                                    uint64_t max_lazy, hipStream_t stream);
hipError_t launch_key_switch_spread(const uint64_t* target_base, size_t target_stride, uint64_t* out,
                                    const DeviceContext& ks, uint32_t L, size_t polys, hipStream_t stream);
hipError_t launch_key_switch_mac(const uint64_t* spread, const uint64_t* key, uint64_t* out, const DeviceContext& ks,
                                 uint32_t L, uint32_t top_rows, size_t polys, hipStream_t stream);
// scaleAndRound: in [polys][L][N] -> out [polys][N]; final_scale = Shoup pair of (gamma^-1 scalingFactor) mod t
hipError_t launch_scale_and_round(const uint64_t* in, uint64_t* out, const RnsToolDevice& tool, U64x2 final_scale,
                                  size_t polys, hipStream_t stream);
// out[poly][c] = (c < added_polys ? ct[poly][c] : 0) + divideAndRoundQLast(prod[poly][c])
hipError_t launch_key_switch_finish(const uint64_t* prod, const uint64_t* ct_base, size_t ct_stride, uint64_t* out,
                                    const DeviceContext& ks, uint32_t L, size_t polys, uint32_t added_polys,
                                    hipStream_t stream);

}  // namespace heamd
