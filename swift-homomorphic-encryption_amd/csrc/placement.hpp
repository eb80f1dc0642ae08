// placement.hpp -- XCD-aware workgroup placement shared by the kernels whose workgroups re-read each other's inputs.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace heamd {

// Kernels whose workgroups come in sets of `replicas` that read the SAME source words (the L + 1 key-switching rows
// spread from one ciphertext row, the three tensor-product polynomials of one [Q, Bsk] row, the two key polynomials of
// one spread row, the column groups of a ct x pt inner product that stream the same ciphertext tile): workgroup b is dispatched to XCD b % 8, so the members of a set take dispatch slots s, s + 1, ... of ONE
// XCD -- the first one's read fills that XCD's L2 and the others hit there instead of going back to HBM.  Sets beyond
// the last multiple of 8 keep the plain order.  Performance only: any placement computes the same thing.
__device__ __forceinline__ void locate_replica(uint32_t block, uint32_t sets, uint32_t replicas, uint32_t& set,
                                               uint32_t& replica) {
    constexpr uint32_t kXcds = 8;
    const uint32_t full = sets & ~(kXcds - 1);
    if (block < full * replicas) {
        const uint32_t slot = block / kXcds, xcd = block % kXcds, round = slot / replicas;
        replica = slot - round * replicas;
        set = round * kXcds + xcd;
    } else {
        const uint32_t rest = block - full * replicas, q = rest / replicas;
        set = full + q;
        replica = rest - q * replicas;
    }
    // the divisions by a run-time count go through the vector ALU: hand the (wave-uniform) results back to scalar
    // registers explicitly, so that everything derived from them (moduli, table bases, row offsets) stays scalar
    set = __builtin_amdgcn_readfirstlane(set);
    replica = __builtin_amdgcn_readfirstlane(replica);
}

}  // namespace heamd
