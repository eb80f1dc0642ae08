// arith_probe.hip -- TEST INFRASTRUCTURE (tests/test_gpu_device_math.py): runs the device-side products of
// csrc/device_math.hpp on caller-supplied operands, one lane per operand pair, so that their stated ranges and congruences
// can be checked against Python integers.  Not part of libhe_amd.so; built by __graft_entry__.build() into
// tests/device_probe/libarith_probe.so.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "device_math.hpp"
#include "host_math.hpp"

namespace {

using namespace heamd;

// a 64-bit word the compiler knows to be wave-uniform (it then keeps it in scalar registers)
__device__ __forceinline__ uint64_t uniform_word(uint64_t v) {
    const uint32_t lo = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(v))));
    const uint32_t hi = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<uint32_t>(v >> 32))));
    return lo | (static_cast<uint64_t>(hi) << 32);
}

// kind 0: split_mul_add<false, false>(0, y, ...) on an unsigned word; kind 1: split_mul_signed<false> on a signed word
// (the constant's second word in signed limbs, as the inverse tables hold it); kinds 2 / 3: the same with the constant's
// words wave-uniform (every lane of a launch then takes constant 0)
template <int KIND>
__global__ void product_kernel(const uint64_t* __restrict__ operand, const uint64_t* __restrict__ w,
                               const uint64_t* __restrict__ wt, const uint64_t* __restrict__ factors, uint64_t p,
                               uint64_t* __restrict__ out, size_t count) {
    const size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    if (i >= count) return;
    const uint64_t neg_2p = 0 - 2 * p;
    if constexpr (KIND == 0) {
        out[i] = split_mul_add<false, false>(0, operand[i], w[i], wt[i], factors[i], neg_2p);
    } else if constexpr (KIND == 1) {
        out[i] = split_mul_signed<false>(operand[i], w[i], wt[i], factors[i], neg_2p, p + (p << 31));
    } else {
        const uint64_t w0 = uniform_word(w[0]), t0 = uniform_word(wt[0]), f0 = uniform_word(factors[0]);
        if constexpr (KIND == 2) {
            out[i] = split_mul_add<true, false>(0, operand[i], w0, t0, f0, neg_2p);
        } else {
            uint64_t bias;
            asm volatile("v_mov_b64 %0, %1" : "=v"(bias) : "s"(p + (p << 31)));
            out[i] = split_mul_signed<true>(operand[i], w0, t0, f0, neg_2p, bias);
        }
    }
}

// kinds 4 / 5: fold_mul<false, false> / fold_mul<true, false> (p = 2^b - d: kModeFoldLazy and kModeFoldMinus); kinds 6 / 7:
// fold_mul<false, true> / fold_mul<true, true> (p = 2^60 + e: kModeFoldPlus).  The uniform kinds take constant 0 for every lane.
template <bool UNIFORM, bool PLUS>
__global__ void fold_kernel(const uint64_t* __restrict__ operand, const uint64_t* __restrict__ w, const uint64_t* __restrict__ wt,
                            uint64_t p, uint64_t* __restrict__ out, size_t count) {
    const size_t i = blockIdx.x * size_t(blockDim.x) + threadIdx.x;
    if (i >= count) return;
    const FoldConstants fc = fold_constants<PLUS>(uniform_word(p));
    if constexpr (UNIFORM) out[i] = fold_mul<true, PLUS>(operand[i], uniform_word(w[0]), uniform_word(wt[0]), fc);
    else out[i] = fold_mul<false, PLUS>(operand[i], w[i], wt[i], fc);
}

}  // namespace

// operand[count], constant[count] -> out[count] = the shift-folded product (kinds 4..7 above).  Host pointers.
extern "C" int arith_probe_fold_product(int kind, uint64_t p, const uint64_t* operand, const uint64_t* constant, size_t count,
                                        uint64_t* out) {
    if (kind < 4 || kind > 7 || count == 0) return -1;
    uint64_t* host = static_cast<uint64_t*>(malloc(count * sizeof(uint64_t)));
    for (size_t i = 0; i < count; ++i) host[i] = heamd::split_shifted(constant[i], p);
    uint64_t* device = nullptr;
    hipError_t e = hipMalloc(&device, 4 * count * sizeof(uint64_t));
    if (e != hipSuccess) { free(host); return int(e); }
    uint64_t *d_operand = device, *d_w = device + count, *d_wt = device + 2 * count, *d_out = device + 3 * count;
    e = hipMemcpy(d_operand, operand, count * sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_w, constant, count * sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_wt, host, count * sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const dim3 grid(static_cast<unsigned>((count + 255) / 256)), block(256);
        switch (kind) {
            case 4: hipLaunchKernelGGL((fold_kernel<false, false>), grid, block, 0, 0, d_operand, d_w, d_wt, p, d_out, count); break;
            case 5: hipLaunchKernelGGL((fold_kernel<true, false>), grid, block, 0, 0, d_operand, d_w, d_wt, p, d_out, count); break;
            case 6: hipLaunchKernelGGL((fold_kernel<false, true>), grid, block, 0, 0, d_operand, d_w, d_wt, p, d_out, count); break;
            default: hipLaunchKernelGGL((fold_kernel<true, true>), grid, block, 0, 0, d_operand, d_w, d_wt, p, d_out, count); break;
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, d_out, count * sizeof(uint64_t), hipMemcpyDeviceToHost);
    (void)hipFree(device);
    free(host);
    return int(e);
}

// operand[count], constant[count] (constants below p; kinds 2 / 3 use constant[0] for every operand) -> out[count].
// Host pointers.  Returns 0 or the hipError_t that stopped it.
extern "C" int arith_probe_split_product(int kind, uint64_t p, const uint64_t* operand, const uint64_t* constant, size_t count,
                                         uint64_t* out) {
    if (kind < 0 || kind > 3 || count == 0) return -1;
    const bool signed_limbs = (kind & 1) != 0;
    uint64_t* host = static_cast<uint64_t*>(malloc(3 * count * sizeof(uint64_t)));
    for (size_t i = 0; i < count; ++i) {
        uint64_t shifted = heamd::split_shifted(constant[i], p);
        if (signed_limbs) shifted += (shifted & 0x80000000ull) << 1;  // as PolyContext::upload stores the inverse tables
        host[i] = shifted;
        host[count + i] = heamd::split_factors(constant[i], p);
    }
    uint64_t* device = nullptr;
    hipError_t e = hipMalloc(&device, 5 * count * sizeof(uint64_t));
    if (e != hipSuccess) { free(host); return int(e); }
    uint64_t *d_operand = device, *d_w = device + count, *d_wt = device + 2 * count, *d_f = device + 3 * count,
             *d_out = device + 4 * count;
    e = hipMemcpy(d_operand, operand, count * sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_w, constant, count * sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_wt, host, count * sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(d_f, host + count, count * sizeof(uint64_t), hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        const dim3 grid(static_cast<unsigned>((count + 255) / 256)), block(256);
        switch (kind) {
            case 0: hipLaunchKernelGGL(product_kernel<0>, grid, block, 0, 0, d_operand, d_w, d_wt, d_f, p, d_out, count); break;
            case 1: hipLaunchKernelGGL(product_kernel<1>, grid, block, 0, 0, d_operand, d_w, d_wt, d_f, p, d_out, count); break;
            case 2: hipLaunchKernelGGL(product_kernel<2>, grid, block, 0, 0, d_operand, d_w, d_wt, d_f, p, d_out, count); break;
            default: hipLaunchKernelGGL(product_kernel<3>, grid, block, 0, 0, d_operand, d_w, d_wt, d_f, p, d_out, count); break;
        }
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, d_out, count * sizeof(uint64_t), hipMemcpyDeviceToHost);
    (void)hipFree(device);
    free(host);
    return int(e);
}
