// power_probe.hip -- the two halves of the forward NTT on their own, each for a given number of seconds, so that rocm-smi
// can be sampled beside them (bench_tools/power_probe.py): register-resident limb-wise butterflies at full occupancy (no
// memory), and the streaming copy of a 1 GiB slab at the transform's access width (8 bytes per lane, non-temporal).
//   hipcc --offload-arch=gfx950 -O3 -o power_probe power_probe.hip ;  ./power_probe butterflies|fold|copy SECONDS
// (butterflies: the limb-wise products of rounds 2-4; fold: the shift-folded products the transforms run on since round 5)
// Prints the sustained rate: T butterflies/s, or TB/s read + written.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "../swift-homomorphic-encryption_amd/csrc/device_math.hpp"
using namespace heamd;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void __launch_bounds__(256) butterflies(uint64_t* out, uint64_t p, int iters) {
    uint64_t v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = (0x9E3779B97F4A7C15ull * (r + 1) + threadIdx.x * 977u + blockIdx.x) % p;
    const uint64_t w = (0xD1B54A32D192ED03ull * (threadIdx.x + 1)) % p;
    const uint64_t wt = static_cast<uint64_t>((static_cast<unsigned __int128>(w) << 32) % p);
    const uint64_t f = static_cast<uint64_t>((static_cast<unsigned __int128>(w) << 32) / (static_cast<unsigned __int128>(p) * 2));
    const uint64_t ft = static_cast<uint64_t>((static_cast<unsigned __int128>(wt) << 32) / (static_cast<unsigned __int128>(p) * 2));
    const uint64_t factors = f | (ft << 32), neg_2p = 0 - 2 * p;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int stride = 8 >> j;
#pragma unroll
            for (int base = 0; base < 16; base += 2 * stride)
#pragma unroll
                for (int o = 0; o < stride; ++o) {
                    const uint64_t x = v[base + o], y = v[base + o + stride];
                    const uint64_t sum = split_mul_add<false, true>(x, y, w, wt, factors, neg_2p);
                    v[base + o] = sum;
                    v[base + o + stride] = ((x << 1) + 8 * p) - sum;
                }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] &= 0x00FFFFFFFFFFFFFFull;
    }
    uint64_t sum = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum ^= v[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

// the production arithmetic since round 5: the product folded by a shift (fold_mul, 5 multiply-adds, no factors), sums
// never brought back (the mask below stands in for the transform's bounded growth)
__global__ void __launch_bounds__(256) butterflies_fold(uint64_t* out, uint64_t p, int iters) {
    uint64_t v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = (0x9E3779B97F4A7C15ull * (r + 1) + threadIdx.x * 977u + blockIdx.x) % p;
    const uint64_t w = (0xD1B54A32D192ED03ull * (threadIdx.x + 1)) % p;
    const uint64_t wt = static_cast<uint64_t>((static_cast<unsigned __int128>(w) << 32) % p);
    const FoldConstants fc = fold_constants<false>(p);
    const uint64_t half_bound = 8 * p;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int stride = 8 >> j;
#pragma unroll
            for (int base = 0; base < 16; base += 2 * stride)
#pragma unroll
                for (int o = 0; o < stride; ++o) {
                    const uint64_t x = v[base + o], y = v[base + o + stride];
                    const uint64_t r = fold_mul<false, false>(y, w, wt, fc);
                    v[base + o] = x + r;
                    v[base + o + stride] = x + half_bound - r;
                }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] &= 0x00FFFFFFFFFFFFFFull;
    }
    uint64_t sum = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum ^= v[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

__global__ void __launch_bounds__(256) copy8(const uint64_t* __restrict__ in, uint64_t* __restrict__ out, size_t words) {
    for (size_t i = blockIdx.x * size_t(256) + threadIdx.x; i < words; i += size_t(gridDim.x) * 256)
        __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}

int main(int argc, char** argv) {
    if (argc < 3) return 2;
    const double seconds = atof(argv[2]);
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const auto t0 = std::chrono::steady_clock::now();
    auto elapsed = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(); };
    if (strcmp(argv[1], "butterflies") == 0 || strcmp(argv[1], "fold") == 0) {
        const bool fold = strcmp(argv[1], "fold") == 0;
        const int blocks = prop.multiProcessorCount * 8, iters = 2000;
        uint64_t* out;
        CHECK(hipMalloc(&out, size_t(blocks) * 256 * 8));
        double done = 0;
        while (elapsed() < seconds) {
            for (int k = 0; k < 8; ++k) {
                if (fold) butterflies_fold<<<blocks, 256>>>(out, 36028797018652673ull, iters);
                else butterflies<<<blocks, 256>>>(out, 36028797018652673ull, iters);
            }
            CHECK(hipDeviceSynchronize());
            done += 8.0 * blocks * 256 * iters * 32;
        }
        printf("butterflies %.3f T/s\n", done / elapsed() / 1e12);
    } else {
        const size_t words = size_t(1) << 27;  // 1 GiB
        uint64_t *in, *out;
        CHECK(hipMalloc(&in, words * 8));
        CHECK(hipMalloc(&out, words * 8));
        CHECK(hipMemset(in, 1, words * 8));
        double bytes = 0;
        while (elapsed() < seconds) {
            for (int k = 0; k < 16; ++k) copy8<<<prop.multiProcessorCount * 32, 256>>>(in, out, words);
            CHECK(hipDeviceSynchronize());
            bytes += 16.0 * 2 * words * 8;
        }
        printf("copy %.3f TB/s\n", bytes / elapsed() / 1e12);
    }
    return 0;
}
