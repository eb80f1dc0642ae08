// butterfly_probe.hip -- register-resident forward butterflies of three arithmetic schedules at full occupancy, no memory:
// what the issue rate under the power cap gives each of them (round 3: is a fold-by-shift butterfly for the 60 / 61-bit
// moduli worth building?).  hipcc --offload-arch=gfx950 -O3 -o butterfly_probe butterfly_probe.hip
//   split   limb-wise Shoup, fold-free (production, p < 2^55)            device_math.hpp split_mul_add
//   approx  Harvey with the 3-multiply quotient, one csub (p < 2^61)     device_math.hpp shoup_lazy4_fma
//   fold    p = 2^b - d: w y = b0 w + b1 wt as an exact 96-bit sum V, then (V mod 2^(b+2)) + (V >> (b+2)) 4d; one csub
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "../swift-homomorphic-encryption_amd/csrc/device_math.hpp"
using namespace heamd;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ uint64_t csub_sel(uint64_t x, uint64_t m) { return x >= m ? x - m : x; }

enum { SPLIT, APPROX, FOLD, FOLD_PLUS };
template <int VARIANT>
__global__ void __launch_bounds__(256) bfly(uint64_t* out, uint64_t p, int iters, uint64_t* check) {
    uint64_t v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = (0x9E3779B97F4A7C15ull * (r + 1) + threadIdx.x * 977u + blockIdx.x) % p;
    // one twiddle per lane (constants derived on the host side of the kernel: here by 128-bit arithmetic once)
    const uint64_t w = (0xD1B54A32D192ED03ull * (threadIdx.x + 1)) % p;
    const uint64_t wt = static_cast<uint64_t>((static_cast<unsigned __int128>(w) << 32) % p);
    const uint64_t wf = static_cast<uint64_t>((static_cast<unsigned __int128>(w) << 64) / p);
    const uint64_t f = static_cast<uint64_t>((static_cast<unsigned __int128>(w) << 32) / (static_cast<unsigned __int128>(p) * 2));
    const uint64_t ft = static_cast<uint64_t>((static_cast<unsigned __int128>(wt) << 32) / (static_cast<unsigned __int128>(p) * 2));
    const uint64_t factors = f | (ft << 32);
    constexpr bool PLUS = VARIANT == FOLD_PLUS;
    const FoldConstants fc = fold_constants<PLUS>(p);
    const uint64_t neg_p = 0 - p, neg_2p = 0 - 2 * p;
    if (check != nullptr && (VARIANT == FOLD || VARIANT == FOLD_PLUS)) {  // the formula against 128-bit arithmetic
        uint64_t bad = 0;
        const uint64_t uw = (0x9E3779B97F4A7C15ull * (blockIdx.x + 1)) % p;  // wave-uniform
        const uint64_t uwt = static_cast<uint64_t>((static_cast<unsigned __int128>(uw) << 32) % p);
        for (int r = 0; r < 16; ++r) {
            uint64_t y = v[r] * 0xFFFFFFFFFFFFull + r;  // any 64-bit word
            if (r == 3) y = ~0ull;
            if (r == 4) y = 0;
            if (r == 5) y = 0xFFFFFFFF00000000ull;
            if (r == 6) y = 0x00000000FFFFFFFFull;
            const uint64_t got = fold_mul<false, PLUS>(y, w, wt, fc);
            const uint64_t want = static_cast<uint64_t>((static_cast<unsigned __int128>(y) * w) % p);
            if (got % p != want || got >= 6 * p) bad |= 1ull << r;
            const uint64_t got_v = fold_mul<false, PLUS>(y, uw, uwt, fc);
            if (got_v % p != static_cast<uint64_t>((static_cast<unsigned __int128>(y) * uw) % p)) bad |= 1ull << (r + 32);
            const uint64_t got_u = fold_mul<true, PLUS>(y, uw, uwt, fc);
            if (got_u != got_v) bad |= 1ull << (r + 48);
            const uint64_t want_u = static_cast<uint64_t>((static_cast<unsigned __int128>(y) * uw) % p);
            if (got_u % p != want_u || got_u >= 6 * p) bad |= 1ull << (r + 16);
        }
        check[blockIdx.x * blockDim.x + threadIdx.x] = bad;
    }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int stride = 8 >> j;
#pragma unroll
            for (int base = 0; base < 16; base += 2 * stride)
#pragma unroll
                for (int o = 0; o < stride; ++o) {
                    uint64_t x = v[base + o];
                    const uint64_t y = v[base + o + stride];
                    if constexpr (VARIANT == SPLIT) {
                        const uint64_t sum = split_mul_add<false, true>(x, y, w, wt, factors, neg_2p);
                        v[base + o] = sum;
                        v[base + o + stride] = ((x << 1) + 8 * p) - sum;
                    } else if constexpr (VARIANT == APPROX) {
                        x = csub_sel(x, 4 * p);
                        const uint64_t sum = shoup_lazy4_fma<false>(x, y, w, wf, neg_p);
                        v[base + o] = sum;
                        v[base + o + stride] = ((x << 1) + 4 * p) - sum;
                    } else {
                        x = csub_sel(x, 8 * p);
                        const uint64_t r = fold_mul<false, PLUS>(y, w, wt, fc);
                        v[base + o] = x + r;
                        v[base + o + stride] = x + 6 * p - r;
                    }
                }
        }
        if constexpr (VARIANT == SPLIT) {  // the fold-free schedule's words must not run away in an endless loop
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] &= 0x00FFFFFFFFFFFFFFull;
        }
    }
    uint64_t sum = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) sum ^= v[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
}

template <int VARIANT>
void run(const char* name, uint64_t p) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * 8, iters = 20000;  // 8 waves per SIMD
    uint64_t *out, *check;
    CHECK(hipMalloc(&out, size_t(blocks) * 256 * 8));
    CHECK(hipMalloc(&check, size_t(blocks) * 256 * 8));
    bfly<VARIANT><<<blocks, 256>>>(out, p, 200, check);
    CHECK(hipDeviceSynchronize());
    if (VARIANT == FOLD || VARIANT == FOLD_PLUS) {
        uint64_t* host = (uint64_t*)malloc(size_t(blocks) * 256 * 8);
        CHECK(hipMemcpy(host, check, size_t(blocks) * 256 * 8, hipMemcpyDeviceToHost));
        size_t bad = 0;
        uint64_t any = 0;
        for (size_t i = 0; i < size_t(blocks) * 256; ++i) { bad += host[i] != 0; any |= host[i]; }
        printf("fold formula check: %zu lanes disagree with 128-bit arithmetic (cases %016llx)\n", bad, (unsigned long long)any);
        free(host);
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    CHECK(hipEventRecord(e0));
    bfly<VARIANT><<<blocks, 256>>>(out, p, iters, nullptr);
    CHECK(hipEventRecord(e1));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    const double butterflies = double(blocks) * 256 * iters * 32;
    printf("%-8s p=%llu  %.1f ms  %.3f T butterflies/s\n", name, (unsigned long long)p, ms, butterflies / (ms * 1e-3) / 1e12);
    CHECK(hipFree(out));
    CHECK(hipFree(check));
}

int main() {
    const uint64_t p55 = 36028797018652673ull;      // 2^55 - 311295
    const uint64_t p60 = 1152921504606584833ull;    // 2^60 - 262143 (NTT-friendly for N = 8192: 1 mod 2^14? only the size matters here)
    run<SPLIT>("split", p55);
    run<APPROX>("approx", p60);
    run<FOLD>("fold", p60);
    run<FOLD>("fold55", p55);
    run<FOLD>("fold57", 144115188075593729ull);  // 2^57 - 262143
    run<APPROX>("approx", 1152921504606994433ull);   // the first BEHZ auxiliary prime for N = 8192: 2^60 + 147457
    run<FOLD_PLUS>("fold+", 1152921504606994433ull);
    run<FOLD_PLUS>("fold+", 1152921504607518721ull);  // the fifth: 2^60 + 671745
    return 0;
}
