// microbench.hip -- instruction-rate and streaming-bandwidth probes for gfx950 (MI355X).
//
// Why: the NTT is bound by the issue rate of 32-bit integer multiplies (a 64-bit Shoup butterfly is ~9 of them),
// so the honest roofline for it needs the measured rates of v_mad_u64_u32 / v_mul_lo_u32 / v_mul_hi_u32 next to
// plain VALU ops.  Output: one line per probe, "name cycles_per_wave_instr_per_SIMD @waves_per_simd".
// Build: hipcc --offload-arch=gfx950 -O3 -o microbench microbench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                       \
    do {                                                                               \
        hipError_t e_ = (x);                                                           \
        if (e_ != hipSuccess) {                                                        \
            fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                                   \
        }                                                                              \
    } while (0)

constexpr int kChains = 8;     // independent dependency chains per lane
constexpr int kUnroll = 16;    // instructions per chain per loop iteration
constexpr int kIters = 2000;

enum Probe { MAD_U64_U32, MAD_I64_I32, MUL_LO_U32, MUL_HI_U32, LSHL_ADD_U64, ADD_U32, ADDCO_PAIR, CNDMASK, MAD_U32_U24, MUL_HI_U32_U24, NOT_B32, FMA_F64, CMP_GT_I64, CMP_GT_I32 };

template <int PROBE>
__global__ void __launch_bounds__(256) probe_kernel(uint64_t* out, uint32_t seed, long long* cycles) {
    uint64_t acc[kChains];
    uint32_t a = seed * 2654435761u + threadIdx.x, b = seed ^ 0x9E3779B9u;
#pragma unroll
    for (int c = 0; c < kChains; ++c) acc[c] = (uint64_t(a + c) << 32) | (b + c);
    const long long t0 = wall_clock64();
    const long long c0 = clock64();
    for (int it = 0; it < kIters; ++it) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
#pragma unroll
            for (int c = 0; c < kChains; ++c) {
                if constexpr (PROBE == MAD_U64_U32) {
                    uint64_t d, carry;
                    asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "v"(b), "v"(acc[c]));
                    acc[c] = d;
                } else if constexpr (PROBE == MAD_I64_I32) {
                    uint64_t d, carry;
                    asm volatile("v_mad_i64_i32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(carry) : "v"(a), "v"(b), "v"(acc[c]));
                    acc[c] = d;
                } else if constexpr (PROBE == MUL_LO_U32) {
                    uint32_t d;
                    asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(d) : "v"(uint32_t(acc[c])), "v"(b));
                    acc[c] = d;
                } else if constexpr (PROBE == MUL_HI_U32) {
                    uint32_t d;
                    asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(d) : "v"(uint32_t(acc[c])), "v"(b));
                    acc[c] = d | 0x80000000u;
                } else if constexpr (PROBE == LSHL_ADD_U64) {
                    uint64_t d;
                    asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(d) : "v"(acc[c]), "v"(uint64_t(b)));
                    acc[c] = d;
                } else if constexpr (PROBE == ADD_U32) {
                    uint32_t d;
                    asm volatile("v_add_u32 %0, %1, %2" : "=v"(d) : "v"(uint32_t(acc[c])), "v"(b));
                    acc[c] = d;
                } else if constexpr (PROBE == ADDCO_PAIR) {
                    uint32_t lo, hi;
                    asm volatile("v_add_co_u32 %0, vcc, %2, %4\n\tv_addc_co_u32 %1, vcc, %3, %5, vcc"
                                 : "=&v"(lo), "=v"(hi)
                                 : "v"(uint32_t(acc[c])), "v"(uint32_t(acc[c] >> 32)), "v"(a), "v"(b)
                                 : "vcc");
                    acc[c] = (uint64_t(hi) << 32) | lo;
                } else if constexpr (PROBE == CNDMASK) {
                    uint32_t d;
                    asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(d) : "v"(uint32_t(acc[c])), "v"(b) : );
                    acc[c] = d;
                } else if constexpr (PROBE == MAD_U32_U24) {
                    uint32_t d;
                    asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(d) : "v"(uint32_t(acc[c])), "v"(b), "v"(a));
                    acc[c] = d;
                } else if constexpr (PROBE == MUL_HI_U32_U24) {
                    uint32_t d;
                    asm volatile("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(d) : "v"(uint32_t(acc[c])), "v"(b));
                    acc[c] = d | 0x00800000u;
                } else if constexpr (PROBE == NOT_B32) {
                    uint32_t d;
                    asm volatile("v_not_b32 %0, %1" : "=v"(d) : "v"(uint32_t(acc[c])));
                    acc[c] = d;
                } else if constexpr (PROBE == CMP_GT_I64) {
                    asm volatile("v_cmp_gt_i64 vcc, 0, %0" : : "v"(acc[c]) : "vcc");
                } else if constexpr (PROBE == CMP_GT_I32) {
                    asm volatile("v_cmp_gt_i32 vcc, 0, %0" : : "v"(uint32_t(acc[c] >> 32)) : "vcc");
                } else if constexpr (PROBE == FMA_F64) {
                    double d;
                    asm volatile("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(__longlong_as_double(acc[c])), "v"(1.0000001), "v"(0.5));
                    acc[c] = __double_as_longlong(d);
                }
            }
        }
    }
    const long long c1 = clock64();
    const long long t1 = wall_clock64();
    uint64_t sum = 0;
#pragma unroll
    for (int c = 0; c < kChains; ++c) sum += acc[c];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0) {
        cycles[2 * blockIdx.x] = c1 - c0;
        cycles[2 * blockIdx.x + 1] = t1 - t0;
    }
}

template <int PROBE>
void run_probe(const char* name, int instrs_per_op) {
    int device_cus = 256;
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    device_cus = prop.multiProcessorCount;
    for (int waves_per_simd : {1, 2, 4, 8}) {
        const int threads = 256;  // 4 waves -> one per SIMD
        const int blocks = device_cus * waves_per_simd;
        uint64_t* out;
        long long* cycles;
        CHECK(hipMalloc(&out, size_t(blocks) * threads * 8));
        CHECK(hipMalloc(&cycles, size_t(blocks) * 16));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        probe_kernel<PROBE><<<blocks, threads>>>(out, 1, cycles);  // warm-up
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        probe_kernel<PROBE><<<blocks, threads>>>(out, 2, cycles);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<long long> h(size_t(blocks) * 2);
        CHECK(hipMemcpy(h.data(), cycles, h.size() * 8, hipMemcpyDeviceToHost));
        double mean_cycles = 0, mean_wall = 0;
        for (int i = 0; i < blocks; ++i) {
            mean_cycles += double(h[2 * i]);
            mean_wall += double(h[2 * i + 1]);
        }
        mean_cycles /= blocks;
        mean_wall /= blocks;
        const double ops_per_wave = double(kIters) * kUnroll * kChains;
        // a SIMD hosts waves_per_simd waves concurrently: cycles per wave-op per SIMD
        const double cyc_per_op = mean_cycles / (ops_per_wave * waves_per_simd);
        const double total_lane_ops = ops_per_wave * 64.0 * 4 * blocks * instrs_per_op;
        printf("%-16s waves/SIMD=%d  shader_cycles/wave_op/SIMD=%7.3f  wall_ticks=%.0f  kernel_ms=%.3f  Tlane_instr/s=%.3f\n",
               name, waves_per_simd, cyc_per_op, mean_wall, ms, total_lane_ops / (ms * 1e-3) / 1e12);
        CHECK(hipFree(out));
        CHECK(hipFree(cycles));
    }
}

// ---- butterfly probes: register-resident 5-stage passes (80 butterflies per lane per iteration), no memory ----
#include "../swift-homomorphic-encryption_amd/csrc/device_math.hpp"
using namespace heamd;

// conditional-subtract candidates
__device__ __forceinline__ uint64_t csub_select(uint64_t x, uint64_t m) { return x >= m ? x - m : x; }
__device__ __forceinline__ uint64_t csub_mask(uint64_t x, uint64_t m) {
    const uint64_t d = x - m;
    const uint64_t mask = static_cast<uint64_t>(static_cast<int64_t>(d) >> 63);
    return d + (m & mask);
}
__device__ __forceinline__ uint64_t csub_min32(uint64_t x, uint64_t m) {
    // valid when x, m < 2^63: (x - m) wraps above 2^63 iff x < m; pick the smaller by comparing high words first
    const uint64_t d = x - m;
    return static_cast<int64_t>(d) < 0 ? x : d;
}

enum BflyVariant { FWD_APPROX_SELECT, FWD_APPROX_MASK, FWD_APPROX_MIN, FWD_APPROX_NOCSUB, FWD_EXACT_SELECT, INV_APPROX_SELECT, INV_APPROX_MASK, INV_APPROX_NOCSUB, CSUB_ONLY_SELECT, CSUB_ONLY_MASK, CSUB_ONLY_MIN, MUL_ONLY_APPROX, MUL_ONLY_EXACT, FWD_HEADROOM, INV_HEADROOM, MUL_ONLY_HEADROOM };

template <int VARIANT>
__global__ void __launch_bounds__(256) bfly_kernel(uint64_t* out, uint64_t p, uint64_t seed, int iters, long long* cycles) {
    uint64_t v[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) v[r] = (seed * (r + 1) + threadIdx.x * 977u) % p;
    U64x2 tw[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        tw[k].x = (seed * 31 + k * 1234567 + threadIdx.x) % p;
        tw[k].y = static_cast<uint64_t>((static_cast<unsigned __int128>(tw[k].x) << 64) / p);
    }
    const uint64_t neg_p = opaque(0 - p);
    const uint64_t hb = 4 * p, two_p = 2 * p;
    const long long t0 = wall_clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const int stride = 16 >> j;
#pragma unroll
            for (int base = 0; base < 32; base += 2 * stride) {
                const U64x2 w = tw[(base / (2 * stride) + j) & 3];
#pragma unroll
                for (int o = 0; o < stride; ++o) {
                    uint64_t x = v[base + o], y = v[base + o + stride];
                    if constexpr (VARIANT == FWD_APPROX_SELECT || VARIANT == FWD_APPROX_MASK || VARIANT == FWD_APPROX_MIN || VARIANT == FWD_APPROX_NOCSUB) {
                        if constexpr (VARIANT == FWD_APPROX_SELECT) x = csub_select(x, hb);
                        if constexpr (VARIANT == FWD_APPROX_MASK) x = csub_mask(x, hb);
                        if constexpr (VARIANT == FWD_APPROX_MIN) x = csub_min32(x, hb);
                        const uint64_t t = shoup_lazy4(y, w.x, w.y, neg_p);
                        v[base + o] = x + t;
                        v[base + o + stride] = x + hb - t;
                    } else if constexpr (VARIANT == FWD_EXACT_SELECT) {
                        x = csub_select(x, two_p);
                        const uint64_t t = shoup_lazy(y, w.x, w.y, neg_p);
                        v[base + o] = x + t;
                        v[base + o + stride] = x + two_p - t;
                    } else if constexpr (VARIANT == INV_APPROX_SELECT || VARIANT == INV_APPROX_MASK || VARIANT == INV_APPROX_NOCSUB) {
                        uint64_t sum = x + y;
                        const uint64_t diff = x + hb - y;
                        if constexpr (VARIANT == INV_APPROX_SELECT) sum = csub_select(sum, hb);
                        if constexpr (VARIANT == INV_APPROX_MASK) sum = csub_mask(sum, hb);
                        v[base + o] = sum;
                        v[base + o + stride] = shoup_lazy4(diff, w.x, w.y, neg_p);
                    } else if constexpr (VARIANT == FWD_HEADROOM) {
                        const uint64_t t = shoup_headroom(y & 0x3fffffffffffffffull, w.x, w.y >> 1, 0 - two_p);
                        v[base + o] = x + t;
                        v[base + o + stride] = x + 2 * hb - t;
                    } else if constexpr (VARIANT == INV_HEADROOM) {
                        const uint64_t sum = x + y;
                        const uint64_t diff = x + hb - y;
                        v[base + o] = sum;
                        v[base + o + stride] = shoup_headroom(diff & 0x3fffffffffffffffull, w.x, w.y >> 1, 0 - two_p);
                    } else if constexpr (VARIANT == MUL_ONLY_HEADROOM) {
                        v[base + o] = shoup_headroom(y & 0x3fffffffffffffffull, w.x, w.y >> 1, 0 - two_p);
                        v[base + o + stride] = x;
                    } else if constexpr (VARIANT == CSUB_ONLY_SELECT) {
                        v[base + o] = csub_select(x + y, hb);
                        v[base + o + stride] = csub_select(y + w.x, hb);
                    } else if constexpr (VARIANT == CSUB_ONLY_MASK) {
                        v[base + o] = csub_mask(x + y, hb);
                        v[base + o + stride] = csub_mask(y + w.x, hb);
                    } else if constexpr (VARIANT == CSUB_ONLY_MIN) {
                        v[base + o] = csub_min32(x + y, hb);
                        v[base + o + stride] = csub_min32(y + w.x, hb);
                    } else if constexpr (VARIANT == MUL_ONLY_APPROX) {
                        v[base + o] = shoup_lazy4(y, w.x, w.y, neg_p);
                        v[base + o + stride] = x;
                    } else if constexpr (VARIANT == MUL_ONLY_EXACT) {
                        v[base + o] = shoup_lazy(y, w.x, w.y, neg_p);
                        v[base + o + stride] = x;
                    }
                }
            }
        }
    }
    const long long t1 = wall_clock64();
    uint64_t sum = 0;
#pragma unroll
    for (int r = 0; r < 32; ++r) sum ^= v[r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int VARIANT>
void run_bfly(const char* name, int ops_per_iter) {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const uint64_t p = 36028797018652673ull;
    for (int wg_per_cu : {1, 2, 4}) {
        const int blocks = prop.multiProcessorCount * wg_per_cu, iters = 400;
        uint64_t* out;
        long long* cycles;
        CHECK(hipMalloc(&out, size_t(blocks) * 256 * 8));
        CHECK(hipMalloc(&cycles, size_t(blocks) * 8));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        bfly_kernel<VARIANT><<<blocks, 256>>>(out, p, 12345, 10, cycles);
        CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0));
        bfly_kernel<VARIANT><<<blocks, 256>>>(out, p, 6789, iters, cycles);
        CHECK(hipEventRecord(e1));
        CHECK(hipDeviceSynchronize());
        float ms = 0;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double ops = double(blocks) * 256 * iters * ops_per_iter;  // lane-level butterflies (or csub pairs)
        printf("%-20s waves/SIMD=%d  %.3f ms  %.3f T lane-ops/s  -> ns per wave-op per SIMD %.2f\n", name, wg_per_cu, ms,
               ops / (ms * 1e-3) / 1e12, (ms * 1e6) / (double(iters) * ops_per_iter * wg_per_cu));
        CHECK(hipFree(out));
        CHECK(hipFree(cycles));
    }
}

// ---- streaming copy: 8 B/lane vs 16 B/lane, to see what HBM rate the NTT's access widths can reach ----
template <typename T>
__global__ void __launch_bounds__(256) copy_kernel(const T* __restrict__ in, T* __restrict__ out, size_t n) {
    for (size_t i = blockIdx.x * size_t(256) + threadIdx.x; i < n; i += size_t(gridDim.x) * 256) out[i] = in[i];
}
// strided 64B-per-lane pattern of the last forward pass: lane owns 8 contiguous words
__global__ void __launch_bounds__(256) copy_lane64B_kernel(const ulonglong2* __restrict__ in, ulonglong2* __restrict__ out, size_t n16) {
    // n16 = number of 16-byte elements; each lane moves 4 consecutive ones
    for (size_t i = (blockIdx.x * size_t(256) + threadIdx.x) * 4; i + 3 < n16; i += size_t(gridDim.x) * 256 * 4) {
        ulonglong2 a = in[i], b = in[i + 1], c = in[i + 2], d = in[i + 3];
        out[i] = a; out[i + 1] = b; out[i + 2] = c; out[i + 3] = d;
    }
}

void run_copy() {
    const size_t bytes = size_t(1) << 30;
    void *in, *out;
    CHECK(hipMalloc(&in, bytes));
    CHECK(hipMalloc(&out, bytes));
    CHECK(hipMemset(in, 1, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int variant = 0; variant < 3; ++variant) {
        for (int blocks : {2048, 8192, 65536}) {
            float best = 1e9;
            for (int rep = 0; rep < 5; ++rep) {
                CHECK(hipEventRecord(e0));
                if (variant == 0) copy_kernel<uint64_t><<<blocks, 256>>>((const uint64_t*)in, (uint64_t*)out, bytes / 8);
                if (variant == 1) copy_kernel<ulonglong2><<<blocks, 256>>>((const ulonglong2*)in, (ulonglong2*)out, bytes / 16);
                if (variant == 2) copy_lane64B_kernel<<<blocks, 256>>>((const ulonglong2*)in, (ulonglong2*)out, bytes / 16);
                CHECK(hipEventRecord(e1));
                CHECK(hipDeviceSynchronize());
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const char* names[] = {"copy_8B_per_lane", "copy_16B_per_lane", "copy_64B_per_lane"};
            printf("%-18s blocks=%6d  %.3f ms  %.1f GB/s (read+write)\n", names[variant], blocks, best, 2.0 * bytes / (best * 1e-3) / 1e9);
        }
    }
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  clock=%d kHz  wall_clock_rate=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate, 0);
    int wall_rate = 0;
    CHECK(hipDeviceGetAttribute(&wall_rate, hipDeviceAttributeWallClockRate, 0));
    printf("wall clock rate: %d kHz\n", wall_rate);
    run_probe<ADD_U32>("v_add_u32", 1);
    run_probe<NOT_B32>("v_not_b32", 1);
    run_probe<MAD_U64_U32>("v_mad_u64_u32", 1);
    run_probe<MAD_I64_I32>("v_mad_i64_i32", 1);
    run_probe<MUL_LO_U32>("v_mul_lo_u32", 1);
    run_probe<MUL_HI_U32>("v_mul_hi_u32", 1);
    run_probe<LSHL_ADD_U64>("v_lshl_add_u64", 1);
    run_probe<ADDCO_PAIR>("add_co+addc pair", 2);
    run_probe<CNDMASK>("v_cndmask_b32", 1);
    run_probe<MAD_U32_U24>("v_mad_u32_u24", 1);
    run_probe<MUL_HI_U32_U24>("v_mul_hi_u32_u24", 1);
    run_probe<FMA_F64>("v_fma_f64", 1);
    run_probe<CMP_GT_I64>("v_cmp_gt_i64", 1);
    run_probe<CMP_GT_I32>("v_cmp_gt_i32", 1);
    run_bfly<FWD_APPROX_SELECT>("fwd_approx_select", 80);
    run_bfly<FWD_APPROX_MASK>("fwd_approx_mask", 80);
    run_bfly<FWD_APPROX_MIN>("fwd_approx_min", 80);
    run_bfly<FWD_APPROX_NOCSUB>("fwd_approx_nocsub", 80);
    run_bfly<FWD_EXACT_SELECT>("fwd_exact_select", 80);
    run_bfly<INV_APPROX_SELECT>("inv_approx_select", 80);
    run_bfly<INV_APPROX_MASK>("inv_approx_mask", 80);
    run_bfly<INV_APPROX_NOCSUB>("inv_approx_nocsub", 80);
    run_bfly<FWD_HEADROOM>("fwd_headroom(+and)", 80);
    run_bfly<INV_HEADROOM>("inv_headroom(+and)", 80);
    run_bfly<MUL_ONLY_HEADROOM>("mul_only_headroom", 80);
    run_bfly<CSUB_ONLY_SELECT>("csub_only_select", 160);
    run_bfly<CSUB_ONLY_MASK>("csub_only_mask", 160);
    run_bfly<CSUB_ONLY_MIN>("csub_only_min", 160);
    run_bfly<MUL_ONLY_APPROX>("mul_only_approx", 80);
    run_bfly<MUL_ONLY_EXACT>("mul_only_exact", 80);
    run_copy();
    return 0;
}
