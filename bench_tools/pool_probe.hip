// pool_probe.hip -- does a stream-ordered allocate / launch / free cycle return to the host without waiting for the GPU?
// Times every API call of `cycles` cycles of { hipMallocFromPoolAsync, a kernel of ~`ms` milliseconds, hipFreeAsync } on
// (a) the legacy default stream, (b) a non-blocking stream, from (1) a pool the program created, (2) the device's default
// pool.  A call that returns in microseconds while milliseconds of work are queued is asynchronous; one that takes as long
// as a kernel has waited for the GPU.     hipcc --offload-arch=gfx950 -O2 -o bench_tools/pool_probe bench_tools/pool_probe.hip
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>

__global__ void spin(uint64_t* out, long iterations) {
    uint64_t x = threadIdx.x;
    for (long i = 0; i < iterations; ++i) x = x * 6364136223846793005ull + 1442695040888963407ull;
    if (x == 42) out[0] = x;
}

static double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
#define CHECK(x)                                                                 \
    do {                                                                         \
        hipError_t e_ = (x);                                                     \
        if (e_ != hipSuccess) {                                                  \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));           \
            std::exit(1);                                                        \
        }                                                                        \
    } while (0)

static void run(const char* label, hipStream_t stream, hipMemPool_t pool, long iterations, int cycles, size_t bytes,
                int frees_per_cycle) {
    std::printf("== %s\n", label);
    CHECK(hipDeviceSynchronize());
    for (int c = 0; c < cycles; ++c) {
        void* p[8] = {};
        double t0 = now_ms();
        for (int k = 0; k < frees_per_cycle; ++k) CHECK(hipMallocFromPoolAsync(&p[k], bytes, pool, stream));
        double t1 = now_ms();
        spin<<<1, 64, 0, stream>>>(static_cast<uint64_t*>(p[0]), iterations);
        double t2 = now_ms();
        for (int k = 0; k < frees_per_cycle; ++k) CHECK(hipFreeAsync(p[k], stream));
        double t3 = now_ms();
        std::printf("  cycle %d: malloc %.3f ms  launch %.3f ms  free %.3f ms\n", c, t1 - t0, t2 - t1, t3 - t2);
    }
    double t0 = now_ms();
    CHECK(hipStreamSynchronize(stream));
    std::printf("  drain %.3f ms\n", now_ms() - t0);
}

int main(int argc, char** argv) {
    const long iterations = argc > 1 ? std::atol(argv[1]) : 4000000;  // ~5 ms
    const int cycles = 6;
    const size_t bytes = size_t(64) << 20;
    hipStream_t stream = nullptr;
    CHECK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    hipMemPoolProps props = {};
    props.allocType = hipMemAllocationTypePinned;
    props.handleTypes = hipMemHandleTypeNone;
    props.location.type = hipMemLocationTypeDevice;
    props.location.id = 0;
    hipMemPool_t own = nullptr, dflt = nullptr;
    CHECK(hipMemPoolCreate(&own, &props));
    uint64_t keep = ~0ull;
    CHECK(hipMemPoolSetAttribute(own, hipMemPoolAttrReleaseThreshold, &keep));
    CHECK(hipDeviceGetDefaultMemPool(&dflt, 0));
    CHECK(hipMemPoolSetAttribute(dflt, hipMemPoolAttrReleaseThreshold, &keep));
    // warm: module load, first allocations
    run("warm-up (own pool, stream)", stream, own, 1000, 2, bytes, 1);
    run("own pool, non-blocking stream, 1 buffer", stream, own, iterations, cycles, bytes, 1);
    run("own pool, non-blocking stream, 4 buffers", stream, own, iterations, cycles, bytes, 4);
    run("own pool, legacy default stream, 1 buffer", nullptr, own, iterations, cycles, bytes, 1);
    run("default pool, non-blocking stream, 1 buffer", stream, dflt, iterations, cycles, bytes, 1);
    run("default pool, legacy default stream, 1 buffer", nullptr, dflt, iterations, cycles, bytes, 1);
    return 0;
}
