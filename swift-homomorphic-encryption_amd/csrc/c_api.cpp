// c_api.cpp -- the extern "C" boundary declared in include/he_amd.h.
#include "../../include/he_amd.h"

#include <hip/hip_runtime.h>

#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "api_internal.hpp"
#include "kernels.hpp"
#include "poly_context.hpp"

using heamd::as_stream;
using heamd::invalid_argument;
using heamd::PolyContext;
using heamd::Scratch;

namespace {

// Runs `body(device_ptr, stream)` on a device copy of a host slab and copies the result back (blocking).
template <typename Body>
int with_device_copy(uint64_t* host, size_t words, Body body) {
    hipStream_t stream = hipStreamPerThread;
    void* device = nullptr;
    HEAMD_HIP_TRY(hipMalloc(&device, words * sizeof(uint64_t) + 16));
    int status = HE_OK;
    hipError_t e = hipMemcpyAsync(device, host, words * sizeof(uint64_t), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) {
        status = body(static_cast<uint64_t*>(device), stream);
        if (status == HE_OK) e = hipMemcpyAsync(host, device, words * sizeof(uint64_t), hipMemcpyDeviceToHost, stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(device);
    if (status != HE_OK) return status;
    if (e != hipSuccess) return heamd::device_failure(e, "host-pointer round trip");
    return HE_OK;
}

int ntt_device(const he_poly_context* ctx, uint64_t* slab, size_t batch, bool inverse, hipStream_t stream) {
    if (ctx == nullptr) return invalid_argument("null context");
    const PolyContext& pc = *ctx->impl;
    // validateNttModuli (PolyContext.swift:175-181) comes first in forwardNtt(poly:) / inverseNtt(poly:)
    if (!pc.all_ntt(pc.moduli_count())) {
        heamd::set_last_error("a modulus of this context is not an NTT modulus for degree " + std::to_string(pc.degree()));
        return HE_ERR_INVALID_NTT_MODULUS;
    }
    if (batch == 0) return HE_OK;
    if (slab == nullptr) return invalid_argument("null slab");
    int status = pc.check_device();
    if (status != HE_OK) return status;
    HEAMD_HIP_TRY(heamd::launch_ntt(inverse, slab, pc.device_context(), 0, pc.moduli_count(),
                                    batch * pc.moduli_count(), stream));
    return HE_OK;
}

int ntt_rows_device(const he_poly_context* ctx, uint64_t modulus, uint64_t* rows_ptr, size_t rows, bool inverse,
                    hipStream_t stream) {
    if (ctx == nullptr) return invalid_argument("null context");
    const PolyContext& pc = *ctx->impl;
    // PolyContext.forwardNtt(dataPtr:modulus:) walks the chain for a context whose last modulus matches and
    // throws invalidPolyContext otherwise, or when that context has no nttContext (PolyRq+Ntt.swift:329-341).
    const int index = pc.modulus_index(modulus);
    if (index < 0 || !pc.host_constants()[index].has_ntt) {
        heamd::set_last_error("modulus " + std::to_string(modulus) + " has no NTT context in this PolyContext");
        return HE_ERR_INVALID_POLY_CONTEXT;
    }
    if (rows == 0) return HE_OK;
    if (rows_ptr == nullptr) return invalid_argument("null rows");
    int status = pc.check_device();
    if (status != HE_OK) return status;
    heamd::DeviceContext dc = pc.device_context();
    dc.approx_ok = modulus < (uint64_t(1) << 61) ? 1 : 0;
    dc.headroom_ok = (modulus < (uint64_t(1) << 55) && modulus >= (uint64_t(1) << 40)) ? 1 : 0;
    HEAMD_HIP_TRY(heamd::launch_ntt(inverse, rows_ptr, dc, static_cast<uint32_t>(index), 1, rows, stream));
    return HE_OK;
}

int elementwise(const he_poly_context* ctx, heamd::ElementwiseOp op, uint64_t* lhs, const uint64_t* rhs, size_t batch,
                he_stream s) {
    if (ctx == nullptr) return invalid_argument("null context");
    const PolyContext& pc = *ctx->impl;
    if (batch == 0) return HE_OK;
    if (lhs == nullptr || (rhs == nullptr && op != heamd::ElementwiseOp::Neg)) return invalid_argument("null slab");
    int status = pc.check_device();
    if (status != HE_OK) return status;
    HEAMD_HIP_TRY(heamd::launch_elementwise(op, lhs, rhs, pc.device_context(), batch * pc.moduli_count(), as_stream(s)));
    return HE_OK;
}

}  // namespace

// ---- scratch (api_internal.hpp scratch_allocate / scratch_release) ----------------------------------------------------
// Two sources, per device:
//  * a HIP memory pool the library creates (release threshold 0): what scratch comes from while the host has not asked for a
//    cache, and always while the stream is being captured into a graph (allocation and free become graph nodes);
//  * the library's own stream-ordered block cache, once he_set_scratch_cache(bytes > 0) was called: blocks from hipMalloc,
//    kept on a free list with the stream they were released on and an event recorded there.  A later request on the SAME
//    stream takes a block without any driver call (stream order makes it safe); a request on another stream first makes that
//    stream wait for the block's event.  Why not the HIP pool with a high release threshold: hipFreeAsync of ROCm 7.2 holds
//    the calling thread until the work enqueued before the PREVIOUS release of that block has finished
//    (bench_tools/pool_probe.hip, profiles/r06b_pool_probe.txt: 106 ms kernels, every free from the second cycle on returns
//    after 106 ms) -- every call that takes scratch would return one call late instead of being enqueue-only.
namespace {
constexpr int kMaxDevices = 64;
struct CachedBlock {
    void* ptr;
    size_t bytes;
    hipStream_t stream;  // released on
    hipEvent_t freed;    // recorded on `stream` at the release
    uint64_t tick;
};
struct ScratchState {
    std::mutex mutex;
    hipMemPool_t pool = nullptr;
    bool pool_failed = false;  // pool creation refused: fall back to the default pool, untouched
    uint64_t limit = 0;        // he_set_scratch_cache: bytes of released scratch the block cache may keep; 0 = cache off
    size_t cached_bytes = 0;
    uint64_t tick = 0;
    std::vector<CachedBlock> free_blocks;
    std::map<void*, size_t> lent;  // blocks of the cache that are in use -> their size
    std::vector<hipEvent_t> spare_events;
};
ScratchState& scratch_state(int device) {
    static ScratchState* states = new ScratchState[kMaxDevices];  // never destroyed: HIP may already be gone at exit
    return states[device];
}
bool current_device(int* device) {
    return hipGetDevice(device) == hipSuccess && *device >= 0 && *device < kMaxDevices;
}
// (under the state's mutex) the device's HIP pool, or nullptr
hipMemPool_t hip_pool_locked(ScratchState& state, int device) {
    if (state.pool == nullptr && !state.pool_failed) {
        hipMemPoolProps props = {};
        props.allocType = hipMemAllocationTypePinned;
        props.handleTypes = hipMemHandleTypeNone;
        props.location.type = hipMemLocationTypeDevice;
        props.location.id = device;
        hipMemPool_t pool = nullptr;
        if (hipMemPoolCreate(&pool, &props) == hipSuccess && pool != nullptr) {
            uint64_t threshold = 0;
            (void)hipMemPoolSetAttribute(pool, hipMemPoolAttrReleaseThreshold, &threshold);
            state.pool = pool;
        } else {
            state.pool_failed = true;
        }
        (void)hipGetLastError();
    }
    return state.pool;
}
bool stream_is_capturing(hipStream_t stream) {
    hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &capture) != hipSuccess) {
        (void)hipGetLastError();  // the legacy default stream cannot be queried -- nor captured
        return false;
    }
    return capture != hipStreamCaptureStatusNone;
}
size_t block_size(size_t bytes) {
    const size_t grain = bytes >= (size_t(1) << 20) ? (size_t(1) << 20) : (size_t(64) << 10);
    return (bytes + grain - 1) / grain * grain;
}
// (under the mutex) gives cached blocks whose release has completed back to the driver, oldest first, until at most `keep`
// bytes stay cached; blocks still in flight stay.  hipFree may wait for the device: only trims and over-limit releases
// come here.
void evict_locked(ScratchState& state, size_t keep) {
    while (state.cached_bytes > keep) {
        size_t pick = state.free_blocks.size();
        for (size_t i = 0; i < state.free_blocks.size(); ++i) {
            if (hipEventQuery(state.free_blocks[i].freed) != hipSuccess) {
                (void)hipGetLastError();
                continue;
            }
            if (pick == state.free_blocks.size() || state.free_blocks[i].tick < state.free_blocks[pick].tick) pick = i;
        }
        if (pick == state.free_blocks.size()) return;
        CachedBlock block = state.free_blocks[pick];
        state.free_blocks.erase(state.free_blocks.begin() + static_cast<long>(pick));
        state.cached_bytes -= block.bytes;
        state.spare_events.push_back(block.freed);
        (void)hipFree(block.ptr);
    }
}
}  // namespace

// he_stream_destroy: a later stream may get the same handle -- the blocks released on this one no longer count as released
// on "the same stream" (their events still order them)
void heamd::scratch_forget_stream(hipStream_t stream) {
    if (stream == nullptr) return;
    int device = 0;
    if (!current_device(&device)) return;
    ScratchState& state = scratch_state(device);
    std::lock_guard<std::mutex> lock(state.mutex);
    for (CachedBlock& block : state.free_blocks)
        if (block.stream == stream) block.stream = reinterpret_cast<hipStream_t>(~uintptr_t(0));
}

hipError_t heamd::scratch_allocate(void** out, size_t bytes, hipStream_t stream) {
    int device = 0;
    if (!current_device(&device)) return hipMallocAsync(out, bytes, stream);
    ScratchState& state = scratch_state(device);
    std::unique_lock<std::mutex> lock(state.mutex);
    if (state.limit == 0 || stream_is_capturing(stream)) {
        hipMemPool_t pool = hip_pool_locked(state, device);
        lock.unlock();
        return pool != nullptr ? hipMallocFromPoolAsync(out, bytes, pool, stream) : hipMallocAsync(out, bytes, stream);
    }
    const size_t want = block_size(bytes);
    // the smallest cached block that holds the request without wasting more than half of itself; one released on this very
    // stream first (no wait), then any other
    size_t pick = state.free_blocks.size();
    for (size_t i = 0; i < state.free_blocks.size(); ++i) {
        const CachedBlock& b = state.free_blocks[i];
        if (b.bytes < want || b.bytes > 2 * want) continue;
        if (pick == state.free_blocks.size()) {
            pick = i;
            continue;
        }
        const CachedBlock& best = state.free_blocks[pick];
        const bool same = b.stream == stream, best_same = best.stream == stream;
        if (same != best_same ? same : b.bytes < best.bytes) pick = i;
    }
    if (pick != state.free_blocks.size()) {
        CachedBlock block = state.free_blocks[pick];
        if (block.stream != stream) {
            const hipError_t e = hipStreamWaitEvent(stream, block.freed, 0);
            if (e != hipSuccess) return e;
        }
        state.free_blocks.erase(state.free_blocks.begin() + static_cast<long>(pick));
        state.cached_bytes -= block.bytes;
        state.spare_events.push_back(block.freed);
        state.lent[block.ptr] = block.bytes;
        *out = block.ptr;
        return hipSuccess;
    }
    void* ptr = nullptr;
    hipError_t e = hipMalloc(&ptr, want);
    if (e == hipErrorOutOfMemory) {  // the cache itself may be what fills the device
        (void)hipGetLastError();
        evict_locked(state, 0);
        e = hipMalloc(&ptr, want);
    }
    if (e != hipSuccess) return e;
    state.lent[ptr] = want;
    *out = ptr;
    return hipSuccess;
}

namespace {
// (under the state's mutex) takes `ptr` back into `state`'s cache if it was lent from it
bool release_into(ScratchState& state, void* ptr, hipStream_t stream);
}  // namespace

void heamd::scratch_release(void* ptr, hipStream_t stream) {
    if (ptr == nullptr) return;
    int device = 0;
    const bool known = current_device(&device);
    if (known && release_into(scratch_state(device), ptr, stream)) return;
    // a block of another device's cache (the caller changed the current device between the two ends of a call: the event is
    // then recorded on a stream of the block's own device, which `stream` is)
    for (int other = 0; other < kMaxDevices; ++other)
        if (!(known && other == device) && release_into(scratch_state(other), ptr, stream)) return;
    (void)hipFreeAsync(ptr, stream);  // from the HIP pool
}

namespace {
bool release_into(ScratchState& state, void* ptr, hipStream_t stream) {
    {
        std::lock_guard<std::mutex> lock(state.mutex);
        auto it = state.lent.find(ptr);
        if (it != state.lent.end()) {
            const size_t bytes = it->second;
            state.lent.erase(it);
            hipEvent_t event = nullptr;
            if (!state.spare_events.empty()) {
                event = state.spare_events.back();
                state.spare_events.pop_back();
            } else if (hipEventCreateWithFlags(&event, hipEventDisableTiming) != hipSuccess) {
                event = nullptr;
            }
            if (event == nullptr || hipEventRecord(event, stream) != hipSuccess) {
                // no way to tell when the block is free again: wait for the stream, then hand it back
                (void)hipGetLastError();
                (void)hipStreamSynchronize(stream);
                (void)hipFree(ptr);
                if (event != nullptr) state.spare_events.push_back(event);
                return true;
            }
            state.free_blocks.push_back(CachedBlock{ptr, bytes, stream, event, ++state.tick});
            state.cached_bytes += bytes;
            if (state.cached_bytes > state.limit) evict_locked(state, static_cast<size_t>(state.limit));
            return true;
        }
    }
    return false;
}
}  // namespace

extern "C" {

const char* he_status_string(int status) {
    switch (status) {
        case HE_OK: return "ok";
        case HE_ERR_INVALID_DEGREE: return "invalidDegree";
        case HE_ERR_INVALID_MODULUS: return "invalidModulus";
        case HE_ERR_COPRIME_MODULI: return "coprimeModuli";
        case HE_ERR_EMPTY_MODULUS: return "emptyModulus";
        case HE_ERR_INVALID_NTT_MODULUS: return "invalidNttModulus";
        case HE_ERR_INVALID_POLY_CONTEXT: return "invalidPolyContext";
        case HE_ERR_POLY_CONTEXT_MISMATCH: return "polyContextMismatch";
        case HE_ERR_INVALID_CIPHERTEXT: return "invalidCiphertext";
        case HE_ERR_INCOMPATIBLE_CIPHERTEXTS: return "incompatibleCiphertexts";
        case HE_ERR_INCOMPATIBLE_CIPHERTEXT_AND_PLAINTEXT: return "incompatibleCiphertextAndPlaintext";
        case HE_ERR_MISSING_RELINEARIZATION_KEY: return "missingRelinearizationKey";
        case HE_ERR_UNEQUAL_CONTEXTS: return "unequalContexts";
        case HE_ERR_NOT_ENOUGH_PRIMES: return "notEnoughPrimes";
        case HE_ERR_NOT_INVERTIBLE: return "notInvertible";
        case HE_ERR_INVALID_ENCRYPTION_PARAMETERS: return "invalidEncryptionParameters";
        case HE_ERR_INVALID_ARGUMENT: return "invalidArgument";
        case HE_ERR_DEVICE: return "deviceError";
        case HE_ERR_UNSUPPORTED: return "unsupportedHeOperation";
        case HE_ERR_MISSING_GALOIS_KEY: return "missingGaloisKey";
        case HE_ERR_SERIALIZED_BUFFER_SIZE_MISMATCH: return "serializedBufferSizeMismatch";
        case HE_ERR_INVALID_COEFFICIENT_PACKING: return "invalidCoefficientPacking";
        default: return "unknown";
    }
}

const char* he_last_error_message(void) { return heamd::last_error(); }
const char* he_version(void) { return "he_amd 0.1 (gfx950, HIP)"; }

int he_device_count(int* out_count) {
    if (out_count == nullptr) return invalid_argument("null out_count");
    *out_count = 0;
    HEAMD_HIP_TRY(hipGetDeviceCount(out_count));
    return HE_OK;
}
int he_get_device(int* out_device) {
    if (out_device == nullptr) return invalid_argument("null out_device");
    HEAMD_HIP_TRY(hipGetDevice(out_device));
    return HE_OK;
}
int he_set_device(int device) {
    HEAMD_HIP_TRY(hipSetDevice(device));
    return HE_OK;
}

int he_set_scratch_cache(uint64_t bytes) {
    int device = 0;
    if (!current_device(&device)) {
        heamd::set_last_error("no current device");
        return HE_ERR_DEVICE;
    }
    ScratchState& state = scratch_state(device);
    std::lock_guard<std::mutex> lock(state.mutex);
    state.limit = bytes;
    if (state.cached_bytes > bytes) evict_locked(state, static_cast<size_t>(bytes));
    return HE_OK;
}
int he_device_trim_scratch(uint64_t keep_bytes) {
    int device = 0;
    if (!current_device(&device)) return HE_OK;
    ScratchState& state = scratch_state(device);
    std::lock_guard<std::mutex> lock(state.mutex);
    evict_locked(state, static_cast<size_t>(keep_bytes));
    if (state.pool != nullptr) HEAMD_HIP_TRY(hipMemPoolTrimTo(state.pool, static_cast<size_t>(keep_bytes)));
    return HE_OK;
}
int he_scratch_cached_bytes(uint64_t* out_bytes) {
    if (out_bytes == nullptr) return invalid_argument("null out_bytes");
    *out_bytes = 0;
    int device = 0;
    if (!current_device(&device)) return HE_OK;
    ScratchState& state = scratch_state(device);
    std::lock_guard<std::mutex> lock(state.mutex);
    *out_bytes = state.cached_bytes;
    return HE_OK;
}

int he_device_malloc(void** out_ptr, size_t bytes) {
    if (out_ptr == nullptr) return invalid_argument("null out_ptr");
    *out_ptr = nullptr;
    HEAMD_HIP_TRY(hipMalloc(out_ptr, bytes ? bytes : 1));
    return HE_OK;
}
int he_device_free(void* ptr) {
    heamd::RelaxedCapture relaxed;
    if (ptr == nullptr) return HE_OK;
    HEAMD_HIP_TRY(hipFree(ptr));
    return HE_OK;
}
int he_host_malloc(void** out_ptr, size_t bytes) {
    if (out_ptr == nullptr) return invalid_argument("null out_ptr");
    *out_ptr = nullptr;
    HEAMD_HIP_TRY(hipHostMalloc(out_ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return HE_OK;
}
int he_host_free(void* ptr) {
    heamd::RelaxedCapture relaxed;
    if (ptr == nullptr) return HE_OK;
    HEAMD_HIP_TRY(hipHostFree(ptr));
    return HE_OK;
}
int he_memcpy_h2d(void* dst_device, const void* src_host, size_t bytes, he_stream stream) {
    HEAMD_HIP_TRY(hipMemcpyAsync(dst_device, src_host, bytes, hipMemcpyHostToDevice, as_stream(stream)));
    return HE_OK;
}
int he_memcpy_d2h(void* dst_host, const void* src_device, size_t bytes, he_stream stream) {
    HEAMD_HIP_TRY(hipMemcpyAsync(dst_host, src_device, bytes, hipMemcpyDeviceToHost, as_stream(stream)));
    return HE_OK;
}
int he_stream_synchronize(he_stream stream) {
    HEAMD_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return HE_OK;
}
int he_stream_create(he_stream* out) {
    if (out == nullptr) return invalid_argument("null out");
    hipStream_t stream = nullptr;
    HEAMD_HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    *out = stream;
    return HE_OK;
}
int he_stream_destroy(he_stream stream) {
    heamd::RelaxedCapture relaxed;
    if (stream == nullptr) return HE_OK;
    heamd::scratch_forget_stream(as_stream(stream));
    HEAMD_HIP_TRY(hipStreamDestroy(as_stream(stream)));
    return HE_OK;
}

// ---- completion primitives for the reference's `...Async` twins (HeSchemeAsync.swift:16-141): a Swift `async`
// wrapper enqueues the `_device` call and suspends on he_stream_add_callback (or polls / waits on an event) instead of
// blocking a thread of the cooperative pool in he_stream_synchronize.
int he_event_create(he_event* out) {
    if (out == nullptr) return invalid_argument("null out");
    hipEvent_t event = nullptr;
    HEAMD_HIP_TRY(hipEventCreateWithFlags(&event, hipEventDisableTiming));
    *out = event;
    return HE_OK;
}
int he_event_destroy(he_event event) {
    heamd::RelaxedCapture relaxed;
    if (event == nullptr) return HE_OK;
    HEAMD_HIP_TRY(hipEventDestroy(static_cast<hipEvent_t>(event)));
    return HE_OK;
}
int he_event_record(he_event event, he_stream stream) {
    if (event == nullptr) return invalid_argument("null event");
    HEAMD_HIP_TRY(hipEventRecord(static_cast<hipEvent_t>(event), as_stream(stream)));
    return HE_OK;
}
int he_event_synchronize(he_event event) {
    if (event == nullptr) return invalid_argument("null event");
    HEAMD_HIP_TRY(hipEventSynchronize(static_cast<hipEvent_t>(event)));
    return HE_OK;
}
int he_event_query(he_event event, int* out_done) {
    if (event == nullptr || out_done == nullptr) return invalid_argument("null event");
    const hipError_t e = hipEventQuery(static_cast<hipEvent_t>(event));
    if (e == hipErrorNotReady) {
        (void)hipGetLastError();
        *out_done = 0;
        return HE_OK;
    }
    HEAMD_HIP_TRY(e);
    *out_done = 1;
    return HE_OK;
}
int he_stream_wait_event(he_stream stream, he_event event) {
    if (event == nullptr) return invalid_argument("null event");
    HEAMD_HIP_TRY(hipStreamWaitEvent(as_stream(stream), static_cast<hipEvent_t>(event), 0));
    return HE_OK;
}
namespace {
struct HostCallback {
    he_host_callback function;
    void* user_data;
};
void run_host_callback(void* raw) {
    HostCallback* call = static_cast<HostCallback*>(raw);
    call->function(call->user_data);
    delete call;
}
}  // namespace
int he_stream_add_callback(he_stream stream, he_host_callback callback, void* user_data) {
    if (callback == nullptr) return invalid_argument("null callback");
    HostCallback* call = new HostCallback{callback, user_data};
    const hipError_t e = hipLaunchHostFunc(as_stream(stream), run_host_callback, call);
    if (e != hipSuccess) {
        delete call;
        return heamd::device_failure(e, "hipLaunchHostFunc");
    }
    return HE_OK;
}

// ------------------------------------------------------------------------------------------ PolyContext
static int poly_context_create(uint32_t degree, const uint64_t* moduli, uint32_t moduli_count, bool host_only,
                               he_poly_context** out) {
    if (out == nullptr) return invalid_argument("null out");
    *out = nullptr;
    std::unique_ptr<PolyContext> impl;
    const int status = PolyContext::create(degree, moduli, moduli_count, impl, host_only);
    if (status != HE_OK) {
        if (status != HE_ERR_DEVICE) heamd::set_last_error(std::string("PolyContext.init: ") + he_status_string(status));
        return status;
    }
    *out = new he_poly_context{impl.release(), true};
    return HE_OK;
}
int he_poly_context_create(uint32_t degree, const uint64_t* moduli, uint32_t moduli_count, he_poly_context** out) {
    return poly_context_create(degree, moduli, moduli_count, false, out);
}
int he_poly_context_create_host_only(uint32_t degree, const uint64_t* moduli, uint32_t moduli_count,
                                     he_poly_context** out) {
    return poly_context_create(degree, moduli, moduli_count, true, out);
}
void he_poly_context_destroy(he_poly_context* ctx) {
    heamd::RelaxedCapture relaxed;
    delete ctx;
}
uint32_t he_poly_context_degree(const he_poly_context* ctx) { return ctx ? ctx->impl->degree() : 0; }
uint32_t he_poly_context_moduli_count(const he_poly_context* ctx) { return ctx ? ctx->impl->moduli_count() : 0; }
int he_poly_context_moduli(const he_poly_context* ctx, uint64_t* out_moduli) {
    if (ctx == nullptr || out_moduli == nullptr) return invalid_argument("null pointer");
    const auto& moduli = ctx->impl->moduli();
    for (size_t i = 0; i < moduli.size(); ++i) out_moduli[i] = moduli[i];
    return HE_OK;
}
uint64_t he_poly_context_max_lazy_product_accumulation_count(const he_poly_context* ctx) {
    return ctx ? ctx->impl->max_lazy_product_accumulation_count(ctx->impl->moduli_count()) : 0;
}
int he_poly_context_q_remainder(const he_poly_context* ctx, uint64_t modulus, uint64_t* out) {
    if (ctx == nullptr || out == nullptr || modulus == 0) return invalid_argument("null pointer or zero modulus");
    const auto& moduli = ctx->impl->moduli();
    *out = heamd::product_mod(moduli.data(), moduli.size(), modulus);
    return HE_OK;
}
int he_poly_context_copy_ntt_tables(const he_poly_context* ctx, uint32_t rns_index, uint64_t* root_powers,
                                    uint64_t* root_factors, uint64_t* inverse_root_powers,
                                    uint64_t* inverse_root_factors, uint64_t* inverse_degree,
                                    uint64_t* inverse_degree_root) {
    if (ctx == nullptr) return invalid_argument("null context");
    const PolyContext& pc = *ctx->impl;
    if (rns_index >= pc.moduli_count() || !pc.host_constants()[rns_index].has_ntt) return HE_ERR_INVALID_NTT_MODULUS;
    const heamd::U64x2* forward = pc.host_forward_twiddles(rns_index);
    const heamd::U64x2* inverse = pc.host_inverse_twiddles(rns_index);
    for (uint32_t k = 0; k < pc.degree(); ++k) {
        if (root_powers) root_powers[k] = forward[k].x;
        if (root_factors) root_factors[k] = forward[k].y;
        if (inverse_root_powers) inverse_root_powers[k] = inverse[k].x;
        if (inverse_root_factors) inverse_root_factors[k] = inverse[k].y;
    }
    if (inverse_degree) *inverse_degree = pc.host_constants()[rns_index].inv_degree;
    if (inverse_degree_root) *inverse_degree_root = pc.host_constants()[rns_index].inv_degree_root;
    return HE_OK;
}
int he_generate_primes(const int32_t* significant_bit_counts, uint32_t count, int preferring_small,
                       uint32_t ntt_degree, uint64_t* out_primes) {
    if ((count > 0 && (significant_bit_counts == nullptr || out_primes == nullptr)) ||
        !heamd::is_power_of_two(ntt_degree))
        return invalid_argument("generatePrimes arguments");
    std::vector<int> bits(significant_bit_counts, significant_bit_counts + count);
    std::vector<heamd::u64> primes;
    if (!heamd::generate_primes(bits, preferring_small != 0, ntt_degree, primes)) return HE_ERR_NOT_ENOUGH_PRIMES;
    for (uint32_t i = 0; i < count; ++i) out_primes[i] = primes[i];
    return HE_OK;
}

// ------------------------------------------------------------------------------------------ NTT
int he_ntt_forward_device(const he_poly_context* ctx, uint64_t* device_slab, size_t batch, he_stream stream) {
    return ntt_device(ctx, device_slab, batch, false, as_stream(stream));
}
int he_ntt_inverse_device(const he_poly_context* ctx, uint64_t* device_slab, size_t batch, he_stream stream) {
    return ntt_device(ctx, device_slab, batch, true, as_stream(stream));
}
int he_ntt_forward_rows_device(const he_poly_context* ctx, uint64_t modulus, uint64_t* device_rows, size_t rows,
                               he_stream stream) {
    return ntt_rows_device(ctx, modulus, device_rows, rows, false, as_stream(stream));
}
int he_ntt_inverse_rows_device(const he_poly_context* ctx, uint64_t modulus, uint64_t* device_rows, size_t rows,
                               he_stream stream) {
    return ntt_rows_device(ctx, modulus, device_rows, rows, true, as_stream(stream));
}
static int ntt_host(const he_poly_context* ctx, uint64_t* host_slab, size_t batch, bool inverse) {
    if (ctx == nullptr) return invalid_argument("null context");
    const PolyContext& pc = *ctx->impl;
    if (!pc.all_ntt(pc.moduli_count())) return HE_ERR_INVALID_NTT_MODULUS;
    if (batch == 0) return HE_OK;
    if (host_slab == nullptr) return invalid_argument("null slab");
    int status = pc.check_device();
    if (status != HE_OK) return status;
    // One blocking copy - kernel - copy on the calling thread's stream.  hipMemcpyAsync from pageable memory runs at the
    // link's rate here (51 GB/s of host traffic for a 64 MiB slab, both directions counted); a pipeline of page-locked
    // staging blocks filled by worker threads was built and measured SLOWER (43 GB/s: the host-side memcpy into the blocks
    // costs more than the overlap gains) -- profiles/r05d_host_seam_pipelined_vs_blocking.txt.
    const size_t words = batch * pc.moduli_count() * pc.degree();
    return with_device_copy(host_slab, words, [&](uint64_t* device, hipStream_t stream) {
        return ntt_device(ctx, device, batch, inverse, stream);
    });
}
int he_ntt_forward(const he_poly_context* ctx, uint64_t* host_slab, size_t batch) {
    return ntt_host(ctx, host_slab, batch, false);
}
int he_ntt_inverse(const he_poly_context* ctx, uint64_t* host_slab, size_t batch) {
    return ntt_host(ctx, host_slab, batch, true);
}
// The transform with a named kernel schedule: every accepted variant computes the same, canonical NTT; anything
// else is HE_ERR_INVALID_ARGUMENT.
int he_ntt_device_variant(const he_poly_context* ctx, uint64_t* device_slab, size_t batch, int inverse, int variant,
                          he_stream stream) {
    if (ctx == nullptr) return invalid_argument("null context");
    const PolyContext& pc = *ctx->impl;
    if (!pc.all_ntt(pc.moduli_count())) return HE_ERR_INVALID_NTT_MODULUS;
    switch (variant) {
        case heamd::kNttVariantAuto: case heamd::kNttVariantExact: case heamd::kNttVariantGeneric:
        case heamd::kNttVariantWide: case heamd::kNttVariantApprox: break;
        default: return invalid_argument("unknown NTT variant");
    }
    if (batch == 0) return HE_OK;
    if (device_slab == nullptr) return invalid_argument("null slab");
    int status = pc.check_device();
    if (status != HE_OK) return status;
    HEAMD_HIP_TRY(heamd::launch_ntt(inverse != 0, device_slab, pc.device_context(), 0, pc.moduli_count(),
                                    batch * pc.moduli_count(), as_stream(stream), variant));
    return HE_OK;
}

// ------------------------------------------------------------------------------------------ element-wise
int he_poly_add_device(const he_poly_context* ctx, uint64_t* lhs, const uint64_t* rhs, size_t batch, he_stream s) {
    return elementwise(ctx, heamd::ElementwiseOp::Add, lhs, rhs, batch, s);
}
int he_poly_sub_device(const he_poly_context* ctx, uint64_t* lhs, const uint64_t* rhs, size_t batch, he_stream s) {
    return elementwise(ctx, heamd::ElementwiseOp::Sub, lhs, rhs, batch, s);
}
int he_poly_neg_device(const he_poly_context* ctx, uint64_t* data, size_t batch, he_stream s) {
    return elementwise(ctx, heamd::ElementwiseOp::Neg, data, nullptr, batch, s);
}
int he_poly_mul_device(const he_poly_context* ctx, uint64_t* lhs, const uint64_t* rhs, size_t batch, he_stream s) {
    return elementwise(ctx, heamd::ElementwiseOp::Mul, lhs, rhs, batch, s);
}
int he_poly_apply_galois_device(const he_poly_context* ctx, const uint64_t* in, uint64_t* out, size_t batch,
                                uint64_t element, int eval_format, he_stream s) {
    if (ctx == nullptr) return invalid_argument("null context");
    const PolyContext& pc = *ctx->impl;
    const uint64_t n = pc.degree();
    // isValidGaloisElement (Galois.swift:100-105) is a precondition of applyGalois
    if ((element & 1) == 0 || element <= 1 || element >= 2 * n) return invalid_argument("invalid Galois element");
    if (batch == 0) return HE_OK;
    if (in == nullptr || out == nullptr || in == out) return invalid_argument("null or aliased slabs");
    int status = pc.check_device();
    if (status != HE_OK) return status;
    const size_t rows = batch * pc.moduli_count();
    if (eval_format != 0) {
        HEAMD_HIP_TRY(heamd::launch_galois_eval(in, out, pc.device_context(), static_cast<uint32_t>(element), rows,
                                                as_stream(s)));
    } else {
        uint64_t inverse = element;  // Newton: g^-1 mod 2^k for odd g
        for (int k = 0; k < 6; ++k) inverse *= 2 - element * inverse;
        HEAMD_HIP_TRY(heamd::launch_galois_coeff(in, out, pc.device_context(),
                                                 static_cast<uint32_t>(inverse & (2 * n - 1)), rows, as_stream(s)));
    }
    return HE_OK;
}

int he_poly_multiply_power_of_x_device(const he_poly_context* ctx, const uint64_t* in, uint64_t* out, size_t batch,
                                       int64_t power, he_stream s) {
    if (ctx == nullptr) return invalid_argument("null context");
    const PolyContext& pc = *ctx->impl;
    if (batch == 0) return HE_OK;
    if (in == nullptr || out == nullptr || in == out) return invalid_argument("null or aliased slabs");
    int status = pc.check_device();
    if (status != HE_OK) return status;
    const int64_t twice = 2 * static_cast<int64_t>(pc.degree());
    int64_t shift = power % twice;  // x^(2N) = 1
    if (shift < 0) shift += twice;
    HEAMD_HIP_TRY(heamd::launch_multiply_power_of_x(in, out, pc.device_context(), static_cast<uint32_t>(shift),
                                                    batch * pc.moduli_count(), as_stream(s)));
    return HE_OK;
}

namespace {
// bit widths and row offsets of PolyRq.serialize (PolyRq+Serialize.swift:69-99); HE_OK or the reference's errors
int serialize_layout(const PolyContext& pc, int skip_lsbs, heamd::SerializeLayout& layout) {
    if (pc.moduli_count() > heamd::kMaxSerializedRows) {
        heamd::set_last_error("serialization supports at most 64 residue rows");
        return HE_ERR_UNSUPPORTED;
    }
    layout.rows = pc.moduli_count();
    layout.byte_offset[0] = 0;
    for (uint32_t r = 0; r < layout.rows; ++r) {
        const uint64_t q = pc.moduli()[r];
        int bits = 64 - __builtin_clzll(q);               // significant bits
        if ((q & (q - 1)) == 0) bits -= 1;                // ceilLog2 (ModularArithmetic/Scalar.swift:266-269)
        // CoefficientPacking.validate (CoefficientPacking.swift:27-31)
        if (!(bits > 0 && bits > skip_lsbs && skip_lsbs >= 0)) {
            heamd::set_last_error("invalid coefficient packing: bitsPerCoeff " + std::to_string(bits) + ", skipLSBs " +
                                  std::to_string(skip_lsbs));
            return HE_ERR_INVALID_COEFFICIENT_PACKING;
        }
        layout.width[r] = static_cast<uint32_t>(bits - skip_lsbs);
        layout.byte_offset[r + 1] = layout.byte_offset[r] + (uint64_t(pc.degree()) * layout.width[r] + 7) / 8;
    }
    return HE_OK;
}
}  // namespace

size_t he_poly_serialization_byte_count(const he_poly_context* ctx, int skip_lsbs) {
    if (ctx == nullptr) return 0;
    heamd::SerializeLayout layout{};
    if (serialize_layout(*ctx->impl, skip_lsbs, layout) != HE_OK) return 0;
    return static_cast<size_t>(layout.byte_offset[layout.rows]);
}

int he_poly_serialize_device(const he_poly_context* ctx, const uint64_t* device_slab, size_t batch, int skip_lsbs,
                             uint8_t* device_bytes, he_stream s) {
    if (ctx == nullptr) return invalid_argument("null context");
    const PolyContext& pc = *ctx->impl;
    heamd::SerializeLayout layout{};
    int status = serialize_layout(pc, skip_lsbs, layout);
    if (status != HE_OK) return status;
    if (batch == 0) return HE_OK;
    if (device_slab == nullptr || device_bytes == nullptr) return invalid_argument("null buffer");
    status = pc.check_device();
    if (status != HE_OK) return status;
    HEAMD_HIP_TRY(heamd::launch_serialize(device_slab, device_bytes, layout, pc.device_context().log_degree,
                                          static_cast<uint32_t>(skip_lsbs), batch, as_stream(s)));
    return HE_OK;
}

int he_poly_deserialize_device(const he_poly_context* ctx, const uint8_t* device_bytes, size_t bytes_per_poly,
                               size_t batch, int skip_lsbs, uint64_t* device_slab, he_stream s) {
    if (ctx == nullptr) return invalid_argument("null context");
    const PolyContext& pc = *ctx->impl;
    heamd::SerializeLayout layout{};
    int status = serialize_layout(pc, skip_lsbs, layout);
    if (status != HE_OK) return status;
    if (bytes_per_poly < layout.byte_offset[layout.rows]) {  // PolyRq+Serialize.swift:41-51
        heamd::set_last_error("serialized buffer holds " + std::to_string(bytes_per_poly) + " bytes, expected " +
                              std::to_string(layout.byte_offset[layout.rows]));
        return HE_ERR_SERIALIZED_BUFFER_SIZE_MISMATCH;
    }
    if (batch == 0) return HE_OK;
    if (device_slab == nullptr || device_bytes == nullptr) return invalid_argument("null buffer");
    status = pc.check_device();
    if (status != HE_OK) return status;
    HEAMD_HIP_TRY(heamd::launch_deserialize(device_bytes, device_slab, layout, pc.device_context().log_degree,
                                            static_cast<uint32_t>(skip_lsbs), bytes_per_poly, batch, as_stream(s)));
    return HE_OK;
}

// ------------------------------------------------------------------------------------------ PolyRq<UInt32>
namespace {
int ntt32(const he_poly_context* ctx, uint32_t* slab, size_t batch, bool inverse, he_stream s) {
    if (ctx == nullptr) return invalid_argument("null context");
    const PolyContext& pc = *ctx->impl;
    if (!pc.all_ntt(pc.moduli_count())) return HE_ERR_INVALID_NTT_MODULUS;
    heamd::DeviceContext32 dc{};
    int status = pc.device_context32(pc.moduli_count(), dc);
    if (status != HE_OK) return status;
    if (batch == 0) return HE_OK;
    if (slab == nullptr) return invalid_argument("null slab");
    hipError_t e = heamd::launch_ntt32(inverse, slab, dc, 0, pc.moduli_count(), batch * pc.moduli_count(), as_stream(s));
    if (e == hipErrorNotSupported) {
        heamd::set_last_error("UInt32 transform supports degrees up to 32768");
        return HE_ERR_UNSUPPORTED;
    }
    HEAMD_HIP_TRY(e);
    return HE_OK;
}
int elementwise32(const he_poly_context* ctx, heamd::ElementwiseOp op, uint32_t* lhs, const uint32_t* rhs, size_t batch,
                  he_stream s) {
    if (ctx == nullptr) return invalid_argument("null context");
    const PolyContext& pc = *ctx->impl;
    heamd::DeviceContext32 dc{};
    int status = pc.device_context32(pc.moduli_count(), dc);
    if (status != HE_OK) return status;
    if (batch == 0) return HE_OK;
    if (lhs == nullptr || (rhs == nullptr && op != heamd::ElementwiseOp::Neg)) return invalid_argument("null slab");
    HEAMD_HIP_TRY(heamd::launch_elementwise32(op, lhs, rhs, nullptr, dc, batch * pc.moduli_count(), as_stream(s)));
    return HE_OK;
}
}  // namespace

int he_ntt_forward_device_u32(const he_poly_context* ctx, uint32_t* device_slab, size_t batch, he_stream s) {
    return ntt32(ctx, device_slab, batch, false, s);
}
int he_ntt_inverse_device_u32(const he_poly_context* ctx, uint32_t* device_slab, size_t batch, he_stream s) {
    return ntt32(ctx, device_slab, batch, true, s);
}
int he_poly_add_device_u32(const he_poly_context* ctx, uint32_t* lhs, const uint32_t* rhs, size_t batch, he_stream s) {
    return elementwise32(ctx, heamd::ElementwiseOp::Add, lhs, rhs, batch, s);
}
// ---- word-size bridge for Bfv<UInt32> callers
int he_words_widen_u32_device(const uint32_t* in, uint64_t* out, size_t words, he_stream s) {
    if (words == 0) return HE_OK;
    if (in == nullptr || out == nullptr) return invalid_argument("null slab");
    if ((reinterpret_cast<uintptr_t>(in) & 15u) != 0 || (reinterpret_cast<uintptr_t>(out) & 15u) != 0)
        return invalid_argument("slabs must be 16-byte aligned");
    HEAMD_HIP_TRY(heamd::launch_widen_words(in, out, words, as_stream(s)));
    return HE_OK;
}
int he_words_copy_device(const uint64_t* in, uint64_t* out, size_t words, int non_temporal, he_stream s) {
    if (words == 0) return HE_OK;
    if (in == nullptr || out == nullptr) return invalid_argument("null slab");
    HEAMD_HIP_TRY(heamd::launch_stream_copy(in, out, words, non_temporal != 0, as_stream(s)));
    return HE_OK;
}
int he_words_narrow_u64_device(const uint64_t* in, uint32_t* out, size_t words, he_stream s) {
    if (words == 0) return HE_OK;
    if (in == nullptr || out == nullptr) return invalid_argument("null slab");
    if ((reinterpret_cast<uintptr_t>(in) & 15u) != 0 || (reinterpret_cast<uintptr_t>(out) & 15u) != 0)
        return invalid_argument("slabs must be 16-byte aligned");
    HEAMD_HIP_TRY(heamd::launch_narrow_words(in, out, words, as_stream(s)));
    return HE_OK;
}

int he_poly_sub_device_u32(const he_poly_context* ctx, uint32_t* lhs, const uint32_t* rhs, size_t batch, he_stream s) {
    return elementwise32(ctx, heamd::ElementwiseOp::Sub, lhs, rhs, batch, s);
}
int he_poly_neg_device_u32(const he_poly_context* ctx, uint32_t* data, size_t batch, he_stream s) {
    return elementwise32(ctx, heamd::ElementwiseOp::Neg, data, nullptr, batch, s);
}
int he_poly_mul_device_u32(const he_poly_context* ctx, uint32_t* lhs, const uint32_t* rhs, size_t batch, he_stream s) {
    return elementwise32(ctx, heamd::ElementwiseOp::Mul, lhs, rhs, batch, s);
}
int he_poly_mul_scalar_device_u32(const he_poly_context* ctx, uint32_t* data, const uint32_t* scalar_residues,
                                  size_t batch, he_stream s) {
    if (ctx == nullptr || scalar_residues == nullptr) return invalid_argument("null pointer");
    const PolyContext& pc = *ctx->impl;
    heamd::DeviceContext32 dc{};
    int status = pc.device_context32(pc.moduli_count(), dc);
    if (status != HE_OK) return status;
    if (batch == 0) return HE_OK;
    if (data == nullptr) return invalid_argument("null slab");
    std::vector<uint64_t> pairs(2 * pc.moduli_count());
    for (uint32_t i = 0; i < pc.moduli_count(); ++i) {
        const uint64_t p = pc.moduli()[i];
        if (scalar_residues[i] >= p) return invalid_argument("scalar residue not reduced");
        pairs[2 * i] = scalar_residues[i];
        pairs[2 * i + 1] = heamd::shoup_factor(scalar_residues[i], p);
    }
    hipStream_t stream = as_stream(s);
    Scratch scratch(stream);
    HEAMD_HIP_TRY(scratch.allocate(pairs.size() * sizeof(uint64_t)));
    HEAMD_HIP_TRY(hipMemcpyAsync(scratch.get(), pairs.data(), pairs.size() * sizeof(uint64_t), hipMemcpyHostToDevice,
                                 stream));
    HEAMD_HIP_TRY(hipStreamSynchronize(stream));  // the pageable host vector must outlive the async copy
    HEAMD_HIP_TRY(heamd::launch_elementwise32(heamd::ElementwiseOp::MulScalar, data, nullptr,
                                              static_cast<const uint64_t*>(scratch.get()), dc,
                                              batch * pc.moduli_count(), stream));
    return HE_OK;
}
int he_poly_divide_and_round_q_last_device_u32(const he_poly_context* ctx, const uint32_t* device_in,
                                               uint32_t* device_out, size_t batch, he_stream s) {
    if (ctx == nullptr) return invalid_argument("null context");
    const PolyContext& pc = *ctx->impl;
    if (pc.moduli_count() < 2) return HE_ERR_INVALID_POLY_CONTEXT;  // PolyRq.swift:366-368
    heamd::DeviceContext32 dc{};
    int status = pc.device_context32(pc.moduli_count(), dc);
    if (status != HE_OK) return status;
    if (batch == 0) return HE_OK;
    if (device_in == nullptr || device_out == nullptr) return invalid_argument("null slab");
    HEAMD_HIP_TRY(heamd::launch_divide_and_round_q_last32(device_in, device_out, dc, pc.moduli_count(), batch,
                                                          as_stream(s)));
    return HE_OK;
}

int he_poly_random_from_seeds_device(const he_poly_context* ctx, const uint8_t* device_seeds, size_t batch,
                                     uint64_t* device_slab, he_stream s) {
    if (ctx == nullptr) return invalid_argument("null context");
    const PolyContext& pc = *ctx->impl;
    if (batch == 0) return HE_OK;
    if (device_seeds == nullptr || device_slab == nullptr) return invalid_argument("null buffer");
    int status = pc.check_device();
    if (status != HE_OK) return status;
    heamd::Scratch chain(as_stream(s));
    HEAMD_HIP_TRY(chain.allocate(heamd::seeded_uniform_scratch_bytes(pc.device_context(), batch)));
    HEAMD_HIP_TRY(heamd::launch_seeded_uniform(device_seeds, device_slab, pc.device_context(), batch, chain.get(),
                                               as_stream(s)));
    return HE_OK;
}

int he_poly_mul_scalar_device(const he_poly_context* ctx, uint64_t* data, const uint64_t* scalar_residues,
                              size_t batch, he_stream s) {
    if (ctx == nullptr || scalar_residues == nullptr) return invalid_argument("null pointer");
    const PolyContext& pc = *ctx->impl;
    if (batch == 0) return HE_OK;
    int status = pc.check_device();
    if (status != HE_OK) return status;
    // MultiplyConstantModulus(multiplicand:divisionModulus:) per row (PolyRq.swift:235-238)
    std::vector<heamd::U64x2> pairs(pc.moduli_count());
    for (uint32_t i = 0; i < pc.moduli_count(); ++i) {
        const uint64_t p = pc.moduli()[i];
        if (scalar_residues[i] >= p) return invalid_argument("scalar residue not reduced");
        pairs[i] = heamd::U64x2{scalar_residues[i], heamd::shoup_factor(scalar_residues[i], p)};
    }
    hipStream_t stream = as_stream(s);
    Scratch scratch(stream);
    HEAMD_HIP_TRY(scratch.allocate(pairs.size() * sizeof(heamd::U64x2)));
    HEAMD_HIP_TRY(hipMemcpyAsync(scratch.get(), pairs.data(), pairs.size() * sizeof(heamd::U64x2),
                                 hipMemcpyHostToDevice, stream));
    // the pageable host vector must outlive the async copy
    HEAMD_HIP_TRY(hipStreamSynchronize(stream));
    HEAMD_HIP_TRY(heamd::launch_elementwise(heamd::ElementwiseOp::MulScalar, data,
                                            static_cast<const uint64_t*>(scratch.get()), pc.device_context(),
                                            batch * pc.moduli_count(), stream));
    return HE_OK;
}

int he_poly_divide_and_round_q_last_device(const he_poly_context* ctx, const uint64_t* in, uint64_t* out,
                                           size_t batch, he_stream s) {
    if (ctx == nullptr) return invalid_argument("null context");
    const PolyContext& pc = *ctx->impl;
    if (pc.moduli_count() < 2) return HE_ERR_INVALID_POLY_CONTEXT;  // no next context (PolyRq.swift:366-368)
    if (batch == 0) return HE_OK;
    if (in == nullptr || out == nullptr) return invalid_argument("null slab");
    int status = pc.check_device();
    if (status != HE_OK) return status;
    HEAMD_HIP_TRY(heamd::launch_divide_and_round_q_last(in, out, pc.device_context(), pc.moduli_count(), batch,
                                                        as_stream(s)));
    return HE_OK;
}
int he_poly_divide_and_round_q_last(const he_poly_context* ctx, const uint64_t* host_in, uint64_t* host_out,
                                    size_t batch) {
    if (ctx == nullptr) return invalid_argument("null context");
    const PolyContext& pc = *ctx->impl;
    if (pc.moduli_count() < 2) return HE_ERR_INVALID_POLY_CONTEXT;
    if (batch == 0) return HE_OK;
    if (host_in == nullptr || host_out == nullptr) return invalid_argument("null slab");
    int status = pc.check_device();
    if (status != HE_OK) return status;
    const size_t n = pc.degree(), L = pc.moduli_count();
    const size_t in_words = batch * L * n, out_words = batch * (L - 1) * n;
    hipStream_t stream = hipStreamPerThread;
    void* device = nullptr;
    HEAMD_HIP_TRY(hipMalloc(&device, (in_words + out_words) * sizeof(uint64_t)));
    uint64_t* d_in = static_cast<uint64_t*>(device);
    uint64_t* d_out = d_in + in_words;
    hipError_t e = hipMemcpyAsync(d_in, host_in, in_words * sizeof(uint64_t), hipMemcpyHostToDevice, stream);
    if (e == hipSuccess) e = heamd::launch_divide_and_round_q_last(d_in, d_out, pc.device_context(), L, batch, stream);
    if (e == hipSuccess) e = hipMemcpyAsync(host_out, d_out, out_words * sizeof(uint64_t), hipMemcpyDeviceToHost, stream);
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
    (void)hipFree(device);
    if (e != hipSuccess) return heamd::device_failure(e, "he_poly_divide_and_round_q_last");
    return HE_OK;
}

int he_poly_adding_lazy_product_device(const he_poly_context* ctx, const uint64_t* lhs, const uint64_t* rhs,
                                       uint64_t* acc_lo_hi, he_stream s) {
    if (ctx == nullptr || lhs == nullptr || rhs == nullptr || acc_lo_hi == nullptr) return invalid_argument("null");
    int status = ctx->impl->check_device();
    if (status != HE_OK) return status;
    HEAMD_HIP_TRY(heamd::launch_adding_lazy_product(lhs, rhs, acc_lo_hi, ctx->impl->device_context(), as_stream(s)));
    return HE_OK;
}
int he_poly_reduce_accumulator_device(const he_poly_context* ctx, const uint64_t* acc_lo_hi, uint64_t* out,
                                      he_stream s) {
    if (ctx == nullptr || acc_lo_hi == nullptr || out == nullptr) return invalid_argument("null");
    int status = ctx->impl->check_device();
    if (status != HE_OK) return status;
    HEAMD_HIP_TRY(heamd::launch_reduce_accumulator(acc_lo_hi, out, ctx->impl->device_context(), as_stream(s)));
    return HE_OK;
}

}  // extern "C"

