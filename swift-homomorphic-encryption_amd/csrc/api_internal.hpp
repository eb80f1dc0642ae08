// api_internal.hpp -- definitions shared by the C-ABI translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/he_amd.h"
#include "poly_context.hpp"

// Opaque handle of include/he_amd.h.  Sub-contexts handed out by a he_bfv_context are non-owning views.
struct he_poly_context {
    heamd::PolyContext* impl;
    bool owned;
    ~he_poly_context() {
        if (owned) delete impl;
    }
};

namespace heamd {

inline hipStream_t as_stream(he_stream s) { return static_cast<hipStream_t>(s); }

// Destroy / free entry points run whenever the host's memory management says so -- a Swift deinit, a Python finaliser -- also
// while some stream of the process is being captured in hipStreamCaptureModeGlobal, where hipFree / hipStreamDestroy /
// hipEventDestroy from ANY thread invalidate that capture (seen: a garbage-collected context of an earlier test finalised inside
// torch.cuda.graph, "operation failed due to a previous error during capture").  The calling thread's capture mode is relaxed
// for the duration of such a call; nothing these calls touch belongs to a capture.
class RelaxedCapture {
  public:
    RelaxedCapture() {
        exchanged_ = hipThreadExchangeStreamCaptureMode(&mode_) == hipSuccess;
        if (!exchanged_) (void)hipGetLastError();
    }
    ~RelaxedCapture() {
        if (exchanged_) (void)hipThreadExchangeStreamCaptureMode(&mode_);
    }
    RelaxedCapture(const RelaxedCapture&) = delete;
    RelaxedCapture& operator=(const RelaxedCapture&) = delete;

  private:
    hipStreamCaptureMode mode_ = hipStreamCaptureModeRelaxed;
    bool exchanged_ = false;
};

inline int invalid_argument(const char* what) {
    set_last_error(std::string("invalid argument: ") + what);
    return HE_ERR_INVALID_ARGUMENT;
}

// Stream-ordered scratch (c_api.cpp): from a memory pool the LIBRARY owns (one per device, release threshold 0 -- never the
// device's default pool, which belongs to the host process and its other HIP users) until the host opts in with
// he_set_scratch_cache(bytes); from then on from the library's own block cache, which hands a stream the blocks it released
// without a driver call.  scratch_release is the only way a block goes back.
hipError_t scratch_allocate(void** out, size_t bytes, hipStream_t stream);
void scratch_release(void* ptr, hipStream_t stream);
void scratch_forget_stream(hipStream_t stream);

// The recursion tree of PirUtil.expand for one (ciphertext count, output count) on one ring: the data movement of every
// level (pir_api.cpp).  Planned on the first use, kept by the context with its table on the device, so that later
// expansions of the same shape neither plan nor copy nor wait.
struct ExpandPlan {
    struct Level {
        size_t nodes = 0, leaf_offset = 0, leaf_count = 0, parent_offset = 0, parent_count = 0;
        bool gather = false;
    };
    std::vector<Level> levels;
    size_t widest = 0;
    uint32_t* table_device = nullptr;
    ExpandPlan() = default;
    ExpandPlan(const ExpandPlan&) = delete;
    ExpandPlan& operator=(const ExpandPlan&) = delete;
    ~ExpandPlan() {
        if (table_device != nullptr) (void)hipFree(table_device);
    }
};
class ExpandPlanCache {
  public:
    using Key = std::pair<size_t, size_t>;  // (ciphertext count, output count)
    std::shared_ptr<const ExpandPlan> find(const Key& key) {
        std::lock_guard<std::mutex> lock(mutex_);
        auto it = plans_.find(key);
        return it == plans_.end() ? nullptr : it->second;
    }
    std::shared_ptr<const ExpandPlan> insert(const Key& key, std::shared_ptr<const ExpandPlan> plan) {
        std::lock_guard<std::mutex> lock(mutex_);
        if (plans_.size() >= 64) plans_.clear();  // shapes in use are few; plans still referenced live on with their users
        return plans_.emplace(key, std::move(plan)).first->second;
    }

  private:
    std::mutex mutex_;
    std::map<Key, std::shared_ptr<const ExpandPlan>> plans_;
};
// bfv_api.cpp: the cache a context owns
ExpandPlanCache& expand_plans(const he_bfv_context* ctx);

// bfv_api.cpp: one PirUtil.expand level through the fused Galois key switch (not exported)
constexpr int kExpandStepUnavailable = -1;
// `rotated`: what the key switch rotates when it is not the parents themselves -- the parents already taken through the
// level's element all but one time (a level whose element is reached by repeated application); nullptr: the parents.
int bfv_expand_step_fused(const he_bfv_context* ctx, uint32_t L, const uint64_t* parents, uint64_t element,
                          const uint64_t* const* keys, size_t groups, size_t group_size, uint64_t* next, uint32_t shift,
                          const uint32_t* leaf_table, size_t leaf_stride, void* workspace, size_t workspace_bytes,
                          hipStream_t stream, const uint64_t* rotated = nullptr);

// bfv_api.cpp: `items` ct x ct inner products that share lhs, the right-hand side given in Eval form over Q (the first remaining
// dimension of a PIR response; not exported)
constexpr int kInnerProductEvalUnavailable = -2;
int bfv_inner_product_shared_eval_rhs(const he_bfv_context* ctx, uint32_t L, const uint64_t* lhs, const uint64_t* rhs_eval,
                                      size_t count, size_t items, uint64_t* out, hipStream_t stream);

// pir_api.cpp: he_pir_dim0_columns_device without its last step (the results stay in Eval form) and
// he_pir_remaining_dimensions_chunks_device for such results (not exported)
int pir_dim0_columns_eval(const he_bfv_context* ctx, const uint64_t* dim0_query_eval, size_t d0, const uint64_t* database,
                          const uint8_t* present_device, size_t columns, uint64_t* out, he_stream s);
int pir_remaining_dimensions_chunks_eval(const he_bfv_context* ctx, const uint32_t* dimensions, uint32_t dimension_count,
                                         size_t chunk_count, uint64_t* intermediate_eval, const uint64_t* remaining_query,
                                         size_t remaining_query_count, const uint64_t* relinearization_key, uint64_t* out,
                                         he_stream s);

// Stream-ordered scratch buffer (scratch_allocate / scratch_release on the same stream).
class Scratch {
  public:
    explicit Scratch(hipStream_t stream) : stream_(stream) {}
    ~Scratch() {
        if (ptr_ != nullptr) scratch_release(ptr_, stream_);
    }
    Scratch(const Scratch&) = delete;
    Scratch& operator=(const Scratch&) = delete;
    hipError_t allocate(size_t bytes) { return scratch_allocate(&ptr_, bytes ? bytes : 1, stream_); }
    void* get() const { return ptr_; }

  private:
    hipStream_t stream_;
    void* ptr_ = nullptr;
};

}  // namespace heamd
