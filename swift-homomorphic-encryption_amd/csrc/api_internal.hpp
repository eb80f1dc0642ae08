// api_internal.hpp -- definitions shared by the C-ABI translation units.
#pragma once

#include <hip/hip_runtime.h>

#include <string>

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

inline int invalid_argument(const char* what) {
    set_last_error(std::string("invalid argument: ") + what);
    return HE_ERR_INVALID_ARGUMENT;
}

// Stream-ordered scratch buffer (hipMallocAsync / hipFreeAsync on the same stream).
class Scratch {
  public:
    explicit Scratch(hipStream_t stream) : stream_(stream) {}
    ~Scratch() {
        if (ptr_ != nullptr) (void)hipFreeAsync(ptr_, stream_);
    }
    Scratch(const Scratch&) = delete;
    Scratch& operator=(const Scratch&) = delete;
    hipError_t allocate(size_t bytes) { return hipMallocAsync(&ptr_, bytes ? bytes : 1, stream_); }
    void* get() const { return ptr_; }

  private:
    hipStream_t stream_;
    void* ptr_ = nullptr;
};

}  // namespace heamd
