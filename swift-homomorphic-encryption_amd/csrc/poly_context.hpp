// poly_context.hpp -- host-side mirror of PolyContext<UInt64>
// (reference Sources/HomomorphicEncryption/PolyRq/PolyContext.swift:19-123) plus its device-resident image.
#pragma once

#include <hip/hip_runtime.h>

#include <memory>
#include <mutex>
#include <string>
#include <vector>

#include "device_context.hpp"
#include "host_math.hpp"

namespace heamd {

// thread-local detail string behind he_last_error_message()
void set_last_error(const std::string& message);
const char* last_error();
// Maps a HIP failure to HE_ERR_DEVICE, recording hipGetErrorString.
int device_failure(hipError_t error, const char* where);

#define HEAMD_HIP_TRY(expr)                                                  \
    do {                                                                     \
        hipError_t heamd_hip_status_ = (expr);                               \
        if (heamd_hip_status_ != hipSuccess) return ::heamd::device_failure(heamd_hip_status_, #expr); \
    } while (0)

class PolyContext {
  public:
    // PolyContext.init(degree:moduli:) (PolyContext.swift:131-141) with the designated initialiser's checks run
    // prefix by prefix (PolyContext.swift:45-123).  Returns an he_status.
    // host_only = build and keep the host-side precomputation but upload nothing (every compute entry point then
    // fails with HE_ERR_DEVICE); used to check the setup math where no GPU exists.
    static int create(uint32_t degree, const uint64_t* moduli, uint32_t moduli_count,
                      std::unique_ptr<PolyContext>& out, bool host_only = false);
    ~PolyContext();
    PolyContext(const PolyContext&) = delete;
    PolyContext& operator=(const PolyContext&) = delete;

    uint32_t degree() const { return degree_; }
    uint32_t log_degree() const { return log_degree_; }
    uint32_t moduli_count() const { return static_cast<uint32_t>(moduli_.size()); }
    const std::vector<u64>& moduli() const { return moduli_; }
    const std::vector<DeviceModulus>& host_constants() const { return host_moduli_; }
    const U64x2* host_forward_twiddles(uint32_t rns_index) const { return host_forward_.data() + size_t(rns_index) * degree_; }
    const U64x2* host_inverse_twiddles(uint32_t rns_index) const { return host_inverse_.data() + size_t(rns_index) * degree_; }
    const U64x2* host_inverse_q_last(uint32_t last) const { return host_inverse_q_last_.data() + size_t(last) * moduli_.size(); }
    bool host_only() const { return device_block_ == nullptr; }
    int device() const { return device_; }
    // validateNttModuli (PolyContext.swift:175-181) for the first `count` moduli
    bool all_ntt(uint32_t count) const;
    int modulus_index(u64 modulus) const;  // -1 when absent
    u64 max_lazy_product_accumulation_count(uint32_t count) const;  // PolyContext.swift:246-253
    // Device image restricted to the first `count` moduli (the chain element PolyContext.getContext(moduliCount:)).
    DeviceContext device_context(uint32_t count) const;
    DeviceContext device_context() const { return device_context(moduli_count()); }
    // true when the caller's current HIP device is the one this context lives on
    int check_device() const;
    // PolyRq<UInt32>: every modulus <= 2^30 - 1 (ModularArithmetic/Scalar.swift:498-511).  The 4-byte-word device
    // image (32-bit twiddle pairs) is built on first use.  HE_OK / HE_ERR_INVALID_MODULUS / HE_ERR_DEVICE.
    int device_context32(uint32_t count, DeviceContext32& out) const;

  private:
    PolyContext() = default;
    int upload();

    uint32_t degree_ = 0, log_degree_ = 0;
    std::vector<u64> moduli_;
    std::vector<DeviceModulus> host_moduli_;
    std::vector<U64x2> host_forward_, host_inverse_, host_inverse_q_last_;
    int device_ = -1;
    DeviceContext dev_{};
    void* device_block_ = nullptr;
    mutable std::mutex word32_lock_;
    mutable void* device_block32_ = nullptr;
    mutable DeviceContext32 dev32_{};
};

// N^-1 and N^-1 psi^(-N/2) of a modulus with every derived constant (64-bit and limb-wise Shoup forms)
void set_inverse_degree_constants(DeviceModulus& m, u64 inverse_degree, u64 inverse_degree_root);

// Validation of one chain element (designated init) -- shared with the BFV context builder.
int validate_poly_context_prefix(uint32_t degree, const uint64_t* moduli, uint32_t count, bool has_next);

}  // namespace heamd
