// bfv_api.cpp -- B3 entry points (Context<Bfv<UInt64>> and the HeScheme operations on the hot path).
// PLACEHOLDER for the first GPU bring-up of the B1/B2 kernels: every entry point reports unsupportedHeOperation.
#include "../../include/he_amd.h"

#include "poly_context.hpp"

struct he_bfv_context {};

static int unsupported(const char* what) {
    heamd::set_last_error(std::string(what) + ": not built yet");
    return HE_ERR_UNSUPPORTED;
}

extern "C" {
int he_bfv_context_create(uint32_t, uint64_t, const uint64_t*, uint32_t, he_bfv_context** out) {
    if (out) *out = nullptr;
    return unsupported("he_bfv_context_create");
}
int he_bfv_context_create_host_only(uint32_t, uint64_t, const uint64_t*, uint32_t, he_bfv_context** out) {
    if (out) *out = nullptr;
    return unsupported("he_bfv_context_create_host_only");
}
void he_bfv_context_destroy(he_bfv_context* ctx) { delete ctx; }
uint32_t he_bfv_ciphertext_moduli_count(const he_bfv_context*) { return 0; }
const he_poly_context* he_bfv_ciphertext_context(const he_bfv_context*, uint32_t) { return nullptr; }
const he_poly_context* he_bfv_key_switching_context(const he_bfv_context*, uint32_t) { return nullptr; }
const he_poly_context* he_bfv_qbsk_context(const he_bfv_context*, uint32_t) { return nullptr; }
int he_bfv_copy_bsk_moduli(const he_bfv_context*, uint64_t*) { return unsupported("he_bfv_copy_bsk_moduli"); }
int he_rns_lift_q_to_qbsk_device(const he_bfv_context*, uint32_t, const uint64_t*, uint64_t*, size_t, he_stream) {
    return unsupported("he_rns_lift_q_to_qbsk_device");
}
int he_rns_floor_qbsk_to_q_device(const he_bfv_context*, uint32_t, const uint64_t*, uint64_t*, size_t, he_stream) {
    return unsupported("he_rns_floor_qbsk_to_q_device");
}
size_t he_bfv_mul_workspace_bytes(const he_bfv_context*, uint32_t, size_t) { return 0; }
size_t he_bfv_relinearize_workspace_bytes(const he_bfv_context*, uint32_t, size_t) { return 0; }
size_t he_bfv_inner_product_workspace_bytes(const he_bfv_context*, uint32_t, size_t) { return 0; }
int he_bfv_mul_device(const he_bfv_context*, uint32_t, const uint64_t*, const uint64_t*, uint64_t*, size_t, void*,
                      size_t, he_stream) {
    return unsupported("he_bfv_mul_device");
}
int he_bfv_relinearize_device(const he_bfv_context*, uint32_t, const uint64_t*, const uint64_t*, uint64_t*, size_t,
                              void*, size_t, he_stream) {
    return unsupported("he_bfv_relinearize_device");
}
int he_bfv_mod_switch_down_device(const he_bfv_context*, uint32_t, uint32_t, const uint64_t*, uint64_t*, size_t,
                                  he_stream) {
    return unsupported("he_bfv_mod_switch_down_device");
}
int he_bfv_mul_plain_device(const he_bfv_context*, uint32_t, uint32_t, uint64_t*, const uint64_t*, size_t, he_stream) {
    return unsupported("he_bfv_mul_plain_device");
}
int he_bfv_inner_product_plain_device(const he_bfv_context*, uint32_t, uint32_t, const uint64_t*, const uint64_t*,
                                      const uint8_t*, size_t, size_t, uint64_t*, he_stream) {
    return unsupported("he_bfv_inner_product_plain_device");
}
int he_bfv_inner_product_device(const he_bfv_context*, uint32_t, const uint64_t*, const uint64_t*, size_t, uint64_t*,
                                void*, size_t, he_stream) {
    return unsupported("he_bfv_inner_product_device");
}
}
