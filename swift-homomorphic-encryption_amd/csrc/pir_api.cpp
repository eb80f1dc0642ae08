// pir_api.cpp -- B4 application hook: the PIR server's per-chunk response kept resident on the device.
//
// PirUtilProtocol.computeResponseForOneChunk (reference Sources/PrivateInformationRetrieval/IndexPir/
// PirUtil.swift:408-486; MulPirServer's twin at IndexPir/MulPir.swift:369-410):
//   1. per database column: Bfv.innerProduct(ciphertexts: dim-0 query, plaintexts: column) (Eval), then canonical
//      (Coeff) format                                                                    PirUtil.swift:428-446
//   2. per remaining dimension d: results <- relinearize(innerProduct(query slice, results slice))   :448-479
//   3. modSwitchDownToSingle                                                                         :481-485
// The reference fans these out over Swift tasks; here every stage is one batched launch (or a short loop of
// launches) of the kernels behind the B3 entry points on the caller's stream -- no stage returns to the host.
#include <vector>

#include "api_internal.hpp"

using heamd::as_stream;
using heamd::invalid_argument;
using heamd::Scratch;

#define HEAMD_TRY_STATUS(expr)            \
    do {                                  \
        const int status_ = (expr);       \
        if (status_ != HE_OK) return status_; \
    } while (0)

extern "C" int he_pir_compute_response_chunk_device(const he_bfv_context* ctx, const uint32_t* dimensions,
                                                    uint32_t dimension_count, const uint64_t* dim0_query_eval,
                                                    const uint64_t* remaining_query, size_t remaining_query_count,
                                                    const uint64_t* database,
                                                    const uint8_t* present, const uint64_t* relinearization_key,
                                                    uint64_t* out, he_stream s) {
    if (ctx == nullptr) return invalid_argument("null context");
    if (dimensions == nullptr || dimension_count == 0) return invalid_argument("empty dimensions");
    if (dim0_query_eval == nullptr || database == nullptr || out == nullptr) return invalid_argument("null operand");
    const uint32_t L = he_bfv_ciphertext_moduli_count(ctx);
    const he_poly_context* q_ctx = he_bfv_ciphertext_context(ctx, L);
    if (q_ctx == nullptr) return invalid_argument("context has no ciphertext level");
    const size_t n = he_poly_context_degree(q_ctx);
    size_t per_chunk = 1, consumed = 0;
    for (uint32_t i = 0; i < dimension_count; ++i) {
        if (dimensions[i] == 0) return invalid_argument("zero dimension");
        per_chunk *= dimensions[i];
        if (i > 0) consumed += dimensions[i];
    }
    const size_t d0 = dimensions[0], columns = per_chunk / d0;
    if (consumed > 0 && remaining_query == nullptr) return invalid_argument("null remaining query");
    // precondition of the reference (PirUtil.swift:422), and the slices taken at :454 must exist
    if (!(columns == 1 || columns == remaining_query_count) || consumed > remaining_query_count)
        return invalid_argument("dimensions do not match the query");
    hipStream_t stream = as_stream(s);
    const size_t poly = size_t(L) * n, ct2 = 2 * poly, ct3 = 3 * poly;

    // results[columns][2][L][N]; `next` receives each dimension's relinearized products; products[items][3][L][N]
    Scratch results_mem(stream), next_mem(stream), products_mem(stream), level_mem(stream);
    HEAMD_HIP_TRY(results_mem.allocate(columns * ct2 * sizeof(uint64_t)));
    uint64_t* results = static_cast<uint64_t*>(results_mem.get());
    // 1. dim-0: every column's ct . pt inner product in one launch, then back to Coeff
    HEAMD_TRY_STATUS(he_bfv_inner_product_plain_device(ctx, L, 2, dim0_query_eval, database, present, d0, columns, results,
                                                       s));
    HEAMD_TRY_STATUS(he_ntt_inverse_device(q_ctx, results, columns * 2, s));
    // 2. remaining dimensions
    size_t count = columns, cursor = 0;
    if (dimension_count > 1) {
        const size_t max_items = columns / dimensions[1];
        HEAMD_HIP_TRY(next_mem.allocate((max_items ? max_items : 1) * ct2 * sizeof(uint64_t)));
        HEAMD_HIP_TRY(products_mem.allocate((max_items ? max_items : 1) * ct3 * sizeof(uint64_t)));
    }
    uint64_t* next = static_cast<uint64_t*>(next_mem.get());
    uint64_t* products = static_cast<uint64_t*>(products_mem.get());
    for (uint32_t i = 1; i < dimension_count; ++i) {
        const size_t d = dimensions[i];
        if (count % d != 0) return invalid_argument("intermediate results do not divide by the dimension");
        const size_t items = count / d;
        const uint64_t* query = remaining_query + cursor * ct2;
        for (size_t j = 0; j < items; ++j)
            HEAMD_TRY_STATUS(he_bfv_inner_product_device(ctx, L, query, results + j * d * ct2, d, products + j * ct3,
                                                         nullptr, 0, s));
        HEAMD_TRY_STATUS(he_bfv_relinearize_device(ctx, L, products, relinearization_key, next, items, nullptr, 0, s));
        // the relinearized products become the next dimension's operands
        HEAMD_HIP_TRY(hipMemcpyAsync(results, next, items * ct2 * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream));
        count = items;
        cursor += d;
    }
    if (count != 1) return invalid_argument("dimensions leave more than one ciphertext");  // PirUtil.swift:481-482
    // 3. modSwitchDownToSingle: L - 1 divideAndRoundQLast steps on the 2-poly ciphertext
    if (L == 1) {
        HEAMD_HIP_TRY(hipMemcpyAsync(out, results, 2 * n * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream));
        return HE_OK;
    }
    HEAMD_HIP_TRY(level_mem.allocate(2 * ct2 * sizeof(uint64_t)));
    uint64_t* ping = static_cast<uint64_t*>(level_mem.get());
    uint64_t* pong = ping + ct2;
    const uint64_t* current = results;
    for (uint32_t level = L; level > 1; --level) {
        uint64_t* target = level == 2 ? out : (current == ping ? pong : ping);
        HEAMD_TRY_STATUS(he_bfv_mod_switch_down_device(ctx, level, 2, current, target, 1, s));
        current = target;
    }
    return HE_OK;
}
