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

// ---------------------------------------------------------------------------------------------------------------------
// PirUtil.expand (PirUtil.swift:196-355): oblivious expansion of query ciphertexts into encrypted selection bits.
// The reference recurses per ciphertext (expandCiphertext -> expandCiphertextForOneStep); every node of one tree
// level uses the same Galois element and shift, so here a level is ONE batch: applyGalois (x count), ct - c1,
// multiplyPowerOfX(-2^(logStep-1)), c1 + ct over all nodes of all trees at that depth.  The output order is the
// recursion's (interleave of the two halves, PirUtil.swift:297-299), computed on the host as a plan.
namespace {

struct ExpandNode {
    size_t output_count;
    int expected_height;  // of its tree
    long child0 = -1, child1 = -1;  // indices into the next level
    long leaf_slot = -1;            // position in the final output when output_count == 1
};

int floor_log2_size(size_t x) {
    int r = 0;
    while (x >>= 1) ++r;
    return r;
}
int ceil_log2_size(size_t x) { return floor_log2_size(x) + ((x & (x - 1)) == 0 ? 0 : 1); }

// leaf order of the subtree rooted at levels[depth][index]: PirUtil.swift:297-299
void leaf_order(const std::vector<std::vector<ExpandNode>>& levels, size_t depth, size_t index,
                std::vector<std::pair<size_t, size_t>>& out) {
    const ExpandNode& node = levels[depth][index];
    if (node.output_count == 1) {
        out.emplace_back(depth, index);
        return;
    }
    std::vector<std::pair<size_t, size_t>> first, second;
    leaf_order(levels, depth + 1, static_cast<size_t>(node.child0), first);
    leaf_order(levels, depth + 1, static_cast<size_t>(node.child1), second);
    for (size_t k = 0; k < second.size(); ++k) {
        out.push_back(first[k]);
        out.push_back(second[k]);
    }
    for (size_t k = second.size(); k < first.size(); ++k) out.push_back(first[k]);
}

}  // namespace

extern "C" int he_pir_expand_device(const he_bfv_context* ctx, const uint64_t* ciphertexts, size_t ciphertext_count,
                                    size_t output_count, const uint64_t* galois_elements,
                                    const uint64_t* const* galois_keys, size_t galois_key_count, uint64_t* out,
                                    he_stream s) {
    if (ctx == nullptr) return invalid_argument("null context");
    const uint32_t L = he_bfv_ciphertext_moduli_count(ctx);
    const he_poly_context* q_ctx = he_bfv_ciphertext_context(ctx, L);
    if (q_ctx == nullptr) return invalid_argument("context has no ciphertext level");
    const size_t n = he_poly_context_degree(q_ctx);
    // preconditions of PirUtil.expand (PirUtil.swift:325-326)
    if (ciphertext_count == 0 || !((ciphertext_count - 1) * n < output_count && ciphertext_count * n >= output_count))
        return invalid_argument("output count does not match the number of query ciphertexts");
    if (ciphertexts == nullptr || out == nullptr) return invalid_argument("null ciphertexts");
    if (galois_key_count > 0 && (galois_elements == nullptr || galois_keys == nullptr))
        return invalid_argument("null Galois keys");
    hipStream_t stream = as_stream(s);
    const size_t ct_words = 2 * size_t(L) * n, ct_bytes = ct_words * sizeof(uint64_t);
    const int log_degree = floor_log2_size(n);

    // ---- plan: the recursion tree of every input ciphertext, level by level
    std::vector<std::vector<ExpandNode>> levels(1);
    size_t remaining = output_count;
    for (size_t i = 0; i < ciphertext_count; ++i) {
        const size_t to_generate = remaining < n ? remaining : n;
        remaining -= to_generate;
        ExpandNode root;
        root.output_count = to_generate;
        root.expected_height = ceil_log2_size(to_generate);
        levels[0].push_back(root);
    }
    for (size_t depth = 0; depth < levels.size(); ++depth) {
        std::vector<ExpandNode> next;
        for (ExpandNode& node : levels[depth]) {
            if (node.output_count <= 1) continue;
            const size_t second = node.output_count >> 1, first = node.output_count - second;
            node.child0 = static_cast<long>(next.size());
            next.push_back(ExpandNode{first, node.expected_height});
            node.child1 = static_cast<long>(next.size());
            next.push_back(ExpandNode{second, node.expected_height});
        }
        if (!next.empty()) levels.push_back(std::move(next));
    }
    {
        std::vector<std::pair<size_t, size_t>> order;
        for (size_t i = 0; i < levels[0].size(); ++i) leaf_order(levels, 0, i, order);
        if (order.size() != output_count) return invalid_argument("expansion plan does not cover the outputs");
        for (size_t slot = 0; slot < order.size(); ++slot)
            levels[order[slot].first][order[slot].second].leaf_slot = static_cast<long>(slot);
    }

    // ---- execute
    size_t widest = 0;
    for (const auto& level : levels) widest = level.size() > widest ? level.size() : widest;
    Scratch cur_mem(stream), next_mem(stream), parent_mem(stream), rotated_mem(stream), tmp_mem(stream);
    HEAMD_HIP_TRY(cur_mem.allocate(widest * ct_bytes));
    HEAMD_HIP_TRY(next_mem.allocate(widest * ct_bytes));
    HEAMD_HIP_TRY(parent_mem.allocate(widest * ct_bytes));
    HEAMD_HIP_TRY(rotated_mem.allocate(widest * ct_bytes));
    HEAMD_HIP_TRY(tmp_mem.allocate(widest * ct_bytes));
    uint64_t* cur = static_cast<uint64_t*>(cur_mem.get());
    uint64_t* next = static_cast<uint64_t*>(next_mem.get());
    uint64_t* parents = static_cast<uint64_t*>(parent_mem.get());
    uint64_t* rotated = static_cast<uint64_t*>(rotated_mem.get());
    uint64_t* tmp = static_cast<uint64_t*>(tmp_mem.get());
    HEAMD_HIP_TRY(hipMemcpyAsync(cur, ciphertexts, ciphertext_count * ct_bytes, hipMemcpyDeviceToDevice, stream));
    for (size_t depth = 0; depth < levels.size(); ++depth) {
        const int log_step = static_cast<int>(depth) + 1;
        const std::vector<ExpandNode>& level = levels[depth];
        // leaves of this level go straight to their output slot (PirUtil.swift:262-268)
        std::vector<size_t> internal;
        for (size_t i = 0; i < level.size(); ++i) {
            if (level[i].output_count != 1) {
                internal.push_back(i);
                continue;
            }
            uint64_t* dst = out + static_cast<size_t>(level[i].leaf_slot) * ct_words;
            HEAMD_HIP_TRY(hipMemcpyAsync(dst, cur + i * ct_words, ct_bytes, hipMemcpyDeviceToDevice, stream));
            if (!(log_step > level[i].expected_height))
                HEAMD_TRY_STATUS(he_poly_add_device(q_ctx, dst, cur + i * ct_words, 2, s));  // output += ciphertext
        }
        if (internal.empty()) continue;
        if (log_step > log_degree) return invalid_argument("logStep exceeds log2(degree)");  // precondition :212
        const size_t batch = internal.size();
        for (size_t k = 0; k < batch; ++k)  // gather the parents contiguously
            HEAMD_HIP_TRY(hipMemcpyAsync(parents + k * ct_words, cur + internal[k] * ct_words, ct_bytes,
                                         hipMemcpyDeviceToDevice, stream));
        // expandCiphertextForOneStep (PirUtil.swift:204-236)
        const uint64_t target = (uint64_t(1) << (log_degree - log_step + 1)) + 1;
        long best = -1;
        for (size_t k = 0; k < galois_key_count; ++k)
            if (galois_elements[k] <= target && (best < 0 || galois_elements[k] > galois_elements[best]))
                best = static_cast<long>(k);
        if (best < 0 || galois_keys[best] == nullptr) {
            heamd::set_last_error("no Galois element <= " + std::to_string(target) + " in the evaluation key");
            return HE_ERR_MISSING_GALOIS_KEY;
        }
        const uint64_t element = galois_elements[best];
        const int applications = 1 << (floor_log2_size(target - 1) - floor_log2_size(element - 1));
        const uint64_t* source = parents;
        for (int a = 0; a < applications; ++a) {  // c1.applyGalois(element) repeatedly until x -> x^target
            uint64_t* dst = (a % 2 == 0) ? rotated : tmp;
            HEAMD_TRY_STATUS(he_bfv_apply_galois_device(ctx, L, source, element, galois_keys[best], dst, batch, nullptr, 0,
                                                        s));
            source = dst;
        }
        uint64_t* c1 = const_cast<uint64_t*>(source);
        uint64_t* difference = (c1 == rotated) ? tmp : rotated;
        // difference = (ciphertext - c1) * x^(-2^(logStep-1)); c1 += ciphertext
        HEAMD_HIP_TRY(hipMemcpyAsync(next, parents, batch * ct_bytes, hipMemcpyDeviceToDevice, stream));
        HEAMD_TRY_STATUS(he_poly_sub_device(q_ctx, next, c1, batch * 2, s));
        HEAMD_TRY_STATUS(he_poly_multiply_power_of_x_device(q_ctx, next, difference, batch * 2,
                                                            -(int64_t(1) << (log_step - 1)), s));
        HEAMD_TRY_STATUS(he_poly_add_device(q_ctx, c1, parents, batch * 2, s));
        // children interleaved as the plan numbered them: child0 = c1 (p0), child1 = difference (p1)
        for (size_t k = 0; k < batch; ++k) {
            const ExpandNode& node = level[internal[k]];
            HEAMD_HIP_TRY(hipMemcpyAsync(next + static_cast<size_t>(node.child0) * ct_words, c1 + k * ct_words, ct_bytes,
                                         hipMemcpyDeviceToDevice, stream));
        }
        // `next` held the subtraction result until multiplyPowerOfX consumed it; child0 copies above overwrite it
        // only after that kernel (stream order), child1 copies come from `difference`
        for (size_t k = 0; k < batch; ++k) {
            const ExpandNode& node = level[internal[k]];
            HEAMD_HIP_TRY(hipMemcpyAsync(next + static_cast<size_t>(node.child1) * ct_words, difference + k * ct_words,
                                         ct_bytes, hipMemcpyDeviceToDevice, stream));
        }
        uint64_t* swap = cur;
        cur = next;
        next = swap;
    }
    return HE_OK;
}
