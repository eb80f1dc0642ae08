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

#include <cstdlib>
#include <memory>

#include "api_internal.hpp"
#include "kernels.hpp"
#include "side_lane.hpp"

using heamd::as_stream;
using heamd::invalid_argument;
using heamd::Scratch;
using heamd::SideLane;

#define HEAMD_TRY_STATUS(expr)            \
    do {                                  \
        const int status_ = (expr);       \
        if (status_ != HE_OK) return status_; \
    } while (0)

namespace {

struct ChunkShape {
    uint32_t L = 0;
    size_t n = 0, d0 = 0, columns = 0, per_chunk = 0, consumed = 0;
    const he_poly_context* q_ctx = nullptr;
};

// the reference's preconditions on (dimensions, query) -- PirUtil.swift:420-422 and the slices taken at :454
int chunk_shape(const he_bfv_context* ctx, const uint32_t* dimensions, uint32_t dimension_count,
                const uint64_t* remaining_query, size_t remaining_query_count, ChunkShape& shape) {
    if (ctx == nullptr) return invalid_argument("null context");
    if (dimensions == nullptr || dimension_count == 0) return invalid_argument("empty dimensions");
    shape.L = he_bfv_ciphertext_moduli_count(ctx);
    shape.q_ctx = he_bfv_ciphertext_context(ctx, shape.L);
    if (shape.q_ctx == nullptr) return invalid_argument("context has no ciphertext level");
    shape.n = he_poly_context_degree(shape.q_ctx);
    shape.per_chunk = 1;
    for (uint32_t i = 0; i < dimension_count; ++i) {
        if (dimensions[i] == 0) return invalid_argument("zero dimension");
        shape.per_chunk *= dimensions[i];
        if (i > 0) shape.consumed += dimensions[i];
    }
    shape.d0 = dimensions[0];
    shape.columns = shape.per_chunk / shape.d0;
    if (shape.consumed > 0 && remaining_query == nullptr) return invalid_argument("null remaining query");
    if (!(shape.columns == 1 || shape.columns == remaining_query_count) || shape.consumed > remaining_query_count)
        return invalid_argument("dimensions do not match the query");
    return HE_OK;
}

}  // namespace

namespace {
// every column's ct . pt inner product in one launch (PirUtil.swift:428-437), then back to Coeff (:438); packed: the
// database holds packed plaintexts (he_amd.h he_bfv_pack_plaintexts_device)
int dim0_columns(const he_bfv_context* ctx, const uint64_t* dim0_query_eval, size_t d0, const uint64_t* database, bool packed,
                 const uint8_t* present_device, size_t columns, uint64_t* out, he_stream s, bool leave_in_eval = false) {
    if (ctx == nullptr) return invalid_argument("null context");
    if (columns == 0) return HE_OK;
    if (dim0_query_eval == nullptr || database == nullptr || out == nullptr) return invalid_argument("null operand");
    const uint32_t L = he_bfv_ciphertext_moduli_count(ctx);
    const he_poly_context* q_ctx = he_bfv_ciphertext_context(ctx, L);
    if (q_ctx == nullptr) return invalid_argument("context has no ciphertext level");
    if (packed)
        HEAMD_TRY_STATUS(he_bfv_inner_product_plain_packed_device(ctx, L, 2, dim0_query_eval, database, present_device, d0,
                                                                  columns, out, s));
    else
        HEAMD_TRY_STATUS(he_bfv_inner_product_plain_resident_device(ctx, L, 2, dim0_query_eval, database, present_device,
                                                                    d0, columns, out, s));
    if (leave_in_eval) return HE_OK;  // (remaining_dimensions takes them as they are: results_in_eval)
    return he_ntt_inverse_device(q_ctx, out, columns * 2, s);
}
}  // namespace

extern "C" int he_pir_dim0_columns_device(const he_bfv_context* ctx, const uint64_t* dim0_query_eval, size_t d0,
                                          const uint64_t* database, const uint8_t* present_device, size_t columns,
                                          uint64_t* out, he_stream s) {
    return dim0_columns(ctx, dim0_query_eval, d0, database, false, present_device, columns, out, s);
}
extern "C" int he_pir_dim0_columns_packed_device(const he_bfv_context* ctx, const uint64_t* dim0_query_eval, size_t d0,
                                                 const uint64_t* packed_database, const uint8_t* present_device,
                                                 size_t columns, uint64_t* out, he_stream s) {
    return dim0_columns(ctx, dim0_query_eval, d0, packed_database, true, present_device, columns, out, s);
}

namespace {
// PirUtil.swift:448-485 for `chunks` chunks at once: intermediate [chunks][columns][2][L][N] Coeff (consumed) ->
// out [chunks][2][1][N].  The result groups of all chunks share each dimension's query slice, so every stage of a
// dimension is one batch over chunks x groups.
int remaining_dimensions(const he_bfv_context* ctx, const uint32_t* dimensions, uint32_t dimension_count,
                         const ChunkShape& shape, size_t chunks, uint64_t* results, const uint64_t* remaining_query,
                         const uint64_t* relinearization_key, uint64_t* out, he_stream s, bool results_in_eval = false) {
    hipStream_t stream = as_stream(s);
    const uint32_t L = shape.L;
    const size_t n = shape.n, poly = size_t(L) * n, ct2 = 2 * poly, ct3 = 3 * poly;
    Scratch next_mem(stream), products_mem(stream);
    size_t count = shape.columns, cursor = 0;  // ciphertexts per chunk
    if (dimension_count > 1) {
        const size_t max_items = chunks * (shape.columns / dimensions[1]);
        HEAMD_HIP_TRY(next_mem.allocate((max_items ? max_items : 1) * ct2 * sizeof(uint64_t)));
        HEAMD_HIP_TRY(products_mem.allocate((max_items ? max_items : 1) * ct3 * sizeof(uint64_t)));
    }
    // the relinearized products of a dimension become the next one's operands: the two slabs swap roles
    uint64_t* current = results;
    uint64_t* other = static_cast<uint64_t*>(next_mem.get());
    uint64_t* products = static_cast<uint64_t*>(products_mem.get());
    for (uint32_t i = 1; i < dimension_count; ++i) {
        const size_t d = dimensions[i];
        if (count % d != 0) return invalid_argument("intermediate results do not divide by the dimension");
        const size_t items = chunks * (count / d);  // groups of d consecutive results, chunk after chunk
        const uint64_t* query = remaining_query + cursor * ct2;
        // results_in_eval: the dim-0 inner products came as their kernel left them, in Eval form -- the first dimension's ct x ct
        // inner products then read those words as the Q rows of their lifted operands instead of transforming them back and
        // forth (the tail of 8 chunks of 256 x 64: 0.91 -> 0.74 ms, profiles/r06y_pir_tail_eval_rows.txt)
        int status = heamd::kInnerProductEvalUnavailable;
        if (results_in_eval && i == 1)
            status = heamd::bfv_inner_product_shared_eval_rhs(ctx, L, query, current, d, items, products, stream);
        if (status == heamd::kInnerProductEvalUnavailable) {
            if (results_in_eval && i == 1) {  // PirUtil.swift:438
                const he_poly_context* q_ctx = he_bfv_ciphertext_context(ctx, L);
                HEAMD_TRY_STATUS(he_ntt_inverse_device(q_ctx, current, chunks * count * 2, s));
            }
            status = he_bfv_inner_product_shared_device(ctx, L, query, current, d, items, products, s);
        }
        HEAMD_TRY_STATUS(status);
        HEAMD_TRY_STATUS(he_bfv_relinearize_device(ctx, L, products, relinearization_key, other, items, nullptr, 0, s));
        uint64_t* swap = current;
        current = other;
        other = swap;
        count /= d;
        cursor += d;
    }
    if (count != 1) return invalid_argument("dimensions leave more than one ciphertext");  // PirUtil.swift:481-482
    if (results_in_eval && dimension_count == 1) {  // PirUtil.swift:438 with nothing after it
        const he_poly_context* q_ctx = he_bfv_ciphertext_context(ctx, L);
        HEAMD_TRY_STATUS(he_ntt_inverse_device(q_ctx, current, chunks * 2, s));
    }
    // modSwitchDownToSingle (:483) on the chunks' 2-polynomial ciphertexts
    return he_bfv_mod_switch_down_to_single_device(ctx, L, 2, current, out, chunks, s);
}

// The chunks of a response are independent, and the reference answers them as concurrent tasks (PirUtil.swift:538-563):
// while one task streams its chunk of the database another is in its ct x ct stage.  The device counterpart -- the chunks in
// pieces, a piece's dim-0 pass over the database on the caller's stream and its remaining dimensions on a lane of the context
// beside the NEXT piece's dim-0 pass -- is built and tested (tests/test_gpu_pir.py) and OFF: measured on 8 chunks of 256 x 64
// it is slower than one dim-0 launch over all chunks followed by one batch per remaining stage (6.46 ms; pieces of 4 / 2 / 1
// chunks 6.91 / 7.32 / 8.07 ms, the lane at either priority: profiles/r06_pir_overlap.txt).  Why: the dim-0 kernel is not
// idle on the multiplier side -- on 224 of the 256 CUs it already loses 6 % (profiles/r06f_cu_mask_probe.txt) -- and the
// remaining dimensions' 1024-lane transform workgroups only get a CU once the dim-0 kernel's queued 256-lane workgroups stop
// refilling it (a 129 us transform takes 1.3-2.5 ms beside it), so the lane's work mostly runs after the pass it was meant to
// hide under, on smaller, less efficient launches.  HEAMD_PIR_PIECE_CHUNKS=k forces pieces of k chunks (the sweep and the
// tests); unset, a response is one piece.
size_t overlap_piece_chunks(uint32_t dimension_count, size_t chunks) {
    if (dimension_count < 2 || chunks < 2) return chunks;
    if (const char* forced = std::getenv("HEAMD_PIR_PIECE_CHUNKS")) {
        const size_t piece = static_cast<size_t>(std::strtoull(forced, nullptr, 10));
        return piece == 0 || piece > chunks ? chunks : piece;
    }
    return chunks;
}

// `chunks` chunks with a device-resident mask, enqueue-only: per piece one dim-0 launch over the columns of its chunks, then
// the remaining dimensions of those chunks together (beside the next piece's dim-0 launch)
int response_chunks_resident(const he_bfv_context* ctx, const uint32_t* dimensions, uint32_t dimension_count,
                             const ChunkShape& shape, size_t chunks, const uint64_t* dim0_query_eval,
                             const uint64_t* remaining_query, const uint64_t* database, size_t chunk_words, bool packed,
                             const uint8_t* present_device, const uint64_t* relinearization_key, uint64_t* out,
                             he_stream s) {
    hipStream_t stream = as_stream(s);
    const size_t ct2 = 2 * size_t(shape.L) * shape.n, out_words = 2 * shape.n;
    Scratch results_mem(stream);
    HEAMD_HIP_TRY(results_mem.allocate(chunks * shape.columns * ct2 * sizeof(uint64_t)));
    uint64_t* results = static_cast<uint64_t*>(results_mem.get());
    const size_t piece = overlap_piece_chunks(dimension_count, chunks);
    if (piece >= chunks) {
        // a chunk is [columns][d0] plaintexts and the chunks are contiguous: all their columns form one column range
        HEAMD_TRY_STATUS(dim0_columns(ctx, dim0_query_eval, shape.d0, database, packed, present_device, chunks * shape.columns,
                                      results, s, true));
        return remaining_dimensions(ctx, dimensions, dimension_count, shape, chunks, results, remaining_query,
                                    relinearization_key, out, s, true);
    }
    heamd::LaneLease lease(heamd::lane_pool(ctx), stream);
    if (lease.lane == nullptr) {
        // a chunk is [columns][d0] plaintexts and the chunks are contiguous: all their columns form one column range
        HEAMD_TRY_STATUS(dim0_columns(ctx, dim0_query_eval, shape.d0, database, packed, present_device, chunks * shape.columns,
                                      results, s));
        return remaining_dimensions(ctx, dimensions, dimension_count, shape, chunks, results, remaining_query,
                                    relinearization_key, out, s);
    }
    SideLane& lane = *lease.lane;
    int status = HE_OK;
    hipError_t e = hipSuccess;
    bool forked = false;
    for (size_t first = 0; first < chunks && status == HE_OK && e == hipSuccess; first += piece) {
        const size_t now = chunks - first < piece ? chunks - first : piece;
        uint64_t* piece_results = results + first * shape.columns * ct2;
        status = dim0_columns(ctx, dim0_query_eval, shape.d0, database + first * chunk_words, packed,
                              present_device ? present_device + first * shape.per_chunk : nullptr, now * shape.columns,
                              piece_results, s);
        if (status != HE_OK) break;
        e = hipEventRecord(lane.stage[0], stream);  // (re-recorded per piece: a wait takes the event as it is when enqueued)
        if (e == hipSuccess) e = hipStreamWaitEvent(lane.stream, lane.stage[0], 0);
        if (e != hipSuccess) break;
        forked = true;
        status = remaining_dimensions(ctx, dimensions, dimension_count, shape, now, piece_results, remaining_query,
                                      relinearization_key, out + first * out_words, static_cast<he_stream>(lane.stream));
    }
    if (forked) {  // the join is enqueued whatever happened in between: the caller's stream never runs ahead of the lane's work
        const hipError_t recorded = hipEventRecord(lane.joined, lane.stream);
        const hipError_t waited = recorded == hipSuccess ? hipStreamWaitEvent(stream, lane.joined, 0) : recorded;
        if (e == hipSuccess) e = waited;
    }
    if (status != HE_OK) return status;
    HEAMD_HIP_TRY(e);
    return HE_OK;
}
}  // namespace

extern "C" int he_pir_remaining_dimensions_device(const he_bfv_context* ctx, const uint32_t* dimensions,
                                                  uint32_t dimension_count, uint64_t* intermediate,
                                                  const uint64_t* remaining_query, size_t remaining_query_count,
                                                  const uint64_t* relinearization_key, uint64_t* out, he_stream s) {
    ChunkShape shape;
    HEAMD_TRY_STATUS(chunk_shape(ctx, dimensions, dimension_count, remaining_query, remaining_query_count, shape));
    if (intermediate == nullptr || out == nullptr) return invalid_argument("null operand");
    return remaining_dimensions(ctx, dimensions, dimension_count, shape, 1, intermediate, remaining_query,
                                relinearization_key, out, s);
}

// The remaining dimensions of `chunk_count` chunks at once (the chunk loop's second half, for callers that produced the dim-0
// results themselves -- a device group, device_group.cpp): intermediate [chunk][columns][2][L][N] Coeff (consumed) ->
// out [chunk][2][1][N]; every stage one batch over the result groups of all chunks.
extern "C" int he_pir_remaining_dimensions_chunks_device(const he_bfv_context* ctx, const uint32_t* dimensions,
                                                         uint32_t dimension_count, size_t chunk_count, uint64_t* intermediate,
                                                         const uint64_t* remaining_query, size_t remaining_query_count,
                                                         const uint64_t* relinearization_key, uint64_t* out, he_stream s) {
    ChunkShape shape;
    HEAMD_TRY_STATUS(chunk_shape(ctx, dimensions, dimension_count, remaining_query, remaining_query_count, shape));
    if (chunk_count == 0) return HE_OK;
    if (intermediate == nullptr || out == nullptr) return invalid_argument("null operand");
    return remaining_dimensions(ctx, dimensions, dimension_count, shape, chunk_count, intermediate, remaining_query,
                                relinearization_key, out, s);
}

// The two halves for a caller that joins them itself (a device group: device_group.cpp) with the dim-0 results passed on in
// EVAL form, as response_chunks_resident does inside one device (api_internal.hpp; not exported)
int heamd::pir_dim0_columns_eval(const he_bfv_context* ctx, const uint64_t* dim0_query_eval, size_t d0, const uint64_t* database,
                                 const uint8_t* present_device, size_t columns, uint64_t* out, he_stream s) {
    return dim0_columns(ctx, dim0_query_eval, d0, database, false, present_device, columns, out, s, true);
}
int heamd::pir_remaining_dimensions_chunks_eval(const he_bfv_context* ctx, const uint32_t* dimensions, uint32_t dimension_count,
                                                size_t chunk_count, uint64_t* intermediate_eval, const uint64_t* remaining_query,
                                                size_t remaining_query_count, const uint64_t* relinearization_key, uint64_t* out,
                                                he_stream s) {
    ChunkShape shape;
    HEAMD_TRY_STATUS(chunk_shape(ctx, dimensions, dimension_count, remaining_query, remaining_query_count, shape));
    if (chunk_count == 0) return HE_OK;
    if (intermediate_eval == nullptr || out == nullptr) return invalid_argument("null operand");
    return remaining_dimensions(ctx, dimensions, dimension_count, shape, chunk_count, intermediate_eval, remaining_query,
                                relinearization_key, out, s, true);
}

extern "C" int he_pir_compute_response_chunk_device(const he_bfv_context* ctx, const uint32_t* dimensions,
                                                    uint32_t dimension_count, const uint64_t* dim0_query_eval,
                                                    const uint64_t* remaining_query, size_t remaining_query_count,
                                                    const uint64_t* database,
                                                    const uint8_t* present, const uint64_t* relinearization_key,
                                                    uint64_t* out, he_stream s) {
    ChunkShape shape;
    HEAMD_TRY_STATUS(chunk_shape(ctx, dimensions, dimension_count, remaining_query, remaining_query_count, shape));
    if (dim0_query_eval == nullptr || database == nullptr || out == nullptr) return invalid_argument("null operand");
    hipStream_t stream = as_stream(s);
    Scratch mask_mem(stream);
    const uint8_t* present_device = nullptr;
    if (present != nullptr) {
        HEAMD_HIP_TRY(mask_mem.allocate(shape.per_chunk));
        HEAMD_HIP_TRY(hipMemcpyAsync(mask_mem.get(), present, shape.per_chunk, hipMemcpyHostToDevice, stream));
        HEAMD_HIP_TRY(hipStreamSynchronize(stream));  // `present` is a borrowed pageable host buffer
        present_device = static_cast<const uint8_t*>(mask_mem.get());
    }
    return response_chunks_resident(ctx, dimensions, dimension_count, shape, 1, dim0_query_eval, remaining_query, database,
                                    shape.per_chunk * size_t(shape.L) * shape.n, false, present_device, relinearization_key,
                                    out, s);
}

namespace {
int compute_response(const he_bfv_context* ctx, const uint32_t* dimensions, uint32_t dimension_count,
                     const uint64_t* dim0_query_eval, const uint64_t* remaining_query, size_t remaining_query_count,
                     const uint64_t* database, bool packed, const uint8_t* present_device, size_t chunk_count,
                     const uint64_t* relinearization_key, uint64_t* out, he_stream s) {
    ChunkShape shape;
    HEAMD_TRY_STATUS(chunk_shape(ctx, dimensions, dimension_count, remaining_query, remaining_query_count, shape));
    if (chunk_count == 0) return HE_OK;
    if (dim0_query_eval == nullptr || database == nullptr || out == nullptr) return invalid_argument("null operand");
    size_t plaintext_words = size_t(shape.L) * shape.n;
    if (packed) {
        plaintext_words = he_bfv_packed_plaintext_words(ctx, shape.L);
        if (plaintext_words == 0) return HE_ERR_UNSUPPORTED;
    }
    const size_t chunk_words = shape.per_chunk * plaintext_words, out_words = 2 * shape.n;
    // The chunks are independent (the reference maps them over tasks, PirUtil.swift:533-563) and share the query: they are
    // answered together, `group` chunks per pass -- one dim-0 launch over all their columns, one batch per stage of
    // every remaining dimension -- with the group sized to keep the intermediate ciphertexts under ~1 GiB.
    const size_t intermediate_bytes = shape.columns * 2 * size_t(shape.L) * shape.n * sizeof(uint64_t);
    size_t group = (size_t(1) << 30) / (intermediate_bytes ? intermediate_bytes : 1);
    group = group == 0 ? 1 : group;
    for (size_t first = 0; first < chunk_count; first += group) {
        const size_t now = chunk_count - first < group ? chunk_count - first : group;
        HEAMD_TRY_STATUS(response_chunks_resident(
            ctx, dimensions, dimension_count, shape, now, dim0_query_eval, remaining_query, database + first * chunk_words,
            chunk_words, packed, present_device ? present_device + first * shape.per_chunk : nullptr, relinearization_key,
            out + first * out_words, s));
    }
    return HE_OK;
}
}  // namespace

extern "C" int he_pir_compute_response_device(const he_bfv_context* ctx, const uint32_t* dimensions,
                                              uint32_t dimension_count, const uint64_t* dim0_query_eval,
                                              const uint64_t* remaining_query, size_t remaining_query_count,
                                              const uint64_t* database, const uint8_t* present_device,
                                              size_t chunk_count, const uint64_t* relinearization_key, uint64_t* out,
                                              he_stream s) {
    return compute_response(ctx, dimensions, dimension_count, dim0_query_eval, remaining_query, remaining_query_count,
                            database, false, present_device, chunk_count, relinearization_key, out, s);
}
extern "C" int he_pir_compute_response_packed_device(const he_bfv_context* ctx, const uint32_t* dimensions,
                                                     uint32_t dimension_count, const uint64_t* dim0_query_eval,
                                                     const uint64_t* remaining_query, size_t remaining_query_count,
                                                     const uint64_t* packed_database, const uint8_t* present_device,
                                                     size_t chunk_count, const uint64_t* relinearization_key,
                                                     uint64_t* out, he_stream s) {
    return compute_response(ctx, dimensions, dimension_count, dim0_query_eval, remaining_query, remaining_query_count,
                            packed_database, true, present_device, chunk_count, relinearization_key, out, s);
}

// ---- the same chunk loop for Bfv<UInt32> on packed 4-byte slabs (the reference's 27/28-bit PIR parameter sets,
// EncryptionParameters.swift:313-345): half the database bytes of the 8-byte route --------------------------------
namespace {
// scratch of the 4-byte remaining-dimension stages for up to `group` chunks at a time
struct Word32Stages {
    Scratch next_mem, products_mem, level_mem;
    uint32_t *next = nullptr, *products = nullptr, *ping = nullptr, *pong = nullptr;
    explicit Word32Stages(hipStream_t stream) : next_mem(stream), products_mem(stream), level_mem(stream) {}
    hipError_t allocate(const ChunkShape& shape, size_t group) {
        const size_t ct2 = 2 * size_t(shape.L) * shape.n, ct3 = 3 * size_t(shape.L) * shape.n, widest = group * shape.columns;
        if (hipError_t e = next_mem.allocate(widest * ct2 * sizeof(uint32_t)); e != hipSuccess) return e;
        if (hipError_t e = products_mem.allocate(widest * ct3 * sizeof(uint32_t)); e != hipSuccess) return e;
        if (hipError_t e = level_mem.allocate(2 * group * ct2 * sizeof(uint32_t)); e != hipSuccess) return e;
        next = static_cast<uint32_t*>(next_mem.get());
        products = static_cast<uint32_t*>(products_mem.get());
        ping = static_cast<uint32_t*>(level_mem.get());
        pong = ping + group * ct2;
        return hipSuccess;
    }
};

// PirUtil.swift:448-485 for `chunks` chunks on 4-byte slabs: results [chunks][columns][2][L][N] Coeff (consumed) ->
// target [chunks][2][1][N]; every stage one batch over the result groups of all chunks
int remaining_dimensions_u32(const he_bfv_context* ctx, const uint32_t* dimensions, uint32_t dimension_count,
                             const ChunkShape& shape, size_t chunks, uint32_t* results, Word32Stages& stages,
                             const uint32_t* remaining_query, const uint32_t* relinearization_key, uint32_t* target,
                             he_stream s) {
    hipStream_t stream = as_stream(s);
    const uint32_t L = shape.L;
    const size_t ct2 = 2 * size_t(L) * shape.n, out_words = 2 * shape.n;
    uint32_t* current = results;
    uint32_t* other = stages.next;
    size_t count = shape.columns, cursor = 0;
    for (uint32_t i = 1; i < dimension_count; ++i) {
        const size_t d = dimensions[i];
        if (count % d != 0) return invalid_argument("intermediate results do not divide by the dimension");
        const size_t items = chunks * (count / d);
        HEAMD_TRY_STATUS(he_bfv_inner_product_shared_device_u32(ctx, L, remaining_query + cursor * ct2, current, d, items,
                                                                stages.products, s));
        HEAMD_TRY_STATUS(he_bfv_relinearize_device_u32(ctx, L, stages.products, relinearization_key, other, items, nullptr, 0, s));
        uint32_t* swap = current;
        current = other;
        other = swap;
        count /= d;
        cursor += d;
    }
    if (count != 1) return invalid_argument("dimensions leave more than one ciphertext");  // PirUtil.swift:481-482
    // modSwitchDownToSingle (:483)
    if (L == 1) {
        HEAMD_HIP_TRY(hipMemcpyAsync(target, current, chunks * out_words * sizeof(uint32_t), hipMemcpyDeviceToDevice, stream));
        return HE_OK;
    }
    const uint32_t* source = current;
    for (uint32_t level = L; level > 1; --level) {
        uint32_t* step = level == 2 ? target : (source == stages.ping ? stages.pong : stages.ping);
        HEAMD_TRY_STATUS(he_bfv_mod_switch_down_device_u32(ctx, level, 2, source, step, chunks, s));
        source = step;
    }
    return HE_OK;
}
}  // namespace

extern "C" int he_pir_compute_response_device_u32(const he_bfv_context* ctx, const uint32_t* dimensions,
                                                  uint32_t dimension_count, const uint32_t* dim0_query_eval,
                                                  const uint32_t* remaining_query, size_t remaining_query_count,
                                                  const uint32_t* database, const uint8_t* present_device,
                                                  size_t chunk_count, const uint32_t* relinearization_key, uint32_t* out,
                                                  he_stream s) {
    ChunkShape shape;
    HEAMD_TRY_STATUS(chunk_shape(ctx, dimensions, dimension_count, reinterpret_cast<const uint64_t*>(remaining_query),
                                 remaining_query_count, shape));
    if (chunk_count == 0) return HE_OK;
    if (dim0_query_eval == nullptr || database == nullptr || out == nullptr) return invalid_argument("null operand");
    hipStream_t stream = as_stream(s);
    const uint32_t L = shape.L;
    const size_t poly = size_t(L) * shape.n, ct2 = 2 * poly;
    const size_t chunk_words = shape.per_chunk * poly, out_words = 2 * shape.n;
    size_t group = (size_t(1) << 30) / (shape.columns * ct2 * sizeof(uint32_t));
    group = group == 0 ? 1 : (group < chunk_count ? group : chunk_count);
    Scratch results_mem(stream);
    Word32Stages stages(stream);
    HEAMD_HIP_TRY(results_mem.allocate(group * shape.columns * ct2 * sizeof(uint32_t)));
    HEAMD_HIP_TRY(stages.allocate(shape, group));
    uint32_t* results = static_cast<uint32_t*>(results_mem.get());
    for (size_t first = 0; first < chunk_count; first += group) {
        const size_t chunks = chunk_count - first < group ? chunk_count - first : group;
        // PirUtil.swift:428-438: every column of every chunk of the group in one launch, then back to Coeff
        HEAMD_TRY_STATUS(he_bfv_inner_product_plain_resident_device_u32(
            ctx, L, 2, dim0_query_eval, database + first * chunk_words,
            present_device ? present_device + first * shape.per_chunk : nullptr, shape.d0, chunks * shape.columns, results, s));
        HEAMD_TRY_STATUS(he_ntt_inverse_device_u32(shape.q_ctx, results, chunks * shape.columns * 2, s));
        HEAMD_TRY_STATUS(remaining_dimensions_u32(ctx, dimensions, dimension_count, shape, chunks, results, stages,
                                                  remaining_query, relinearization_key, out + first * out_words, s));
    }
    return HE_OK;
}

// Several queries (1..4) over one packed 4-byte database in one call: as he_pir_compute_response_queries_device, the
// dim-0 inner products of all of them share one pass over the database.
namespace {
int compute_response_queries_u32(const he_bfv_context* ctx, const uint32_t* dimensions, uint32_t dimension_count,
                                 size_t queries, const uint32_t* dim0_queries_eval, const uint32_t* remaining_queries,
                                 size_t remaining_query_count, size_t remaining_stride, const uint32_t* database,
                                 const uint8_t* present_device, size_t chunk_count,
                                 const uint32_t* const* relinearization_keys, uint32_t* out, he_stream s) {
    ChunkShape shape;
    HEAMD_TRY_STATUS(chunk_shape(ctx, dimensions, dimension_count, reinterpret_cast<const uint64_t*>(remaining_queries),
                                 remaining_query_count, shape));
    if (queries == 0 || chunk_count == 0) return HE_OK;
    if (queries > 4) return invalid_argument("at most 4 queries share one pass over the database");
    if (dim0_queries_eval == nullptr || database == nullptr || out == nullptr) return invalid_argument("null operand");
    if (dimension_count > 1 && relinearization_keys == nullptr) return invalid_argument("null relinearization keys");
    hipStream_t stream = as_stream(s);
    const uint32_t L = shape.L;
    const size_t ct_words = 2 * size_t(L) * shape.n, ct_bytes = ct_words * sizeof(uint32_t);
    const size_t chunk_words = shape.per_chunk * size_t(L) * shape.n, out_words = 2 * shape.n;
    size_t group = (size_t(1) << 30) / (shape.columns * queries * ct_bytes);
    group = group == 0 ? 1 : (group < chunk_count ? group : chunk_count);
    Scratch all_mem(stream), one_mem(stream);
    Word32Stages stages(stream);
    HEAMD_HIP_TRY(all_mem.allocate(group * shape.columns * queries * ct_bytes));
    HEAMD_HIP_TRY(one_mem.allocate(group * shape.columns * ct_bytes));
    HEAMD_HIP_TRY(stages.allocate(shape, group));
    uint32_t* all = static_cast<uint32_t*>(all_mem.get());  // [chunk][column][query][2][L][N]
    uint32_t* one = static_cast<uint32_t*>(one_mem.get());  // [chunk][column][2][L][N] of one query
    for (size_t first = 0; first < chunk_count; first += group) {
        const size_t now = chunk_count - first < group ? chunk_count - first : group;
        const size_t columns = now * shape.columns;
        HEAMD_TRY_STATUS(he_bfv_inner_product_plain_resident_device_u32(
            ctx, L, static_cast<uint32_t>(2 * queries), dim0_queries_eval, database + first * chunk_words,
            present_device ? present_device + first * shape.per_chunk : nullptr, shape.d0, columns, all, s));
        HEAMD_TRY_STATUS(he_ntt_inverse_device_u32(shape.q_ctx, all, columns * 2 * queries, s));
        for (size_t q = 0; q < queries; ++q) {
            HEAMD_HIP_TRY(hipMemcpy2DAsync(one, ct_bytes, all + q * ct_words, queries * ct_bytes, ct_bytes, columns,
                                           hipMemcpyDeviceToDevice, stream));
            HEAMD_TRY_STATUS(remaining_dimensions_u32(
                ctx, dimensions, dimension_count, shape, now, one, stages,
                remaining_queries ? remaining_queries + q * remaining_stride * ct_words : nullptr,
                relinearization_keys ? relinearization_keys[q] : nullptr, out + (q * chunk_count + first) * out_words, s));
        }
    }
    return HE_OK;
}
}  // namespace

extern "C" int he_pir_compute_response_queries_device_u32(const he_bfv_context* ctx, const uint32_t* dimensions,
                                                          uint32_t dimension_count, size_t queries,
                                                          const uint32_t* dim0_queries_eval,
                                                          const uint32_t* remaining_queries, size_t remaining_query_count,
                                                          const uint32_t* database, const uint8_t* present_device,
                                                          size_t chunk_count, const uint32_t* const* relinearization_keys,
                                                          uint32_t* out, he_stream s) {
    return compute_response_queries_u32(ctx, dimensions, dimension_count, queries, dim0_queries_eval, remaining_queries,
                                        remaining_query_count, remaining_query_count, database, present_device, chunk_count,
                                        relinearization_keys, out, s);
}

// Several queries over the same database in one call: the dim-0 inner products of all of them stream the database once
// (their ciphertext vectors side by side, he_amd.h he_bfv_inner_product_plain_device with polys = 2 x queries); the
// remaining dimensions, which involve only query ciphertexts and intermediate results, then run query by query.
namespace {
// remaining_stride: ciphertexts from one query's remaining ciphertexts to the next query's
int compute_response_queries(const he_bfv_context* ctx, const uint32_t* dimensions, uint32_t dimension_count, size_t queries,
                             const uint64_t* dim0_queries_eval, const uint64_t* remaining_queries,
                             size_t remaining_query_count, size_t remaining_stride, const uint64_t* database,
                             const uint8_t* present_device, size_t chunk_count,
                             const uint64_t* const* relinearization_keys, uint64_t* out, he_stream s) {
    ChunkShape shape;
    HEAMD_TRY_STATUS(chunk_shape(ctx, dimensions, dimension_count, remaining_queries, remaining_query_count, shape));
    if (queries == 0 || chunk_count == 0) return HE_OK;
    if (queries > 4) return invalid_argument("at most 4 queries share one pass over the database");
    if (dim0_queries_eval == nullptr || database == nullptr || out == nullptr) return invalid_argument("null operand");
    if (dimension_count > 1 && relinearization_keys == nullptr) return invalid_argument("null relinearization keys");
    hipStream_t stream = as_stream(s);
    const uint32_t L = shape.L;
    const size_t ct_words = 2 * size_t(L) * shape.n, ct_bytes = ct_words * sizeof(uint64_t);
    const size_t chunk_words = shape.per_chunk * size_t(L) * shape.n, out_words = 2 * shape.n;
    // groups of chunks that keep the intermediate ciphertexts of all queries under ~1 GiB
    const size_t intermediate_bytes = shape.columns * queries * ct_bytes;
    size_t group = (size_t(1) << 30) / (intermediate_bytes ? intermediate_bytes : 1);
    group = group == 0 ? 1 : (group < chunk_count ? group : chunk_count);
    Scratch all_mem(stream), one_mem(stream);
    HEAMD_HIP_TRY(all_mem.allocate(group * shape.columns * queries * ct_bytes));
    HEAMD_HIP_TRY(one_mem.allocate(group * shape.columns * ct_bytes));
    uint64_t* all = static_cast<uint64_t*>(all_mem.get());  // [chunk][column][query][2][L][N]
    uint64_t* one = static_cast<uint64_t*>(one_mem.get());  // [chunk][column][2][L][N] of one query
    // As response_chunks_resident (one piece unless forced): a piece's dim-0 pass for every query (PirUtil.swift:428-438) on
    // the caller's stream, the queries' remaining dimensions on a lane beside the next piece's pass.  (`one` is only touched
    // on the stream the remaining dimensions run on: its uses are ordered there.)
    const size_t piece = overlap_piece_chunks(dimension_count, group);
    std::unique_ptr<heamd::LaneLease> lease;
    if (piece < group) lease.reset(new heamd::LaneLease(heamd::lane_pool(ctx), stream));
    SideLane* lane = lease ? lease->lane : nullptr;
    hipStream_t tail_stream = lane != nullptr ? lane->stream : stream;
    int status = HE_OK;
    hipError_t e = hipSuccess;
    bool forked = false;
    for (size_t first = 0, now = 0; first < chunk_count && status == HE_OK && e == hipSuccess; first += now) {
        now = chunk_count - first < piece ? chunk_count - first : piece;
        now = group - first % group < now ? group - first % group : now;  // (a piece does not straddle the slabs' end)
        const size_t columns = now * shape.columns;
        uint64_t* piece_all = all + (first % group) * shape.columns * queries * ct_words;
        if (lane != nullptr && first != 0 && first % group == 0) {
            // the slabs wrap around: the lane's work on the previous `group` chunks first
            e = hipEventRecord(lane->joined, lane->stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(stream, lane->joined, 0);
            if (e != hipSuccess) break;
        }
        status = he_bfv_inner_product_plain_resident_device(
            ctx, L, static_cast<uint32_t>(2 * queries), dim0_queries_eval, database + first * chunk_words,
            present_device ? present_device + first * shape.per_chunk : nullptr, shape.d0, columns, piece_all, s);
        // (left in Eval form: remaining_dimensions, results_in_eval)
        if (status != HE_OK) break;
        if (lane != nullptr) {
            e = hipEventRecord(lane->stage[0], stream);
            if (e == hipSuccess) e = hipStreamWaitEvent(lane->stream, lane->stage[0], 0);
            if (e != hipSuccess) break;
            forked = true;
        }
        for (size_t q = 0; q < queries && status == HE_OK && e == hipSuccess; ++q) {
            // this query's results, column after column (a strided copy: they sit `queries` ciphertexts apart)
            e = heamd::launch_copy_records(piece_all + q * ct_words, queries * ct_words, one, ct_words, ct_words, columns, tail_stream);
            if (e == hipErrorInvalidValue) {  // (a degree below the copy kernel's granule)
                (void)hipGetLastError();
                e = hipMemcpy2DAsync(one, ct_bytes, piece_all + q * ct_words, queries * ct_bytes, ct_bytes, columns,
                                     hipMemcpyDeviceToDevice, tail_stream);
            }
            if (e != hipSuccess) break;
            status = remaining_dimensions(
                ctx, dimensions, dimension_count, shape, now, one,
                remaining_queries ? remaining_queries + q * remaining_stride * ct_words : nullptr,
                relinearization_keys ? relinearization_keys[q] : nullptr, out + (q * chunk_count + first) * out_words,
                static_cast<he_stream>(tail_stream), true);
        }
    }
    if (forked) {  // the join is enqueued whatever happened in between
        const hipError_t recorded = hipEventRecord(lane->joined, lane->stream);
        const hipError_t waited = recorded == hipSuccess ? hipStreamWaitEvent(stream, lane->joined, 0) : recorded;
        if (e == hipSuccess) e = waited;
    }
    if (status != HE_OK) return status;
    HEAMD_HIP_TRY(e);
    return HE_OK;
}
}  // namespace

extern "C" int he_pir_compute_response_queries_device(const he_bfv_context* ctx, const uint32_t* dimensions,
                                                      uint32_t dimension_count, size_t queries,
                                                      const uint64_t* dim0_queries_eval, const uint64_t* remaining_queries,
                                                      size_t remaining_query_count, const uint64_t* database,
                                                      const uint8_t* present_device, size_t chunk_count,
                                                      const uint64_t* const* relinearization_keys, uint64_t* out,
                                                      he_stream s) {
    return compute_response_queries(ctx, dimensions, dimension_count, queries, dim0_queries_eval, remaining_queries,
                                    remaining_query_count, remaining_query_count, database, present_device, chunk_count,
                                    relinearization_keys, out, s);
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

// plans (or finds) the expansion of `ciphertext_count` ciphertexts of degree n into `output_count`
int expand_plan(const he_bfv_context* ctx, size_t n, size_t ciphertext_count, size_t output_count,
                std::shared_ptr<const heamd::ExpandPlan>& out) {
    heamd::ExpandPlanCache& cache = heamd::expand_plans(ctx);
    const heamd::ExpandPlanCache::Key key(ciphertext_count, output_count);
    out = cache.find(key);
    if (out) return HE_OK;
    if (output_count > 0x7fffffffull) return invalid_argument("too many outputs");
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
    // the data movement of every level in one table: per level its leaves (-> output slot, doubled when the leaf sits
    // above its tree's height) and, when leaves and internal nodes mix, the gather of the parents
    auto plan = std::make_shared<heamd::ExpandPlan>();
    plan->levels.resize(levels.size());
    std::vector<uint32_t> table;
    for (size_t depth = 0; depth < levels.size(); ++depth) {
        const std::vector<ExpandNode>& level = levels[depth];
        plan->widest = level.size() > plan->widest ? level.size() : plan->widest;
        const int log_step = static_cast<int>(depth) + 1;
        heamd::ExpandPlan::Level& m = plan->levels[depth];
        m.nodes = level.size();
        m.leaf_offset = table.size();
        for (size_t i = 0; i < level.size(); ++i) {
            if (level[i].output_count != 1) continue;
            const uint32_t doubled = log_step > level[i].expected_height ? 0u : 1u;  // PirUtil.swift:262-268
            table.push_back(static_cast<uint32_t>(i));
            table.push_back((static_cast<uint32_t>(level[i].leaf_slot) << 1) | doubled);
            ++m.leaf_count;
        }
        m.parent_count = level.size() - m.leaf_count;
        m.gather = m.leaf_count != 0 && m.parent_count != 0;
        m.parent_offset = table.size();
        if (m.gather) {
            uint32_t k = 0;
            for (size_t i = 0; i < level.size(); ++i) {
                if (level[i].output_count == 1) continue;
                table.push_back(static_cast<uint32_t>(i));
                table.push_back(k++ << 1);
            }
        }
    }
    // the table stays with the plan: one blocking upload per shape
    HEAMD_HIP_TRY(hipMalloc(reinterpret_cast<void**>(&plan->table_device), (table.size() + 1) * sizeof(uint32_t)));
    if (!table.empty())
        HEAMD_HIP_TRY(hipMemcpy(plan->table_device, table.data(), table.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    out = cache.insert(key, std::move(plan));
    return HE_OK;
}

}  // namespace

extern "C" int he_pir_expand_batch_device(const he_bfv_context* ctx, const uint64_t* ciphertexts, size_t queries,
                                          size_t ciphertext_count, size_t output_count, const uint64_t* galois_elements,
                                          const uint64_t* const* galois_keys, size_t galois_key_count, uint64_t* out,
                                          he_stream s) {
    if (ctx == nullptr) return invalid_argument("null context");
    if (queries == 0) return HE_OK;

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

    // ---- plan: the recursion tree of every input ciphertext, level by level (once per shape and context)
    std::shared_ptr<const heamd::ExpandPlan> plan;
    HEAMD_TRY_STATUS(expand_plan(ctx, n, ciphertext_count, output_count, plan));
    const std::vector<heamd::ExpandPlan::Level>& moves = plan->levels;
    const uint32_t* table_device = plan->table_device;
    const size_t widest = plan->widest;
    const heamd::DeviceContext q_device = q_ctx->impl->device_context();
    Scratch cur_mem(stream), next_mem(stream), parent_mem(stream), rotated_mem(stream), tmp_mem(stream),
        workspace_mem(stream);
    // every level runs over all queries at once: buffers hold [query][node of the level].  The gathered parents, the key
    // switch's workspace and the fallback's two rotation buffers only ever hold the parents of a level (at most half of the
    // widest level: every parent has two children in the next one); the rotation buffers are allocated when the fallback
    // is first taken.
    size_t most_parents = 1;
    for (const heamd::ExpandPlan::Level& level : moves) most_parents = std::max(most_parents, level.parent_count);
    const size_t workspace_bytes = he_bfv_apply_galois_workspace_bytes(ctx, L, queries * most_parents);
    HEAMD_HIP_TRY(cur_mem.allocate(queries * widest * ct_bytes));
    HEAMD_HIP_TRY(next_mem.allocate(queries * widest * ct_bytes));
    HEAMD_HIP_TRY(parent_mem.allocate(queries * most_parents * ct_bytes));
    HEAMD_HIP_TRY(workspace_mem.allocate(workspace_bytes));
    const uint64_t* cur = ciphertexts;  // level 0 reads the caller's ciphertexts in place
    uint64_t* buffers[2] = {static_cast<uint64_t*>(cur_mem.get()), static_cast<uint64_t*>(next_mem.get())};
    uint64_t* gathered = static_cast<uint64_t*>(parent_mem.get());
    uint64_t *rotated = nullptr, *tmp = nullptr;

    // ---- execute: every stage of a level is one launch over all nodes of all trees of all queries at that depth
    // (the inner product with the Galois key alone runs per run of queries that share a key)
    int status = HE_OK;
    std::vector<const uint64_t*> level_keys(queries);
    for (size_t depth = 0; depth < moves.size() && status == HE_OK; ++depth) {
        const int log_step = static_cast<int>(depth) + 1;
        const heamd::ExpandPlan::Level& m = moves[depth];
        const size_t level_nodes = m.nodes;
        if (m.leaf_count != 0)
            HEAMD_HIP_TRY(heamd::launch_expand_move(cur, out, table_device + m.leaf_offset, q_device, m.leaf_count, queries,
                                                    level_nodes, output_count, stream));
        if (m.parent_count == 0) continue;
        if (log_step > log_degree) {  // precondition, PirUtil.swift:212
            status = invalid_argument("logStep exceeds log2(degree)");
            break;
        }
        const size_t batch = m.parent_count;
        const uint64_t* parents = cur;
        if (m.gather) {
            HEAMD_HIP_TRY(heamd::launch_expand_move(cur, gathered, table_device + m.parent_offset, q_device, batch, queries,
                                                    level_nodes, batch, stream));
            parents = gathered;
        }
        // expandCiphertextForOneStep (PirUtil.swift:204-236)
        const uint64_t target = (uint64_t(1) << (log_degree - log_step + 1)) + 1;
        long best = -1;
        for (size_t k = 0; k < galois_key_count; ++k)
            if (galois_elements[k] <= target && (best < 0 || galois_elements[k] > galois_elements[best]))
                best = static_cast<long>(k);
        bool have_keys = best >= 0;
        for (size_t q = 0; have_keys && q < queries; ++q) {
            level_keys[q] = galois_keys[q * galois_key_count + static_cast<size_t>(best)];
            have_keys = level_keys[q] != nullptr;
        }
        if (!have_keys) {
            heamd::set_last_error("no Galois element <= " + std::to_string(target) + " in the evaluation key");
            status = HE_ERR_MISSING_GALOIS_KEY;
            break;
        }
        const uint64_t element = galois_elements[best];
        // the reference applies the element 2^(log2(target - 1) - log2(element - 1)) times and then traps unless the
        // composition IS the target element (precondition(currElement == targetElement), PirUtil.swift:222-231): with an
        // evaluation key outside the 2^k + 1 ladder this call fails instead of returning another expansion
        if (element < 3 || (element & 1) == 0) {
            heamd::set_last_error("Galois element " + std::to_string(element) + " cannot reach " + std::to_string(target));
            status = HE_ERR_MISSING_GALOIS_KEY;
            break;
        }
        const int applications = 1 << (floor_log2_size(target - 1) - floor_log2_size(element - 1));
        uint64_t composed = 1;
        for (int a = 0; a < applications; ++a) composed = (composed * element) % (2 * n);
        if (composed != target) {
            heamd::set_last_error("Galois element " + std::to_string(element) + " applied " + std::to_string(applications) +
                                  " times is not " + std::to_string(target) + " (mod 2N)");
            status = HE_ERR_MISSING_GALOIS_KEY;
            break;
        }
        // children: c1 + ciphertext and (ciphertext - c1) x^(-2^(logStep-1)), interleaved as the plan numbered them
        const uint32_t shift = static_cast<uint32_t>(2 * n - (size_t(1) << (log_step - 1)));
        uint64_t* next = buffers[depth % 2];
        // The children leave the LAST application's key switch directly (the element has its own key: the only one) -- and
        // when all of them are leaves, for their output slots (the next level's leaf table is in node order).  With
        // repeated application (keyCompression configurations, PirUtil.swift:221-231) the applications before the last
        // one are plain Galois key switches of the parents; the last one rotates their result and forms the children
        // with the parents themselves.
        const bool to_outputs = depth + 1 < moves.size() && moves[depth + 1].parent_count == 0 &&
                                moves[depth + 1].leaf_count == 2 * batch;
        if (applications > 1 && rotated == nullptr) {
            HEAMD_HIP_TRY(rotated_mem.allocate(queries * most_parents * ct_bytes));
            HEAMD_HIP_TRY(tmp_mem.allocate(queries * most_parents * ct_bytes));
            rotated = static_cast<uint64_t*>(rotated_mem.get());
            tmp = static_cast<uint64_t*>(tmp_mem.get());
        }
        const uint64_t* c1 = parents;
        for (int a = 0; a + 1 < applications && status == HE_OK; ++a) {  // applyGalois(element), all but the last time
            uint64_t* dst = (a % 2 == 0) ? rotated : tmp;
            status = he_bfv_apply_galois_grouped_device(ctx, L, c1, element, level_keys.data(), queries, batch, dst,
                                                        workspace_mem.get(), workspace_bytes, s);
            c1 = dst;
        }
        if (status != HE_OK) break;
        status = heamd::bfv_expand_step_fused(
            ctx, L, parents, element, level_keys.data(), queries, batch, to_outputs ? out : next, shift,
            to_outputs ? table_device + moves[depth + 1].leaf_offset : nullptr, output_count, workspace_mem.get(),
            workspace_bytes, stream, c1 == parents ? nullptr : c1);
        if (status == HE_OK) {
            if (to_outputs) break;  // nothing below this level
            cur = next;
            continue;
        }
        if (status != heamd::kExpandStepUnavailable) break;
        // no fused key switch for this degree: the last application on its own, then the step kernel
        if (rotated == nullptr) {
            HEAMD_HIP_TRY(rotated_mem.allocate(queries * most_parents * ct_bytes));
            HEAMD_HIP_TRY(tmp_mem.allocate(queries * most_parents * ct_bytes));
            rotated = static_cast<uint64_t*>(rotated_mem.get());
            tmp = static_cast<uint64_t*>(tmp_mem.get());
        }
        uint64_t* last = (c1 == rotated) ? tmp : rotated;
        status = he_bfv_apply_galois_grouped_device(ctx, L, c1, element, level_keys.data(), queries, batch, last,
                                                    workspace_mem.get(), workspace_bytes, s);
        if (status != HE_OK) break;
        HEAMD_HIP_TRY(heamd::launch_expand_step(parents, last, next, q_device, shift, queries * batch, stream));
        cur = next;
    }
    if (status != HE_OK) return status;
    return HE_OK;
}

extern "C" int he_pir_expand_device(const he_bfv_context* ctx, const uint64_t* ciphertexts, size_t ciphertext_count,
                                    size_t output_count, const uint64_t* galois_elements,
                                    const uint64_t* const* galois_keys, size_t galois_key_count, uint64_t* out,
                                    he_stream s) {
    return he_pir_expand_batch_device(ctx, ciphertexts, 1, ciphertext_count, output_count, galois_elements, galois_keys,
                                      galois_key_count, out, s);
}

// ---------------------------------------------------------------------------------------------------------------------
// PirUtilProtocol.computeResponse(to:using:databases:parameter:context:callOptions:) with one database
// (PirUtil.swift:490-568): expand the query's ciphertexts into expandedQueryCount = sum(dimensions) selection
// ciphertexts per queried index (:500-505, :565-566), take each index's first dimensions[0] of them to Eval
// (:520-533), and answer every chunk (:545-563).  The indices of one Query share the database: up to four of them at
// a time share one pass over it.
extern "C" int he_pir_compute_response_to_query_device(
    const he_bfv_context* ctx, const uint32_t* dimensions, uint32_t dimension_count, const uint64_t* query_ciphertexts,
    size_t query_ciphertext_count, size_t indices_count, const uint64_t* galois_elements,
    const uint64_t* const* galois_keys, size_t galois_key_count, const uint64_t* relinearization_key,
    const uint64_t* const* databases, const uint8_t* const* present_masks, size_t database_count, size_t chunk_count,
    uint64_t* out, he_stream s) {
    if (ctx == nullptr) return invalid_argument("null context");
    if (dimensions == nullptr || dimension_count == 0) return invalid_argument("empty dimensions");
    size_t expanded_count = 0;
    for (uint32_t i = 0; i < dimension_count; ++i) expanded_count += dimensions[i];
    const size_t remaining_count = expanded_count - dimensions[0];
    ChunkShape shape;
    // the remaining query is checked against the shape below; a placeholder stands in for it here
    HEAMD_TRY_STATUS(chunk_shape(ctx, dimensions, dimension_count, remaining_count ? query_ciphertexts : nullptr,
                                 remaining_count, shape));
    // PirError.invalidBatchSize (PirUtil.swift:498-500): one database for every index, or one per index
    if (databases == nullptr || !(database_count == 1 || database_count >= indices_count))
        return invalid_argument("database count matches neither one nor the number of indices");
    if (indices_count == 0 || chunk_count == 0) return HE_OK;
    if (query_ciphertexts == nullptr || out == nullptr) return invalid_argument("null operand");
    for (size_t d = 0; d < (database_count == 1 ? size_t(1) : indices_count); ++d)
        if (databases[d] == nullptr) return invalid_argument("null database");
    const bool shared = database_count == 1;
    hipStream_t stream = as_stream(s);
    const size_t ct_words = 2 * size_t(shape.L) * shape.n, ct_bytes = ct_words * sizeof(uint64_t);
    const size_t total = expanded_count * indices_count, out_words = 2 * shape.n;
    Scratch expanded_mem(stream), side_mem(stream);
    HEAMD_HIP_TRY(expanded_mem.allocate(total * ct_bytes));
    uint64_t* expanded = static_cast<uint64_t*>(expanded_mem.get());  // [index][sum(dimensions)][2][L][N] Coeff
    HEAMD_TRY_STATUS(he_pir_expand_device(ctx, query_ciphertexts, query_ciphertext_count, total, galois_elements,
                                          galois_keys, galois_key_count, expanded, s));
    // indices with a database of their own (PirUtil.swift:518) are answered one by one; those that share one go four at
    // a time
    constexpr size_t kGroup = 4;
    const size_t kTogether = shared ? kGroup : 1;
    const size_t widest = indices_count < kTogether ? indices_count : kTogether;
    uint64_t* side = nullptr;  // [dimensions[0]][indices of a group][2][L][N]
    if (widest > 1) {
        HEAMD_HIP_TRY(side_mem.allocate(shape.d0 * widest * ct_bytes));
        side = static_cast<uint64_t*>(side_mem.get());
    }
    const uint64_t* keys[kGroup] = {relinearization_key, relinearization_key, relinearization_key, relinearization_key};
    for (size_t first = 0; first < indices_count; first += kTogether) {
        const size_t now = indices_count - first < kTogether ? indices_count - first : kTogether;
        uint64_t* mine = expanded + first * expanded_count * ct_words;  // this group's selection ciphertexts
        const uint64_t* rest = remaining_count ? mine + shape.d0 * ct_words : nullptr;
        uint64_t* group_out = out + first * chunk_count * out_words;
        const uint64_t* database = databases[shared ? 0 : first];
        const uint8_t* present_device = present_masks ? present_masks[shared ? 0 : first] : nullptr;
        if (now == 1) {
            // convertToEvalFormat (PirUtil.swift:520-533) where the expansion left the dim-0 ciphertexts
            HEAMD_TRY_STATUS(he_ntt_forward_device(shape.q_ctx, mine, shape.d0 * 2, s));
            HEAMD_TRY_STATUS(he_pir_compute_response_device(ctx, dimensions, dimension_count, mine, rest, remaining_count,
                                                            database, present_device, chunk_count, relinearization_key,
                                                            group_out, s));
            continue;
        }
        // the group's dim-0 ciphertexts side by side, then to Eval
        for (size_t q = 0; q < now; ++q)
            HEAMD_HIP_TRY(hipMemcpy2DAsync(side + q * ct_words, now * ct_bytes, mine + q * expanded_count * ct_words,
                                           ct_bytes, ct_bytes, shape.d0, hipMemcpyDeviceToDevice, stream));
        HEAMD_TRY_STATUS(he_ntt_forward_device(shape.q_ctx, side, shape.d0 * now * 2, s));
        HEAMD_TRY_STATUS(compute_response_queries(ctx, dimensions, dimension_count, now, side, rest, remaining_count,
                                                  expanded_count, database, present_device, chunk_count,
                                                  dimension_count > 1 ? keys : nullptr, group_out, s));
    }
    return HE_OK;
}

// The whole-query call for Bfv<UInt32> on packed 4-byte slabs: the expansion (bound by its key switches, not by bytes)
// runs on widened words with the 8-byte kernels -- the Galois keys are therefore taken as 8-byte slabs, as
// he_pir_expand_device takes them for a UInt32 context -- everything that touches the database is 4-byte; indices that
// share the database share its pass four at a time.
extern "C" int he_pir_compute_response_to_query_device_u32(
    const he_bfv_context* ctx, const uint32_t* dimensions, uint32_t dimension_count, const uint32_t* query_ciphertexts,
    size_t query_ciphertext_count, size_t indices_count, const uint64_t* galois_elements,
    const uint64_t* const* galois_keys_wide, size_t galois_key_count, const uint32_t* relinearization_key,
    const uint32_t* const* databases, const uint8_t* const* present_masks, size_t database_count, size_t chunk_count,
    uint32_t* out, he_stream s) {
    if (ctx == nullptr) return invalid_argument("null context");
    if (dimensions == nullptr || dimension_count == 0) return invalid_argument("empty dimensions");
    size_t expanded_count = 0;
    for (uint32_t i = 0; i < dimension_count; ++i) expanded_count += dimensions[i];
    const size_t remaining_count = expanded_count - dimensions[0];
    ChunkShape shape;
    HEAMD_TRY_STATUS(chunk_shape(ctx, dimensions, dimension_count,
                                 remaining_count ? reinterpret_cast<const uint64_t*>(query_ciphertexts) : nullptr,
                                 remaining_count, shape));
    if (databases == nullptr || !(database_count == 1 || database_count >= indices_count))
        return invalid_argument("database count matches neither one nor the number of indices");
    if (indices_count == 0 || chunk_count == 0) return HE_OK;
    if (query_ciphertexts == nullptr || out == nullptr) return invalid_argument("null operand");
    hipStream_t stream = as_stream(s);
    const size_t ct_words = 2 * size_t(shape.L) * shape.n;
    const size_t total = expanded_count * indices_count, out_words = 2 * shape.n;
    Scratch query_mem(stream), wide_mem(stream), expanded_mem(stream);
    HEAMD_HIP_TRY(query_mem.allocate(query_ciphertext_count * ct_words * sizeof(uint64_t)));
    HEAMD_HIP_TRY(wide_mem.allocate(total * ct_words * sizeof(uint64_t)));
    HEAMD_HIP_TRY(expanded_mem.allocate(total * ct_words * sizeof(uint32_t)));
    uint64_t* query_wide = static_cast<uint64_t*>(query_mem.get());
    uint64_t* expanded_wide = static_cast<uint64_t*>(wide_mem.get());
    uint32_t* expanded = static_cast<uint32_t*>(expanded_mem.get());  // [index][sum(dimensions)][2][L][N] Coeff
    HEAMD_TRY_STATUS(he_words_widen_u32_device(query_ciphertexts, query_wide, query_ciphertext_count * ct_words, s));
    HEAMD_TRY_STATUS(he_pir_expand_device(ctx, query_wide, query_ciphertext_count, total, galois_elements, galois_keys_wide,
                                          galois_key_count, expanded_wide, s));
    HEAMD_TRY_STATUS(he_words_narrow_u64_device(expanded_wide, expanded, total * ct_words, s));
    // indices with a database of their own are answered one by one; those that share one go four at a time
    const bool shared = database_count == 1;
    constexpr size_t kGroup = 4;
    const size_t together = shared ? kGroup : 1;
    const size_t widest = indices_count < together ? indices_count : together;
    for (size_t d = 0; d < (shared ? size_t(1) : indices_count); ++d)
        if (databases[d] == nullptr) return invalid_argument("null database");
    Scratch side_mem(stream);
    uint32_t* side = nullptr;  // [dimensions[0]][indices of a group][2][L][N]
    if (widest > 1) {
        HEAMD_HIP_TRY(side_mem.allocate(shape.d0 * widest * ct_words * sizeof(uint32_t)));
        side = static_cast<uint32_t*>(side_mem.get());
    }
    const uint32_t* keys[kGroup] = {relinearization_key, relinearization_key, relinearization_key, relinearization_key};
    const size_t ct_bytes = ct_words * sizeof(uint32_t);
    for (size_t first = 0; first < indices_count; first += together) {
        const size_t now = indices_count - first < together ? indices_count - first : together;
        uint32_t* mine = expanded + first * expanded_count * ct_words;
        const uint32_t* rest = remaining_count ? mine + shape.d0 * ct_words : nullptr;
        uint32_t* group_out = out + first * chunk_count * out_words;
        const uint32_t* database = databases[shared ? 0 : first];
        const uint8_t* present_device = present_masks ? present_masks[shared ? 0 : first] : nullptr;
        if (now == 1) {
            HEAMD_TRY_STATUS(he_ntt_forward_device_u32(shape.q_ctx, mine, shape.d0 * 2, s));
            HEAMD_TRY_STATUS(he_pir_compute_response_device_u32(ctx, dimensions, dimension_count, mine, rest, remaining_count,
                                                                database, present_device, chunk_count, relinearization_key,
                                                                group_out, s));
            continue;
        }
        for (size_t q = 0; q < now; ++q)
            HEAMD_HIP_TRY(hipMemcpy2DAsync(side + q * ct_words, now * ct_bytes, mine + q * expanded_count * ct_words, ct_bytes,
                                           ct_bytes, shape.d0, hipMemcpyDeviceToDevice, stream));
        HEAMD_TRY_STATUS(he_ntt_forward_device_u32(shape.q_ctx, side, shape.d0 * now * 2, s));
        HEAMD_TRY_STATUS(compute_response_queries_u32(ctx, dimensions, dimension_count, now, side, rest, remaining_count,
                                                      expanded_count, database, present_device, chunk_count,
                                                      dimension_count > 1 ? keys : nullptr, group_out, s));
    }
    return HE_OK;
}
