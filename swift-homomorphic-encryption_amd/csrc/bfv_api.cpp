// bfv_api.cpp -- B3 entry points: Context<Bfv<UInt64>> and the HeScheme operations on the hot path
// (reference Sources/HomomorphicEncryption/Bfv/*.swift).  Each operation is a short pipeline of HIP kernels on the
// caller's stream; nothing here touches the data on the host.
#include <memory>
#include <mutex>
#include <type_traits>
#include <vector>

#include "api_internal.hpp"
#include "bfv_context.hpp"
#include "kernels.hpp"
#include "rns_kernels.hpp"
#include "side_lane.hpp"

using heamd::as_stream;
using heamd::BfvContext;
using heamd::DeviceContext;
using heamd::invalid_argument;
using heamd::PolyContext;
using heamd::RnsToolLevel;
using heamd::Scratch;
using heamd::SideLane;

struct he_bfv_context {
    std::unique_ptr<BfvContext> impl;
    // non-owning he_poly_context views handed out by he_bfv_*_context(), index = ciphertext moduli count
    std::vector<std::unique_ptr<he_poly_context>> ciphertext, key_switching, qbsk;
    mutable heamd::ExpandPlanCache expand_plans;  // PirUtil.expand shapes seen so far (api_internal.hpp)
    mutable heamd::LanePool lanes;                 // side_lane.hpp: streams for work of one call that runs beside its neighbour
};

namespace heamd {
ExpandPlanCache& expand_plans(const he_bfv_context* ctx) { return ctx->expand_plans; }
LanePool& lane_pool(const he_bfv_context* ctx) { return ctx->lanes; }
}  // namespace heamd

namespace {

int bfv_create(uint32_t degree, uint64_t t, const uint64_t* q, uint32_t count, bool host_only, he_bfv_context** out,
               int word_bits = 64) {
    if (out == nullptr) return invalid_argument("null out");
    *out = nullptr;
    std::unique_ptr<BfvContext> impl;
    const int status = BfvContext::create(degree, t, q, count, impl, host_only, word_bits);
    if (status != HE_OK) {
        if (status != HE_ERR_DEVICE) heamd::set_last_error(std::string("Context.init: ") + he_status_string(status));
        return status;
    }
    auto* handle = new he_bfv_context{};
    const uint32_t L = impl->top_level();
    handle->ciphertext.resize(L + 1);
    handle->key_switching.resize(L + 1);
    handle->qbsk.resize(L + 1);
    for (uint32_t k = 1; k <= L; ++k) {
        handle->ciphertext[k].reset(new he_poly_context{const_cast<PolyContext*>(impl->ciphertext(k)), false});
        if (impl->key_switching(k) != nullptr)
            handle->key_switching[k].reset(new he_poly_context{const_cast<PolyContext*>(impl->key_switching(k)), false});
        handle->qbsk[k].reset(new he_poly_context{impl->tool(k)->qbsk.get(), false});
    }
    handle->impl = std::move(impl);
    *out = handle;
    return HE_OK;
}

// Common argument checks; returns the level's tool on success.
int check_level(const he_bfv_context* ctx, uint32_t moduli_count, const RnsToolLevel** tool) {
    if (ctx == nullptr) return invalid_argument("null context");
    if (!ctx->impl->valid(moduli_count)) return invalid_argument("moduli_count out of range");
    if (ctx->impl->host_only()) {
        heamd::set_last_error("context was created host-only (no device tables)");
        return HE_ERR_DEVICE;
    }
    const int status = ctx->impl->ciphertext(moduli_count)->check_device();
    if (status != HE_OK) return status;
    *tool = ctx->impl->tool(moduli_count);
    if ((*tool)->device.L > heamd::rns_max_supported_moduli()) {
        heamd::set_last_error("more than 16 ciphertext moduli are not supported by the BEHZ kernels");
        return HE_ERR_UNSUPPORTED;
    }
    return HE_OK;
}

size_t qbsk_poly_words(const BfvContext& ctx, uint32_t L) { return size_t(2 * L + 1) * ctx.degree(); }

// Resolves caller workspace vs stream-ordered scratch.
int resolve_workspace(void* workspace, size_t workspace_bytes, size_t needed, Scratch& scratch, uint64_t** out) {
    if (workspace != nullptr) {
        if (workspace_bytes < needed) return invalid_argument("workspace too small");
        *out = static_cast<uint64_t*>(workspace);
        return HE_OK;
    }
    HEAMD_HIP_TRY(scratch.allocate(needed));
    *out = static_cast<uint64_t*>(scratch.get());
    return HE_OK;
}

// ---- transforms by slab word type: 8-byte words run the tiled kernels with fused loads (ntt_kernels.hip), 4-byte words
// (Bfv<UInt32>: every modulus, Bsk primes included, is <= 2^30 - 1) the 4-byte kernels of word32_kernels.hip ------------
hipError_t ntt_records(bool inverse, uint64_t* slab, const PolyContext& pc, const DeviceContext& dc, uint32_t record_rows,
                       size_t records, hipStream_t stream) {
    (void)pc;
    return heamd::launch_ntt_mixed(inverse, slab, dc, record_rows, records, stream);
}
hipError_t ntt_records(bool inverse, uint32_t* slab, const PolyContext& pc, const DeviceContext& dc, uint32_t record_rows,
                       size_t records, hipStream_t stream) {
    heamd::DeviceContext32 dc32{};
    if (pc.device_context32(record_rows, dc32) != HE_OK) return hipErrorInvalidValue;
    dc32.moduli = dc.moduli;  // the caller may have substituted constants (t N^-1)
    return heamd::launch_ntt32(inverse, slab, dc32, 0, record_rows, records * record_rows, stream);
}

// Eval form over [Q, Bsk] -> Coeff over Q: (x t) -> inverse NTT -> floorQBskToQ  (Bfv+Multiply.swift:31-48)
template <typename W>
int drop_extended_base(const RnsToolLevel& tool, W* eval_qbsk, W* out, size_t polys, hipStream_t stream) {
    DeviceContext scaled = tool.qbsk->device_context();
    scaled.moduli = tool.qbsk_moduli_scaled_by_t;  // folds the multiplication by t into N^-1
    scaled.scaled_inverse_degree = 1;
    const uint32_t rows = tool.qbsk->moduli_count();
    HEAMD_HIP_TRY(ntt_records(true, eval_qbsk, *tool.qbsk, scaled, rows, polys, stream));
    HEAMD_HIP_TRY(heamd::launch_floor_qbsk_to_q(eval_qbsk, out, tool.device, polys, stream));
    return HE_OK;
}

// tensor product (Bfv+Multiply.swift:80-82) and dropExtendedBase's inverse NTT: one kernel where the degree has a tiled
// 8-byte transform (the products are formed as the inverse transform loads its row), two launches otherwise
int tensor_and_inverse(const RnsToolLevel& tool, uint64_t* lifted, uint64_t* tensor, size_t batch, hipStream_t stream,
                       bool* in_coeff_form) {
    DeviceContext scaled = tool.qbsk->device_context();
    scaled.moduli = tool.qbsk_moduli_scaled_by_t;
    scaled.scaled_inverse_degree = 1;
    const uint32_t rows = tool.qbsk->moduli_count();
    hipError_t fused = heamd::launch_ntt_tensor_inverse(lifted, tensor, scaled, rows, batch, stream);
    if (fused == hipErrorNotSupported) {
        (void)hipGetLastError();
        HEAMD_HIP_TRY(heamd::launch_tensor(lifted, tensor, tool.qbsk->device_context(), batch, stream));
        *in_coeff_form = false;
        return HE_OK;
    }
    HEAMD_HIP_TRY(fused);
    *in_coeff_form = true;
    return HE_OK;
}
int tensor_and_inverse(const RnsToolLevel& tool, uint32_t* lifted, uint32_t* tensor, size_t batch, hipStream_t stream,
                       bool* in_coeff_form) {
    const uint32_t rows = tool.qbsk->moduli_count();
    heamd::DeviceContext32 scaled{};
    if (tool.qbsk->device_context32(rows, scaled) != HE_OK) return invalid_argument("no 4-byte image of the [Q, Bsk] context");
    scaled.moduli = tool.qbsk_moduli_scaled_by_t;  // folds the multiplication by t into N^-1
    hipError_t fused = heamd::launch_ntt32_tensor_inverse(lifted, tensor, scaled, rows, batch, stream);
    if (fused == hipErrorNotSupported) {
        (void)hipGetLastError();
        HEAMD_HIP_TRY(heamd::launch_tensor(lifted, tensor, tool.qbsk->device_context(), batch, stream));
        *in_coeff_form = false;
        return HE_OK;
    }
    HEAMD_HIP_TRY(fused);
    *in_coeff_form = true;
    return HE_OK;
}

// computeBehzPolys (Bfv+Multiply.swift:51-57) for `items` pairs of two-polynomial ciphertexts: lift each of the four
// polynomials into lifted[item][0..3] (lhs -> slots 0, 1; rhs -> slots 2, 3), then forward NTT over [Q, Bsk].  Rows
// [0, L) of a lifted polynomial are the input itself (RnsTool.swift:329-330): where the tiled transform can read them
// from the ciphertexts, the lift does not write the copy.
hipError_t lift_pairs_to_eval(const RnsToolLevel& tool, uint32_t L, size_t n, size_t ext, const uint64_t* lhs,
                              const uint64_t* rhs, uint64_t* lifted, size_t items, hipStream_t stream) {
    const DeviceContext qbsk = tool.qbsk->device_context();
    const uint32_t rows = 2 * L + 1;
    const bool from_source = heamd::ntt_lifted_forward_supported(qbsk, rows, L, items * 4);
    hipError_t e = heamd::launch_lift_pair_q_to_qbsk_strided(lhs, rhs, lifted, tool.device, items, 2, 2 * L * n, 4 * ext, 0, 2 * ext,
                                                             stream, !from_source);
    if (e != hipSuccess) return e;
    if (from_source) return heamd::launch_ntt_lifted_forward(lifted, qbsk, rows, items * 4, lhs, rhs, 2 * L * n, L, stream);
    return ntt_records(false, lifted, *tool.qbsk, qbsk, rows, items * 4, stream);
}
hipError_t lift_pairs_to_eval(const RnsToolLevel& tool, uint32_t L, size_t n, size_t ext, const uint32_t* lhs,
                              const uint32_t* rhs, uint32_t* lifted, size_t items, hipStream_t stream) {
    const uint32_t rows = 2 * L + 1;
    heamd::DeviceContext32 qbsk32{};
    if (tool.qbsk->device_context32(rows, qbsk32) != HE_OK) return hipErrorInvalidValue;
    const bool from_source = heamd::ntt32_lifted_forward_supported(qbsk32) && items * 4 * rows <= (size_t(1) << 30);
    hipError_t e = heamd::launch_lift_pair_q_to_qbsk_strided(lhs, rhs, lifted, tool.device, items, 2, 2 * L * n, 4 * ext, 0, 2 * ext,
                                                             stream, !from_source);
    if (e != hipSuccess) return e;
    if (from_source) return heamd::launch_ntt32_lifted_forward(lifted, qbsk32, rows, items, lhs, rhs, 2 * L * n, L, stream);
    return ntt_records(false, lifted, *tool.qbsk, tool.qbsk->device_context(), rows, items * 4, stream);
}

// multiplyWithoutScaling's transforms and tensor product and dropExtendedBase's inverse transforms row by row
// (behz_kernels.hip): the lift writes the Bsk rows of the four operands, then ONE kernel per row band takes every
// (item, [Q, Bsk] row) from its four Coeff rows to its three scaled Coeff product rows -- the Eval rows never reach HBM.
// *fused = false (nothing launched): the degree or the batch has no such kernel; the caller runs the unfused pipeline.
constexpr bool kBehzRowsFused = true;
constexpr bool kBehzCiphertextRowsBesideLift = true;
constexpr size_t kBehzFloorParts = 2;  // <= SideLane::kStages + 1
constexpr size_t kBehzFirstPartEighths = 4;  // two parts: the first one's share of the batch, in eighths
int mul_rows_fused(const he_bfv_context* ctx, const RnsToolLevel& tool, uint32_t L, size_t n, size_t ext, const uint64_t* lhs,
                   const uint64_t* rhs, uint64_t* lifted, uint64_t* tensor, uint64_t* out, size_t batch, hipStream_t stream,
                   bool* fused, bool* floored) {
    *floored = false;
    *fused = false;
    if (!kBehzRowsFused) return HE_OK;
    DeviceContext scaled = tool.qbsk->device_context();
    scaled.moduli = tool.qbsk_moduli_scaled_by_t;  // folds dropExtendedBase's multiplication by t into N^-1
    scaled.scaled_inverse_degree = 1;
    const uint32_t rows = 2 * L + 1;
    if (!heamd::behz_rows_fused_supported(scaled, rows, L, batch)) return HE_OK;
    *fused = true;
    // (the fold butterflies of the Bsk band take the lift's rows as its reduction leaves them, below 5p)
    const bool lazy = heamd::behz_lifted_rows_may_be_lazy(scaled, rows, L);
    // The row bands that read the ciphertexts themselves (the Q rows) do not depend on the lift: on the context's side lane they
    // run beside it -- a 128-register row-fused workgroup leaves room on its CU for the lift's 256-lane workgroups of 40
    // registers (ct x ct +2.6 %, profiles/r05v_behz_q_band_beside_lift_ab.txt).  From four workgroup generations of Q rows up;
    // not while the caller's stream is being captured into a graph, nor when every lane of the context is leased
    // (side_lane.hpp).
    if (kBehzCiphertextRowsBesideLift && batch * L >= 1024) {
        heamd::LaneLease lease(heamd::lane_pool(ctx), stream);
        if (lease.lane != nullptr) {
            SideLane& lane = *lease.lane;
            HEAMD_HIP_TRY(hipEventRecord(lane.forked, stream));
            HEAMD_HIP_TRY(hipStreamWaitEvent(lane.stream, lane.forked, 0));
            hipError_t e = heamd::launch_behz_rows_fused(lhs, rhs, 2 * L * n, lifted, tensor, scaled, rows, L, batch, lane.stream,
                                                         heamd::kBehzCiphertextRows);
            if (e == hipSuccess)
                e = heamd::launch_lift_pair_q_to_qbsk_strided(lhs, rhs, lifted, tool.device, batch, 2, 2 * L * n, 4 * ext, 0, 2 * ext, stream, false, lazy);
            // The Bsk band and the floor in kBehzFloorParts parts of the batch: a part's floor -- 256-lane workgroups of 41
            // registers -- follows the Q band on the lane and runs beside the NEXT part's Bsk band on the caller's stream;
            // only the last part's floor is left to run on its own.
            const size_t parts = (kBehzFloorParts > 1 && batch >= 256 * kBehzFloorParts) ? kBehzFloorParts : 1;
            // (two parts: the first one larger -- its floor has the whole of the second part's Bsk band to run beside, and the
            // second part's floor is what is left exposed)
            size_t bounds[SideLane::kStages + 2] = {0};
            for (size_t k = 1; k <= parts; ++k) bounds[k] = batch * k / parts;
            if (parts == 2) bounds[1] = batch * kBehzFirstPartEighths / 8;
            for (size_t k = 0; k < parts; ++k) {
                const size_t first = bounds[k], items = bounds[k + 1] - first;
                if (items == 0) continue;
                if (e == hipSuccess)
                    e = heamd::launch_behz_rows_fused(lhs + first * 2 * L * n, rhs + first * 2 * L * n, 2 * L * n, lifted + first * 4 * ext,
                                                      tensor + first * 3 * ext, scaled, rows, L, items, stream, heamd::kBehzLiftedRows);
                if (k + 1 < parts) {
                    if (e == hipSuccess) e = hipEventRecord(lane.stage[k], stream);
                    if (e == hipSuccess) e = hipStreamWaitEvent(lane.stream, lane.stage[k], 0);
                    if (e == hipSuccess)
                        e = heamd::launch_floor_qbsk_to_q(tensor + first * 3 * ext, out + first * 3 * L * n, tool.device, items * 3, lane.stream);
                }
            }
            // the join is enqueued whatever happened in between: the caller's stream never runs ahead of the lane's work
            const hipError_t recorded = hipEventRecord(lane.joined, lane.stream);
            const hipError_t waited = recorded == hipSuccess ? hipStreamWaitEvent(stream, lane.joined, 0) : recorded;
            HEAMD_HIP_TRY(e);
            HEAMD_HIP_TRY(waited);
            if (parts > 1) {  // the last part's floor, behind the join (it reads the Q band's rows too)
                const size_t first = bounds[parts - 1];
                HEAMD_HIP_TRY(heamd::launch_floor_qbsk_to_q(tensor + first * 3 * ext, out + first * 3 * L * n, tool.device,
                                                            (batch - first) * 3, stream));
                *floored = true;
            }
            return HE_OK;
        }
    }
    HEAMD_HIP_TRY(heamd::launch_lift_pair_q_to_qbsk_strided(lhs, rhs, lifted, tool.device, batch, 2, 2 * L * n, 4 * ext, 0, 2 * ext, stream, false, lazy));
    HEAMD_HIP_TRY(heamd::launch_behz_rows_fused(lhs, rhs, 2 * L * n, lifted, tensor, scaled, rows, L, batch, stream));
    return HE_OK;
}

// Bfv.mulAssign(ct, ct) (Bfv/Bfv+Multiply.swift:18-85) on slabs of W
template <typename W>
int mul_pipeline(const he_bfv_context* ctx, const RnsToolLevel* tool, uint32_t L, const W* lhs, const W* rhs, W* out,
                 size_t batch, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    const size_t n = ctx->impl->degree(), ext = qbsk_poly_words(*ctx->impl, L);
    Scratch scratch(stream);
    uint64_t* raw = nullptr;
    int status = resolve_workspace(workspace, workspace_bytes, batch * 7 * ext * sizeof(W), scratch, &raw);
    if (status != HE_OK) return status;
    W* ws = reinterpret_cast<W*>(raw);
    W* lifted = ws;                   // [batch][4][2L+1][N]  (a0, a1, b0, b1)
    W* tensor = ws + batch * 4 * ext; // [batch][3][2L+1][N]
    if constexpr (std::is_same<W, uint64_t>::value) {
        bool fused = false;
        bool floored = false;
        status = mul_rows_fused(ctx, *tool, L, n, ext, lhs, rhs, lifted, tensor, out, batch, stream, &fused, &floored);
        if (status != HE_OK) return status;
        if (fused) {
            if (!floored) HEAMD_HIP_TRY(heamd::launch_floor_qbsk_to_q(tensor, out, tool->device, batch * 3, stream));
            return HE_OK;
        }
    }
    HEAMD_HIP_TRY(lift_pairs_to_eval(*tool, L, n, ext, lhs, rhs, lifted, batch, stream));
    bool in_coeff_form = false;
    status = tensor_and_inverse(*tool, lifted, tensor, batch, stream, &in_coeff_form);
    if (status != HE_OK) return status;
    if (!in_coeff_form) return drop_extended_base(*tool, tensor, out, batch * 3, stream);
    HEAMD_HIP_TRY(heamd::launch_floor_qbsk_to_q(tensor, out, tool->device, batch * 3, stream));
    return HE_OK;
}

// Bfv+Keys.swift:165-179: decompose the target over q_0..q_{L-1}, lift each piece to every key-switching modulus
// and take it to Eval.  One fused kernel where the degree has a tiled 8-byte NTT, two launches otherwise.
hipError_t spread_to_eval(const uint64_t* target, size_t target_stride, uint64_t* spread, const PolyContext& ks_ctx,
                          uint32_t L, size_t polys, hipStream_t stream) {
    const DeviceContext ks = ks_ctx.device_context();
    hipError_t e = heamd::launch_ntt_spread(target, target_stride, L, polys, spread, ks, 0, stream);
    if (e != hipErrorNotSupported) return e;
    (void)hipGetLastError();
    e = heamd::launch_key_switch_spread(target, target_stride, spread, ks, L, polys, stream);
    if (e != hipSuccess) return e;
    return heamd::launch_ntt(false, spread, ks, 0, L + 1, polys * L * (L + 1), stream);
}
hipError_t spread_to_eval(const uint32_t* target, size_t target_stride, uint32_t* spread, const PolyContext& ks_ctx,
                          uint32_t L, size_t polys, hipStream_t stream) {
    const DeviceContext ks = ks_ctx.device_context();
    heamd::DeviceContext32 ks32{};
    if (ks_ctx.device_context32(L + 1, ks32) != HE_OK) return hipErrorInvalidValue;
    hipError_t e = heamd::launch_ntt32_spread(target, target_stride, L, polys, spread, ks32, stream);
    if (e != hipErrorNotSupported) return e;
    (void)hipGetLastError();
    e = heamd::launch_key_switch_spread(target, target_stride, spread, ks, L, polys, stream);
    if (e != hipSuccess) return e;
    return ntt_records(false, spread, ks_ctx, ks, L + 1, polys * L, stream);
}
// Bfv+Keys.swift:180-207: the lazy inner product of the decomposed target with the key, then back to Coeff.  One
// kernel where the degree has a tiled 8-byte NTT (the sums are formed as the inverse transform loads its row).
hipError_t key_mac_to_coeff(const uint64_t* spread, const uint64_t* key, uint64_t* prod, const PolyContext& ks_ctx,
                            uint32_t L, uint32_t top_rows, size_t polys, hipStream_t stream) {
    const DeviceContext ks = ks_ctx.device_context();
    hipError_t e = heamd::launch_ntt_key_mac_inverse(spread, key, prod, ks, L, top_rows, polys, stream);
    if (e != hipErrorNotSupported) return e;
    (void)hipGetLastError();
    e = heamd::launch_key_switch_mac(spread, key, prod, ks, L, top_rows, polys, stream);
    if (e != hipSuccess) return e;
    return heamd::launch_ntt(true, prod, ks, 0, L + 1, polys * 2 * (L + 1), stream);
}
hipError_t key_mac_to_coeff(const uint32_t* spread, const uint32_t* key, uint32_t* prod, const PolyContext& ks_ctx,
                            uint32_t L, uint32_t top_rows, size_t polys, hipStream_t stream) {
    const DeviceContext ks = ks_ctx.device_context();
    heamd::DeviceContext32 ks32{};
    if (ks_ctx.device_context32(L + 1, ks32) != HE_OK) return hipErrorInvalidValue;
    hipError_t e = heamd::launch_ntt32_key_mac_inverse(spread, key, prod, ks32, L, top_rows, polys, stream);
    if (e != hipErrorNotSupported) return e;
    (void)hipGetLastError();
    e = heamd::launch_key_switch_mac(spread, key, prod, ks, L, top_rows, polys, stream);
    if (e != hipSuccess) return e;
    return ntt_records(true, prod, ks_ctx, ks, L + 1, polys * 2, stream);
}

// The inner product with the key, the inverse transforms and the key switch's last step (relinearize: added_polys = 2) in
// two launches of one kernel where the degree has it (ntt_kernels.hip kInverseFromKeyMacFinish); hipErrorNotSupported
// otherwise (nothing launched).
hipError_t key_mac_and_finish(const uint64_t* spread, const uint64_t* key, uint64_t* prod, const uint64_t* ct_base,
                              size_t ct_stride, uint64_t* out, const PolyContext& ks_ctx, uint32_t L, uint32_t top_rows,
                              size_t polys, uint32_t added_polys, hipStream_t stream) {
    const heamd::KeySwitchEnd end{ct_base, ct_stride, out, added_polys};
    return heamd::launch_ntt_key_mac_inverse_finish(spread, key, prod, end, ks_ctx.device_context(), L, top_rows, polys,
                                                    stream);
}
hipError_t key_mac_and_finish(const uint32_t* spread, const uint32_t* key, uint32_t* prod, const uint32_t* ct_base,
                              size_t ct_stride, uint32_t* out, const PolyContext& ks_ctx, uint32_t L, uint32_t top_rows,
                              size_t polys, uint32_t added_polys, hipStream_t stream) {
    heamd::DeviceContext32 ks32{};
    if (ks_ctx.device_context32(L + 1, ks32) != HE_OK) return hipErrorNotSupported;
    return heamd::launch_ntt32_key_mac_inverse_finish(spread, key, prod, ct_base, ct_stride, out, ks32, L, top_rows, polys,
                                                      added_polys, stream);
}

// _computeKeySwitchingUpdate (Bfv+Keys.swift:123-208) on polynomial `target` of every item, then
// out[item][c] = (c < added_polys ? ct[item][c] : 0) + update[item][c]  (relinearize: added 2; applyGalois: added 1)
template <typename W>
int key_switch_pipeline(const he_bfv_context* ctx, uint32_t L, const W* target, size_t target_stride, const W* ct_base,
                        size_t ct_stride, const W* key, W* out, size_t batch, uint32_t added_polys, W* spread, W* prod,
                        hipStream_t stream) {
    const PolyContext* ks_ctx = ctx->impl->key_switching(L);
    HEAMD_HIP_TRY(spread_to_eval(target, target_stride, spread, *ks_ctx, L, batch, stream));
    const hipError_t fused = key_mac_and_finish(spread, key, prod, ct_base, ct_stride, out, *ks_ctx, L,
                                                ctx->impl->top_level() + 1, batch, added_polys, stream);
    if (fused != hipErrorNotSupported) {
        HEAMD_HIP_TRY(fused);
        return HE_OK;
    }
    HEAMD_HIP_TRY(key_mac_to_coeff(spread, key, prod, *ks_ctx, L, ctx->impl->top_level() + 1, batch, stream));
    HEAMD_HIP_TRY(heamd::launch_key_switch_finish(prod, ct_base, ct_stride, out, ks_ctx->device_context(), L, batch,
                                                  added_polys, stream));
    return HE_OK;
}

// The same with one key per group of `group_size` consecutive items (ciphertexts of different clients in one batch):
// decomposition and finish run over the whole batch, only the inner product with the key is launched per run of equal
// keys.
template <typename W>
int key_switch_pipeline_grouped(const he_bfv_context* ctx, uint32_t L, const W* target, size_t target_stride,
                                const W* ct_base, size_t ct_stride, const W* const* keys, size_t groups, size_t group_size,
                                W* out, uint32_t added_polys, W* spread, W* prod, hipStream_t stream) {
    const PolyContext* ks_ctx = ctx->impl->key_switching(L);
    const size_t n = ctx->impl->degree(), batch = groups * group_size;
    HEAMD_HIP_TRY(spread_to_eval(target, target_stride, spread, *ks_ctx, L, batch, stream));
    for (size_t g = 0; g < groups;) {
        size_t run = 1;
        while (g + run < groups && keys[g + run] == keys[g]) ++run;
        const size_t first = g * group_size, polys = run * group_size;
        HEAMD_HIP_TRY(key_mac_to_coeff(static_cast<const W*>(spread) + first * L * (L + 1) * n, keys[g],
                                       prod + first * 2 * (L + 1) * n, *ks_ctx, L, ctx->impl->top_level() + 1, polys,
                                       stream));
        g += run;
    }
    HEAMD_HIP_TRY(heamd::launch_key_switch_finish(static_cast<const W*>(prod), ct_base, ct_stride, out,
                                                  ks_ctx->device_context(), L, batch, added_polys, stream));
    return HE_OK;
}

// g^-1 mod 2N for odd g (Newton iteration doubles the correct low bits)
uint32_t inverse_mod_power_of_two(uint64_t g, uint64_t modulus) {
    uint64_t x = g;  // correct to 3 bits
    for (int k = 0; k < 6; ++k) x *= 2 - g * x;
    return static_cast<uint32_t>(x & (modulus - 1));
}

// Bfv.applyGalois (Bfv.swift:174-198) without a rotated copy of the ciphertext: the automorphism of c1 happens as the
// decomposition's transform loads its rows, that of c0 as the last kernel adds it to the update; expand_shift != 0
// turns that last kernel into one step of PirUtil.expand (rns_kernels.hpp, launch_galois_finish).  One key per group of
// `group_size` consecutive ciphertexts.  hipErrorNotSupported (nothing launched): the degree has no tiled transform.
// `own` (expand steps): the ciphertexts the children are formed with when they are not `ct` (launch_galois_finish).
hipError_t galois_switch_fused(const he_bfv_context* ctx, uint32_t L, const uint64_t* ct, uint32_t galois_inverse,
                               const uint64_t* const* keys, size_t groups, size_t group_size, uint64_t* out,
                               uint32_t expand_shift, const heamd::ExpandTargets& targets, uint64_t* spread, uint64_t* prod,
                               hipStream_t stream, const uint64_t* own = nullptr) {
    const PolyContext* ks_ctx = ctx->impl->key_switching(L);
    const DeviceContext ks = ks_ctx->device_context();
    const size_t n = ctx->impl->degree(), batch = groups * group_size, ct_stride = 2 * size_t(L) * n;
    hipError_t e = heamd::launch_ntt_spread(ct + size_t(L) * n, ct_stride, L, batch, spread, ks, galois_inverse, stream);
    if (e != hipSuccess) return e;
    // the key switch's end (galois(c0) + update0 | update1, or the two children of an expand step) in the key-MAC
    // transform's store where the degree has that kernel; the separate finish kernel otherwise
    // -- decided on the smallest run of queries with one key: every run is two launches of its own there, against one
    // key-MAC launch per run and ONE finish kernel over the whole batch (profiles/r04m_galois_fused_end_ab.txt)
    size_t smallest_run = groups;
    for (size_t g = 0; g < groups;) {
        size_t run = 1;
        while (g + run < groups && keys[g + run] == keys[g]) ++run;
        smallest_run = run < smallest_run ? run : smallest_run;
        g += run;
    }
    const bool fused_end = heamd::ntt_key_mac_finish_supported(ks, L, smallest_run * group_size);
    for (size_t g = 0; g < groups;) {
        size_t run = 1;
        while (g + run < groups && keys[g + run] == keys[g]) ++run;
        const size_t first = g * group_size, polys = run * group_size;
        const uint64_t* const run_spread = static_cast<const uint64_t*>(spread) + first * L * (L + 1) * n;
        uint64_t* const run_prod = prod + first * 2 * (L + 1) * n;
        if (fused_end) {
            heamd::KeySwitchEnd end{ct, ct_stride, out, 1u};
            end.galois_inverse = galois_inverse;
            end.expand_shift = expand_shift;
            end.own_base = own;
            end.targets_table = targets.table;
            end.targets_group_size = targets.group_size;
            end.targets_group_stride = targets.group_stride;
            end.poly_base = first;
            e = heamd::launch_ntt_key_mac_inverse_finish(run_spread, keys[g], run_prod, end, ks, L, ctx->impl->top_level() + 1,
                                                         polys, stream);
        } else {
            e = key_mac_to_coeff(run_spread, keys[g], run_prod, *ks_ctx, L, ctx->impl->top_level() + 1, polys, stream);
        }
        if (e != hipSuccess) return e;
        g += run;
    }
    if (fused_end) return hipSuccess;
    return heamd::launch_galois_finish(static_cast<const uint64_t*>(prod), ct, ct_stride, out, ks, L, batch,
                                       galois_inverse, expand_shift, targets, stream, own);
}

template <typename W>
int relinearize_pipeline(const he_bfv_context* ctx, uint32_t L, const W* ct3, const W* key, W* out, size_t batch,
                         void* workspace, size_t workspace_bytes, hipStream_t stream) {
    const size_t n = ctx->impl->degree();
    Scratch scratch(stream);
    uint64_t* raw = nullptr;
    const size_t words = batch * (size_t(L) * (L + 1) + 2 * (L + 1)) * n;
    int status = resolve_workspace(workspace, workspace_bytes, words * sizeof(W), scratch, &raw);
    if (status != HE_OK) return status;
    W* spread = reinterpret_cast<W*>(raw);                // [batch][L][L+1][N]
    W* prod = spread + batch * L * (L + 1) * n;           // [batch][2][L+1][N]
    const size_t ct_stride = 3 * size_t(L) * n;
    return key_switch_pipeline(ctx, L, ct3 + 2 * size_t(L) * n, ct_stride, ct3, ct_stride, key, out, batch, 2, spread, prod,
                               stream);
}

}  // namespace

namespace heamd {
// One level of PirUtil.expand (PirUtil.swift:204-236) for parents whose Galois element has its own key: the children
// (parent + applyGalois(parent), (parent - applyGalois(parent)) x^shift) leave the key switch's last kernel directly.
// leaf_table != nullptr: the children are leaves and go to their output slots in `next` = the expansion's output
// (rns_kernels.hpp, ExpandTargets; leaf_stride = outputs per query).
// Returns kExpandStepUnavailable when the degree has no tiled transform (the caller composes the step from
// he_bfv_apply_galois_grouped_device and the expand-step kernel instead).
int bfv_expand_step_fused(const he_bfv_context* ctx, uint32_t L, const uint64_t* parents, uint64_t element,
                          const uint64_t* const* keys, size_t groups, size_t group_size, uint64_t* next, uint32_t shift,
                          const uint32_t* leaf_table, size_t leaf_stride, void* workspace, size_t workspace_bytes,
                          hipStream_t stream, const uint64_t* rotated) {
    const size_t n = ctx->impl->degree(), batch = groups * group_size;
    if (batch == 0) return HE_OK;
    if (workspace_bytes < he_bfv_apply_galois_workspace_bytes(ctx, L, batch)) return invalid_argument("workspace too small");
    uint64_t* spread = static_cast<uint64_t*>(workspace);    // [batch][L][L+1][N]
    uint64_t* prod = spread + batch * L * (L + 1) * n;       // [batch][2][L+1][N]
    const hipError_t e = galois_switch_fused(ctx, L, rotated != nullptr ? rotated : parents,
                                             inverse_mod_power_of_two(element, 2 * n), keys, groups, group_size, next, shift,
                                             heamd::ExpandTargets{leaf_table, group_size, leaf_stride}, spread, prod, stream,
                                             rotated != nullptr ? parents : nullptr);
    if (e == hipErrorNotSupported) {
        (void)hipGetLastError();
        return kExpandStepUnavailable;
    }
    HEAMD_HIP_TRY(e);
    return HE_OK;
}
}  // namespace heamd

extern "C" {

int he_bfv_context_create(uint32_t degree, uint64_t plaintext_modulus, const uint64_t* coefficient_moduli,
                          uint32_t moduli_count, he_bfv_context** out) {
    return bfv_create(degree, plaintext_modulus, coefficient_moduli, moduli_count, false, out);
}
int he_bfv_context_create_host_only(uint32_t degree, uint64_t plaintext_modulus, const uint64_t* coefficient_moduli,
                                    uint32_t moduli_count, he_bfv_context** out) {
    return bfv_create(degree, plaintext_modulus, coefficient_moduli, moduli_count, true, out);
}
int he_bfv_context_create_u32(uint32_t degree, uint64_t plaintext_modulus, const uint64_t* coefficient_moduli,
                              uint32_t moduli_count, he_bfv_context** out) {
    return bfv_create(degree, plaintext_modulus, coefficient_moduli, moduli_count, false, out, 32);
}
void he_bfv_context_destroy(he_bfv_context* ctx) {
    heamd::RelaxedCapture relaxed;  // (api_internal.hpp: safe beside another thread's -- or this thread's -- graph capture)
    delete ctx;
}
uint32_t he_bfv_ciphertext_moduli_count(const he_bfv_context* ctx) { return ctx ? ctx->impl->top_level() : 0; }
const he_poly_context* he_bfv_ciphertext_context(const he_bfv_context* ctx, uint32_t k) {
    return (ctx && ctx->impl->valid(k)) ? ctx->ciphertext[k].get() : nullptr;
}
const he_poly_context* he_bfv_key_switching_context(const he_bfv_context* ctx, uint32_t k) {
    return (ctx && ctx->impl->valid(k)) ? ctx->key_switching[k].get() : nullptr;
}
const he_poly_context* he_bfv_qbsk_context(const he_bfv_context* ctx, uint32_t k) {
    return (ctx && ctx->impl->valid(k)) ? ctx->qbsk[k].get() : nullptr;
}
int he_bfv_copy_bsk_moduli(const he_bfv_context* ctx, uint64_t* out_bsk) {
    if (ctx == nullptr || out_bsk == nullptr) return invalid_argument("null pointer");
    const auto& list = ctx->impl->bsk_mtilde();
    for (size_t i = 0; i + 1 < list.size(); ++i) out_bsk[i] = list[i];
    return HE_OK;
}

int he_rns_lift_q_to_qbsk_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* in, uint64_t* out,
                                 size_t batch, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    if (batch == 0) return HE_OK;
    if (in == nullptr || out == nullptr) return invalid_argument("null slab");
    HEAMD_HIP_TRY(heamd::launch_lift_q_to_qbsk(in, out, tool->device, batch, as_stream(s)));
    return HE_OK;
}
int he_rns_floor_qbsk_to_q_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* in, uint64_t* out,
                                  size_t batch, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    if (batch == 0) return HE_OK;
    if (in == nullptr || out == nullptr) return invalid_argument("null slab");
    HEAMD_HIP_TRY(heamd::launch_floor_qbsk_to_q(in, out, tool->device, batch, as_stream(s)));
    return HE_OK;
}

// ------------------------------------------------------------------------------------------ ct x ct
size_t he_bfv_mul_workspace_bytes(const he_bfv_context* ctx, uint32_t moduli_count, size_t batch) {
    if (ctx == nullptr || !ctx->impl->valid(moduli_count)) return 0;
    return batch * 7 * qbsk_poly_words(*ctx->impl, moduli_count) * sizeof(uint64_t);  // 4 lifted + 3 tensor polys
}

extern "C++" {
namespace {
template <typename W>
int mul_entry(const he_bfv_context* ctx, uint32_t moduli_count, const W* lhs, const W* rhs, W* out, size_t batch,
              void* workspace, size_t workspace_bytes, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    if (sizeof(W) == 4 && ctx->impl->word_bits() != 32) return invalid_argument("4-byte slabs need a Bfv<UInt32> context");
    if (batch == 0) return HE_OK;
    if (lhs == nullptr || rhs == nullptr || out == nullptr) return invalid_argument("null ciphertext");
    {
        // out must not overlap lhs or rhs: with the batch in parts, one part's floor stores its products while another
        // part's row band still reads the operands -- whether that happens depends on the batch size and the moduli, so an
        // overlap would be right for some shapes and silently wrong for others
        const size_t words = size_t(moduli_count) * ctx->impl->degree();
        const uintptr_t out_begin = reinterpret_cast<uintptr_t>(out), out_end = out_begin + batch * 3 * words * sizeof(W);
        for (const W* operand : {lhs, rhs}) {
            const uintptr_t begin = reinterpret_cast<uintptr_t>(operand), end = begin + batch * 2 * words * sizeof(W);
            if (out_begin < end && begin < out_end) return invalid_argument("ct x ct: out overlaps an operand");
        }
    }
    return mul_pipeline(ctx, tool, moduli_count, lhs, rhs, out, batch, workspace, workspace_bytes, as_stream(s));
}
}  // namespace
}  // extern "C++"

int he_bfv_mul_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* lhs, const uint64_t* rhs,
                      uint64_t* out, size_t batch, void* workspace, size_t workspace_bytes, he_stream s) {
    return mul_entry(ctx, moduli_count, lhs, rhs, out, batch, workspace, workspace_bytes, s);
}
int he_bfv_mul_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, const uint32_t* lhs, const uint32_t* rhs,
                          uint32_t* out, size_t batch, void* workspace, size_t workspace_bytes, he_stream s) {
    return mul_entry(ctx, moduli_count, lhs, rhs, out, batch, workspace, workspace_bytes, s);
}

// ------------------------------------------------------------------------------------------ relinearize
size_t he_bfv_relinearize_workspace_bytes(const he_bfv_context* ctx, uint32_t moduli_count, size_t batch) {
    if (ctx == nullptr || !ctx->impl->valid(moduli_count)) return 0;
    const size_t L = moduli_count, n = ctx->impl->degree();
    return batch * (L * (L + 1) + 2 * (L + 1)) * n * sizeof(uint64_t);
}

extern "C++" {
namespace {
template <typename W>
int relinearize_entry(const he_bfv_context* ctx, uint32_t moduli_count, const W* ct3, const W* key, W* out, size_t batch,
                      void* workspace, size_t workspace_bytes, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    if (sizeof(W) == 4 && ctx->impl->word_bits() != 32) return invalid_argument("4-byte slabs need a Bfv<UInt32> context");
    if (key == nullptr || !ctx->impl->has_key_switching()) {
        heamd::set_last_error("no relinearization key");
        return HE_ERR_MISSING_RELINEARIZATION_KEY;  // Bfv.swift:208-210
    }
    if (batch == 0) return HE_OK;
    if (ct3 == nullptr || out == nullptr) return invalid_argument("null ciphertext");
    {
        // out must not overlap ct3: the last kernel adds the update to (c0, c1) of one item while it stores another item's
        // result (items are 3 L N words apart in ct3 and 2 L N apart in out), in no particular order -- and which kernel
        // that is depends on the batch size, so an overlap would be right for some batches and silently wrong for others.
        // The one overlap that is word-for-word in place -- a single ciphertext relinearized onto its own (c0, c1) -- is
        // allowed: it takes the element-wise finish kernel (a batch of one is never fused).
        const size_t words = size_t(moduli_count) * ctx->impl->degree();
        const uintptr_t in_begin = reinterpret_cast<uintptr_t>(ct3), in_end = in_begin + batch * 3 * words * sizeof(W);
        const uintptr_t out_begin = reinterpret_cast<uintptr_t>(out), out_end = out_begin + batch * 2 * words * sizeof(W);
        const bool overlap = out_begin < in_end && in_begin < out_end;
        if (overlap && !(batch == 1 && out_begin == in_begin))
            return invalid_argument("relinearize: out overlaps ct3 (only a single ciphertext may be relinearized in place)");
    }
    return relinearize_pipeline(ctx, moduli_count, ct3, key, out, batch, workspace, workspace_bytes, as_stream(s));
}
}  // namespace
}  // extern "C++"

int he_bfv_relinearize_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* ct3,
                              const uint64_t* key, uint64_t* out, size_t batch, void* workspace,
                              size_t workspace_bytes, he_stream s) {
    return relinearize_entry(ctx, moduli_count, ct3, key, out, batch, workspace, workspace_bytes, s);
}
int he_bfv_relinearize_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, const uint32_t* ct3,
                                  const uint32_t* key, uint32_t* out, size_t batch, void* workspace,
                                  size_t workspace_bytes, he_stream s) {
    return relinearize_entry(ctx, moduli_count, ct3, key, out, batch, workspace, workspace_bytes, s);
}

// ------------------------------------------------------------------------------------------ Galois automorphism
extern "C++" {
namespace {
bool is_valid_galois_element(uint64_t element, uint64_t degree) {  // Galois.swift:100-105
    return degree != 0 && (degree & (degree - 1)) == 0 && (element & 1) == 1 && element < (degree << 1) && element > 1;
}
}  // namespace
}  // extern "C++"

// GaloisElement.swappingRows(degree:) / rotatingColumns(by:degree:) (PolyRq/Galois.swift:174-212): the elements that
// Bfv.swapRows / rotateColumns pass to applyGalois (HeScheme.swift:1047-1094)
int he_galois_element_swapping_rows(uint64_t degree, uint64_t* out_element) {
    if (out_element == nullptr) return invalid_argument("null out");
    *out_element = (degree << 1) - 1;
    return HE_OK;
}
int he_galois_element_rotating_columns(int64_t step, uint64_t degree, uint64_t* out_element) {
    if (out_element == nullptr) return invalid_argument("null out");
    if (degree == 0 || (degree & (degree - 1)) != 0) return HE_ERR_INVALID_DEGREE;
    uint64_t positive = static_cast<uint64_t>(step < 0 ? -step : step);
    if (!(positive < (degree >> 1) && positive > 0)) return invalid_argument("rotation step out of range");  // invalidRotationStep
    positive &= (degree << 1) - 1;
    if (step > 0) positive = (degree >> 1) - positive;
    // GaloisElementGenerator.value = 3, to the power `positive` mod 2N
    const uint64_t modulus = degree << 1;
    uint64_t result = 1, base = 3 % modulus;
    for (uint64_t e = positive; e != 0; e >>= 1) {
        if (e & 1) result = result * base % modulus;
        base = base * base % modulus;
    }
    *out_element = result;
    return HE_OK;
}

size_t he_bfv_apply_galois_workspace_bytes(const he_bfv_context* ctx, uint32_t moduli_count, size_t batch) {
    if (ctx == nullptr || !ctx->impl->valid(moduli_count)) return 0;
    const size_t L = moduli_count, n = ctx->impl->degree();
    return batch * (2 * L + L * (L + 1) + 2 * (L + 1)) * n * sizeof(uint64_t);
}

extern "C++" {
namespace {
template <typename W>
int apply_galois_entry(const he_bfv_context* ctx, uint32_t moduli_count, const W* ct, uint64_t element, const W* galois_key,
                       W* out, size_t batch, void* workspace, size_t workspace_bytes, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    if (sizeof(W) == 4 && ctx->impl->word_bits() != 32) return invalid_argument("4-byte slabs need a Bfv<UInt32> context");
    if (galois_key == nullptr || !ctx->impl->has_key_switching()) {
        heamd::set_last_error("no Galois key for this element");
        return HE_ERR_MISSING_GALOIS_KEY;  // Bfv.swift:184-189
    }
    if (!is_valid_galois_element(element, ctx->impl->degree())) return invalid_argument("invalid Galois element");
    if (batch == 0) return HE_OK;
    if (ct == nullptr || out == nullptr) return invalid_argument("null ciphertext");
    hipStream_t stream = as_stream(s);
    const uint32_t L = moduli_count;
    const size_t n = ctx->impl->degree();
    const PolyContext* q_ctx = ctx->impl->ciphertext(L);
    Scratch scratch(stream);
    uint64_t* raw = nullptr;
    const size_t words = batch * (2 * size_t(L) + size_t(L) * (L + 1) + 2 * (L + 1)) * n;
    status = resolve_workspace(workspace, workspace_bytes, words * sizeof(W), scratch, &raw);
    if (status != HE_OK) return status;
    W* rotated = reinterpret_cast<W*>(raw);                // [batch][2][L][N]
    W* spread = rotated + batch * 2 * L * n;               // [batch][L][L+1][N]
    W* prod = spread + batch * L * (L + 1) * n;            // [batch][2][L+1][N]
    const size_t ct_stride = 2 * size_t(L) * n;
    // Bfv.swift:190-196: c0' = galois(c0) + update0, c1' = update1, update = keySwitch(galois(c1))
    const uint32_t galois_inverse = inverse_mod_power_of_two(element, 2 * n);
    if constexpr (sizeof(W) == 8) {
        if (static_cast<const void*>(ct) != static_cast<const void*>(out)) {  // the fused kernels read ct to the end
            const uint64_t* key = reinterpret_cast<const uint64_t*>(galois_key);
            const hipError_t e = galois_switch_fused(ctx, L, reinterpret_cast<const uint64_t*>(ct), galois_inverse, &key, 1,
                                                     batch, reinterpret_cast<uint64_t*>(out), 0,
                                                     heamd::ExpandTargets{nullptr, 1, 0}, reinterpret_cast<uint64_t*>(spread),
                                                     reinterpret_cast<uint64_t*>(prod), stream);
            if (e == hipSuccess) return HE_OK;
            if (e != hipErrorNotSupported) HEAMD_HIP_TRY(e);
            (void)hipGetLastError();
        }
    }
    HEAMD_HIP_TRY(heamd::launch_galois_coeff(ct, rotated, q_ctx->device_context(L), galois_inverse, batch * 2 * L, stream));
    return key_switch_pipeline(ctx, L, static_cast<const W*>(rotated) + size_t(L) * n, ct_stride,
                               static_cast<const W*>(rotated), ct_stride, galois_key, out, batch, 1, spread, prod, stream);
}
}  // namespace
}  // extern "C++"

int he_bfv_apply_galois_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* ct, uint64_t element,
                               const uint64_t* galois_key, uint64_t* out, size_t batch, void* workspace,
                               size_t workspace_bytes, he_stream s) {
    return apply_galois_entry(ctx, moduli_count, ct, element, galois_key, out, batch, workspace, workspace_bytes, s);
}
int he_bfv_apply_galois_grouped_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* ct,
                                       uint64_t element, const uint64_t* const* galois_keys, size_t groups,
                                       size_t group_size, uint64_t* out, void* workspace, size_t workspace_bytes,
                                       he_stream s) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    if (galois_keys == nullptr || !ctx->impl->has_key_switching()) {
        heamd::set_last_error("no Galois key for this element");
        return HE_ERR_MISSING_GALOIS_KEY;
    }
    for (size_t g = 0; g < groups; ++g)
        if (galois_keys[g] == nullptr) {
            heamd::set_last_error("no Galois key for this element");
            return HE_ERR_MISSING_GALOIS_KEY;
        }
    if (!is_valid_galois_element(element, ctx->impl->degree())) return invalid_argument("invalid Galois element");
    const size_t batch = groups * group_size;
    if (batch == 0) return HE_OK;
    if (ct == nullptr || out == nullptr) return invalid_argument("null ciphertext");
    hipStream_t stream = as_stream(s);
    const uint32_t L = moduli_count;
    const size_t n = ctx->impl->degree();
    const PolyContext* q_ctx = ctx->impl->ciphertext(L);
    Scratch scratch(stream);
    uint64_t* ws = nullptr;
    status = resolve_workspace(workspace, workspace_bytes, he_bfv_apply_galois_workspace_bytes(ctx, L, batch), scratch, &ws);
    if (status != HE_OK) return status;
    uint64_t* rotated = ws;                                     // [batch][2][L][N]
    uint64_t* spread = rotated + batch * 2 * L * n;             // [batch][L][L+1][N]
    uint64_t* prod = spread + batch * L * (L + 1) * n;          // [batch][2][L+1][N]
    const size_t ct_stride = 2 * size_t(L) * n;
    const uint32_t galois_inverse = inverse_mod_power_of_two(element, 2 * n);
    if (ct != out) {
        const hipError_t e =
            galois_switch_fused(ctx, L, ct, galois_inverse, galois_keys, groups, group_size, out, 0,
                                heamd::ExpandTargets{nullptr, 1, 0}, spread, prod, stream);
        if (e == hipSuccess) return HE_OK;
        if (e != hipErrorNotSupported) HEAMD_HIP_TRY(e);
        (void)hipGetLastError();
    }
    HEAMD_HIP_TRY(heamd::launch_galois_coeff(ct, rotated, q_ctx->device_context(L), galois_inverse, batch * 2 * L, stream));
    return key_switch_pipeline_grouped<uint64_t>(ctx, L, rotated + size_t(L) * n, ct_stride, rotated, ct_stride,
                                                 galois_keys, groups, group_size, out, 1, spread, prod, stream);
}
int he_bfv_apply_galois_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, const uint32_t* ct, uint64_t element,
                                   const uint32_t* galois_key, uint32_t* out, size_t batch, void* workspace,
                                   size_t workspace_bytes, he_stream s) {
    return apply_galois_entry(ctx, moduli_count, ct, element, galois_key, out, batch, workspace, workspace_bytes, s);
}

// ------------------------------------------------------------------------------------------ scaleAndRound
int he_rns_scale_and_round_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* in,
                                  uint64_t scaling_factor, uint64_t* out, size_t batch, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    if (batch == 0) return HE_OK;
    if (in == nullptr || out == nullptr) return invalid_argument("null polynomial");
    const uint64_t t = ctx->impl->plaintext_modulus();
    if (scaling_factor >= t) return invalid_argument("scaling factor not reduced mod t");
    // inverseGammaModT.multiplyMod(scalingFactor) (RnsTool.swift:298)
    const uint64_t scaled = heamd::mul_mod(tool->device.inv_gamma_mod_t, scaling_factor, t);
    const heamd::U64x2 final_scale{scaled, heamd::shoup_factor(scaled, t)};
    HEAMD_HIP_TRY(heamd::launch_scale_and_round(in, out, tool->device, final_scale, batch, as_stream(s)));
    return HE_OK;
}

// ------------------------------------------------------------------------------------------ plaintext <-> Eval
int he_bfv_plaintext_to_eval_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* plaintext,
                                    uint64_t* out, size_t batch, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    if (batch == 0) return HE_OK;
    if (plaintext == nullptr || out == nullptr) return invalid_argument("null plaintext");
    hipStream_t stream = as_stream(s);
    const PolyContext* q_ctx = ctx->impl->ciphertext(moduli_count);
    const DeviceContext dc = q_ctx->device_context(moduli_count);
    // Plaintext.convertToEvalFormat (Plaintext.swift:149-170): centered lift, then forwardNtt -- one kernel where the
    // degree has a tiled NTT (the lift rides the transform's load), two launches otherwise
    hipError_t fused = heamd::launch_ntt_lift(plaintext, ctx->impl->plaintext_modulus(), batch, out, dc, stream);
    if (fused != hipErrorNotSupported) {
        HEAMD_HIP_TRY(fused);
        return HE_OK;
    }
    (void)hipGetLastError();
    HEAMD_HIP_TRY(heamd::launch_plaintext_lift(plaintext, out, dc, ctx->impl->plaintext_modulus(), batch, stream));
    HEAMD_HIP_TRY(heamd::launch_ntt(false, out, dc, 0, moduli_count, batch * moduli_count, stream));
    return HE_OK;
}

int he_bfv_plaintext_to_coeff_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* plaintext_eval,
                                     uint64_t* out, size_t batch, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    if (batch == 0) return HE_OK;
    if (plaintext_eval == nullptr || out == nullptr) return invalid_argument("null plaintext");
    hipStream_t stream = as_stream(s);
    const PolyContext* q_ctx = ctx->impl->ciphertext(moduli_count);
    const DeviceContext dc = q_ctx->device_context(moduli_count);
    const size_t n = ctx->impl->degree();
    // Plaintext.convertToCoeffFormat (Plaintext.swift:176-191) keeps residue row 0 only and rows are independent
    // under the inverse NTT, so only row 0 is transformed
    HEAMD_HIP_TRY(heamd::launch_first_rows(plaintext_eval, out, dc, batch, stream));
    HEAMD_HIP_TRY(heamd::launch_ntt(true, out, dc, 0, 1, batch, stream));
    HEAMD_HIP_TRY(heamd::launch_plaintext_unlift(out, q_ctx->moduli()[0], ctx->impl->plaintext_modulus(), batch * n,
                                                 stream));
    return HE_OK;
}

// ------------------------------------------------------------------------------------------ mod switch, ct x pt
int he_bfv_mod_switch_down_device(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count,
                                  const uint64_t* in, uint64_t* out, size_t batch, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    if (moduli_count < 2) return HE_ERR_INVALID_POLY_CONTEXT;  // PolyRq.swift:366-368
    if (batch == 0 || poly_count == 0) return HE_OK;
    if (in == nullptr || out == nullptr) return invalid_argument("null ciphertext");
    const PolyContext* pc = ctx->impl->ciphertext(moduli_count);
    HEAMD_HIP_TRY(heamd::launch_divide_and_round_q_last(in, out, pc->device_context(), moduli_count,
                                                        batch * poly_count, as_stream(s)));
    return HE_OK;
}

// Ciphertext.modSwitchDownToSingle (Bfv.swift:163-171): moduli_count -> 1 moduli, one kernel for 2..8 moduli
int he_bfv_mod_switch_down_to_single_device(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count,
                                            const uint64_t* in, uint64_t* out, size_t batch, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    if (batch == 0 || poly_count == 0) return HE_OK;
    if (in == nullptr || out == nullptr) return invalid_argument("null ciphertext");
    hipStream_t stream = as_stream(s);
    const size_t polys = batch * poly_count, n = ctx->impl->degree();
    if (moduli_count == 1) {  // already there
        if (in != out) HEAMD_HIP_TRY(hipMemcpyAsync(out, in, polys * n * sizeof(uint64_t), hipMemcpyDeviceToDevice, stream));
        return HE_OK;
    }
    const PolyContext* pc = ctx->impl->ciphertext(moduli_count);
    hipError_t e = heamd::launch_mod_switch_down_to_single(in, out, pc->device_context(), moduli_count, polys, stream);
    if (e != hipErrorNotSupported) {
        HEAMD_HIP_TRY(e);
        return HE_OK;
    }
    (void)hipGetLastError();
    // more than 8 moduli (or degree 1): step by step through two scratch slabs
    Scratch level_mem(stream);
    HEAMD_HIP_TRY(level_mem.allocate(2 * polys * size_t(moduli_count - 1) * n * sizeof(uint64_t)));
    uint64_t* ping = static_cast<uint64_t*>(level_mem.get());
    uint64_t* pong = ping + polys * size_t(moduli_count - 1) * n;
    const uint64_t* current = in;
    for (uint32_t level = moduli_count; level > 1; --level) {
        uint64_t* target = level == 2 ? out : (current == ping ? pong : ping);
        HEAMD_HIP_TRY(heamd::launch_divide_and_round_q_last(current, target, ctx->impl->ciphertext(level)->device_context(),
                                                            level, polys, stream));
        current = target;
    }
    return HE_OK;
}

int he_bfv_mul_plain_device(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count, uint64_t* ct,
                            const uint64_t* pt, size_t batch, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    if (batch == 0 || poly_count == 0) return HE_OK;
    if (ct == nullptr || pt == nullptr) return invalid_argument("null operand");
    const PolyContext* pc = ctx->impl->ciphertext(moduli_count);
    HEAMD_HIP_TRY(heamd::launch_mul_plain(ct, pt, pc->device_context(), poly_count, batch, as_stream(s)));
    return HE_OK;
}

extern "C++" {
namespace {
// Bfv.addAssignCoeff / subAssignCoeff(_: inout CoeffCiphertext, _: CoeffPlaintext) (Bfv/Bfv.swift:110-117) =
// plaintextTranslate (Bfv/Bfv+Encrypt.swift:75-140)
template <typename W>
int plaintext_translate(const he_bfv_context* ctx, const RnsToolLevel* tool, uint32_t poly_count, W* ct, const W* plaintexts,
                        bool subtract, size_t batch, he_stream s) {
    if (poly_count == 0) return invalid_argument("a ciphertext has at least one polynomial");
    if (batch == 0) return HE_OK;
    if (ct == nullptr || plaintexts == nullptr) return invalid_argument("null operand");
    (void)ctx;
    HEAMD_HIP_TRY(heamd::launch_plaintext_translate(ct, plaintexts, tool->device, poly_count, subtract, batch, as_stream(s)));
    return HE_OK;
}
}  // namespace
}  // extern "C++"

int he_bfv_add_plain_device(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count, uint64_t* ct,
                            const uint64_t* plaintexts, size_t batch, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    const int status = check_level(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    return plaintext_translate(ctx, tool, poly_count, ct, plaintexts, false, batch, s);
}
int he_bfv_sub_plain_device(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count, uint64_t* ct,
                            const uint64_t* plaintexts, size_t batch, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    const int status = check_level(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    return plaintext_translate(ctx, tool, poly_count, ct, plaintexts, true, batch, s);
}

extern "C++" {
namespace {
// Bfv.innerProduct(ciphertexts:plaintexts:) (Bfv/Bfv.swift:476-505) with the nil-plaintext mask resident on the device
template <typename W>
int inner_product_plain(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count, const W* cts, const W* pts,
                        const uint8_t* present_device, size_t count, size_t columns, W* out, hipStream_t stream) {
    const PolyContext* pc = ctx->impl->ciphertext(moduli_count);
    const uint64_t max_lazy = pc->max_lazy_product_accumulation_count(moduli_count);
    // the carry-counting accumulator's reduction wants sums below 2^127: at most 2^127 / (p_max - 1)^2 products
    uint64_t cadence = max_lazy;
    bool narrow_moduli = true;  // every modulus below 2^56
    for (uint32_t i = 0; i < moduli_count; ++i) {
        narrow_moduli = narrow_moduli && (pc->moduli()[i] >> 56) == 0;
        const unsigned __int128 below = pc->moduli()[i] - 1;
        if (below == 0) continue;
        // a window starts from the previous window's folded residue (< p), so cadence (p - 1)^2 + p - 1 must stay
        // below 2^127
        const unsigned __int128 limit = ((static_cast<unsigned __int128>(1) << 127) - pc->moduli()[i]) / (below * below);
        if (limit < cadence) cadence = static_cast<uint64_t>(limit);
    }
    HEAMD_HIP_TRY(heamd::launch_inner_product_plain(cts, pts, present_device, out, pc->device_context(), poly_count,
                                                    count, columns, max_lazy, cadence, narrow_moduli, stream));
    return HE_OK;
}
int check_inner_product_plain(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count, const void* cts,
                              const void* pts, size_t count, size_t columns, const void* out, bool* nothing_to_do) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    *nothing_to_do = false;
    if (count == 0) return invalid_argument("empty ciphertext vector");  // precondition, Bfv.swift:481-483
    if (columns == 0) {
        *nothing_to_do = true;
        return HE_OK;
    }
    if (cts == nullptr || pts == nullptr || out == nullptr) return invalid_argument("null operand");
    if (poly_count < 1 || poly_count > 8 || poly_count == 5 || poly_count == 7)
        return invalid_argument("poly_count must be 1, 2, 3, 4, 6 or 8");
    return HE_OK;
}
}  // namespace
}  // extern "C++"

int he_bfv_inner_product_plain_device(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count,
                                      const uint64_t* cts, const uint64_t* pts, const uint8_t* present, size_t count,
                                      size_t columns, uint64_t* out, he_stream s) {
    bool nothing = false;
    int status = check_inner_product_plain(ctx, moduli_count, poly_count, cts, pts, count, columns, out, &nothing);
    if (status != HE_OK || nothing) return status;
    hipStream_t stream = as_stream(s);
    Scratch scratch(stream);
    const uint8_t* present_device = nullptr;
    if (present != nullptr) {
        HEAMD_HIP_TRY(scratch.allocate(count * columns));
        HEAMD_HIP_TRY(hipMemcpyAsync(scratch.get(), present, count * columns, hipMemcpyHostToDevice, stream));
        HEAMD_HIP_TRY(hipStreamSynchronize(stream));  // `present` is a borrowed pageable host buffer
        present_device = static_cast<const uint8_t*>(scratch.get());
    }
    return inner_product_plain(ctx, moduli_count, poly_count, cts, pts, present_device, count, columns, out, stream);
}

int he_bfv_inner_product_plain_resident_device(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count,
                                               const uint64_t* cts, const uint64_t* pts, const uint8_t* present_device,
                                               size_t count, size_t columns, uint64_t* out, he_stream s) {
    bool nothing = false;
    int status = check_inner_product_plain(ctx, moduli_count, poly_count, cts, pts, count, columns, out, &nothing);
    if (status != HE_OK || nothing) return status;
    return inner_product_plain(ctx, moduli_count, poly_count, cts, pts, present_device, count, columns, out, as_stream(s));
}

// ------------------------------------------------------------------------------------------ packed plaintexts
extern "C++" {
namespace {
// bits(q_r) per word, rows in whole 8-byte words (degree a multiple of 64)
int packed_layout(const he_bfv_context* ctx, uint32_t moduli_count, heamd::PackedLayout& layout) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    const PolyContext* pc = ctx->impl->ciphertext(moduli_count);
    if (moduli_count > heamd::kMaxPackedRows || pc->degree() < 256) {
        heamd::set_last_error("packed plaintexts need degree >= 256 and at most 8 moduli");
        return HE_ERR_UNSUPPORTED;
    }
    layout.rows = moduli_count;
    layout.word_offset[0] = 0;
    for (uint32_t r = 0; r < moduli_count; ++r) {
        layout.width[r] = static_cast<uint32_t>(64 - __builtin_clzll(pc->moduli()[r]));
        layout.word_offset[r + 1] = layout.word_offset[r] + static_cast<uint32_t>(pc->degree() / 64 * layout.width[r]);
    }
    return HE_OK;
}
}  // namespace
}  // extern "C++"

size_t he_bfv_packed_plaintext_words(const he_bfv_context* ctx, uint32_t moduli_count) {
    heamd::PackedLayout layout{};
    if (ctx == nullptr || packed_layout(ctx, moduli_count, layout) != HE_OK) return 0;
    return layout.word_offset[layout.rows];
}

int he_bfv_pack_plaintexts_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* plaintexts_eval,
                                  size_t count, uint64_t* packed, he_stream s) {
    heamd::PackedLayout layout{};
    int status = packed_layout(ctx, moduli_count, layout);
    if (status != HE_OK) return status;
    if (count == 0) return HE_OK;
    if (plaintexts_eval == nullptr || packed == nullptr) return invalid_argument("null operand");
    const PolyContext* pc = ctx->impl->ciphertext(moduli_count);
    HEAMD_HIP_TRY(heamd::launch_pack_rows(plaintexts_eval, packed, layout, pc->device_context().log_degree, count,
                                          as_stream(s)));
    return HE_OK;
}

int he_bfv_inner_product_plain_packed_device(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count,
                                             const uint64_t* cts, const uint64_t* packed_pts, const uint8_t* present_device,
                                             size_t count, size_t columns, uint64_t* out, he_stream s) {
    bool nothing = false;
    int status = check_inner_product_plain(ctx, moduli_count, poly_count, cts, packed_pts, count, columns, out, &nothing);
    if (status != HE_OK || nothing) return status;
    if (poly_count > 3) return invalid_argument("packed plaintexts take poly_count 1..3");
    heamd::PackedLayout layout{};
    status = packed_layout(ctx, moduli_count, layout);
    if (status != HE_OK) return status;
    const PolyContext* pc = ctx->impl->ciphertext(moduli_count);
    uint64_t cadence = pc->max_lazy_product_accumulation_count(moduli_count);
    bool narrow_moduli = true;
    for (uint32_t i = 0; i < moduli_count; ++i) {  // as in inner_product_plain above
        narrow_moduli = narrow_moduli && (pc->moduli()[i] >> 56) == 0;
        const unsigned __int128 below = pc->moduli()[i] - 1;
        if (below == 0) continue;
        const unsigned __int128 limit = ((static_cast<unsigned __int128>(1) << 127) - pc->moduli()[i]) / (below * below);
        if (limit < cadence) cadence = static_cast<uint64_t>(limit);
    }
    if (cadence == 0) {
        heamd::set_last_error("moduli too wide for the packed inner product");
        return HE_ERR_UNSUPPORTED;
    }
    HEAMD_HIP_TRY(heamd::launch_inner_product_plain_packed(cts, packed_pts, layout, present_device, out,
                                                           pc->device_context(), poly_count, count, columns, cadence,
                                                           narrow_moduli, as_stream(s)));
    return HE_OK;
}

// ------------------------------------------------------------------------------------------ inner product ct . ct
size_t he_bfv_inner_product_workspace_bytes(const he_bfv_context* ctx, uint32_t moduli_count, size_t count) {
    if (ctx == nullptr || !ctx->impl->valid(moduli_count)) return 0;
    return (count * 4 + 3) * qbsk_poly_words(*ctx->impl, moduli_count) * sizeof(uint64_t);
}

extern "C++" {
namespace {
// Bfv.innerProduct(_: [CanonicalCiphertext], _: [CanonicalCiphertext]) (Bfv/Bfv.swift:315-361)
template <typename W>
int inner_product_pipeline(const he_bfv_context* ctx, const RnsToolLevel* tool, uint32_t L, const W* lhs, const W* rhs,
                           size_t count, W* out, void* workspace, size_t workspace_bytes, hipStream_t stream) {
    const size_t n = ctx->impl->degree(), ext = qbsk_poly_words(*ctx->impl, L), rows = 2 * L + 1;
    Scratch scratch(stream);
    uint64_t* raw = nullptr;
    int status = resolve_workspace(workspace, workspace_bytes, (count * 4 + 3) * ext * sizeof(W), scratch, &raw);
    if (status != HE_OK) return status;
    W* lifted = reinterpret_cast<W*>(raw);   // [count][4][2L+1][N]
    W* sum = lifted + count * 4 * ext;       // [3][2L+1][N]
    HEAMD_HIP_TRY(lift_pairs_to_eval(*tool, L, n, ext, lhs, rhs, lifted, count, stream));
    const DeviceContext qbsk = tool->qbsk->device_context();
    // maxProductCount = maxLazyProductAccumulationCount() / 2 because poly1 takes two products per pair (Bfv.swift:339)
    const uint64_t max_lazy = tool->qbsk->max_lazy_product_accumulation_count(static_cast<uint32_t>(rows)) / 2;
    HEAMD_HIP_TRY(heamd::launch_tensor_accumulate(static_cast<const W*>(lifted), sum, qbsk, count, max_lazy ? max_lazy : 1,
                                                  stream));
    return drop_extended_base(*tool, sum, out, 3, stream);
}

// `items` inner products that share the left vector: lhs [count][2][L][N], rhs [items][count][2][L][N] -> out
// [items][3][L][N].  Every stage is one launch over all items (the left vector is lifted and transformed once).
template <typename W>
int inner_product_shared_pipeline(const he_bfv_context* ctx, const RnsToolLevel* tool, uint32_t L, const W* lhs,
                                  const W* rhs, size_t count, size_t items, W* out, hipStream_t stream) {
    const size_t ext = qbsk_poly_words(*ctx->impl, L), rows = 2 * L + 1;
    Scratch scratch(stream);
    HEAMD_HIP_TRY(scratch.allocate((count * 2 + items * count * 2 + items * 3) * ext * sizeof(W)));
    W* lifted_l = static_cast<W*>(scratch.get());      // [count][2][2L+1][N]
    W* lifted_r = lifted_l + count * 2 * ext;           // [items][count][2][2L+1][N]
    W* sum = lifted_r + items * count * 2 * ext;        // [items][3][2L+1][N]
    HEAMD_HIP_TRY(heamd::launch_lift_q_to_qbsk(lhs, lifted_l, tool->device, count * 2, stream));
    HEAMD_HIP_TRY(heamd::launch_lift_q_to_qbsk(rhs, lifted_r, tool->device, items * count * 2, stream));
    const DeviceContext qbsk = tool->qbsk->device_context();
    // lifted_l and lifted_r are adjacent: one transform launch over both
    HEAMD_HIP_TRY(ntt_records(false, lifted_l, *tool->qbsk, qbsk, static_cast<uint32_t>(rows), (1 + items) * count * 2, stream));
    if constexpr (std::is_same<W, uint64_t>::value) {
        // the carry-counting sums where they exist (rns_kernels.hip tensor_accumulate_shared_sums_kernel)
        const uint64_t cadence = heamd::tensor_sums_cadence(tool->qbsk->moduli().data(), static_cast<uint32_t>(rows));
        const hipError_t sums = heamd::launch_tensor_accumulate_shared_sums(lifted_l, lifted_r, nullptr, 0, sum, qbsk, count, items,
                                                                           cadence, stream);
        if (sums != hipErrorNotSupported) {
            HEAMD_HIP_TRY(sums);
            return drop_extended_base(*tool, sum, out, items * 3, stream);
        }
        (void)hipGetLastError();
    }
    const uint64_t max_lazy = tool->qbsk->max_lazy_product_accumulation_count(static_cast<uint32_t>(rows)) / 2;
    HEAMD_HIP_TRY(heamd::launch_tensor_accumulate_shared(static_cast<const W*>(lifted_l), static_cast<const W*>(lifted_r), sum,
                                                         qbsk, count, items, max_lazy ? max_lazy : 1, stream));
    return drop_extended_base(*tool, sum, out, items * 3, stream);
}
}  // namespace
}  // extern "C++"

// inner_product_shared_pipeline for a right-hand side that arrives in EVAL form over Q (the dim-0 inner products of a PIR
// response as their kernel leaves them, PirUtil.swift:428-437): the Q rows of its lifted records would be those words themselves
// -- the forward transform of the inverse transform of a canonical row is the row -- so the sums read them where they lie, and
// only the Bsk rows go through the inverse transform (out of place, into a Coeff copy), the lift and the forward transform.
// rhs_eval is not written.  kInnerProductEvalUnavailable: nothing launched, the caller takes rhs to Coeff
// (convertToCoeffFormat, PirUtil.swift:438) and calls he_bfv_inner_product_shared_device.
extern "C++" {
namespace heamd {
int bfv_inner_product_shared_eval_rhs(const he_bfv_context* ctx, uint32_t L, const uint64_t* lhs, const uint64_t* rhs_eval,
                                      size_t count, size_t items, uint64_t* out, hipStream_t stream) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level(ctx, L, &tool);
    if (status != HE_OK) return status;
    const PolyContext* q_ctx = ctx->impl->ciphertext(L);
    const DeviceContext qbsk = tool->qbsk->device_context();
    const uint32_t rows = 2 * L + 1;
    const size_t n = ctx->impl->degree(), ext = qbsk_poly_words(*ctx->impl, L), polys = items * count * 2;
    const uint64_t cadence = heamd::tensor_sums_cadence(tool->qbsk->moduli().data(), rows);
    if (q_ctx == nullptr || polys == 0 || cadence == 0 || qbsk.log_degree < 12 || qbsk.log_degree > 13 ||
        polys * rows > (size_t(1) << 30))
        return kInnerProductEvalUnavailable;
    const DeviceContext q_device = q_ctx->device_context();
    if (q_device.approx_ok == 0) return kInnerProductEvalUnavailable;  // (no out-of-place transform for such moduli)
    Scratch scratch(stream), coeff_mem(stream);
    HEAMD_HIP_TRY(scratch.allocate((count * 2 + polys + items * 3) * ext * sizeof(uint64_t)));
    HEAMD_HIP_TRY(coeff_mem.allocate(polys * size_t(L) * n * sizeof(uint64_t)));
    uint64_t* lifted_l = static_cast<uint64_t*>(scratch.get());  // [count][2][2L+1][N]
    uint64_t* lifted_r = lifted_l + count * 2 * ext;              // [items][count][2][2L+1][N]: only the Bsk rows are used
    uint64_t* sum = lifted_r + polys * ext;                       // [items][3][2L+1][N]
    uint64_t* coeff = static_cast<uint64_t*>(coeff_mem.get());    // [items][count][2][L][N]
    HEAMD_HIP_TRY(heamd::launch_lift_q_to_qbsk(lhs, lifted_l, tool->device, count * 2, stream));
    HEAMD_HIP_TRY(ntt_records(false, lifted_l, *tool->qbsk, qbsk, rows, count * 2, stream));
    HEAMD_HIP_TRY(heamd::launch_ntt_inverse_out_of_place(rhs_eval, coeff, q_device, 0, L, polys * L, stream));
    HEAMD_HIP_TRY(heamd::launch_lift_q_to_qbsk_strided(static_cast<const uint64_t*>(coeff), lifted_r, tool->device, polys, 1,
                                                       size_t(L) * n, ext, 0, stream, false));
    HEAMD_HIP_TRY(heamd::launch_ntt_record_band(false, lifted_r, qbsk, rows, L, rows - L, polys, stream));
    HEAMD_HIP_TRY(heamd::launch_tensor_accumulate_shared_sums(lifted_l, lifted_r, rhs_eval, L, sum, qbsk, count, items, cadence,
                                                              stream));
    return drop_extended_base(*tool, sum, out, items * 3, stream);
}
}  // namespace heamd
}  // extern "C++"

int he_bfv_inner_product_shared_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* lhs,
                                       const uint64_t* rhs, size_t count, size_t items, uint64_t* out, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    if (count == 0) return invalid_argument("empty ciphertext vector");
    if (items == 0) return HE_OK;
    if (lhs == nullptr || rhs == nullptr || out == nullptr) return invalid_argument("null ciphertext");
    return inner_product_shared_pipeline(ctx, tool, moduli_count, lhs, rhs, count, items, out, as_stream(s));
}

int he_bfv_inner_product_device(const he_bfv_context* ctx, uint32_t moduli_count, const uint64_t* lhs,
                                const uint64_t* rhs, size_t count, uint64_t* out, void* workspace,
                                size_t workspace_bytes, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    if (count == 0) return invalid_argument("empty ciphertext vector");
    if (lhs == nullptr || rhs == nullptr || out == nullptr) return invalid_argument("null ciphertext");
    return inner_product_pipeline(ctx, tool, moduli_count, lhs, rhs, count, out, workspace, workspace_bytes, as_stream(s));
}

// ------------------------------------------------------------------------------------------ Bfv<UInt32> on 4-byte slabs
// The remaining entry points of this file on packed [UInt32] slabs: the same kernels instantiated on 4-byte words
// (widened in registers), the transforms through word32_kernels.hip.  The context must come from
// he_bfv_context_create_u32.
extern "C++" {
namespace {
int check_level_u32(const he_bfv_context* ctx, uint32_t moduli_count, const RnsToolLevel** tool) {
    const int status = check_level(ctx, moduli_count, tool);
    if (status != HE_OK) return status;
    if (ctx->impl->word_bits() != 32) return invalid_argument("4-byte slabs need a Bfv<UInt32> context");
    return HE_OK;
}
}  // namespace
}  // extern "C++"

int he_rns_lift_q_to_qbsk_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, const uint32_t* in, uint32_t* out,
                                     size_t batch, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level_u32(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    if (batch == 0) return HE_OK;
    if (in == nullptr || out == nullptr) return invalid_argument("null slab");
    HEAMD_HIP_TRY(heamd::launch_lift_q_to_qbsk(in, out, tool->device, batch, as_stream(s)));
    return HE_OK;
}
int he_rns_floor_qbsk_to_q_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, const uint32_t* in,
                                      uint32_t* out, size_t batch, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level_u32(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    if (batch == 0) return HE_OK;
    if (in == nullptr || out == nullptr) return invalid_argument("null slab");
    HEAMD_HIP_TRY(heamd::launch_floor_qbsk_to_q(in, out, tool->device, batch, as_stream(s)));
    return HE_OK;
}
int he_rns_scale_and_round_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, const uint32_t* in,
                                      uint64_t scaling_factor, uint32_t* out, size_t batch, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level_u32(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    if (batch == 0) return HE_OK;
    if (in == nullptr || out == nullptr) return invalid_argument("null polynomial");
    const uint64_t t = ctx->impl->plaintext_modulus();
    if (scaling_factor >= t) return invalid_argument("scaling factor not reduced mod t");
    const uint64_t scaled = heamd::mul_mod(tool->device.inv_gamma_mod_t, scaling_factor, t);
    const heamd::U64x2 final_scale{scaled, heamd::shoup_factor(scaled, t)};
    HEAMD_HIP_TRY(heamd::launch_scale_and_round(in, out, tool->device, final_scale, batch, as_stream(s)));
    return HE_OK;
}

int he_bfv_plaintext_to_eval_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, const uint32_t* plaintext,
                                        uint32_t* out, size_t batch, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level_u32(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    if (batch == 0) return HE_OK;
    if (plaintext == nullptr || out == nullptr) return invalid_argument("null plaintext");
    hipStream_t stream = as_stream(s);
    const PolyContext* q_ctx = ctx->impl->ciphertext(moduli_count);
    const DeviceContext dc = q_ctx->device_context(moduli_count);
    HEAMD_HIP_TRY(heamd::launch_plaintext_lift(plaintext, out, dc, ctx->impl->plaintext_modulus(), batch, stream));
    HEAMD_HIP_TRY(ntt_records(false, out, *q_ctx, dc, moduli_count, batch, stream));
    return HE_OK;
}
int he_bfv_plaintext_to_coeff_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, const uint32_t* plaintext_eval,
                                         uint32_t* out, size_t batch, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level_u32(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    if (batch == 0) return HE_OK;
    if (plaintext_eval == nullptr || out == nullptr) return invalid_argument("null plaintext");
    hipStream_t stream = as_stream(s);
    const PolyContext* q_ctx = ctx->impl->ciphertext(moduli_count);
    const DeviceContext dc = q_ctx->device_context(moduli_count);
    HEAMD_HIP_TRY(heamd::launch_first_rows(plaintext_eval, out, dc, batch, stream));
    HEAMD_HIP_TRY(ntt_records(true, out, *q_ctx, dc, 1, batch, stream));  // row 0 only (Plaintext.swift:176-191)
    HEAMD_HIP_TRY(heamd::launch_plaintext_unlift(out, q_ctx->moduli()[0], ctx->impl->plaintext_modulus(),
                                                 batch * ctx->impl->degree(), stream));
    return HE_OK;
}

int he_bfv_mod_switch_down_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count,
                                      const uint32_t* in, uint32_t* out, size_t batch, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level_u32(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    if (moduli_count < 2) return HE_ERR_INVALID_POLY_CONTEXT;  // PolyRq.swift:366-368
    if (batch == 0 || poly_count == 0) return HE_OK;
    if (in == nullptr || out == nullptr) return invalid_argument("null ciphertext");
    const PolyContext* pc = ctx->impl->ciphertext(moduli_count);
    heamd::DeviceContext32 dc32{};
    status = pc->device_context32(moduli_count, dc32);
    if (status != HE_OK) return status;
    HEAMD_HIP_TRY(heamd::launch_divide_and_round_q_last32(in, out, dc32, moduli_count, batch * poly_count, as_stream(s)));
    return HE_OK;
}

int he_bfv_mul_plain_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count, uint32_t* ct,
                                const uint32_t* pt, size_t batch, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level_u32(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    if (batch == 0 || poly_count == 0) return HE_OK;
    if (ct == nullptr || pt == nullptr) return invalid_argument("null operand");
    const PolyContext* pc = ctx->impl->ciphertext(moduli_count);
    HEAMD_HIP_TRY(heamd::launch_mul_plain32(ct, pt, pc->device_context(), poly_count, batch, as_stream(s)));
    return HE_OK;
}

int he_bfv_add_plain_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count, uint32_t* ct,
                                const uint32_t* plaintexts, size_t batch, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    const int status = check_level_u32(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    return plaintext_translate(ctx, tool, poly_count, ct, plaintexts, false, batch, s);
}
int he_bfv_sub_plain_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count, uint32_t* ct,
                                const uint32_t* plaintexts, size_t batch, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    const int status = check_level_u32(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    return plaintext_translate(ctx, tool, poly_count, ct, plaintexts, true, batch, s);
}

int he_bfv_inner_product_plain_resident_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, uint32_t poly_count,
                                                   const uint32_t* cts, const uint32_t* pts,
                                                   const uint8_t* present_device, size_t count, size_t columns,
                                                   uint32_t* out, he_stream s) {
    if (ctx != nullptr && ctx->impl->word_bits() != 32) return invalid_argument("4-byte slabs need a Bfv<UInt32> context");
    bool nothing = false;
    int status = check_inner_product_plain(ctx, moduli_count, poly_count, cts, pts, count, columns, out, &nothing);
    if (status != HE_OK || nothing) return status;
    return inner_product_plain(ctx, moduli_count, poly_count, cts, pts, present_device, count, columns, out, as_stream(s));
}

int he_bfv_inner_product_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, const uint32_t* lhs,
                                    const uint32_t* rhs, size_t count, uint32_t* out, void* workspace,
                                    size_t workspace_bytes, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level_u32(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    if (count == 0) return invalid_argument("empty ciphertext vector");
    if (lhs == nullptr || rhs == nullptr || out == nullptr) return invalid_argument("null ciphertext");
    return inner_product_pipeline(ctx, tool, moduli_count, lhs, rhs, count, out, workspace, workspace_bytes, as_stream(s));
}

int he_bfv_inner_product_shared_device_u32(const he_bfv_context* ctx, uint32_t moduli_count, const uint32_t* lhs,
                                           const uint32_t* rhs, size_t count, size_t items, uint32_t* out, he_stream s) {
    const RnsToolLevel* tool = nullptr;
    int status = check_level_u32(ctx, moduli_count, &tool);
    if (status != HE_OK) return status;
    if (count == 0) return invalid_argument("empty ciphertext vector");
    if (items == 0) return HE_OK;
    if (lhs == nullptr || rhs == nullptr || out == nullptr) return invalid_argument("null ciphertext");
    return inner_product_shared_pipeline(ctx, tool, moduli_count, lhs, rhs, count, items, out, as_stream(s));
}

}  // extern "C"
