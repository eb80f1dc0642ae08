// device_group.cpp -- the path's multi-GPU split behind the C ABI: one process, one he_bfv_context and one
// stream per member device, the units of a call (polynomials, database columns) split over the members by the rule of
// he_shard_bounds, nothing on the data path crossing devices but the replicated query on the way in and the finished
// shards on the way out (peer copies on the members' streams, joined into the caller's stream by events).
//
// The reference runs the same partition over tasks of one process: the columns of a chunk in groups (PirUtil.swift:424-445),
// the polynomials of a batch one task each (Bfv.swift:266-287) -- every unit independent, contexts shared.  Here contexts
// are replicated per device (a few MiB of tables) and the database lives sharded by column where it was uploaded.
#include <memory>
#include <mutex>
#include <vector>

#include "api_internal.hpp"

using heamd::as_stream;
using heamd::invalid_argument;
using heamd::Scratch;

struct he_device_group {
    struct Member {
        int device = 0;
        he_bfv_context* ctx = nullptr;
        hipStream_t stream = nullptr;
        hipEvent_t done = nullptr;  // recorded on `stream` when the member's share of a call is enqueued
        bool remote = false;        // its results reach the home device by a peer copy
    };
    std::vector<Member> members;
    hipEvent_t home_ready = nullptr;  // on member 0's device: the caller's stream has produced the call's inputs
    std::mutex mutex;                 // the events are the group's: one call enqueues at a time
};

namespace {

// the calling thread's device, restored on scope exit
class DeviceGuard {
  public:
    DeviceGuard() { ok_ = hipGetDevice(&saved_) == hipSuccess; }
    ~DeviceGuard() {
        if (ok_) (void)hipSetDevice(saved_);
    }

  private:
    int saved_ = 0;
    bool ok_ = false;
};

void bounds(size_t total, size_t members, size_t member, size_t& begin, size_t& end) {
    const size_t base = total / members, extra = total % members;
    begin = member * base + (member < extra ? member : extra);
    end = begin + base + (member < extra ? 1 : 0);
}

void destroy_members(he_device_group* group) {
    for (he_device_group::Member& m : group->members) {
        if (hipSetDevice(m.device) != hipSuccess) continue;
        if (m.stream != nullptr) (void)hipStreamSynchronize(m.stream);
        if (m.done != nullptr) (void)hipEventDestroy(m.done);
        if (m.stream != nullptr) {
            heamd::scratch_forget_stream(m.stream);
            (void)hipStreamDestroy(m.stream);
        }
        he_bfv_context_destroy(m.ctx);
    }
    if (!group->members.empty() && group->home_ready != nullptr && hipSetDevice(group->members[0].device) == hipSuccess)
        (void)hipEventDestroy(group->home_ready);
}

// device-to-device bytes on `stream` (which belongs to the device that is current)
hipError_t copy_between(void* dst, int dst_device, const void* src, int src_device, size_t bytes, hipStream_t stream) {
    if (dst_device == src_device) return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, stream);
    return hipMemcpyPeerAsync(dst, dst_device, src, src_device, bytes, stream);
}

}  // namespace

extern "C" {

int he_shard_bounds(size_t total, uint32_t members, uint32_t member, size_t* out_begin, size_t* out_end) {
    if (out_begin == nullptr || out_end == nullptr) return invalid_argument("null out");
    if (members == 0 || member >= members) return invalid_argument("member out of range");
    bounds(total, members, member, *out_begin, *out_end);
    return HE_OK;
}

int he_device_group_create(const int* devices, uint32_t device_count, uint32_t flags, uint32_t degree,
                           uint64_t plaintext_modulus, const uint64_t* coefficient_moduli, uint32_t moduli_count,
                           he_device_group** out) {
    if (out == nullptr) return invalid_argument("null out");
    *out = nullptr;
    if (devices == nullptr || device_count == 0) return invalid_argument("empty device list");
    if ((flags & ~uint32_t(HE_GROUP_STAGE_ALL)) != 0) return invalid_argument("unknown group flags");
    int available = 0;
    HEAMD_HIP_TRY(hipGetDeviceCount(&available));
    for (uint32_t i = 0; i < device_count; ++i)
        if (devices[i] < 0 || devices[i] >= available) return invalid_argument("no such device");
    DeviceGuard guard;
    std::unique_ptr<he_device_group> group(new he_device_group());
    group->members.resize(device_count);
    int status = HE_OK;
    for (uint32_t i = 0; i < device_count && status == HE_OK; ++i) {
        he_device_group::Member& m = group->members[i];
        m.device = devices[i];
        m.remote = i != 0 && (m.device != devices[0] || (flags & HE_GROUP_STAGE_ALL) != 0);
        hipError_t e = hipSetDevice(m.device);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&m.stream, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&m.done, hipEventDisableTiming);
        if (e == hipSuccess && i == 0) e = hipEventCreateWithFlags(&group->home_ready, hipEventDisableTiming);
        if (e == hipSuccess && m.device != devices[0]) {
            // direct peer copies where the fabric allows them (xGMI); without access the runtime stages the copy
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, m.device, devices[0]) == hipSuccess && can != 0) {
                const hipError_t enabled = hipDeviceEnablePeerAccess(devices[0], 0);
                if (enabled != hipSuccess && enabled != hipErrorPeerAccessAlreadyEnabled) (void)hipGetLastError();
            }
            (void)hipGetLastError();
        }
        if (e != hipSuccess) {
            status = heamd::device_failure(e, "device group member");
            break;
        }
        status = he_bfv_context_create(degree, plaintext_modulus, coefficient_moduli, moduli_count, &m.ctx);
    }
    if (status != HE_OK) {
        destroy_members(group.get());
        return status;
    }
    *out = group.release();
    return HE_OK;
}

void he_device_group_destroy(he_device_group* group) {
    if (group == nullptr) return;
    heamd::RelaxedCapture relaxed;
    DeviceGuard guard;
    destroy_members(group);
    delete group;
}

uint32_t he_device_group_size(const he_device_group* group) {
    return group == nullptr ? 0 : static_cast<uint32_t>(group->members.size());
}
int he_device_group_device(const he_device_group* group, uint32_t member, int* out_device) {
    if (group == nullptr || out_device == nullptr) return invalid_argument("null argument");
    if (member >= group->members.size()) return invalid_argument("member out of range");
    *out_device = group->members[member].device;
    return HE_OK;
}
const he_bfv_context* he_device_group_context(const he_device_group* group, uint32_t member) {
    return (group == nullptr || member >= group->members.size()) ? nullptr : group->members[member].ctx;
}
he_stream he_device_group_stream(const he_device_group* group, uint32_t member) {
    return (group == nullptr || member >= group->members.size()) ? nullptr : group->members[member].stream;
}

int he_device_group_synchronize(he_device_group* group) {
    if (group == nullptr) return invalid_argument("null group");
    DeviceGuard guard;
    for (he_device_group::Member& m : group->members) {
        HEAMD_HIP_TRY(hipSetDevice(m.device));
        HEAMD_HIP_TRY(hipStreamSynchronize(m.stream));
    }
    return HE_OK;
}

// PolyContext.forwardNtt / inverseNtt over a batch whose polynomials live sharded: member m transforms its resident shard
// slab_shards[m] = [he_shard_bounds(batch, size, m)][L][N] on its own stream; nothing moves.
static int ntt_group(he_device_group* group, uint32_t moduli_count, uint64_t* const* slab_shards, size_t batch, bool inverse) {
    if (group == nullptr) return invalid_argument("null group");
    if (batch == 0) return HE_OK;
    if (slab_shards == nullptr) return invalid_argument("null shards");
    DeviceGuard guard;
    std::lock_guard<std::mutex> lock(group->mutex);
    const size_t size = group->members.size();
    for (size_t i = 0; i < size; ++i) {
        he_device_group::Member& m = group->members[i];
        size_t begin = 0, end = 0;
        bounds(batch, size, i, begin, end);
        if (end == begin) continue;
        if (slab_shards[i] == nullptr) return invalid_argument("null shard");
        const he_poly_context* pc = he_bfv_ciphertext_context(m.ctx, moduli_count);
        if (pc == nullptr) return invalid_argument("moduli_count out of range");
        HEAMD_HIP_TRY(hipSetDevice(m.device));
        const int status = inverse ? he_ntt_inverse_device(pc, slab_shards[i], end - begin, m.stream)
                                   : he_ntt_forward_device(pc, slab_shards[i], end - begin, m.stream);
        if (status != HE_OK) return status;
    }
    return HE_OK;
}
int he_ntt_forward_group(he_device_group* group, uint32_t moduli_count, uint64_t* const* slab_shards, size_t batch) {
    return ntt_group(group, moduli_count, slab_shards, batch, false);
}
int he_ntt_inverse_group(he_device_group* group, uint32_t moduli_count, uint64_t* const* slab_shards, size_t batch) {
    return ntt_group(group, moduli_count, slab_shards, batch, true);
}

}  // extern "C"
namespace {
// leave_in_eval: the members skip the inverse transform of their columns (PirUtil.swift:438); the home device's remaining
// dimensions take them as they are (pir_api.cpp remaining_dimensions, results_in_eval)
int dim0_columns_group(he_device_group* group, const uint64_t* dim0_query_eval, size_t d0,
                       const uint64_t* const* database_shards, const uint8_t* const* present_shards, size_t columns,
                       uint64_t* out, he_stream home_stream, bool leave_in_eval) {
    if (group == nullptr) return invalid_argument("null group");
    if (columns == 0) return HE_OK;
    if (dim0_query_eval == nullptr || database_shards == nullptr || out == nullptr) return invalid_argument("null operand");
    DeviceGuard guard;
    std::lock_guard<std::mutex> lock(group->mutex);
    const size_t size = group->members.size();
    const he_device_group::Member& home = group->members[0];
    const uint32_t L = he_bfv_ciphertext_moduli_count(home.ctx);
    const size_t n = he_poly_context_degree(he_bfv_ciphertext_context(home.ctx, L));
    const size_t ct_words = 2 * size_t(L) * n, query_bytes = d0 * ct_words * sizeof(uint64_t);
    hipStream_t caller = as_stream(home_stream);
    HEAMD_HIP_TRY(hipSetDevice(home.device));
    HEAMD_HIP_TRY(hipEventRecord(group->home_ready, caller));
    int status = HE_OK;
    std::vector<bool> enqueued(size, false);
    for (size_t i = 0; i < size && status == HE_OK; ++i) {
        he_device_group::Member& m = group->members[i];
        size_t begin = 0, end = 0;
        bounds(columns, size, i, begin, end);
        const size_t mine = end - begin;
        if (mine == 0) continue;
        if (database_shards[i] == nullptr) {
            status = invalid_argument("null database shard");
            break;
        }
        hipError_t e = hipSetDevice(m.device);
        if (e == hipSuccess) e = hipStreamWaitEvent(m.stream, group->home_ready, 0);
        // (the scratch below is released on the member's stream, with the member's device current)
        Scratch query_copy(m.stream), shard_out(m.stream);
        const uint64_t* query = dim0_query_eval;
        uint64_t* result = out + begin * ct_words;
        if (m.remote && e == hipSuccess) {
            // the query is replicated (512 MiB at BASELINE configs[4]'s d0 = 1024: 3 ms over one xGMI link, against the
            // member's 34 GB pass), the member's columns come back as one copy
            e = query_copy.allocate(query_bytes);
            if (e == hipSuccess)
                e = copy_between(query_copy.get(), m.device, dim0_query_eval, home.device, query_bytes, m.stream);
            if (e == hipSuccess) e = shard_out.allocate(mine * ct_words * sizeof(uint64_t));
            query = static_cast<const uint64_t*>(query_copy.get());
            result = static_cast<uint64_t*>(shard_out.get());
        }
        if (e != hipSuccess) {
            status = heamd::device_failure(e, "device group: member setup");
            break;
        }
        status = leave_in_eval
                     ? heamd::pir_dim0_columns_eval(m.ctx, query, d0, database_shards[i],
                                                    present_shards != nullptr ? present_shards[i] : nullptr, mine, result, m.stream)
                     : he_pir_dim0_columns_device(m.ctx, query, d0, database_shards[i],
                                                  present_shards != nullptr ? present_shards[i] : nullptr, mine, result, m.stream);
        if (status != HE_OK) break;
        if (m.remote)
            e = copy_between(out + begin * ct_words, home.device, result, m.device, mine * ct_words * sizeof(uint64_t), m.stream);
        if (e == hipSuccess) e = hipEventRecord(m.done, m.stream);
        if (e != hipSuccess) {
            status = heamd::device_failure(e, "device group: gather");
            break;
        }
        enqueued[i] = true;
    }
    // the caller's stream continues once every member that got work has delivered (also after a failure: it never runs
    // ahead of copies already enqueued into `out`)
    const hipError_t back = hipSetDevice(home.device);
    for (size_t i = 0; i < size && back == hipSuccess; ++i)
        if (enqueued[i]) (void)hipStreamWaitEvent(caller, group->members[i].done, 0);
    if (status != HE_OK) return status;
    HEAMD_HIP_TRY(back);
    return HE_OK;
}
// Every member's share of the column range is a whole number of chunks: each member then answers ITS chunks from start to finish
// (he_pir_compute_response_device on its own stream: dim-0 pass, remaining dimensions, modSwitchDownToSingle) and only the
// finished single-modulus responses -- 2 N words per chunk -- come back.  Nothing is left for the home device to do alone:
// with the columns split inside chunks the remaining dimensions of ALL chunks run there after the gather (0.74 ms for 8 chunks
// of 256 x 64 against a dim-0 pass of 5.6 ms / members).  The queries and the key are replicated to remote members.
int response_by_whole_chunks(he_device_group* group, const uint32_t* dimensions, uint32_t dimension_count, size_t columns_per_chunk,
                             const uint64_t* dim0_query_eval, const uint64_t* remaining_query, size_t remaining_query_count,
                             const uint64_t* const* database_shards, const uint8_t* const* present_shards, size_t chunk_count,
                             const uint64_t* relinearization_key, uint64_t* out, he_stream home_stream) {
    if (dim0_query_eval == nullptr || database_shards == nullptr || out == nullptr) return invalid_argument("null operand");
    DeviceGuard guard;
    std::lock_guard<std::mutex> lock(group->mutex);
    const size_t size = group->members.size();
    const he_device_group::Member& home = group->members[0];
    const uint32_t L = he_bfv_ciphertext_moduli_count(home.ctx);
    const size_t n = he_poly_context_degree(he_bfv_ciphertext_context(home.ctx, L));
    const size_t ct_bytes = 2 * size_t(L) * n * sizeof(uint64_t), out_words = 2 * n;
    const size_t dim0_bytes = dimensions[0] * ct_bytes, remaining_bytes = remaining_query_count * ct_bytes;
    const size_t key_bytes = size_t(L) * 2 * (L + 1) * n * sizeof(uint64_t);  // [L][2][L+1][N] (he_amd.h he_bfv_relinearize_device)
    hipStream_t caller = as_stream(home_stream);
    HEAMD_HIP_TRY(hipSetDevice(home.device));
    HEAMD_HIP_TRY(hipEventRecord(group->home_ready, caller));
    int status = HE_OK;
    std::vector<bool> enqueued(size, false);
    for (size_t i = 0; i < size && status == HE_OK; ++i) {
        he_device_group::Member& m = group->members[i];
        size_t begin = 0, end = 0;
        bounds(columns_per_chunk * chunk_count, size, i, begin, end);
        const size_t first = begin / columns_per_chunk, mine = (end - begin) / columns_per_chunk;
        if (mine == 0) continue;
        if (database_shards[i] == nullptr) {
            status = invalid_argument("null database shard");
            break;
        }
        hipError_t e = hipSetDevice(m.device);
        if (e == hipSuccess) e = hipStreamWaitEvent(m.stream, group->home_ready, 0);
        Scratch dim0_copy(m.stream), remaining_copy(m.stream), key_copy(m.stream), shard_out(m.stream);
        const uint64_t *dim0 = dim0_query_eval, *remaining = remaining_query, *key = relinearization_key;
        uint64_t* result = out + first * out_words;
        if (m.remote && e == hipSuccess) {
            e = dim0_copy.allocate(dim0_bytes);
            if (e == hipSuccess) e = copy_between(dim0_copy.get(), m.device, dim0_query_eval, home.device, dim0_bytes, m.stream);
            dim0 = static_cast<const uint64_t*>(dim0_copy.get());
            if (e == hipSuccess && remaining_query != nullptr && remaining_bytes != 0) {
                e = remaining_copy.allocate(remaining_bytes);
                if (e == hipSuccess) e = copy_between(remaining_copy.get(), m.device, remaining_query, home.device, remaining_bytes, m.stream);
                remaining = static_cast<const uint64_t*>(remaining_copy.get());
            }
            if (e == hipSuccess && relinearization_key != nullptr) {
                e = key_copy.allocate(key_bytes);
                if (e == hipSuccess) e = copy_between(key_copy.get(), m.device, relinearization_key, home.device, key_bytes, m.stream);
                key = static_cast<const uint64_t*>(key_copy.get());
            }
            if (e == hipSuccess) e = shard_out.allocate(mine * out_words * sizeof(uint64_t));
            result = static_cast<uint64_t*>(shard_out.get());
        }
        if (e != hipSuccess) {
            status = heamd::device_failure(e, "device group: member setup");
            break;
        }
        status = he_pir_compute_response_device(m.ctx, dimensions, dimension_count, dim0, remaining, remaining_query_count,
                                                database_shards[i], present_shards != nullptr ? present_shards[i] : nullptr, mine,
                                                key, result, m.stream);
        if (status != HE_OK) break;
        if (m.remote)
            e = copy_between(out + first * out_words, home.device, result, m.device, mine * out_words * sizeof(uint64_t), m.stream);
        if (e == hipSuccess) e = hipEventRecord(m.done, m.stream);
        if (e != hipSuccess) {
            status = heamd::device_failure(e, "device group: gather");
            break;
        }
        enqueued[i] = true;
    }
    const hipError_t back = hipSetDevice(home.device);
    for (size_t i = 0; i < size && back == hipSuccess; ++i)
        if (enqueued[i]) (void)hipStreamWaitEvent(caller, group->members[i].done, 0);
    if (status != HE_OK) return status;
    HEAMD_HIP_TRY(back);
    return HE_OK;
}
}  // namespace
extern "C" {

int he_pir_dim0_columns_group(he_device_group* group, const uint64_t* dim0_query_eval, size_t d0,
                              const uint64_t* const* database_shards, const uint8_t* const* present_shards, size_t columns,
                              uint64_t* out, he_stream home_stream) {
    return dim0_columns_group(group, dim0_query_eval, d0, database_shards, present_shards, columns, out, home_stream, false);
}

int he_pir_compute_response_group(he_device_group* group, const uint32_t* dimensions, uint32_t dimension_count,
                                  const uint64_t* dim0_query_eval, const uint64_t* remaining_query,
                                  size_t remaining_query_count, const uint64_t* const* database_shards,
                                  const uint8_t* const* present_shards, size_t chunk_count,
                                  const uint64_t* relinearization_key, uint64_t* out, he_stream home_stream) {
    if (group == nullptr) return invalid_argument("null group");
    if (dimensions == nullptr || dimension_count == 0) return invalid_argument("empty dimensions");
    if (chunk_count == 0) return HE_OK;
    size_t per_chunk = 1;
    for (uint32_t i = 0; i < dimension_count; ++i) {
        if (dimensions[i] == 0) return invalid_argument("zero dimension");
        per_chunk *= dimensions[i];
    }
    // the columns of all chunks are one column range (a chunk is [columns][d0] plaintexts, the chunks are contiguous)
    const size_t d0 = dimensions[0], columns = per_chunk / d0 * chunk_count;
    const he_device_group::Member& home = group->members[0];
    const uint32_t L = he_bfv_ciphertext_moduli_count(home.ctx);
    const size_t n = he_poly_context_degree(he_bfv_ciphertext_context(home.ctx, L));
    // whole chunks per member (he_shard_bounds of the column range falls on chunk boundaries for every member): no serial tail
    const size_t members = group->members.size(), columns_per_chunk = per_chunk / d0;
    bool whole_chunks = members > 1 && dimension_count > 1;
    for (size_t i = 0; i < members && whole_chunks; ++i) {
        size_t begin = 0, end = 0;
        bounds(columns, members, i, begin, end);
        whole_chunks = begin % columns_per_chunk == 0 && end % columns_per_chunk == 0;
    }
    if (whole_chunks)
        return response_by_whole_chunks(group, dimensions, dimension_count, columns_per_chunk, dim0_query_eval, remaining_query,
                                        remaining_query_count, database_shards, present_shards, chunk_count, relinearization_key,
                                        out, home_stream);
    DeviceGuard guard;
    HEAMD_HIP_TRY(hipSetDevice(home.device));
    Scratch intermediate(as_stream(home_stream));
    HEAMD_HIP_TRY(intermediate.allocate(columns * 2 * size_t(L) * n * sizeof(uint64_t)));
    uint64_t* results = static_cast<uint64_t*>(intermediate.get());
    int status = dim0_columns_group(group, dim0_query_eval, d0, database_shards, present_shards, columns, results, home_stream,
                                    true);
    if (status == HE_OK) {
        HEAMD_HIP_TRY(hipSetDevice(home.device));
        status = heamd::pir_remaining_dimensions_chunks_eval(home.ctx, dimensions, dimension_count, chunk_count, results,
                                                             remaining_query, remaining_query_count, relinearization_key, out,
                                                             home_stream);
    }
    return status;
}

int he_pir_compute_response_chunk_group(he_device_group* group, const uint32_t* dimensions, uint32_t dimension_count,
                                        const uint64_t* dim0_query_eval, const uint64_t* remaining_query,
                                        size_t remaining_query_count, const uint64_t* const* database_shards,
                                        const uint8_t* const* present_shards, const uint64_t* relinearization_key,
                                        uint64_t* out, he_stream home_stream) {
    return he_pir_compute_response_group(group, dimensions, dimension_count, dim0_query_eval, remaining_query,
                                         remaining_query_count, database_shards, present_shards, 1, relinearization_key, out,
                                         home_stream);
}

}  // extern "C"
