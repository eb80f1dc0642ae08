// side_lane.hpp -- second streams a he_bfv_context owns, for work of ONE call that can run beside its neighbour.
//
// A call leases a lane, forks it off the caller's stream with an event, enqueues on both, and joins the lane back
// before it returns; as seen from the caller the call is still enqueue-only on its own stream.  Two uses:
//   * ct x ct (bfv_api.cpp mul_rows_fused): the row band that reads the ciphertexts beside the lift, a part's floor beside
//     the next part's Bsk band;
//   * the PIR server (pir_api.cpp), as an experiment switch only: the remaining dimensions of one piece of the chunks beside
//     the next piece's dim-0 pass over the database, the way the reference runs them as concurrent tasks
//     (PirUtil.swift:538-563) -- measured slower than the single-stream order on this part (pir_api.cpp).
// Lanes are FIFO, so a lane shared by two caller streams would queue one caller's work behind the other's backlog: the
// pool hands a caller stream the lane it used last, a fresh one to a stream it has not seen (up to kMaxLanes per context),
// and only then the least recently used lane.  A lane is leased for the host-side enqueue only; a call that finds every
// lane leased (or is being captured into a graph) runs its single-stream order instead.  A call running ON a lane may lease
// another one (the PIR tail's ct x ct does).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <memory>
#include <mutex>
#include <vector>

namespace heamd {

struct SideLane {
    static constexpr int kStages = 4;
    hipStream_t stream = nullptr;
    hipEvent_t forked = nullptr, joined = nullptr;
    hipEvent_t stage[kStages] = {};  // the caller's stream has reached a point the lane's next piece of work depends on
    // pool bookkeeping (under the pool's mutex)
    bool leased = false;
    hipStream_t last_caller = nullptr;
    bool used = false;
    uint64_t last_use = 0;

    SideLane() = default;
    SideLane(const SideLane&) = delete;
    SideLane& operator=(const SideLane&) = delete;
    ~SideLane() {
        if (forked != nullptr) (void)hipEventDestroy(forked);
        if (joined != nullptr) (void)hipEventDestroy(joined);
        for (hipEvent_t e : stage)
            if (e != nullptr) (void)hipEventDestroy(e);
        if (stream != nullptr) (void)hipStreamDestroy(stream);
    }
    // Stream and events.  Another thread may be capturing in hipStreamCaptureModeGlobal, where creation calls are not
    // allowed: this thread's capture mode is relaxed around them.
    hipError_t create() {
        hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
        const bool exchanged = hipThreadExchangeStreamCaptureMode(&mode) == hipSuccess;
        if (!exchanged) (void)hipGetLastError();
        hipEvent_t* all[2 + kStages] = {&forked, &joined};
        for (int k = 0; k < kStages; ++k) all[2 + k] = &stage[k];
        // Highest priority: what runs on a lane is the smaller share of a call, beside longer kernels of the caller's stream
        // (ct x ct + relinearize 213.7 k/s against 209.9 k/s at normal priority, profiles/r06e_lane_priority.txt).
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) {
            (void)hipGetLastError();
            least = greatest = 0;
        }
        hipError_t e = hipStreamCreateWithPriority(&stream, hipStreamNonBlocking, greatest);
        for (hipEvent_t* event : all)
            if (e == hipSuccess) e = hipEventCreateWithFlags(event, hipEventDisableTiming);
        if (exchanged) (void)hipThreadExchangeStreamCaptureMode(&mode);
        return e;
    }
};

class LanePool {
  public:
    static constexpr size_t kMaxLanes = 8;
    // nullptr: no lane for this call (all leased, or none could be created)
    SideLane* lease(hipStream_t caller) {
        std::lock_guard<std::mutex> lock(mutex_);
        SideLane *same = nullptr, *fresh = nullptr, *oldest = nullptr;
        for (const std::unique_ptr<SideLane>& lane : lanes_) {
            if (lane->leased) continue;
            if (lane->used && lane->last_caller == caller) same = lane.get();
            if (!lane->used) fresh = lane.get();
            if (oldest == nullptr || lane->last_use < oldest->last_use) oldest = lane.get();
        }
        SideLane* pick = same != nullptr ? same : fresh;
        if (pick == nullptr && lanes_.size() < kMaxLanes) {
            std::unique_ptr<SideLane> made(new SideLane());
            if (made->create() == hipSuccess) {
                lanes_.push_back(std::move(made));
                pick = lanes_.back().get();
            } else {
                (void)hipGetLastError();
            }
        }
        if (pick == nullptr) pick = oldest;
        if (pick == nullptr) return nullptr;
        pick->leased = true;
        pick->used = true;
        pick->last_caller = caller;
        pick->last_use = ++clock_;
        return pick;
    }
    void give_back(SideLane* lane) {
        std::lock_guard<std::mutex> lock(mutex_);
        lane->leased = false;
    }
    size_t lane_count() {
        std::lock_guard<std::mutex> lock(mutex_);
        return lanes_.size();
    }

  private:
    std::mutex mutex_;
    std::vector<std::unique_ptr<SideLane>> lanes_;
    uint64_t clock_ = 0;
};

// A lane for the duration of one call's enqueue; empty (lane == nullptr) while `caller` is being captured into a graph --
// the legacy default stream cannot be queried and counts as captured -- or when the pool has nothing to give.
class LaneLease {
  public:
    LaneLease(LanePool& pool, hipStream_t caller) : pool_(pool) {
        hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(caller, &capture) != hipSuccess) {
            (void)hipGetLastError();
            return;
        }
        if (capture == hipStreamCaptureStatusNone) lane = pool.lease(caller);
    }
    ~LaneLease() {
        if (lane != nullptr) pool_.give_back(lane);
    }
    LaneLease(const LaneLease&) = delete;
    LaneLease& operator=(const LaneLease&) = delete;
    SideLane* lane = nullptr;

  private:
    LanePool& pool_;
};

}  // namespace heamd

struct he_bfv_context;
namespace heamd {
// bfv_api.cpp: the pool a context owns
LanePool& lane_pool(const he_bfv_context* ctx);
}  // namespace heamd
