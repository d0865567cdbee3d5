// naqs_poll.hpp — the ONE way a kernel of this library waits for a word another workgroup publishes.
//
// Hand-over format: value and flag are one 64-bit word, `tag << 32 | payload`, written with a relaxed
// agent-scope atomic store and read with a relaxed agent-scope atomic load — no fences, no L2 write-back.  A consumer polls
// until the tag is the one it expects.  The producers are either already running (lower workgroup index: in-order dispatch)
// or finished, so a wait is normally a few hundred nanoseconds.
//
// Every wait is BOUNDED: if the tag has not appeared after `budget` ticks of the 100 MHz constant clock (default 2 s: four
// orders of magnitude above any kernel of the library), the waiting lane records (site, workgroup, waited-for index) in the
// handle's error word — mapped host memory, first failure wins — and returns `false`; the caller then leaves WITHOUT
// writing results.  The host sees the word at its next look (every API entry, and wherever it already polls the sampler's
// published words) and returns NAQS_ERR_HIP with `naqs_last_hip_error_string()` naming the site.  The reference's only
// failure path in this loop is the sampler's MaxBatchSizeExceededError (src/naqs/network/nade.py:39-40, 710-712 ->
// src/optimizer/energy.py:939-946); a hang has no counterpart there, which is why it must become an error here.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace naqs {

// where a wait sits (the error word names it)
enum PollSite : uint32_t {
    POLL_NONE = 0,
    POLL_LOGPSI_SPLIT = 1,     // phase kernels, column split: the consumer half waits for the producer half's partial rows
    POLL_SAMPLE_LOOKBACK = 2,  // sampler, one level per launch: decoupled look-back over the preceding workgroups' child counts
    POLL_SAMPLE_LOOKBACK_MULTI = 3,   // sampler, several levels per launch: the same look-back at the launch's last level
    POLL_SEED_SUMS = 4,        // vmc_seed_delta_kernel: workgroups > 0 wait for workgroup 0's sums
    POLL_PACK_BOUNDS = 5,      // weight re-pack: the scale chain waits for the 64 per-workgroup maxima
    POLL_N_SITES
};

// per handle, in device memory (read on the slow path only); `err` points into mapped host memory
struct PollCtl {
    unsigned long long budget;      // ticks of wall_clock64() (100 MHz)
    uint32_t drop_site;             // debugging (NAQS_DEBUG_DROP_STORE=<site>): the producers of this site skip one store
    uint32_t pad;
    unsigned long long *err;        // 0 = fine; else site << 56 | (workgroup & 0xFFFFFF) << 32 | waited-for index
};

#if defined(__HIPCC__)
// true when this producer should skip its store (debug knob; never set in production)
__device__ __forceinline__ bool poll_drop(const PollCtl *ctl, uint32_t site) {
    return ctl != nullptr && ctl->drop_site == site;
}

__device__ __forceinline__ void poll_fail(const PollCtl *ctl, uint32_t site, uint32_t index) {
    if (ctl == nullptr || ctl->err == nullptr) return;
    const unsigned long long code = ((unsigned long long)site << 56) | ((unsigned long long)(blockIdx.x & 0xFFFFFFu) << 32) | index;
    unsigned long long expected = 0ull;
    __hip_atomic_compare_exchange_strong(ctl->err, &expected, code, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Poll `src` until its upper half equals `tag`; SLEEP = the s_sleep argument between looks.  -> true and the word in `word`,
// or false after the budget (error word set).  The fast path (tag already there) is one load and one compare.
template <int SLEEP = 2>
__device__ __forceinline__ bool poll_tagged(const unsigned long long *src, const uint32_t tag, unsigned long long &word,
                                            const PollCtl *ctl, const uint32_t site, const uint32_t index) {
    word = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if ((uint32_t)(word >> 32) == tag) return true;
    const unsigned long long t0 = wall_clock64();
    const unsigned long long budget = ctl != nullptr ? ctl->budget : 200000000ull;
    for (uint32_t n = 1;; ++n) {
        __builtin_amdgcn_s_sleep(SLEEP);
        word = __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((uint32_t)(word >> 32) == tag) return true;
        if ((n & 63u) == 0u && wall_clock64() - t0 > budget) {
            poll_fail(ctl, site, index);
            return false;
        }
    }
}
#endif

// host side (naqs_hip.hip).  Round 6: a control block and an error word PER HANDLE that launches waiting kernels (every network
// handle), not per device — two runs share a GPU in the farm (experiments.run --per-gpu 2), and with one word per device the
// run whose kernel gave up could find its error taken by its neighbour's next call: the neighbour failed for nothing and the
// victim went on with results that were never written.  A handle looks at ITS word; naqs_device_check(device) looks at all
// of the device's.
struct PollHandle {
    PollCtl *dev = nullptr;                    // device copy of the control block, or nullptr (waits stay bounded by the default, unreported)
    unsigned long long *err_host = nullptr;    // mapped host word
    int device = -1;
};
int poll_handle_create(int device, PollHandle *out);     // NAQS_OK even when nothing could be allocated (dev == nullptr then)
void poll_handle_destroy(PollHandle *h);
int poll_check(const PollHandle &h);           // NAQS_OK, or NAQS_ERR_HIP after recording what the word said (and clearing it)
int poll_check_device(int device);             // the same over every live handle of the device

}  // namespace naqs
