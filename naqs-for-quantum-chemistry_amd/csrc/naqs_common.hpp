// naqs_common.hpp — host-side helpers shared by the translation units of libnaqs_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <atomic>
#include <cstdint>
#include <cstdlib>

#include "naqs_hip.h"

#define NAQS_API extern "C" __attribute__((visibility("default")))

namespace naqs {

extern thread_local hipError_t g_last_hip;   // defined in naqs_hip.hip

#define HIP_TRY(expr)                                  \
    do {                                               \
        hipError_t e__ = (expr);                       \
        if (e__ != hipSuccess) {                       \
            ::naqs::g_last_hip = e__;                  \
            return NAQS_ERR_HIP;                       \
        }                                              \
    } while (0)

constexpr int WAVE = 64;

// every kernel launch of the library goes through this (naqs_launch_count(): bench.py's launches per training step)
extern std::atomic<int64_t> g_launches;      // defined in naqs_hip.hip
#define NAQS_KLAUNCH(...)                                                  \
    do {                                                                   \
        ::naqs::g_launches.fetch_add(1, std::memory_order_relaxed);        \
        hipLaunchKernelGGL(__VA_ARGS__);                                   \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool switched = false;
    int init(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) return NAQS_ERR_NO_DEVICE;
        if (prev != dev) {
            HIP_TRY(hipSetDevice(dev));
            switched = true;
        }
        return NAQS_OK;
    }
    ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

inline int env_int(const char *name, int dflt) {
    const char *v = std::getenv(name);
    return v ? std::atoi(v) : dflt;
}

// HIP-event pairs around the launches of one kernel (bench.py's roofline leg)
struct EventRing {
    hipEvent_t *ev = nullptr;
    int cap = 0, used = 0;
    int stride = 1, tick = 0;            // record every stride-th launch (an event pair costs ~4 us of queue time)
    int enable(int max_records) {
        for (int i = 0; i < cap; ++i) (void)hipEventDestroy(ev[i]);
        std::free(ev);
        ev = nullptr; cap = used = 0; tick = 0;
        if (max_records <= 0) return NAQS_OK;
        ev = static_cast<hipEvent_t *>(std::calloc((size_t)2 * max_records, sizeof(hipEvent_t)));
        if (!ev) return NAQS_ERR_NOMEM;
        for (int i = 0; i < 2 * max_records; ++i) { HIP_TRY(hipEventCreate(&ev[i])); cap = i + 1; }
        return NAQS_OK;
    }
    bool armed() { return used + 2 <= cap && (tick++ % stride) == 0; }
    int begin(hipStream_t s) { HIP_TRY(hipEventRecord(ev[used], s)); return NAQS_OK; }
    int end(hipStream_t s) { HIP_TRY(hipEventRecord(ev[used + 1], s)); used += 2; return NAQS_OK; }
    int read(double *total_ms, int64_t *launches) {
        double tot = 0;
        for (int i = 0; i + 1 < used; i += 2) {
            HIP_TRY(hipEventSynchronize(ev[i + 1]));
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, ev[i], ev[i + 1]));
            tot += ms;
        }
        *total_ms = tot;
        *launches = used / 2;
        used = 0;
        return NAQS_OK;
    }
};

struct ElocFeed;
// implemented in naqs_hip.hip; used by the fused entry point naqs_logpsi_eloc
int eloc_begin(naqs_ham *h, int64_t M, hipStream_t s, ElocFeed *feed);
int eloc_main(naqs_ham *h, int64_t M, const ElocFeed &feed, double *eloc_dev, const double *w_dev, double *out4_dev,
              hipStream_t s);
int ham_device(const naqs_ham *h);

}  // namespace naqs
