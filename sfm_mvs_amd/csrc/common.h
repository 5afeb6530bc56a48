// Shared host-side helpers for libsfmhip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include "../../include/sfm_hip.h"

namespace sfm {

void set_error(const char* fmt, ...);

inline hipStream_t as_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Optional per-kernel HIP-event timing (sfm_profile_enable): slot ids
enum ProfSlot { kProfKnnFilter = 0, kProfKnnRefine = 1, kProfTriangulate = 2, kProfBaDense = 3, kProfResidual = 4, kProfBaSchur = 5, kProfSiftPyramid = 6, kProfSiftDescriptor = 7, kProfSlots = 8 };
void prof_begin(int slot, hipStream_t s);
void prof_end(int slot, hipStream_t s, int launches = 1);
int prof_repeat();   // launches per event pair requested with sfm_profile_enable(n > 1) (1 when profiling is off)

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Every point at which a library call WAITS for the device (the RANSAC entry points read hypothesis scores back chunk by chunk,
// the Schur solver its convergence scalars) goes through this: sfm_host_sync_count() is what bench.py divides by the cameras
// registered to state "host synchronisations per frame" (VERDICT r04 item 5).
void note_host_sync();
inline hipError_t stream_sync(hipStream_t s) { note_host_sync(); return hipStreamSynchronize(s); }

// Camera chunks of the dense BA kernels (grid = point tiles x chunks, a workgroup walks ncam / chunks cameras for its tile
// of points): the launch takes ceil(tiles * chunks / slots) rounds of ceil(ncam / chunks) cameras each, `slots` = the
// workgroups the chip holds at once for that kernel (256 CUs x its occupancy).  Round 2 aimed at ">= 1024 workgroups" and
// got 1176 for config 4 — 2.3 rounds on 512 slots, i.e. three: a quarter of the sweep was a third round a third full.
// Every chunk also costs a partial row per point (written, then folded): ~0.6 % of a sweep each.
inline int pick_camera_chunks(int tiles, long long ncam, int slots, int max_chunks = 64) {
    long long cap = ncam / 16;
    if (cap > max_chunks) cap = max_chunks;
    if (cap < 1) cap = 1;
    int best = 1;
    double best_cost = 1e300;
    for (int n = 1; n <= (int)cap; ++n) {
        const long long rounds = ((long long)tiles * n + slots - 1) / slots, cams = (ncam + n - 1) / n;
        const double cost = (double)(rounds * cams) * (1.0 + 0.006 * n);
        if (cost < best_cost) { best_cost = cost; best = n; }
    }
    return best;
}

// Carve aligned sub-buffers out of the caller's workspace.
struct Carver {
    char* base;
    size_t off = 0;
    explicit Carver(void* p) : base(static_cast<char*>(p)) {}
    template <typename T> T* take(size_t count) {
        off = align_up(off, 256);
        T* p = reinterpret_cast<T*>(base + off);
        off += count * sizeof(T);
        return p;
    }
    size_t used() const { return align_up(off, 256); }
};

}  // namespace sfm

#define SFM_CHECK_ARG(cond, ...)                 \
    do {                                         \
        if (!(cond)) {                           \
            sfm::set_error(__VA_ARGS__);         \
            return SFM_ERR_ARG;                  \
        }                                        \
    } while (0)

#define SFM_CHECK_HIP(expr)                                                           \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) {                                                       \
            sfm::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_),     \
                           __FILE__, __LINE__);                                       \
            return SFM_ERR_DEVICE;                                                    \
        }                                                                             \
    } while (0)

#define SFM_CHECK_LAUNCH()                                                            \
    do {                                                                              \
        hipError_t e_ = hipGetLastError();                                            \
        if (e_ != hipSuccess) {                                                       \
            sfm::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), \
                           __FILE__, __LINE__);                                       \
            return SFM_ERR_DEVICE;                                                    \
        }                                                                             \
    } while (0)
