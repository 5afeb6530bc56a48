// Error channel + ABI version of libsfmhip.so.
#include "common.h"
#include <cstring>
#include <mutex>
#include <utility>
#include <atomic>
#include <vector>

namespace sfm {
static std::atomic<long long> g_host_syncs{0};
void note_host_sync() { g_host_syncs.fetch_add(1, std::memory_order_relaxed); }

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// ---- measurement hook: HIP events recorded on the launch stream around selected kernels --------
namespace {
struct ProfState {
    bool on = false;
    int repeat = 1;                      // launches per event pair for kernels that support it (sfm_profile_enable(n > 1))
    size_t launches[kProfSlots] = {0};
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[kProfSlots];
    size_t used[kProfSlots] = {0};
    hipEvent_t pending[kProfSlots] = {nullptr};
};
ProfState g_prof;
std::mutex g_prof_mu;
}  // namespace

void prof_begin(int slot, hipStream_t s) {
    if (!g_prof.on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    auto& v = g_prof.ev[slot];
    if (g_prof.used[slot] == v.size()) {
        hipEvent_t a, b;
        if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return;
        v.emplace_back(a, b);
    }
    (void)hipEventRecord(v[g_prof.used[slot]].first, s);
}

void prof_end(int slot, hipStream_t s, int launches) {
    if (!g_prof.on) return;
    std::lock_guard<std::mutex> lk(g_prof_mu);
    auto& v = g_prof.ev[slot];
    if (g_prof.used[slot] >= v.size()) return;
    (void)hipEventRecord(v[g_prof.used[slot]].second, s);
    ++g_prof.used[slot];
    g_prof.launches[slot] += (size_t)launches;
}

int prof_repeat() { return g_prof.on ? g_prof.repeat : 1; }
}  // namespace sfm

extern "C" int sfm_profile_enable(int on) {
    std::lock_guard<std::mutex> lk(sfm::g_prof_mu);
    sfm::g_prof.on = on != 0;          // slots keep accumulating across on/off toggles; sfm_profile_read resets one
    sfm::g_prof.repeat = on > 1 ? on : 1;
    return SFM_OK;
}

extern "C" int sfm_profile_read(int slot, double* total_ms, int64_t* launches) {
    SFM_CHECK_ARG(slot >= 0 && slot < sfm::kProfSlots && total_ms && launches, "sfm_profile_read: bad argument");
    std::lock_guard<std::mutex> lk(sfm::g_prof_mu);
    double tot = 0;
    const size_t n = sfm::g_prof.used[slot];
    for (size_t i = 0; i < n; ++i) {
        auto& e = sfm::g_prof.ev[slot][i];
        SFM_CHECK_HIP(hipEventSynchronize(e.second));
        float ms = 0;
        SFM_CHECK_HIP(hipEventElapsedTime(&ms, e.first, e.second));
        tot += ms;
    }
    *total_ms = tot;
    *launches = (int64_t)sfm::g_prof.launches[slot];
    sfm::g_prof.used[slot] = 0;
    sfm::g_prof.launches[slot] = 0;
    return SFM_OK;
}

extern "C" int64_t sfm_host_sync_count(void) { return (int64_t)sfm::g_host_syncs.load(std::memory_order_relaxed); }
extern "C" int sfm_abi_version(void) { return SFM_ABI_VERSION; }

// What this BINARY was built from: the code hash of every source file (scripts/knn_code_hash.py: comments and whitespace do not
// count), handed in by the Makefile: "knn.hip:<sha256> assoc.hip:<first 16 digits> ... sfm_hip.h:<16>".  The committed fuzz logs
// and PMC traffic stamps name it; tests/test_gpu_knn.py and bench.py compare them with the LOADED library's id, file by file —
// the KNN logs against knn.hip, the SIFT logs against sift.hip, the geometry / pipeline logs against theirs.
#ifndef SFM_BUILD_ID
#define SFM_BUILD_ID "knn.hip:unknown"
#endif
extern "C" const char* sfm_build_id(void) {
#ifdef SFM_DEV_BUILD
    return SFM_BUILD_ID " dev-build";
#else
    return SFM_BUILD_ID;
#endif
}
extern "C" const char* sfm_last_error(void) { return sfm::g_err; }
