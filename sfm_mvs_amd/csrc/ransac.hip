// The RANSAC entry points of the path as single C-ABI calls:
//
//   sfm_find_essential_mat   cv2.findEssentialMat(pts0, pts1, K, RANSAC, 0.999, 0.4)        sfm.py:307
//   sfm_recover_pose         cv2.recoverPose(E, pts0, pts1, K)                              sfm.py:311
//   sfm_solve_pnp_ransac     cv2.solvePnPRansac(X, p, K, zeros(5,1), <stray arg>)           sfm.py:67
//
// Division of labour (the MI355X-native shape of OpenCV's RANSACPointSetRegistrator): hypothesis GENERATION is a
// few dozen doubles of sequential arithmetic per iteration and stays on the host (host_solvers.h: five-point, EPnP,
// the ITERATIVE DLT initialisation, the 6x6 Levenberg-Marquardt step); everything that is a sweep over the
// correspondences runs on the device — Sampson / reprojection scoring of a whole CHUNK of hypotheses in one launch
// (H x N lanes, integer counts + masks), the cheirality vote of recoverPose, and the Gauss-Newton residual / J^T J
// sweep of the refinement.  The subsets RANSAC draws do not depend on the scores, so iterations are generated in
// chunks (8, 16, ... 64 iterations), scored in ONE launch, and the host then replays OpenCV's sequential bookkeeping
// over the returned counts: every model of an iteration is examined, the best is replaced only on a STRICTLY larger
// count (> max(best, modelPoints-1)), `niters` is re-estimated then and looked at only between iterations — the result
// is the one a one-model-at-a-time loop produces.  Per chunk: one upload (models), one launch, one download (counts).
// The winning mask never leaves the device except for the PnP inlier list.
//
// Points come in as device pointers (float32, as the reference holds them); the few values the host-side solvers need
// (the sampled correspondences, and the inliers for the DLT initialisation) are read from one host copy made per call.
#include "common.h"
#include "host_solvers.h"
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <cstring>
#include <chrono>
#include <cstdlib>
#include <utility>
#include <vector>

namespace {

namespace hs = sfm::host;

// points.col(0) = (points.col(0) - cx) / fx as OpenCV evaluates it: ONE scaled conversion x * (1/fx) + (-cx * (1/fx)),
// multiply and add separately rounded (the library is built with -ffp-contract=off).
struct KNorm {
    double ifx, ify, bx, by;
    explicit KNorm(const double* K) : ifx(1. / K[0]), ify(1. / K[4]), bx(-K[2] * (1. / K[0])), by(-K[5] * (1. / K[4])) {}
};

__global__ __launch_bounds__(256) void k_normalise_kernel(const float* __restrict__ pts, int64_t n, KNorm k,
                                                          double* __restrict__ out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 p = reinterpret_cast<const float2*>(pts)[i];
    reinterpret_cast<double2*>(out)[i] = make_double2((double)p.x * k.ifx + k.bx, (double)p.y * k.ify + k.by);
}

// ---- Gauss-Newton sweep of solvePnP(ITERATIVE) over the inlier set ---------------------------------------------------
// One lane per inlier (grid-stride inside ONE to 64 workgroups): projection of the float32 object point through
// (R, t, K) in fp64, residual against the float32 observation, and — mode 1 — the 2x6 Jacobian rows of OpenCV's
// projectPoints (dp/drvec through dR/drvec, dp/dtvec) folded into the upper triangle of J^T J (21), J^T e (6) and
// |e|^2.  R, t and dR/dr arrive as kernel arguments (computed on the host, so the sweep has no transcendental math).
// Sums are reduced in ONE FIXED TREE — the same tree oracle/solvers_oracle.c (lm_tree_sums) walks, so that Levenberg-Marquardt's
// accept / reject tests, hence the refined pose, hence the whole camera chain, are bit-reproducible between the two:
//   term of point o:  Ju[a] Ju[b] + Jv[a] Jv[b]  |  Ju[a] ru + Jv[a] rv  |  ru ru + rv rv
//   slots:            G = min(ceil(m / 1024), 64) workgroups of 1024 lanes; point o -> lane o mod 1024 G, a lane adds its points
//                     in increasing o starting from 0
//   workgroup:        each wave folds its 64 lanes by the butterfly v[l] += v[l ^ s], s = 32 .. 1; the workgroup's sum is
//                     0 + wave 0 + ... + wave 15
//   total:            G = 1: the workgroup's sum;  G > 1: 0 + workgroup 0 + workgroup 1 + ...   (pnp_sweep_fold_kernel)
// (OpenCV's own order — cvMulTransposed / cvGEMM / cvNorm SIMD loops — is not pinned by anything in the reference.)
struct PnpCam {
    double R[9], t[3], dR[27], fx, fy, cx, cy;
};
constexpr int kSweepAcc = 28;        // 21 + 6 + 1
constexpr int kSweepThreads = 1024, kSweepMaxBlocks = 64;

// one correspondence's terms added to a lane's 28 running sums (shared by the launch-per-sweep kernel and the sweep server)
template <int MODE>
__device__ __forceinline__ void pnp_sweep_point(const PnpCam& cam, const float* __restrict__ X, const float* __restrict__ uv, int64_t i,
                                                double (&acc)[kSweepAcc]) {
    const double Xw = X[3 * i], Yw = X[3 * i + 1], Zw = X[3 * i + 2];
    double x = cam.R[0] * Xw + cam.R[1] * Yw + cam.R[2] * Zw + cam.t[0];
    double y = cam.R[3] * Xw + cam.R[4] * Yw + cam.R[5] * Zw + cam.t[1];
    double z = cam.R[6] * Xw + cam.R[7] * Yw + cam.R[8] * Zw + cam.t[2];
    z = z != 0.0 ? 1. / z : 1.;
    x *= z;
    y *= z;
    const double ru = (x * cam.fx + cam.cx) - (double)uv[2 * i], rv = (y * cam.fy + cam.cy) - (double)uv[2 * i + 1];
    acc[27] += ru * ru + rv * rv;
    if (MODE == 1) {
        double Ju[6], Jv[6];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double* d = cam.dR + 9 * j;
            const double dx0 = Xw * d[0] + Yw * d[1] + Zw * d[2];
            const double dy0 = Xw * d[3] + Yw * d[4] + Zw * d[5];
            const double dz0 = Xw * d[6] + Yw * d[7] + Zw * d[8];
            Ju[j] = cam.fx * (z * (dx0 - x * dz0));
            Jv[j] = cam.fy * (z * (dy0 - y * dz0));
        }
        Ju[3] = cam.fx * z; Ju[4] = 0;          Ju[5] = cam.fx * (-x * z);
        Jv[3] = 0;          Jv[4] = cam.fy * z; Jv[5] = cam.fy * (-y * z);
        int q = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int b = a; b < 6; ++b) acc[q++] += Ju[a] * Ju[b] + Jv[a] * Jv[b];
#pragma unroll
        for (int a = 0; a < 6; ++a) acc[21 + a] += Ju[a] * ru + Jv[a] * rv;
    }
}

// A lane's 28 sums -> its wave's, wacc[wave][k].  The TREE is the butterfly v[l] += v[l ^ s], s = 32 .. 1 (what the oracle walks);
// the SCHEDULE is a reduce-scatter: at step s a lane keeps half of its values and hands the other half to lane l ^ s, which
// keeps exactly those — every pair sum a[l] + a[l ^ s] is still formed (by one of the two lanes instead of both: IEEE addition
// commutes, so the bits are the same), but 28 values cost 14 + 7 + 4 + 2 + 1 + 1 = 29 exchanges instead of 28 x 6 = 168.
// After the last step lanes l and l ^ 1 hold the wave's sum of value
//   idx(l) = 14 b5 + 7 b4 + (4 b3 + 2 b2 + b1)        (b_i = bit i of l; 4 b3 + 2 b2 + b1 = 7 is an unused slot).
template <int N>
__device__ __forceinline__ void scatter_step(double (&v)[14], int bit, int s) {
    constexpr int H = (N + 1) / 2;
#pragma unroll
    for (int j = 0; j < H; ++j) {
        const double hi = H + j < N ? v[H + j] : 0.0;
        const double send = bit ? v[j] : hi, keep = bit ? hi : v[j];
        v[j] = keep + __shfl_xor(send, s, 64);
    }
}
template <int KFIRST>
__device__ __forceinline__ void pnp_sweep_wave_fold(const double (&acc)[kSweepAcc], double (*wacc)[kSweepAcc]) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (KFIRST != 0) {                                   // |e|^2 alone: the plain butterfly
        double v = acc[kSweepAcc - 1];
#pragma unroll
        for (int s = 32; s >= 1; s >>= 1) v += __shfl_xor(v, s, 64);
        if (lane == 0) wacc[wave][kSweepAcc - 1] = v;
        return;
    }
    double v[14];
    {
        const int b = lane & 32;
#pragma unroll
        for (int j = 0; j < 14; ++j) {
            const double send = b ? acc[j] : acc[14 + j], keep = b ? acc[14 + j] : acc[j];
            v[j] = keep + __shfl_xor(send, 32, 64);
        }
    }
    scatter_step<14>(v, lane & 16, 16);
    scatter_step<7>(v, lane & 8, 8);
    scatter_step<4>(v, lane & 4, 4);
    scatter_step<2>(v, lane & 2, 2);
    v[0] = v[0] + __shfl_xor(v[0], 1, 64);
    const int low = (lane >> 1) & 7;
    if ((lane & 1) == 0 && low != 7) wacc[wave][(lane & 32 ? 14 : 0) + (lane & 16 ? 7 : 0) + low] = v[0];
}

template <int MODE>
__global__ __launch_bounds__(kSweepThreads) void pnp_sweep_kernel(PnpCam cam, const float* __restrict__ X, const float* __restrict__ uv,
                                                                  const int32_t* __restrict__ sel, int64_t m,
                                                                  double* __restrict__ partials) {
    __shared__ double wacc[kSweepThreads / 64][kSweepAcc];
    double acc[kSweepAcc];
#pragma unroll
    for (int k = 0; k < kSweepAcc; ++k) acc[k] = 0;
    for (int64_t o = blockIdx.x * (int64_t)kSweepThreads + threadIdx.x; o < m; o += (int64_t)gridDim.x * kSweepThreads)
        pnp_sweep_point<MODE>(cam, X, uv, sel ? sel[o] : o, acc);
    constexpr int kFirst = MODE == 1 ? 0 : 27;
    if (blockIdx.x * (int64_t)kSweepThreads + (threadIdx.x & ~63) < m)
        pnp_sweep_wave_fold<kFirst>(acc, wacc);
    else if ((threadIdx.x & 63) < kSweepAcc)             // a wave without points: the butterfly of zeros is +0
        wacc[threadIdx.x >> 6][threadIdx.x & 63] = 0.0;
    __syncthreads();
    if (threadIdx.x >= kFirst && threadIdx.x < kSweepAcc) {
        double s = 0;
        for (int w = 0; w < kSweepThreads / 64; ++w) s += wacc[w][threadIdx.x];
        partials[(int64_t)blockIdx.x * kSweepAcc + threadIdx.x] = s;
    }
}

// ---- the PnP SERVER: sfm_solve_pnp_ransac without a launch, a copy or a stream synchronisation per step -------------------------
// solvePnPRansac is a chain of small dependent steps — read the correspondences, score a chunk of hypotheses, list the inliers,
// then cvFindExtrinsicCameraParams2's loop  parameters -> sweep -> 6x6 damped SVD solve -> accept / reject  (5-6 links) — each of
// which round 5 paid for with a launch, a completion signal and a host wake-up (~30 us per link; the work itself is 1-3 us).  The
// algebra stays on the host: the 6x6 one-sided Jacobi SVD is ~90 strictly sequential rotations, each a dependent chain of a division,
// two square roots and a hypot — ~2 us on a host core, an estimated 30-35 us on ONE lane of a CU (a dependent fp64 operation issues
// every 13 ns there: scripts/ubench/idle_clock.hip) — so moving it to the device would cost what the round trip costs.  The DEVICE
// side becomes a server instead: a few workgroups, launched once per call, resident for its duration, that answer requests posted in
// the caller's pinned, fine-grained mailbox (PnpMailbox).  The host writes the payload, then ONE 64-bit word (request number,
// command, argument); lane 0 of every workgroup polls that word (system-scope loads); the answer goes back into the mailbox and the
// leader bumps `done_seq`, on which the host spins in its own cache.  A link is two PCIe hops + the work.  Commands:
//   copy-in    the correspondences (float32 X, uv) -> mailbox: what the host-side hypothesis generators sample from
//   score      H models (R, t computed by the HOST: the Rodrigues the oracle and cv2 evaluate on the CPU) -> one 64-bit inlier mask per
//              wave and model; the host counts bits, replays OpenCV's bookkeeping and reads the winner's inlier list off the mask
//   sweep      [first of a selection: the winner's ascending inlier list -> the lanes' selection (point of tree lane L) and the caller's
//              `inliers_dev`;]  R, t, dR/dr -> the 28 sums of J^T J, J^T e, |e|^2 over the inliers, pnp_sweep_kernel's FIXED TREE
//   quit
// Layout: the tree's lane L (virtual workgroup L / 1024, wave (L % 1024) / 64) is point L of the current selection.  <= 1 024
// points: ONE workgroup of 1 024 lanes.  More: workgroups of 256 lanes (one tree wave per SIMD: a wave alone on its SIMD issues four
// times as fast as sixteen sharing a CU); results meet in device memory behind one ticket per workgroup and request, and the leader
// (workgroup 0) adds   0 + wave 0 + ... + wave 15   per virtual workgroup, then   0 + workgroup 0 + ...   — pnp_sweep_kernel's
// order; a wave without points contributes +0, which changes no partial sum.  The server leaves on "quit", or on its own after
// kServerTimeoutTicks without a request (the host then starts another and replays the selection: a descheduled host thread must not
// be able to hang a queue).
constexpr int kServerMaxG = 8, kServerMaxN = 1024 * kServerMaxG;   // correspondences a served call may have; larger ones keep the launch path
constexpr int kServerMaxWaves = kServerMaxN / 64, kServerMaxModels = 64;
constexpr long long kServerTimeoutTicks = 100000000; // 1 s of the 100 MHz wall clock
enum : uint32_t { kCmdNone = 0, kCmdSweep = 1, kCmdQuit = 2, kCmdCopyIn = 3, kCmdScore = 4, kServerLeft = 0xFFFFFFFFu };
struct alignas(64) PnpMailbox {
    unsigned long long request, pad0[7];             // host -> device, written last: number << 40 | command << 32 | argument
    PnpCam cam;                                      // sweep: R, t, dR/dr, intrinsics (score: intrinsics only)
    alignas(64) uint32_t done_seq, pad1[15];         // device -> host (written last); kServerLeft: the server gave up waiting
    double sums[kSweepAcc];
    alignas(64) double models[kServerMaxModels * 12];               // score: R (9, row-major), t (3) per model
    unsigned long long bits[kServerMaxModels * kServerMaxWaves];    // score: [model][wave] inlier masks
};
constexpr unsigned long long kSeqMask = 0xFFFFFFull;

__device__ __forceinline__ unsigned long long sys_load_u64(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); }

template <int kServerThreads>
__global__ __launch_bounds__(kServerThreads) void pnp_server_kernel(PnpMailbox* __restrict__ mb, const float* __restrict__ X, const float* __restrict__ uv,
                                                                    int n, float thr2, float* __restrict__ hX, float* __restrict__ huv,
                                                                    const int32_t* __restrict__ hinl, int32_t* __restrict__ inliers_dev,
                                                                    uint32_t first_seq, double* __restrict__ part, unsigned int* __restrict__ arrive) {
    constexpr int kWavesPerWg = kServerThreads / 64;
    constexpr bool kSingle = kServerThreads == 1024;
    __shared__ double wacc[kWavesPerWg][kSweepAcc];
    __shared__ double fold[kSingle ? 1 : kServerMaxWaves][kSweepAcc + 1];
    __shared__ double models[kServerMaxModels * 12];
    __shared__ PnpCam cam;
    __shared__ unsigned long long s_req;
    const int tid = threadIdx.x, wg = blockIdx.x, nwg = gridDim.x;
    const int L = wg * kServerThreads + tid, lane = tid & 63;     // the tree's lane; its wave: L >> 6
    int m = 0, nwaves = 0, i0 = -1;                                // the current selection (set by "inliers")
    unsigned int served = 0;
    // every workgroup's stores have left, and (several workgroups) every workgroup has arrived, before the leader answers
    // Every store that leaves a workgroup here is WRITE-THROUGH — the mailbox is fine-grained host memory (uncached on the device),
    // `part` is written with agent-scope atomic stores — so "has left" is a wait for the lane's own outstanding stores, not a cache
    // write-back (a system-scope release fence per wave walks the L2: 37 us per request with 24 workgroups, measured).
    auto finish = [&](uint32_t seq) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // EVERY lane: its own stores have been acknowledged
        __syncthreads();
        if (!kSingle) {
            ++served;
            if (tid == 0) {
                __hip_atomic_fetch_add(arrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (wg == 0) {
                    const unsigned int want = served * (unsigned int)nwg;
                    while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != want) __builtin_amdgcn_s_sleep(1);
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                }
            }
            __syncthreads();
        }
        (void)seq;
    };
    auto answer = [&](uint32_t seq) {                              // (leader, after its last mailbox store)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (wg == 0 && tid == 0) __hip_atomic_store(&mb->done_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    };
    for (uint32_t seq = first_seq & (uint32_t)kSeqMask;; seq = (seq + 1) & (uint32_t)kSeqMask) {
        if (tid == 0) {
            const long long t0 = wall_clock64();
            unsigned long long r = 0;
            for (;;) {
                const unsigned long long w = sys_load_u64(&mb->request);
                if ((uint32_t)(w >> 40) == seq) {
                    __atomic_thread_fence(__ATOMIC_ACQUIRE);
                    r = w;
                    break;
                }
                if (wall_clock64() - t0 > kServerTimeoutTicks) break;
                __builtin_amdgcn_s_sleep(1);
            }
            s_req = r;
        }
        __syncthreads();
        const unsigned long long req = s_req;
        const uint32_t cmd = (uint32_t)(req >> 32) & 0xFFu, arg = (uint32_t)req;
        if (cmd == kCmdQuit || cmd == kCmdNone) {
            if (wg == 0 && tid == 0)
                __hip_atomic_store(&mb->done_seq, cmd == kCmdQuit ? seq : (uint32_t)kServerLeft, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
            return;
        }
        if (cmd == kCmdCopyIn) {
            // plain, lane-contiguous dword stores: a wave's 256 bytes leave as whole 64-byte writes (one system-scope store per value
            // was 90 ns EACH over PCIe: 139 us for 300 points)
            const int stride = nwg * kServerThreads;
            for (int e = L; e < 3 * n; e += stride) hX[e] = X[e];
            for (int e = L; e < 2 * n; e += stride) huv[e] = uv[e];
            __atomic_thread_fence(__ATOMIC_RELEASE);                 // (plain stores: written back to host memory here — the one request that needs it)
            finish(seq);
            answer(seq);
            continue;
        }
        if (cmd == kCmdScore) {
            // score_pnp_kernel's arithmetic on R, t the host evaluated: err = (float)(ex^2 + ey^2) of the float32 differences, in <= thr2
            const int H = (int)arg;
            for (int e = tid; e < 12 * H; e += kServerThreads)
                models[e] = __hip_atomic_load(&mb->models[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (tid < 4) reinterpret_cast<double*>(&cam.fx)[tid] = __hip_atomic_load(&mb->cam.fx + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __syncthreads();
            float Xw = 0.f, Yw = 0.f, Zw = 0.f, ou = 0.f, ov = 0.f;
            if (L < n) { Xw = X[3 * L]; Yw = X[3 * L + 1]; Zw = X[3 * L + 2]; ou = uv[2 * L]; ov = uv[2 * L + 1]; }
            if ((L & ~63) < n)
                for (int h = 0; h < H; ++h) {
                    const double* e = models + 12 * h;
                    double x = e[0] * (double)Xw + e[1] * (double)Yw + e[2] * (double)Zw + e[9];
                    double y = e[3] * (double)Xw + e[4] * (double)Yw + e[5] * (double)Zw + e[10];
                    double z = e[6] * (double)Xw + e[7] * (double)Yw + e[8] * (double)Zw + e[11];
                    z = z != 0.0 ? 1. / z : 1.;
                    x *= z;
                    y *= z;
                    const double u = x * cam.fx + cam.cx, v = y * cam.fy + cam.cy;
                    const float ex = ou - (float)u, ey = ov - (float)v;
                    const float err = (float)((double)ex * (double)ex + (double)ey * (double)ey);
                    const unsigned long long bal = __ballot(L < n && err <= thr2);
                    if (lane == 0) __hip_atomic_store(&mb->bits[(size_t)h * kServerMaxWaves + (L >> 6)], bal, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            finish(seq);
            answer(seq);
            continue;
        }
        // ---- sweep (argument m + 1 > 0: FIRST take the host's ascending inlier list as the selection — point of tree lane L — and
        // write the caller's `inliers_dev`; a separate, unacknowledged "inliers" request could be overwritten by the sweep request
        // that follows it before a workgroup had polled it: found by scripts/fuzz_geometry.py on the calls whose DLT initialisation
        // returns at once)
        if (arg) {
            m = (int)arg - 1;
            nwaves = (m + 63) >> 6;
            i0 = -1;
            if (L < m) {
                i0 = __hip_atomic_load(&hinl[L], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                inliers_dev[L] = i0;
            }
        }
        if (tid < (int)(sizeof(PnpCam) / sizeof(double)))
            reinterpret_cast<double*>(&cam)[tid] =
                __hip_atomic_load(reinterpret_cast<const double*>(&mb->cam) + tid, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        __syncthreads();
        double acc[kSweepAcc];
#pragma unroll
        for (int k = 0; k < kSweepAcc; ++k) acc[k] = 0;
        if (i0 >= 0) pnp_sweep_point<1>(cam, X, uv, i0, acc);
        if ((L & ~63) < m) pnp_sweep_wave_fold<0>(acc, wacc);       // (a wave without points: never read)
        __syncthreads();
        const int G = (m + 1023) >> 10;
        if (kSingle) {
            if (tid < kSweepAcc) {
                double s = 0;
                for (int w = 0; w < nwaves; ++w) s += wacc[w][tid];
                __hip_atomic_store(&mb->sums[tid], s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        } else {
            if (tid < kWavesPerWg * kSweepAcc && wg * kWavesPerWg + tid / kSweepAcc < nwaves)
                __hip_atomic_store(&part[(size_t)(wg * kWavesPerWg + tid / kSweepAcc) * kSweepAcc + tid % kSweepAcc], wacc[tid / kSweepAcc][tid % kSweepAcc],
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            finish(seq);
            if (wg == 0) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // every reader's cache, not only lane 0's
                for (int e = tid; e < nwaves * kSweepAcc; e += kServerThreads)
                    fold[e / kSweepAcc][e % kSweepAcc] = __hip_atomic_load(&part[e], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __syncthreads();
                if (tid < kSweepAcc) {
                    double total = 0;
                    for (int g = 0; g < G; ++g) {
                        double s = 0;
                        const int w1 = min(16 * (g + 1), nwaves);
                        for (int w = 16 * g; w < w1; ++w) s += fold[w][tid];
                        total = G == 1 ? s : total + s;
                    }
                    __hip_atomic_store(&mb->sums[tid], total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                }
            }
        }
        answer(seq);
    }
}

__global__ void pnp_sweep_fold_kernel(const double* __restrict__ partials, int blocks, double* __restrict__ out) {
    const int k = threadIdx.x;
    if (k >= kSweepAcc) return;
    double s = 0;
    for (int b = 0; b < blocks; ++b) s += partials[(int64_t)b * kSweepAcc + k];
    out[k] = s;
}

// cv2.projectPoints on float64 object points (the reference's bundle-adjustment residual, sfm.py:119-121, is fp64 end to
// end: SciPy's finite-difference steps of ~1.5e-8 relative vanish in a float32 round trip)
__global__ __launch_bounds__(256) void project_f64_kernel(PnpCam cam, const double* __restrict__ X, int64_t n, double* __restrict__ out) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double Xw = X[3 * i], Yw = X[3 * i + 1], Zw = X[3 * i + 2];
    double x = cam.R[0] * Xw + cam.R[1] * Yw + cam.R[2] * Zw + cam.t[0];
    double y = cam.R[3] * Xw + cam.R[4] * Yw + cam.R[5] * Zw + cam.t[1];
    double z = cam.R[6] * Xw + cam.R[7] * Yw + cam.R[8] * Zw + cam.t[2];
    z = z != 0.0 ? 1. / z : 1.;
    out[2 * i] = (x * z) * cam.fx + cam.cx;
    out[2 * i + 1] = (y * z) * cam.fy + cam.cy;
}

// cv::Rodrigues vector -> matrix with dR/dr (3 x 9), host side of the sweep
void rodrigues_with_jac(const double* rv, double* R, double* J) {
    const double theta = std::sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    if (theta < DBL_EPSILON) {
        for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
        std::memset(J, 0, 27 * sizeof(double));
        J[5] = J[15] = J[19] = -1;
        J[7] = J[11] = J[21] = 1;
        return;
    }
    // sin and cos from ONE glibc sincos() call — what gcc makes of cv::Rodrigues' `cos(theta)` / `sin(theta)` pair (and of the oracle's) — not
    // from sin() and cos(): glibc rounds the two differently for 0.12 % of the arguments (247 of 200 000 random rotation vectors, 1 ulp),
    // and one such ulp in a trial step's R flips a Levenberg-Marquardt accept / reject at convergence: the 1e-10 pose differences that 3 %
    // of the sequences of scripts/fuzz_pipeline.py showed in rounds 4-5 (found in round 6 by tracing both sides' sweeps)
    double c, s;
    ::sincos(theta, &s, &c);
    const double c1 = 1. - c, itheta = 1. / theta;
    const double r[3] = {rv[0] * itheta, rv[1] * itheta, rv[2] * itheta};
    const double rrt[9] = {r[0] * r[0], r[0] * r[1], r[0] * r[2], r[0] * r[1], r[1] * r[1], r[1] * r[2], r[0] * r[2], r[1] * r[2], r[2] * r[2]};
    const double rx[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
    for (int k = 0; k < 9; ++k) R[k] = c * ((k % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[k] + s * rx[k];
    const double drrt[27] = {r[0] + r[0], r[1], r[2], r[1], 0, 0, r[2], 0, 0, 0, r[0], 0, r[0], r[1] + r[1], r[2], 0, r[2], 0,
                             0, 0, r[0], 0, 0, r[1], r[0], r[1], r[2] + r[2]};
    const double drx[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
    for (int i = 0; i < 3; ++i) {
        const double ri = r[i];
        const double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta, a3 = (c - s * itheta) * ri, a4 = s * itheta;
        for (int k = 0; k < 9; ++k)
            J[i * 9 + k] = a0 * ((k % 4 == 0) ? 1.0 : 0.0) + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * rx[k] + a4 * drx[i * 9 + k];
    }
}

// Chunk sizes of the batched RANSAC: small first (with a high inlier ratio `niters` collapses after the first good
// model and hypotheses generated beyond it are wasted host work), doubling up to 64 iterations per launch.
struct Chunker {
    int chunk;
    explicit Chunker(int first = 8) : chunk(first) {}
    int next(int remaining) {
        const int m = std::min(chunk, remaining);
        chunk = std::min(2 * chunk, 64);
        return m;
    }
};

// Sequential bookkeeping of RANSACPointSetRegistrator::run over one scored chunk.  owner[j] = iteration (relative to
// `it0`) that produced model j; returns the index of the chunk's model that became the overall best (-1: none) and sets
// `stop` when the loop condition `iter < niters` failed at an iteration boundary inside the chunk.
struct RansacState {
    int niters, best = 0, model_points;
    double confidence;
    int64_t count;
    int last_improve = -1, models_scored = 0;
    // iterations the sequential loop enters: `niters` only shrinks, and only inside the iteration that improved
    int iterations_run() const { return std::max(niters, last_improve + 1); }
};

int replay_chunk(RansacState& st, int it0, const std::vector<int>& owner, const int32_t* counts, bool& stop) {
    int best_j = -1, last_k = -1;
    stop = false;
    for (size_t j = 0; j < owner.size(); ++j) {
        const int k = owner[j];
        if (k != last_k) {                       // entering iteration it0 + k: the loop condition is evaluated here only
            if (it0 + k >= st.niters) {
                stop = true;
                break;
            }
            last_k = k;
        }
        ++st.models_scored;
        const int good = counts[j];
        if (good > std::max(st.best, st.model_points - 1)) {
            st.best = good;
            best_j = (int)j;
            st.last_improve = it0 + k;
            st.niters = hs::ransac_update_num_iters(st.confidence, (double)(st.count - good) / (double)st.count, st.model_points, st.niters);
        }
    }
    return best_j;
}

// dev: SFM_PNP_PROF=1 prints where sfm_solve_pnp_ransac's host time goes (accumulated, at process exit)
struct PnpProf {
    bool on = getenv("SFM_PNP_PROF") != nullptr;
    double t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    long calls = 0, chunks = 0, sweeps = 0;
    ~PnpProf() {
        if (on && calls)
            fprintf(stderr, "[pnp prof] calls %ld  per call us: copy-in %.1f  epnp %.1f  score+sync %.1f  mask/inliers %.1f  dlt %.1f  lm-sweeps %.1f  lm-host %.1f | chunks/call %.2f sweeps/call %.2f\n",
                    calls, t[0] / calls, t[1] / calls, t[2] / calls, t[3] / calls, t[4] / calls, t[5] / calls, t[6] / calls, (double)chunks / calls, (double)sweeps / calls);
    }
};
PnpProf g_pnp_prof;
inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

// Pinned, device-visible host memory for what the host-returning entry points move across the bus per call (sampled
// correspondences in, counts / sums / masks out): copies to and from pageable memory go through the runtime's staging
// path (~20 us each), pinned ones are plain DMA, and the sweeps store their 28 sums straight into it.  One buffer per
// host thread, grown on demand, released at thread exit — the only memory this library allocates itself.
struct HostMailbox {
    void* p = nullptr;
    size_t cap = 0;
    void* get(size_t bytes) {
        if (bytes > cap) {
            if (p) (void)hipHostFree(p);
            p = nullptr;
            cap = 0;
            const size_t want = sfm::align_up(bytes + (bytes >> 1), 4096);
            if (hipHostMalloc(&p, want, hipHostMallocMapped | hipHostMallocPortable | hipHostMallocCoherent) != hipSuccess) {   // coherent = fine-grained: the sweep server polls it
                p = nullptr;
                return nullptr;
            }
            cap = want;
            std::memset(p, 0, want);
        }
        return p;
    }
    ~HostMailbox() {
        if (p) (void)hipHostFree(p);
    }
};
thread_local HostMailbox g_mailbox;


// Host side of pnp_server_kernel (see there).  One per sfm_solve_pnp_ransac call, on the caller's stream; the destructor posts "quit"
// on every exit path, so a server never outlives its call.
std::atomic<long long> g_mailbox_polls{0};
bool g_no_sweep_server = false;                     // sfm_debug_pnp_sweep_server(0): the launch-per-step path (A/B, tests)
struct PnpServer {
    PnpMailbox* mb;
    const float *X, *uv;
    int n;
    float thr2;
    float *hX, *huv;                                // mailbox: the correspondences as the host reads them
    int32_t* hinl;                                  // mailbox: the inlier list the host posts
    int32_t* inliers_dev;
    hipStream_t stream;
    double* part;                                   // device: [<= 128 waves][28] wave sums, then the arrival counter
    uint32_t seq = 0;
    int m_sel = -1;                                 // the selection (hinl[0 .. m_sel)) the sweeps run on
    bool send_sel = false;                          // ... and whether the next sweep request has to carry it (new selection, or a restarted server)
    bool running = false;
    ~PnpServer() { (void)stop(); }
    int start() {
        // request numbers continue across calls (the mailbox is per host thread): a request can never be mistaken for an old one
        seq = (uint32_t)((__atomic_load_n(&mb->request, __ATOMIC_RELAXED) >> 40) & kSeqMask);
        __atomic_store_n(&mb->request, (unsigned long long)seq << 40, __ATOMIC_RELAXED);
        __atomic_store_n(&mb->done_seq, seq, __ATOMIC_RELEASE);
        unsigned int* arrive = reinterpret_cast<unsigned int*>(part + (size_t)kSweepAcc * kServerMaxWaves);
        const uint32_t first = (seq + 1) & (uint32_t)kSeqMask;
        if (n <= 1024) {
            hipLaunchKernelGGL(pnp_server_kernel<1024>, dim3(1), dim3(1024), 0, stream, mb, X, uv, n, thr2, hX, huv, hinl, inliers_dev, first, part, arrive);
        } else {
            SFM_CHECK_HIP(hipMemsetAsync(arrive, 0, sizeof(unsigned int), stream));
            hipLaunchKernelGGL(pnp_server_kernel<256>, dim3((n + 255) / 256), dim3(256), 0, stream, mb, X, uv, n, thr2, hX, huv, hinl, inliers_dev, first, part, arrive);
        }
        SFM_CHECK_LAUNCH();
        running = true;
        send_sel = m_sel >= 0;
        return SFM_OK;
    }
    // wait until the server has answered request `want`; 0 = answered, 1 = the server left (timeout on its side), < 0 = error
    int wait(uint32_t want) {
        const double t0 = now_us();
        for (long long spins = 1;; ++spins) {
            const uint32_t d = __atomic_load_n(&mb->done_seq, __ATOMIC_ACQUIRE);
            if (d == want) { g_mailbox_polls.fetch_add(spins, std::memory_order_relaxed); return 0; }
            if (d == (uint32_t)kServerLeft) return 1;
            __builtin_ia32_pause();
            if ((spins & 0xFFF) == 0) {
                if (hipStreamQuery(stream) == hipSuccess && __atomic_load_n(&mb->done_seq, __ATOMIC_ACQUIRE) != want) return 1;   // the kernel is gone
                if (now_us() - t0 > 5e6) {
                    sfm::set_error("sfm_solve_pnp_ransac: the PnP server did not answer within 5 s");
                    return SFM_ERR_DEVICE;
                }
            }
        }
    }
    void post(uint32_t cmd, uint32_t arg) {
        seq = (seq + 1) & (uint32_t)kSeqMask;
        __atomic_store_n(&mb->request, (unsigned long long)seq << 40 | (unsigned long long)cmd << 32 | arg, __ATOMIC_RELEASE);
    }
    // Post a request whose payload is already in the mailbox and wait for the answer.  EVERY request is answered before the next is
    // posted (the mailbox holds one).  A server that left is restarted and the request repeated (all of them are idempotent).
    int request(uint32_t cmd, uint32_t arg = 0) {
        for (int attempt = 0; attempt < 3; ++attempt) {
            post(cmd, cmd == kCmdSweep && send_sel ? (uint32_t)m_sel + 1u : arg);
            const int r = wait(seq);
            if (r <= 0) {
                if (r == 0 && cmd == kCmdSweep) send_sel = false;
                return r < 0 ? r : SFM_OK;
            }
            running = false;
            SFM_CHECK_HIP(sfm::stream_sync(stream));
            const int rs = start();
            if (rs != SFM_OK) return rs;
        }
        sfm::set_error("sfm_solve_pnp_ransac: the PnP server keeps leaving");
        return SFM_ERR_DEVICE;
    }
    void select(int m) {                            // (hinl[0 .. m) holds the list: it travels with the next sweep request)
        m_sel = m;
        send_sel = true;
    }
    int stop() {
        if (!running) return SFM_OK;
        running = false;
        post(kCmdQuit, 0);
        const int r = wait(seq);
        return r < 0 ? r : SFM_OK;
    }
};

size_t essential_ws(int64_t n) {
    const size_t hmax = 64 * 10;
    return sfm::align_up(sizeof(double) * 4 * (size_t)n, 256) + sfm::align_up(sizeof(double) * 9 * hmax, 256) +
           sfm::align_up(sizeof(int32_t) * hmax, 256) + sfm::align_up(hmax * (size_t)n, 256) + 1024;
}

}  // namespace

// ---------------------------------------------------------------------------------------------- findEssentialMat
extern "C" size_t sfm_find_essential_mat_ws_bytes(int64_t n) { return n < 0 ? 0 : essential_ws(n); }

extern "C" int sfm_find_essential_mat(const float* pts0_dev, const float* pts1_dev, int64_t n, const double* K, double prob,
                                      double threshold, int max_iters, double* E_host, int32_t* info_host, uint8_t* mask_dev,
                                      void* ws, size_t ws_bytes, void* stream_) {
    SFM_CHECK_ARG(K && E_host && info_host, "sfm_find_essential_mat: null pointer");
    SFM_CHECK_ARG(n >= 0 && n < (int64_t)1 << 30, "sfm_find_essential_mat: bad n");
    info_host[0] = info_host[1] = info_host[2] = info_host[3] = 0;      // models in E, inliers, iterations run, models scored
    if (n < 5) return SFM_OK;                                           // OpenCV: count < modelPoints -> no model
    SFM_CHECK_ARG(pts0_dev && pts1_dev && mask_dev, "sfm_find_essential_mat: null pointer");
    if (!ws || ws_bytes < essential_ws(n)) {
        sfm::set_error("sfm_find_essential_mat: workspace too small (%zu < %zu)", ws_bytes, essential_ws(n));
        return SFM_ERR_WORKSPACE;
    }
    hipStream_t stream = sfm::as_stream(stream_);
    sfm::Carver c(reinterpret_cast<void*>(sfm::align_up((size_t)(uintptr_t)ws, 256)));
    double* x0n = c.take<double>(2 * (size_t)n);
    double* x1n = c.take<double>(2 * (size_t)n);
    const size_t hmax = 640;
    double* models_dev = c.take<double>(9 * hmax);
    int32_t* counts_dev = c.take<int32_t>(hmax);
    uint8_t* masks_dev = c.take<uint8_t>(hmax * (size_t)n);

    const KNorm kn(K);
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(k_normalise_kernel, dim3(nb), dim3(256), 0, stream, pts0_dev, n, kn, x0n);
    hipLaunchKernelGGL(k_normalise_kernel, dim3(nb), dim3(256), 0, stream, pts1_dev, n, kn, x1n);
    SFM_CHECK_LAUNCH();
    std::vector<float> h0(2 * (size_t)n), h1(2 * (size_t)n);
    SFM_CHECK_HIP(hipMemcpyAsync(h0.data(), pts0_dev, sizeof(float) * 2 * (size_t)n, hipMemcpyDeviceToHost, stream));
    SFM_CHECK_HIP(hipMemcpyAsync(h1.data(), pts1_dev, sizeof(float) * 2 * (size_t)n, hipMemcpyDeviceToHost, stream));
    SFM_CHECK_HIP(sfm::stream_sync(stream));
    auto sample = [&](const int* idx, double* s0, double* s1) {
        for (int k = 0; k < 5; ++k) {
            s0[2 * k] = (double)h0[2 * idx[k]] * kn.ifx + kn.bx; s0[2 * k + 1] = (double)h0[2 * idx[k] + 1] * kn.ify + kn.by;
            s1[2 * k] = (double)h1[2 * idx[k]] * kn.ifx + kn.bx; s1[2 * k + 1] = (double)h1[2 * idx[k] + 1] * kn.ify + kn.by;
        }
    };
    if (n == 5) {                     // count == modelPoints: the solver's models as they are (stacked), every point an inlier
        const int idx[5] = {0, 1, 2, 3, 4};
        double s0[10], s1[10];
        sample(idx, s0, s1);
        const int k = hs::five_point(s0, s1, E_host);
        info_host[0] = k;
        info_host[1] = k > 0 ? 5 : 0;
        if (k > 0) SFM_CHECK_HIP(hipMemsetAsync(mask_dev, 1, 5, stream));
        return SFM_OK;
    }
    double thr = threshold;
    thr /= (K[0] + K[4]) / 2;
    const float thr2 = (float)(thr * thr);
    hs::CvRng rng;
    RansacState st;
    st.niters = std::max(max_iters, 1);
    st.model_points = 5;
    st.confidence = prob;
    st.count = n;
    Chunker ch;
    std::vector<double> models;
    std::vector<int> owner;
    std::vector<int32_t> counts;
    int it = 0;
    while (it < st.niters) {
        const int m = ch.next(st.niters - it);
        models.clear();
        owner.clear();
        for (int k = 0; k < m; ++k) {
            int idx[5];
            double s0[10], s1[10], E10[90];
            rng.subset((int)n, 5, idx);
            sample(idx, s0, s1);
            const int nm = hs::five_point(s0, s1, E10);
            for (int q = 0; q < nm; ++q) {
                models.insert(models.end(), E10 + 9 * q, E10 + 9 * q + 9);
                owner.push_back(k);
            }
        }
        const int H = (int)owner.size();
        if (H > 0) {
            counts.resize(H);
            SFM_CHECK_HIP(hipMemcpyAsync(models_dev, models.data(), sizeof(double) * 9 * (size_t)H, hipMemcpyHostToDevice, stream));
            const int rc = sfm_score_essential(models_dev, H, x0n, x1n, n, thr2, counts_dev, masks_dev, stream_);
            if (rc != SFM_OK) return rc;
            SFM_CHECK_HIP(hipMemcpyAsync(counts.data(), counts_dev, sizeof(int32_t) * (size_t)H, hipMemcpyDeviceToHost, stream));
            SFM_CHECK_HIP(sfm::stream_sync(stream));
            bool stop;
            const int bj = replay_chunk(st, it, owner, counts.data(), stop);
            if (bj >= 0) {
                std::memcpy(E_host, models.data() + 9 * (size_t)bj, 9 * sizeof(double));
                SFM_CHECK_HIP(hipMemcpyAsync(mask_dev, masks_dev + (size_t)bj * (size_t)n, (size_t)n, hipMemcpyDeviceToDevice, stream));
            }
            if (stop) {
                it = st.niters;
                break;
            }
        }
        it += m;
    }
    info_host[0] = st.best > 0 ? 1 : 0;
    info_host[1] = st.best;
    info_host[2] = st.iterations_run();
    info_host[3] = st.models_scored;
    return SFM_OK;
}

// ---------------------------------------------------------------------------------------------- recoverPose
extern "C" size_t sfm_recover_pose_ws_bytes(int64_t n) {
    return n < 0 ? 0 : sfm::align_up(sizeof(double) * 4 * (size_t)n, 256) + sfm::align_up(4 * (size_t)n, 256) + 1024;
}

extern "C" int sfm_recover_pose(const double* E, const float* pts0_dev, const float* pts1_dev, int64_t n, const double* K,
                                double distance_thresh, int rows, double* R_host, double* t_host, int32_t* good_host,
                                uint8_t* mask_dev, void* ws, size_t ws_bytes, void* stream_) {
    SFM_CHECK_ARG(E && K && R_host && t_host && good_host, "sfm_recover_pose: null pointer");
    SFM_CHECK_ARG(n >= 0 && (rows == 4 || rows == 6), "sfm_recover_pose: bad n / rows");
    SFM_CHECK_ARG(n == 0 || (pts0_dev && pts1_dev), "sfm_recover_pose: null points");
    if (!ws || ws_bytes < sfm_recover_pose_ws_bytes(n)) {
        sfm::set_error("sfm_recover_pose: workspace too small (%zu < %zu)", ws_bytes, sfm_recover_pose_ws_bytes(n));
        return SFM_ERR_WORKSPACE;
    }
    hipStream_t stream = sfm::as_stream(stream_);
    double R1[9], R2[9], tt[3], Ps[48];
    hs::decompose_essential(E, R1, R2, tt);
    const double* Rs[4] = {R1, R2, R1, R2};
    const double sg[4] = {1, 1, -1, -1};
    for (int cnd = 0; cnd < 4; ++cnd)
        for (int i = 0; i < 3; ++i) {
            for (int j = 0; j < 3; ++j) Ps[12 * cnd + 4 * i + j] = Rs[cnd][3 * i + j];
            Ps[12 * cnd + 4 * i + 3] = sg[cnd] * tt[i];
        }
    int32_t g[4] = {0, 0, 0, 0};
    uint8_t* masks_dev = nullptr;
    if (n > 0) {
        sfm::Carver c(reinterpret_cast<void*>(sfm::align_up((size_t)(uintptr_t)ws, 256)));
        double* x0n = c.take<double>(2 * (size_t)n);
        double* x1n = c.take<double>(2 * (size_t)n);
        masks_dev = c.take<uint8_t>(4 * (size_t)n);
        int32_t* counts_dev = c.take<int32_t>(4);
        const KNorm kn(K);
        const unsigned nb = (unsigned)((n + 255) / 256);
        hipLaunchKernelGGL(k_normalise_kernel, dim3(nb), dim3(256), 0, stream, pts0_dev, n, kn, x0n);
        hipLaunchKernelGGL(k_normalise_kernel, dim3(nb), dim3(256), 0, stream, pts1_dev, n, kn, x1n);
        SFM_CHECK_LAUNCH();
        const int rc = sfm_recover_pose_score(Ps, 4, x0n, x1n, n, distance_thresh, rows, counts_dev, masks_dev, stream_);
        if (rc != SFM_OK) return rc;
        SFM_CHECK_HIP(hipMemcpyAsync(g, counts_dev, sizeof(g), hipMemcpyDeviceToHost, stream));
        SFM_CHECK_HIP(sfm::stream_sync(stream));
    }
    int k;      // OpenCV's cascade of >= tests in candidate order
    if (g[0] >= g[1] && g[0] >= g[2] && g[0] >= g[3]) k = 0;
    else if (g[1] >= g[0] && g[1] >= g[2] && g[1] >= g[3]) k = 1;
    else if (g[2] >= g[0] && g[2] >= g[1] && g[2] >= g[3]) k = 2;
    else k = 3;
    std::memcpy(R_host, Rs[k], 9 * sizeof(double));
    for (int i = 0; i < 3; ++i) t_host[i] = sg[k] * tt[i];
    *good_host = g[k];
    if (mask_dev && n > 0) SFM_CHECK_HIP(hipMemcpyAsync(mask_dev, masks_dev + (size_t)k * (size_t)n, (size_t)n, hipMemcpyDeviceToDevice, stream));
    return SFM_OK;
}

// ---------------------------------------------------------------------------------------------- solvePnPRansac
extern "C" size_t sfm_solve_pnp_ransac_ws_bytes(int64_t n) {
    if (n < 0) return 0;
    const size_t hmax = 64;
    return sfm::align_up(sizeof(double) * 6 * hmax, 256) + sfm::align_up(sizeof(int32_t) * hmax, 256) + sfm::align_up(hmax * (size_t)n, 256) +
           sfm::align_up((size_t)n, 256) + sfm::align_up(sizeof(double) * kSweepAcc * (kSweepMaxBlocks + 1), 256) +
           sfm::align_up(sizeof(double) * kSweepAcc * kServerMaxWaves + 64, 256) + 1024;
}

extern "C" int sfm_solve_pnp_ransac(const float* X_dev, const float* uv_dev, int64_t n, const double* K, int iterations,
                                    float reproj_error, double confidence, double* rvec_host, double* tvec_host,
                                    int32_t* info_host, int32_t* inliers_dev, void* ws, size_t ws_bytes, void* stream_) {
    SFM_CHECK_ARG(K && rvec_host && tvec_host && info_host, "sfm_solve_pnp_ransac: null pointer");
    info_host[0] = info_host[1] = info_host[2] = info_host[3] = 0;      // ok, inliers, init status, LM iterations
    SFM_CHECK_ARG(n >= 4 && n < (int64_t)1 << 30, "sfm_solve_pnp_ransac: at least 4 correspondences are required (OpenCV asserts npoints >= 4; got %lld)",
                  (long long)n);
    SFM_CHECK_ARG(X_dev && uv_dev && inliers_dev, "sfm_solve_pnp_ransac: null pointer");
    if (!ws || ws_bytes < sfm_solve_pnp_ransac_ws_bytes(n)) {
        sfm::set_error("sfm_solve_pnp_ransac: workspace too small (%zu < %zu)", ws_bytes, sfm_solve_pnp_ransac_ws_bytes(n));
        return SFM_ERR_WORKSPACE;
    }
    hipStream_t stream = sfm::as_stream(stream_);
    sfm::Carver c(reinterpret_cast<void*>(sfm::align_up((size_t)(uintptr_t)ws, 256)));
    const size_t hmax = 64;
    (void)c.take<double>(6 * hmax);                      // (models are read from the pinned mailbox)
    int32_t* counts_dev = c.take<int32_t>(hmax);
    uint8_t* masks_dev = c.take<uint8_t>(hmax * (size_t)n);
    uint8_t* best_dev = c.take<uint8_t>((size_t)n);
    double* sweep_dev = c.take<double>((size_t)kSweepAcc * (kSweepMaxBlocks + 1));
    double* server_part = c.take<double>((size_t)kSweepAcc * kServerMaxWaves + 8);      // the PnP server's wave sums, then its arrival counter

    double tp = now_us();
    auto lap = [&](int k) { const double t1 = now_us(); g_pnp_prof.t[k] += t1 - tp; tp = t1; };
    ++g_pnp_prof.calls;
    // mailbox layout: X [3n] f32 | uv [2n] f32 | mask [n] u8 | poses [6 hmax] f64 | counts [hmax] i32 | sums [28] f64 |
    //                 chunk masks [kSmallMasks n] u8 (small chunks: their masks ride along with the counts, so the
    //                 winner's mask is already on the host when RANSAC stops — one synchronisation less per call)
    constexpr int kSmallMasks = 4;
    const size_t o_uv = sizeof(float) * 3 * (size_t)n, o_mask = o_uv + sizeof(float) * 2 * (size_t)n;
    const size_t o_pose = sfm::align_up(o_mask + (size_t)n, 64), o_cnt = o_pose + sizeof(double) * 6 * hmax;
    const size_t o_sum = sfm::align_up(o_cnt + sizeof(int32_t) * hmax, 64);
    const size_t o_lm = sfm::align_up(o_sum + sizeof(double) * kSweepAcc, 64);
    const size_t o_inl = o_lm + sizeof(PnpMailbox);                        // served calls: the inlier list the host posts
    const size_t o_cm = sfm::align_up(o_inl + sizeof(int32_t) * (size_t)n, 64);
    const bool ride_along = (size_t)n * kSmallMasks <= (1u << 18);
    char* mb = static_cast<char*>(g_mailbox.get(o_cm + (ride_along ? (size_t)n * kSmallMasks : 0)));
    if (!mb) {
        sfm::set_error("sfm_solve_pnp_ransac: hipHostMalloc failed");
        return SFM_ERR_DEVICE;
    }
    float* hX = reinterpret_cast<float*>(mb);
    float* huv = reinterpret_cast<float*>(mb + o_uv);
    uint8_t* hmask = reinterpret_cast<uint8_t*>(mb + o_mask);
    double* hposes = reinterpret_cast<double*>(mb + o_pose);
    int32_t* hcounts = reinterpret_cast<int32_t*>(mb + o_cnt);
    double* sums = reinterpret_cast<double*>(mb + o_sum);
    uint8_t* hchunk = reinterpret_cast<uint8_t*>(mb + o_cm);
    bool host_mask_valid = false;
    const float thr2 = (float)((double)reproj_error * (double)reproj_error);
    // 6 .. 8 192 correspondences: the whole call is served by ONE resident launch (pnp_server_kernel) through the mailbox — no copy,
    // no further launch, no stream synchronisation; otherwise a copy / launch / synchronisation per step, as in round 5
    PnpMailbox* const pmb = reinterpret_cast<PnpMailbox*>(mb + o_lm);
    int32_t* const hinl = reinterpret_cast<int32_t*>(mb + o_inl);
    const bool served = n >= 6 && n <= kServerMaxN && !g_no_sweep_server;
    PnpServer server{pmb, X_dev, uv_dev, (int)n, thr2, hX, huv, hinl, inliers_dev, stream, server_part};
    if (served) {
        int rc0 = server.start();
        if (rc0 == SFM_OK) rc0 = server.request(kCmdCopyIn, 0);
        if (rc0 != SFM_OK) return rc0;
    } else {
        SFM_CHECK_HIP(hipMemcpyAsync(hX, X_dev, sizeof(float) * 3 * (size_t)n, hipMemcpyDeviceToHost, stream));
        SFM_CHECK_HIP(hipMemcpyAsync(huv, uv_dev, sizeof(float) * 2 * (size_t)n, hipMemcpyDeviceToHost, stream));
        SFM_CHECK_HIP(sfm::stream_sync(stream));
    }
    lap(0);
    const double ifx = 1. / K[0], ify = 1. / K[4];
    // solvePnP(EPNP) on a sample: undistortPoints writes float32 normalised coordinates (the image points' type) and
    // epnp::init_points maps them back with u = x fu + uc
    // (reject_nonfinite: a RANSAC hypothesis with a NaN scores zero inliers in OpenCV and is dropped here; the plain
    // solvePnP(EPNP) of the five-point case returns whatever the solver produced)
    auto epnp_model = [&](const int* idx, double* model, bool reject_nonfinite = true) -> bool {
        double Xs[15], us[10], R[9], t[3];
        for (int k = 0; k < 5; ++k) {
            for (int j = 0; j < 3; ++j) Xs[3 * k + j] = (double)hX[3 * (size_t)idx[k] + j];
            us[2 * k] = (double)(float)(((double)huv[2 * (size_t)idx[k]] - K[2]) * ifx) * K[0] + K[2];
            us[2 * k + 1] = (double)(float)(((double)huv[2 * (size_t)idx[k] + 1] - K[5]) * ify) * K[4] + K[5];
        }
        hs::Epnp(K, Xs, us, 5).compute_pose(R, t);
        if (reject_nonfinite) {
            for (int k = 0; k < 9; ++k)
                if (!std::isfinite(R[k])) return false;
            for (int k = 0; k < 3; ++k)
                if (!std::isfinite(t[k])) return false;
        }
        hs::rodrigues_mat2vec(R, model);
        std::memcpy(model + 3, t, sizeof(t));
        return true;
    };
    if (n == 4) {                     // OpenCV switches to model_points = 4 / SOLVEPNP_P3P: plain solvePnP(P3P), every point an inlier
        double Xs[12], us[8], R[9], t[3], model[6];
        for (int k = 0; k < 4; ++k) {
            for (int j = 0; j < 3; ++j) Xs[3 * k + j] = (double)hX[3 * k + j];
            us[2 * k] = (double)(float)(((double)huv[2 * k] - K[2]) * ifx) * K[0] + K[2];
            us[2 * k + 1] = (double)(float)(((double)huv[2 * k + 1] - K[5]) * ify) * K[4] + K[5];
        }
        if (!hs::p3p(K, Xs, us, R, t)) return SFM_OK;
        hs::rodrigues_mat2vec(R, model);
        std::memcpy(rvec_host, model, 24);
        std::memcpy(tvec_host, t, 24);
        const int32_t all[4] = {0, 1, 2, 3};
        SFM_CHECK_HIP(hipMemcpyAsync(inliers_dev, all, sizeof(all), hipMemcpyHostToDevice, stream));
        SFM_CHECK_HIP(sfm::stream_sync(stream));
        info_host[0] = 1;
        info_host[1] = 4;
        return SFM_OK;
    }
    if (n == 5) {                     // model_points == npoints: plain solvePnP(EPNP), every point an inlier
        const int idx[5] = {0, 1, 2, 3, 4};
        double model[6];
        epnp_model(idx, model, false);
        std::memcpy(rvec_host, model, 24);
        std::memcpy(tvec_host, model + 3, 24);
        const int32_t all[5] = {0, 1, 2, 3, 4};
        SFM_CHECK_HIP(hipMemcpyAsync(inliers_dev, all, sizeof(all), hipMemcpyHostToDevice, stream));
        SFM_CHECK_HIP(sfm::stream_sync(stream));
        info_host[0] = 1;
        info_host[1] = 5;
        return SFM_OK;
    }
    hs::CvRng rng;
    RansacState st;
    st.niters = std::max(iterations, 1);
    st.model_points = 5;
    st.confidence = confidence;
    st.count = n;
    Chunker ch(2);      // every EPnP solve is 18 us of host time: a clean scene (inlier ratio > 0.95) is done after 1-2
                        // iterations; 10 % outliers need log(0.01) / log(1 - 0.9^5) ~ 6 = chunks of 2 + 4
    std::vector<int> owner;
    double best_model[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long best_bits[kServerMaxWaves];            // served calls: the best model's inlier masks
    int it = 0;
    while (it < st.niters) {
        const int m = ch.next(st.niters - it);
        owner.clear();
        for (int k = 0; k < m; ++k) {
            int idx[5];
            double model[6];
            rng.subset((int)n, 5, idx);
            if (!epnp_model(idx, model)) continue;
            std::memcpy(hposes + 6 * owner.size(), model, sizeof(model));       // the scoring kernel reads the models in place
            owner.push_back(k);
        }
        const int H = (int)owner.size();
        lap(1);
        ++g_pnp_prof.chunks;
        if (H > 0 && served) {
            // R, t of every model on the host (cv::Rodrigues as the error function of OpenCV's RANSAC evaluates it, on the CPU), one
            // request, one 64-bit inlier mask per wave and model back; the counts are the masks' bit counts
            for (int j = 0; j < H; ++j) {
                double R[9], J[27];
                rodrigues_with_jac(hposes + 6 * (size_t)j, R, J);
                std::memcpy(pmb->models + 12 * (size_t)j, R, sizeof(R));
                std::memcpy(pmb->models + 12 * (size_t)j + 9, hposes + 6 * (size_t)j + 3, 3 * sizeof(double));
            }
            pmb->cam.fx = K[0]; pmb->cam.fy = K[4]; pmb->cam.cx = K[2]; pmb->cam.cy = K[5];
            const int rc = server.request(kCmdScore, (uint32_t)H);
            if (rc != SFM_OK) return rc;
            const int nw = (int)((n + 63) >> 6);
            for (int j = 0; j < H; ++j) {
                int cnt = 0;
                for (int w = 0; w < nw; ++w) cnt += __builtin_popcountll(pmb->bits[(size_t)j * kServerMaxWaves + w]);
                hcounts[j] = cnt;
            }
            bool stop;
            const int bj = replay_chunk(st, it, owner, hcounts, stop);
            if (bj >= 0) {
                std::memcpy(best_model, hposes + 6 * (size_t)bj, sizeof(best_model));
                std::memcpy(best_bits, pmb->bits + (size_t)bj * kServerMaxWaves, sizeof(unsigned long long) * (size_t)nw);
            }
            lap(2);
            if (stop) break;
        } else if (H > 0) {
            const int rc = sfm_score_pnp(hposes, H, K, X_dev, uv_dev, n, thr2, counts_dev, masks_dev, stream_);
            if (rc != SFM_OK) return rc;
            SFM_CHECK_HIP(hipMemcpyAsync(hcounts, counts_dev, sizeof(int32_t) * (size_t)H, hipMemcpyDeviceToHost, stream));
            const bool with_masks = ride_along && H <= kSmallMasks;
            if (with_masks) SFM_CHECK_HIP(hipMemcpyAsync(hchunk, masks_dev, (size_t)H * (size_t)n, hipMemcpyDeviceToHost, stream));
            SFM_CHECK_HIP(sfm::stream_sync(stream));
            bool stop;
            const int bj = replay_chunk(st, it, owner, hcounts, stop);
            if (bj >= 0) {
                std::memcpy(best_model, hposes + 6 * (size_t)bj, sizeof(best_model));
                SFM_CHECK_HIP(hipMemcpyAsync(best_dev, masks_dev + (size_t)bj * (size_t)n, (size_t)n, hipMemcpyDeviceToDevice, stream));
                host_mask_valid = with_masks;
                if (with_masks) std::memcpy(hmask, hchunk + (size_t)bj * (size_t)n, (size_t)n);
            }
            lap(2);
            if (stop) break;
        }
        it += m;
    }
    if (st.best <= 0) return SFM_OK;
    // inlier list (ascending, as OpenCV pushes them) — the one piece of the mask the host needs
    std::vector<int32_t> inl;
    inl.reserve((size_t)st.best);
    if (served) {
        for (int w = 0; w < (int)((n + 63) >> 6); ++w)
            for (unsigned long long b = best_bits[w]; b; b &= b - 1) inl.push_back((int32_t)(64 * w + __builtin_ctzll(b)));
    } else {
        if (!host_mask_valid) {
            SFM_CHECK_HIP(hipMemcpyAsync(hmask, best_dev, (size_t)n, hipMemcpyDeviceToHost, stream));
            SFM_CHECK_HIP(sfm::stream_sync(stream));
        }
        for (int64_t i = 0; i < n; ++i)
            if (hmask[(size_t)i]) inl.push_back((int32_t)i);
    }
    const int64_t m_in = (int64_t)inl.size();
    if (served) {                                             // the server takes the list as its selection and writes `inliers_dev`
        std::memcpy(hinl, inl.data(), sizeof(int32_t) * (size_t)m_in);
        server.select((int)m_in);
    } else {
        SFM_CHECK_HIP(hipMemcpyAsync(inliers_dev, inl.data(), sizeof(int32_t) * (size_t)m_in, hipMemcpyHostToDevice, stream));
    }
    lap(3);
    // solvePnP(ITERATIVE) on the inliers: DLT initialisation on the host, Levenberg-Marquardt with the sweeps on the device
    const int blocks = (int)std::min<int64_t>((m_in + kSweepThreads - 1) / kSweepThreads, kSweepMaxBlocks);
    double param[6];
    const int init_status = hs::pnp_dlt_init<float>(hX, huv, inl.data(), m_in, K, param, param + 3);
    if (init_status != 0) std::memcpy(param, best_model, sizeof(param));      // planar / < 6 inliers: refine the RANSAC model
    lap(4);
    auto sweep = [&](const double* p, bool jac) -> int {
        lap(6);
        ++g_pnp_prof.sweeps;
        PnpCam cam;
        rodrigues_with_jac(p, cam.R, cam.dR);
        cam.t[0] = p[3]; cam.t[1] = p[4]; cam.t[2] = p[5];
        cam.fx = K[0]; cam.fy = K[4]; cam.cx = K[2]; cam.cy = K[5];
        if (served) {
            std::memcpy(&pmb->cam, &cam, sizeof(cam));
            const int rcs = server.request(kCmdSweep, 0);
            if (rcs == SFM_OK) std::memcpy(sums, pmb->sums, sizeof(double) * kSweepAcc);
            lap(5);
            return rcs;
        }
        double* out = blocks == 1 ? sums : sweep_dev;           // one workgroup: its 28 sums go straight to the pinned mailbox
        if (jac)
            hipLaunchKernelGGL(pnp_sweep_kernel<1>, dim3(blocks), dim3(kSweepThreads), 0, stream, cam, X_dev, uv_dev, inliers_dev, m_in, out);
        else
            hipLaunchKernelGGL(pnp_sweep_kernel<0>, dim3(blocks), dim3(kSweepThreads), 0, stream, cam, X_dev, uv_dev, inliers_dev, m_in, out);
        if (blocks > 1)
            hipLaunchKernelGGL(pnp_sweep_fold_kernel, dim3(1), dim3(64), 0, stream, sweep_dev, blocks, sums);
        SFM_CHECK_LAUNCH();
        SFM_CHECK_HIP(sfm::stream_sync(stream));
        lap(5);
        return SFM_OK;
    };
    // CvLevMarq (J / err interface) as cvFindExtrinsicCameraParams2 drives it: <= 20 iterations, epsilon FLT_EPSILON,
    // lambda = exp(k log 10) from k = -3, the DIAGONAL of J^T J scaled by 1 + lambda, SVD solve
    const double LOG10 = std::log(10.);
    int lambdaLg10 = -3, iters = 0;
    double prev[6], JtJ[36], JtErr[6], prevErrNorm = DBL_MAX, errNorm = 0;
    auto step = [&]() {
        const double lambda = std::exp(lambdaLg10 * LOG10);
        double A[36], dx[6];
        std::memcpy(A, JtJ, sizeof(A));
        for (int i = 0; i < 6; ++i) A[7 * i] *= 1. + lambda;
        hs::Svd sv;
        sv.compute(A, 6, 6);
        sv.back_subst(JtErr, dx);
        for (int i = 0; i < 6; ++i) param[i] = prev[i] - dx[i];
    };
    // Every sweep returns J^T J, J^T e AND |e|^2 at its parameters: the error check of a trial step and the normal
    // equations of the next iteration (which OpenCV evaluates at the same, accepted, parameters) are one launch and one
    // download instead of two — same sums, half the host round trips.
    int rc = sweep(param, true);
    if (rc != SFM_OK) return rc;
    for (;;) {
        int q = 0;
        for (int a = 0; a < 6; ++a)
            for (int b = a; b < 6; ++b) JtJ[6 * a + b] = JtJ[6 * b + a] = sums[q++];
        for (int a = 0; a < 6; ++a) JtErr[a] = sums[21 + a];
        std::memcpy(prev, param, sizeof(prev));
        if (iters == 0) prevErrNorm = std::sqrt(sums[27]);
        step();
        if ((rc = sweep(param, true)) != SFM_OK) return rc;
        for (;;) {
            errNorm = std::sqrt(sums[27]);
            if (errNorm > prevErrNorm && ++lambdaLg10 <= 16) {
                step();
                if ((rc = sweep(param, true)) != SFM_OK) return rc;
                continue;
            }
            break;
        }
        lambdaLg10 = std::max(lambdaLg10 - 1, -16);
        double dn = 0, pn = 0;
        for (int i = 0; i < 6; ++i) {
            dn += (param[i] - prev[i]) * (param[i] - prev[i]);
            pn += prev[i] * prev[i];
        }
        ++iters;
        if (iters >= 20 || std::sqrt(dn) / std::sqrt(pn) < FLT_EPSILON) break;
        prevErrNorm = errNorm;
    }
    lap(6);
    if (served && (rc = server.stop()) != SFM_OK) return rc;
    std::memcpy(rvec_host, param, 24);
    std::memcpy(tvec_host, param + 3, 24);
    info_host[0] = 1;
    info_host[1] = (int32_t)m_in;
    info_host[2] = init_status;
    info_host[3] = iters;
    return SFM_OK;
}

// ---------------------------------------------------------------------------------------------- host solver exports
// (the generators above, reachable on their own: unit-tested against the oracle on the CPU)
extern "C" int sfm_host_epnp(const double* K, const double* Xw, const double* uv, int n, double* R_out, double* t_out) {
    SFM_CHECK_ARG(K && Xw && uv && R_out && t_out, "sfm_host_epnp: null pointer");
    SFM_CHECK_ARG(n >= 4 && n <= hs::kEpnpMaxPts, "sfm_host_epnp: n must be in [4, %d] (got %d)", hs::kEpnpMaxPts, n);
    hs::Epnp(K, Xw, uv, n).compute_pose(R_out, t_out);
    return SFM_OK;
}

extern "C" int sfm_host_p3p(const double* K, const double* Xw, const double* uv, double* R_out, double* t_out, int32_t* ok_out) {
    SFM_CHECK_ARG(K && Xw && uv && R_out && t_out && ok_out, "sfm_host_p3p: null pointer");
    *ok_out = hs::p3p(K, Xw, uv, R_out, t_out) ? 1 : 0;
    return SFM_OK;
}

extern "C" int sfm_host_five_point(const double* x1n, const double* x2n, double* E_out, int32_t* count_out) {
    SFM_CHECK_ARG(x1n && x2n && E_out && count_out, "sfm_host_five_point: null pointer");
    *count_out = hs::five_point(x1n, x2n, E_out);
    return SFM_OK;
}

extern "C" int sfm_host_decompose_essential(const double* E, double* R1, double* R2, double* t) {
    SFM_CHECK_ARG(E && R1 && R2 && t, "sfm_host_decompose_essential: null pointer");
    hs::decompose_essential(E, R1, R2, t);
    return SFM_OK;
}

extern "C" int sfm_host_pnp_dlt_init(const double* K, const double* X, const double* uv, int64_t n, double* rvec, double* tvec,
                                     int32_t* status_out) {
    SFM_CHECK_ARG(K && X && uv && rvec && tvec && status_out && n >= 1, "sfm_host_pnp_dlt_init: bad argument");
    *status_out = hs::pnp_dlt_init<double>(X, uv, nullptr, n, K, rvec, tvec);
    return SFM_OK;
}

extern "C" int sfm_project_points_f64(const double* rvec, const double* tvec, const double* K, const double* X_dev, int64_t n,
                                      double* proj_dev, void* stream_) {
    SFM_CHECK_ARG(rvec && tvec && K && n >= 0, "sfm_project_points_f64: bad argument");
    if (n == 0) return SFM_OK;
    SFM_CHECK_ARG(X_dev && proj_dev, "sfm_project_points_f64: null pointer");
    PnpCam cam;
    rodrigues_with_jac(rvec, cam.R, cam.dR);
    cam.t[0] = tvec[0]; cam.t[1] = tvec[1]; cam.t[2] = tvec[2];
    cam.fx = K[0]; cam.fy = K[4]; cam.cx = K[2]; cam.cy = K[5];
    hipLaunchKernelGGL(project_f64_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, sfm::as_stream(stream_), cam, X_dev, n, proj_dev);
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}

extern "C" int sfm_host_rodrigues(const double* src, int src_is_matrix, double* dst, double* jac) {
    SFM_CHECK_ARG(src && dst, "sfm_host_rodrigues: null pointer");
    if (src_is_matrix) {
        hs::rodrigues_mat2vec(src, dst);
    } else {
        double J[27];
        rodrigues_with_jac(src, dst, jac ? jac : J);
    }
    return SFM_OK;
}

// Mailbox polls of the sweep server's host side since load (spins of the waiting loop: a measure of time, not of API calls), and the
// A/B switch between the server and the launch-per-sweep path (on = 1 default; returns the previous setting).
extern "C" int64_t sfm_host_poll_count(void) { return (int64_t)g_mailbox_polls.load(std::memory_order_relaxed); }
extern "C" int sfm_debug_pnp_sweep_server(int on) {
    const int was = g_no_sweep_server ? 0 : 1;
    g_no_sweep_server = on == 0;
    return was;
}

// Where sfm_solve_pnp_ransac's HOST time went since the last reset (accumulated over calls, microseconds):
// out[0] calls, out[1..7] = copy-in, EPnP hypotheses on the host, scoring + wait, mask / inlier bookkeeping, DLT initialisation,
// Levenberg-Marquardt sweeps (device + wait), Levenberg-Marquardt host algebra; out[8] hypothesis chunks, out[9] LM sweeps.
extern "C" int sfm_pnp_profile_read(double* out10, int reset) {
    SFM_CHECK_ARG(out10 != nullptr, "sfm_pnp_profile_read: null pointer");
    out10[0] = (double)g_pnp_prof.calls;
    for (int k = 0; k < 7; ++k) out10[1 + k] = g_pnp_prof.t[k];
    out10[8] = (double)g_pnp_prof.chunks;
    out10[9] = (double)g_pnp_prof.sweeps;
    if (reset) {
        for (double& v : g_pnp_prof.t) v = 0.0;
        g_pnp_prof.calls = g_pnp_prof.chunks = g_pnp_prof.sweeps = 0;
    }
    return SFM_OK;
}
