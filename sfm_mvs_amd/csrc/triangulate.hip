// cv2.triangulatePoints(P1, P2, points1, points2) + `cloud / cloud[3]`   (sfm.py:53-54)
//
// One lane per correspondence.  The 4x4 (or legacy 6x4) DLT system is built in fp64 registers and
// its smallest right singular vector is found by the same one-sided Jacobi sweep OpenCV's cv::SVD
// runs (pair order, skip threshold 10*DBL_EPSILON, <=30 sweeps, 2-lane dot/norm accumulation,
// selection sort) so that the float32 result rounds like the reference's.  Everything stays in
// VGPRs (A^T 4xROWS + V 4x4 doubles, all indices compile-time); lanes that converge early idle
// until their wave's slowest point is done.  fp64-VALU bound: ~1.5-3 kFLOP per point vs 32 B.
//
// north_star words this as "one-warp-per-point SVD"; a 4x4 problem has 6 column pairs per sweep —
// spreading it over 64 lanes would leave >90 % of the wave idle and add cross-lane traffic, so a
// lane owns a point and a wave solves 64 points in lockstep (see docs/geometry.md).
#include "common.h"
#include <algorithm>
#include <cfloat>
#include <cstdlib>

namespace {

struct ProjPair {
    double p[2][12];
};

__device__ __forceinline__ double svd_hypot(double a, double b) {
    a = fabs(a);
    b = fabs(b);
    if (a > b) {
        b /= a;
        return a * sqrt(1 + b * b);
    }
    if (b > 0) {
        a /= b;
        return b * sqrt(1 + a * a);
    }
    return 0;
}

// rotate pair (I,J): all indices are template constants so At/V/W stay in registers
template <int M, int I, int J>
__device__ __forceinline__ void jacobi_pair(double (&At)[4][M], double (&V)[4][4], double (&W)[4], bool& changed) {
    const double eps = DBL_EPSILON * 10;
    double a = W[I], b = W[J];
    double s0 = 0, s1 = 0;
#pragma unroll
    for (int k = 0; k < M; k += 2) {
        s0 = s0 + At[I][k] * At[J][k];
        s1 = s1 + At[I][k + 1] * At[J][k + 1];
    }
    double p = s0 + s1;
    if (fabs(p) <= eps * sqrt(a * b)) return;
    p *= 2;
    const double beta = a - b, gamma = svd_hypot(p, beta);
    double c, s;
    if (beta < 0) {
        const double delta = (gamma - beta) * 0.5;
        s = sqrt(delta / gamma);
        c = p / (gamma * s * 2);
    } else {
        c = sqrt((gamma + beta) / (gamma * 2));
        s = p / (gamma * c * 2);
    }
    double a0 = 0, a1 = 0, b0 = 0, b1 = 0;
#pragma unroll
    for (int k = 0; k < M; k += 2) {
        const double t0 = c * At[I][k] + s * At[J][k], t1 = c * At[J][k] - s * At[I][k];
        const double u0 = c * At[I][k + 1] + s * At[J][k + 1], u1 = c * At[J][k + 1] - s * At[I][k + 1];
        At[I][k] = t0; At[J][k] = t1; At[I][k + 1] = u0; At[J][k + 1] = u1;
        a0 = a0 + t0 * t0; b0 = b0 + t1 * t1;
        a1 = a1 + u0 * u0; b1 = b1 + u1 * u1;
    }
    W[I] = a0 + a1;
    W[J] = b0 + b1;
    changed = true;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double t0 = c * V[I][k] + s * V[J][k], t1 = -s * V[I][k] + c * V[J][k];
        V[I][k] = t0; V[J][k] = t1;
    }
}

// Right singular vector of the smallest singular value of the DLT system whose COLUMNS are At[k][*].
template <int M>
__device__ __forceinline__ void dlt_nullvec(double (&At)[4][M], double (&X)[4]) {
    double V[4][4], W[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < M; ++k) sd += At[a][k] * At[a][k];
        W[a] = sd;
#pragma unroll
        for (int k = 0; k < 4; ++k) V[a][k] = (a == k) ? 1.0 : 0.0;
    }
    for (int iter = 0; iter < 30; ++iter) {
        bool changed = false;
        jacobi_pair<M, 0, 1>(At, V, W, changed);
        jacobi_pair<M, 0, 2>(At, V, W, changed);
        jacobi_pair<M, 0, 3>(At, V, W, changed);
        jacobi_pair<M, 1, 2>(At, V, W, changed);
        jacobi_pair<M, 1, 3>(At, V, W, changed);
        jacobi_pair<M, 2, 3>(At, V, W, changed);
        if (!changed) break;
    }
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        double sd = 0;
#pragma unroll
        for (int k = 0; k < M; ++k) sd += At[a][k] * At[a][k];
        W[a] = sqrt(sd);
    }
    // selection sort (descending, strict '<') on (W, row id): which V row ends up last
    int id[4] = {0, 1, 2, 3};
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        double wj = W[a];
        int j = a;
#pragma unroll
        for (int k = a + 1; k < 4; ++k)
            if (wj < W[k]) { wj = W[k]; j = k; }
        // swap slots a and j (j is runtime: predicated over the static candidates)
#pragma unroll
        for (int k = a + 1; k < 4; ++k)
            if (j == k) {
                const double tw = W[a]; W[a] = W[k]; W[k] = tw;
                const int ti = id[a]; id[a] = id[k]; id[k] = ti;
            }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) X[k] = id[3] == 0 ? V[0][k] : id[3] == 1 ? V[1][k] : id[3] == 2 ? V[2][k] : V[3][k];
}

// FUSED: x * P3 - P1 as one fma (the fast path's own system: its result is only kept where it provably casts like the
// faithful one's); the faithful path rounds the product and the difference separately, as the reference's arithmetic does.
template <int M, bool FUSED = false>
__device__ __forceinline__ void dlt_build(double (&At)[4][M], const double* __restrict__ Pa, const double* __restrict__ Pb,
                                          double xa, double ya, double xb, double yb) {
    constexpr int PER = M / 2;
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const double* P = v == 0 ? Pa : Pb;
        const double x = v == 0 ? xa : xb, y = v == 0 ? ya : yb;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if constexpr (FUSED) {
                At[k][v * PER + 0] = fma(x, P[8 + k], -P[k]);
                At[k][v * PER + 1] = fma(y, P[8 + k], -P[4 + k]);
            } else {
                At[k][v * PER + 0] = x * P[8 + k] - P[k];
                At[k][v * PER + 1] = y * P[8 + k] - P[4 + k];
            }
            if (PER == 3) At[k][v * PER + 2] = x * P[4 + k] - y * P[k];
        }
    }
}

// Fast path for the normalised form (`cloud / cloud[3]`, the only form sfm.py uses): the smallest right singular vector of
// the 4x4 DLT matrix A as the smallest eigenvector of A^T A + mu I, by inverse iteration on an LDL^T factorisation
// (mu = 1e-12 trace keeps the pivots positive and does not move the eigenvectors).  The iteration contracts by
// (lambda4 + mu)/(lambda3 + mu) — 2 steps on exact data, <= 6 at 3 px noise on Gustav geometry — and costs ~600 fp64
// instructions per point against ~13 k for the OpenCV-faithful Jacobi sweeps (whose correctly-rounded divisions and
// square roots dominate).  Unit-normalised in fp64, cast to float32, divided by w in float32 exactly as the faithful
// path + sfm.py:54 do, the result is BIT-IDENTICAL to it on > 99.9 % of points and within 1 ulp otherwise (the two
// vectors differ by ~1e-10 before the cast).  A lane that has not converged after kFastIters steps (a start vector
// orthogonal to the solution, lambda3 ~ lambda4: degenerate geometry) runs the Jacobi path instead.  Returns false then.
#ifndef SFM_TRI_ITERS
#define SFM_TRI_ITERS 12
#endif
#ifndef SFM_TRI_ABANDON
#define SFM_TRI_ABANDON 0.0
#endif
constexpr int kFastIters = SFM_TRI_ITERS;
constexpr double kFastAbandon = SFM_TRI_ABANDON;   // > 0: a lane whose step shrinks by less than this factor per iteration gives up (-> Jacobi)

// 1 / d and 1 / sqrt(d) for the fast path (d > 0, far from the ends of the exponent range): the hardware estimate (v_rcp_f64 /
// v_rsq_f64: ~2^-24 relative) and two Newton steps -> <= 1-2 ulp in 5 / 9 instructions instead of the 12 / 27 of the
// correctly rounded division and square root.  The fast path does not need correct rounding — its result is compared against
// float32 rounding boundaries with a margin >= 2^-40, 10^4 x these errors — the OpenCV-faithful Jacobi path keeps `/` and sqrt.
__device__ __forceinline__ double fast_rcp(double d) {
    double y = __builtin_amdgcn_rcp(d);
    y = fma(fma(-d, y, 1.0), y, y);
    return fma(fma(-d, y, 1.0), y, y);
}
__device__ __forceinline__ double fast_rsqrt(double d) {
    double y = __builtin_amdgcn_rsq(d);
    y = fma(0.5 * y, fma(-d, y * y, 1.0), y);
    return fma(0.5 * y, fma(-d, y * y, 1.0), y);
}

__device__ __forceinline__ bool dlt_nullvec_fast(const double (&At)[4][4], double (&X)[4], double* sens = nullptr) {
    // M = A^T A (At[k] is column k of A)
    double m[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            double sd = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) sd = fma(At[a][k], At[b][k], sd);
            m[a][b] = sd;
        }
    const double mu = 1e-12 * (m[0][0] + m[1][1] + m[2][2] + m[3][3]);
    // LDL^T of M + mu I
    const double d0 = m[0][0] + mu;
    if (!(d0 > 0)) return false;
    const double r0 = fast_rcp(d0);
    const double l10 = m[1][0] * r0, l20 = m[2][0] * r0, l30 = m[3][0] * r0;
    const double d1 = fma(-l10, m[1][0], m[1][1] + mu);
    if (!(d1 > 0)) return false;
    const double r1 = fast_rcp(d1);
    const double l21 = fma(-l20, m[1][0], m[2][1]) * r1, l31 = fma(-l30, m[1][0], m[3][1]) * r1;
    const double t21 = l21 * d1, t31 = l31 * d1;
    const double d2 = fma(-l21, t21, fma(-l20, m[2][0], m[2][2] + mu));
    if (!(d2 > 0)) return false;
    const double r2 = fast_rcp(d2);
    const double l32 = fma(-l31, t21, fma(-l30, m[2][0], m[3][2])) * r2;
    const double d3 = fma(-l32, l32 * d2, fma(-l31, t31, fma(-l30, m[3][0], m[3][3] + mu)));
    if (!(d3 > 0)) return false;
    const double r3 = fast_rcp(d3);
    double v0 = 0.5, v1 = 0.5, v2 = 0.5, v3 = 0.5;
    bool done = false;
    // contraction of the iteration = ratio of two consecutive steps, taken while the step is still well above the rounding
    // floor (a floored step would overstate it by orders of magnitude); the first step's "ratio" is against the arbitrary start
    // vector and is skipped.  Numerator and denominator are kept and divided ONCE after the loop.
    double diff_prev = 0.0, rho_num = 0.0, rho_den = 1.0, nn_last = 1.0;
    for (int it = 0; it < kFastIters && !done; ++it) {
        // L y = v
        const double y0 = v0;
        const double y1 = fma(-l10, y0, v1);
        const double y2 = fma(-l21, y1, fma(-l20, y0, v2));
        const double y3 = fma(-l32, y2, fma(-l31, y1, fma(-l30, y0, v3)));
        // D z = y,  L^T w = z
        const double w3 = y3 * r3;
        const double w2 = fma(-l32, w3, y2 * r2);
        const double w1 = fma(-l31, w3, fma(-l21, w2, y1 * r1));
        const double w0 = fma(-l30, w3, fma(-l20, w2, fma(-l10, w1, y0 * r0)));
        const double nn = fma(w0, w0, fma(w1, w1, fma(w2, w2, w3 * w3)));
        const double inv = fast_rsqrt(nn);
        const double sgn = (fma(w0, v0, fma(w1, v1, fma(w2, v2, w3 * v3))) < 0) ? -inv : inv;
        const double n0 = w0 * sgn, n1 = w1 * sgn, n2 = w2 * sgn, n3 = w3 * sgn;
        const double diff = fmax(fmax(fabs(n0 - v0), fabs(n1 - v1)), fmax(fabs(n2 - v2), fabs(n3 - v3)));
        v0 = n0; v1 = n1; v2 = n2; v3 = n3;
        done = diff < 1e-13;
        if (kFastAbandon > 0.0 && it >= 1 && !done && diff > kFastAbandon * diff_prev) return false;
        if (it >= 1 && (diff >= 1e-14 || rho_num == 0.0)) {
            rho_num = fmax(diff, 1e-16);
            rho_den = fmax(diff_prev, 1e-300);
        }
        diff_prev = diff;
        nn_last = nn;                                                   // sqrt(nn) -> 1 / (lambda4 + mu) as v converges
    }
    X[0] = v0; X[1] = v1; X[2] = v2; X[3] = v3;
    // How far can this vector be from the one another backward-stable algorithm returns?  ~ eps * lambda1 / lambda3: with
    // rho = (lambda4 + mu) / (lambda3 + mu) the contraction the iteration showed (ratio of consecutive steps) and
    // lambda1 <= trace, lambda1 / lambda3 ~ trace * rho * growth.  Measured over well- and ill-conditioned geometries
    // (scripts/dev_tri_calib.py, 4e6 points): |fast - Jacobi| <= 2.5 sens, 3.5e-15 at the rounding floor.
    if (sens) {
        const double rho = rho_num == 0.0 ? 1.0 : rho_num / rho_den;
        *sens = 2.220446049250313e-16 * (m[0][0] + m[1][1] + m[2][2] + m[3][3]) * fmin(1.0, rho) * sqrt(nn_last);
    }
    return done;
}

// Guarded fast path (normalise_w = 3): the inverse-iteration vector and the Jacobi vector are the same unit vector up to
// ~1e-11 (fp64 rounding of two different algorithms), so their float32 casts — and with them the whole float32 result —
// are identical unless a component sits within that distance of a float32 ROUNDING BOUNDARY (the midpoint of two adjacent
// floats).  The distance between two backward-stable solutions grows with the conditioning of the null vector, ~ eps
// lambda1 / lambda3 of A^T A, which the iteration itself reveals (dlt_nullvec_fast: `sens`).  A lane whose four components
// all keep a margin of max(2^-40, 16 sens) — in units of the unit vector's norm — from the nearest midpoint keeps the fast
// result; the others (a fraction of a percent on well-conditioned geometry, all of them when the baseline vanishes) and the
// lanes whose iteration did not converge are marked and redone by triangulate_fixup_kernel with the Jacobi sweeps,
// compacted so that no wave runs the slow path for a single lane.
constexpr double kCastGuard = 9.094947017729282e-13;    // 2^-40: 300 x the rounding floor of the two vectors' difference (<= 3.5e-15 measured)
constexpr double kSensFactor = 16.0;                    // margin = max(kCastGuard, kSensFactor * sens); measured |fast on the fused system - Jacobi on the unfused one| <= 2.0 sens (profiles/r05_tri_guard_calibration.txt)
constexpr unsigned kRedoMark = 0x7FC0DEADu;             // quiet-NaN payload written to all four outputs of a point to redo

__device__ __forceinline__ bool cast_margin_ok(double x, double margin) {
    const double ax = fabs(x);
    const float af = (float)ax;                                        // round to nearest even
    const unsigned bits = __float_as_uint(af);
    if (bits == 0u) return true;                                       // |x| < 2^-150: zero on both paths
    if (bits >= 0x7F7FFFFFu) return false;
    const double lo = 0.5 * ((double)af + (double)__uint_as_float(bits - 1u));   // midpoint to the float below
    const double hi = 0.5 * ((double)af + (double)__uint_as_float(bits + 1u));   // ... and above
    return ax - lo > margin && hi - ax > margin;                    // (absolute: the vector has unit norm)
}

template <int M>
__device__ __forceinline__ void store_point(const double (&Xd)[4], int normalise_w, int64_t n, int64_t i, float* __restrict__ X4) {
    float X[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) X[k] = (float)Xd[k];
    if (normalise_w) {
        const float w = X[3];
#pragma unroll
        for (int k = 0; k < 4; ++k) X[k] = X[k] / w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) X4[k * n + i] = X[k];
}

// Second pass of the guarded fast path: a workgroup scans chunks of kFixChunk points for the redo mark (row 3 first: one
// 4-byte read per point; row 0 only where row 3 is marked), queues their indices in LDS, and runs the OpenCV-faithful Jacobi
// path on the queue, 256 points at a time.  The grid is at most one resident round of the chip (1024 workgroups) walking
// the chunks, and a chunk is large (fixup_chunk: n / 512 rounded up, 2048 .. 20 480 points) so that the points a pass marks
// come in few, well-filled waves: a Jacobi solve is ~13 000 fp64 instructions whether one lane of the wave needs it or all 64
// (round 3: 4096-point chunks, 2 442 workgroups of one or two sparse waves each, in 2.4 rounds).
constexpr int kFixStep = 256 * 8, kFixChunkMax = 10 * kFixStep;      // chunk = a multiple of 2048 points, at most 20 480 (16-bit queue entries: 40 KiB of LDS)
inline int fixup_chunk(int64_t n) {
    // Few, large chunks: a Jacobi solve costs a wave ~13 000 fp64 instructions whether one of its lanes needs it or all 64, and
    // the waves of the workgroups resident on a SIMD take turns — what counts is the number of (partly filled) waves per SIMD.
    // With ~0.7 % of the points marked (PMC, profiles/r04_other_pmc.md) 10 240-point chunks gave 72 marked points per workgroup =
    // a full wave + one of 8 lanes, on 977 workgroups: ~1.9 waves per SIMD; n / 512 rounded up (<= 20 480) gives ~140 = 2.2 waves
    // on ~490 workgroups: ~1.3 per SIMD.
    const int64_t per = (n + 511) / 512;
    const int64_t c = (per + kFixStep - 1) / kFixStep * kFixStep;
    return (int)std::min<int64_t>(std::max<int64_t>(c, kFixStep), kFixChunkMax);
}
__global__ __launch_bounds__(256) void triangulate_fixup_kernel(ProjPair P, const float* __restrict__ x1, const float* __restrict__ x2, int64_t n,
                                                               int64_t spt, int64_t sxy, int chunk, float* __restrict__ X4,
                                                               const int* __restrict__ list /*null, or the first pass's reject list: {count, overflow, indices ...}*/) {
    extern __shared__ unsigned short queue[];                       // [chunk] offsets inside the chunk (< 20 480)
    __shared__ int qn;
    if (list && list[1] == 0) {                                     // (uniform) every rejected point is on the first pass's compact list:
        const int total = list[0];                                  // no scan — at 1e7 points it was 35 of this pass's 57 us — and every wave but the last is full
        for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
            const int64_t i = list[2 + e];
            double At[4][4], Xd[4];
            dlt_build<4>(At, P.p[0], P.p[1], (double)x1[i * spt], (double)x1[i * spt + sxy], (double)x2[i * spt], (double)x2[i * spt + sxy]);
            dlt_nullvec<4>(At, Xd);
            store_point<4>(Xd, 1, n, i, X4);
        }
        return;
    }
    for (int64_t base = (int64_t)blockIdx.x * chunk; base < n; base += (int64_t)gridDim.x * chunk) {
        __syncthreads();                                            // (the previous chunk's queue has been worked off)
        if (threadIdx.x == 0) qn = 0;
        __syncthreads();
        for (int o0 = threadIdx.x; o0 < chunk; o0 += kFixStep) {    // eight independent reads in flight per lane
            unsigned w[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int64_t i = base + o0 + 256 * k;
                w[k] = i < n ? __float_as_uint(X4[3 * n + i]) : 0u;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if (w[k] == kRedoMark && __float_as_uint(X4[base + o0 + 256 * k]) == kRedoMark) queue[atomicAdd(&qn, 1)] = (unsigned short)(o0 + 256 * k);
        }
        __syncthreads();
        const int total = qn;
        for (int e = threadIdx.x; e < total; e += 256) {
            const int64_t i = base + queue[e];
            double At[4][4], Xd[4];
            dlt_build<4>(At, P.p[0], P.p[1], (double)x1[i * spt], (double)x1[i * spt + sxy], (double)x2[i * spt], (double)x2[i * spt + sxy]);
            dlt_nullvec<4>(At, Xd);
            store_point<4>(Xd, 1, n, i, X4);
        }
    }
}

template <int M>
__global__ __launch_bounds__(256) void triangulate_kernel(ProjPair P, const float* __restrict__ x1,
                                                          const float* __restrict__ x2, int64_t n, int64_t spt,
                                                          int64_t sxy, int normalise_w, double sens_factor, double base_guard,
                                                          float* __restrict__ X4) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    double At[4][M];   // At[k][row]: column k of the DLT matrix
    dlt_build<M>(At, P.p[0], P.p[1], (double)x1[i * spt], (double)x1[i * spt + sxy], (double)x2[i * spt],
                 (double)x2[i * spt + sxy]);
    double Xd[4];
    bool have = false;
    if constexpr (M == 4) {
        double sens = 0;
        // (normalise_w = 3, the guarded path, never comes here: it is triangulate_guarded_kernel + triangulate_fixup_kernel)
        if (normalise_w == 2) have = dlt_nullvec_fast(At, Xd, &sens);
#ifdef SFM_DEV_BUILD
        if (normalise_w == 4) {
            // dev calibration of the guard (scripts/dev/dev_tri_calib.py of the round-5 tree; not in release builds) on the SHIPPED configuration: the
            // inverse iteration on the FMA-fused system (dlt_build<4, true>: what triangulate_guarded_kernel / tri_matches_points_kernel
            // solve) against the Jacobi path on the unfused one (what the fix-up pass and the faithful kernel solve).  Fusing moves
            // every entry of A by ~1 ulp, i.e. the null vector by the order of `sens` itself, so the two must be measured together.
            // X4[0] = max |fast - jacobi| over the components (sign-aligned), X4[1] = sens
            double Af[4][4], Xj[4];
            dlt_build<4, true>(Af, P.p[0], P.p[1], (double)x1[i * spt], (double)x1[i * spt + sxy], (double)x2[i * spt], (double)x2[i * spt + sxy]);
            have = dlt_nullvec_fast(Af, Xd, &sens);
            dlt_nullvec<M>(At, Xj);
            const double sg = (Xj[0] * Xd[0] + Xj[1] * Xd[1] + Xj[2] * Xd[2] + Xj[3] * Xd[3]) < 0 ? -1.0 : 1.0;
            double d = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) d = fmax(d, fabs(Xd[k] - sg * Xj[k]));
            X4[i] = have ? (float)d : -1.f;
            X4[n + i] = (float)sens;
            X4[2 * n + i] = 0.f;
            X4[3 * n + i] = 0.f;
            return;
        }
#endif
    }
    if (!have) dlt_nullvec<M>(At, Xd);
    store_point<M>(Xd, normalise_w, n, i, X4);
}

// First pass of the guarded fast path (normalise_w = 3), written for the vector ALU: nothing but the inverse iteration is
// compiled in (the Jacobi sweeps of the general kernel cost it a fifth of its registers), a lane walks points i, i + T,
// i + 2T ... of a grid sized to the chip, and the four coordinates of its NEXT point are requested before the current one is
// solved — the ~2 us a load takes under traffic is covered by the ~640 fp64 instructions of a solve instead of being waited
// for by a freshly launched workgroup (round 3: 39 000 workgroups of one point per lane at 1e7 points, waitcnt share 0.39).
// PACKED: the reference's layout — transposed views of (N, 2) arrays, sfm.py:47-48 — is read as one float2 per point.
template <bool PACKED>
__global__ __launch_bounds__(256) void triangulate_guarded_kernel(ProjPair P, const float* __restrict__ x1, const float* __restrict__ x2, int64_t n,
                                                                  int64_t spt, int64_t sxy, double sens_factor, double base_guard,
                                                                  float* __restrict__ X4, int* __restrict__ list /*null, or {count, overflow, indices[cap]}*/, int cap) {
    // Rejected points are marked in X4 (the scan pass finds them) and, when the caller provides a list, also queued per workgroup
    // (LDS) and appended to it with ONE global atomic per workgroup at the end of its walk (per point that would be ~70 000
    // arrivals on one address at 1e7 points: ~0.8 ms).
    constexpr int kQ = 1024;
    __shared__ int rq[kQ];
    __shared__ int rqn, rbase;
    if (list) { if (threadIdx.x == 0) rqn = 0; __syncthreads(); }
    const int64_t T = (int64_t)gridDim.x * 256;
    int64_t i = blockIdx.x * (int64_t)256 + threadIdx.x;
    float xa = 0.f, ya = 0.f, xb = 0.f, yb = 0.f;
    auto fetch = [&](int64_t k, float& ax, float& ay, float& bx, float& by) {
        if constexpr (PACKED) {
            const float2 a = reinterpret_cast<const float2*>(x1)[k], b = reinterpret_cast<const float2*>(x2)[k];
            ax = a.x; ay = a.y; bx = b.x; by = b.y;
        } else {
            ax = x1[k * spt]; ay = x1[k * spt + sxy]; bx = x2[k * spt]; by = x2[k * spt + sxy];
        }
    };
    if (i < n) fetch(i, xa, ya, xb, yb);
    while (i < n) {
        const int64_t nx = i + T;
        float nxa = 0.f, nya = 0.f, nxb = 0.f, nyb = 0.f;
        if (nx < n) fetch(nx, nxa, nya, nxb, nyb);
        double At[4][4], Xd[4], sens = 0;
        dlt_build<4, true>(At, P.p[0], P.p[1], (double)xa, (double)ya, (double)xb, (double)yb);
        const bool have = dlt_nullvec_fast(At, Xd, &sens);
        const double margin = fmax(base_guard, sens_factor * sens);
        const bool keep = have && margin < 1e-3 && cast_margin_ok(Xd[0], margin) && cast_margin_ok(Xd[1], margin) &&
                          cast_margin_ok(Xd[2], margin) && cast_margin_ok(Xd[3], margin);
        if (keep) {
            store_point<4>(Xd, 1, n, i, X4);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) X4[k * n + i] = __uint_as_float(kRedoMark);
            if (list) {
                const int at = atomicAdd(&rqn, 1);
                if (at < kQ) rq[at] = (int)i;                         // (list mode is offered for n < 2^31 only)
            }
        }
        xa = nxa; ya = nya; xb = nxb; yb = nyb;
        i = nx;
    }
    if (list) {
        __syncthreads();
        const int total = rqn;                                      // (uniform)
        if (total > kQ) {                                           // more rejects than the queue holds: leave it to the scan
            if (threadIdx.x == 0) atomicExch(&list[1], 1);
            return;
        }
        if (threadIdx.x == 0) rbase = total ? atomicAdd(&list[0], total) : 0;
        __syncthreads();
        if (rbase + total > cap) {
            if (threadIdx.x == 0) atomicExch(&list[1], 1);
            return;
        }
        for (int e = threadIdx.x; e < total; e += 256) list[2 + rbase + e] = rq[e];
    }
}

// ---------------------------------------------------------------- matches -> points in one pass (sfm.py:262-268, then :53-54)
// The Lowe loop, the keypoint gather and the triangulation of up to kTmBatch image pairs, from the KNN blocks the matcher
// wrote ({trainIdx x2}, {distance x2} per query) straight to [4 x cap] point blocks — no survivor list, no gathered
// coordinate arrays and no host round trip in between (the pair-sharded path used to re-derive the survivors with a stable
// sort per pair and gather with index kernels: 20 % of the config-5 job for 0.1 ms of arithmetic).
//   count:  one workgroup per kTmChunk queries -> its number of ratio survivors;
//   points: a workgroup sums the counts of its predecessors (ascending queryIdx = the order of the Python loop), lists its
//           own survivors in LDS in order, and runs the GUARDED fast DLT on them (normalise_w = 3 of sfm_triangulate_dlt:
//           same code, same constants); the lanes the guard rejects are queued in LDS and redone by the same workgroup with
//           the OpenCV-faithful Jacobi sweeps, so a point's value is that of triangulate_kernel<4> + fixup, bit for bit.
//           Columns >= the pair's survivor count are zeroed, the count is stored behind the block.
constexpr int kTmBatch = 8;
constexpr int kTmChunk = 1024;
struct TriMatchArgs {
    const int* idx[kTmBatch];        // [nq][2] trainIdx of the two neighbours
    const float* dist[kTmBatch];     // [nq][2] their distances
    const float2* kp0[kTmBatch];     // query image's keypoints (KeyPoint.pt)
    const float2* kp1[kTmBatch];     // train image's
    float* X4[kTmBatch];             // [4][ldx] points
    int* count[kTmBatch];            // survivors of the pair (device scalar)
    int nq[kTmBatch];
    double P[kTmBatch][2][12];
};

__device__ __forceinline__ unsigned tm_ratio_bits(const int* __restrict__ idx, const float* __restrict__ dist, int nq, int qb, double ratio, int (&ti)[4]) {
    unsigned pass = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int q = qb + k;
        ti[k] = -1;
        if (q < nq) {
            const int2 ii = *reinterpret_cast<const int2*>(idx + 2 * q);
            const float2 dd = *reinterpret_cast<const float2*>(dist + 2 * q);
            ti[k] = ii.x;
            pass |= ((ii.y >= 0) && ((double)dd.x < ratio * (double)dd.y) ? 1u : 0u) << k;   // sfm.py:264: float32 promoted to double, strict <
        }
    }
    return pass;
}

__global__ __launch_bounds__(256) void tri_matches_count_kernel(TriMatchArgs A, double ratio, int* __restrict__ block_count) {
    __shared__ int wsum[4];
    const int pb = blockIdx.y;
    const int qb = blockIdx.x * kTmChunk + threadIdx.x * 4;
    int ti[4];
    int c = __popc(tm_ratio_bits(A.idx[pb], A.dist[pb], A.nq[pb], qb, ratio, ti));
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_count[pb * gridDim.x + blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

__global__ __launch_bounds__(256) void tri_matches_points_kernel(TriMatchArgs A, double ratio, const int* __restrict__ block_count, int64_t ldx,
                                                                 double sens_factor, double base_guard) {
    __shared__ int list_q[kTmChunk], list_t[kTmChunk], redo[kTmChunk];
    __shared__ int wsum[8], base_s, total_s, nredo;
    const int pb = blockIdx.y, nblk = gridDim.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nq = A.nq[pb];
    // exclusive prefix of the preceding workgroups' counts, and the pair's total
    int part = 0, tot = 0;
    for (int b = threadIdx.x; b < nblk; b += 256) {
        const int c = block_count[pb * nblk + b];
        tot += c;
        if (b < (int)blockIdx.x) part += c;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        part += __shfl_xor(part, m, 64);
        tot += __shfl_xor(tot, m, 64);
    }
    if (lane == 0) { wsum[wave] = part; wsum[4 + wave] = tot; }
    if (threadIdx.x == 0) nredo = 0;
    __syncthreads();
    if (threadIdx.x == 0) {
        base_s = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        total_s = wsum[4] + wsum[5] + wsum[6] + wsum[7];
    }
    __syncthreads();
    const int base = base_s, total = total_s;
    __syncthreads();
    // this workgroup's survivors, in ascending queryIdx order
    const int qb = blockIdx.x * kTmChunk + threadIdx.x * 4;
    int ti[4];
    const unsigned pass = tm_ratio_bits(A.idx[pb], A.dist[pb], nq, qb, ratio, ti);
    const int mine = __popc(pass);
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    const int m_wg = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    int pos = woff + incl - mine;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (pass & (1u << k)) {
            list_q[pos] = qb + k;
            list_t[pos] = ti[k];
            ++pos;
        }
    __syncthreads();
    const float2* __restrict__ kp0 = A.kp0[pb];
    const float2* __restrict__ kp1 = A.kp1[pb];
    float* __restrict__ X4 = A.X4[pb];
    for (int e = threadIdx.x; e < m_wg; e += 256) {
        const float2 a = kp0[list_q[e]], b = kp1[list_t[e]];
        double At[4][4], Xd[4], sens = 0;
        dlt_build<4, true>(At, A.P[pb][0], A.P[pb][1], (double)a.x, (double)a.y, (double)b.x, (double)b.y);
        const bool have = dlt_nullvec_fast(At, Xd, &sens);
        const double margin = fmax(base_guard, sens_factor * sens);
        const bool keep = have && margin < 1e-3 && cast_margin_ok(Xd[0], margin) && cast_margin_ok(Xd[1], margin) &&
                          cast_margin_ok(Xd[2], margin) && cast_margin_ok(Xd[3], margin);
        if (keep) store_point<4>(Xd, 1, ldx, base + e, X4);
        else redo[atomicAdd(&nredo, 1)] = e;
    }
    __syncthreads();
    const int nr = nredo;
    for (int r = threadIdx.x; r < nr; r += 256) {                  // (stored by position: the queue's order does not matter)
        const int e = redo[r];
        const float2 a = kp0[list_q[e]], b = kp1[list_t[e]];
        double At[4][4], Xd[4];
        dlt_build<4>(At, A.P[pb][0], A.P[pb][1], (double)a.x, (double)a.y, (double)b.x, (double)b.y);
        dlt_nullvec<4>(At, Xd);
        store_point<4>(Xd, 1, ldx, base + e, X4);
    }
    // columns past the survivors: zero (an exchange slot still holds the points of an earlier pair)
    for (int64_t c = (int64_t)blockIdx.x * kTmChunk + threadIdx.x; c < min((int64_t)(blockIdx.x + 1) * kTmChunk, ldx); c += 256)
        if (c >= total) {
#pragma unroll
            for (int k = 0; k < 4; ++k) X4[k * ldx + c] = 0.f;
        }
    if (blockIdx.x == 0 && threadIdx.x == 0 && A.count[pb]) *A.count[pb] = total;
}

// cv2.recoverPose's cheirality vote (sfm.py:311): for pose candidate m = blockIdx.y triangulate every
// K-normalised correspondence against [I|0] / [R|t] in fp64 (points are double there) and test
//   Q2*Q3 > 0,  Q2/Q3 < dist,  0 < z' < dist  with z' = third row of [R|t] * (Q/Q3).
struct PoseCands {
    double p[4][12];
};

template <int M>
__global__ __launch_bounds__(256) void recover_pose_kernel(PoseCands C, const double* __restrict__ x1n,
                                                           const double* __restrict__ x2n, int64_t n, double dist,
                                                           int* __restrict__ counts, unsigned char* __restrict__ mask) {
    const int m = blockIdx.y;
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    bool good = false;
    if (i < n) {
        const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
        const double* P1 = C.p[m];
        double At[4][M];
        dlt_build<M>(At, P0, P1, x1n[2 * i], x1n[2 * i + 1], x2n[2 * i], x2n[2 * i + 1]);
        double Q[4];
        dlt_nullvec<M>(At, Q);
        good = Q[2] * Q[3] > 0;
        const double q0 = Q[0] / Q[3], q1 = Q[1] / Q[3], q2 = Q[2] / Q[3], q3 = Q[3] / Q[3];
        good = (q2 < dist) && good;
        const double z = ((P1[8] * q0 + P1[9] * q1) + P1[10] * q2) + P1[11] * q3;
        good = (z > 0) && good;
        good = (z < dist) && good;
        if (mask) mask[(int64_t)m * n + i] = good ? 255 : 0;
    }
    const int cnt = __popcll(__ballot(good));
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&counts[m], cnt);
}

}  // namespace

extern "C" int sfm_triangulate_dlt(const double* P1, const double* P2, const float* x1, const float* x2, int64_t n,
                                   int64_t stride_pt, int64_t stride_xy, int rows, int normalise_w, float* X4,
                                   void* stream_) {
    SFM_CHECK_ARG(rows == 4 || rows == 6, "sfm_triangulate_dlt: rows must be 4 or 6 (got %d)", rows);
#ifdef SFM_DEV_BUILD
    constexpr int kMaxMode = 4;     // + the calibration mode of scripts/dev/dev_tri_calib.py of the round-5 tree
#else
    constexpr int kMaxMode = 3;
#endif
    SFM_CHECK_ARG(normalise_w >= 0 && normalise_w <= kMaxMode && (normalise_w < 2 || rows == 4),
                  "sfm_triangulate_dlt: normalise_w must be 0, 1, 2 (fast path) or 3 (guarded fast path); 2 and 3 need rows = 4");
    SFM_CHECK_ARG(n >= 0, "sfm_triangulate_dlt: negative n");
    if (n == 0) return SFM_OK;
    SFM_CHECK_ARG(P1 && P2 && x1 && x2 && X4, "sfm_triangulate_dlt: null pointer");
    ProjPair P;
    for (int k = 0; k < 12; ++k) {
        P.p[0][k] = P1[k];
        P.p[1][k] = P2[k];
    }
    const dim3 grid((unsigned)((n + 255) / 256));
    // The guard's constants are part of the bit-identity claim (an EMPIRICAL bound, see the kernel's comment): a release build
    // takes no override from the environment.
#ifdef SFM_DEV_BUILD
    static const double sens_factor = [] { const char* e = getenv("SFM_TRI_SENS"); return e ? atof(e) : kSensFactor; }();
    static const double base_guard = [] { const char* e = getenv("SFM_TRI_GUARD"); return e ? atof(e) : kCastGuard; }();
#else
    constexpr double sens_factor = kSensFactor, base_guard = kCastGuard;
#endif
    sfm::prof_begin(sfm::kProfTriangulate, sfm::as_stream(stream_));
    int* list = nullptr;
    int cap = 0;
    if (normalise_w == 3 && n >= (1 << 18) && n < INT_MAX) {
        // large calls: a stream-ordered scratch list for the first pass's rejects ({count, overflow flag, indices}; ~0.7 % of the
        // points are rejected, room for 1/8 of them).  If the allocation is refused the marks are scanned as for small calls.
        cap = (int)(n / 8);
        if (hipMallocAsync(reinterpret_cast<void**>(&list), sizeof(int) * (size_t)(cap + 2), sfm::as_stream(stream_)) != hipSuccess) {
            (void)hipGetLastError();
            list = nullptr;
        } else if (hipMemsetAsync(list, 0, 2 * sizeof(int), sfm::as_stream(stream_)) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFreeAsync(list, sfm::as_stream(stream_));
            list = nullptr;
        }
    }
    if (normalise_w == 3) {
        // (rows == 4) a grid sized to the chip: 256 CUs x the kernel's resident workgroups, a lane walks its points
        const dim3 pgrid((unsigned)std::min<int64_t>((n + 255) / 256, 256 * 6));
        if (stride_pt == 2 && stride_xy == 1 && ((uintptr_t)x1 & 7) == 0 && ((uintptr_t)x2 & 7) == 0)
            hipLaunchKernelGGL(triangulate_guarded_kernel<true>, pgrid, dim3(256), 0, sfm::as_stream(stream_), P, x1, x2, n, stride_pt, stride_xy,
                               sens_factor, base_guard, X4, list, cap);
        else
            hipLaunchKernelGGL(triangulate_guarded_kernel<false>, pgrid, dim3(256), 0, sfm::as_stream(stream_), P, x1, x2, n, stride_pt, stride_xy,
                               sens_factor, base_guard, X4, list, cap);
    } else if (rows == 4)
        hipLaunchKernelGGL(triangulate_kernel<4>, grid, dim3(256), 0, sfm::as_stream(stream_), P, x1, x2, n, stride_pt,
                           stride_xy, normalise_w, sens_factor, base_guard, X4);
    else
        hipLaunchKernelGGL(triangulate_kernel<6>, grid, dim3(256), 0, sfm::as_stream(stream_), P, x1, x2, n, stride_pt,
                           stride_xy, normalise_w, sens_factor, base_guard, X4);
    if (normalise_w == 3)
    {
        const int chunk = fixup_chunk(n);
        hipLaunchKernelGGL(triangulate_fixup_kernel, dim3((unsigned)std::min<int64_t>((n + chunk - 1) / chunk, 1024)), dim3(256), sizeof(unsigned short) * (size_t)chunk,
                           sfm::as_stream(stream_), P, x1, x2, n, stride_pt, stride_xy, chunk, X4, (const int*)list);
        if (list) (void)hipFreeAsync(list, sfm::as_stream(stream_));
    }
    sfm::prof_end(sfm::kProfTriangulate, sfm::as_stream(stream_));
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}

extern "C" size_t sfm_triangulate_matches_batch_ws_bytes(int batch, int64_t cap) {
    if (batch < 1 || batch > kTmBatch || cap < 0) return 0;
    return sfm::align_up(sizeof(int) * (size_t)batch * (size_t)((cap + kTmChunk - 1) / kTmChunk + 1), 256);
}

extern "C" int sfm_triangulate_matches_batch(int batch, const int32_t* const* knn_idx, const float* const* knn_dist, const int64_t* nq_host,
                                             double ratio, const float* const* kp0, const float* const* kp1, const double* P_host, int64_t cap,
                                             float* const* X4, int32_t* const* count, void* ws, size_t ws_bytes, void* stream_) {
    SFM_CHECK_ARG(batch >= 1 && batch <= kTmBatch, "sfm_triangulate_matches_batch: batch must be 1..%d (got %d)", kTmBatch, batch);
    SFM_CHECK_ARG(cap >= 0 && cap < INT32_MAX, "sfm_triangulate_matches_batch: bad cap");
    SFM_CHECK_ARG(knn_idx && knn_dist && nq_host && kp0 && kp1 && P_host && X4, "sfm_triangulate_matches_batch: null pointer");
    if (cap == 0) return SFM_OK;
    TriMatchArgs A;
    for (int b = 0; b < kTmBatch; ++b) {
        const int s = b < batch ? b : 0;
        SFM_CHECK_ARG(nq_host[s] >= 0 && nq_host[s] <= cap, "sfm_triangulate_matches_batch: pair %d has %lld queries, the point block %lld columns", s,
                      (long long)nq_host[s], (long long)cap);
        SFM_CHECK_ARG(X4[s] && (nq_host[s] == 0 || (knn_idx[s] && knn_dist[s] && kp0[s] && kp1[s])), "sfm_triangulate_matches_batch: null pointer (pair %d)", s);
        SFM_CHECK_ARG(((uintptr_t)knn_idx[s] & 7) == 0 && ((uintptr_t)knn_dist[s] & 7) == 0 && ((uintptr_t)kp0[s] & 7) == 0 && ((uintptr_t)kp1[s] & 7) == 0,
                      "sfm_triangulate_matches_batch: KNN blocks and keypoint arrays must be 8-byte aligned");
        A.idx[b] = knn_idx[s];
        A.dist[b] = knn_dist[s];
        A.kp0[b] = reinterpret_cast<const float2*>(kp0[s]);
        A.kp1[b] = reinterpret_cast<const float2*>(kp1[s]);
        A.X4[b] = X4[s];
        A.count[b] = count ? count[s] : nullptr;
        A.nq[b] = (int)nq_host[s];
        for (int k = 0; k < 24; ++k) A.P[b][k / 12][k % 12] = P_host[(size_t)s * 24 + k];
    }
    if (!ws || ws_bytes < sfm_triangulate_matches_batch_ws_bytes(batch, cap)) {
        sfm::set_error("sfm_triangulate_matches_batch: workspace too small (%zu < %zu)", ws_bytes, sfm_triangulate_matches_batch_ws_bytes(batch, cap));
        return SFM_ERR_WORKSPACE;
    }
    hipStream_t stream = sfm::as_stream(stream_);
    const dim3 grid((unsigned)((cap + kTmChunk - 1) / kTmChunk), (unsigned)batch);
    int* counts = static_cast<int*>(ws);
    sfm::prof_begin(sfm::kProfTriangulate, stream);
    hipLaunchKernelGGL(tri_matches_count_kernel, grid, dim3(256), 0, stream, A, ratio, counts);
    hipLaunchKernelGGL(tri_matches_points_kernel, grid, dim3(256), 0, stream, A, ratio, counts, cap, kSensFactor, kCastGuard);
    sfm::prof_end(sfm::kProfTriangulate, stream);
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}

extern "C" int sfm_recover_pose_score(const double* P_host, int h, const double* x1n, const double* x2n, int64_t n,
                                      double dist_thresh, int rows, int32_t* counts, uint8_t* mask, void* stream_) {
    SFM_CHECK_ARG(h >= 1 && h <= 4 && n >= 0 && (rows == 4 || rows == 6), "sfm_recover_pose_score: bad sizes");
    SFM_CHECK_ARG(P_host && counts && (n == 0 || (x1n && x2n)), "sfm_recover_pose_score: null pointer");
    hipStream_t stream = sfm::as_stream(stream_);
    SFM_CHECK_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)h, stream));
    if (n == 0) return SFM_OK;
    PoseCands C;
    for (int m = 0; m < 4; ++m)
        for (int k = 0; k < 12; ++k) C.p[m][k] = m < h ? P_host[m * 12 + k] : 0.0;
    const dim3 grid((unsigned)((n + 255) / 256), (unsigned)h);
    if (rows == 4)
        hipLaunchKernelGGL(recover_pose_kernel<4>, grid, dim3(256), 0, stream, C, x1n, x2n, n, dist_thresh, counts, mask);
    else
        hipLaunchKernelGGL(recover_pose_kernel<6>, grid, dim3(256), 0, stream, C, x1n, x2n, n, dist_thresh, counts, mask);
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}
