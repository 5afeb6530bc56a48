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
// lane owns a point and a wave solves 64 points in lockstep.
#include "common.h"
#include <cfloat>

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

template <int M>
__global__ __launch_bounds__(256) void triangulate_kernel(ProjPair P, const float* __restrict__ x1,
                                                          const float* __restrict__ x2, int64_t n, int64_t spt,
                                                          int64_t sxy, int normalise_w, float* __restrict__ X4) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int PER = M / 2;
    double At[4][M];   // At[k][row]: column k of the DLT matrix
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const float* xs = v == 0 ? x1 : x2;
        const double x = (double)xs[i * spt];
        const double y = (double)xs[i * spt + sxy];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            At[k][v * PER + 0] = x * P.p[v][8 + k] - P.p[v][k];
            At[k][v * PER + 1] = y * P.p[v][8 + k] - P.p[v][4 + k];
            if (PER == 3) At[k][v * PER + 2] = x * P.p[v][4 + k] - y * P.p[v][k];
        }
    }
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
    float X[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const double v = id[3] == 0 ? V[0][k] : id[3] == 1 ? V[1][k] : id[3] == 2 ? V[2][k] : V[3][k];
        X[k] = (float)v;
    }
    if (normalise_w) {
        const float w = X[3];
#pragma unroll
        for (int k = 0; k < 4; ++k) X[k] = X[k] / w;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) X4[k * n + i] = X[k];
}

}  // namespace

extern "C" int sfm_triangulate_dlt(const double* P1, const double* P2, const float* x1, const float* x2, int64_t n,
                                   int64_t stride_pt, int64_t stride_xy, int rows, int normalise_w, float* X4,
                                   void* stream_) {
    SFM_CHECK_ARG(rows == 4 || rows == 6, "sfm_triangulate_dlt: rows must be 4 or 6 (got %d)", rows);
    SFM_CHECK_ARG(n >= 0, "sfm_triangulate_dlt: negative n");
    if (n == 0) return SFM_OK;
    SFM_CHECK_ARG(P1 && P2 && x1 && x2 && X4, "sfm_triangulate_dlt: null pointer");
    ProjPair P;
    for (int k = 0; k < 12; ++k) {
        P.p[0][k] = P1[k];
        P.p[1][k] = P2[k];
    }
    const dim3 grid((unsigned)((n + 255) / 256));
    sfm::prof_begin(sfm::kProfTriangulate, sfm::as_stream(stream_));
    if (rows == 4)
        hipLaunchKernelGGL(triangulate_kernel<4>, grid, dim3(256), 0, sfm::as_stream(stream_), P, x1, x2, n, stride_pt,
                           stride_xy, normalise_w, X4);
    else
        hipLaunchKernelGGL(triangulate_kernel<6>, grid, dim3(256), 0, sfm::as_stream(stream_), P, x1, x2, n, stride_pt,
                           stride_xy, normalise_w, X4);
    sfm::prof_end(sfm::kProfTriangulate, sfm::as_stream(stream_));
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}
