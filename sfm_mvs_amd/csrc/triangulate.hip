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

template <int M>
__device__ __forceinline__ void dlt_build(double (&At)[4][M], const double* __restrict__ Pa, const double* __restrict__ Pb,
                                          double xa, double ya, double xb, double yb) {
    constexpr int PER = M / 2;
#pragma unroll
    for (int v = 0; v < 2; ++v) {
        const double* P = v == 0 ? Pa : Pb;
        const double x = v == 0 ? xa : xb, y = v == 0 ? ya : yb;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            At[k][v * PER + 0] = x * P[8 + k] - P[k];
            At[k][v * PER + 1] = y * P[8 + k] - P[4 + k];
            if (PER == 3) At[k][v * PER + 2] = x * P[4 + k] - y * P[k];
        }
    }
}

template <int M>
__global__ __launch_bounds__(256) void triangulate_kernel(ProjPair P, const float* __restrict__ x1,
                                                          const float* __restrict__ x2, int64_t n, int64_t spt,
                                                          int64_t sxy, int normalise_w, float* __restrict__ X4) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    double At[4][M];   // At[k][row]: column k of the DLT matrix
    dlt_build<M>(At, P.p[0], P.p[1], (double)x1[i * spt], (double)x1[i * spt + sxy], (double)x2[i * spt],
                 (double)x2[i * spt + sxy]);
    double Xd[4];
    dlt_nullvec<M>(At, Xd);
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
