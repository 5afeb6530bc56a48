// Reprojection residual / Gauss-Newton sweep and RANSAC scoring kernels.
//   A5  ReprojectionError  = Rodrigues + projectPoints + norm          sfm.py:79-100
//   A6  solvePnPRansac: model scoring + LM normal equations            sfm.py:67
//   A7  findEssentialMat: Sampson scoring                              sfm.py:307
//   A8  OptimReprojectionError / BundleAdjustment residual sweep       sfm.py:104-157
//
// Projection is OpenCV's distortion-free projectPoints in fp64:
//   X' = R(rvec) X + t;  z = 1/Z' (1 if Z' == 0);  x = X' z;  y = Y' z;  u = x fx + cx;  v = y fy + cy.
// Jacobians are the analytic dp/d(rvec,tvec) and dp/dX of that formula (dR/drvec from the Rodrigues
// closed form).  One lane = one observation; cameras are expanded once per call into a table
// (R, t, dR/dr) by cam_prepare_kernel so the sweep itself has no transcendental math.
#include "common.h"
#include <cfloat>

namespace {

constexpr int kCamStride = 40;   // 9 R + 3 t + 27 dR/dr (+1 pad) doubles

struct Intrin {
    double fx, fy, cx, cy;
};

__device__ void rodrigues_dev(const double* __restrict__ rv, double* __restrict__ R, double* __restrict__ J) {
    const double theta = sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    if (theta < DBL_EPSILON) {
        for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
        if (J) {
            for (int k = 0; k < 27; ++k) J[k] = 0;
            J[5] = J[15] = J[19] = -1;
            J[7] = J[11] = J[21] = 1;
        }
        return;
    }
    const double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = 1. / theta;
    const double r[3] = {rv[0] * itheta, rv[1] * itheta, rv[2] * itheta};
    const double rrt[9] = {r[0] * r[0], r[0] * r[1], r[0] * r[2], r[0] * r[1], r[1] * r[1],
                           r[1] * r[2], r[0] * r[2], r[1] * r[2], r[2] * r[2]};
    const double rx[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
    for (int k = 0; k < 9; ++k) R[k] = c * ((k % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[k] + s * rx[k];
    if (J) {
        const double drrt[27] = {r[0] + r[0], r[1], r[2], r[1], 0, 0, r[2], 0, 0,
                                 0, r[0], 0, r[0], r[1] + r[1], r[2], 0, r[2], 0,
                                 0, 0, r[0], 0, 0, r[1], r[0], r[1], r[2] + r[2]};
        const double drx[27] = {0, 0, 0, 0, 0, -1, 0, 1, 0, 0, 0, 1, 0, 0, 0, -1, 0, 0, 0, -1, 0, 1, 0, 0, 0, 0, 0};
        for (int i = 0; i < 3; ++i) {
            const double ri = r[i];
            const double a0 = -s * ri, a1 = (s - 2 * c1 * itheta) * ri, a2 = c1 * itheta;
            const double a3 = (c - s * itheta) * ri, a4 = s * itheta;
            for (int k = 0; k < 9; ++k)
                J[i * 9 + k] = a0 * ((k % 4 == 0) ? 1.0 : 0.0) + a1 * rrt[k] + a2 * drrt[i * 9 + k] + a3 * rx[k] +
                               a4 * drx[i * 9 + k];
        }
    }
}

__global__ void cam_prepare_kernel(const double* __restrict__ cams, int64_t ncam, double* __restrict__ table) {
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c >= ncam) return;
    double R[9], J[27];
    rodrigues_dev(cams + 6 * c, R, J);
    double* e = table + c * kCamStride;
    for (int k = 0; k < 9; ++k) e[k] = R[k];
    for (int k = 0; k < 3; ++k) e[9 + k] = cams[6 * c + 3 + k];
    for (int k = 0; k < 27; ++k) e[12 + k] = J[k];
}

__device__ __forceinline__ void project(const double* __restrict__ e, const Intrin& K, double Xw, double Yw, double Zw,
                                        double& u, double& v, double& x, double& y, double& z) {
    x = e[0] * Xw + e[1] * Yw + e[2] * Zw + e[9];
    y = e[3] * Xw + e[4] * Yw + e[5] * Zw + e[10];
    z = e[6] * Xw + e[7] * Yw + e[8] * Zw + e[11];
    z = z != 0.0 ? 1. / z : 1.;
    x *= z;
    y *= z;
    u = x * K.fx + K.cx;
    v = y * K.fy + K.cy;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, 64);
    return v;
}

constexpr int kNAcc = 29;   // 21 upper-triangular JtJ + 6 Jtr + sumsq (f32 diff, the metric) + fp64 |proj-obs|^2

// One lane per observation.  MODE 0: projection / residual / inliers only.  MODE 1: + Gauss-Newton blocks.
// single_cam: every observation belongs to camera 0 → the 6x6 block is reduced in fixed order
// (wave shuffle tree → 4 waves → per-block partial → final_reduce_kernel), no atomics.
template <int MODE>
__global__ __launch_bounds__(256) void residual_kernel(
    const double* __restrict__ table, Intrin K, const float* __restrict__ X, int64_t ldx, const float* __restrict__ obs,
    const int* __restrict__ cam_idx, const int* __restrict__ pt_idx, int64_t nobs, float* __restrict__ proj,
    unsigned char* __restrict__ inlier, float thr2, int single_cam, double* __restrict__ partials /*[grid][kNAcc]*/,
    double* __restrict__ JtJ_cam, double* __restrict__ Jtr_cam, double* __restrict__ JtJ_pt,
    double* __restrict__ Jtr_pt) {
    __shared__ double wacc[4][kNAcc];
    const int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    const bool live = o < nobs;
    double acc[kNAcc];
#pragma unroll
    for (int k = 0; k < kNAcc; ++k) acc[k] = 0;

    if (live) {
        const int64_t ci = cam_idx ? cam_idx[o] : 0;
        const int64_t pi = pt_idx ? pt_idx[o] : o;
        const double* e = table + ci * kCamStride;
        const double Xw = X[pi * ldx], Yw = X[pi * ldx + 1], Zw = X[pi * ldx + 2];
        const float ox = obs[2 * o], oy = obs[2 * o + 1];
        double u, v, x, y, z;
        project(e, K, Xw, Yw, Zw, u, v, x, y, z);
        const float pu = (float)u, pv = (float)v;
        if (proj) {
            proj[2 * o] = pu;
            proj[2 * o + 1] = pv;
        }
        const float dxf = pu - ox, dyf = pv - oy;
        acc[27] = (double)dxf * (double)dxf + (double)dyf * (double)dyf;
        if (inlier) {
            const float ex = ox - pu, ey = oy - pv;
            const float err = (float)((double)ex * (double)ex + (double)ey * (double)ey);
            inlier[o] = err <= thr2 ? 1 : 0;
        }
        if (MODE == 1) {
            const double ru = u - (double)ox, rv = v - (double)oy;
            acc[28] = ru * ru + rv * rv;
            double Ju[6], Jv[6];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double* d = e + 12 + 9 * j;
                const double dx0 = Xw * d[0] + Yw * d[1] + Zw * d[2];
                const double dy0 = Xw * d[3] + Yw * d[4] + Zw * d[5];
                const double dz0 = Xw * d[6] + Yw * d[7] + Zw * d[8];
                Ju[j] = K.fx * (z * (dx0 - x * dz0));
                Jv[j] = K.fy * (z * (dy0 - y * dz0));
            }
            Ju[3] = K.fx * z; Ju[4] = 0;        Ju[5] = K.fx * (-x * z);
            Jv[3] = 0;        Jv[4] = K.fy * z; Jv[5] = K.fy * (-y * z);
            if (JtJ_pt || Jtr_pt) {
                double Pu[3], Pv[3];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    Pu[k] = K.fx * (z * (e[k] - x * e[6 + k]));
                    Pv[k] = K.fy * (z * (e[3 + k] - y * e[6 + k]));
                }
                if (JtJ_pt)
#pragma unroll
                    for (int a = 0; a < 3; ++a)
#pragma unroll
                        for (int b = 0; b < 3; ++b) unsafeAtomicAdd(&JtJ_pt[pi * 9 + a * 3 + b], Pu[a] * Pu[b] + Pv[a] * Pv[b]);
                if (Jtr_pt)
#pragma unroll
                    for (int a = 0; a < 3; ++a) unsafeAtomicAdd(&Jtr_pt[pi * 3 + a], Pu[a] * ru + Pv[a] * rv);
            }
            if (single_cam) {
                int q = 0;
#pragma unroll
                for (int a = 0; a < 6; ++a)
#pragma unroll
                    for (int b = a; b < 6; ++b) acc[q++] = Ju[a] * Ju[b] + Jv[a] * Jv[b];
#pragma unroll
                for (int a = 0; a < 6; ++a) acc[21 + a] = Ju[a] * ru + Jv[a] * rv;
            } else {
                if (JtJ_cam)
#pragma unroll
                    for (int a = 0; a < 6; ++a)
#pragma unroll
                        for (int b = 0; b < 6; ++b) unsafeAtomicAdd(&JtJ_cam[ci * 36 + a * 6 + b], Ju[a] * Ju[b] + Jv[a] * Jv[b]);
                if (Jtr_cam)
#pragma unroll
                    for (int a = 0; a < 6; ++a) unsafeAtomicAdd(&Jtr_cam[ci * 6 + a], Ju[a] * ru + Jv[a] * rv);
            }
        }
    }

    // fixed-order block reduction → one partial row per block
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    constexpr int kFirst = (MODE == 1) ? 0 : 27;
    constexpr int kLast = (MODE == 1) ? kNAcc : 28;
#pragma unroll
    for (int k = kFirst; k < kLast; ++k) {
        const double s = wave_sum(acc[k]);
        if (lane == 0) wacc[wave][k] = s;
    }
    __syncthreads();
    if (threadIdx.x < kLast && threadIdx.x >= kFirst) {
        const int k = threadIdx.x;
        partials[(int64_t)blockIdx.x * kNAcc + k] = ((wacc[0][k] + wacc[1][k]) + wacc[2][k]) + wacc[3][k];
    }
}

// Folds partial rows (one per residual workgroup, kNAcc doubles each) in a FIXED two-level shape — the result depends on
// the row count only, never on timing.  1024 threads = 32 groups x 32 entry lanes: group g adds rows g, g + 32, g + 64 ...
// of its workgroup's chunk in ascending order (a group reads one whole 232-byte row per step: coalesced, and the loads of
// a group's next rows do not depend on its running sums), then the 32 group sums of an entry are added pairwise
// (stride 16, 8, 4, 2, 1).  Round 2 walked all rows with one thread per entry: 167 us at 782 rows (200k points), 33 %
// of a 500-call trace whose residual kernel takes 5 us; this is one 1024-thread workgroup for <= kReduceChunk rows
// (25 dependent adds per thread at 782 rows) and a first level of ceil(rows / kReduceChunk) workgroups above that.
constexpr int kReduceChunk = 2048;   // rows folded by one workgroup

__device__ __forceinline__ double fold_rows(const double* __restrict__ rows, int nrows, double (*lds)[32]) {
    const int k = threadIdx.x & 31, g = threadIdx.x >> 5;
    double s = 0;
    if (k < kNAcc)
        for (int r = g; r < nrows; r += 32) s += rows[(int64_t)r * kNAcc + k];
    lds[g][k] = s;
    __syncthreads();
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
        if (g < m) lds[g][k] = lds[g][k] + lds[g + m][k];
        __syncthreads();
    }
    return lds[0][k];
}

__global__ __launch_bounds__(1024) void reduce_rows_kernel(const double* __restrict__ partials, int nrows, double* __restrict__ out_rows) {
    __shared__ double lds[32][32];
    const int r0 = blockIdx.x * kReduceChunk;
    const double s = fold_rows(partials + (int64_t)r0 * kNAcc, min(kReduceChunk, nrows - r0), lds);
    if (threadIdx.x < kNAcc) out_rows[(int64_t)blockIdx.x * kNAcc + threadIdx.x] = s;
}

__global__ __launch_bounds__(1024) void final_reduce_kernel(const double* __restrict__ partials, int nblocks, int with_jac,
                                                            double* __restrict__ sumsq, double* __restrict__ JtJ_cam, double* __restrict__ Jtr_cam,
                                                            double* __restrict__ res2) {
    __shared__ double lds[32][32];
    const double s = fold_rows(partials, nblocks, lds);
    const int k = threadIdx.x;
    if (k >= kNAcc) return;
    if (!with_jac && k != 27) return;
    if (k == 28) {
        if (res2) *res2 += s;
    } else if (k == 27) {
        if (sumsq) *sumsq += s;
    } else if (k >= 21) {
        if (Jtr_cam) Jtr_cam[k - 21] += s;
    } else if (JtJ_cam) {
        int a = 0, rem = k;   // k-th upper-triangular entry → (a,b)
        while (rem >= 6 - a) { rem -= 6 - a; ++a; }
        const int b = a + rem;
        JtJ_cam[a * 6 + b] += s;
        if (a != b) JtJ_cam[b * 6 + a] += s;
    }
}

// ---------------------------------------------------------------- RANSAC scoring
// grid = (ceil(n/256), h): block (bx, m) scores hypothesis m on 256 points; integer atomics only.
__global__ __launch_bounds__(256) void score_essential_kernel(const double* __restrict__ Es, const double* __restrict__ x1n,
                                                              const double* __restrict__ x2n, int64_t n, float thr2,
                                                              int* __restrict__ counts, unsigned char* __restrict__ mask) {
    const int m = blockIdx.y;
    const double* E = Es + 9 * m;
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    bool in = false;
    if (i < n) {
        const double a1 = x1n[2 * i], b1 = x1n[2 * i + 1], a2 = x2n[2 * i], b2 = x2n[2 * i + 1];
        // Ex1 = E*(x1,1);  Etx2 = E^T*(x2,1);  Sampson = (x2^T E x1)^2 / (Ex1_0^2 + Ex1_1^2 + Etx2_0^2 + Etx2_1^2)
        const double e0 = E[0] * a1 + E[1] * b1 + E[2] * 1., e1 = E[3] * a1 + E[4] * b1 + E[5] * 1.,
                     e2 = E[6] * a1 + E[7] * b1 + E[8] * 1.;
        const double f0 = E[0] * a2 + E[3] * b2 + E[6] * 1., f1 = E[1] * a2 + E[4] * b2 + E[7] * 1.;
        const double x2tEx1 = a2 * e0 + b2 * e1 + 1. * e2;
        const double a = e0 * e0, b = e1 * e1, c = f0 * f0, d = f1 * f1;
        const float err = (float)(x2tEx1 * x2tEx1 / (a + b + c + d));
        in = err <= thr2;
        if (mask) mask[(int64_t)m * n + i] = in ? 1 : 0;
    }
    const int cnt = __popcll(__ballot(in));
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&counts[m], cnt);
}

__global__ __launch_bounds__(256) void score_pnp_kernel(const double* __restrict__ poses, Intrin K,
                                                        const float* __restrict__ X, const float* __restrict__ obs,
                                                        int64_t n, float thr2, int* __restrict__ counts,
                                                        unsigned char* __restrict__ mask) {
    __shared__ double cam[12];
    const int m = blockIdx.y;
    if (threadIdx.x == 0) {
        double R[9];
        rodrigues_dev(poses + 6 * m, R, nullptr);
        for (int k = 0; k < 9; ++k) cam[k] = R[k];
        for (int k = 0; k < 3; ++k) cam[9 + k] = poses[6 * m + 3 + k];
    }
    __syncthreads();
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    bool in = false;
    if (i < n) {
        double u, v, x, y, z;
        project(cam, K, X[3 * i], X[3 * i + 1], X[3 * i + 2], u, v, x, y, z);
        const float ex = obs[2 * i] - (float)u, ey = obs[2 * i + 1] - (float)v;
        const float err = (float)((double)ex * (double)ex + (double)ey * (double)ey);
        in = err <= thr2;
        if (mask) mask[(int64_t)m * n + i] = in ? 1 : 0;
    }
    const int cnt = __popcll(__ballot(in));
    if ((threadIdx.x & 63) == 0 && cnt) atomicAdd(&counts[m], cnt);
}

Intrin make_intrin(const double* K) { return Intrin{K[0], K[4], K[2], K[5]}; }

}  // namespace

extern "C" size_t sfm_project_residual_ws_bytes(int64_t nobs, int64_t ncam, int64_t npt) {
    (void)npt;
    if (nobs < 0 || ncam < 0) return 0;
    const size_t blocks = (size_t)((nobs + 255) / 256);
    const size_t level1 = (blocks + kReduceChunk - 1) / kReduceChunk;
    return sfm::align_up((size_t)ncam * kCamStride * sizeof(double), 256) +
           sfm::align_up((blocks + 1 + level1) * kNAcc * sizeof(double), 256) + 512;
}

extern "C" int sfm_project_residual(const double* cams, int64_t ncam, const double* K_host, const float* X, int64_t npt,
                                    int64_t ldx, const float* obs, const int32_t* cam_idx, const int32_t* pt_idx,
                                    int64_t nobs, float* proj, double* sumsq, uint8_t* inlier, float thr2,
                                    double* JtJ_cam, double* Jtr_cam, double* JtJ_pt, double* Jtr_pt, double* res2,
                                    void* ws, size_t ws_bytes, void* stream_) {
    SFM_CHECK_ARG(ncam >= 1 && npt >= 0 && nobs >= 0 && ldx >= 3, "sfm_project_residual: bad sizes");
    SFM_CHECK_ARG(cams && K_host, "sfm_project_residual: null camera/intrinsics");
    if (nobs == 0) return SFM_OK;
    SFM_CHECK_ARG(X && obs, "sfm_project_residual: null X/obs");
    SFM_CHECK_ARG(pt_idx || nobs <= npt, "sfm_project_residual: pt_idx NULL requires nobs <= npt");
    SFM_CHECK_ARG(nobs <= (int64_t)256 * kReduceChunk * kReduceChunk, "sfm_project_residual: more than 2^30 observations per call");
    const size_t need = sfm_project_residual_ws_bytes(nobs, ncam, npt);
    if (!ws || ws_bytes < need) {
        sfm::set_error("sfm_project_residual: workspace too small (%zu < %zu)", ws_bytes, need);
        return SFM_ERR_WORKSPACE;
    }
    hipStream_t stream = sfm::as_stream(stream_);
    sfm::Carver c(reinterpret_cast<void*>(sfm::align_up((size_t)(uintptr_t)ws, 256)));
    double* table = c.take<double>((size_t)ncam * kCamStride);
    const int blocks = (int)((nobs + 255) / 256);
    const int level1 = (blocks + kReduceChunk - 1) / kReduceChunk;
    double* partials = c.take<double>((size_t)(blocks + 1 + level1) * kNAcc);

    hipLaunchKernelGGL(cam_prepare_kernel, dim3((unsigned)((ncam + 63) / 64)), dim3(64), 0, stream, cams, ncam, table);
    SFM_CHECK_LAUNCH();
    const bool jac = JtJ_cam || Jtr_cam || JtJ_pt || Jtr_pt || res2;
    const int single = (cam_idx == nullptr) ? 1 : 0;
    const Intrin K = make_intrin(K_host);
    if (jac)
        hipLaunchKernelGGL(residual_kernel<1>, dim3(blocks), dim3(256), 0, stream, table, K, X, ldx, obs, cam_idx, pt_idx,
                           nobs, proj, inlier, thr2, single, partials, JtJ_cam, Jtr_cam, JtJ_pt, Jtr_pt);
    else
        hipLaunchKernelGGL(residual_kernel<0>, dim3(blocks), dim3(256), 0, stream, table, K, X, ldx, obs, cam_idx, pt_idx,
                           nobs, proj, inlier, thr2, single, partials, JtJ_cam, Jtr_cam, JtJ_pt, Jtr_pt);
    SFM_CHECK_LAUNCH();
    if (sumsq || res2 || (jac && single)) {
        const double* rows = partials;
        int nrows = blocks;
        if (level1 > 1) {                                  // (level1 <= kReduceChunk up to 1e9 observations: checked above)
            double* rows1 = partials + (size_t)(blocks + 1) * kNAcc;
            hipLaunchKernelGGL(reduce_rows_kernel, dim3((unsigned)level1), dim3(1024), 0, stream, partials, blocks, rows1);
            SFM_CHECK_LAUNCH();
            rows = rows1;
            nrows = level1;
        }
        hipLaunchKernelGGL(final_reduce_kernel, dim3(1), dim3(1024), 0, stream, rows, nrows, jac ? 1 : 0, sumsq,
                           single ? JtJ_cam : nullptr, single ? Jtr_cam : nullptr, res2);
        SFM_CHECK_LAUNCH();
    }
    return SFM_OK;
}

extern "C" int sfm_score_essential(const double* E, int h, const double* x1n, const double* x2n, int64_t n, float thr2,
                                   int32_t* counts, uint8_t* mask, void* stream_) {
    SFM_CHECK_ARG(h >= 0 && n >= 0 && h <= 65535, "sfm_score_essential: bad sizes");
    if (h == 0) return SFM_OK;
    SFM_CHECK_ARG(E && counts && (n == 0 || (x1n && x2n)), "sfm_score_essential: null pointer");
    hipStream_t stream = sfm::as_stream(stream_);
    SFM_CHECK_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)h, stream));
    if (n == 0) return SFM_OK;
    hipLaunchKernelGGL(score_essential_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)h), dim3(256), 0, stream, E, x1n,
                       x2n, n, thr2, counts, mask);
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}

extern "C" int sfm_score_pnp(const double* poses, int h, const double* K_host, const float* X, const float* obs, int64_t n,
                             float thr2, int32_t* counts, uint8_t* mask, void* stream_) {
    SFM_CHECK_ARG(h >= 0 && n >= 0 && h <= 65535, "sfm_score_pnp: bad sizes");
    if (h == 0) return SFM_OK;
    SFM_CHECK_ARG(poses && K_host && counts && (n == 0 || (X && obs)), "sfm_score_pnp: null pointer");
    hipStream_t stream = sfm::as_stream(stream_);
    SFM_CHECK_HIP(hipMemsetAsync(counts, 0, sizeof(int32_t) * (size_t)h, stream));
    if (n == 0) return SFM_OK;
    hipLaunchKernelGGL(score_pnp_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)h), dim3(256), 0, stream, poses,
                       make_intrin(K_host), X, obs, n, thr2, counts, mask);
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}
