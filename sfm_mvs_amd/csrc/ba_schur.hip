// Matrix-free products with the camera-point coupling W of the dense-visibility bundle-adjustment normal equations
//
//     [ B   W ] [dc]   [g_c]          B_i = sum_j Jc_ij^T Jc_ij  (6x6 per camera)      W_ij = Jc_ij^T Jp_ij  (6x3)
//     [ W^T C ] [dp] = [g_p]          C_j = sum_i Jp_ij^T Jp_ij  (3x3 per point)
//
// — what a Schur-complement solver needs on top of sfm_ba_dense_sweep (which yields B, C, g_c, g_p): the reduced camera
// system S dc = g_c - W C^-1 g_p with S = B - W C^-1 W^T is solved by conjugate gradients, and every S x costs
//     u = W^T x   (sfm_ba_schur_wt: per point   u_j = sum_i Jp_ij^T (Jc_ij x_i))
//     w = W v     (sfm_ba_schur_w : per camera  w_i = sum_j Jc_ij^T (Jp_ij v_j)),   v_j = C_j^-1 u_j
// The 1e8 W blocks of config 4 (19 GB in fp64) are never formed: Jc and Jp depend on the cameras and points only — not
// on the observations — so a product re-derives them in registers and reads NO observation data; it is pure fp64 VALU
// work (~150 DP instructions per camera-point pair and product).
// SURVEY 8f-3 ("real sparse BA solver ... Schur complement + PCG on the J^T J blocks the sweep already emits");
// replaces the reference's scipy least_squares call with finite differences (sfm.py:138-157).
//
// Decomposition as in ba_dense.hip: 256 lanes x PP points per workgroup, blockIdx.y = camera chunk; per-point sums live
// in registers across the camera loop, per-camera sums are folded through LDS in a fixed order → deterministic.
#include "common.h"
#include <cfloat>

namespace {

constexpr int kCamStride = 40;
constexpr int kSegStride = 33;     // doubles; 32 lanes + 1 pad
constexpr int kValStride = 8 * kSegStride + 1;

struct Intrin {
    double fx, fy, cx, cy;
};

__device__ void rodrigues_schur(const double* __restrict__ rv, double* __restrict__ R, double* __restrict__ J) {
    const double theta = sqrt(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]);
    if (theta < DBL_EPSILON) {
        for (int k = 0; k < 9; ++k) R[k] = (k % 4 == 0) ? 1.0 : 0.0;
        for (int k = 0; k < 27; ++k) J[k] = 0;
        J[5] = J[15] = J[19] = -1;
        J[7] = J[11] = J[21] = 1;
        return;
    }
    const double c = cos(theta), s = sin(theta), c1 = 1. - c, itheta = 1. / theta;
    const double r[3] = {rv[0] * itheta, rv[1] * itheta, rv[2] * itheta};
    const double rrt[9] = {r[0] * r[0], r[0] * r[1], r[0] * r[2], r[0] * r[1], r[1] * r[1],
                           r[1] * r[2], r[0] * r[2], r[1] * r[2], r[2] * r[2]};
    const double rx[9] = {0, -r[2], r[1], r[2], 0, -r[0], -r[1], r[0], 0};
    for (int k = 0; k < 9; ++k) R[k] = c * ((k % 4 == 0) ? 1.0 : 0.0) + c1 * rrt[k] + s * rx[k];
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

// camera table: R (9), t (3), dR/dr (27), pad — same layout as ba_dense.hip
__global__ void schur_cam_prepare_kernel(const double* __restrict__ cams, int64_t ncam, double* __restrict__ table) {
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c >= ncam) return;
    double R[9], J[27];
    rodrigues_schur(cams + 6 * c, R, J);
    double* e = table + c * kCamStride;
    for (int k = 0; k < 9; ++k) e[k] = R[k];
    for (int k = 0; k < 3; ++k) e[9 + k] = cams[6 * c + 3 + k];
    for (int k = 0; k < 27; ++k) e[12 + k] = J[k];
    e[39] = 0;
}

// The two 2-row Jacobians of one camera-point pair: Ju, Jv (d(u,v)/d(rvec,tvec), 6 each) and Pu, Pv (d(u,v)/dX, 3 each),
// exactly the expressions of ba_dense_kernel.
__device__ __forceinline__ void pair_jacobians(const double* __restrict__ e, const Intrin& K, double Xw, double Yw, double Zw,
                                               double (&Ju)[6], double (&Jv)[6], double (&Pu)[3], double (&Pv)[3]) {
    double x = e[0] * Xw + e[1] * Yw + e[2] * Zw + e[9];
    double y = e[3] * Xw + e[4] * Yw + e[5] * Zw + e[10];
    double z = e[6] * Xw + e[7] * Yw + e[8] * Zw + e[11];
    z = z != 0.0 ? 1. / z : 1.;
    x *= z;
    y *= z;
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
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        Pu[k] = K.fx * (z * (e[k] - x * e[6 + k]));
        Pv[k] = K.fy * (z * (e[3 + k] - y * e[6 + k]));
    }
}

// u_j (partial over a camera chunk) = sum_i Jp_ij^T (Jc_ij x_i)
template <int PP>
__global__ __launch_bounds__(256) void schur_wt_kernel(const double* __restrict__ table, Intrin K, int ncam,
                                                       const float* __restrict__ X, int64_t npt, int64_t ldx,
                                                       const double* __restrict__ xc /*[ncam][6]*/, int nch,
                                                       double* __restrict__ pt_part /*[nch][npt][3]*/) {
    const int tid = threadIdx.x;
    const int tile = blockIdx.x, ch = blockIdx.y;
    const int c_begin = (int)((int64_t)ncam * ch / nch), c_end = (int)((int64_t)ncam * (ch + 1) / nch);
    const int64_t p0 = (int64_t)tile * (256 * PP) + tid;
    double Xw[PP], Yw[PP], Zw[PP], acc[PP][3];
    bool live[PP];
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
        const int64_t p = p0 + 256 * pp;
        live[pp] = p < npt;
        const int64_t ps = live[pp] ? p : 0;
        Xw[pp] = X[ps * ldx];
        Yw[pp] = X[ps * ldx + 1];
        Zw[pp] = X[ps * ldx + 2];
        acc[pp][0] = acc[pp][1] = acc[pp][2] = 0;
    }
    for (int c = c_begin; c < c_end; ++c) {
        const double* __restrict__ e = table + (int64_t)c * kCamStride;
        const double* __restrict__ xi = xc + (int64_t)c * 6;
#pragma unroll
        for (int pp = 0; pp < PP; ++pp) {
            if (!live[pp]) continue;
            double Ju[6], Jv[6], Pu[3], Pv[3];
            pair_jacobians(e, K, Xw[pp], Yw[pp], Zw[pp], Ju, Jv, Pu, Pv);
            double tu = 0, tv = 0;
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                tu += Ju[a] * xi[a];
                tv += Jv[a] * xi[a];
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) acc[pp][k] += Pu[k] * tu + Pv[k] * tv;
        }
    }
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
        const int64_t p = p0 + 256 * pp;
        if (live[pp]) {
            double* dst = pt_part + ((int64_t)ch * npt + p) * 3;
            dst[0] = acc[pp][0]; dst[1] = acc[pp][1]; dst[2] = acc[pp][2];
        }
    }
}

__global__ __launch_bounds__(256) void schur_pt_fold_kernel(const double* __restrict__ pt_part, int nch, int64_t npt,
                                                            double* __restrict__ u) {
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= npt) return;
    double s0 = 0, s1 = 0, s2 = 0;
    for (int ch = 0; ch < nch; ++ch) {
        const double* src = pt_part + ((int64_t)ch * npt + p) * 3;
        s0 += src[0]; s1 += src[1]; s2 += src[2];
    }
    u[p * 3 + 0] = s0; u[p * 3 + 1] = s1; u[p * 3 + 2] = s2;
}

// w_i (partial over a point tile) = sum_j Jc_ij^T (Jp_ij v_j)
template <int PP>
__global__ __launch_bounds__(256) void schur_w_kernel(const double* __restrict__ table, Intrin K, int ncam,
                                                      const float* __restrict__ X, int64_t npt, int64_t ldx,
                                                      const double* __restrict__ v /*[npt][3]*/, int nch,
                                                      double* __restrict__ cam_part /*[tiles][ncam][6]*/) {
    extern __shared__ __attribute__((aligned(16))) double red[];   // 6 * kValStride doubles
    const int tid = threadIdx.x;
    const int tile = blockIdx.x, ch = blockIdx.y;
    const int c_begin = (int)((int64_t)ncam * ch / nch), c_end = (int)((int64_t)ncam * (ch + 1) / nch);
    const int64_t p0 = (int64_t)tile * (256 * PP) + tid;
    double Xw[PP], Yw[PP], Zw[PP], vj[PP][3];
    bool live[PP];
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
        const int64_t p = p0 + 256 * pp;
        live[pp] = p < npt;
        const int64_t ps = live[pp] ? p : 0;
        Xw[pp] = X[ps * ldx];
        Yw[pp] = X[ps * ldx + 1];
        Zw[pp] = X[ps * ldx + 2];
        vj[pp][0] = v[ps * 3]; vj[pp][1] = v[ps * 3 + 1]; vj[pp][2] = v[ps * 3 + 2];
    }
    const int wseg = tid >> 5, wlane = tid & 31;   // LDS slot of this lane's contribution
    const int rk = tid >> 3, rseg = tid & 7;       // reducer role: value rk (< 6), segment rseg
    for (int c = c_begin; c < c_end; ++c) {
        const double* __restrict__ e = table + (int64_t)c * kCamStride;
        double cacc[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int pp = 0; pp < PP; ++pp) {
            if (!live[pp]) continue;
            double Ju[6], Jv[6], Pu[3], Pv[3];
            pair_jacobians(e, K, Xw[pp], Yw[pp], Zw[pp], Ju, Jv, Pu, Pv);
            const double su = Pu[0] * vj[pp][0] + Pu[1] * vj[pp][1] + Pu[2] * vj[pp][2];
            const double sv = Pv[0] * vj[pp][0] + Pv[1] * vj[pp][1] + Pv[2] * vj[pp][2];
#pragma unroll
            for (int a = 0; a < 6; ++a) cacc[a] += Ju[a] * su + Jv[a] * sv;
        }
        __syncthreads();   // previous camera's readers are done
#pragma unroll
        for (int k = 0; k < 6; ++k) red[k * kValStride + wseg * kSegStride + wlane] = cacc[k];
        __syncthreads();
        if (rk < 6) {
            const double* src = red + rk * kValStride + rseg * kSegStride;
            double s = 0;
#pragma unroll 8
            for (int i = 0; i < 32; ++i) s += src[i];
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            if (rseg == 0) cam_part[((int64_t)tile * ncam + c) * 6 + rk] = s;
        }
    }
}

// one block per camera: thread k sums partial k over tiles in tile order
__global__ void schur_cam_fold_kernel(const double* __restrict__ cam_part, int tiles, int ncam, double* __restrict__ w) {
    const int c = blockIdx.x, k = threadIdx.x;
    if (k >= 6) return;
    double s = 0;
    for (int t = 0; t < tiles; ++t) s += cam_part[((int64_t)t * ncam + c) * 6 + k];
    w[c * 6 + k] = s;
}

// Indexed (sparse-visibility) variants: one lane per observation (cam_idx[o], pt_idx[o]); sums by fp64 hardware atomics
// (order-dependent in the last bits, like the indexed residual sweep).  MODE 0: u[pt] += Jp^T (Jc x[cam]);
// MODE 1: w[cam] += Jc^T (Jp v[pt]).  Outputs must be zeroed by the caller (the entry points do it).
template <int MODE>
__global__ __launch_bounds__(256) void schur_indexed_kernel(const double* __restrict__ table, Intrin K, const float* __restrict__ X,
                                                           int64_t ldx, const int* __restrict__ cam_idx,
                                                           const int* __restrict__ pt_idx, int64_t nobs,
                                                           const double* __restrict__ in, double* __restrict__ out) {
    const int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (o >= nobs) return;
    const int64_t ci = cam_idx[o], pi = pt_idx[o];
    double Ju[6], Jv[6], Pu[3], Pv[3];
    pair_jacobians(table + ci * kCamStride, K, X[pi * ldx], X[pi * ldx + 1], X[pi * ldx + 2], Ju, Jv, Pu, Pv);
    if (MODE == 0) {
        const double* xi = in + ci * 6;
        double tu = 0, tv = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            tu += Ju[a] * xi[a];
            tv += Jv[a] * xi[a];
        }
#pragma unroll
        for (int k = 0; k < 3; ++k) unsafeAtomicAdd(&out[pi * 3 + k], Pu[k] * tu + Pv[k] * tv);
    } else {
        const double* vj = in + pi * 3;
        const double su = Pu[0] * vj[0] + Pu[1] * vj[1] + Pu[2] * vj[2];
        const double sv = Pv[0] * vj[0] + Pv[1] * vj[1] + Pv[2] * vj[2];
#pragma unroll
        for (int a = 0; a < 6; ++a) unsafeAtomicAdd(&out[ci * 6 + a], Ju[a] * su + Jv[a] * sv);
    }
}

struct SchurPlan {
    int pp, tiles, nch;
};

SchurPlan schur_plan(int64_t ncam, int64_t npt, bool cam_side) {
    SchurPlan d;
    d.pp = npt >= 64 * 1024 ? 4 : 2;
    d.tiles = (int)((npt + 256 * d.pp - 1) / (256 * d.pp));
    int nch = d.tiles > 0 ? (1024 + d.tiles - 1) / d.tiles : 1;
    if (nch > ncam / 16) nch = (int)(ncam / 16);
    if (nch < 1) nch = 1;
    if (nch > 64) nch = 64;
    d.nch = nch;
    (void)cam_side;
    return d;
}

struct SchurWs {
    double *table, *part;
    size_t bytes;
};

SchurWs schur_carve(void* base, int64_t ncam, int64_t npt, const SchurPlan& d) {
    sfm::Carver c(base);
    SchurWs w;
    w.table = c.take<double>((size_t)ncam * kCamStride);
    const size_t pt = (size_t)d.nch * npt * 3, cam = (size_t)d.tiles * ncam * 6;
    w.part = c.take<double>(pt > cam ? pt : cam);
    w.bytes = c.used();
    return w;
}

int schur_common(const char* who, const double* cams, int64_t ncam, const double* K_host, const float* X, int64_t npt, int64_t ldx,
                 const void* in, void* out, void* ws, size_t ws_bytes) {
    SFM_CHECK_ARG(ncam >= 1 && npt >= 1 && ldx >= 3 && ncam < (1 << 24), "%s: bad sizes", who);
    SFM_CHECK_ARG(cams && K_host && X && in && out, "%s: null pointer", who);
    const size_t need = sfm_ba_schur_ws_bytes(ncam, npt);
    if (!ws || ws_bytes < need) {
        sfm::set_error("%s: workspace too small (%zu < %zu)", who, ws_bytes, need);
        return SFM_ERR_WORKSPACE;
    }
    return SFM_OK;
}

}  // namespace

extern "C" size_t sfm_ba_schur_ws_bytes(int64_t ncam, int64_t npt) {
    if (ncam < 1 || npt < 1) return 0;
    return schur_carve(nullptr, ncam, npt, schur_plan(ncam, npt, false)).bytes + 256;
}

extern "C" int sfm_ba_schur_wt(const double* cams, int64_t ncam, const double* K_host, const float* X, int64_t npt, int64_t ldx,
                               const double* x_cam, double* u_pt, void* ws, size_t ws_bytes, void* stream_) {
    const int rc = schur_common("sfm_ba_schur_wt", cams, ncam, K_host, X, npt, ldx, x_cam, u_pt, ws, ws_bytes);
    if (rc != SFM_OK) return rc;
    hipStream_t stream = sfm::as_stream(stream_);
    const SchurPlan d = schur_plan(ncam, npt, false);
    const SchurWs w = schur_carve(reinterpret_cast<void*>(sfm::align_up((size_t)(uintptr_t)ws, 256)), ncam, npt, d);
    const Intrin K{K_host[0], K_host[4], K_host[2], K_host[5]};
    hipLaunchKernelGGL(schur_cam_prepare_kernel, dim3((unsigned)((ncam + 63) / 64)), dim3(64), 0, stream, cams, ncam, w.table);
    SFM_CHECK_LAUNCH();
    const dim3 grid((unsigned)d.tiles, (unsigned)d.nch);
    sfm::prof_begin(sfm::kProfBaSchur, stream);
    if (d.pp == 4)
        hipLaunchKernelGGL(schur_wt_kernel<4>, grid, dim3(256), 0, stream, w.table, K, (int)ncam, X, npt, ldx, x_cam, d.nch, w.part);
    else
        hipLaunchKernelGGL(schur_wt_kernel<2>, grid, dim3(256), 0, stream, w.table, K, (int)ncam, X, npt, ldx, x_cam, d.nch, w.part);
    sfm::prof_end(sfm::kProfBaSchur, stream);
    SFM_CHECK_LAUNCH();
    hipLaunchKernelGGL(schur_pt_fold_kernel, dim3((unsigned)((npt + 255) / 256)), dim3(256), 0, stream, w.part, d.nch, npt, u_pt);
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}

extern "C" int sfm_ba_schur_w(const double* cams, int64_t ncam, const double* K_host, const float* X, int64_t npt, int64_t ldx,
                              const double* v_pt, double* w_cam, void* ws, size_t ws_bytes, void* stream_) {
    const int rc = schur_common("sfm_ba_schur_w", cams, ncam, K_host, X, npt, ldx, v_pt, w_cam, ws, ws_bytes);
    if (rc != SFM_OK) return rc;
    hipStream_t stream = sfm::as_stream(stream_);
    const SchurPlan d = schur_plan(ncam, npt, true);
    const SchurWs w = schur_carve(reinterpret_cast<void*>(sfm::align_up((size_t)(uintptr_t)ws, 256)), ncam, npt, d);
    const Intrin K{K_host[0], K_host[4], K_host[2], K_host[5]};
    hipLaunchKernelGGL(schur_cam_prepare_kernel, dim3((unsigned)((ncam + 63) / 64)), dim3(64), 0, stream, cams, ncam, w.table);
    SFM_CHECK_LAUNCH();
    const size_t lds = (size_t)6 * kValStride * sizeof(double);
    const dim3 grid((unsigned)d.tiles, (unsigned)d.nch);
    sfm::prof_begin(sfm::kProfBaSchur, stream);
    if (d.pp == 4)
        hipLaunchKernelGGL(schur_w_kernel<4>, grid, dim3(256), lds, stream, w.table, K, (int)ncam, X, npt, ldx, v_pt, d.nch, w.part);
    else
        hipLaunchKernelGGL(schur_w_kernel<2>, grid, dim3(256), lds, stream, w.table, K, (int)ncam, X, npt, ldx, v_pt, d.nch, w.part);
    sfm::prof_end(sfm::kProfBaSchur, stream);
    SFM_CHECK_LAUNCH();
    hipLaunchKernelGGL(schur_cam_fold_kernel, dim3((unsigned)ncam), dim3(64), 0, stream, w.part, d.tiles, (int)ncam, w_cam);
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}

// Sparse visibility: observation o sees point pt_idx[o] from camera cam_idx[o].  mode 0: out = W^T in (in [ncam x 6],
// out [npt x 3]); mode 1: out = W in (in [npt x 3], out [ncam x 6]).  ws_dev: ncam * 40 doubles (+256 B).
extern "C" size_t sfm_ba_schur_indexed_ws_bytes(int64_t ncam) {
    return ncam < 1 ? 0 : (size_t)ncam * kCamStride * sizeof(double) + 512;
}

extern "C" int sfm_ba_schur_indexed(const double* cams, int64_t ncam, const double* K_host, const float* X, int64_t npt,
                                    int64_t ldx, const int32_t* cam_idx, const int32_t* pt_idx, int64_t nobs, int mode,
                                    const double* in, double* out, void* ws, size_t ws_bytes, void* stream_) {
    SFM_CHECK_ARG(ncam >= 1 && npt >= 1 && ldx >= 3 && nobs >= 0 && (mode == 0 || mode == 1), "sfm_ba_schur_indexed: bad sizes / mode");
    SFM_CHECK_ARG(cams && K_host && X && in && out && (nobs == 0 || (cam_idx && pt_idx)), "sfm_ba_schur_indexed: null pointer");
    if (!ws || ws_bytes < sfm_ba_schur_indexed_ws_bytes(ncam)) {
        sfm::set_error("sfm_ba_schur_indexed: workspace too small (%zu < %zu)", ws_bytes, sfm_ba_schur_indexed_ws_bytes(ncam));
        return SFM_ERR_WORKSPACE;
    }
    hipStream_t stream = sfm::as_stream(stream_);
    double* table = reinterpret_cast<double*>(sfm::align_up((size_t)(uintptr_t)ws, 256));
    const Intrin K{K_host[0], K_host[4], K_host[2], K_host[5]};
    SFM_CHECK_HIP(hipMemsetAsync(out, 0, sizeof(double) * (size_t)(mode == 0 ? npt * 3 : ncam * 6), stream));
    if (nobs == 0) return SFM_OK;
    hipLaunchKernelGGL(schur_cam_prepare_kernel, dim3((unsigned)((ncam + 63) / 64)), dim3(64), 0, stream, cams, ncam, table);
    SFM_CHECK_LAUNCH();
    const dim3 grid((unsigned)((nobs + 255) / 256));
    if (mode == 0)
        hipLaunchKernelGGL(schur_indexed_kernel<0>, grid, dim3(256), 0, stream, table, K, X, ldx, cam_idx, pt_idx, nobs, in, out);
    else
        hipLaunchKernelGGL(schur_indexed_kernel<1>, grid, dim3(256), 0, stream, table, K, X, ldx, cam_idx, pt_idx, nobs, in, out);
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}
