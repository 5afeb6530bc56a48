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

// This file's sums are checked against the oracle to 1e-10, not bit for bit (unlike the KNN refine, the triangulation and
// the SIFT kernels, for which the library is built with -ffp-contract=off): let the compiler fuse a * b + c here — the
// sweeps are bound by the fp64 vector ALU's instruction count, and the unfused forms cost a third more instructions.
#pragma clang fp contract(fast)

namespace {

constexpr int kCamStride = 40;
constexpr int kWaveValStride = 8 * 9 + 1;   // a wave's slab of schur_w_kernel: [value][8 groups x (8 + pad)] + pad

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

// The same Jacobians with the structure of the translation columns spelled out — Ju = (Ju[0..2], fxz, 0, Ju5),
// Jv = (Jv[0..2], 0, fyz, Jv5) — for the dense products below, which never multiply the structural zeros and take 1 / z
// from the hardware estimate + two Newton steps (as ba_dense_kernel; the sums are checked to 1e-10).
struct PairJ {
    double Ju[3], Jv[3], fxz, fyz, Ju5, Jv5, Pu[3], Pv[3];
};
__device__ __forceinline__ PairJ pair_jacobians_structured(const double* __restrict__ e, const Intrin& K, double Xw, double Yw, double Zw) {
    PairJ J;
    double x = e[0] * Xw + e[1] * Yw + e[2] * Zw + e[9];
    double y = e[3] * Xw + e[4] * Yw + e[5] * Zw + e[10];
    double z = e[6] * Xw + e[7] * Yw + e[8] * Zw + e[11];
    if (z != 0.0) {
        double r = __builtin_amdgcn_rcp(z);
        r = fma(fma(-z, r, 1.0), r, r);
        r = fma(fma(-z, r, 1.0), r, r);
        z = r;
    } else {
        z = 1.;
    }
    x *= z;
    y *= z;
    J.fxz = K.fx * z;
    J.fyz = K.fy * z;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const double* d = e + 12 + 9 * j;
        const double dx0 = Xw * d[0] + Yw * d[1] + Zw * d[2];
        const double dy0 = Xw * d[3] + Yw * d[4] + Zw * d[5];
        const double dz0 = Xw * d[6] + Yw * d[7] + Zw * d[8];
        J.Ju[j] = J.fxz * (dx0 - x * dz0);
        J.Jv[j] = J.fyz * (dy0 - y * dz0);
    }
    J.Ju5 = -x * J.fxz;
    J.Jv5 = -y * J.fyz;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        J.Pu[k] = J.fxz * (e[k] - x * e[6 + k]);
        J.Pv[k] = J.fyz * (e[3 + k] - y * e[6 + k]);
    }
    return J;
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
            const PairJ J = pair_jacobians_structured(e, K, Xw[pp], Yw[pp], Zw[pp]);
            double tu = J.fxz * xi[3], tv = J.fyz * xi[4];
            tu = fma(J.Ju5, xi[5], tu);
            tv = fma(J.Jv5, xi[5], tv);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                tu = fma(J.Ju[a], xi[a], tu);
                tv = fma(J.Jv[a], xi[a], tv);
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                acc[pp][k] = fma(J.Pu[k], tu, acc[pp][k]);
                acc[pp][k] = fma(J.Pv[k], tv, acc[pp][k]);
            }
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
                                                      double* __restrict__ cam_part /*[4 * tiles][ncam][6]: a row per wave*/) {
    extern __shared__ __attribute__((aligned(16))) double red[];   // 4 waves x 6 * kWaveValStride doubles
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
    // The six camera-side sums are folded PER WAVE (as in ba_dense_kernel: a wave's own LDS slab, a wave barrier, a partial
    // row per wave — no __syncthreads in the camera loop): lane 8k + g adds partials 8g .. 8g + 7 of value k in order, three
    // exchanges join the eight groups.
    const int lane = tid & 63, wave = tid >> 6;
    double* const wred = red + wave * (6 * kWaveValStride);
    const int rk = lane >> 3, rseg = lane & 7;     // reducer role: value rk (< 6), group rseg
    for (int c = c_begin; c < c_end; ++c) {
        const double* __restrict__ e = table + (int64_t)c * kCamStride;
        double cacc[6] = {0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int pp = 0; pp < PP; ++pp) {
            if (!live[pp]) continue;
            const PairJ J = pair_jacobians_structured(e, K, Xw[pp], Yw[pp], Zw[pp]);
            const double su = J.Pu[0] * vj[pp][0] + J.Pu[1] * vj[pp][1] + J.Pu[2] * vj[pp][2];
            const double sv = J.Pv[0] * vj[pp][0] + J.Pv[1] * vj[pp][1] + J.Pv[2] * vj[pp][2];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                cacc[a] = fma(J.Ju[a], su, cacc[a]);
                cacc[a] = fma(J.Jv[a], sv, cacc[a]);
            }
            cacc[3] = fma(J.fxz, su, cacc[3]);
            cacc[4] = fma(J.fyz, sv, cacc[4]);
            cacc[5] = fma(J.Ju5, su, cacc[5]);
            cacc[5] = fma(J.Jv5, sv, cacc[5]);
        }
#pragma unroll
        for (int k = 0; k < 6; ++k) wred[k * kWaveValStride + (lane >> 3) * 9 + (lane & 7)] = cacc[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (rk < 6) {
            const double* src = wred + rk * kWaveValStride + rseg * 9;
            double s = 0;
#pragma unroll
            for (int i = 0; i < 8; ++i) s += src[i];
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            if (rseg == 0) cam_part[(((int64_t)tile * 4 + wave) * ncam + c) * 6 + rk] = s;
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// One workgroup per camera folds its `tiles` partial 6-vectors in a fixed two-level shape: 256 threads = 32 groups x 8 entry
// lanes, group g adds tiles g, g + 32, ... in ascending order, then the 32 group sums are added pairwise (stride 16 .. 1).
// (Round 2: six threads walking every tile — 48 us of a 0.6 ms product.)
__global__ __launch_bounds__(256) void schur_cam_fold_kernel(const double* __restrict__ cam_part, int tiles, int ncam, double* __restrict__ w) {
    __shared__ double lds[32][8];
    const int c = blockIdx.x, k = threadIdx.x & 7, g = threadIdx.x >> 3;
    double s = 0;
    if (k < 6)
        for (int t = g; t < tiles; t += 32) s += cam_part[((int64_t)t * ncam + c) * 6 + k];
    lds[g][k] = s;
    __syncthreads();
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
        if (g < m) lds[g][k] = lds[g][k] + lds[g + m][k];
        __syncthreads();
    }
    if (threadIdx.x < 6) w[c * 6 + threadIdx.x] = lds[0][threadIdx.x];
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
    d.nch = sfm::pick_camera_chunks(d.tiles, ncam, cam_side ? 1024 : 1280);      // (schur_w: 4 waves per SIMD, schur_wt: 5)
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
    const size_t pt = (size_t)d.nch * npt * 3, cam = (size_t)4 * d.tiles * ncam * 6;
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
    const size_t lds = (size_t)4 * 6 * kWaveValStride * sizeof(double);
    const dim3 grid((unsigned)d.tiles, (unsigned)d.nch);
    sfm::prof_begin(sfm::kProfBaSchur, stream);
    if (d.pp == 4)
        hipLaunchKernelGGL(schur_w_kernel<4>, grid, dim3(256), lds, stream, w.table, K, (int)ncam, X, npt, ldx, v_pt, d.nch, w.part);
    else
        hipLaunchKernelGGL(schur_w_kernel<2>, grid, dim3(256), lds, stream, w.table, K, (int)ncam, X, npt, ldx, v_pt, d.nch, w.part);
    sfm::prof_end(sfm::kProfBaSchur, stream);
    SFM_CHECK_LAUNCH();
    hipLaunchKernelGGL(schur_cam_fold_kernel, dim3((unsigned)ncam), dim3(256), 0, stream, w.part, 4 * d.tiles, (int)ncam, w_cam);
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}

// ---------------------------------------------------------------- the damped Schur step, PCG on the device
// sfm_ba_schur_solve: one Levenberg-Marquardt step of the DENSE problem through the reduced camera system
//     S dc = g_c - W Cd^-1 g_p,   S = Bd - W Cd^-1 W^T,   dp = Cd^-1 (g_p - W^T dc)
// with block-Jacobi-preconditioned conjugate gradients, S applied matrix-free by the two product kernels above.  Round 2
// drove this recurrence from Python (a dozen torch launches per CG iteration: 0.51 s per solve of config 4, a quarter of
// it kernels); here every vector operation of an iteration on the camera side (3000 doubles at config 4) is ONE
// single-workgroup kernel with fixed-order reductions, the scalars (r.z, r.r, the stopping bound) never leave the device,
// and the host only looks at r.r every fifth iteration (one 16-byte read, the cadence of the Python loop it replaces).
namespace {

// Bd = B with the diagonal scaled by (1 + lam); Minv = Bd^-1 (Gauss-Jordan with partial pivoting, one lane per camera).
// A singular or non-finite pivot (a camera without observations at lam = 0) sets bit 0 of *status and leaves the
// identity as that camera's preconditioner block (the plain inverse produced inf / NaN silently).
__global__ __launch_bounds__(64) void schur_damp_cam_kernel(const double* __restrict__ B, int ncam, double lam, double* __restrict__ Bd,
                                                            double* __restrict__ Minv, int* __restrict__ status) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ncam) return;
    double a[6][6], b[6][6];
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            double v = B[i * 36 + r * 6 + c];
            if (r == c) v *= 1.0 + lam;
            Bd[i * 36 + r * 6 + c] = v;
            a[r][c] = v;
            b[r][c] = r == c ? 1.0 : 0.0;
        }
    bool bad = false;
#pragma unroll
    for (int col = 0; col < 6; ++col) {
#pragma unroll
        for (int r = col + 1; r < 6; ++r) {
            const bool sw = fabs(a[r][col]) > fabs(a[col][col]);
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                const double ta = a[col][c], tb = b[col][c];
                a[col][c] = sw ? a[r][c] : ta;
                a[r][c] = sw ? ta : a[r][c];
                b[col][c] = sw ? b[r][c] : tb;
                b[r][c] = sw ? tb : b[r][c];
            }
        }
        const double piv = a[col][col];
        if (!(fabs(piv) > 0.0) || !(fabs(piv) < 1e300)) bad = true;
        const double d = 1.0 / piv;
#pragma unroll
        for (int c = 0; c < 6; ++c) {
            a[col][c] *= d;
            b[col][c] *= d;
        }
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            if (r == col) continue;
            const double f = a[r][col];
#pragma unroll
            for (int c = 0; c < 6; ++c) {
                a[r][c] -= f * a[col][c];
                b[r][c] -= f * b[col][c];
            }
        }
    }
    if (bad) atomicOr(status, 1);
#pragma unroll
    for (int r = 0; r < 6; ++r)
#pragma unroll
        for (int c = 0; c < 6; ++c) Minv[i * 36 + r * 6 + c] = bad ? (r == c ? 1.0 : 0.0) : b[r][c];
}

// Cd^-1 of the damped symmetric 3 x 3 point blocks (adjugate / determinant; a vanishing determinant keeps the adjugate,
// as the round-2 host code did, and sets bit 1 of *status)
__global__ __launch_bounds__(256) void schur_damp_pt_kernel(const double* __restrict__ C, int64_t npt, double lam, double* __restrict__ Cinv,
                                                           int* __restrict__ status) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= npt) return;
    const double* s = C + i * 9;
    const double a = s[0] * (1.0 + lam), b = s[1], c = s[2], d = s[4] * (1.0 + lam), e = s[5], f = s[8] * (1.0 + lam);
    const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    const double c11 = a * f - c * c, c12 = b * c - a * e, c22 = a * d - b * b;
    const double det = a * c00 + b * c01 + c * c02;
    const bool ok = fabs(det) > 1e-300;
    if (!ok) atomicOr(status, 2);
    const double inv = 1.0 / (ok ? det : 1.0);
    double* o = Cinv + i * 9;
    o[0] = c00 * inv; o[1] = c01 * inv; o[2] = c02 * inv;
    o[3] = c01 * inv; o[4] = c11 * inv; o[5] = c12 * inv;
    o[6] = c02 * inv; o[7] = c12 * inv; o[8] = c22 * inv;
}

// y_j = Cinv_j (sign * x_j + g_j)   (g may be null)
__global__ __launch_bounds__(256) void schur_pt_apply_kernel(const double* __restrict__ Cinv, const double* __restrict__ x, const double* __restrict__ g,
                                                            double sign, int64_t npt, double* __restrict__ y) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= npt) return;
    double v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) v[k] = sign * x[i * 3 + k] + (g ? g[i * 3 + k] : 0.0);
    const double* m = Cinv + i * 9;
#pragma unroll
    for (int r = 0; r < 3; ++r) y[i * 3 + r] = m[3 * r] * v[0] + m[3 * r + 1] * v[1] + m[3 * r + 2] * v[2];
}

constexpr int kCgThreads = 1024;
enum { kScalRz = 0, kScalRr = 1, kScalStop2 = 2, kScalWords = 8 };

// sum over the workgroup in a fixed shape (lane-strided partial sums were formed in element order by the caller)
__device__ __forceinline__ double cg_block_sum(double v, double* lds) {
    lds[threadIdx.x] = v;
    __syncthreads();
    for (int m = kCgThreads / 2; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) lds[threadIdx.x] += lds[threadIdx.x + m];
        __syncthreads();
    }
    const double s = lds[0];
    __syncthreads();
    return s;
}

__device__ __forceinline__ double block6_row(const double* __restrict__ M, const double* __restrict__ v, int e) {
    const int c = e / 6, r = e - 6 * c;
    const double* m = M + (int64_t)c * 36 + 6 * r;
    const double* x = v + (int64_t)c * 6;
    return m[0] * x[0] + m[1] * x[1] + m[2] * x[2] + m[3] * x[3] + m[4] * x[4] + m[5] * x[5];
}

// rhs = (g_c - w0) * free;  x = 0;  r = rhs;  z = Minv r * free;  p = z;  scalars r.z, r.r, stop^2 = tol^2 rhs.rhs
__global__ __launch_bounds__(kCgThreads) void schur_cg_init_kernel(const double* __restrict__ gc, const double* __restrict__ w0,
                                                                  const double* __restrict__ Minv, int n, int fix_first, double tol2,
                                                                  double* __restrict__ x, double* __restrict__ r, double* __restrict__ p,
                                                                  double* __restrict__ scal) {
    __shared__ double lds[kCgThreads];
    double rr = 0;
    for (int e = threadIdx.x; e < n; e += kCgThreads) {
        const double v = (fix_first && e < 6) ? 0.0 : gc[e] - w0[e];
        r[e] = v;
        x[e] = 0.0;
        rr += v * v;
    }
    __threadfence_block();
    __syncthreads();
    rr = cg_block_sum(rr, lds);
    double rz = 0;
    for (int e = threadIdx.x; e < n; e += kCgThreads) {
        const double z = (fix_first && e < 6) ? 0.0 : block6_row(Minv, r, e);
        p[e] = z;
        rz += r[e] * z;
    }
    rz = cg_block_sum(rz, lds);
    if (threadIdx.x == 0) {
        scal[kScalRz] = rz;
        scal[kScalRr] = rr;
        scal[kScalStop2] = tol2 * rr;
    }
}

// One CG iteration on the camera side, given w = W Cd^-1 W^T p:  Sp = (Bd p - w) * free;  alpha = r.z / p.Sp;  x += alpha p;
// r -= alpha Sp;  z = Minv r * free;  beta = r.z' / r.z;  p = z + beta p
__global__ __launch_bounds__(kCgThreads) void schur_cg_step_kernel(const double* __restrict__ Bd, const double* __restrict__ Minv,
                                                                  const double* __restrict__ w, int n, int fix_first, double* __restrict__ x,
                                                                  double* __restrict__ r, double* __restrict__ p, double* __restrict__ tmp,
                                                                  double* __restrict__ scal) {
    __shared__ double lds[kCgThreads];
    double pSp = 0;
    for (int e = threadIdx.x; e < n; e += kCgThreads) {
        const double sp = (fix_first && e < 6) ? 0.0 : block6_row(Bd, p, e) - w[e];
        tmp[e] = sp;
        pSp += p[e] * sp;
    }
    pSp = cg_block_sum(pSp, lds);
    const double rz = scal[kScalRz];
    const double alpha = rz / pSp;
    double rr = 0;
    for (int e = threadIdx.x; e < n; e += kCgThreads) {
        x[e] += alpha * p[e];
        const double v = r[e] - alpha * tmp[e];
        r[e] = v;
        rr += v * v;
    }
    __threadfence_block();
    __syncthreads();
    rr = cg_block_sum(rr, lds);
    double rz_new = 0;
    for (int e = threadIdx.x; e < n; e += kCgThreads) {
        const double z = (fix_first && e < 6) ? 0.0 : block6_row(Minv, r, e);
        tmp[e] = z;
        rz_new += r[e] * z;
    }
    rz_new = cg_block_sum(rz_new, lds);
    const double beta = rz_new / rz;
    for (int e = threadIdx.x; e < n; e += kCgThreads) p[e] = tmp[e] + beta * p[e];
    if (threadIdx.x == 0) {
        scal[kScalRz] = rz_new;
        scal[kScalRr] = rr;
    }
}

struct SolveWs {
    SchurWs prod;
    double *Bd, *Minv, *Cinv, *u, *v, *wv, *r, *p, *tmp, *scal;
    int* status;
    size_t bytes;
};

SolveWs solve_carve(void* base, int64_t ncam, int64_t npt, const SchurPlan& d) {
    SolveWs w;
    w.prod = schur_carve(base, ncam, npt, d);
    sfm::Carver c(static_cast<char*>(base) + w.prod.bytes);
    w.Bd = c.take<double>((size_t)ncam * 36);
    w.Minv = c.take<double>((size_t)ncam * 36);
    w.Cinv = c.take<double>((size_t)npt * 9);
    w.u = c.take<double>((size_t)npt * 3);
    w.v = c.take<double>((size_t)npt * 3);
    w.wv = c.take<double>((size_t)ncam * 6);
    w.r = c.take<double>((size_t)ncam * 6);
    w.p = c.take<double>((size_t)ncam * 6);
    w.tmp = c.take<double>((size_t)ncam * 6);
    w.scal = c.take<double>(kScalWords);
    w.status = c.take<int>(4);
    w.bytes = w.prod.bytes + c.used();
    return w;
}

void launch_wt(hipStream_t stream, const SchurPlan& d, const SchurWs& w, const Intrin& K, int64_t ncam, const float* X, int64_t npt, int64_t ldx,
               const double* x_cam, double* u_pt) {
    const dim3 grid((unsigned)d.tiles, (unsigned)d.nch);
    if (d.pp == 4)
        hipLaunchKernelGGL(schur_wt_kernel<4>, grid, dim3(256), 0, stream, w.table, K, (int)ncam, X, npt, ldx, x_cam, d.nch, w.part);
    else
        hipLaunchKernelGGL(schur_wt_kernel<2>, grid, dim3(256), 0, stream, w.table, K, (int)ncam, X, npt, ldx, x_cam, d.nch, w.part);
    hipLaunchKernelGGL(schur_pt_fold_kernel, dim3((unsigned)((npt + 255) / 256)), dim3(256), 0, stream, w.part, d.nch, npt, u_pt);
}

void launch_w(hipStream_t stream, const SchurPlan& d, const SchurWs& w, const Intrin& K, int64_t ncam, const float* X, int64_t npt, int64_t ldx,
              const double* v_pt, double* w_cam) {
    const size_t lds = (size_t)4 * 6 * kWaveValStride * sizeof(double);
    const dim3 grid((unsigned)d.tiles, (unsigned)d.nch);
    if (d.pp == 4)
        hipLaunchKernelGGL(schur_w_kernel<4>, grid, dim3(256), lds, stream, w.table, K, (int)ncam, X, npt, ldx, v_pt, d.nch, w.part);
    else
        hipLaunchKernelGGL(schur_w_kernel<2>, grid, dim3(256), lds, stream, w.table, K, (int)ncam, X, npt, ldx, v_pt, d.nch, w.part);
    hipLaunchKernelGGL(schur_cam_fold_kernel, dim3((unsigned)ncam), dim3(256), 0, stream, w.part, 4 * d.tiles, (int)ncam, w_cam);
}

}  // namespace

extern "C" size_t sfm_ba_schur_solve_ws_bytes(int64_t ncam, int64_t npt) {
    if (ncam < 1 || npt < 1) return 0;
    return solve_carve(nullptr, ncam, npt, schur_plan(ncam, npt, false)).bytes + 512;
}

// JtJ_cam [ncam x 36], Jtr_cam [ncam x 6], JtJ_pt [npt x 9], Jtr_pt [npt x 3]: the blocks sfm_ba_dense_sweep returned at
// (cams, X).  dc_dev [ncam x 6], dp_dev [npt x 3]: the step (update = parameters - step).  iters_host: CG iterations run;
// status_host: bit 0 a singular camera block (identity used as its preconditioner), bit 1 a singular point block.
// Synchronises `stream` (it returns host scalars and reads the residual norm every fifth iteration).
extern "C" int sfm_ba_schur_solve(const double* cams, int64_t ncam, const double* K_host, const float* X, int64_t npt, int64_t ldx,
                                  const double* JtJ_cam, const double* Jtr_cam, const double* JtJ_pt, const double* Jtr_pt, double lam,
                                  int fix_first_camera, double cg_tol, int cg_iters, double* dc_dev, double* dp_dev, int32_t* iters_host,
                                  int32_t* status_host, void* ws, size_t ws_bytes, void* stream_) {
    SFM_CHECK_ARG(ncam >= 1 && npt >= 1 && ldx >= 3 && ncam < (1 << 24) && cg_iters >= 0, "sfm_ba_schur_solve: bad sizes");
    SFM_CHECK_ARG(cams && K_host && X && JtJ_cam && Jtr_cam && JtJ_pt && Jtr_pt && dc_dev && dp_dev, "sfm_ba_schur_solve: null pointer");
    const size_t need = sfm_ba_schur_solve_ws_bytes(ncam, npt);
    if (!ws || ws_bytes < need) {
        sfm::set_error("sfm_ba_schur_solve: workspace too small (%zu < %zu)", ws_bytes, need);
        return SFM_ERR_WORKSPACE;
    }
    hipStream_t stream = sfm::as_stream(stream_);
    const SchurPlan d = schur_plan(ncam, npt, false);
    const SolveWs w = solve_carve(reinterpret_cast<void*>(sfm::align_up((size_t)(uintptr_t)ws, 256)), ncam, npt, d);
    const Intrin K{K_host[0], K_host[4], K_host[2], K_host[5]};
    const int n = (int)ncam * 6;
    SFM_CHECK_HIP(hipMemsetAsync(w.status, 0, sizeof(int) * 4, stream));
    hipLaunchKernelGGL(schur_cam_prepare_kernel, dim3((unsigned)((ncam + 63) / 64)), dim3(64), 0, stream, cams, ncam, w.prod.table);
    hipLaunchKernelGGL(schur_damp_cam_kernel, dim3((unsigned)((ncam + 63) / 64)), dim3(64), 0, stream, JtJ_cam, (int)ncam, lam, w.Bd, w.Minv, w.status);
    const unsigned pblocks = (unsigned)((npt + 255) / 256);
    hipLaunchKernelGGL(schur_damp_pt_kernel, dim3(pblocks), dim3(256), 0, stream, JtJ_pt, npt, lam, w.Cinv, w.status);
    SFM_CHECK_LAUNCH();
    // rhs = g_c - W Cd^-1 g_p
    hipLaunchKernelGGL(schur_pt_apply_kernel, dim3(pblocks), dim3(256), 0, stream, w.Cinv, Jtr_pt, (const double*)nullptr, 1.0, npt, w.v);
    const SchurPlan dw = schur_plan(ncam, npt, true);                // (the camera-side product has its own occupancy, hence its own chunking)
    launch_w(stream, dw, w.prod, K, ncam, X, npt, ldx, w.v, w.wv);
    hipLaunchKernelGGL(schur_cg_init_kernel, dim3(1), dim3(kCgThreads), 0, stream, Jtr_cam, w.wv, w.Minv, n, fix_first_camera ? 1 : 0,
                       cg_tol * cg_tol, dc_dev, w.r, w.p, w.scal);
    SFM_CHECK_LAUNCH();
    int it = 0;
    while (it < cg_iters) {
        if (it % 5 == 0) {
            double h[kScalWords];
            SFM_CHECK_HIP(hipMemcpyAsync(h, w.scal, sizeof(h), hipMemcpyDeviceToHost, stream));
            SFM_CHECK_HIP(sfm::stream_sync(stream));
            if (h[kScalRr] <= h[kScalStop2]) break;
        }
        sfm::prof_begin(sfm::kProfBaSchur, stream);
        launch_wt(stream, d, w.prod, K, ncam, X, npt, ldx, w.p, w.u);                                   // u = W^T p
        hipLaunchKernelGGL(schur_pt_apply_kernel, dim3(pblocks), dim3(256), 0, stream, w.Cinv, w.u, (const double*)nullptr, 1.0, npt, w.v);
        launch_w(stream, dw, w.prod, K, ncam, X, npt, ldx, w.v, w.wv);                                   // w = W Cd^-1 u
        sfm::prof_end(sfm::kProfBaSchur, stream, 2);
        hipLaunchKernelGGL(schur_cg_step_kernel, dim3(1), dim3(kCgThreads), 0, stream, w.Bd, w.Minv, w.wv, n, fix_first_camera ? 1 : 0, dc_dev,
                           w.r, w.p, w.tmp, w.scal);
        SFM_CHECK_LAUNCH();
        ++it;
    }
    // dp = Cd^-1 (g_p - W^T dc)
    launch_wt(stream, d, w.prod, K, ncam, X, npt, ldx, dc_dev, w.u);
    hipLaunchKernelGGL(schur_pt_apply_kernel, dim3(pblocks), dim3(256), 0, stream, w.Cinv, w.u, Jtr_pt, -1.0, npt, dp_dev);
    SFM_CHECK_LAUNCH();
    int st[4] = {0, 0, 0, 0};
    SFM_CHECK_HIP(hipMemcpyAsync(st, w.status, sizeof(st), hipMemcpyDeviceToHost, stream));
    SFM_CHECK_HIP(sfm::stream_sync(stream));
    if (iters_host) *iters_host = it;
    if (status_host) *status_host = st[0];
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
