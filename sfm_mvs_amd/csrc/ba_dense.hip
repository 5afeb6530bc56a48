// Dense-visibility Gauss-Newton sweep (BASELINE config 4: 500 cameras x 200 000 points = 1e8
// observations): the residual function of sfm.py:104-136 (OptimReprojectionError) with its analytic
// Jacobians, accumulated into the per-camera 6x6 / 6 and per-point 3x3 / 3 normal-equation blocks a
// sparse BA solver consumes — instead of the reference's (5N+22) finite-difference evaluations per
// Jacobian (scipy least_squares, sfm.py:146).
//
// Layout / roofline: obs is [ncam][npt][2] float32, streamed exactly once with 8-byte coalesced
// loads (8 B per observation is the algorithmic HBM traffic); points are float32 [npt][ldx].
// Arithmetic is fp64 (~210 DP instructions per observation) so the kernel sits on the fp64 VALU
// roof, not HBM — both are reported by bench.py --workload ba.
//
// Decomposition: block = 256 lanes, lane owns PP points (registers hold their 3x3+3 accumulators
// for the whole camera loop); blockIdx.y selects a camera chunk.  Per camera the 27+1 camera-side
// sums of the block's 256*PP observations are reduced in a fixed order through LDS
// ([value][segment][lane] with odd strides: conflict-free writes, <=2-way reads) and written as one
// partial row per (tile, camera).  Two small fixed-order kernels fold the partials → deterministic,
// no floating-point atomics anywhere.
#include "common.h"
#include <cfloat>

// This file's sums are checked against the oracle to 1e-10, not bit for bit (unlike the KNN refine, the triangulation and
// the SIFT kernels, for which the library is built with -ffp-contract=off): let the compiler fuse a * b + c here — the
// sweeps are bound by the fp64 vector ALU's instruction count, and the unfused forms cost a third more instructions.
#pragma clang fp contract(fast)

namespace {

constexpr int kCamStride = 24;     // R (9), t (3), the three rotation-derivative axes w_j (9), padding: 42 scalar registers of table per camera
constexpr int kNAcc = 28;          // 21 upper JtJ + 6 Jtr + 1 sumsq
constexpr int kSegStride = 33;     // doubles; 32 lanes + 1 pad
constexpr int kValStride = 2 * kSegStride + 1;   // a wave's slab: [value][half][32 lanes + pad] + pad

struct Intrin {
    double fx, fy, cx, cy;
};

__device__ void rodrigues_dev2(const double* __restrict__ rv, double* __restrict__ R, double* __restrict__ J) {
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

// Per-camera table.  The derivative of the rotated point with respect to the Rodrigues vector is a cross product:
//     d(R X)/dr_j = (dR/dr_j) X = [w_j]x (R X),     [w_j]x = (dR/dr_j) R^T  (skew-symmetric: R R^T = I differentiated)
// so instead of OpenCV's 27-entry dR/dr (cv::Rodrigues' Jacobian, which the oracle multiplies out) the sweep needs the three
// axes w_j — 9 numbers — and the rotated point it has already computed: 6 instead of 9 instructions per axis and observation,
// and a table of 21 doubles per camera (42 scalar registers) instead of 39 (78: more than the kernel could keep across the four
// points of a lane, round 3).  Same value up to rounding (the sums are checked to 1e-10 against the oracle's dR/dr form).
__global__ void dense_cam_prepare_kernel(const double* __restrict__ cams, int64_t ncam, double* __restrict__ table) {
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c >= ncam) return;
    double R[9], J[27];
    rodrigues_dev2(cams + 6 * c, R, J);
    double* e = table + c * kCamStride;
    for (int k = 0; k < 9; ++k) e[k] = R[k];
    for (int k = 0; k < 3; ++k) e[9 + k] = cams[6 * c + 3 + k];
    for (int j = 0; j < 3; ++j) {
        double S[9];                                       // (dR/dr_j) R^T
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) S[3 * a + b] = J[9 * j + 3 * a] * R[3 * b] + J[9 * j + 3 * a + 1] * R[3 * b + 1] + J[9 * j + 3 * a + 2] * R[3 * b + 2];
        e[12 + 3 * j + 0] = 0.5 * (S[7] - S[5]);
        e[12 + 3 * j + 1] = 0.5 * (S[2] - S[6]);
        e[12 + 3 * j + 2] = 0.5 * (S[3] - S[1]);
    }
    e[21] = e[22] = e[23] = 0;
}

// FULL: every point of the tile exists (all tiles but the last): no per-point liveness test anywhere, so a camera's four
// points per lane are ONE basic block — the camera's table is fetched once (scalar loads) instead of once per point section
// (round 3: four sections, each behind its own s_waitcnt on the scalar loads), and the four independent chains interleave.
template <int PP, bool FULL>
__device__ __forceinline__ void ba_dense_body(const double* __restrict__ table, const Intrin& K, int ncam,
                                              const float* __restrict__ X, int64_t npt, int64_t ldx,
                                              const float* __restrict__ obs, int nch,
                                              double* __restrict__ cam_part, double* __restrict__ pt_part, double* __restrict__ red) {
    const int tid = threadIdx.x;
    const int tile = blockIdx.x, ch = blockIdx.y;
    const int c_begin = (int)((int64_t)ncam * ch / nch), c_end = (int)((int64_t)ncam * (ch + 1) / nch);
    const int64_t p0 = (int64_t)tile * (256 * PP) + tid;

    double Xw[PP], Yw[PP], Zw[PP];
    bool live[PP];
    double pacc[PP][9];   // 6 upper 3x3 + 3
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
        const int64_t p = p0 + 256 * pp;
        live[pp] = FULL || p < npt;
        const int64_t ps = live[pp] ? p : 0;
        Xw[pp] = X[ps * ldx];
        Yw[pp] = X[ps * ldx + 1];
        Zw[pp] = X[ps * ldx + 2];
#pragma unroll
        for (int k = 0; k < 9; ++k) pacc[pp][k] = 0;
    }

    const int lane = tid & 63, wave = tid >> 6;
    const int whalf = lane >> 5, wlane = lane & 31;   // LDS slot of this lane's contribution inside the wave's slab
    double* const wred = red + wave * (kNAcc * kValStride);

    // The observations of camera c + 1 are requested while camera c is worked on.  (Tried in round 4 and not kept: a second
    // buffer with the loop unrolled by two — two cameras of distance, but 16 more registers than the 256 that two waves per SIMD
    // allow: 15 spills, 1.02 ms; touching camera c + 2's lines with an extra 4-byte load per lane — 0.86 ms against 0.80: the
    // sweep does not wait for its observations, it waits for the per-camera fold and the scalar table loads.)
    float2 ob_next[PP];
    const float* const ob0 = obs + p0 * 2;
    auto ob_ptr = [&](int pp, int64_t cam) -> const float* {
        if constexpr (FULL) return ob0 + cam * npt * 2 + 512 * pp;   // (one base address, the points' 2 KiB offsets are immediates)
        else return obs + (cam * npt + (live[pp] ? p0 + 256 * pp : 0)) * 2;
    };
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) ob_next[pp] = *reinterpret_cast<const float2*>(ob_ptr(pp, min(c_begin, ncam - 1)));
    for (int c = c_begin; c < c_end; ++c) {
        const double* __restrict__ e = table + (int64_t)c * kCamStride;
        double cacc[kNAcc];
#pragma unroll
        for (int k = 0; k < kNAcc; ++k) cacc[k] = 0;
        float2 ob[PP];
        const int64_t c1 = min(c + 1, c_end - 1);                    // (unconditional: a conditional reload is a select at the back edge)
#pragma unroll
        for (int pp = 0; pp < PP; ++pp) {
            ob[pp] = ob_next[pp];
            ob_next[pp] = *reinterpret_cast<const float2*>(ob_ptr(pp, c1));
        }
#pragma unroll
        for (int pp = 0; pp < PP; ++pp) {
            if (!FULL && !live[pp]) continue;
            const double x0 = e[0] * Xw[pp] + e[1] * Yw[pp] + e[2] * Zw[pp];      // R X: also the lever arm of the rotation derivatives
            const double y0 = e[3] * Xw[pp] + e[4] * Yw[pp] + e[5] * Zw[pp];
            const double z0 = e[6] * Xw[pp] + e[7] * Yw[pp] + e[8] * Zw[pp];
            double x = x0 + e[9], y = y0 + e[10], z = z0 + e[11];
            // 1 / z: the hardware estimate (~26 bits) + two Newton steps — a third of the IEEE division's instructions,
            // full double accuracy up to the last bit or two (the sums are checked to 1e-10)
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
            const double u = x * K.fx + K.cx, v = y * K.fy + K.cy;
            const float dxf = (float)u - ob[pp].x, dyf = (float)v - ob[pp].y;
            cacc[27] += (double)dxf * (double)dxf + (double)dyf * (double)dyf;
            const double ru = u - (double)ob[pp].x, rv = v - (double)ob[pp].y;
            const double fxz = K.fx * z, fyz = K.fy * z;
            // camera-side Jacobian rows: Ju = (Ju0, Ju1, Ju2, fx z, 0, -fx x z), Jv = (Jv0, Jv1, Jv2, 0, fy z, -fy y z)
            double Ju[3], Jv[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const double* w = e + 12 + 3 * j;                                    // d(R X)/dr_j = w_j x (R X)
                const double dx0 = w[1] * z0 - w[2] * y0;
                const double dy0 = w[2] * x0 - w[0] * z0;
                const double dz0 = w[0] * y0 - w[1] * x0;
                Ju[j] = fxz * (dx0 - x * dz0);
                Jv[j] = fyz * (dy0 - y * dz0);
            }
            const double Ju5 = -x * fxz, Jv5 = -y * fyz;
            // J^T J (upper triangle, row-major q = 0..20) and J^T r as one fused multiply-add per NONZERO product: the
            // structural zeros of the translation columns (Ju[4] = Jv[3] = 0) are never multiplied
            {
                // rows 0..2
                int q = 0;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
#pragma unroll
                    for (int b2 = a; b2 < 3; ++b2) {
                        cacc[q] = fma(Ju[a], Ju[b2], cacc[q]);
                        cacc[q] = fma(Jv[a], Jv[b2], cacc[q]);
                        ++q;
                    }
                    cacc[q] = fma(Ju[a], fxz, cacc[q]); ++q;                    // (a, 3)
                    cacc[q] = fma(Jv[a], fyz, cacc[q]); ++q;                    // (a, 4)
                    cacc[q] = fma(Ju[a], Ju5, cacc[q]);
                    cacc[q] = fma(Jv[a], Jv5, cacc[q]); ++q;                    // (a, 5)
                }
                cacc[15] = fma(fxz, fxz, cacc[15]);                             // (3, 3); (3, 4) = cacc[16] stays 0
                cacc[17] = fma(fxz, Ju5, cacc[17]);                             // (3, 5)
                cacc[18] = fma(fyz, fyz, cacc[18]);                             // (4, 4)
                cacc[19] = fma(fyz, Jv5, cacc[19]);                             // (4, 5)
                cacc[20] = fma(Ju5, Ju5, cacc[20]);
                cacc[20] = fma(Jv5, Jv5, cacc[20]);                             // (5, 5)
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    cacc[21 + a] = fma(Ju[a], ru, cacc[21 + a]);
                    cacc[21 + a] = fma(Jv[a], rv, cacc[21 + a]);
                }
                cacc[24] = fma(fxz, ru, cacc[24]);
                cacc[25] = fma(fyz, rv, cacc[25]);
                cacc[26] = fma(Ju5, ru, cacc[26]);
                cacc[26] = fma(Jv5, rv, cacc[26]);
            }
            double Pu[3], Pv[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                Pu[k] = fxz * (e[k] - x * e[6 + k]);
                Pv[k] = fyz * (e[3 + k] - y * e[6 + k]);
            }
            {
                int q = 0;
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b2 = a; b2 < 3; ++b2) {
                        pacc[pp][q] = fma(Pu[a], Pu[b2], pacc[pp][q]);
                        pacc[pp][q] = fma(Pv[a], Pv[b2], pacc[pp][q]);
                        ++q;
                    }
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    pacc[pp][6 + a] = fma(Pu[a], ru, pacc[pp][6 + a]);
                    pacc[pp][6 + a] = fma(Pv[a], rv, pacc[pp][6 + a]);
                }
            }
        }
        // Fixed-order reduction of the 28 camera-side sums, PER WAVE and without a workgroup barrier: the wave's 64 partials
        // of a value sit in its own LDS slab ([value][half][32 lanes], odd strides); lane 2k + h adds the 32 partials of
        // half h of value k in ascending order, one exchange joins the halves, and the wave writes ITS partial row
        // (4 rows per tile and camera: the fold kernel adds them).  A wave's LDS operations execute in issue order, so
        // neither the readers of this camera nor the writers of the next need more than a wave barrier — and the four
        // waves drift apart, which is the point: with two __syncthreads per camera around a workgroup-wide fold this
        // step cost 38 % of the kernel (ablation), all of it LDS time nobody overlapped.
#pragma unroll
        for (int k = 0; k < kNAcc; ++k) wred[k * kValStride + whalf * kSegStride + wlane] = cacc[k];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < 2 * kNAcc) {
            const double* src = wred + (lane >> 1) * kValStride + (lane & 1) * kSegStride;
            // (four interleaved chains of eight, then a fixed tree: one chain of 32 dependent adds is 32 add latencies)
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
#pragma unroll
            for (int i = 0; i < 32; i += 4) {
                s0 += src[i];
                s1 += src[i + 1];
                s2 += src[i + 2];
                s3 += src[i + 3];
            }
            double s = (s0 + s1) + (s2 + s3);
            s += __shfl_xor(s, 1, 64);
            if ((lane & 1) == 0) cam_part[(((int64_t)tile * 4 + wave) * ncam + c) * kNAcc + (lane >> 1)] = s;
        }
        __builtin_amdgcn_wave_barrier();
    }

#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
        const int64_t p = p0 + 256 * pp;
        if (FULL || live[pp]) {
            double* dst = pt_part + ((int64_t)ch * npt + p) * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) dst[k] = pacc[pp][k];
        }
    }
}

template <int PP>
__global__ __launch_bounds__(256) void ba_dense_kernel(const double* __restrict__ table, Intrin K, int ncam,
                                                       const float* __restrict__ X, int64_t npt, int64_t ldx,
                                                       const float* __restrict__ obs, int nch,
                                                       double* __restrict__ cam_part /*[4 * tiles][ncam][28]: a row per WAVE*/,
                                                       double* __restrict__ pt_part /*[nch][npt][9]*/) {
    extern __shared__ __attribute__((aligned(16))) double red[];   // 4 waves x kNAcc * kValStride doubles
    if (((int64_t)blockIdx.x + 1) * (256 * PP) <= npt)              // (uniform) a full tile
        ba_dense_body<PP, true>(table, K, ncam, X, npt, ldx, obs, nch, cam_part, pt_part, red);
    else
        ba_dense_body<PP, false>(table, K, ncam, X, npt, ldx, obs, nch, cam_part, pt_part, red);
}

// One workgroup per camera folds its `tiles` partial rows (kNAcc doubles each) in a FIXED two-level shape: 1024 threads =
// 32 groups x 32 entry lanes, group g adds tiles g, g + 32, ... in ascending order, then the 32 group sums of an entry are
// added pairwise (stride 16 .. 1).  (Round 2: one thread per entry walking every tile — 76 us per sweep at 196 tiles.)
__global__ __launch_bounds__(1024) void dense_cam_reduce_kernel(const double* __restrict__ cam_part, int tiles, int ncam,
                                                                double* __restrict__ JtJ_cam, double* __restrict__ Jtr_cam,
                                                                double* __restrict__ cam_sumsq) {
    __shared__ double lds[32][32];
    const int c = blockIdx.x, k = threadIdx.x & 31, g = threadIdx.x >> 5;
    double s = 0;
    if (k < kNAcc)
        for (int t = g; t < tiles; t += 32) s += cam_part[((int64_t)t * ncam + c) * kNAcc + k];
    lds[g][k] = s;
    __syncthreads();
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) {
        if (g < m) lds[g][k] = lds[g][k] + lds[g + m][k];
        __syncthreads();
    }
    if (threadIdx.x >= kNAcc) return;
    s = lds[0][k];
    if (k == 27) {
        cam_sumsq[c] = s;
    } else if (k >= 21) {
        if (Jtr_cam) Jtr_cam[c * 6 + (k - 21)] = s;
    } else if (JtJ_cam) {
        int a = 0, rem = k;
        while (rem >= 6 - a) { rem -= 6 - a; ++a; }
        const int b = a + rem;
        JtJ_cam[c * 36 + a * 6 + b] = s;
        JtJ_cam[c * 36 + b * 6 + a] = s;
    }
}

// sum of the per-camera sums: 1024 lanes take cameras tid, tid + 1024, ... in order, then a fixed tree
__global__ __launch_bounds__(1024) void dense_sumsq_kernel(const double* __restrict__ cam_sumsq, int ncam, double* __restrict__ sumsq) {
    __shared__ double lds[1024];
    double s = 0;
    for (int c = threadIdx.x; c < ncam; c += 1024) s += cam_sumsq[c];
    lds[threadIdx.x] = s;
    __syncthreads();
    for (int m = 512; m >= 1; m >>= 1) {
        if ((int)threadIdx.x < m) lds[threadIdx.x] += lds[threadIdx.x + m];
        __syncthreads();
    }
    if (threadIdx.x == 0) *sumsq = lds[0];
}

__global__ __launch_bounds__(256) void dense_pt_reduce_kernel(const double* __restrict__ pt_part, int nch, int64_t npt,
                                                              double* __restrict__ JtJ_pt, double* __restrict__ Jtr_pt) {
    const int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (p >= npt) return;
    double s[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) s[k] = 0;
    for (int ch = 0; ch < nch; ++ch) {
        const double* src = pt_part + ((int64_t)ch * npt + p) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) s[k] += src[k];
    }
    if (JtJ_pt) {
        double* d = JtJ_pt + p * 9;
        d[0] = s[0]; d[1] = s[1]; d[2] = s[2];
        d[3] = s[1]; d[4] = s[3]; d[5] = s[4];
        d[6] = s[2]; d[7] = s[4]; d[8] = s[5];
    }
    if (Jtr_pt) {
        Jtr_pt[p * 3 + 0] = s[6];
        Jtr_pt[p * 3 + 1] = s[7];
        Jtr_pt[p * 3 + 2] = s[8];
    }
}

struct DensePlan {
    int pp, tiles, nch;
};

DensePlan dense_plan(int64_t ncam, int64_t npt) {
    DensePlan d;
    d.pp = npt >= 64 * 1024 ? 4 : 2;
    d.tiles = (int)((npt + 256 * d.pp - 1) / (256 * d.pp));
    d.nch = sfm::pick_camera_chunks(d.tiles, ncam, 512);      // (two workgroups per CU: 202 VGPRs, 59 KB of LDS)
    return d;
}

struct DenseWs {
    double *table, *cam_part, *pt_part, *cam_sumsq;
    size_t bytes;
};

DenseWs dense_carve(void* base, int64_t ncam, int64_t npt, const DensePlan& d) {
    sfm::Carver c(base);
    DenseWs w;
    w.table = c.take<double>((size_t)ncam * kCamStride);
    w.cam_sumsq = c.take<double>((size_t)ncam);
    w.cam_part = c.take<double>((size_t)4 * d.tiles * ncam * kNAcc);
    w.pt_part = c.take<double>((size_t)d.nch * npt * 9);
    w.bytes = c.used();
    return w;
}

}  // namespace

extern "C" size_t sfm_ba_dense_sweep_ws_bytes(int64_t ncam, int64_t npt) {
    if (ncam < 1 || npt < 1) return 0;
    return dense_carve(nullptr, ncam, npt, dense_plan(ncam, npt)).bytes + 256;
}

extern "C" int sfm_ba_dense_sweep(const double* cams, int64_t ncam, const double* K_host, const float* X, int64_t npt,
                                  int64_t ldx, const float* obs, double* sumsq, double* JtJ_cam, double* Jtr_cam,
                                  double* JtJ_pt, double* Jtr_pt, void* ws, size_t ws_bytes, void* stream_) {
    SFM_CHECK_ARG(ncam >= 1 && npt >= 1 && ldx >= 3 && ncam < (1 << 24), "sfm_ba_dense_sweep: bad sizes");
    SFM_CHECK_ARG(cams && K_host && X && obs, "sfm_ba_dense_sweep: null pointer");
    SFM_CHECK_ARG(((uintptr_t)obs & 7) == 0, "sfm_ba_dense_sweep: obs must be 8-byte aligned");
    const DensePlan d = dense_plan(ncam, npt);
    const size_t need = sfm_ba_dense_sweep_ws_bytes(ncam, npt);
    if (!ws || ws_bytes < need) {
        sfm::set_error("sfm_ba_dense_sweep: workspace too small (%zu < %zu)", ws_bytes, need);
        return SFM_ERR_WORKSPACE;
    }
    hipStream_t stream = sfm::as_stream(stream_);
    const DenseWs w = dense_carve(reinterpret_cast<void*>(sfm::align_up((size_t)(uintptr_t)ws, 256)), ncam, npt, d);
    const Intrin K{K_host[0], K_host[4], K_host[2], K_host[5]};
    hipLaunchKernelGGL(dense_cam_prepare_kernel, dim3((unsigned)((ncam + 63) / 64)), dim3(64), 0, stream, cams, ncam, w.table);
    SFM_CHECK_LAUNCH();
    const size_t lds = (size_t)4 * kNAcc * kValStride * sizeof(double);
    const dim3 grid((unsigned)d.tiles, (unsigned)d.nch);
    sfm::prof_begin(sfm::kProfBaDense, stream);
    if (d.pp == 4)
        hipLaunchKernelGGL(ba_dense_kernel<4>, grid, dim3(256), lds, stream, w.table, K, (int)ncam, X, npt, ldx, obs, d.nch,
                           w.cam_part, w.pt_part);
    else
        hipLaunchKernelGGL(ba_dense_kernel<2>, grid, dim3(256), lds, stream, w.table, K, (int)ncam, X, npt, ldx, obs, d.nch,
                           w.cam_part, w.pt_part);
    sfm::prof_end(sfm::kProfBaDense, stream);
    SFM_CHECK_LAUNCH();
    hipLaunchKernelGGL(dense_cam_reduce_kernel, dim3((unsigned)ncam), dim3(1024), 0, stream, w.cam_part, 4 * d.tiles, (int)ncam,
                       JtJ_cam, Jtr_cam, w.cam_sumsq);
    SFM_CHECK_LAUNCH();
    if (sumsq) {
        hipLaunchKernelGGL(dense_sumsq_kernel, dim3(1), dim3(1024), 0, stream, w.cam_sumsq, (int)ncam, sumsq);
        SFM_CHECK_LAUNCH();
    }
    if (JtJ_pt || Jtr_pt) {
        hipLaunchKernelGGL(dense_pt_reduce_kernel, dim3((unsigned)((npt + 255) / 256)), dim3(256), 0, stream, w.pt_part, d.nch,
                           npt, JtJ_pt, Jtr_pt);
        SFM_CHECK_LAUNCH();
    }
    return SFM_OK;
}
