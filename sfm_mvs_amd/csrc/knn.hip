// Brute-force 128-D L2 2-NN (cv2.BFMatcher().knnMatch(des0, des1, k=2), sfm.py:259-260)
// for gfx950 — "certified filter + exact refine":
//
//   1. knn_norms_kernel   ||t||^2 per train row (+ global max), fp32.
//   2. knn_filter_kernel  s(q,t) = ||t||^2 - 2 q.t on v_mfma_f32_32x32x2_f32 with the
//                         OPERANDS SWAPPED (A = train tile from LDS, B = query fragment
//                         resident in VGPRs) so that a lane owns ONE query column and the
//                         running top-3 per lane needs no cross-lane traffic.  Every
//                         (train split, half-wave) pair is an independent "stream" that
//                         emits its 3 best (s, idx).
//   3. knn_refine_kernel  one wave per query: re-evaluates the few candidates that can
//                         still be in the top-2 with the reference's direct-form float32
//                         arithmetic (sub, mul, add — no FMA — in OpenCV's 2x4-lane
//                         accumulation order, then sqrtf), orders them by (dist, idx) and
//                         CERTIFIES the answer against the lower bound of everything the
//                         filter discarded.  Uncertifiable queries are queued …
//   4. knn_fallback_kernel … and resolved by an exact direct-form scan of all trains.
//
// The GEMM-form value is therefore never returned: indices and distances are bit-identical
// to the direct-form oracle (oracle/sfm_oracle.c: orc_knn2_l2_f32) for any finite input.
#include "common.h"
#include <cfloat>
#include <climits>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kDim = 128;
constexpr int kTileT = 32;             // train rows per LDS tile (= MFMA M)
constexpr int kLdsRow = kDim + 4;      // +4 floats: ds_read_b128 of 32 rows is conflict-free
constexpr int kWaves = 4;
constexpr int kThreads = kWaves * 64;
constexpr int kMaxSplit = 32;
constexpr int kTargetBlocks = 512;     // 256 CUs x 2 resident workgroups
constexpr float kInf = __builtin_huge_valf();

struct Plan {
    int qg;           // 32-query groups per wave (1 or 2)
    int rows_per_block;
    int n_rb;         // query row blocks
    int tiles;        // train tiles of 32
    int S;            // train splits
    int NS;           // streams per query = 2*S
};

Plan make_plan(int64_t nq, int64_t nt) {
    Plan p;
    p.qg = (nq >= 256 * 24) ? 2 : 1;
    p.rows_per_block = kWaves * 32 * p.qg;
    p.n_rb = (int)((nq + p.rows_per_block - 1) / p.rows_per_block);
    p.tiles = (int)((nt + kTileT - 1) / kTileT);
    int s = p.n_rb > 0 ? kTargetBlocks / p.n_rb : 1;
    if (s < 1) s = 1;
    if (s > kMaxSplit) s = kMaxSplit;
    if (s > p.tiles) s = p.tiles > 0 ? p.tiles : 1;
    p.S = s;
    p.NS = 2 * s;
    return p;
}

// ---------------------------------------------------------------- norms
__global__ __launch_bounds__(256) void knn_norms_kernel(const float* __restrict__ T, int64_t ldt, int nt,
                                                        float* __restrict__ tn, unsigned* __restrict__ tmax_bits) {
    // 32 lanes per row, one float4 each.
    const int row = blockIdx.x * 8 + (threadIdx.x >> 5);
    const int l = threadIdx.x & 31;
    float s = 0.f;
    if (row < nt) {
        const float4 v = *reinterpret_cast<const float4*>(T + (int64_t)row * ldt + 4 * l);
        s = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
#pragma unroll
    for (int m = 16; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if (row < nt && l == 0) {
        tn[row] = s;
        atomicMax(tmax_bits, __float_as_uint(s));   // s >= 0: uint order == float order
    }
}

// ---------------------------------------------------------------- filter
__device__ __forceinline__ void top3_insert(float v, int id, float (&s)[3], int (&ix)[3]) {
    const bool lt0 = v < s[0], lt1 = v < s[1], lt2 = v < s[2];
    s[2] = lt1 ? s[1] : (lt2 ? v : s[2]);
    ix[2] = lt1 ? ix[1] : (lt2 ? id : ix[2]);
    s[1] = lt0 ? s[0] : (lt1 ? v : s[1]);
    ix[1] = lt0 ? ix[0] : (lt1 ? id : ix[1]);
    s[0] = lt0 ? v : s[0];
    ix[0] = lt0 ? id : ix[0];
}

struct StageRegs {
    float4 v[4];
    float tn;
};

__device__ __forceinline__ void stage_load(StageRegs& r, const float* __restrict__ T, int64_t ldt, int nt,
                                           const float* __restrict__ tn, int tile) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int f = tid + kThreads * n;
        const int row = tile * kTileT + (f >> 5);
        const int c4 = f & 31;
        r.v[n] = (row < nt) ? *reinterpret_cast<const float4*>(T + (int64_t)row * ldt + 4 * c4)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    r.tn = kInf;
    if (tid < kTileT) {
        const int row = tile * kTileT + tid;
        if (row < nt) r.tn = tn[row];
    }
}

__device__ __forceinline__ void stage_store(const StageRegs& r, float* __restrict__ buf) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const int f = tid + kThreads * n;
        *reinterpret_cast<float4*>(buf + (f >> 5) * kLdsRow + 4 * (f & 31)) = r.v[n];
    }
    if (tid < kTileT) buf[kTileT * kLdsRow + tid] = r.tn;
}

constexpr int kBufFloats = kTileT * kLdsRow + kTileT;   // tile + its ||t||^2 row

template <int QG>
__global__ __launch_bounds__(kThreads, 2) void knn_filter_kernel(
    const float* __restrict__ Q, int64_t ldq, int nq, const float* __restrict__ T, int64_t ldt, int nt,
    const float* __restrict__ tn, int S, int tiles, float* __restrict__ cand_s, int* __restrict__ cand_i) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int j = lane & 31;   // MFMA column  = query within group / train row for A loads
    const int h = lane >> 5;   // k-half selector / C row block selector
    const int rb = blockIdx.x / S;
    const int sp = blockIdx.x - rb * S;
    const int q0 = rb * (kWaves * 32 * QG) + wave * (32 * QG);
    const int t_begin = (int)((int64_t)tiles * sp / S);
    const int t_end = (int)((int64_t)tiles * (sp + 1) / S);

    // Query fragment (MFMA B operand), pre-scaled by -2 (exact), resident for the whole block.
    // k-order: MFMA step 4c+e of half h consumes k = 8c + 4h + e — the same permutation
    // is applied to the A (train) operand, so the dot product is complete.
    float bq[QG][64];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const int row = q0 + 32 * g + j;
        const bool ok = row < nq;
        const float* src = Q + (int64_t)(ok ? row : 0) * ldq + 4 * h;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float4 v = *reinterpret_cast<const float4*>(src + 8 * c);
            if (!ok) v = make_float4(0.f, 0.f, 0.f, 0.f);
            bq[g][4 * c + 0] = -2.f * v.x;
            bq[g][4 * c + 1] = -2.f * v.y;
            bq[g][4 * c + 2] = -2.f * v.z;
            bq[g][4 * c + 3] = -2.f * v.w;
        }
    }

    float bs[QG][3];
    int bi[QG][3];
#pragma unroll
    for (int g = 0; g < QG; ++g)
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            bs[g][r] = kInf;
            bi[g][r] = -1;
        }

    StageRegs sr;
    if (t_begin < t_end) {
        stage_load(sr, T, ldt, nt, tn, t_begin);
        stage_store(sr, smem);
    }
    __syncthreads();

    for (int t = t_begin; t < t_end; ++t) {
        const float* buf = smem + ((t - t_begin) & 1) * kBufFloats;
        float* nbuf = smem + (((t - t_begin) & 1) ^ 1) * kBufFloats;
        const bool more = (t + 1 < t_end);
        if (more) stage_load(sr, T, ldt, nt, tn, t + 1);   // in flight behind the MFMAs

        // C init = ||t||^2 of the 16 train rows this lane owns: row(r) = (r&3) + 8*(r>>2) + 4h
        f32x16 acc[QG];
        {
            const float* tnp = buf + kTileT * kLdsRow + 4 * h;
            f32x16 c0;
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const float4 v = *reinterpret_cast<const float4*>(tnp + 8 * b);
                c0[4 * b + 0] = v.x;
                c0[4 * b + 1] = v.y;
                c0[4 * b + 2] = v.z;
                c0[4 * b + 3] = v.w;
            }
#pragma unroll
            for (int g = 0; g < QG; ++g) acc[g] = c0;
        }

        const float* arow = buf + j * kLdsRow + 4 * h;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float4 a = *reinterpret_cast<const float4*>(arow + 8 * c);
#pragma unroll
            for (int g = 0; g < QG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, bq[g][4 * c + 0], acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < QG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, bq[g][4 * c + 1], acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < QG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, bq[g][4 * c + 2], acc[g], 0, 0, 0);
#pragma unroll
            for (int g = 0; g < QG; ++g) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, bq[g][4 * c + 3], acc[g], 0, 0, 0);
        }

        // Lane-local running top-3 (ascending train index within the lane ⇒ strict '<' keeps
        // the earlier index on equal s; ordering among equals is settled exactly by refine).
        const int id0 = t * kTileT + 4 * h;
#pragma unroll
        for (int g = 0; g < QG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) top3_insert(acc[g][r], id0 + (r & 3) + 8 * (r >> 2), bs[g], bi[g]);

        if (more) stage_store(sr, nbuf);
        __syncthreads();
    }

    // One stream per (split, half-wave): 3 candidates each.
    const int NS = 2 * S;
    const int stream = 2 * sp + h;
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const int row = q0 + 32 * g + j;
        if (row < nq) {
            const int64_t o = ((int64_t)row * NS + stream) * 3;
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                cand_s[o + r] = bs[g][r];
                cand_i[o + r] = bi[g][r];
            }
        }
    }
}

// ---------------------------------------------------------------- exact direct-form distance
// Reference arithmetic (OpenCV normL2Sqr_, SSE2 path): two 4-lane accumulators over blocks of 8,
// mul and add separately rounded; lanes summed as (d0+d1) then ((s0+s1)+s2)+s3.
// Compiled with -ffp-contract=off so none of this fuses.
__device__ __forceinline__ float exact_l2sq_128(const float* __restrict__ qrow /*LDS*/, const float* __restrict__ trow) {
    float acc[8];
#pragma unroll
    for (int l = 0; l < 8; ++l) acc[l] = 0.f;
#pragma unroll 4
    for (int i = 0; i < 16; ++i) {
        const float4 t0 = *reinterpret_cast<const float4*>(trow + 8 * i);
        const float4 t1 = *reinterpret_cast<const float4*>(trow + 8 * i + 4);
        const float4 q0 = *reinterpret_cast<const float4*>(qrow + 8 * i);
        const float4 q1 = *reinterpret_cast<const float4*>(qrow + 8 * i + 4);
        float d;
        d = q0.x - t0.x; acc[0] = acc[0] + d * d;
        d = q0.y - t0.y; acc[1] = acc[1] + d * d;
        d = q0.z - t0.z; acc[2] = acc[2] + d * d;
        d = q0.w - t0.w; acc[3] = acc[3] + d * d;
        d = q1.x - t1.x; acc[4] = acc[4] + d * d;
        d = q1.y - t1.y; acc[5] = acc[5] + d * d;
        d = q1.z - t1.z; acc[6] = acc[6] + d * d;
        d = q1.w - t1.w; acc[7] = acc[7] + d * d;
    }
    const float s0 = acc[0] + acc[4], s1 = acc[1] + acc[5], s2 = acc[2] + acc[6], s3 = acc[3] + acc[7];
    return ((s0 + s1) + s2) + s3;
}

struct Best2 {
    float d[2];     // sqrtf distance
    float dsq[2];   // its square (certificate)
    int i[2];
};

__device__ __forceinline__ bool key_less(float da, int ia, float db, int ib) {
    return da < db || (da == db && ia < ib);
}

__device__ __forceinline__ void best2_insert(Best2& b, float d, float dsq, int i) {
    if (key_less(d, i, b.d[0], b.i[0])) {
        b.d[1] = b.d[0]; b.dsq[1] = b.dsq[0]; b.i[1] = b.i[0];
        b.d[0] = d; b.dsq[0] = dsq; b.i[0] = i;
    } else if (key_less(d, i, b.d[1], b.i[1])) {
        b.d[1] = d; b.dsq[1] = dsq; b.i[1] = i;
    }
}

__device__ __forceinline__ void best2_wave_reduce(Best2& b) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        Best2 o;
        o.d[0] = __shfl_xor(b.d[0], m, 64); o.dsq[0] = __shfl_xor(b.dsq[0], m, 64); o.i[0] = __shfl_xor(b.i[0], m, 64);
        o.d[1] = __shfl_xor(b.d[1], m, 64); o.dsq[1] = __shfl_xor(b.dsq[1], m, 64); o.i[1] = __shfl_xor(b.i[1], m, 64);
        best2_insert(b, o.d[0], o.dsq[0], o.i[0]);
        best2_insert(b, o.d[1], o.dsq[1], o.i[1]);
    }
}

// ---------------------------------------------------------------- refine
__global__ __launch_bounds__(256) void knn_refine_kernel(
    const float* __restrict__ Q, int64_t ldq, int nq, const float* __restrict__ T, int64_t ldt,
    const float* __restrict__ cand_s, const int* __restrict__ cand_i, int NS,
    const unsigned* __restrict__ tmax_bits, int* __restrict__ idx_out, float* __restrict__ dist_out,
    int* __restrict__ flag_count, int* __restrict__ flag_list) {
    __shared__ __attribute__((aligned(16))) float qrows[4][kDim];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int q = blockIdx.x * 4 + wave;
    const bool valid = q < nq;

    float qq = 0.f;
    if (valid && lane < 32) {
        const float4 v = *reinterpret_cast<const float4*>(Q + (int64_t)q * ldq + 4 * lane);
        *reinterpret_cast<float4*>(&qrows[wave][4 * lane]) = v;
        qq = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
    }
    __syncthreads();
    if (!valid) return;
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) qq += __shfl_xor(qq, m, 64);

    // Slack that dominates: GEMM-form rounding of the filter (<= 2*gamma_130*(|q|+|t|)^2),
    // rounding of the direct-form sums (<= 24u*d^2), of ||q||^2, ||t||^2 (gamma_128 each) and the
    // final sqrtf merge (8u*d^2), u = 2^-24.  600u*(|q|+|t|max)^2 covers their sum with >1.5x room.
    const float tmax = __uint_as_float(*tmax_bits);
    const float nsum = sqrtf(qq) + sqrtf(tmax);
    const float eps = 600.f * 5.9604645e-8f * 1.01f * nsum * nsum;

    const int NC = NS * 3;
    const float* cs = cand_s + (int64_t)q * NC;
    const int* ci = cand_i + (int64_t)q * NC;

    // Pass A: two smallest filter scores over all candidates, and tau = the smallest score any
    // discarded train can have (every stream discards only trains >= its 3rd best).
    float m1 = kInf, m2 = kInf, tau = kInf;
    for (int c = lane; c < NC; c += 64) {
        const float s = cs[c];
        if (c % 3 == 2) tau = fminf(tau, s);
        if (s < m1) { m2 = m1; m1 = s; } else if (s < m2) { m2 = s; }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        const float o1 = __shfl_xor(m1, m, 64), o2 = __shfl_xor(m2, m, 64);
        const float lo = fminf(m1, o1), hi = fmaxf(m1, o1);
        m2 = fminf(hi, fminf(m2, o2));
        m1 = lo;
        tau = fminf(tau, __shfl_xor(tau, m, 64));
    }
    // A candidate whose score exceeds the 2nd smallest score by more than 2*eps is strictly
    // farther (even after sqrtf) than two other candidates: it cannot be in the exact top-2.
    const float thr = m2 + 4.f * eps;

    Best2 b;
    b.d[0] = b.d[1] = kInf; b.dsq[0] = b.dsq[1] = kInf; b.i[0] = b.i[1] = INT_MAX;
    for (int c0 = 0; c0 < NC; c0 += 64) {
        const int c = c0 + lane;
        if (c < NC) {
            const float s = cs[c];
            const int id = ci[c];
            if (id >= 0 && s <= thr) {
                const float dsq = exact_l2sq_128(qrows[wave], T + (int64_t)id * ldt);
                best2_insert(b, sqrtf(dsq), dsq, id);
            }
        }
    }
    best2_wave_reduce(b);

    if (lane == 0) {
        const bool have2 = b.i[1] != INT_MAX;
        idx_out[2 * q + 0] = b.i[0] == INT_MAX ? -1 : b.i[0];
        idx_out[2 * q + 1] = have2 ? b.i[1] : -1;
        dist_out[2 * q + 0] = b.d[0];
        dist_out[2 * q + 1] = b.d[1];
        // Certificate: everything the filter discarded has exact d^2 >= tau + |q|^2 - eps.
        const bool certified = (tau == kInf) || (have2 && (double)b.dsq[1] + (double)eps < (double)tau + (double)qq);
        if (!certified) flag_list[atomicAdd(flag_count, 1)] = q;
    }
}

// ---------------------------------------------------------------- exact fallback
__global__ __launch_bounds__(256) void knn_fallback_kernel(
    const float* __restrict__ Q, int64_t ldq, const float* __restrict__ T, int64_t ldt, int nt,
    const int* __restrict__ flag_count, const int* __restrict__ flag_list, int* __restrict__ idx_out,
    float* __restrict__ dist_out, int* __restrict__ stats, int S, int NS) {
    __shared__ __attribute__((aligned(16))) float qrow[kDim];
    __shared__ Best2 wbest[4];
    const int nflag = *flag_count;
    if (blockIdx.x == 0 && threadIdx.x == 0 && stats) {
        stats[0] = nflag; stats[1] = S; stats[2] = NS; stats[3] = 0;
    }
    for (int f = blockIdx.x; f < nflag; f += gridDim.x) {
        const int q = flag_list[f];
        __syncthreads();
        if (threadIdx.x < 32)
            *reinterpret_cast<float4*>(&qrow[4 * threadIdx.x]) =
                *reinterpret_cast<const float4*>(Q + (int64_t)q * ldq + 4 * threadIdx.x);
        __syncthreads();
        Best2 b;
        b.d[0] = b.d[1] = kInf; b.dsq[0] = b.dsq[1] = kInf; b.i[0] = b.i[1] = INT_MAX;
        for (int t = threadIdx.x; t < nt; t += 256) {
            const float dsq = exact_l2sq_128(qrow, T + (int64_t)t * ldt);
            best2_insert(b, sqrtf(dsq), dsq, t);
        }
        best2_wave_reduce(b);
        if ((threadIdx.x & 63) == 0) wbest[threadIdx.x >> 6] = b;
        __syncthreads();
        if (threadIdx.x == 0) {
            Best2 r = wbest[0];
            for (int w = 1; w < 4; ++w) {
                best2_insert(r, wbest[w].d[0], wbest[w].dsq[0], wbest[w].i[0]);
                best2_insert(r, wbest[w].d[1], wbest[w].dsq[1], wbest[w].i[1]);
            }
            idx_out[2 * q + 0] = r.i[0] == INT_MAX ? -1 : r.i[0];
            idx_out[2 * q + 1] = r.i[1] == INT_MAX ? -1 : r.i[1];
            dist_out[2 * q + 0] = r.d[0];
            dist_out[2 * q + 1] = r.d[1];
        }
    }
}

// ---------------------------------------------------------------- ratio test + ordered compaction
// `m.distance < 0.70 * n.distance` (sfm.py:264): float32 distances promoted to double.
__global__ __launch_bounds__(1024) void ratio_compact_kernel(const int* __restrict__ idx, const float* __restrict__ dist,
                                                             int nq, double ratio, int* __restrict__ out_q,
                                                             int* __restrict__ out_t, int* __restrict__ out_count,
                                                             unsigned char* __restrict__ mask) {
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int q0 = 0; q0 < nq; q0 += 1024) {
        const int q = q0 + threadIdx.x;
        bool pass = false;
        int ti = -1;
        if (q < nq) {
            const float d1 = dist[2 * q], d2 = dist[2 * q + 1];
            ti = idx[2 * q];
            pass = (idx[2 * q + 1] >= 0) && ((double)d1 < ratio * (double)d2);
            if (mask) mask[q] = pass ? 1 : 0;
        }
        const unsigned long long bal = __ballot(pass);
        const int prefix = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0, total = 0;
        for (int w = 0; w < 16; ++w) {
            const int c = wsum[w];
            if (w < wave) woff += c;
            total += c;
        }
        const int base = base_s;
        if (pass) {
            out_q[base + woff + prefix] = q;
            out_t[base + woff + prefix] = ti;
        }
        __syncthreads();
        if (threadIdx.x == 0) base_s = base + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *out_count = base_s;
}

__global__ __launch_bounds__(256) void gather_matches_kernel(const float2* __restrict__ kp0, const float2* __restrict__ kp1,
                                                             const int* __restrict__ oq, const int* __restrict__ ot,
                                                             const int* __restrict__ count, int64_t capacity,
                                                             float2* __restrict__ p0, float2* __restrict__ p1) {
    int64_t n = *count;
    if (n > capacity) n = capacity;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        p0[i] = kp0[oq[i]];
        p1[i] = kp1[ot[i]];
    }
}

struct KnnWs {
    float* tn;
    unsigned* tmax;
    int* flag_count;
    int* flag_list;
    float* cand_s;
    int* cand_i;
    size_t bytes;
};

KnnWs carve_ws(void* ws, int64_t nq, int64_t nt, const Plan& p) {
    sfm::Carver c(ws);
    KnnWs w;
    w.tmax = c.take<unsigned>(1);
    w.flag_count = c.take<int>(1);
    w.tn = c.take<float>((size_t)p.tiles * kTileT);
    w.flag_list = c.take<int>((size_t)nq);
    w.cand_s = c.take<float>((size_t)nq * p.NS * 3);
    w.cand_i = c.take<int>((size_t)nq * p.NS * 3);
    w.bytes = c.used();
    return w;
}

}  // namespace

extern "C" size_t sfm_knn2_l2_f32_ws_bytes(int64_t nq, int64_t nt, int dim) {
    if (nq < 0 || nt < 0 || dim != kDim) return 0;
    const Plan p = make_plan(nq, nt);
    return carve_ws(nullptr, nq, nt, p).bytes + 256;
}

extern "C" int sfm_knn2_l2_f32(const float* q, int64_t nq, int64_t ldq, const float* t, int64_t nt, int64_t ldt,
                               int dim, int32_t* idx, float* dist, int32_t* stats, void* ws, size_t ws_bytes,
                               void* stream_) {
    SFM_CHECK_ARG(dim == kDim, "sfm_knn2_l2_f32: dim must be 128 (got %d)", dim);
    SFM_CHECK_ARG(nq >= 0 && nt >= 0 && nq < INT_MAX / 256 && nt < INT_MAX / 2, "sfm_knn2_l2_f32: bad sizes nq=%lld nt=%lld",
                  (long long)nq, (long long)nt);
    if (nq == 0) return SFM_OK;
    SFM_CHECK_ARG(q && idx && dist && (t || nt == 0), "sfm_knn2_l2_f32: null pointer");
    SFM_CHECK_ARG(ldq >= dim && ldt >= dim && ldq % 4 == 0 && ldt % 4 == 0, "sfm_knn2_l2_f32: ldq/ldt must be >= dim and multiples of 4");
    SFM_CHECK_ARG(((uintptr_t)q & 15) == 0 && ((uintptr_t)t & 15) == 0, "sfm_knn2_l2_f32: q/t must be 16-byte aligned");
    const Plan p = make_plan(nq, nt);
    const size_t need = sfm_knn2_l2_f32_ws_bytes(nq, nt, dim);
    if (!ws || ws_bytes < need) {
        sfm::set_error("sfm_knn2_l2_f32: workspace too small (%zu < %zu)", ws_bytes, need);
        return SFM_ERR_WORKSPACE;
    }
    // carve from a 256-aligned base inside the caller's buffer
    char* base = reinterpret_cast<char*>(sfm::align_up((size_t)(uintptr_t)ws, 256));
    const KnnWs w = carve_ws(base, nq, nt, p);
    hipStream_t stream = sfm::as_stream(stream_);

    SFM_CHECK_HIP(hipMemsetAsync(w.tmax, 0, 512, stream));   // tmax + flag_count (adjacent 256-B slots)
    if (nt > 0) {
        hipLaunchKernelGGL(knn_norms_kernel, dim3((unsigned)((nt + 7) / 8)), dim3(256), 0, stream, t, ldt, (int)nt, w.tn,
                           w.tmax);
        SFM_CHECK_LAUNCH();
    }
    const size_t lds = 2 * (size_t)kBufFloats * sizeof(float);
    const dim3 grid((unsigned)(p.n_rb * p.S));
    sfm::prof_begin(sfm::kProfKnnFilter, stream);
    if (p.qg == 2)
        hipLaunchKernelGGL(knn_filter_kernel<2>, grid, dim3(kThreads), lds, stream, q, ldq, (int)nq, t, ldt, (int)nt, w.tn,
                           p.S, p.tiles, w.cand_s, w.cand_i);
    else
        hipLaunchKernelGGL(knn_filter_kernel<1>, grid, dim3(kThreads), lds, stream, q, ldq, (int)nq, t, ldt, (int)nt, w.tn,
                           p.S, p.tiles, w.cand_s, w.cand_i);
    sfm::prof_end(sfm::kProfKnnFilter, stream);
    SFM_CHECK_LAUNCH();
    sfm::prof_begin(sfm::kProfKnnRefine, stream);
    hipLaunchKernelGGL(knn_refine_kernel, dim3((unsigned)((nq + 3) / 4)), dim3(256), 0, stream, q, ldq, (int)nq, t, ldt,
                       w.cand_s, w.cand_i, p.NS, w.tmax, idx, dist, w.flag_count, w.flag_list);
    SFM_CHECK_LAUNCH();
    hipLaunchKernelGGL(knn_fallback_kernel, dim3(256), dim3(256), 0, stream, q, ldq, t, ldt, (int)nt, w.flag_count,
                       w.flag_list, idx, dist, stats, p.S, p.NS);
    sfm::prof_end(sfm::kProfKnnRefine, stream);
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}

extern "C" int sfm_ratio_compact(const int32_t* idx, const float* dist, int64_t nq, double ratio, int32_t* out_q,
                                 int32_t* out_t, int32_t* out_count, uint8_t* mask, void* stream_) {
    SFM_CHECK_ARG(nq >= 0 && nq < INT_MAX, "sfm_ratio_compact: bad nq");
    SFM_CHECK_ARG(out_count && (nq == 0 || (idx && dist && out_q && out_t)), "sfm_ratio_compact: null pointer");
    hipLaunchKernelGGL(ratio_compact_kernel, dim3(1), dim3(1024), 0, sfm::as_stream(stream_), idx, dist, (int)nq, ratio,
                       out_q, out_t, out_count, mask);
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}

extern "C" int sfm_gather_matches(const float* kp0, const float* kp1, const int32_t* out_q, const int32_t* out_t,
                                  const int32_t* count, int64_t capacity, float* pts0, float* pts1, void* stream_) {
    SFM_CHECK_ARG(kp0 && kp1 && out_q && out_t && count && pts0 && pts1 && capacity >= 0, "sfm_gather_matches: bad argument");
    if (capacity == 0) return SFM_OK;
    unsigned blocks = (unsigned)((capacity + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gather_matches_kernel, dim3(blocks), dim3(256), 0, sfm::as_stream(stream_),
                       reinterpret_cast<const float2*>(kp0), reinterpret_cast<const float2*>(kp1), out_q, out_t, count,
                       capacity, reinterpret_cast<float2*>(pts0), reinterpret_cast<float2*>(pts1));
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}
