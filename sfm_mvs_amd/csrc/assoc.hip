// Data association between consecutive pairs — common_points (sfm.py:215-239).
//
// For every row i of pts1 the reference takes `np.where(pts2 == pts1[i, :])[0][0]`: the FIRST row of pts2 whose x OR
// whose y is bit-equal (element-wise broadcast), appends (i, row) to (indx1, indx2) — duplicates in
// indx2 allowed — and then drops the matched rows of pts2 / pts3 by mask-and-compress.  n1 * n2 ~ 1e6-1e7 exact
// float compares per frame: one lane per row of pts1 scanning pts2 through LDS tiles (coalesced, early exit per
// wave), then an ordered ballot compaction.  Integer outputs, bit-exact by construction.
#include "common.h"
#include <climits>

namespace {

constexpr int kTile = 1024;   // rows of pts2 staged per iteration (8 KiB)

// 64 rows of pts1 per workgroup, FOUR lanes per row: lane l of a row's quad scans entries l, l+4, ... of the staged
// pts2 tile branch-free (running minimum of the matching indices), the quad folds its minima after every tile and the
// workgroup stops at the first tile boundary where every row has its answer.  (One lane per row with an early `break`
// ran the whole wave at the pace of its slowest lane, one dependent LDS read per entry: 0.35 ms per frame.)
__global__ __launch_bounds__(256) void first_match_kernel(const float2* __restrict__ p1, int n1, const float2* __restrict__ p2,
                                                          int n2, int* __restrict__ first, unsigned char* __restrict__ keep2) {
    __shared__ float2 tile[kTile];
    const int l = threadIdx.x & 3;
    const int i = blockIdx.x * 64 + (threadIdx.x >> 2);
    const float2 a = i < n1 ? p1[i] : make_float2(0.f, 0.f);
    int found = INT_MAX;
    for (int base = 0; base < n2; base += kTile) {
        __syncthreads();
        for (int k = threadIdx.x; k < kTile && base + k < n2; k += 256) tile[k] = p2[base + k];
        __syncthreads();
        const int m = min(kTile, n2 - base);
#pragma unroll 4
        for (int k = l; k < m; k += 4) {
            const float2 b = tile[k];
            const bool hit = b.x == a.x || b.y == a.y;      // x OR y, exact float equality (NaN never matches, like NumPy)
            found = min(found, hit ? base + k : INT_MAX);
        }
        found = min(found, __shfl_xor(found, 1, 64));
        found = min(found, __shfl_xor(found, 2, 64));
        if (__syncthreads_and(found != INT_MAX || i >= n1)) break;
    }
    if (i < n1 && l == 0) {
        first[i] = found == INT_MAX ? -1 : found;
        if (found != INT_MAX) keep2[found] = 0;       // all writers store the same value
    }
}

__global__ __launch_bounds__(256) void keep_init_kernel(unsigned char* __restrict__ keep2, int n2) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < n2) keep2[r] = 1;
}

// single workgroup ordered compaction of (i, first[i]) pairs with first[i] >= 0
__global__ __launch_bounds__(1024) void assoc_compact_kernel(const int* __restrict__ first, int n1, int* __restrict__ idx1,
                                                             int* __restrict__ idx2, int* __restrict__ count) {
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base_s = 0;
    __syncthreads();
    for (int i0 = 0; i0 < n1; i0 += 1024) {
        const int i = i0 + threadIdx.x;
        const int f = i < n1 ? first[i] : -1;
        const unsigned long long bal = __ballot(f >= 0);
        if (lane == 0) wsum[wave] = __popcll(bal);
        __syncthreads();
        int woff = 0, total = 0;
        for (int w = 0; w < 16; ++w) {
            if (w < wave) woff += wsum[w];
            total += wsum[w];
        }
        if (f >= 0) {
            const int pos = base_s + woff + __popcll(bal & ((1ull << lane) - 1ull));
            idx1[pos] = i;
            idx2[pos] = f;
        }
        __syncthreads();
        if (threadIdx.x == 0) base_s += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) *count = base_s;
}

// Merge of partial 2-NN lists (train set split over S ranks / shards): one lane per query walks the 2 S candidates in shard
// order and keeps the two smallest by (distance, global train index) — the order a single scan of the whole train set
// produces (cv2.BFMatcher keeps the earlier row on a tie).  Missing neighbours are idx < 0.
__global__ __launch_bounds__(256) void knn_merge_top2_kernel(const int* __restrict__ cand /*[S][2][nq][2]: idx plane, dist-bits plane*/, int S,
                                                             int64_t nq, int* __restrict__ out_idx, float* __restrict__ out_dist) {
    const int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (q >= nq) return;
    float d0 = __builtin_huge_valf(), d1 = __builtin_huge_valf();
    int i0 = -1, i1 = -1;
    auto less = [](float da, int ia, float db, int ib) { return ib < 0 || da < db || (da == db && ia < ib); };
    for (int s = 0; s < S; ++s) {
        const int2 ci = *reinterpret_cast<const int2*>(cand + ((int64_t)(2 * s) * nq + q) * 2);
        const int2 cd = *reinterpret_cast<const int2*>(cand + ((int64_t)(2 * s + 1) * nq + q) * 2);
        const int ids[2] = {ci.x, ci.y};
        const float ds[2] = {__int_as_float(cd.x), __int_as_float(cd.y)};
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (ids[r] < 0) continue;
            if (less(ds[r], ids[r], d0, i0)) {
                d1 = d0; i1 = i0;
                d0 = ds[r]; i0 = ids[r];
            } else if (less(ds[r], ids[r], d1, i1)) {
                d1 = ds[r]; i1 = ids[r];
            }
        }
    }
    *reinterpret_cast<int2*>(out_idx + 2 * q) = make_int2(i0, i1);
    *reinterpret_cast<float2*>(out_dist + 2 * q) = make_float2(i0 >= 0 ? d0 : 0.f, i1 >= 0 ? d1 : 0.f);
}

// ---------------------------------------------------------------- mask -> ascending row indices
// `pts0 = pts0[mask.ravel() == 1]` (sfm.py:309, OpenCV's {0,1} essential-matrix mask), `pts0[mask.ravel() > 0]` (sfm.py:313, the
// {0,255} cheirality mask) and the complement of common_points (np.ma mask + compress, sfm.py:233-238): the rows whose mask byte
// passes, in ascending order.  Two multi-workgroup passes like the Lowe compaction (csrc/knn.hip): count per 1024 rows, then
// every workgroup sums its predecessors' counts and writes its rows at the right offset — deterministic, no atomics.
constexpr int kMaskBlock = 1024;
__device__ __forceinline__ unsigned mask_bits(const unsigned char* __restrict__ mask, int64_t n, int64_t r0, int mode) {
    unsigned pass = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (r0 + k < n) {
            const unsigned char m = mask[r0 + k];
            pass |= ((mode == 0 ? m == 1 : m != 0) ? 1u : 0u) << k;
        }
    return pass;
}
__global__ __launch_bounds__(256) void mask_count_kernel(const unsigned char* __restrict__ mask, int64_t n, int mode, int* __restrict__ block_count) {
    __shared__ int wsum[4];
    int c = __popc(mask_bits(mask, n, (int64_t)blockIdx.x * kMaskBlock + threadIdx.x * 4, mode));
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(256) void mask_scatter_kernel(const unsigned char* __restrict__ mask, int64_t n, int mode, const int* __restrict__ block_count,
                                                           int* __restrict__ out, int* __restrict__ count) {
    __shared__ int wsum[4];
    __shared__ int base_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int part = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += 256) part += block_count[b];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) part += __shfl_xor(part, m, 64);
    if (lane == 0) wsum[wave] = part;
    __syncthreads();
    if (threadIdx.x == 0) base_s = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    const int base = base_s;
    __syncthreads();
    const int64_t r0 = (int64_t)blockIdx.x * kMaskBlock + threadIdx.x * 4;
    const unsigned pass = mask_bits(mask, n, r0, mode);
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
    int pos = base + woff + incl - mine;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (pass & (1u << k)) out[pos++] = (int)(r0 + k);
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) *count = pos;
}

}  // namespace

extern "C" size_t sfm_mask_indices_ws_bytes(int64_t n) {
    if (n < 0) return 0;
    return sfm::align_up(sizeof(int) * (size_t)((n + kMaskBlock - 1) / kMaskBlock + 1), 256);
}

extern "C" int sfm_mask_indices(const uint8_t* mask, int64_t n, int mode, int32_t* idx_out, int32_t* count, void* ws, size_t ws_bytes, void* stream_) {
    SFM_CHECK_ARG(n >= 0 && n < ((int64_t)1 << 31) && (mode == 0 || mode == 1), "sfm_mask_indices: bad size or mode (0: byte == 1, 1: byte != 0)");
    SFM_CHECK_ARG(count && (n == 0 || (mask && idx_out)), "sfm_mask_indices: null pointer");
    hipStream_t stream = sfm::as_stream(stream_);
    if (n == 0) {
        SFM_CHECK_HIP(hipMemsetAsync(count, 0, sizeof(int32_t), stream));
        return SFM_OK;
    }
    if (!ws || ws_bytes < sfm_mask_indices_ws_bytes(n)) {
        sfm::set_error("sfm_mask_indices: workspace too small (%zu < %zu)", ws_bytes, sfm_mask_indices_ws_bytes(n));
        return SFM_ERR_WORKSPACE;
    }
    const unsigned blocks = (unsigned)((n + kMaskBlock - 1) / kMaskBlock);
    int* counts = static_cast<int*>(ws);
    hipLaunchKernelGGL(mask_count_kernel, dim3(blocks), dim3(256), 0, stream, mask, n, mode, counts);
    hipLaunchKernelGGL(mask_scatter_kernel, dim3(blocks), dim3(256), 0, stream, mask, n, mode, counts, idx_out, count);
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}

extern "C" int sfm_knn_merge_top2(const int32_t* cand, int shards, int64_t nq, int32_t* out_idx, float* out_dist, void* stream_) {
    SFM_CHECK_ARG(shards >= 1 && nq >= 0 && nq < ((int64_t)1 << 31), "sfm_knn_merge_top2: bad sizes");
    if (nq == 0) return SFM_OK;
    SFM_CHECK_ARG(cand && out_idx && out_dist, "sfm_knn_merge_top2: null pointer");
    SFM_CHECK_ARG(((uintptr_t)cand & 7) == 0 && ((uintptr_t)out_idx & 7) == 0 && ((uintptr_t)out_dist & 7) == 0, "sfm_knn_merge_top2: 8-byte alignment");
    hipStream_t stream = sfm::as_stream(stream_);
    hipLaunchKernelGGL(knn_merge_top2_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, stream, cand, shards, nq, out_idx, out_dist);
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}

extern "C" int sfm_common_points(const float* pts1, int64_t n1, const float* pts2, int64_t n2, int32_t* first_ws,
                                 int32_t* idx1, int32_t* idx2, int32_t* count, uint8_t* keep2, void* stream_) {
    SFM_CHECK_ARG(n1 >= 0 && n2 >= 0 && n1 < (1 << 30) && n2 < (1 << 30), "sfm_common_points: bad sizes");
    SFM_CHECK_ARG(count && (n1 == 0 || (pts1 && first_ws && idx1 && idx2)) && (n2 == 0 || (pts2 && keep2)),
                  "sfm_common_points: null pointer");
    hipStream_t stream = sfm::as_stream(stream_);
    if (n2 > 0) {
        hipLaunchKernelGGL(keep_init_kernel, dim3((unsigned)((n2 + 255) / 256)), dim3(256), 0, stream, keep2, (int)n2);
        SFM_CHECK_LAUNCH();
    }
    if (n1 == 0) {
        SFM_CHECK_HIP(hipMemsetAsync(count, 0, sizeof(int32_t), stream));
        return SFM_OK;
    }
    hipLaunchKernelGGL(first_match_kernel, dim3((unsigned)((n1 + 63) / 64)), dim3(256), 0, stream,
                       reinterpret_cast<const float2*>(pts1), (int)n1, reinterpret_cast<const float2*>(pts2), (int)n2, first_ws,
                       keep2);
    SFM_CHECK_LAUNCH();
    hipLaunchKernelGGL(assoc_compact_kernel, dim3(1), dim3(1024), 0, stream, first_ws, (int)n1, idx1, idx2, count);
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}
