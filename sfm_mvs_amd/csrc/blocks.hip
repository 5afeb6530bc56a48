// Small batched block kernels of the Schur-complement solver (the reference hands the problem to SciPy's
// least_squares, sfm.py:146) and the facade's cv2.norm (sfm.py:93,95):
//   sfm_block_inverse   n independent k x k blocks (k = 3: point blocks C_j, k = 6: camera blocks B_i) inverted in
//                       registers, one lane per block, Gauss-Jordan with partial pivoting
//   sfm_block_matvec    y_i = A_i x_i for n blocks (block-Jacobi preconditioner apply, C^-1 g products)
//   sfm_norm_l2         cv2.norm(a, b, NORM_L2): differences in the inputs' type, squares accumulated in double,
//                       fixed-order reduction (lane tree -> waves -> workgroups): deterministic
// HBM-bound streaming kernels: 8 k^2 B in + 8 k^2 B out per block (inverse), 8 (k^2 + 2k) B per block (matvec).
#include "common.h"

namespace {

template <int K>
__global__ __launch_bounds__(256) void block_inverse_kernel(const double* __restrict__ A, int64_t n, double* __restrict__ out,
                                                            int* __restrict__ bad /*optional: number of singular / non-finite blocks*/) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a[K][K], b[K][K];
    bool singular = false;
#pragma unroll
    for (int r = 0; r < K; ++r)
#pragma unroll
        for (int c = 0; c < K; ++c) {
            a[r][c] = A[i * K * K + r * K + c];
            b[r][c] = r == c ? 1.0 : 0.0;
        }
#pragma unroll
    for (int col = 0; col < K; ++col) {
        // partial pivoting without dynamic register indexing: swap row `col` with every later row whose entry is larger
#pragma unroll
        for (int r = col + 1; r < K; ++r) {
            const bool sw = fabs(a[r][col]) > fabs(a[col][col]);
#pragma unroll
            for (int c = 0; c < K; ++c) {
                const double ta = a[col][c], tb = b[col][c];
                a[col][c] = sw ? a[r][c] : ta;
                a[r][c] = sw ? ta : a[r][c];
                b[col][c] = sw ? b[r][c] : tb;
                b[r][c] = sw ? tb : b[r][c];
            }
        }
        // a zero or non-finite pivot (after partial pivoting: the block is singular, or holds inf / NaN) is REPORTED: the
        // inverse of that block comes out non-finite as before, and the status word lets the caller refuse it
        singular = singular || !(fabs(a[col][col]) > 0.0) || !(fabs(a[col][col]) < __builtin_huge_val());
        const double d = 1.0 / a[col][col];
#pragma unroll
        for (int c = 0; c < K; ++c) {
            a[col][c] *= d;
            b[col][c] *= d;
        }
#pragma unroll
        for (int r = 0; r < K; ++r) {
            if (r == col) continue;
            const double f = a[r][col];
#pragma unroll
            for (int c = 0; c < K; ++c) {
                a[r][c] -= f * a[col][c];
                b[r][c] -= f * b[col][c];
            }
        }
    }
#pragma unroll
    for (int r = 0; r < K; ++r)
#pragma unroll
        for (int c = 0; c < K; ++c) out[i * K * K + r * K + c] = b[r][c];
    if (bad && singular) atomicAdd(bad, 1);                       // (rare path: integer count, order-independent)
}

template <int K>
__global__ __launch_bounds__(256) void block_matvec_kernel(const double* __restrict__ A, const double* __restrict__ x, int64_t n,
                                                           double* __restrict__ y) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i >= n) return;
    double xv[K];
#pragma unroll
    for (int c = 0; c < K; ++c) xv[c] = x[i * K + c];
#pragma unroll
    for (int r = 0; r < K; ++r) {
        double s = 0;
#pragma unroll
        for (int c = 0; c < K; ++c) s += A[i * K * K + r * K + c] * xv[c];
        y[i * K + r] = s;
    }
}

constexpr int kNormBlocksMax = 256;

template <typename T>
__global__ __launch_bounds__(256) void norm_partial_kernel(const T* __restrict__ a, const T* __restrict__ b, int64_t n,
                                                           double* __restrict__ partial) {
    __shared__ double w[4];
    double s = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const T d = b ? (T)(a[i] - b[i]) : a[i];          // difference in the inputs' type, as cv2.norm takes it
        s += (double)d * (double)d;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
    if ((threadIdx.x & 63) == 0) w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = ((w[0] + w[1]) + w[2]) + w[3];
}

__global__ void norm_final_kernel(const double* __restrict__ partial, int blocks, double* __restrict__ out) {
    double s = 0;
    for (int b = 0; b < blocks; ++b) s += partial[b];
    *out = sqrt(s);
}

}  // namespace

extern "C" int sfm_block_inverse_checked(const double* A, int64_t n, int k, double* Ainv, int32_t* bad_count_dev, void* stream_) {
    SFM_CHECK_ARG(n >= 0 && (k == 3 || k == 6), "sfm_block_inverse: k must be 3 or 6 (got %d)", k);
    if (bad_count_dev) SFM_CHECK_HIP(hipMemsetAsync(bad_count_dev, 0, sizeof(int32_t), sfm::as_stream(stream_)));
    if (n == 0) return SFM_OK;
    SFM_CHECK_ARG(A && Ainv, "sfm_block_inverse: null pointer");
    const dim3 grid((unsigned)((n + 255) / 256));
    if (k == 3)
        hipLaunchKernelGGL(block_inverse_kernel<3>, grid, dim3(256), 0, sfm::as_stream(stream_), A, n, Ainv, bad_count_dev);
    else
        hipLaunchKernelGGL(block_inverse_kernel<6>, grid, dim3(256), 0, sfm::as_stream(stream_), A, n, Ainv, bad_count_dev);
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}

extern "C" int sfm_block_inverse(const double* A, int64_t n, int k, double* Ainv, void* stream_) {
    return sfm_block_inverse_checked(A, n, k, Ainv, nullptr, stream_);
}

extern "C" int sfm_block_matvec(const double* A, const double* x, int64_t n, int k, double* y, void* stream_) {
    SFM_CHECK_ARG(n >= 0 && (k == 3 || k == 6), "sfm_block_matvec: k must be 3 or 6 (got %d)", k);
    if (n == 0) return SFM_OK;
    SFM_CHECK_ARG(A && x && y, "sfm_block_matvec: null pointer");
    const dim3 grid((unsigned)((n + 255) / 256));
    if (k == 3)
        hipLaunchKernelGGL(block_matvec_kernel<3>, grid, dim3(256), 0, sfm::as_stream(stream_), A, x, n, y);
    else
        hipLaunchKernelGGL(block_matvec_kernel<6>, grid, dim3(256), 0, sfm::as_stream(stream_), A, x, n, y);
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}

extern "C" size_t sfm_norm_l2_ws_bytes(void) { return sizeof(double) * kNormBlocksMax + 256; }

extern "C" int sfm_norm_l2(const void* a, const void* b, int64_t n, int is_f64, double* out_dev, void* ws, size_t ws_bytes,
                           void* stream_) {
    SFM_CHECK_ARG(n >= 0 && out_dev && (n == 0 || a), "sfm_norm_l2: bad argument");
    if (!ws || ws_bytes < sfm_norm_l2_ws_bytes()) {
        sfm::set_error("sfm_norm_l2: workspace too small (%zu < %zu)", ws_bytes, sfm_norm_l2_ws_bytes());
        return SFM_ERR_WORKSPACE;
    }
    hipStream_t stream = sfm::as_stream(stream_);
    double* partial = reinterpret_cast<double*>(sfm::align_up((size_t)(uintptr_t)ws, 256));
    const int blocks = (int)std::min<int64_t>(std::max<int64_t>((n + 2047) / 2048, 1), kNormBlocksMax);
    if (is_f64)
        hipLaunchKernelGGL(norm_partial_kernel<double>, dim3(blocks), dim3(256), 0, stream, static_cast<const double*>(a),
                           static_cast<const double*>(b), n, partial);
    else
        hipLaunchKernelGGL(norm_partial_kernel<float>, dim3(blocks), dim3(256), 0, stream, static_cast<const float*>(a),
                           static_cast<const float*>(b), n, partial);
    hipLaunchKernelGGL(norm_final_kernel, dim3(1), dim3(1), 0, stream, partial, blocks, out_dev);
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}
