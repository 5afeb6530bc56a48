// SIFT detectAndCompute and the two preprocessing calls in front of it, for gfx950.
//
// Replaces  cv2.pyrDown (sfm.py:40), cv2.cvtColor(BGR2GRAY) (sfm.py:243-244) and
// cv2.xfeatures2d.SIFT_create().detectAndCompute(gray, None) (sfm.py:246-252).
//
// Everything is stream-ordered device work; nothing is copied to the host:
//   upsample 2x (u8 -> f32, bilinear) -> per octave { fused separable Gaussian (LDS tile, rows then columns) that also
//   emits the DoG plane; 2x decimation is a stride-2 read of the next octave's first blur } -> extrema of all octaves (one lane per column and four rows, raw list) -> sub-pixel
//   refinement (one lane per raw extremum) -> gradient (magnitude, orientation) planes -> orientation histograms (one
//   wave per extremum) -> x-bucketed ranking in OpenCV's keypoint order + duplicate removal (ordered compaction) ->
//   descriptors (one wave per keypoint, four lanes per histogram cell).
//
// Arithmetic contract: float32 operations in the order of the sequential algorithm (oracle/sift_oracle.c restates
// it), no contraction (-ffp-contract=off), IEEE divide/sqrt, and fixed polynomial programs for exp / sincos / atan2,
// so keypoints and descriptors are reproducible bit for bit.  Histogram sums, which are order-sensitive, are
// accumulated by an owner lane per bin that walks the samples in raster order.
//
// Memory: one workspace holds the whole scale space: for a W x H input the doubled base is 2W x 2H floats and the
// pyramid (6 Gaussian + 5 DoG planes + 3 float2 gradient planes per octave at the defaults) takes 17 * 4/3 * 16 WH
// bytes plus the keypoint lists (268 MB for the reference's 968 x 648 frames and 131 072 keypoint slots) — resident in
// HBM from the first blur to the last descriptor.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include "common.h"

namespace {

constexpr int kImgBorder = 5, kMaxInterpSteps = 5, kOriBins = 36, kDescWidth = 4, kDescBins = 8;
constexpr float kOriSigFctr = 1.5f, kOriRadius = 4.5f, kOriPeakRatio = 0.8f, kDescSclFctr = 3.f, kDescMagThr = 0.2f, kIntDescrFctr = 512.f;
constexpr int kMaxTaps = 55;   // filter length limit (LDS tile of the blur kernel stays under 64 KB): sigma <= 6.8 per blur
constexpr int kTileW = 64, kTileH = 32;
constexpr int kMaxOctaves = 16;

struct Taps { float k[kMaxTaps + 1]; int n; };

// device counters: [0] refined extrema, [1] keypoints (one per orientation peak), [2] keypoints after duplicate removal,
// [3] raw extrema (before refinement; summed from the kRawSegs segment counters at kCntRawSeg), [4] largest overflowing
// raw segment count (0 = none), [5] the same for the keypoint segments at kCntKpSeg, [kCntHist..) keypoints per x range, [kCntCursor..) scatter cursors of the ordering pass
constexpr int kXBuckets = 256, kCntHist = 64, kCntCursor = kCntHist + kXBuckets;
constexpr int kRawSegs = 64, kSegStride = 32, kCntRawSeg = kCntCursor + kXBuckets, kCntKpSeg = kCntRawSeg + kRawSegs * kSegStride,
              kCounterInts = kCntKpSeg + kRawSegs * kSegStride;
__device__ inline int x_bucket(float x, int W0) { return min(kXBuckets - 1, max(0, (int)(x * ((float)kXBuckets / (float)W0)))); }


// Scale-space geometry (octave o of a doubled base W0 x H0): planes are W0>>o by H0>>o; Gaussian planes of all
// octaves are packed back to back, (nL+3) per octave, DoG planes (nL+2) per octave in a second region.
struct Geom {
    float* g; float* d; float2* grad; int W0, H0, nL, nOct;   // grad: (magnitude, orientation) pairs
    __device__ __host__ int w(int o) const { return W0 >> o; }
    __device__ __host__ int h(int o) const { return H0 >> o; }
    __device__ __host__ size_t plane(int o) const { return (size_t)w(o) * h(o); }
    __device__ __host__ size_t goff(int o) const { size_t s = 0; for (int p = 0; p < o; ++p) s += plane(p) * (nL + 3); return s; }
    __device__ __host__ size_t doff(int o) const { size_t s = 0; for (int p = 0; p < o; ++p) s += plane(p) * (nL + 2); return s; }
    __device__ __host__ float* G(int o, int i) const { return g + goff(o) + plane(o) * i; }
    __device__ __host__ float* D(int o, int i) const { return d + doff(o) + plane(o) * i; }
    // gradient magnitude / orientation of Gaussian layers 1..nL (the layers keypoints live in), nL planes per octave
    __device__ __host__ size_t moff(int o, int layer) const { size_t s = 0; for (int p = 0; p < o; ++p) s += plane(p) * nL; return s + plane(o) * (layer - 1); }
};

__device__ inline int reflect101(int p, int len) {   // BORDER_REFLECT_101, any distance (period 2 len - 2)
    if ((unsigned)p < (unsigned)len) return p;
    if (len == 1) return 0;
    const int period = 2 * len - 2;
    p %= period;
    if (p < 0) p += period;
    return p < len ? p : period - p;
}
__device__ inline int cv_round(float v) { return __float2int_rn(v); }
__device__ inline int cv_floor(float v) { return (int)floorf(v); }

__device__ inline float sift_expf(float x) {   // Cephes expf, fixed operation order (see the oracle)
    if (x < -80.f) x = -80.f;
    if (x > 80.f) x = 80.f;
    const float n = rintf(x * 1.44269504f);
    float r = x - n * 0.693359375f;
    r = r - n * -2.12194440e-4f;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    const float y = p * (r * r) + r + 1.0f;
    return y * __uint_as_float((unsigned)((int)n + 127) << 23);
}

__device__ inline float sift_expf_unclamped(float x) {   // sift_expf for |x| < 80, where its clamps are inactive
    const float n = rintf(x * 1.44269504f);
    float r = x - n * 0.693359375f;
    r = r - n * -2.12194440e-4f;
    float p = 1.9875691500e-4f;
    p = p * r + 1.3981999507e-3f;
    p = p * r + 8.3334519073e-3f;
    p = p * r + 4.1665795894e-2f;
    p = p * r + 1.6666665459e-1f;
    p = p * r + 5.0000001201e-1f;
    const float y = p * (r * r) + r + 1.0f;
    return y * __uint_as_float((unsigned)((int)n + 127) << 23);
}

__device__ inline void sift_sincos(float xf, float* s, float* c) {
    const double x = (double)xf;
    const double kq = rint(x * 0.63661977236758134308);
    const double r = (x - kq * 1.57079632679489655800) - kq * 6.123233995736766e-17;
    const double r2 = r * r;
    double sp = -7.6471637318198164759e-13;
    sp = sp * r2 + 1.6059043836821614599e-10;
    sp = sp * r2 - 2.5052108385441718775e-8;
    sp = sp * r2 + 2.7557319223985890653e-6;
    sp = sp * r2 - 1.9841269841269841270e-4;
    sp = sp * r2 + 8.3333333333333333333e-3;
    sp = sp * r2 - 1.6666666666666666667e-1;
    const double sn = r + r * r2 * sp;
    double cp = 4.7794773323873852974e-14;
    cp = cp * r2 - 1.1470745597729724714e-11;
    cp = cp * r2 + 2.0876756987868098979e-9;
    cp = cp * r2 - 2.7557319223985890653e-7;
    cp = cp * r2 + 2.4801587301587301587e-5;
    cp = cp * r2 - 1.3888888888888888889e-3;
    cp = cp * r2 + 4.1666666666666666667e-2;
    const double cs = 1.0 - 0.5 * r2 + r2 * r2 * cp;
    const int q = (int)((long long)kq & 3);
    *s = (float)(q == 0 ? sn : q == 1 ? cs : q == 2 ? -sn : -cs);
    *c = (float)(q == 0 ? cs : q == 1 ? -sn : q == 2 ? -cs : sn);
}

__device__ inline float fast_atan2_deg(float y, float x) {
    const float p1 = 0.9997878412794807f * 57.29577951308232f, p3 = -0.3258083974640975f * 57.29577951308232f,
                p5 = 0.1555786518463281f * 57.29577951308232f, p7 = -0.04432655554792128f * 57.29577951308232f;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) { c = ay / (ax + (float)DBL_EPSILON); c2 = c * c; a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    else          { c = ax / (ay + (float)DBL_EPSILON); c2 = c * c; a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c; }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// ------------------------------------------------------------------------------------------------ preprocessing

__global__ __launch_bounds__(256) void bgr2gray_kernel(const unsigned char* __restrict__ bgr, int w, int h, long stride, unsigned char* __restrict__ gray) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= w || y >= h) return;
    const unsigned char* p = bgr + (size_t)y * stride + 3 * x;
    gray[(size_t)y * w + x] = (unsigned char)((p[0] * 1868 + p[1] * 9617 + p[2] * 4899 + (1 << 13)) >> 14);
}

__global__ __launch_bounds__(256) void pyrdown_kernel(const unsigned char* __restrict__ src, int w, int h, int ch, unsigned char* __restrict__ dst) {
    const int dw = (w + 1) / 2, dh = (h + 1) / 2;
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= dw || y >= dh) return;
    for (int c = 0; c < ch; ++c) {
        int s = 0;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int sy = reflect101(2 * y + i - 2, h);
            int r = 0;
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int kj = j == 0 || j == 4 ? 1 : j == 2 ? 6 : 4;
                r += kj * src[((size_t)sy * w + reflect101(2 * x + j - 2, w)) * ch + c];
            }
            s += (i == 0 || i == 4 ? 1 : i == 2 ? 6 : 4) * r;
        }
        dst[((size_t)y * dw + x) * ch + c] = (unsigned char)((s + 128) >> 8);
    }
}

// resize(float(gray), 2x, INTER_LINEAR): horizontal taps first, then vertical, indices clamped at the border
__global__ __launch_bounds__(256) void upsample2_kernel(const unsigned char* __restrict__ gray, int w, int h, long stride, float* __restrict__ out) {
    const int dx = blockIdx.x * 64 + (threadIdx.x & 63), dy = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (dx >= 2 * w || dy >= 2 * h) return;
    float fx = (float)((dx + 0.5) * 0.5 - 0.5);
    int sx = cv_floor(fx); fx -= sx;
    if (sx < 0) { sx = 0; fx = 0; }
    if (sx >= w - 1) { sx = w - 1; fx = 0; }
    const int sx1 = sx + 1 < w ? sx + 1 : w - 1;
    float fy = (float)((dy + 0.5) * 0.5 - 0.5);
    int sy = cv_floor(fy); fy -= sy;
    if (sy < 0) { sy = 0; fy = 0; }
    if (sy >= h - 1) { sy = h - 1; fy = 0; }
    const int sy1 = sy + 1 < h ? sy + 1 : h - 1;
    const unsigned char *r0 = gray + (size_t)sy * stride, *r1 = gray + (size_t)sy1 * stride;
    const float a = (float)r0[sx] * (1.f - fx) + (float)r0[sx1] * fx;
    const float b = (float)r1[sx] * (1.f - fx) + (float)r1[sx1] * fx;
    out[(size_t)dy * 2 * w + dx] = a * (1.f - fy) + b * fy;
}


// ------------------------------------------------------------------------------------------------ Gaussian + DoG
// One 64 x 32 output tile per workgroup.  The (32+2R) x (64+2R) input window (BORDER_REFLECT_101) goes to LDS once,
// the row pass writes (32+2R) x 64 partial results back to LDS, the column pass reads them: each input pixel is
// fetched from HBM/L2 about (1+2R/64)(1+2R/32) times instead of 2(2R+1) times.  `dog` (optional) = dst - src.
// `src` is read through a view: pixel (y, x) of the w x h input is src[(y * sstep) * spitch + x * sstep]; sstep = 2 with the
// previous octave's pitch makes the first blur of an octave read the 2x decimation of layer nL directly (the
// decimated plane itself is never materialised: nothing else reads layer 0 of octaves >= 1).
__global__ __launch_bounds__(256) void gauss_blur_kernel(const float* __restrict__ src, float* __restrict__ dst, float* __restrict__ dog,
                                                         int w, int h, Taps taps, int sstep, int spitch) {
    extern __shared__ float blur_lds[];      // tk[kMaxTaps+1] | th[rows x 64] | tin[rows x pitch]
    const int n = taps.n, R = n >> 1;
    const int x0 = blockIdx.x * kTileW, y0 = blockIdx.y * kTileH;
    const int pitch = kTileW + 2 * R, rows = kTileH + 2 * R;
    float* tk = blur_lds;
    float* th = tk + kMaxTaps + 1;
    float* tin = th + rows * kTileW;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 0; i < kMaxTaps; ++i) tk[i] = taps.k[i];
    }
    for (int e = threadIdx.x; e < rows * pitch; e += 256) {
        const int ry = e / pitch, rx = e - ry * pitch;
        tin[e] = src[(size_t)(reflect101(y0 - R + ry, h) * sstep) * spitch + reflect101(x0 - R + rx, w) * sstep];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < rows * kTileW; e += 256) {
        const int ry = e >> 6, cx = e & 63;
        const float* p = tin + ry * pitch + cx;
        float s = p[0] * tk[0];
        for (int j = 1; j < n; ++j) s += p[j] * tk[j];
        th[e] = s;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < kTileH * kTileW; e += 256) {
        const int ty = e >> 6, tx = e & 63;
        const int x = x0 + tx, y = y0 + ty;
        if (x >= w || y >= h) continue;
        const float* p = th + (ty + R) * kTileW + tx;
        float s = tk[R] * p[0];
        for (int j = 1; j <= R; ++j) s += tk[R + j] * (p[j * kTileW] + p[-j * kTileW]);
        dst[(size_t)y * w + x] = s;
        if (dog) dog[(size_t)y * w + x] = s - tin[(ty + R) * pitch + tx + R];
    }
}

// The same filter with the tap count known at compile time (the defaults need 11, 13, 17, 21 and 27 taps): taps live
// in scalar registers, the row pass computes four adjacent outputs per lane from one run of 128-bit LDS reads
// ((N+3)/4 reads per output instead of N, plus no tap reads), the column pass works on float4 columns.  Every output
// is still the sequential dot product in tap order (rows) / the symmetric form (columns): results are identical to
// gauss_blur_kernel's.  A single tile of this kernel is also what bounds the small octaves (one workgroup each).
constexpr int kBlurThreads = 512;    // per 64 x 32 tile: a tile's latency (what the one-workgroup octaves and the last round of the big ones pay) halves vs 256
template <int N>
struct BlurShape {
    static constexpr int R = N / 2, ROWS = kTileH + 2 * R, COLS = kTileW + 2 * R, PITCH = (COLS + 3) & ~3, NV = (N + 3 + 3) / 4;
    static constexpr int kLdsFloats = ROWS * PITCH + ROWS * kTileW;      // tin | th
};
// one 64 x 32 tile (bx, by) of one blur; lds: BlurShape<N>::kLdsFloats floats, 16-byte aligned
template <int N>
__device__ __forceinline__ void blur_fixed_tile(const float* __restrict__ src, float* __restrict__ dst, float* __restrict__ dog, int w, int h,
                                                const Taps& taps, int sstep, int spitch, int bx, int by, float* __restrict__ lds) {
    constexpr int R = BlurShape<N>::R, ROWS = BlurShape<N>::ROWS, COLS = BlurShape<N>::COLS, PITCH = BlurShape<N>::PITCH, NV = BlurShape<N>::NV;
    float* const tin = lds;
    float* const th = lds + ROWS * PITCH;
    const int x0 = bx * kTileW, y0 = by * kTileH;
    constexpr int LOADS = (ROWS * COLS + kBlurThreads - 1) / kBlurThreads;     // all of a lane's loads are issued before the first is consumed
    float ld[LOADS];
    if (x0 >= R && y0 >= R && x0 + kTileW + R <= w && y0 + kTileH + R <= h) {     // interior tile: no border arithmetic
        const float* base = src + (size_t)((y0 - R) * sstep) * spitch + (x0 - R) * sstep;
#pragma unroll
        for (int k = 0; k < LOADS; ++k) {
            const int e = threadIdx.x + kBlurThreads * k, ry = e / COLS, rx = e - ry * COLS;
            ld[k] = e < ROWS * COLS ? base[(size_t)(ry * sstep) * spitch + rx * sstep] : 0.f;
        }
    } else {
#pragma unroll
        for (int k = 0; k < LOADS; ++k) {
            const int e = threadIdx.x + kBlurThreads * k, ry = e / COLS, rx = e - ry * COLS;
            ld[k] = e < ROWS * COLS ? src[(size_t)(reflect101(y0 - R + ry, h) * sstep) * spitch + reflect101(x0 - R + rx, w) * sstep] : 0.f;
        }
    }
#pragma unroll
    for (int k = 0; k < LOADS; ++k) {
        const int e = threadIdx.x + kBlurThreads * k, ry = e / COLS, rx = e - ry * COLS;
        if (e < ROWS * COLS) tin[ry * PITCH + rx] = ld[k];
    }
    __syncthreads();
    for (int it = threadIdx.x; it < ROWS * (kTileW / 4); it += kBlurThreads) {
        const int ry = it >> 4, qx = it & 15;
        const float4* p = reinterpret_cast<const float4*>(tin + ry * PITCH + 4 * qx);
        float v[4 * NV];
#pragma unroll
        for (int k = 0; k < NV; ++k) { const float4 t = p[k]; v[4 * k] = t.x; v[4 * k + 1] = t.y; v[4 * k + 2] = t.z; v[4 * k + 3] = t.w; }
        float o[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            float a = v[m] * taps.k[0];
#pragma unroll
            for (int j = 1; j < N; ++j) a += v[m + j] * taps.k[j];
            o[m] = a;
        }
        *reinterpret_cast<float4*>(th + ry * kTileW + 4 * qx) = make_float4(o[0], o[1], o[2], o[3]);
    }
    __syncthreads();
    for (int it = threadIdx.x; it < kTileH * (kTileW / 4); it += kBlurThreads) {
        const int ty = it >> 4, qx = it & 15;
        const int x = x0 + 4 * qx, y = y0 + ty;
        if (x >= w || y >= h) continue;
        const float* p = th + (ty + R) * kTileW + 4 * qx;
        const float4 c = *reinterpret_cast<const float4*>(p);
        float o[4] = {taps.k[R] * c.x, taps.k[R] * c.y, taps.k[R] * c.z, taps.k[R] * c.w};
#pragma unroll
        for (int j = 1; j <= R; ++j) {
            const float4 a = *reinterpret_cast<const float4*>(p + j * kTileW), b = *reinterpret_cast<const float4*>(p - j * kTileW);
            o[0] += taps.k[R + j] * (a.x + b.x); o[1] += taps.k[R + j] * (a.y + b.y);
            o[2] += taps.k[R + j] * (a.z + b.z); o[3] += taps.k[R + j] * (a.w + b.w);
        }
        const float* cin = tin + (ty + R) * PITCH + 4 * qx + R;
        float* d = dst + (size_t)y * w + x;
        float* g = dog ? dog + (size_t)y * w + x : nullptr;
        if (x + 3 < w) {
            *reinterpret_cast<float4*>(d) = make_float4(o[0], o[1], o[2], o[3]);
            if (g) *reinterpret_cast<float4*>(g) = make_float4(o[0] - cin[0], o[1] - cin[1], o[2] - cin[2], o[3] - cin[3]);
        } else {
            for (int m = 0; m < 4 && x + m < w; ++m) { d[m] = o[m]; if (g) g[m] = o[m] - cin[m]; }
        }
    }
}

template <int N>
__global__ __launch_bounds__(kBlurThreads) void gauss_blur_fixed_kernel(const float* __restrict__ src, float* __restrict__ dst, float* __restrict__ dog,
                                                               int w, int h, Taps taps, int sstep, int spitch) {
    __shared__ __attribute__((aligned(16))) float lds[BlurShape<N>::kLdsFloats];
    blur_fixed_tile<N>(src, dst, dog, w, h, taps, sstep, spitch, blockIdx.x, blockIdx.y, lds);
}

// TWO independent blurs in one launch (tiles of A first, then B's): the scale space is a chain of launches — an octave's layer
// i needs layer i - 1, the next octave needs layer nL — and from octave 3 on every launch is a handful of workgroups, i.e.
// launch latency.  Layers nL + 1 and nL + 2 of an octave (which only the DoG / gradient planes need) depend on the same
// finished layer nL as layers 1 and 2 of the NEXT octave: each of them shares a launch with one of those, and the chain is
// 3 launches per octave instead of 5 (46 -> 30 per frame at the defaults).  Same tile code, same results.
struct BlurJob {
    const float* src; float* dst; float* dog;
    int w, h, sstep, spitch, tiles_x, tiles;
    Taps taps;
};
template <int NA, int NB>
__global__ __launch_bounds__(kBlurThreads) void gauss_blur_pair_kernel(BlurJob A, BlurJob B) {
    constexpr int kLds = BlurShape<NA>::kLdsFloats > BlurShape<NB>::kLdsFloats ? BlurShape<NA>::kLdsFloats : BlurShape<NB>::kLdsFloats;
    __shared__ __attribute__((aligned(16))) float lds[kLds];
    const int b = blockIdx.x;
    if (b < A.tiles) blur_fixed_tile<NA>(A.src, A.dst, A.dog, A.w, A.h, A.taps, A.sstep, A.spitch, b % A.tiles_x, b / A.tiles_x, lds);
    else blur_fixed_tile<NB>(B.src, B.dst, B.dog, B.w, B.h, B.taps, B.sstep, B.spitch, (b - A.tiles) % B.tiles_x, (b - A.tiles) / B.tiles_x, lds);
}

// ------------------------------------------------------------------------------------------------ extrema
struct Cand { int o, layer, r, c; float x, y, size, response; int octave; };

__device__ inline void solve3(const float (&a)[3][3], const float (&b)[3], float (&x)[3]) {   // Matx33f::solve: Cramer
    float d = a[0][0] * (a[1][1] * a[2][2] - a[2][1] * a[1][2]) - a[0][1] * (a[1][0] * a[2][2] - a[2][0] * a[1][2]) +
              a[0][2] * (a[1][0] * a[2][1] - a[2][0] * a[1][1]);
    if (d == 0) { x[0] = x[1] = x[2] = 0; return; }
    d = 1 / d;
    x[0] = d * (b[0] * (a[1][1] * a[2][2] - a[1][2] * a[2][1]) - a[0][1] * (b[1] * a[2][2] - a[1][2] * b[2]) +
                a[0][2] * (b[1] * a[2][1] - a[1][1] * b[2]));
    x[1] = d * (a[0][0] * (b[1] * a[2][2] - a[1][2] * b[2]) - b[0] * (a[1][0] * a[2][2] - a[1][2] * a[2][0]) +
                a[0][2] * (a[1][0] * b[2] - b[1] * a[2][0]));
    x[2] = d * (a[0][0] * (a[1][1] * b[2] - b[1] * a[2][1]) - a[0][1] * (a[1][0] * b[2] - b[1] * a[2][0]) +
                b[0] * (a[1][0] * a[2][1] - a[1][1] * a[2][0]));
}

__device__ bool adjust_local_extrema(const float* __restrict__ dog, int w, int h, size_t plane, int octv, int layer, int r, int c, int nL,
                                     float contrastThreshold, float edgeThreshold, float sigma, Cand* out) {
    const float img_scale = 1.f / 255, deriv_scale = img_scale * 0.5f, second_deriv_scale = img_scale, cross_deriv_scale = img_scale * 0.25f;
    float xi = 0, xr = 0, xc = 0;
    int i = 0;
#define PXI(l, rr, cc) dog[(size_t)(l) * plane + (size_t)(rr) * w + (cc)]
    for (; i < kMaxInterpSteps; ++i) {
        const float dD[3] = {(PXI(layer, r, c + 1) - PXI(layer, r, c - 1)) * deriv_scale, (PXI(layer, r + 1, c) - PXI(layer, r - 1, c)) * deriv_scale,
                             (PXI(layer + 1, r, c) - PXI(layer - 1, r, c)) * deriv_scale};
        const float v2 = PXI(layer, r, c) * 2;
        const float dxx = (PXI(layer, r, c + 1) + PXI(layer, r, c - 1) - v2) * second_deriv_scale;
        const float dyy = (PXI(layer, r + 1, c) + PXI(layer, r - 1, c) - v2) * second_deriv_scale;
        const float dss = (PXI(layer + 1, r, c) + PXI(layer - 1, r, c) - v2) * second_deriv_scale;
        const float dxy = (PXI(layer, r + 1, c + 1) - PXI(layer, r + 1, c - 1) - PXI(layer, r - 1, c + 1) + PXI(layer, r - 1, c - 1)) * cross_deriv_scale;
        const float dxs = (PXI(layer + 1, r, c + 1) - PXI(layer + 1, r, c - 1) - PXI(layer - 1, r, c + 1) + PXI(layer - 1, r, c - 1)) * cross_deriv_scale;
        const float dys = (PXI(layer + 1, r + 1, c) - PXI(layer + 1, r - 1, c) - PXI(layer - 1, r + 1, c) + PXI(layer - 1, r - 1, c)) * cross_deriv_scale;
        const float H[3][3] = {{dxx, dxy, dxs}, {dxy, dyy, dys}, {dxs, dys, dss}};
        float X[3];
        solve3(H, dD, X);
        xi = -X[2]; xr = -X[1]; xc = -X[0];
        if (fabsf(xi) < 0.5f && fabsf(xr) < 0.5f && fabsf(xc) < 0.5f) break;
        if (fabsf(xi) > (float)(INT32_MAX / 3) || fabsf(xr) > (float)(INT32_MAX / 3) || fabsf(xc) > (float)(INT32_MAX / 3)) return false;
        c += cv_round(xc); r += cv_round(xr); layer += cv_round(xi);
        if (layer < 1 || layer > nL || c < kImgBorder || c >= w - kImgBorder || r < kImgBorder || r >= h - kImgBorder) return false;
    }
    if (i >= kMaxInterpSteps) return false;
    const float dD[3] = {(PXI(layer, r, c + 1) - PXI(layer, r, c - 1)) * deriv_scale, (PXI(layer, r + 1, c) - PXI(layer, r - 1, c)) * deriv_scale,
                         (PXI(layer + 1, r, c) - PXI(layer - 1, r, c)) * deriv_scale};
    const float t = dD[0] * xc + dD[1] * xr + dD[2] * xi;
    const float contr = PXI(layer, r, c) * img_scale + t * 0.5f;
    if (fabsf(contr) * nL < contrastThreshold) return false;
    const float v2 = PXI(layer, r, c) * 2.f;
    const float dxx = (PXI(layer, r, c + 1) + PXI(layer, r, c - 1) - v2) * second_deriv_scale;
    const float dyy = (PXI(layer, r + 1, c) + PXI(layer, r - 1, c) - v2) * second_deriv_scale;
    const float dxy = (PXI(layer, r + 1, c + 1) - PXI(layer, r + 1, c - 1) - PXI(layer, r - 1, c + 1) + PXI(layer, r - 1, c - 1)) * cross_deriv_scale;
#undef PXI
    const float tr = dxx + dyy, det = dxx * dyy - dxy * dxy;
    if (det <= 0 || tr * tr * edgeThreshold >= (edgeThreshold + 1) * (edgeThreshold + 1) * det) return false;
    out->o = octv; out->layer = layer; out->r = r; out->c = c;
    out->x = (c + xc) * (1 << octv);
    out->y = (r + xr) * (1 << octv);
    out->octave = octv + (layer << 8) + (cv_round((xi + 0.5f) * 255) << 16);
    out->size = sigma * sift_expf(((layer + xi) / nL) * 0.69314718f) * (1 << octv) * 2;
    out->response = fabsf(contr);
    return true;
}

// tiles of 64 x 16 pixels of the bordered interior, nL layers per octave, all octaves in one launch
constexpr int kExtRows = 4;                                   // rows per thread (a wave: 64 columns x 4 rows; a workgroup: 64 x 16)
__device__ __host__ inline int extrema_tiles(const Geom& g, int o) {
    const int iw = g.w(o) - 2 * kImgBorder, ih = g.h(o) - 2 * kImgBorder;
    return iw > 0 && ih > 0 ? ((iw + 63) / 64) * ((ih + 4 * kExtRows - 1) / (4 * kExtRows)) * g.nL : 0;
}

// A pixel is an extremum when it is >= (<=) all 26 neighbours, i.e. >= their MAXIMUM (<= their minimum): comparisons are
// exact, so max3 / min3 trees give the sequential test's answer with a quarter of its instructions.  A thread takes four
// consecutive rows of one column: the 3-wide row maxima / minima of the six rows it touches (three planes) are formed once
// and shared by its four pixels — 13.5 loads and ~20 vector instructions per pixel instead of 27 and ~100 (53 -> 32 us
// per 968 x 648 frame).  Tiles whose pixels are all below the threshold read their centre values only.
__global__ __launch_bounds__(256) void extrema_kernel(Geom geo, int threshold, int4* __restrict__ raw, int raw_cap, int* __restrict__ counters) {
    int o = 0, t = blockIdx.x;
    for (;; ++o) { const int k = extrema_tiles(geo, o); if (t < k) break; t -= k; }
    const int w = geo.w(o), h = geo.h(o);
    const int tx = (w - 2 * kImgBorder + 63) / 64, ty = (h - 2 * kImgBorder + 4 * kExtRows - 1) / (4 * kExtRows);
    const int layer = 1 + t / (tx * ty);
    t -= (layer - 1) * tx * ty;
    const int lane = threadIdx.x & 63;
    const int c = kImgBorder + (t % tx) * 64 + lane, r0 = kImgBorder + (t / tx) * (4 * kExtRows) + (threadIdx.x >> 6) * kExtRows;
    const size_t plane = geo.plane(o);
    const bool live = c < w - kImgBorder && r0 < h - kImgBorder;      // (then rows r0 - 1 .. r0 + 4 and columns c +- 1 are inside the plane)
    const float* img = geo.D(o, 0) + plane * layer + (size_t)r0 * w + c;
    float val[kExtRows];
    bool cand[kExtRows], ext[kExtRows];
    bool some = false;
#pragma unroll
    for (int k = 0; k < kExtRows; ++k) {
        val[k] = live ? img[k * w] : 0.f;
        cand[k] = live && r0 + k < h - kImgBorder && fabsf(val[k]) > (float)threshold;
        ext[k] = false;
        some |= cand[k];
    }
    if (__any(some)) {
        if (live) {
            float mx[3][kExtRows + 2], mn[3][kExtRows + 2], sx[kExtRows + 2], sn[kExtRows + 2];   // row maxima / minima; own plane: without the centre
#pragma unroll
            for (int p = 0; p < 3; ++p)
#pragma unroll
                for (int q = 0; q < kExtRows + 2; ++q) {
                    const float* row = img + (long)(p - 1) * (long)plane + (q - 1) * w;
                    const float a = row[-1], b = row[0], d = row[1];
                    mx[p][q] = fmaxf(fmaxf(a, b), d);
                    mn[p][q] = fminf(fminf(a, b), d);
                    if (p == 1) { sx[q] = fmaxf(a, d); sn[q] = fminf(a, d); }
                }
#pragma unroll
            for (int k = 0; k < kExtRows; ++k) {
                const float hi = fmaxf(fmaxf(fmaxf(fmaxf(mx[0][k], mx[0][k + 1]), mx[0][k + 2]), fmaxf(fmaxf(mx[2][k], mx[2][k + 1]), mx[2][k + 2])),
                                       fmaxf(fmaxf(mx[1][k], sx[k + 1]), mx[1][k + 2]));
                const float lo = fminf(fminf(fminf(fminf(mn[0][k], mn[0][k + 1]), mn[0][k + 2]), fminf(fminf(mn[2][k], mn[2][k + 1]), mn[2][k + 2])),
                                       fminf(fminf(mn[1][k], sn[k + 1]), mn[1][k + 2]));
                ext[k] = cand[k] && ((val[k] > 0 && val[k] >= hi) || (val[k] < 0 && val[k] <= lo));
            }
        }
        // The raw list is kRawSegs independent segments, each with its own counter on its own cache line: tens of
        // thousands of atomics on ONE address serialise in L2 (they, not the loads, bounded the first version: 250 us).
        unsigned long long bal[kExtRows];
        int total = 0, below = 0;
#pragma unroll
        for (int k = 0; k < kExtRows; ++k) {
            bal[k] = __ballot(ext[k]);
            total += __popcll(bal[k]);
            below += __popcll(bal[k] & ((1ull << lane) - 1ull));
        }
        if (total) {                                                                // wave-uniform: one atomic per wave
            const int seg = blockIdx.x % kRawSegs, seg_cap = raw_cap / kRawSegs;
            int base = 0;
            if (lane == 0) base = atomicAdd(&counters[kCntRawSeg + seg * kSegStride], total);
            int at = __shfl(base, 0) + below;
#pragma unroll
            for (int k = 0; k < kExtRows; ++k)
                if (ext[k]) { if (at < seg_cap) raw[(size_t)seg * seg_cap + at] = make_int4(o, layer, r0 + k, c); ++at; }
        }
    }
}

constexpr int kDescGrid = 8192;       // workgroups of descriptor_kernel: one keypoint per wave up to 32 768 keypoints, strided beyond
constexpr int kOriGrid = 2048;        // workgroups of orientation_kernel: 8 192 waves = every wave slot of the chip once
constexpr int kRawPerKeypoint = 8;   // capacity of the raw-extrema list (before contrast / edge rejection) per keypoint slot

// sub-pixel refinement, contrast and edge tests of the raw extrema: one lane each
__global__ __launch_bounds__(256) void refine_kernel(Geom geo, const int4* __restrict__ raw, int raw_cap, float contrastThreshold, float edgeThreshold,
                                                     float sigma, Cand* __restrict__ cand, int* __restrict__ counters, int cap) {
    const int seg_cap = raw_cap / kRawSegs;
    const int id = blockIdx.x * 256 + threadIdx.x, lane = threadIdx.x & 63;
    // thread -> (segment, slot) INTERLEAVED (id = slot * kRawSegs + segment): the filled slots of all 64 segments are then the
    // first ids of the grid — the workgroups that have work start at once and the (many) empty ones behind them only return;
    // segment-major, each segment's two busy workgroups sat in front of its 62 empty ones and the last segment's started
    // after the whole grid had been dispatched
    const int seg = id % kRawSegs, at = id / kRawSegs;
    bool ok = false;
    Cand k;
    if (at < seg_cap) {
        const int cnt = counters[kCntRawSeg + seg * kSegStride];
        if (at == 0) { atomicAdd(&counters[3], cnt); if (cnt > seg_cap) atomicMax(&counters[4], cnt); }   // totals for the caller
        if (at < min(cnt, seg_cap)) {
            const int4 e = raw[(size_t)seg * seg_cap + at];
            ok = adjust_local_extrema(geo.D(e.x, 0), geo.w(e.x), geo.h(e.x), geo.plane(e.x), e.x, e.y, e.z, e.w, geo.nL, contrastThreshold, edgeThreshold,
                                      sigma, &k);
        }
    }
    const unsigned long long bal = __ballot(ok);
    if (bal) {
        int base = 0;
        if (lane == 0) base = atomicAdd(&counters[0], __popcll(bal));
        base = __shfl(base, 0);
        const int slot = base + __popcll(bal & ((1ull << lane) - 1ull));
        if (ok && slot < cap) cand[slot] = k;
    }
}

// Gradient magnitude and orientation (degrees, OpenCV's fastAtan2) of every interior pixel of Gaussian layers
// 1..nL, all octaves in one launch: the orientation and descriptor kernels visit each pixel many times.
// tiles of 64 x 4 interior pixels; workgroup -> (octave, layer, tile) by a scalar scan, thread -> pixel by two adds.  (The flat
// pixel index per thread it replaced — a 64-bit division and three table walks per pixel — took the same 43 us: the kernel
// is bound by its 80 MB of float2 stores + 40 MB of loads, 2.8 TB/s, not by its arithmetic.)
__device__ __host__ inline int grad_tiles(const Geom& g, int o) {
    const int iw = g.w(o) - 2, ih = g.h(o) - 2;
    return iw > 0 && ih > 0 ? ((iw + 63) / 64) * ((ih + 3) / 4) * g.nL : 0;
}
__global__ __launch_bounds__(256) void gradient_kernel(Geom geo) {
    int o = 0, t = blockIdx.x;
    for (;; ++o) { const int k = grad_tiles(geo, o); if (t < k) break; t -= k; }
    const int w = geo.w(o), h = geo.h(o);
    const int tx = (w - 2 + 63) / 64, ty = (h - 2 + 3) / 4;
    const int layer = 1 + t / (tx * ty);
    t -= (layer - 1) * tx * ty;
    const int x = 1 + (t % tx) * 64 + (threadIdx.x & 63), y = 1 + (t / tx) * 4 + (threadIdx.x >> 6);
    if (x >= w - 1 || y >= h - 1) return;
    const float* img = geo.G(o, layer) + (size_t)y * w + x;
    const float dx = img[1] - img[-1];
    const float dy = img[-w] - img[w];
    geo.grad[geo.moff(o, layer) + (size_t)y * w + x] = make_float2(sqrtf(dx * dx + dy * dy), fast_atan2_deg(dy, dx));
}

// ------------------------------------------------------------------------------------------------ orientation
constexpr int kOriChunk = 1024;
// One wave per refined candidate.  Samples of the (2r+1)^2 window are evaluated 64 at a time into LDS (bin, weight *
// magnitude); lane b < 36 then owns histogram bin b and adds its samples in raster order (float sums are order
// sensitive; this is the order of the sequential algorithm).  Keypoints (one per histogram peak >= 0.8 max) are
// appended unordered; the sort below fixes the order.
__global__ __launch_bounds__(256) void orientation_kernel(Geom geo, const Cand* __restrict__ cand, int* __restrict__ counters, int cap,
                                                          float* __restrict__ kp_raw) {
    __shared__ __attribute__((aligned(16))) float sval[4][kOriChunk];
    __shared__ __attribute__((aligned(16))) unsigned char sbin[4][kOriChunk];      // 255 = no sample
    __shared__ float shist[4][kOriBins + 4];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ncand = min(counters[0], cap);
    // The grid is a FIXED number of workgroups and a wave strides over the candidates: sized for `cap` (the host does not
    // know the count), 32 768 workgroups of which nine in ten only return took 56 us to dispatch — as long as the work.
    for (int id = blockIdx.x * 4 + wave; id < ncand; id += gridDim.x * 4) {
    const Cand k = cand[id];
    const int w = geo.w(k.o), h = geo.h(k.o);
    const float2* grad = geo.grad + geo.moff(k.o, k.layer);
    const float scl_octv = k.size * 0.5f / (1 << k.o);
    const int radius = cv_round(kOriRadius * scl_octv);
    const float sigma = kOriSigFctr * scl_octv;
    const float expf_scale = -1.f / (2.f * sigma * sigma);
    const int side = 2 * radius + 1, total = side * side;
    const int n = kOriBins;
    float acc = 0.f;   // lane b: temphist[b]
    for (int base = 0; base < total; base += kOriChunk) {
        const int m = min(kOriChunk, total - base), m16 = (m + 15) & ~15;
        for (int e = lane; e < m16; e += 64) {
            const int idx = base + e;
            const int i = idx / side - radius, j = idx % side - radius;
            const int y = k.r + i, x = k.c + j;
            int bin = 255; float v = 0.f;
            if (e < m && !(y <= 0 || y >= h - 1 || x <= 0 || x >= w - 1)) {
                const float W = sift_expf((i * i + j * j) * expf_scale);
                const float2 mo = grad[(size_t)y * w + x];
                const float ori = mo.y, mag = mo.x;
                bin = cv_round((n / 360.f) * ori);
                if (bin >= n) bin -= n;
                if (bin < 0) bin += n;
                v = W * mag;
            }
            sbin[wave][e] = (unsigned char)bin; sval[wave][e] = v;
        }
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): the wave's own LDS writes are visible
        // lane b adds the samples of bin b in raster order, 16 samples per LDS round trip; the others add +0, which leaves
        // a non-negative float sum unchanged (every addend W * mag is >= +0)
        for (int e = 0; e < m16; e += 16) {
            const uint4 b16 = *reinterpret_cast<const uint4*>(&sbin[wave][e]);
            const float4* vp = reinterpret_cast<const float4*>(&sval[wave][e]);
            const float4 v0 = vp[0], v1 = vp[1], v2 = vp[2], v3 = vp[3];
            const unsigned bw[4] = {b16.x, b16.y, b16.z, b16.w};
            const float vv[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
#pragma unroll
            for (int t = 0; t < 16; ++t) acc += ((bw[t >> 2] >> (8 * (t & 3))) & 255u) == (unsigned)lane ? vv[t] : 0.f;
        }
        __builtin_amdgcn_wave_barrier();
    }
    float* th = &shist[wave][2];
    if (lane < n) th[lane] = acc;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    if (lane == 0) { th[-1] = th[n - 1]; th[-2] = th[n - 2]; th[n] = th[0]; th[n + 1] = th[1]; }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    float hv = -FLT_MAX;
    if (lane < n) hv = (th[lane - 2] + th[lane + 2]) * (1.f / 16.f) + (th[lane - 1] + th[lane + 1]) * (4.f / 16.f) + th[lane] * (6.f / 16.f);
    float omax = hv;
#pragma unroll
    for (int s = 32; s >= 1; s >>= 1) omax = fmaxf(omax, __shfl_xor(omax, s));
    const float mag_thr = omax * kOriPeakRatio;
    const int l = lane > 0 ? lane - 1 : n - 1, r2 = lane < n - 1 ? lane + 1 : 0;
    const float hl = __shfl(hv, l), hr = __shfl(hv, r2);
    const bool peak = lane < n && hv > hl && hv > hr && hv >= mag_thr;
    const unsigned long long bal = __ballot(peak);
    int slot0 = 0;
    const int seg = blockIdx.x % kRawSegs, seg_cap = cap / kRawSegs;          // segmented list, see extrema_kernel
    if (lane == 0 && bal) {
        slot0 = atomicAdd(&counters[kCntKpSeg + seg * kSegStride], __popcll(bal));
        atomicAdd(&counters[kCntHist + x_bucket(k.x, geo.W0)], __popcll(bal));
    }
    slot0 = __shfl(slot0, 0);
    if (peak) {
        float bin = lane + 0.5f * (hl - hr) / (hl - 2 * hv + hr);
        bin = bin < 0 ? n + bin : bin >= n ? bin - n : bin;
        float angle = 360.f - (360.f / n) * bin;
        if (fabsf(angle - 360.f) < FLT_EPSILON) angle = 0.f;
        const int slot = slot0 + __popcll(bal & ((1ull << lane) - 1ull));
        if (slot < seg_cap) {
            float* q = kp_raw + ((size_t)seg * seg_cap + slot) * 8;
            q[0] = k.x; q[1] = k.y; q[2] = k.size; q[3] = angle; q[4] = k.response; q[5] = __int_as_float(k.octave); q[6] = __int_as_float(-1); q[7] = 0.f;
        }
    }
    __builtin_amdgcn_wave_barrier();                              // (the wave's LDS rows are reused by its next candidate)
    }
}

// ------------------------------------------------------------------------------------------------ ordering
// KeyPointsFilter::removeDuplicatedSorted order: x, y ascending, size descending, angle ascending, response, octave
// descending; keypoints equal in every field are interchangeable, the index breaks the tie.
struct Key { float x, y, size, angle, response; int octave; };
__device__ inline bool key_before(const Key& a, int ia, const Key& b, int ib) {
    if (a.x != b.x) return a.x < b.x;
    if (a.y != b.y) return a.y < b.y;
    if (a.size != b.size) return a.size > b.size;
    if (a.angle != b.angle) return a.angle < b.angle;
    if (a.response != b.response) return a.response > b.response;
    if (a.octave != b.octave) return a.octave > b.octave;
    return ia < ib;
}

// Ordering in two levels.  Keypoints are first binned by x into kXBuckets ranges of the (doubled) image width — the
// histogram is filled by the orientation kernel as it appends keypoints — and scattered bucket by bucket
// (unordered inside a bucket); then each keypoint counts the members of its own bucket that are ordered before it
// (full comparison cascade; ~n / 256 candidates instead of n).  Buckets are monotone in x, the primary key, so
// bucket start + rank inside the bucket is the global rank.

__device__ inline void bucket_starts(const int* __restrict__ counters, int* start /* LDS [kXBuckets + 1] */) {   // 256 threads
    __shared__ int wtot[4];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int v = counters[kCntHist + t];
    int inc = v;
#pragma unroll
    for (int sft = 1; sft < 64; sft <<= 1) { const int o = __shfl_up(inc, sft); if (lane >= sft) inc += o; }
    if (lane == 63) wtot[wave] = inc;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < wave; ++k) base += wtot[k];
    start[t] = base + inc - v;
    if (t == 255) start[256] = base + inc;
    __syncthreads();
}

__global__ __launch_bounds__(256) void bucket_scatter_kernel(const float* __restrict__ kp_raw, int* __restrict__ counters, int cap, int W0,
                                                             float* __restrict__ kp_bucketed) {
    __shared__ int start[kXBuckets + 1];
    bucket_starts(counters, start);
    const int seg_cap = cap / kRawSegs;
    const int id = blockIdx.x * 256 + threadIdx.x;
    const int seg = id % kRawSegs, at = id / kRawSegs;           // interleaved, as in refine_kernel: the filled slots are the grid's first ids
    if (at >= seg_cap) return;
    const int cnt = counters[kCntKpSeg + seg * kSegStride];
    if (at == 0) { atomicAdd(&counters[1], min(cnt, seg_cap)); if (cnt > seg_cap) atomicMax(&counters[5], cnt); }   // dense count for the next kernels
    if (at >= min(cnt, seg_cap)) return;
    const float4* src = reinterpret_cast<const float4*>(kp_raw + ((size_t)seg * seg_cap + at) * 8);
    const float4 a = src[0], b = src[1];
    const int bk = x_bucket(a.x, W0);
    const int pos = start[bk] + atomicAdd(&counters[kCntCursor + bk], 1);
    if (pos < cap) {
        float4* d = reinterpret_cast<float4*>(kp_bucketed + (size_t)pos * 8);
        d[0] = a; d[1] = b;
    }
}

__global__ __launch_bounds__(256) void bucket_rank_kernel(const float* __restrict__ kp_bucketed, const int* __restrict__ counters, int cap, int W0,
                                                          float* __restrict__ kp_sorted) {
    __shared__ int start[kXBuckets + 1];
    __shared__ float4 tile[1024];
    const int n = min(counters[1], cap);
    if ((int)(blockIdx.x * 256) >= n) return;
    bucket_starts(counters, start);
    const int i = blockIdx.x * 256 + threadIdx.x;
    const bool valid = i < n;
    float4 a = {0, 0, 0, 0}, b = {0, 0, 0, 0};
    if (valid) { a = reinterpret_cast<const float4*>(kp_bucketed + (size_t)i * 8)[0]; b = reinterpret_cast<const float4*>(kp_bucketed + (size_t)i * 8)[1]; }
    const Key me = {a.x, a.y, a.z, a.w, b.x, __float_as_int(b.y)};
    const int bk = x_bucket(a.x, W0);
    const int lo = valid ? start[bk] : 0, hi = valid ? min(start[bk + 1], n) : 0;
    // The workgroup's 256 keypoints are consecutive in bucket order, so the buckets they must be ranked within form ONE
    // range of the list: it is staged through LDS in coalesced tiles (a thread walking its bucket in global memory paid a
    // round trip per element: 32 us for 15 000 keypoints).
    const int i_last = min(n, (int)(blockIdx.x * 256) + 256) - 1;
    const int glo = start[x_bucket(kp_bucketed[(size_t)(blockIdx.x * 256) * 8], W0)];
    const int ghi = min(start[x_bucket(kp_bucketed[(size_t)i_last * 8], W0) + 1], n);
    int r = 0;
    for (int t0 = glo; t0 < ghi; t0 += 1024) {
        __syncthreads();
        for (int k = threadIdx.x; k < min(1024, ghi - t0); k += 256) tile[k] = reinterpret_cast<const float4*>(kp_bucketed + (size_t)(t0 + k) * 8)[0];
        __syncthreads();
        const int jl = max(lo, t0), jh = min(hi, t0 + 1024);
        for (int j = jl; j < jh; ++j) {
            const float4 c = tile[j - t0];
            if (c.x != me.x || c.y != me.y) r += (c.x < me.x || (c.x == me.x && c.y < me.y)) ? 1 : 0;
            else {
                const float4 e = reinterpret_cast<const float4*>(kp_bucketed + (size_t)j * 8)[1];
                const Key other = {c.x, c.y, c.z, c.w, e.x, __float_as_int(e.y)};
                r += key_before(other, j, me, i) ? 1 : 0;
            }
        }
    }
    const int pos = lo + r;
    if (valid && pos < cap) {
        float4* d = reinterpret_cast<float4*>(kp_sorted + (size_t)pos * 8);
        d[0] = a; d[1] = b;
    }
}

// single workgroup: drop keypoints equal to their predecessor in (x, y, size, angle), keep the order, undo the 2x
// base (firstOctave = -1): pt and size halve, the octave byte decrements.
__global__ __launch_bounds__(1024) void dedupe_kernel(const float* __restrict__ kp_sorted, int* __restrict__ counters, int cap, float* __restrict__ kp_out,
                                                      int* __restrict__ count_out, int nL, int* __restrict__ perm) {
    __shared__ int wsum[2][16];
    __shared__ int bcnt[64], bpos[64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // A list that overflowed upstream leaves slots of kp_sorted unwritten (the x-range histogram counts keypoints that
    // were never stored): what the later kernels would read there is whatever the workspace held before — a "keypoint"
    // with a 1e30 window keeps the descriptor kernel busy for minutes.  The caller raises on the counts reported below
    // anyway, so nothing downstream runs on such a frame.
    const bool overflow = counters[4] != 0 || counters[5] != 0 || counters[0] > cap;
    const int n = overflow ? 0 : min(counters[1], cap);
    if (threadIdx.x < 64) bcnt[threadIdx.x] = 0;
    // processing order of the descriptor kernel: keypoints bucketed by window size (layer and sub-layer offset are
    // packed in the octave field), largest first — the kernel's last waves are then its shortest (in list order it takes
    // 230 instead of 180 us for 15 000 keypoints; the ordering costs 14 of this kernel's 36 us)
    auto bucket = [nL](int packed) {
        const int key = (((packed >> 8) & 255) - 1) * 256 + ((packed >> 16) & 255);
        return 63 - min(63, max(0, key * 64 / (nL * 256)));
    };
    // One workgroup is latency-bound: every trip of 1 024 keypoints is a round trip to memory and a barrier.  The rows of
    // FOUR trips are requested together (coalesced: trip k of a thread is row i0 + 1024 k + thread), the running total lives
    // in registers (every thread adds the same sixteen wave counts) and the wave counts alternate between two LDS rows:
    // one round trip per four trips, ONE barrier per trip.  (Four CONSECUTIVE rows per thread instead: 57 us — strided
    // rows coalesce worse.)
    constexpr int kAhead = 4;
    int base = 0, trip = 0;
    for (int i0 = 0; i0 < n; i0 += kAhead * 1024) {
        float4 va[kAhead], vb[kAhead], vp[kAhead];
#pragma unroll
        for (int k = 0; k < kAhead; ++k) {
            const int i = i0 + k * 1024 + threadIdx.x;
            va[k] = vb[k] = vp[k] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < n) {
                va[k] = reinterpret_cast<const float4*>(kp_sorted + (size_t)i * 8)[0];
                vb[k] = reinterpret_cast<const float4*>(kp_sorted + (size_t)i * 8)[1];
                if (i > 0) vp[k] = reinterpret_cast<const float4*>(kp_sorted + (size_t)(i - 1) * 8)[0];
            }
        }
#pragma unroll
        for (int k = 0; k < kAhead; ++k, ++trip) {
            if (i0 + k * 1024 >= n) break;                      // (uniform)
            const int i = i0 + k * 1024 + threadIdx.x;
            const float4 a = va[k], b = vb[k], p = vp[k];
            const bool keep = i < n && (i == 0 || !(p.x == a.x && p.y == a.y && p.z == a.z && p.w == a.w));
            const unsigned long long bal = __ballot(keep);
            if (lane == 0) wsum[trip & 1][wave] = __popcll(bal);
            __syncthreads();
            int woff = 0, total = 0;
#pragma unroll
            for (int w = 0; w < 16; ++w) { const int c = wsum[trip & 1][w]; if (w < wave) woff += c; total += c; }
            if (keep) {
                const int pos = base + woff + __popcll(bal & ((1ull << lane) - 1ull));
                int oct = __float_as_int(b.y);
                oct = (oct & ~255) | ((oct - 1) & 255);
                float4* d = reinterpret_cast<float4*>(kp_out + (size_t)pos * 8);
                d[0] = make_float4(a.x * 0.5f, a.y * 0.5f, a.z * 0.5f, a.w);
                d[1] = make_float4(b.x, __int_as_float(oct), b.z, 0.f);
                atomicAdd(&bcnt[bucket(oct)], 1);
            }
            base += total;
        }
    }
    if (threadIdx.x == 0) { counters[2] = base; count_out[0] = base; count_out[1] = counters[5] ? 0x7fffffff : counters[1]; count_out[2] = counters[0]; count_out[3] = counters[4] ? 0x7fffffff : counters[3]; }
    const int nout = min(base, cap);
    __syncthreads();                                           // (also: this workgroup's kp_out stores are visible to its own loads below)
    if (threadIdx.x == 0) { int acc = 0; for (int k = 0; k < 64; ++k) { bpos[k] = acc; acc += bcnt[k]; } }
    __syncthreads();
    for (int i = threadIdx.x; i < nout; i += 4 * 1024) {      // four independent loads in flight per thread
        int oc[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) oc[k] = i + k * 1024 < nout ? __float_as_int(kp_out[(size_t)(i + k * 1024) * 8 + 5]) : 0;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + k * 1024 < nout) perm[atomicAdd(&bpos[bucket(oc[k])], 1)] = i + k * 1024;
    }
}

// ------------------------------------------------------------------------------------------------ descriptors
// One wave per keypoint: FOUR lanes per interior cell of the 6 x 6 x 10 trilinear histogram (the border cells never
// reach the 4 x 4 x 8 descriptor, so they are not accumulated).  float32 sums are order sensitive: every bin must
// receive its contributions in raster order of the rotated window.  Instead of routing samples to bins, each cell
// walks its own footprint — a rotated square of side 2 hist_width centred (in pixel offsets from the keypoint) at
// T = hw^2 R^T (ci-2.5, ri-2.5).  The sixteen footprints are translates of one square, so the whole wave shares ONE
// walk (wave-uniform control flow): rows u and spans [v0(u), v1(u)] of the centred square dilated by the rounding of T
// (+ floor/ceil slack), visited at (i, j) = (round(T) + (u, v)); the exact per-sample test decides membership, the
// template only has to cover it.  The four lanes of a cell evaluate four consecutive samples of a row at once
// (rotation, Gaussian weight, one float2 gradient gather, trilinear split) and then add them to the cell's bins one
// after the other, in order: the evaluation (~120 instructions) runs four wide, only the two additions per sample are
// sequential.  Every bin sees exactly the sequential algorithm's addends in its order; a sample is evaluated by the
// <= 4 cells it touches.  Normalisation, the 0.2 clip and the 512 / u8 quantisation follow; output is float32
// holding integers.  (Four keypoints per wave with one lane per cell does the same work per wave but leaves a wave
// alone on its SIMD when a frame has few keypoints: 1 174 keypoints took 315 us, as long as 15 000.)
constexpr int kDescBinsUsed = kDescBins + 1;   // o0 in 0..7 writes bins o0 and o0+1: bin 9 of the 10 stays zero
constexpr int kDescLdsFloats = 4 * (16 * kDescBinsUsed + 128);   // per wave: the 16 cells' bins, then the keypoint's 128 raw values
__device__ __forceinline__ void descriptor_cells(const Geom& geo, const float* __restrict__ kp, int nkp, const int* __restrict__ perm,
                                                 float* __restrict__ desc, float* sbuf, int wg /* the workgroup's number in the list: keypoints 4 wg .. 4 wg + 3 */) {
    float (*shist)[16 * kDescBinsUsed] = reinterpret_cast<float (*)[16 * kDescBinsUsed]>(sbuf);
    float (*sdst)[128] = reinterpret_cast<float (*)[128]>(sbuf + 4 * 16 * kDescBinsUsed);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, cell = lane >> 2, sub = lane & 3;
    if (wg * 4 >= nkp) return;                     // whole workgroup
    const int slot = wg * 4 + wave;
    const bool have = slot < nkp;                               // (the last workgroup's idle waves still meet the barrier below)
    const int id = perm[have ? slot : nkp - 1];                 // largest windows first
    const int d = kDescWidth, n = kDescBins;
    const int ri = 1 + (cell >> 2), ci = 1 + (cell & 3);        // my cell in the (d+2) x (d+2) grid
    float* mine = shist[wave] + cell * kDescBinsUsed;
    if (sub == 0) {
#pragma unroll
        for (int k = 0; k < kDescBinsUsed; ++k) mine[k] = 0.f;
    }
    const float* q = kp + (size_t)id * 8;
    const int packed = __float_as_int(q[5]);
    int octave = packed & 255; const int layer = (packed >> 8) & 255;
    octave = octave < 128 ? octave : (-128 | octave);
    const float scale = octave >= 0 ? 1.f / (1 << octave) : (float)(1 << -octave);
    const float size = q[2] * scale;
    float ori = 360.f - q[3];
    if (fabsf(ori - 360.f) < FLT_EPSILON) ori = 0.f;
    const int o = octave + 1;
    const int w = geo.w(o), h = geo.h(o);
    const float2* grad = geo.grad + geo.moff(o, layer);
    const int px = cv_round(q[0] * scale), py = cv_round(q[1] * scale);
    const float scl = size * 0.5f;
    float cos_t, sin_t;
    sift_sincos(ori * (float)(3.14159265358979323846 / 180), &sin_t, &cos_t);
    const float hist_width = kDescSclFctr * scl;
    int radius = cv_round(hist_width * 1.4142135623730951f * (d + 1) * 0.5f);
    radius = min(radius, (int)sqrt((double)w * w + (double)h * h));
    cos_t /= hist_width; sin_t /= hist_width;
    const float bins_per_rad = n / 360.f, exp_scale = -1.f / (d * d * 0.5f);
    // my cell's integer translation; the shared template
    const float C = cos_t * hist_width * hist_width, S = sin_t * hist_width * hist_width;
    const float ccen = ci - 2.5f, rcen = ri - 2.5f;
    const int tj = (int)rintf(C * ccen + S * rcen), ti = (int)rintf(C * rcen - S * ccen);
    const float ext = fabsf(cos_t) + fabsf(sin_t);
    const float B = 1.f + 0.5f * ext + 1e-3f;
    // columns j that can enter at all: |j| <= radius and 1 <= px + j <= w - 2 (none: the histogram stays zero)
    const int jlo = max(-radius, 1 - px), jhi = min(radius, w - 2 - px);
    const unsigned jspan = (unsigned)(jhi - jlo);
    const int U = have && jhi >= jlo ? (int)ceilf(B * hist_width * hist_width * ext) : -1;
    const bool use_c = fabsf(cos_t) > 1e-4f, use_s = fabsf(sin_t) > 1e-4f;
    const float inv_c = use_c ? 1.f / cos_t : 0.f, inv_s = use_s ? 1.f / sin_t : 0.f;
    // row u of the template spans [max(m1 u - e1, m2 u - e2), min(m1 u + e1, m2 u + e2)]: |v cos_t - u sin_t| <= B and
    // |v sin_t + u cos_t| <= B solved for v (a bound whose coefficient vanishes is replaced by the window, 4 radius)
    const float m1 = sin_t * inv_c, e1 = use_c ? B * fabsf(inv_c) : 4.f * radius;
    const float m2 = -cos_t * inv_s, e2 = use_s ? B * fabsf(inv_s) : 4.f * radius;
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    for (int u = -U; u <= U; ++u) {                               // wave-uniform
        const float fu = (float)u;
        const float lo = fmaxf(m1 * fu - e1, m2 * fu - e2), hi = fminf(m1 * fu + e1, m2 * fu + e2);
        const int v0 = (int)floorf(lo), v1 = lo <= hi ? (int)ceilf(hi) : v0 - 1;
        const int i = ti + u;
        const int r = py + i;
        const bool row_ok = i >= -radius && i <= radius && (unsigned)(r - 1) < (unsigned)(h - 2);
        for (int vb = v0; vb <= v1; vb += 4) {                    // four consecutive samples of the row, one per lane of the cell
            const int j = tj + vb + sub;
            const float c_rot = j * cos_t - i * sin_t, r_rot = j * sin_t + i * cos_t;
            float rbin = r_rot + d / 2 - 0.5f, cbin = c_rot + d / 2 - 0.5f;
            const int r0 = cv_floor(rbin), c0 = cv_floor(cbin);
            const int dr = ri - (r0 + 1), dc = ci - (c0 + 1);
            // rbin < d and cbin < d follow from dr, dc <= 1 (r0 <= ri - 1 <= 3)
            const bool inside = row_ok & (rbin > -1) & (cbin > -1) & ((unsigned)(j - jlo) <= jspan) & ((unsigned)(dr | dc) <= 1u);
            int o0 = 0;
            float vo0 = 0.f, vo1 = 0.f;
            if (inside) {
                const float W = sift_expf_unclamped((c_rot * c_rot + r_rot * r_rot) * exp_scale);   // argument in (-1.6, 0]
                const float2 mo = grad[(unsigned)(r * w + px + j)];
                float obin = (mo.y - ori) * bins_per_rad;
                const float mag = mo.x * W;
                o0 = cv_floor(obin);
                rbin -= r0; cbin -= c0; obin -= o0;
                o0 &= n - 1;                                   // obin in (-8, 8]: the same as { if (o0 < 0) o0 += n; if (o0 >= n) o0 -= n; }
                const float vr1 = mag * rbin;
                const float vr = dr ? vr1 : mag - vr1;
                const float vrc1 = vr * cbin;
                const float vrc = dc ? vrc1 : vr - vrc1;
                vo1 = vrc * obin; vo0 = vrc - vo1;
            }
            if (__any(inside)) {
                // the four samples enter the cell's bins in row order; LDS operations of one wave execute in program order
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (sub == k && inside) { mine[o0] += vo0; mine[o0 + 1] += vo1; }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_s_waitcnt(0xc07f);
    // circular orientation: bin n folds onto bin 0 (bin n+1, which would fold onto 1, is never written)
    float* dst = sdst[wave];
    if (sub == 0) {
        mine[0] += mine[n];
#pragma unroll
        for (int k = 0; k < kDescBins; ++k) dst[cell * n + k] = mine[k];
    }
    // normalisation: ONE wave does the four keypoints of the workgroup, sixteen lanes each (the two sequential sums cost a
    // wave the same whether it carries one keypoint or four)
    __syncthreads();
    if (wave != 0) return;
    const int kq = lane >> 4, l16 = lane & 15;
    const float* dk = sdst[kq];
    const int len = d * d * n;
    float nrm2 = 0.f;
    for (int k = 0; k < len; ++k) nrm2 += dk[k] * dk[k];       // sequential: the order is part of the result
    const float thr = sqrtf(nrm2) * kDescMagThr;
    nrm2 = 0.f;
    for (int k = 0; k < len; ++k) { const float v = fminf(dk[k], thr); nrm2 += v * v; }
    const float mul = kIntDescrFctr / fmaxf(sqrtf(nrm2), FLT_EPSILON);
    if (wg * 4 + kq < nkp) {
        float out[kDescBins];
#pragma unroll
        for (int k = 0; k < kDescBins; ++k) { const int iv = cv_round(fminf(dk[l16 * kDescBins + k], thr) * mul); out[k] = (float)(iv < 0 ? 0 : iv > 255 ? 255 : iv); }
        float4* o4 = reinterpret_cast<float4*>(desc + (size_t)perm[wg * 4 + kq] * 128 + l16 * kDescBins);
        o4[0] = make_float4(out[0], out[1], out[2], out[3]);
        o4[1] = make_float4(out[4], out[5], out[6], out[7]);
    }
}

// The grid is at most kDescGrid workgroups (the host does not know the count and `cap` is generous: dispatching 32 768 workgroups
// that only return takes ~50 us, a floor under every small frame): one keypoint per wave for the first 4 kDescGrid keypoints;
// a second, small launch (kRest) strides over what is beyond them and returns at once on ordinary frames.
template <bool kRest>
__global__ __launch_bounds__(256) void descriptor_kernel(Geom geo, const float* __restrict__ kp, const int* __restrict__ counters, int cap,
                                                         const int* __restrict__ perm, float* __restrict__ desc) {
    __shared__ float sbuf[kDescLdsFloats];
    const int nkp = min(counters[2], cap);
    if (!kRest) descriptor_cells(geo, kp, nkp, perm, desc, sbuf, blockIdx.x);
    else
        for (int wg = kDescGrid + blockIdx.x; wg * 4 < nkp; wg += gridDim.x) {
            descriptor_cells(geo, kp, nkp, perm, desc, sbuf, wg);
            __syncthreads();                                    // (the workgroup's LDS is reused by its next four keypoints)
        }
}

int gauss_taps(double sigma, Taps* t) {   // getGaussianKernel(cvRound(sigma*8+1)|1, sigma, CV_32F)
    const int n = (int)lrint(sigma * 4 * 2 + 1) | 1;
    if (n > kMaxTaps) return -1;
    const double s2 = -0.5 / (sigma * sigma);
    double sum = 0;
    for (int i = 0; i < n; ++i) { const double x = i - (n - 1) * 0.5; t->k[i] = (float)std::exp(s2 * x * x); sum += t->k[i]; }
    sum = 1. / sum;
    for (int i = 0; i < n; ++i) t->k[i] = (float)(t->k[i] * sum);
    for (int i = n; i <= kMaxTaps; ++i) t->k[i] = 0.f;
    t->n = n;
    return n;
}

int num_octaves(int64_t w, int64_t h) {
    const double m = (double)std::min(2 * w, 2 * h);
    return (int)lrint(std::log(m) / std::log(2.) - 2) + 1;
}

struct Layout { size_t g_floats, d_floats, m_floats, up_floats; int nOct; };
Layout layout(int64_t w, int64_t h, int nL) {
    Layout L{};
    L.nOct = num_octaves(w, h);
    for (int o = 0; o < L.nOct; ++o) {
        const size_t p = (size_t)((2 * w) >> o) * (size_t)((2 * h) >> o);
        L.g_floats += p * (nL + 3);
        L.d_floats += p * (nL + 2);
        L.m_floats += p * nL;
    }
    L.up_floats = (size_t)4 * w * h;
    return L;
}

}  // namespace

extern "C" int sfm_bgr2gray_u8(const uint8_t* bgr, int64_t w, int64_t h, int64_t stride, uint8_t* gray, void* stream) {
    SFM_CHECK_ARG(bgr && gray && w > 0 && h > 0 && stride >= 3 * w && w < (1 << 20) && h < (1 << 20), "sfm_bgr2gray_u8: bad arguments");
    hipLaunchKernelGGL(bgr2gray_kernel, dim3((unsigned)((w + 63) / 64), (unsigned)((h + 3) / 4)), dim3(256), 0, sfm::as_stream(stream), bgr, (int)w,
                       (int)h, (long)stride, gray);
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}

extern "C" int sfm_pyrdown_u8(const uint8_t* src, int64_t w, int64_t h, int channels, uint8_t* dst, void* stream) {
    SFM_CHECK_ARG(src && dst && w > 0 && h > 0 && channels >= 1 && channels <= 4 && w < (1 << 20) && h < (1 << 20), "sfm_pyrdown_u8: bad arguments");
    const int64_t dw = (w + 1) / 2, dh = (h + 1) / 2;
    hipLaunchKernelGGL(pyrdown_kernel, dim3((unsigned)((dw + 63) / 64), (unsigned)((dh + 3) / 4)), dim3(256), 0, sfm::as_stream(stream), src, (int)w,
                       (int)h, channels, dst);
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}

extern "C" size_t sfm_sift_ws_bytes(int64_t w, int64_t h, int n_octave_layers, int64_t max_keypoints) {
    if (w <= 0 || h <= 0 || n_octave_layers < 1 || max_keypoints < 0) return 0;
    const Layout L = layout(w, h, n_octave_layers);
    sfm::Carver c(nullptr);
    c.take<float>(L.up_floats);
    c.take<float>(L.g_floats);
    c.take<float>(L.d_floats);
    c.take<float2>(L.m_floats);
    c.take<int>((size_t)max_keypoints);
    c.take<Cand>((size_t)max_keypoints);
    c.take<float>((size_t)max_keypoints * 8);
    c.take<float>((size_t)max_keypoints * 8);
    c.take<int>(kCounterInts);
    c.take<int4>((size_t)max_keypoints * kRawPerKeypoint);
    return c.used();
}

extern "C" int sfm_sift_detect_and_compute(const uint8_t* gray, int64_t w, int64_t h, int64_t stride, int n_octave_layers, double contrast_threshold,
                                           double edge_threshold, double sigma, int64_t max_keypoints, float* keypoints, float* descriptors,
                                           int32_t* count, void* ws, size_t ws_bytes, void* stream_) {
    SFM_CHECK_ARG(gray && keypoints && count && ws, "sfm_sift_detect_and_compute: null pointer");
    SFM_CHECK_ARG(w >= 8 && h >= 8 && w <= 16384 && h <= 16384 && stride >= w, "sfm_sift_detect_and_compute: image must be 8..16384 pixels a side");
    SFM_CHECK_ARG(n_octave_layers >= 1 && n_octave_layers <= 8 && sigma > 0 && max_keypoints >= 64 && max_keypoints < (1 << 24),
                  "sfm_sift_detect_and_compute: bad parameters (1..8 layers, sigma > 0, 64 <= max_keypoints < 2^24)");
    if (ws_bytes < sfm_sift_ws_bytes(w, h, n_octave_layers, max_keypoints)) {
        sfm::set_error("sfm_sift_detect_and_compute: workspace too small");
        return SFM_ERR_WORKSPACE;
    }
    hipStream_t stream = sfm::as_stream(stream_);
    const int nL = n_octave_layers, cap = (int)max_keypoints;
    const Layout L = layout(w, h, nL);
    SFM_CHECK_ARG(L.nOct >= 1 && L.nOct <= kMaxOctaves, "sfm_sift_detect_and_compute: octave count out of range");
    sfm::Carver c(ws);
    float* up = c.take<float>(L.up_floats);
    Geom geo;
    geo.g = c.take<float>(L.g_floats);
    geo.d = c.take<float>(L.d_floats);
    geo.grad = c.take<float2>(L.m_floats);
    int* perm = c.take<int>((size_t)max_keypoints);
    geo.W0 = (int)(2 * w); geo.H0 = (int)(2 * h); geo.nL = nL; geo.nOct = L.nOct;
    Cand* cand = c.take<Cand>((size_t)cap);
    float* kp_raw = c.take<float>((size_t)cap * 8);
    float* kp_sorted = c.take<float>((size_t)cap * 8);
    int* counters = c.take<int>(kCounterInts);
    const int raw_cap = (int)std::min<size_t>((size_t)cap * kRawPerKeypoint, (size_t)1 << 30);
    int4* raw = c.take<int4>((size_t)cap * kRawPerKeypoint);

    // per-layer blur taps: sig[i]^2 = (sigma k^i)^2 - (sigma k^(i-1))^2, k = 2^(1/nL); the base blur lifts the assumed
    // 0.5 px camera blur (1.0 after doubling) to sigma
    Taps taps[16];
    {
        const float sf = (float)sigma;
        float sd = sf * sf - 0.5f * 0.5f * 4;
        sd = std::sqrt(sd > 0.01f ? sd : 0.01f);
        SFM_CHECK_ARG(gauss_taps((double)sd, &taps[0]) > 0, "sfm_sift_detect_and_compute: sigma needs more than %d filter taps", kMaxTaps);
        const double k = std::pow(2., 1. / nL);
        for (int i = 1; i < nL + 3; ++i) {
            const double sp = std::pow(k, (double)(i - 1)) * sigma, st = sp * k;
            SFM_CHECK_ARG(gauss_taps(std::sqrt(st * st - sp * sp), &taps[i]) > 0, "sfm_sift_detect_and_compute: sigma needs more than %d filter taps",
                          kMaxTaps);
        }
    }
    SFM_CHECK_HIP(hipMemsetAsync(counters, 0, kCounterInts * sizeof(int), stream));
    auto grid2 = [](int ww, int hh) { return dim3((unsigned)((ww + 63) / 64), (unsigned)((hh + 3) / 4)); };
    auto tiles = [](int ww, int hh) { return dim3((unsigned)((ww + kTileW - 1) / kTileW), (unsigned)((hh + kTileH - 1) / kTileH)); };
    auto launch_blur = [&](dim3 grid, const float* in, float* out, float* dg, int ww, int hh, const Taps& t, int sstep, int spitch) {
        switch (t.n) {
#define SFM_BLUR_CASE(N) case N: hipLaunchKernelGGL(gauss_blur_fixed_kernel<N>, grid, dim3(kBlurThreads), 0, stream, in, out, dg, ww, hh, t, sstep, spitch); return true;
            SFM_BLUR_CASE(5) SFM_BLUR_CASE(7) SFM_BLUR_CASE(9) SFM_BLUR_CASE(11) SFM_BLUR_CASE(13) SFM_BLUR_CASE(15) SFM_BLUR_CASE(17)
            SFM_BLUR_CASE(19) SFM_BLUR_CASE(21) SFM_BLUR_CASE(23) SFM_BLUR_CASE(25) SFM_BLUR_CASE(27)
#undef SFM_BLUR_CASE
            default: return false;
        }
    };
    auto blur_lds = [](const Taps& t) {
        const int R = t.n / 2, rows = kTileH + 2 * R;
        return (size_t)(kMaxTaps + 1 + rows * kTileW + rows * (kTileW + 2 * R)) * sizeof(float);
    };
    sfm::prof_begin(sfm::kProfSiftPyramid, stream);
    hipLaunchKernelGGL(upsample2_kernel, grid2(geo.W0, geo.H0), dim3(256), 0, stream, gray, (int)w, (int)h, (long)stride, up);
    SFM_CHECK_LAUNCH();
    auto blur = [&](const float* in, float* out, float* dg, int ww, int hh, const Taps& t, int sstep, int spitch) {
        if (!launch_blur(tiles(ww, hh), in, out, dg, ww, hh, t, sstep, spitch))
            hipLaunchKernelGGL(gauss_blur_kernel, tiles(ww, hh), dim3(256), blur_lds(t), stream, in, out, dg, ww, hh, t, sstep, spitch);
    };
    blur(up, geo.G(0, 0), nullptr, geo.W0, geo.H0, taps[0], 1, geo.W0);
    SFM_CHECK_LAUNCH();
    // layer i of octave o as a job (layer 1 of octaves >= 1 reads every second pixel of the previous octave's layer nL in place)
    auto job = [&](int o, int i) {
        BlurJob j;
        const int ow = geo.w(o), oh = geo.h(o);
        const bool dec = i == 1 && o > 0;
        j.src = dec ? geo.G(o - 1, nL) : geo.G(o, i - 1);
        j.dst = geo.G(o, i); j.dog = geo.D(o, i - 1);
        j.w = ow; j.h = oh; j.sstep = dec ? 2 : 1; j.spitch = dec ? geo.w(o - 1) : ow;
        j.tiles_x = (ow + kTileW - 1) / kTileW; j.tiles = j.tiles_x * ((oh + kTileH - 1) / kTileH);
        j.taps = taps[i];
        return j;
    };
    auto single = [&](int o, int i) { const BlurJob j = job(o, i); blur(j.src, j.dst, j.dog, j.w, j.h, j.taps, j.sstep, j.spitch); };
    // the default parameters' tap counts for the two shared launches; anything else runs the plain chain
    const bool paired = nL >= 2 && taps[nL + 1].n == 21 && taps[1].n == 11 && taps[nL + 2].n == 27 && taps[2].n == 13;
    for (int o = 0; o < geo.nOct; ++o) {
        for (int i = 1; i < nL + 3; ++i) {
            if (paired && o > 0 && i <= 2) {                   // with layer nL + i of the previous octave
                const BlurJob A = job(o - 1, nL + i), B = job(o, i);
                if (i == 1) hipLaunchKernelGGL((gauss_blur_pair_kernel<21, 11>), dim3((unsigned)(A.tiles + B.tiles)), dim3(kBlurThreads), 0, stream, A, B);
                else hipLaunchKernelGGL((gauss_blur_pair_kernel<27, 13>), dim3((unsigned)(A.tiles + B.tiles)), dim3(kBlurThreads), 0, stream, A, B);
            } else if (paired && i > nL && o + 1 < geo.nOct) {
                continue;                                       // (launched with the next octave's layers 1 and 2)
            } else {
                single(o, i);
            }
            SFM_CHECK_LAUNCH();
        }
    }
    sfm::prof_end(sfm::kProfSiftPyramid, stream);
    {
        int total_tiles = 0;
        size_t grad_wgs = 0;
        for (int o = 0; o < geo.nOct; ++o) { total_tiles += extrema_tiles(geo, o); grad_wgs += (size_t)grad_tiles(geo, o); }
        if (total_tiles > 0) {
            const int threshold = (int)std::floor(0.5 * contrast_threshold / nL * 255);
            hipLaunchKernelGGL(extrema_kernel, dim3((unsigned)total_tiles), dim3(256), 0, stream, geo, threshold, raw, raw_cap, counters);
            SFM_CHECK_LAUNCH();
            hipLaunchKernelGGL(refine_kernel, dim3((unsigned)((raw_cap + 255) / 256)), dim3(256), 0, stream, geo, (const int4*)raw, raw_cap,
                               (float)contrast_threshold, (float)edge_threshold, (float)sigma, cand, counters, cap);
            SFM_CHECK_LAUNCH();
        }
        if (grad_wgs > 0) {
            hipLaunchKernelGGL(gradient_kernel, dim3((unsigned)grad_wgs), dim3(256), 0, stream, geo);
            SFM_CHECK_LAUNCH();
        }
    }
    const unsigned wave_blocks = (unsigned)std::min((cap + 3) / 4, kOriGrid);
    hipLaunchKernelGGL(orientation_kernel, dim3(wave_blocks), dim3(256), 0, stream, geo, (const Cand*)cand, counters, cap, kp_raw);
    SFM_CHECK_LAUNCH();
    hipLaunchKernelGGL(bucket_scatter_kernel, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, stream, (const float*)kp_raw, counters, cap, geo.W0,
                       kp_sorted);
    SFM_CHECK_LAUNCH();
    hipLaunchKernelGGL(bucket_rank_kernel, dim3((unsigned)((cap + 255) / 256)), dim3(256), 0, stream, (const float*)kp_sorted, (const int*)counters, cap,
                       geo.W0, kp_raw);
    SFM_CHECK_LAUNCH();
    hipLaunchKernelGGL(dedupe_kernel, dim3(1), dim3(1024), 0, stream, (const float*)kp_raw, counters, cap, keypoints, count, nL, perm);
    SFM_CHECK_LAUNCH();
    if (descriptors) {
        sfm::prof_begin(sfm::kProfSiftDescriptor, stream);
        hipLaunchKernelGGL(descriptor_kernel<false>, dim3((unsigned)std::min((cap + 3) / 4, kDescGrid)), dim3(256), 0, stream, geo,
                           (const float*)keypoints, (const int*)counters, cap, (const int*)perm, descriptors);
        SFM_CHECK_LAUNCH();
        if ((cap + 3) / 4 > kDescGrid) {
            hipLaunchKernelGGL(descriptor_kernel<true>, dim3(2048), dim3(256), 0, stream, geo, (const float*)keypoints, (const int*)counters, cap,
                               (const int*)perm, descriptors);
            SFM_CHECK_LAUNCH();
        }
        sfm::prof_end(sfm::kProfSiftDescriptor, stream);
    }
    return SFM_OK;
}
