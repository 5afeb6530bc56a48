// Dev measurement: how exactly does v_mfma_f32_32x32x16_f16 add?  For random and adversarial operands the result is compared
// with the exact value c + sum_k a_k b_k (fp64: the products of two fp16 are exact, 17 terms fit) and the error is reported
// in units of 2^-24 (|c| + sum |a_k b_k|) — the quantity the KNN certificate's "MFMA chain" term is priced in.
// Every output element of every MFMA is a separate sample (operands are generated from a hash, so a lane can recompute
// the row / column of any element it holds).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ inline uint32_t hash3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}
// fp16 value with exponent in [e0, e0 + span), random sign (if signed_) and mantissa
__device__ inline _Float16 gen16(uint32_t h, int e0, int span, bool signed_) {
    const int e = e0 + (int)((h >> 11) % (uint32_t)span);
    const float m = 1.f + (float)(h & 1023u) * (1.f / 1024.f);
    const float v = ldexpf(m, e) * ((signed_ && (h >> 31)) ? -1.f : 1.f);
    return (_Float16)v;
}
__device__ inline float gen32(uint32_t h, int e0, int span, bool signed_) {
    const int e = e0 + (int)((h >> 24) % (uint32_t)span);
    const float m = 1.f + (float)(h & 0x7FFFFFu) * (1.f / 8388608.f);
    return ldexpf(m, e) * ((signed_ && ((h >> 23) & 1)) ? -1.f : 1.f);
}

struct Regime { int ea, sa, eb, sb, ec, sc, signed_; };

__global__ __launch_bounds__(256) void k(Regime rg, int trials, uint32_t seed, double* maxerr) {
    const int lane = threadIdx.x & 63, wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int rc = lane & 31, h = lane >> 5;
    double worst = 0;
    for (int t = 0; t < trials; ++t) {
        const uint32_t s = seed + 7919u * (uint32_t)(wid * trials + t);
        f16x8 A, B;
        for (int e = 0; e < 8; ++e) {
            A[e] = gen16(hash3(s, rc, 8 * h + e), rg.ea, rg.sa, rg.signed_);               // A[row rc][k = 8h + e]
            B[e] = gen16(hash3(s ^ 0xABCDu, rc, 8 * h + e), rg.eb, rg.sb, rg.signed_);     // B[k][col rc]
        }
        f32x16 C;
        for (int r = 0; r < 16; ++r) C[r] = gen32(hash3(s ^ 0x1234u, 8 * (r >> 2) + 4 * h + (r & 3), rc), rg.ec, rg.sc, rg.signed_);
        const f32x16 D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0);
        for (int r = 0; r < 16; ++r) {
            const int row = 8 * (r >> 2) + 4 * h + (r & 3);
            double ref = (double)C[r], mag = fabs((double)C[r]);
            for (int kk = 0; kk < 16; ++kk) {
                const double a = (double)(float)gen16(hash3(s, row, kk), rg.ea, rg.sa, rg.signed_);
                const double b = (double)(float)gen16(hash3(s ^ 0xABCDu, rc, kk), rg.eb, rg.sb, rg.signed_);
                ref += a * b;
                mag += fabs(a * b);
            }
            const double err = fabs((double)D[r] - ref) / (mag * 5.9604644775390625e-08);
            if (err > worst) worst = err;
        }
    }
    for (int m = 32; m >= 1; m >>= 1) { const double o = __shfl_xor(worst, m, 64); if (o > worst) worst = o; }
    if (lane == 0) maxerr[wid] = worst;
}

int main() {
    const int blocks = 1024, trials = 200;
    double* d; (void)hipMalloc(&d, blocks * 4 * sizeof(double));
    double* hbuf = new double[blocks * 4];
    const Regime regs[] = {
        {0, 1, 0, 1, 4, 1, 0},      // equal exponents, positive: the textbook case
        {-3, 6, -3, 6, 0, 8, 1},    // spread exponents, signed (cancellation)
        {-8, 16, -8, 16, -8, 24, 1},   // wide spread
        {0, 1, 0, 1, 20, 1, 1},     // c dominates
        {6, 2, 6, 2, -10, 4, 1},    // products dominate
        {-14, 4, 0, 4, -10, 8, 1},  // small operands (down to fp16's smallest normals)
        {-1, 2, -1, 2, 5, 3, 0},    // the KNN filter's own regime: operands ~ [0.5, 2), accumulator ~ 32..256, positive
    };
    for (const Regime& rg : regs) {
        double worst = 0;
        for (int rep = 0; rep < 4; ++rep) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, rg, trials, 0x1000u * rep + 17u, d);
            (void)hipMemcpy(hbuf, d, blocks * 4 * sizeof(double), hipMemcpyDeviceToHost);
            for (int i = 0; i < blocks * 4; ++i) worst = std::fmax(worst, hbuf[i]);
        }
        printf("A 2^[%d,%d) B 2^[%d,%d) C 2^[%d,%d) %s: max |D - exact| = %.3f x 2^-24 (|c| + sum |a b|)   [%ld samples]\n", rg.ea, rg.ea + rg.sa, rg.eb,
               rg.eb + rg.sb, rg.ec, rg.ec + rg.sc, rg.signed_ ? "signed" : "positive", worst, 4L * blocks * 4 * trials * 1024);
    }
    return 0;
}
