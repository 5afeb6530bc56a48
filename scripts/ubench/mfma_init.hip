// Dev microbenchmark: pipe time of the filter's per-tile MFMA mix — 8 x v_mfma_f32_32x32x16_f16 per query group plus the
// accumulator initialisation as (0) nothing, (1) v_mfma_f32_32x32x2_f32, (2) v_mfma_f32_32x32x8_bf16; 2 groups per wave.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int INIT, int OCC>
__global__ __launch_bounds__(256, OCC) void k(const unsigned* __restrict__ T, int iters, float* out) {
    u32x4 a, b;
    for (int e = 0; e < 4; ++e) { a[e] = T[threadIdx.x * 4 + e]; b[e] = T[1024 + threadIdx.x * 4 + e]; }
    const float fa = __uint_as_float(T[threadIdx.x]), fb = __uint_as_float(T[threadIdx.x + 7]);
    s16x4 sa, sb;
    for (int e = 0; e < 4; ++e) { sa[e] = (short)T[threadIdx.x + e]; sb[e] = (short)T[99 + threadIdx.x + e]; }
    f32x16 acc[2], sum[2];
    const f32x16 zero = {};
    for (int c = 0; c < 2; ++c) sum[c] = zero;
    for (int t = 0; t < iters; ++t) {
        asm volatile("" : "+v"(a), "+v"(b));                 // (loop-variant operands: nothing is hoisted)
        const f16x8 A = __builtin_bit_cast(f16x8, a), B = __builtin_bit_cast(f16x8, b);
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (INIT == 1) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(c ? fb : fa, fb, zero, 0, 0, 0);
            else if (INIT == 2) acc[c] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(c ? sb : sa, sb, zero, 0, 0, 0);
            else acc[c] = sum[c];
        }
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int c = 0; c < 2; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(c ? B : A, B, acc[c], 0, 0, 0);
        for (int c = 0; c < 2; ++c) sum[c][t & 15] += acc[c][(t >> 4) & 15];
    }
    float r = 0;
    for (int c = 0; c < 2; ++c) for (int q = 0; q < 16; ++q) r += sum[c][q];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int INIT, int OCC>
void run(const unsigned* T, float* out) {
    const int blocks = 256 * OCC, iters = 2000;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<INIT, OCC>), dim3(blocks), dim3(256), 0, 0, T, iters, out);
    (void)hipEventRecord(e0);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<INIT, OCC>), dim3(blocks), dim3(256), 0, 0, T, iters, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double tiles_per_simd = (double)blocks * 4 * iters / 1024.0;     // wave-tiles (2 groups each)
    printf("init=%d waves/SIMD=%d  %7.3f ms  %6.1f ns per wave-tile per SIMD  (%5.0f cycles @2.0GHz; 16 x 32 = 512 for the products)\n", INIT, OCC, ms,
           ms * 1e6 / tiles_per_simd, ms * 1e-3 * 2.0e9 / tiles_per_simd);
}

int main() {
    unsigned* T; float* out;
    (void)hipMalloc(&T, 1 << 20); (void)hipMalloc(&out, 4096 * 256 * 4);
    (void)hipMemset(T, 0x3c, 1 << 20);
    run<0, 2>(T, out); run<1, 2>(T, out); run<2, 2>(T, out);
    run<0, 1>(T, out); run<1, 1>(T, out); run<2, 1>(T, out);
    return 0;
}
