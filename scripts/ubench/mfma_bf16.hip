// Dev microbenchmark: issue behaviour of v_mfma_f32_32x32x16_bf16 dependent chains (1 or 2 accumulators per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int CHAINS, int OCC>
__global__ __launch_bounds__(256, OCC) void k(const unsigned* __restrict__ T, int iters, float* out) {
    u32x4 a, b;
    for (int e = 0; e < 4; ++e) { a[e] = T[threadIdx.x * 4 + e]; b[e] = T[1024 + threadIdx.x * 4 + e]; }
    const bf16x8 A = __builtin_bit_cast(bf16x8, a), B = __builtin_bit_cast(bf16x8, b);
    f32x16 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = (float)(r + c);
    for (int t = 0; t < iters; ++t) {
#pragma unroll
        for (int s = 0; s < 24 / CHAINS; ++s)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, acc[c], 0, 0, 0);
    }
    float r = 0;
    for (int c = 0; c < CHAINS; ++c) for (int q = 0; q < 16; ++q) r += acc[c][q];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int CHAINS, int OCC>
void run(const unsigned* T, float* out) {
    const int blocks = 256 * OCC, iters = 400;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<CHAINS, OCC>), dim3(blocks), dim3(256), 0, 0, T, iters, out);
    (void)hipEventRecord(e0);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<CHAINS, OCC>), dim3(blocks), dim3(256), 0, 0, T, iters, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    const double flop = (double)blocks * 4 * iters * 24 * 2.0 * 32 * 32 * 16;
    const double mfma_per_simd = (double)blocks * 4 * iters * 24 / 1024.0;
    printf("chains=%d waves/SIMD=%d  %7.3f ms  %7.0f TFLOP/s  %5.1f cycles/MFMA/SIMD @2.4GHz\n", CHAINS, OCC, ms, flop / ms / 1e9,
           ms * 1e-3 * 2.4e9 / mfma_per_simd);
}

int main() {
    unsigned* T; float* out;
    (void)hipMalloc(&T, 1 << 20); (void)hipMalloc(&out, 4096 * 256 * 4);
    (void)hipMemset(T, 0x3c, 1 << 20);
    run<1, 1>(T, out); run<2, 1>(T, out); run<4, 1>(T, out);
    run<1, 2>(T, out); run<2, 2>(T, out);
    run<1, 4>(T, out); run<2, 4>(T, out);
    return 0;
}
