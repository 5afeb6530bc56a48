// Dev microbenchmark: what does a PURE 32x32 MFMA stream sustain on this part, and at what shader clock, as a function of
// the operand data?  (is the "power-limited clock" of DESIGN.md 4.1 real?)
//   KIND 0: v_mfma_f32_32x32x16_f16   1: v_mfma_f32_32x32x16_bf16   2: v_mfma_i32_32x32x32_i8
//   four independent accumulator chains per wave, OCC waves per SIMD, no memory traffic in the loop;
//   operands: zeros | one constant (0x3c3c3c3c, as scripts/ubench/mfma_bf16.hip) | random finite values (different per lane,
//   A and B re-drawn from 8 register sets round-robin so that the operand buses toggle every MFMA).
// Prints sustained T(FL)OP/s, the shader clock measured inside the kernel (s_memtime / s_memrealtime) and cycles per MFMA.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int KIND, int OCC>
__global__ __launch_bounds__(256, OCC) void k(const unsigned* __restrict__ T, int iters, float* out, long long* clk) {
    long long c0 = 0, w0 = 0;
    if (threadIdx.x == 0) { c0 = clock64(); w0 = wall_clock64(); }
    u32x4 a[8], b[8];
    for (int s = 0; s < 8; ++s)
        for (int e = 0; e < 4; ++e) {
            a[s][e] = T[((blockIdx.x * 256 + threadIdx.x) * 64 + s * 8 + e) & 0x3FFFF];
            b[s][e] = T[((blockIdx.x * 256 + threadIdx.x) * 64 + s * 8 + 4 + e) & 0x3FFFF];
        }
    using acc_t = std::conditional_t<KIND == 2, i32x16, f32x16>;
    acc_t acc[4];
    for (int c = 0; c < 4; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0;
    for (int t = 0; t < iters; ++t) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if constexpr (KIND == 0) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[s]), __builtin_bit_cast(f16x8, b[(s + c) & 7]), acc[c], 0, 0, 0);
                else if constexpr (KIND == 1) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[s]), __builtin_bit_cast(bf16x8, b[(s + c) & 7]), acc[c], 0, 0, 0);
                else acc[c] = __builtin_amdgcn_mfma_i32_32x32x32_i8(__builtin_bit_cast(i32x4, a[s]), __builtin_bit_cast(i32x4, b[(s + c) & 7]), acc[c], 0, 0, 0);
            }
    }
    float r = 0;
    for (int c = 0; c < 4; ++c) for (int q = 0; q < 16; ++q) r += (float)acc[c][q];
    out[blockIdx.x * 256 + threadIdx.x] = r;
    if (threadIdx.x == 0) {
        clk[2 * blockIdx.x] = clock64() - c0;
        clk[2 * blockIdx.x + 1] = wall_clock64() - w0;
    }
}

template <int KIND, int OCC>
void run(const unsigned* T, float* out, long long* clk, const char* data) {
    const int blocks = 256 * OCC, iters = 1500 / OCC;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((k<KIND, OCC>), dim3(blocks), dim3(256), 0, 0, T, iters, out, clk);   // ramp the clocks / power state
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<KIND, OCC>), dim3(blocks), dim3(256), 0, 0, T, iters, out, clk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<long long> c(2 * blocks);
    (void)hipMemcpy(c.data(), clk, c.size() * 8, hipMemcpyDeviceToHost);
    double ghz = 0;
    for (int b = 0; b < blocks; ++b) ghz += (double)c[2 * b] / (double)c[2 * b + 1] * 0.1;
    ghz /= blocks;
    const double n_mfma = (double)blocks * 4 * iters * 32;
    const double ops = n_mfma * 2.0 * 32 * 32 * (KIND == 2 ? 32 : 16);
    const char* names[3] = {"f32_32x32x16_f16 ", "f32_32x32x16_bf16", "i32_32x32x32_i8  "};
    printf("| %s | %d | %-8s | %8.3f | %6.0f | %5.3f | %5.1f | %5.3f |\n", names[KIND], OCC, data, ms, ops / ms / 1e9, ghz,
           ms * 1e-3 * ghz * 1e9 / (n_mfma / 1024.0), ops / ms / 1e9 / (KIND == 2 ? 5000.0 : 2500.0));
}

int main() {
    unsigned* T; float* out; long long* clk;
    (void)hipMalloc(&T, 1 << 20); (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&clk, 1024 * 16);
    std::vector<unsigned> h(1 << 18);
    printf("| MFMA | waves/SIMD | operands | ms/launch | T(FL)OP/s | shader GHz | cycles/MFMA/SIMD | frac of 2.5 / 5 P |\n|---|---|---|---|---|---|---|---|\n");
    for (int pass = 0; pass < 4; ++pass) {
        const char* data = pass == 0 ? "zero" : pass == 1 ? "constant" : pass == 2 ? "random" : "sift-u8";
        unsigned s = 777u;
        for (auto& v : h) {
            s = s * 1664525u + 1013904223u;
            const unsigned r = s ^ (s >> 15);
            if (pass == 0) v = 0u;
            else if (pass == 1) v = 0x3c3c3c3cu;
            else if (pass == 2) v = (r & 0xBFFFBFFFu) | 0x20002000u;      // finite fp16 / bf16 of mixed sign and moderate exponent; as bytes: arbitrary
            else v = r & 0x3F3F3F3Fu;                                      // small non-negative bytes (SIFT-like integers); as fp16: small normals
        }
        (void)hipMemcpy(T, h.data(), 1 << 20, hipMemcpyHostToDevice);
        run<0, 1>(T, out, clk, data); run<1, 1>(T, out, clk, data); run<2, 1>(T, out, clk, data);
        run<0, 2>(T, out, clk, data); run<2, 2>(T, out, clk, data);
    }
    return 0;
}
