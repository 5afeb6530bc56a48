// Dev microbenchmark: the single-product filter's tile loop in isolation (2 x 32-query groups per wave, A fragments
// from LDS, B resident), to separate matrix-pipe time from LDS / VALU / issue effects.
//   V=0 MFMA only (A in registers)   V=1 + ds_read_b128 per k-step   V=2 + packed-key inserts of the previous tile
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int V, bool AUG>
__global__ __launch_bounds__(512, 2) void k(const unsigned* __restrict__ T, int tiles, int* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[3 * 4096];
    for (int i = threadIdx.x; i < 3 * 4096; i += 512) lds[i] = T[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    u32x4 bh[2][8];
    for (int g = 0; g < 2; ++g) for (int s = 0; s < 8; ++s) for (int e = 0; e < 4; ++e) bh[g][s][e] = T[(threadIdx.x * 64 + g * 32 + s * 4 + e) & 0xFFFF];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds;
    f32x16 acc[2][2];
    for (int a = 0; a < 2; ++a) for (int g = 0; g < 2; ++g) for (int r = 0; r < 16; ++r) acc[a][g][r] = 0.f;
    int k0[2] = {0x7f800000, 0x7f800000}, k1[2] = {0x7f800000, 0x7f800000}, k2[2] = {0x7f800000, 0x7f800000};
    int vmask;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vmask) : "s"(~511));
    u32x4 areg[8];
    for (int s = 0; s < 8; ++s) for (int e = 0; e < 4; ++e) areg[s][e] = T[(lane * 32 + s * 4 + e) & 0xFFFF];
    const float aaug = h ? 1.f : 3.f, baug = h ? 2.f : 1.f;
    auto tile = [&](f32x16(&cur)[2], f32x16(&prev)[2], int t) {
        const unsigned abase = lds0 + (unsigned)(t % 3) * 16384u + (unsigned)j * 256u + ((unsigned)(h ^ (j & 15)) << 4);
        u32x4 ah[8];
        if (V >= 1) {
#pragma unroll
            for (int s = 0; s < 8; ++s) asm volatile("ds_read_b128 %0, %1" : "=v"(ah[s]) : "v"(abase ^ (32u * s)));
        }
        if (AUG) {
            const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            cur[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(aaug, baug, zero, 0, 0, 0);
            cur[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(aaug, baug, zero, 0, 0, 0);
        }
        const int seq0 = __builtin_amdgcn_readfirstlane((t & 31) << 4);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
            if (V >= 1) {
                switch (s) {
                    case 0: asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(ah[0])); break;
                    case 1: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(ah[1])); break;
                    case 2: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(ah[2])); break;
                    case 3: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ah[3])); break;
                    case 4: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(ah[4])); break;
                    case 5: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah[5])); break;
                    case 6: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(ah[6])); break;
                    default: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[7])); break;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
#pragma unroll
            for (int g = 0; g < 2; ++g)
                cur[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, V >= 1 ? ah[s] : areg[s]),
                                                               __builtin_bit_cast(f16x8, bh[g][s]), cur[g], 0, 0, 0);
            if (V == 2) {
#pragma unroll
                for (int g = 0; g < 2; ++g)
#pragma unroll
                    for (int r = 2 * s; r < 2 * s + 2; ++r) {
                        int key;
                        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(prev[g][r]), "v"(vmask), "s"(seq0 + r));
                        const int lo = min(key, k0[g]);
                        const int m1 = max(min(key, k1[g]), min(max(key, k1[g]), k0[g]));
                        k2[g] = max(min(key, k1[g]), min(max(key, k1[g]), k2[g]));
                        k1[g] = m1;
                        k0[g] = lo;
                    }
            } else if (V == 3) {       // pair-min: one insertion per two adjacent scores (candidates would be row pairs)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int m = min(__float_as_int(prev[g][2 * s]), __float_as_int(prev[g][2 * s + 1]));
                    int key;
                    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(m), "v"(vmask), "s"(seq0 + s));
                    const int lo = min(key, k0[g]);
                    const int m1 = max(min(key, k1[g]), min(max(key, k1[g]), k0[g]));
                    k2[g] = max(min(key, k1[g]), min(max(key, k1[g]), k2[g]));
                    k1[g] = m1;
                    k0[g] = lo;
                }
            } else {
                k0[0] = min(k0[0], __float_as_int(prev[0][2 * s]) + __float_as_int(prev[0][2 * s + 1]));
                k0[1] = min(k0[1], __float_as_int(prev[1][2 * s]) + __float_as_int(prev[1][2 * s + 1]));
            }
        }
    };
    for (int t = 0; t < tiles; t += 2) {
        tile(acc[0], acc[1], t);
        tile(acc[1], acc[0], t + 1);
    }
    out[blockIdx.x * 512 + threadIdx.x] = k0[0] + k1[0] + k2[0] + k0[1] + k1[1] + k2[1];
}

template <int V, bool AUG>
void run(const unsigned* T, int* out, int tiles) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<V, AUG>), dim3(256), dim3(512), 0, 0, T, tiles, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<V, AUG>), dim3(256), dim3(512), 0, 0, T, tiles, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    // per SIMD: 2 waves x tiles tile-steps
    const double cyc = ms * 1e-3 * 2.06e9 / (2.0 * tiles);
    printf("V=%d aug=%d tiles=%5d  %8.2f us  %6.0f cycles per wave-tile-step (MFMA demand %d)\n", V, (int)AUG, tiles, ms * 1e3, cyc, 512 + (AUG ? 128 : 0));
}

int main() {
    unsigned* T; int* out;
    (void)hipMalloc(&T, 1 << 20); (void)hipMalloc(&out, 256 * 512 * 4);
    unsigned* h = (unsigned*)malloc(1 << 20);
    for (int i = 0; i < (1 << 18); ++i) h[i] = 0x38003800u + ((i * 2654435761u) & 0x03FF03FFu);   // fp16 values in [0.5, 1)
    (void)hipMemcpy(T, h, 1 << 20, hipMemcpyHostToDevice);
    for (int tiles : {24, 2400}) {
        run<0, false>(T, out, tiles); run<0, true>(T, out, tiles);
        run<1, false>(T, out, tiles); run<1, true>(T, out, tiles);
        run<2, false>(T, out, tiles); run<2, true>(T, out, tiles);
        run<3, true>(T, out, tiles);
    }
    return 0;
}
