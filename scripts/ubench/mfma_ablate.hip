// Dev microbenchmark: what keeps v_mfma_f32_32x32x2_f32 from issuing back to back in the knn filter?
// Variants add one ingredient of the filter's tile loop at a time.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;
constexpr float kInf = __builtin_huge_valf();

__device__ __forceinline__ void top3_insert(float v, int id, float (&s)[3], int (&ix)[3]) {
    const bool lt0 = v < s[0], lt1 = v < s[1], lt2 = v < s[2];
    s[2] = lt1 ? s[1] : (lt2 ? v : s[2]);
    ix[2] = lt1 ? ix[1] : (lt2 ? id : ix[2]);
    s[1] = lt0 ? s[0] : (lt1 ? v : s[1]);
    ix[1] = lt0 ? ix[0] : (lt1 ? id : ix[1]);
    s[0] = lt0 ? v : s[0];
    ix[0] = lt0 ? id : ix[0];
}

// FLAGS bit0: LDS a-frag reads (asm prefetch)  bit1: epilogue  bit2: barrier per tile  bit3: glds staging per tile
template <int FLAGS, int OCC>
__global__ __launch_bounds__(256, OCC) void k(const float* __restrict__ T, int tiles, float* out) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float bq[64];
#pragma unroll
    for (int c = 0; c < 64; ++c) bq[c] = T[(blockIdx.x * 256 + threadIdx.x) * 64 % 4096 + c] * 0.001f;
    if (FLAGS & 1) {
        for (int i = threadIdx.x; i < 2 * 4096 + 64; i += 256) smem[i] = T[i % 4096] * 0.001f;
        __syncthreads();
    }
    float bs[3] = {kInf, kInf, kInf};
    int bi[3] = {-1, -1, -1};
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
    const int hm = h ^ (j & 15);
    __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)T, 0, 1 << 24, 0x00020000);
    const int lane_off = (wave * 8 + h) * 512 + (((lane & 31) ^ ((wave * 8 + h) & 15)) << 4);
    f32x16 keep = {0};
    for (int t = 0; t < tiles; ++t) {
        const int cur = t & 1;
        if (FLAGS & 8) {
#pragma unroll
            for (int n = 0; n < 4; ++n)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(trs, (lptr_t)(smem + (cur ^ 1) * 4096 + (wave * 4 + n) * 256), 16,
                                                         (lane_off ^ (32 * n)) + 2 * n * 512, (t % 64) * 16384, 0, 0);
        }
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = (float)r;
        if (FLAGS & 1) {
            const unsigned abase = lds0 + (unsigned)(cur * 4096 + j * 128) * 4u + ((unsigned)hm << 4);
            f32x4 af[3];
            asm volatile("ds_read_b128 %0, %1" : "=v"(af[0]) : "v"(abase));
            asm volatile("ds_read_b128 %0, %1" : "=v"(af[1]) : "v"(abase ^ 32u));
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                if (c + 2 < 16) {
                    const unsigned ad = (abase ^ (32u * ((c + 2) & 7))) + ((c + 2) >= 8 ? 256u : 0u);
                    asm volatile("ds_read_b128 %0, %1" : "=v"(af[(c + 2) % 3]) : "v"(ad));
                    asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(af[c % 3]));
                } else if (c + 1 < 16) {
                    asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(af[c % 3]));
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[c % 3]));
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c % 3][e], bq[4 * c + e], acc, 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int c = 0; c < 64; ++c) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bq[(c + 1) & 63], bq[c], acc, 0, 0, 0);
        }
        if (FLAGS & 2) {
#pragma unroll
            for (int r = 0; r < 16; ++r) top3_insert(acc[r], t * 32 + r, bs, bi);
        } else {
            keep += acc;
        }
        if (FLAGS & 4) __syncthreads();
    }
    float r = bs[0] + bs[1] + bs[2] + bi[0] + bi[1] + bi[2];
#pragma unroll
    for (int q = 0; q < 16; ++q) r += keep[q];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int FLAGS, int OCC>
void run(const char* name, const float* T, float* out, int blocks) {
    const int tiles = 48;
    const size_t lds = (FLAGS & 1) ? (2 * 4096 + 64) * 4 : 0;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((k<FLAGS, OCC>), dim3(blocks), dim3(256), lds, 0, T, tiles, out);
    hipEventRecord(a);
    const int reps = 5;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((k<FLAGS, OCC>), dim3(blocks), dim3(256), lds, 0, T, tiles, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
    const double flop = (double)blocks * 4 * tiles * 64 * 4096.0;
    printf("%-34s occ=%d blocks=%5d  %8.3f ms  %7.1f TFLOP/s (%.0f%% of 157.3)\n", name, OCC, blocks, ms, flop / ms / 1e9, flop / ms / 1e9 / 157.3 * 100);
}

int main() {
    float *T, *out;
    hipMalloc(&T, 1 << 24); hipMalloc(&out, 4096 * 256 * 4);
    std::vector<float> h((1 << 24) / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) % 1000) * 0.001f;
    hipMemcpy(T, h.data(), 1 << 24, hipMemcpyHostToDevice);
    run<0, 1>("pure MFMA", T, out, 256);
    run<0, 2>("pure MFMA", T, out, 512);
    run<0, 4>("pure MFMA", T, out, 1024);
    run<1, 4>("+lds frag reads", T, out, 1024);
    run<3, 4>("+lds +epilogue", T, out, 1024);
    run<2, 4>("+epilogue only", T, out, 1024);
    run<5, 4>("+lds +barrier", T, out, 1024);
    run<7, 4>("+lds +epi +barrier", T, out, 1024);
    run<15, 4>("+lds +epi +barrier +glds", T, out, 1024);
    run<15, 2>("+lds +epi +barrier +glds", T, out, 512);
    run<7, 2>("+lds +epi +barrier", T, out, 512);
    run<3, 2>("+lds +epilogue", T, out, 512);
    run<2, 1>("+epilogue only", T, out, 256);
    return 0;
}
