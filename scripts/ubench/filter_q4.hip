// Dev microbenchmark: tile loop of the "q4" filter — ONE wave per SIMD (256-thread workgroup, one per CU, 512 registers),
// FOUR 32-query groups per wave whose B fragments live in AGPRs, accumulators single-buffered in VGPRs, the accumulator init
// ||t||^2 + ||q||^2 as one bf16 MFMA on exact bf16 triples, the packed-key epilogue of two groups interleaved into the MFMA
// chains of the other two.
//   V=0 MFMA only   V=1 + A fragments from LDS (rolling refill)   V=2 + packed-key inserts   V=3 + LDS-DMA ring + barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int kSlotBytes = 2 * 8192 + 2 * 512;     // two fp16 tiles + their two init-fragment tiles
constexpr int kRing = 3;

#define MFMA_F16(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b))
#define MFMA_INIT(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "a"(b))

__device__ __forceinline__ void key_insert_quad(const f32x16& a, int r, int seq /*sgpr*/, int vmask, int& k0, int& k1, int& k2) {
    const int m = min(min(__float_as_int(a[r]), __float_as_int(a[r + 1])), min(__float_as_int(a[r + 2]), __float_as_int(a[r + 3])));
    int key;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(m), "v"(vmask), "s"(seq));
    const int lo = min(key, k0);
    const int m1 = max(min(key, k1), min(max(key, k1), k0));
    k2 = max(min(key, k1), min(max(key, k1), k2));
    k1 = m1;
    k0 = lo;
}

template <int V>
__global__ __launch_bounds__(256, 1) void k(const unsigned* __restrict__ T, int tiles, int* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds[];
    for (int i = threadIdx.x; i < kRing * kSlotBytes / 4; i += 256) lds[i] = T[i];
    __syncthreads();
    const int lane = threadIdx.x & 63, j = lane & 31, h = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    u32x4 bq[4][8], bi[4];
    for (int g = 0; g < 4; ++g) {
        for (int s = 0; s < 8; ++s) {
            u32x4 v;
            for (int e = 0; e < 4; ++e) v[e] = T[(threadIdx.x * 128 + g * 32 + s * 4 + e) & 0xFFFF];
            asm volatile("" : "=a"(bq[g][s]) : "0"(v));
        }
        u32x4 v = {0x3F803F80u, 0x3F803F80u, 0u, 0u};
        if (h) v = u32x4{0u, 0u, 0u, 0u};
        asm volatile("" : "=a"(bi[g]) : "0"(v));
    }
    asm volatile("s_nop 4");
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned*)lds;
    f32x16 acc[4];
    for (int g = 0; g < 4; ++g) for (int r = 0; r < 16; ++r) acc[g][r] = __builtin_huge_valf();
    int k0[4], k1[4], k2[4];
    for (int g = 0; g < 4; ++g) k0[g] = k1[g] = k2[g] = 0x7f800000;
    int vmask;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vmask) : "s"(~255));
    unsigned fa[8];
    for (int st = 0; st < 8; ++st) fa[st] = lds0 + (((unsigned)j * 256u + ((unsigned)(h ^ (j & 15)) << 4)) ^ (32u * st));
    unsigned fi = lds0 + 16384u + (h ? 1008u : (unsigned)j * 16u);     // init fragment (h = 1 lanes: a zero line)
    u32x4 ah[8], ai;
    for (int s = 0; s < 8; ++s) for (int e = 0; e < 4; ++e) ah[s][e] = T[(lane * 32 + s * 4 + e) & 0xFFFF];
    for (int e = 0; e < 4; ++e) ai[e] = T[(lane * 4 + e) & 0xFFFF];
    int rbuf = 0;
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)T, 0, 1 << 20, 0x00020000);
    const int lane_off = (lane >> 4) * 256 + ((lane & 15) << 4);

    auto stage = [&](int tile, int buf) {       // this wave's quarter of a slot: 4 x 1 KiB of the two tiles + 256 B of init fragments
        const unsigned dst0 = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)buf * kSlotBytes + (unsigned)wave * 4096u);
        const int soff = __builtin_amdgcn_readfirstlane(((tile * 8192 + wave * 4096) & 0x7FFFF));
#pragma unroll
        for (int n = 0; n < 4; ++n)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(trs, (lptr_t)(size_t)(dst0 + n * 1024), 16, lane_off + n * 1024, soff, 0, 0);
        const unsigned dsti = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)buf * kSlotBytes + 16384u + (unsigned)wave * 256u);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(trs, (lptr_t)(size_t)dsti, 4, lane * 4, soff, 0, 0);
    };
    if (V >= 3) { stage(0, 0); stage(2, 1); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); }

    auto tile = [&](int t, auto half_c) {
        constexpr int kHalf = decltype(half_c)::value;
        if (V >= 3 && kHalf == 0 && t + 4 < tiles) stage(t + 4, rbuf + 2 >= kRing ? rbuf + 2 - kRing : rbuf + 2);
        const int seq_prev = __builtin_amdgcn_readfirstlane(((t - 1) & 63) << 2), seq_cur = __builtin_amdgcn_readfirstlane((t & 63) << 2);
        // ---- phase A: chains of groups 0, 1; epilogue of groups 2, 3 of the previous tile
        if (V >= 1) asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(ai));
        MFMA_INIT(acc[0], ai, bi[0]);
        MFMA_INIT(acc[1], ai, bi[1]);
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            if (V >= 1) {
                switch (st) {
                    case 0: asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(ah[0])); break;
                    case 1: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(ah[1])); break;
                    case 2: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(ah[2])); break;
                    case 3: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ah[3])); break;
                    case 4: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(ah[4])); break;
                    case 5: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah[5])); break;
                    case 6: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(ah[6])); break;
                    default: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[7])); break;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            MFMA_F16(acc[0], ah[st], bq[0][st]);
            MFMA_F16(acc[1], ah[st], bq[1][st]);
            if (V >= 2) {
                const int g = 2 + (st >> 2);
                key_insert_quad(acc[g], 4 * (st & 3), seq_prev + (st & 3), vmask, k0[g], k1[g], k2[g]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- phase B: chains of groups 2, 3; epilogue of groups 0, 1 of this tile; A fragments of the next tile requested
        MFMA_INIT(acc[2], ai, bi[2]);
        MFMA_INIT(acc[3], ai, bi[3]);
        if (V >= 1) {
            if constexpr (kHalf == 0) asm volatile("ds_read_b128 %0, %1 offset:512" : "=v"(ai) : "v"(fi));
            else {
                const int nb = rbuf == kRing - 1 ? 0 : rbuf + 1;
                fi += (unsigned)(nb == 0 ? -(kRing - 1) * kSlotBytes : kSlotBytes);
                asm volatile("ds_read_b128 %0, %1" : "=v"(ai) : "v"(fi));
            }
        }
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            __builtin_amdgcn_sched_barrier(0);
            MFMA_F16(acc[2], ah[st], bq[2][st]);
            MFMA_F16(acc[3], ah[st], bq[3][st]);
            if (V >= 1) {
                if constexpr (kHalf == 0) asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(ah[st]) : "v"(fa[st]));
                else {
                    const int nb = rbuf == kRing - 1 ? 0 : rbuf + 1;
                    fa[st] += (unsigned)(nb == 0 ? -(kRing - 1) * kSlotBytes : kSlotBytes);
                    asm volatile("ds_read_b128 %0, %1" : "=v"(ah[st]) : "v"(fa[st]));
                }
            }
            if (V >= 2) {
                const int g = st >> 2;
                key_insert_quad(acc[g], 4 * (st & 3), seq_cur + (st & 3), vmask, k0[g], k1[g], k2[g]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (kHalf == 1) {
            rbuf = rbuf == kRing - 1 ? 0 : rbuf + 1;
            if (V >= 3 && t + 1 < tiles) {
                // the next slot must have landed for every wave; the slot after it (5 pieces, just issued or about to be) stays in flight
                if (t + 3 < tiles) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
            }
        }
    };
    for (int t = 0; t < tiles; t += 2) {
        tile(t, std::integral_constant<int, 0>{});
        tile(t + 1, std::integral_constant<int, 1>{});
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ai), "+v"(ah[0]), "+v"(ah[1]), "+v"(ah[2]), "+v"(ah[3]), "+v"(ah[4]), "+v"(ah[5]), "+v"(ah[6]), "+v"(ah[7]));
    asm volatile("s_nop 15\n\ts_nop 15");
    int s = 0;
    for (int g = 0; g < 4; ++g) s += k0[g] + k1[g] + k2[g] + __float_as_int(acc[g][0]) + __float_as_int(acc[g][15]);
    out[blockIdx.x * 256 + threadIdx.x] = s;
}


// V=4: no LDS at all.  The train image is stored in FRAGMENT ORDER ([tile][9 fragments][64 lanes][16 B]: 8 k-steps + the
// init fragment), so one buffer_load_dwordx4 per fragment reads 1 KiB contiguous straight into the MFMA's A registers; a
// ring of D tiles of fragments in VGPRs (refilled as the fragments die) replaces the LDS ring, the barriers, the LDS-DMA
// and the ds_reads.  The four waves of a workgroup walk the same tiles (L1 / L2 hits), nothing synchronises them.
typedef int i32x4 __attribute__((ext_vector_type(4)));
template <int D>
__global__ __launch_bounds__(256, 1) void k4(const unsigned* __restrict__ T, int img_tiles, int tiles, int* out) {
    const int lane = threadIdx.x & 63, h = lane >> 5;
    u32x4 bq[4][8], bi[4];
    for (int g = 0; g < 4; ++g) {
        for (int s = 0; s < 8; ++s) {
            u32x4 v;
            for (int e = 0; e < 4; ++e) v[e] = T[(threadIdx.x * 128 + g * 32 + s * 4 + e) & 0xFFFF];
            asm volatile("" : "=a"(bq[g][s]) : "0"(v));
        }
        u32x4 v = {0x3F803F80u, 0x3F803F80u, 0u, 0u};
        if (h) v = u32x4{0u, 0u, 0u, 0u};
        asm volatile("" : "=a"(bi[g]) : "0"(v));
    }
    asm volatile("s_nop 4");
    f32x16 acc[4];
    for (int g = 0; g < 4; ++g) for (int r = 0; r < 16; ++r) acc[g][r] = __builtin_huge_valf();
    int k0[4], k1[4], k2[4];
    for (int g = 0; g < 4; ++g) k0[g] = k1[g] = k2[g] = 0x7f800000;
    int vmask;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vmask) : "s"(~255));
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)T, 0, img_tiles * 9216, 0x00020000);
    const int voff = lane * 16;
    // this workgroup's tile range starts somewhere in the image (as a stream-K range would) and wraps
    int tile0 = (int)((blockIdx.x * 977u) % (unsigned)img_tiles);
    i32x4 fr[D][9];
    auto load_frag = [&](int slot, int f, int tile) {
        const int soff = __builtin_amdgcn_readfirstlane((tile % img_tiles) * 9216 + f * 1024);
        fr[slot][f] = __builtin_amdgcn_raw_buffer_load_b128(trs, voff, soff, 0);
    };
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int f = 0; f < 9; ++f) load_frag(d, f, tile0 + d);

    auto tile = [&](int t, auto slot_c) {
        constexpr int S = decltype(slot_c)::value;
        const int seq_prev = __builtin_amdgcn_readfirstlane(((t - 1) & 63) << 2), seq_cur = __builtin_amdgcn_readfirstlane((t & 63) << 2);
        const u32x4 ai = __builtin_bit_cast(u32x4, fr[S][8]);
        MFMA_INIT(acc[0], ai, bi[0]);
        MFMA_INIT(acc[1], ai, bi[1]);
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 a = __builtin_bit_cast(u32x4, fr[S][st]);
            MFMA_F16(acc[0], a, bq[0][st]);
            MFMA_F16(acc[1], a, bq[1][st]);
            const int g = 2 + (st >> 2);
            key_insert_quad(acc[g], 4 * (st & 3), seq_prev + (st & 3), vmask, k0[g], k1[g], k2[g]);
            __builtin_amdgcn_sched_barrier(0);
        }
        MFMA_INIT(acc[2], ai, bi[2]);
        MFMA_INIT(acc[3], ai, bi[3]);
        __builtin_amdgcn_sched_barrier(0);
        load_frag(S, 8, tile0 + t + D);
#pragma unroll
        for (int st = 0; st < 8; ++st) {
            __builtin_amdgcn_sched_barrier(0);
            const u32x4 a = __builtin_bit_cast(u32x4, fr[S][st]);
            MFMA_F16(acc[2], a, bq[2][st]);
            MFMA_F16(acc[3], a, bq[3][st]);
            __builtin_amdgcn_sched_barrier(0);
            load_frag(S, st, tile0 + t + D);
            const int g = st >> 2;
            key_insert_quad(acc[g], 4 * (st & 3), seq_cur + (st & 3), vmask, k0[g], k1[g], k2[g]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int t = 0; t < tiles; t += D) {
        tile(t, std::integral_constant<int, 0>{});
        if constexpr (D > 1) tile(t + 1, std::integral_constant<int, 1>{});
        if constexpr (D > 2) tile(t + 2, std::integral_constant<int, 2>{});
        if constexpr (D > 3) tile(t + 3, std::integral_constant<int, 3>{});
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    int s = 0;
    for (int g = 0; g < 4; ++g) s += k0[g] + k1[g] + k2[g] + __float_as_int(acc[g][0]) + __float_as_int(acc[g][15]);
    for (int d = 0; d < D; ++d) for (int f = 0; f < 9; ++f) s += fr[d][f][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int D>
void run4(const unsigned* T, int* out, int tiles, int img_tiles) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k4<D>), dim3(256), dim3(256), 0, 0, T, img_tiles, tiles, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k4<D>), dim3(256), dim3(256), 0, 0, T, img_tiles, tiles, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double cyc = ms * 1e-3 * 2.0e9 / tiles;
    const double tf = 256.0 * 4 * tiles * 4096.0 * 256 / (ms * 1e-3) / 1e12;
    printf("q4 V=4 (direct, ring %d) image %4d tiles, tiles=%5d  %8.2f us  %6.0f cycles@2GHz per tile  algorithmic %.0f TF  err=%s\n", D, img_tiles, tiles, ms * 1e3, cyc, tf,
           hipGetErrorString(hipGetLastError()));
}

template <int V>
void run(const unsigned* T, int* out, int tiles) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    (void)hipFuncSetAttribute((const void*)k<V>, hipFuncAttributeMaxDynamicSharedMemorySize, kRing * kSlotBytes);
    hipLaunchKernelGGL((k<V>), dim3(256), dim3(256), kRing * kSlotBytes, 0, T, tiles, out);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) hipLaunchKernelGGL((k<V>), dim3(256), dim3(256), kRing * kSlotBytes, 0, T, tiles, out);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
    const double cyc = ms * 1e-3 * 2.0e9 / tiles;     // one wave per SIMD
    const double tf = 256.0 * 4 * tiles * 4096.0 * 256 / (ms * 1e-3) / 1e12;
    printf("q4 V=%d tiles=%5d  %8.2f us  %6.0f cycles@2GHz per tile (MFMA demand 1152; 128 q x 32 t)  algorithmic %.0f TF  err=%s\n", V, tiles, ms * 1e3, cyc, tf,
           hipGetErrorString(hipGetLastError()));
}

int main() {
    unsigned* T; int* out;
    const size_t bytes = (size_t)8 * 313 * 9216;                 // eight 10k-row fragment images
    (void)hipMalloc(&T, bytes); (void)hipMalloc(&out, 256 * 256 * 4);
    unsigned* h = (unsigned*)malloc(bytes);
    for (size_t i = 0; i < bytes / 4; ++i) h[i] = 0x38003800u + (((unsigned)i * 2654435761u) & 0x03FF03FFu);   // fp16 values in [0.5, 1)
    (void)hipMemcpy(T, h, bytes, hipMemcpyHostToDevice);
    for (int tiles : {24, 2400}) {
        run<0>(T, out, tiles); run<1>(T, out, tiles); run<2>(T, out, tiles); run<3>(T, out, tiles);
        run4<2>(T, out, tiles, 313); run4<3>(T, out, tiles, 313); run4<4>(T, out, tiles, 313); run4<3>(T, out, tiles, 8 * 313);
    }
    return 0;
}
