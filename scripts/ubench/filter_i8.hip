// Dev microbenchmark: tile loop of an exact-integer filter on v_mfma_i32_32x32x32_i8 — train fragments streamed
// L2 -> registers from a fragment-order image ([tile][4 k-steps + init][64 lanes][16 B]), NG 32-query groups per wave
// whose B fragments live in AGPRs, one init MFMA per tile shared by the groups, packed-key top-3 epilogue of half the
// groups interleaved into the chains of the other half.
//   NG   groups per wave (4, 8)          OCC  waves per SIMD (1: 256 workgroups, 2: 512)
//   RQ   accumulator registers per candidate record (4, 8, 16; 0: no epilogue)     LD  0: no fragment refills in the loop
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <type_traits>
#include <vector>
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int kTileBytes = 5 * 1024;
// OCC == 2: the A fragments live in AGPRs too (the loads target them directly): 128 VGPRs + 128 AGPRs per wave
#define MFMA_I8(acc, a, b) do { if constexpr (OCC == 2) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "a"(b)); else asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b)); } while (0)
#define MFMA_I8_C(acc, a, b, c) do { if constexpr (OCC == 2) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %3" : "=&v"(acc) : "a"(a), "a"(b), "v"(c)); else asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "a"(b), "v"(c)); } while (0)
#define MFMA_I8_INIT(acc, a, b) do { if constexpr (OCC == 2) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=v"(acc) : "a"(a), "a"(b)); else asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "a"(b)); } while (0)

template <int RQ>
__device__ __forceinline__ int rec_min(const i32x16& a, int r) {
    if constexpr (RQ == 4) return min(min(a[r], a[r + 1]), min(a[r + 2], a[r + 3]));
    else if constexpr (RQ == 8) return min(min(min(a[r], a[r + 1]), min(a[r + 2], a[r + 3])), min(min(a[r + 4], a[r + 5]), min(a[r + 6], a[r + 7])));
    else {
        int m = a[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) m = min(m, a[i]);
        return m;
    }
}
__device__ __forceinline__ int key_pack(int m, int seq) {
    int key;
    asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(key) : "v"(m), "s"(seq));
    return key;
}
__device__ __forceinline__ void key_put(int key, int& k0, int& k1, int& k2) {
    const int lo = min(key, k0);
    const int m1 = max(min(key, k1), min(max(key, k1), k0));
    k2 = max(min(key, k1), min(max(key, k1), k2));
    k1 = m1;
    k0 = lo;
}

template <int NG, int OCC, int RQ, int LD, int D>
__global__ __launch_bounds__(256, OCC) void kf(const unsigned* __restrict__ T, int img_tiles, int tiles, int* out, long long* clk) {
    const int lane = threadIdx.x & 63, h = lane >> 5;
    long long c0 = 0, w0 = 0;
    if (clk && threadIdx.x == 0) { c0 = clock64(); w0 = wall_clock64(); }
    u32x4 bq[NG][4], bi;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            u32x4 v;
            for (int e = 0; e < 4; ++e) v[e] = T[(threadIdx.x * 128 + g * 16 + s * 4 + e) & 0xFFFF];
            asm volatile("" : "=a"(bq[g][s]) : "0"(v));
        }
    {
        u32x4 v = {h ? 0x80808080u : 0x80808001u, 0x80808080u, 0x80808080u, 0x80808080u};
        asm volatile("" : "=a"(bi) : "0"(v));
    }
    asm volatile("s_nop 4");
    i32x16 acc[NG], cinit;
#pragma unroll
    for (int g = 0; g < NG; ++g)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[g][r] = 0x7fffff00;
    int k0[NG], k1[NG], k2[NG];
#pragma unroll
    for (int g = 0; g < NG; ++g) k0[g] = k1[g] = k2[g] = 0x7fffffff;
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)T, 0, img_tiles * kTileBytes, 0x00020000);
    const int voff = lane * 16;
    const int tile0 = (int)((blockIdx.x * 977u) % (unsigned)img_tiles);
    i32x4 fr[D][5];
    auto load_frag = [&](int slot, int f, int tile) {
        const int soff = __builtin_amdgcn_readfirstlane((tile % img_tiles) * kTileBytes + f * 1024);
        fr[slot][f] = __builtin_amdgcn_raw_buffer_load_b128(trs, voff, soff, 0);
    };
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int f = 0; f < 5; ++f) load_frag(d, f, tile0 + d);
    MFMA_I8_INIT(cinit, __builtin_bit_cast(u32x4, fr[0][4]), bi);
    asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");
    constexpr int P = NG / 2;
    constexpr int NREC = RQ ? 16 / RQ : 0;            // records per group and tile

    auto tile = [&](int t, auto slot_c) {
        constexpr int S = decltype(slot_c)::value;
        const int seq_prev = __builtin_amdgcn_readfirstlane(((t - 1) & 63) << 2), seq_cur = __builtin_amdgcn_readfirstlane((t & 63) << 2);
        // ---- phase A: chains of groups 0 .. P-1, epilogue of groups P .. NG-1 (previous tile): 4 P MFMAs, P * NREC records
        int e = 0;                                     // records dealt so far (compile-time after unrolling)
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const u32x4 a = __builtin_bit_cast(u32x4, fr[S][st]);
#pragma unroll
            for (int g = 0; g < P; ++g) {
                __builtin_amdgcn_sched_barrier(0);
                if (st == 0) MFMA_I8_C(acc[g], a, bq[g][st], cinit);
                else MFMA_I8(acc[g], a, bq[g][st]);
                if constexpr (RQ > 0) {
                    const int slot = st * P + g;                       // MFMA slot 0 .. 4P-1; records spread evenly
                    if ((slot * P * NREC) / (4 * P) != ((slot + 1) * P * NREC) / (4 * P)) {
                        const int rec = (slot * P * NREC) / (4 * P);
                        const int eg = P + rec / NREC, er = rec % NREC;
                        key_put(key_pack(rec_min<RQ>(acc[eg], RQ * er), seq_prev + er), k0[eg], k1[eg], k2[eg]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        (void)e;
        // ---- phase B
        if (LD) load_frag(S, 4, tile0 + t + D);
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            const u32x4 a = __builtin_bit_cast(u32x4, fr[S][st]);
#pragma unroll
            for (int g = 0; g < P; ++g) {
                __builtin_amdgcn_sched_barrier(0);
                if (st == 0) MFMA_I8_C(acc[P + g], a, bq[P + g][st], cinit);
                else MFMA_I8(acc[P + g], a, bq[P + g][st]);
                if (st == 0 && g == P - 1) {
                    // (the hazard "cinit overwritten while the last chain's first MFMA still reads it" does not exist: SrcC is read at issue)
                }
                if (st == 1 && g == 0) MFMA_I8_INIT(cinit, __builtin_bit_cast(u32x4, fr[(S + 1) % D][4]), bi);
                if constexpr (RQ > 0) {
                    const int slot = st * P + g;
                    if ((slot * P * NREC) / (4 * P) != ((slot + 1) * P * NREC) / (4 * P)) {
                        const int rec = (slot * P * NREC) / (4 * P);
                        const int eg = rec / NREC, er = rec % NREC;
                        key_put(key_pack(rec_min<RQ>(acc[eg], RQ * er), seq_cur + er), k0[eg], k1[eg], k2[eg]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if (LD) load_frag(S, st, tile0 + t + D);
        }
    };
    for (int t = 0; t < tiles; t += D) {
        tile(t, std::integral_constant<int, 0>{});
        if constexpr (D > 1) tile(t + 1, std::integral_constant<int, 1>{});
        if constexpr (D > 2) tile(t + 2, std::integral_constant<int, (D > 2 ? 2 : 0)>{});
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    int s = 0;
#pragma unroll
    for (int g = 0; g < NG; ++g) s += k0[g] + k1[g] + k2[g] + acc[g][0] + acc[g][15];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int f = 0; f < 5; ++f) s += fr[d][f][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (clk && threadIdx.x == 0) {
        clk[2 * blockIdx.x] = clock64() - c0;
        clk[2 * blockIdx.x + 1] = wall_clock64() - w0;
    }
}

template <int NG, int OCC, int RQ, int LD, int D>
void run(const unsigned* T, int* out, long long* clk, int tiles, int img_tiles, const char* data) {
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * OCC;
    hipLaunchKernelGGL((kf<NG, OCC, RQ, LD, D>), dim3(blocks), dim3(256), 0, 0, T, img_tiles, tiles, out, clk);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    const int reps = 8;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((kf<NG, OCC, RQ, LD, D>), dim3(blocks), dim3(256), 0, 0, T, img_tiles, tiles, out, clk);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    std::vector<long long> c(2 * blocks);
    (void)hipMemcpy(c.data(), clk, c.size() * 8, hipMemcpyDeviceToHost);
    double ghz = 0;
    for (int b = 0; b < blocks; ++b) ghz += (double)c[2 * b] / (double)c[2 * b + 1] * 0.1;
    ghz /= blocks;
    const double tile_groups = (double)blocks * 4 * tiles * NG;
    const double tops = tile_groups * 262144.0 / (ms * 1e-3) / 1e12;
    const double cyc = ms * 1e-3 * ghz * 1e9 / ((double)tiles * NG * OCC);        // shader cycles per tile-group per SIMD
    printf("i8 NG=%d waves/SIMD=%d RQ=%2d LD=%d ring=%d data=%-6s tiles=%5d  %8.2f us  %5.2f GHz  %6.1f cyc/tile-group/SIMD (128 = pipe)  %6.0f TOPS  %s\n", NG, OCC, RQ, LD, D,
           data, tiles, ms * 1e3, ghz, cyc, tops, hipGetErrorString(hipGetLastError()));
}

int main() {
    const int img_tiles = 313;
    unsigned* T; int* out; long long* clk;
    (void)hipMalloc(&T, 4 << 20); (void)hipMalloc(&out, 1024 * 256 * 4); (void)hipMalloc(&clk, 1024 * 16);
    std::vector<unsigned> hostT(1 << 20);
    for (int pass = 0; pass < 2; ++pass) {
        const char* data = pass == 0 ? "random" : "zero";
        unsigned s = 12345u;
        for (auto& v : hostT) { s = s * 1664525u + 1013904223u; v = pass == 0 ? (s ^ (s >> 13)) : 0u; }
        (void)hipMemcpy(T, hostT.data(), 4 << 20, hipMemcpyHostToDevice);
        // per SIMD the same number of tile-groups in every configuration: tiles * NG * OCC = 4800
        run<4, 1, 0, 0, 3>(T, out, clk, 1200, img_tiles, data);
        run<4, 1, 0, 1, 3>(T, out, clk, 1200, img_tiles, data);
        run<4, 1, 4, 1, 3>(T, out, clk, 1200, img_tiles, data);
        run<4, 1, 8, 1, 3>(T, out, clk, 1200, img_tiles, data);
        run<4, 1, 16, 1, 3>(T, out, clk, 1200, img_tiles, data);
        run<4, 2, 0, 1, 3>(T, out, clk, 600, img_tiles, data);
        run<4, 2, 4, 1, 3>(T, out, clk, 600, img_tiles, data);
        run<4, 2, 8, 1, 3>(T, out, clk, 600, img_tiles, data);
        run<4, 2, 4, 1, 2>(T, out, clk, 600, img_tiles, data);
        run<8, 1, 0, 1, 3>(T, out, clk, 600, img_tiles, data);
        run<8, 1, 4, 1, 3>(T, out, clk, 600, img_tiles, data);
        run<8, 1, 8, 1, 3>(T, out, clk, 600, img_tiles, data);
        run<8, 1, 16, 1, 3>(T, out, clk, 600, img_tiles, data);
    }
    return 0;
}
