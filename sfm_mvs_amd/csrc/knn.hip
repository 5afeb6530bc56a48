// Brute-force 128-D L2 2-NN (cv2.BFMatcher().knnMatch(des0, des1, k=2), sfm.py:259-260)
// for gfx950 — "certified filter + exact refine":
//
//   1. knn_prep_kernel    one pass over Q and T: the fp16 operand image, fp32 ||.||^2, exactness / range flags, the
//                         work-partition tables (knn_norms_kernel for the fp32-MFMA variant);
//      knn_split_images_kernel  the bf16 hi/mid planes, only when the flags ask for the split arithmetic.
//   2. knn_filter_*       s(q,t) = ||t||^2 + c - 2 q.t on the matrix pipe with the OPERANDS SWAPPED (A = train tile, B = query
//                         fragments resident in registers) so that a lane owns ONE query column and the running top-3
//                         per lane needs no cross-lane traffic.
//                         Default: knn_filter_q4_kernel — v_mfma_f32_32x32x16_{f16,bf16}, one fp16 product or three bf16
//                         hi/mid products, chosen on the device from the data; ONE wave per SIMD owning 4 x 32 queries
//                         whose fragments sit in accumulation registers; train tiles stream from a FRAGMENT-ORDER image
//                         straight into the MFMA's A registers (no LDS); c = the pair's largest ||q||^2, so that one
//                         init MFMA per tile serves all four groups; cost-weighted stream-K split over batch x row blocks
//                         x tiles.  (knn_filter_split2_kernel: the round-2 LDS-ring kernel, c = the query's own
//                         ||q||^2, kept as the `lds` variants; knn_filter_kernel: fp32 MFMA.)
//                         Every (workgroup segment, 64-tile substream, half-wave) triple is an independent "stream" that
//                         emits its 3 best records (a record = a quad of adjacent trains).
//   3. knn_refine_kernel  16 lanes per query: screens the rows of the records that can still matter against the fp16
//                         image, re-evaluates the survivors with the reference's direct-form float32 arithmetic (sub,
//                         mul, add — no FMA — in OpenCV's 2x4-lane accumulation order, then sqrtf), orders them by
//                         (dist, idx) and CERTIFIES the answer stream by stream against the lower bound of what each
//                         stream discarded.  Streams that cannot be certified are rescanned in the same kernel by the
//                         whole workgroup (fp16 screen of the stream's trains, exact evaluation of the few that pass).
//   4. ratio_scatter_*    Lowe survivors (counted by the refine kernel) written in ascending queryIdx order.
//
// filter = auto also carries the INTEGER body of the same filter kernel (v_mfma_i32_32x32x32_i8, half the MFMAs per distance):
//   * on u8-integer data (what cv2 SIFT emits) the scores are exact and the certificate is an integer one (refine_i8_body);
//   * on float data with compact support the rows are QUANTISED to 8 bits on one grid per pair (taken from a sample of the
//     rows, every residual norm measured by the prep pass), the body ranks the quantised distances exactly and
//     refine_q8_body selects / certifies with | ||q - t|| - s sqrt(D) | <= ||q - q^|| + ||t - t^|| before the same float32
//     re-evaluation; pairs whose grid turns out not to fit are repaired to the fp16 image (knn_split_images_kernel).
// Up to 8 equally shaped pairs share one set of launches (sfm_match_batch_l2_f32).
//
// The GEMM-form value is therefore never returned: indices and distances are bit-identical
// to the direct-form oracle (oracle/sfm_oracle.c: orc_knn2_l2_f32) for any finite input whose squared row norms are finite in
// float32 (|x| up to ~1e18: the filters' scores carry ||t||^2 + ||q||^2).
#include "common.h"
#include <cfloat>
#include <climits>
#include <cstdlib>
#include <type_traits>
#include <algorithm>
#include <mutex>
#include <vector>

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x3 __attribute__((ext_vector_type(3), aligned(4)));
typedef int i32x3 __attribute__((ext_vector_type(3), aligned(4)));

constexpr int kDim = 128;
constexpr int kTileT = 32;             // train rows per LDS tile (= MFMA M)


constexpr int kResidentWaves = 4096;   // 256 CUs x 16 waves (4 per SIMD at <= 128 VGPRs)
constexpr int kMaxSlots = 258;         // cap on filter blocks that may touch one query row block (all 256 CUs on one)
constexpr float kInf = __builtin_huge_valf();
constexpr int kSubTilesHost = 64;      // = kSubTiles (tiles per substream), needed by make_plan before its definition
constexpr int kI8SubTilesHost = 128;   // = kI8SubTiles (exact-integer body)
constexpr int kQ8SubTilesHost = 32;    // = kQ8SubTiles (the same body on 8-bit QUANTISED float data: shorter streams, see refine_q8_body)

// Work decomposition ("stream-K" over the flattened (query row block, train tile) unit space):
// block b owns units [units*b/G, units*(b+1)/G).  Every block gets the same number of units (+-1),
// so all 2x256 resident workgroups finish together whatever nq, nt are; a block that crosses a
// row-block boundary flushes its candidates and reloads the query fragment.
constexpr int kMaxBatch = 8;           // image pairs of equal shape matched by ONE set of launches (sfm_match_batch_l2_f32)

// Per-pair user buffers of a batched call (kernel argument, by value; entries >= B are unused).
struct BatchPtrs {
    const float* q[kMaxBatch];
    const float* t[kMaxBatch];
    int* idx[kMaxBatch];
    float* dist[kMaxBatch];
    int* stats[kMaxBatch];
    int* out_q[kMaxBatch];
    int* out_t[kMaxBatch];
    int* out_count[kMaxBatch];
    unsigned char* mask[kMaxBatch];
};

// Filter variants (the `filter` argument of the entry points, include/sfm_hip.h): results are bit-identical whichever runs.
constexpr int kFilterAuto = 0, kFilterF32 = 1, kFilterSplit = 2, kFilterLds = 3, kFilterLdsSplit = 4, kFilterHalf = 5,
              kFilterNoQuant = 6;   // as kFilterAuto, but float data never run QUANTISED on the integer body (exact u8 data still do)
constexpr int kFilterI8Plan = 100;     // internal: the plan of the exact-integer body that kFilterAuto carries beside its fp16 plan
constexpr int kFilterI8PlanQ8 = 101;   // internal: the same with the short substreams quantised data run with (sizes the slot arrays)

struct Plan {
    int split;         // 1: 16-bit MFMA filter (default), 0: fp32-MFMA filter
    int q4;            // 1: knn_filter_q4_kernel (register-streamed fragments, 4 query groups per wave), 0: the LDS-ring kernels
    int force_mode;    // -1: arithmetic chosen on the device from the data; kModeSplit: pinned to the bf16 split
    int qg;            // 32-query groups per wave (4: q4; 2: knn_filter_split2_kernel at 2 waves/SIMD)
    int nq_pad;        // query rows (of ONE pair) padded to whole row blocks
    int waves;         // waves per filter workgroup (4, 8 or 16); 16 waves are resident per CU either way
    int rows_per_block;
    int B;             // image pairs in the batch: the unit space is B x (row blocks of a pair) x tiles, one partition over all of it
    int n_rb1;         // query row blocks of one pair
    int n_rb;          // query row blocks of the whole batch = B * n_rb1 (pair b owns [b * n_rb1, (b + 1) * n_rb1))
    int tiles;         // train tiles of 32
    int64_t units;     // n_rb * tiles
    int G;             // filter blocks
    int smax;          // candidate slots reserved per row block (>= blocks touching it)
    int nsub;          // substreams (32 tiles each) per slot
    int seg_cost;      // partition weight of a segment in tile-steps (0: plain equal-units split)
    int sub_tiles;     // tiles per substream (64: the 16-bit filters' key holds 6 bits of tile + 2 of quad; 256: the i8 body's key holds 8 bits of tile)
    int i8;            // 1: kFilterAuto / kFilterNoQuant — the launch set also carries the exact-integer (i8 MFMA) body, chosen on the device for u8-integer data
    int q8;            // 1: kFilterAuto — float pairs whose sampled values have compact support are QUANTISED to 8 bits for that body
};

__host__ __device__ inline int64_t unit_begin(int64_t units, int G, int b) { return units * b / G; }

// block that owns unit u
__host__ __device__ inline int block_of_unit(int64_t units, int G, int64_t u) {
    int b = (int)(u * G / units);
    if (b >= G) b = G - 1;
    while (b + 1 < G && unit_begin(units, G, b + 1) <= u) ++b;
    while (b > 0 && unit_begin(units, G, b) > u) --b;
    return b;
}

// The partition as a function: block b owns units [part_begin(b), part_begin(b+1)).  seg_cost = 0 is unit_begin().
// seg_cost > 0 (pipelined split2 filter) equalises COST instead of units: every segment — the part of a block's range
// inside one query row block — pays seg_cost tile-steps for its prologue (query fragments, ring start) and flush, so
// a block that crosses a row-block boundary gets correspondingly fewer tiles (measured at 10k x 10k: crossing blocks
// ran 5 us = 4 tile-steps longer than the rest and set the kernel's duration).  In "cost space" a row block is
// tiles + seg_cost long and every block gets the same length (+-1); mapping back drops the seg_cost gaps.
struct Partition {
    int64_t units;
    int tiles, G, seg_cost;
    int64_t total;     // cost-space length of the whole problem: units + seg_cost per row block
};

__host__ __device__ inline Partition make_partition(int64_t units, int tiles, int G, int seg_cost) {
    Partition pt{units, tiles, G, seg_cost, 0};
    if (seg_cost > 0) {
        const int64_t n_rb = units / tiles;
        pt.total = units + (int64_t)seg_cost * n_rb;
    }
    return pt;
}

__host__ __device__ inline int64_t part_begin(const Partition& pt, int b) {
    if (pt.seg_cost == 0) return unit_begin(pt.units, pt.G, b);
    const int64_t y = pt.total * b / pt.G, period = pt.tiles + pt.seg_cost;
    const int64_t rb = y / period, rem = y - rb * period;
    const int64_t x = rb * pt.tiles + (rem < pt.tiles ? rem : pt.tiles);
    return x < pt.units ? x : pt.units;
}
constexpr int kSegCostTiles = 5;
// Tuning switches read from the environment exist in DEV builds only (make CXXFLAGS+=-DSFM_DEV_BUILD): a release
// library takes no override — what the parity sweeps ran on is what ships (sfm_build_id() names it).
#ifdef SFM_DEV_BUILD
int dev_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
constexpr int dev_env_int(const char*, int dflt) { return dflt; }
#endif
const int g_seg_cost = dev_env_int("SFM_KNN_SEGCOST", kSegCostTiles);
const int g_q8 = dev_env_int("SFM_KNN_Q8", 1);                  // 0 = kFilterAuto never quantises (= kFilterNoQuant)
const int g_seg_cost_q4 = dev_env_int("SFM_KNN_SEGCOST_Q4", 5); // q4 kernel: a segment's prologue in tile-steps

// One workgroup fills the partition tables the filter / refine kernels read: begin[G+1], and per query row block the
// first and last block that touches it.
constexpr int kPartLds = 1025;                                  // blocks whose table fits the LDS copy used for the searches
// Compact candidate streams (q4 filter): a workgroup's segment of L tiles in a row block is ceil(L / kSubTiles) substreams;
// the substreams of a row block are numbered densely in workgroup order — rb_last[n_rb + rb] = their count, wg_sbase[b] = the
// number of the first substream of block b's FIRST segment (a later segment of a block opens its row block: number 0) — so
// the records a query owns are a dense prefix of its slot array and nothing is written for substreams that do not exist.
__device__ inline void fill_partition_tables(const Partition pt, int n_rb, int64_t* __restrict__ begin, int* __restrict__ rb_first,
                                             int* __restrict__ rb_last, int* __restrict__ wg_sbase = nullptr, int sub_tiles = kSubTilesHost) {
    // the binary searches below are 9-10 DEPENDENT reads each: from global memory that was 6-7 us and made this one
    // workgroup the critical path of the whole prep launch, so they run on an LDS copy of the table
    __shared__ int64_t tab[kPartLds];
    const int G = pt.G, tiles = pt.tiles;
    const bool in_lds = G + 1 <= kPartLds;
    for (int b = threadIdx.x; b <= G; b += blockDim.x) {
        const int64_t v = b == G ? pt.units : part_begin(pt, b);
        begin[b] = v;
        if (in_lds) tab[b] = v;
    }
    __threadfence_block();
    __syncthreads();
    const int64_t* look = in_lds ? tab : begin;
    for (int rb = threadIdx.x; rb < n_rb; rb += blockDim.x) {
        const int64_t u0 = (int64_t)rb * tiles, u1 = (int64_t)(rb + 1) * tiles - 1;
        int lo = 0, hi = G - 1;                                // smallest b with begin[b+1] > u0
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (look[mid + 1] > u0) hi = mid; else lo = mid + 1;
        }
        rb_first[rb] = lo;
        lo = 0, hi = G - 1;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (look[mid + 1] > u1) hi = mid; else lo = mid + 1;
        }
        rb_last[rb] = lo;
        if (wg_sbase) {
            int acc = 0;
            for (int b = rb_first[rb]; b <= lo; ++b) {
                const int64_t s0 = look[b] > u0 ? look[b] : u0, s1 = look[b + 1] < u1 + 1 ? look[b + 1] : u1 + 1;
                if (look[b] >= u0) wg_sbase[b] = acc;              // (block b's range begins in this row block)
                acc += (int)((s1 - s0 + sub_tiles - 1) / sub_tiles);
            }
            rb_last[n_rb + rb] = acc;
        }
    }
}

Plan make_plan_uncached(int64_t nq, int64_t nt, int B, int filter);

// Planning walks the partition (O(G + row blocks)): keep the last few plans (a pipeline alternates between its full
// batch and the partial batch that ends a sequence).
Plan make_plan(int64_t nq, int64_t nt, int B, int filter) {
    struct Entry {
        int64_t nq = -1, nt = -1;
        int B = -1, mode = -1, seg = -1;
        unsigned long long used = 0;
        Plan plan;
    };
    static std::mutex mu;
    static Entry cache[4];
    static unsigned long long tick = 0;
    std::lock_guard<std::mutex> lk(mu);
    Entry* victim = &cache[0];
    for (Entry& e : cache) {
        if (e.nq == nq && e.nt == nt && e.B == B && e.mode == filter && e.seg == g_seg_cost) {
            e.used = ++tick;
            return e.plan;
        }
        if (e.used < victim->used) victim = &e;
    }
    victim->plan = make_plan_uncached(nq, nt, B, filter);
    victim->nq = nq; victim->nt = nt; victim->B = B; victim->mode = filter; victim->seg = g_seg_cost;
    victim->used = ++tick;
    return victim->plan;
}

Plan make_plan_uncached(int64_t nq, int64_t nt, int B, int filter) {
    Plan p;
    p.B = B;
    static const int env_w = dev_env_int("SFM_KNN_WAVES", 0);
    p.waves = (env_w == 4 || env_w == 8 || env_w == 16) ? env_w : 8;
    p.split = filter == kFilterF32 ? 0 : 1;
    p.q4 = (filter == kFilterLds || filter == kFilterLdsSplit || filter == kFilterF32) ? 0 : 1;
    p.force_mode = (filter == kFilterSplit || filter == kFilterLdsSplit) ? 2 /*kModeSplit*/ : -1;
    p.i8 = (filter == kFilterAuto || filter == kFilterNoQuant) ? 1 : 0;
    p.q8 = (filter == kFilterAuto && g_q8) ? 1 : 0;
    const bool i8plan = filter == kFilterI8Plan || filter == kFilterI8PlanQ8;   // the partition of the i8 body: 8 query groups per wave, 1024-query row blocks
    p.sub_tiles = i8plan ? (filter == kFilterI8PlanQ8 ? kQ8SubTilesHost : kI8SubTilesHost) : kSubTilesHost;   // (i8 plan: the SHORTEST substreams the launch set may choose on the device — sizes the slot arrays)
    p.qg = i8plan ? 8 : p.q4 ? 4 : p.split ? 2 : 1;
    if (p.q4) p.waves = 4;                     // one 4-wave workgroup per CU: one wave per SIMD, 512 registers each
    else if (p.split && !(env_w == 4 || env_w == 8 || env_w == 16)) p.waves = p.qg == 2 ? 4 : 16;   // split2: two 4-wave workgroups per CU (their barrier stalls interleave; ~3 % over one 8-wave group)
    p.rows_per_block = p.waves * 32 * p.qg;
    p.n_rb1 = (int)((nq + p.rows_per_block - 1) / p.rows_per_block);
    p.n_rb = B * p.n_rb1;
    p.nq_pad = p.n_rb1 * p.rows_per_block;
    if (p.i8) p.nq_pad = (int)((nq + 1023) / 1024) * 1024;      // (the query images serve the i8 body's 1024-query row blocks too)
    p.tiles = (int)((nt + kTileT - 1) / kTileT);
    p.units = (int64_t)p.n_rb * p.tiles;
    static const int env_res = dev_env_int("SFM_KNN_RESIDENT", 0);
    int64_t g = (env_res > 0 ? env_res : kResidentWaves / p.qg) / p.waves;
    if (i8plan) g = 256;                       // (one workgroup per CU, as the q4 bodies: they are bodies of one kernel)
    if (g > p.units) g = p.units;
    if (g > (int64_t)p.n_rb * (kMaxSlots - 2)) g = (int64_t)p.n_rb * (kMaxSlots - 2);
    if (g < 1) g = 1;
    p.G = (int)g;
    p.seg_cost = p.q4 ? g_seg_cost_q4 : (p.split && p.qg == 2) ? g_seg_cost : 0;
    {   // candidate slots per row block / substreams per slot from the actual partition
        std::vector<int64_t> begin((size_t)p.G + 1);
        const Partition pt = make_partition(p.units, p.tiles, p.G, p.seg_cost);
        for (int b = 0; b <= p.G; ++b) begin[b] = b == p.G ? p.units : part_begin(pt, b);
        int64_t maxseg = 1;
        for (int b = 0; b < p.G; ++b) maxseg = std::max(maxseg, begin[b + 1] - begin[b]);
        int touch = 1, b0 = 0, b1 = 0;
        for (int rb = 0; rb < p.n_rb; ++rb) {
            const int64_t u0 = (int64_t)rb * p.tiles, u1 = (int64_t)(rb + 1) * p.tiles - 1;
            while (b0 + 1 < p.G && begin[b0 + 1] <= u0) ++b0;
            while (b1 + 1 < p.G && begin[b1 + 1] <= u1) ++b1;
            touch = std::max(touch, b1 - b0 + 1);
        }
        p.smax = touch;
        if (maxseg > p.tiles) maxseg = p.tiles;
        p.nsub = (int)((maxseg + p.sub_tiles - 1) / p.sub_tiles);
        if (p.nsub < 1) p.nsub = 1;
    }
    return p;
}

// ---------------------------------------------------------------- norms
// ||t||^2 per train row (32 lanes per row, one float4 each) + one max per block (no atomics: a
// single hot atomicMax address costs ~12 ns per arrival).  Block 0 also zeroes the rescan counter,
// so the pipeline needs no memset launch.
constexpr int kNormBlocks = 256;

__global__ __launch_bounds__(256) void knn_norms_kernel(const float* __restrict__ T, int64_t ldt, int nt,
                                                        float* __restrict__ tn, float* __restrict__ bmax,
                                                        int* __restrict__ stats, int* __restrict__ zero, int nzero,
                                                        int64_t units, int tiles, int G, int seg_cost, int n_rb,
                                                        int64_t* __restrict__ wg_begin, int* __restrict__ rb_first,
                                                        int* __restrict__ rb_last) {
    __shared__ float wmax[4];
    if (blockIdx.x == gridDim.x - 1) {                     // the extra workgroup: partition tables, nothing else
        fill_partition_tables(make_partition(units, tiles, G, seg_cost), n_rb, wg_begin, rb_first, rb_last);
        return;
    }
    const int nblk = gridDim.x - 1;
    const int l = threadIdx.x & 31;
    float mx = 0.f;
    for (int row = blockIdx.x * 8 + (threadIdx.x >> 5); row < nt; row += nblk * 8) {
        const float4 v = *reinterpret_cast<const float4*>(T + (int64_t)row * ldt + 4 * l);
        float s = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) s += __shfl_xor(s, m, 64);
        if (l == 0) tn[row] = s;
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
        bmax[blockIdx.x] = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
        if (blockIdx.x == 0 && stats) stats[0] = 0;   // rescanned-query counter (refine kernel)
        if (blockIdx.x == 1 || nblk == 1)
            for (int i = 0; i < nzero; ++i) zero[i] = 0;   // Lowe-ratio survivor counters of the fused match call
    }
}

// ---------------------------------------------------------------- filter
__device__ __forceinline__ void top3_insert(float v, int id, float (&s)[3], int (&ix)[3]) {
    const bool lt0 = v < s[0], lt1 = v < s[1], lt2 = v < s[2];
    s[2] = lt1 ? s[1] : (lt2 ? v : s[2]);
    ix[2] = lt1 ? ix[1] : (lt2 ? id : ix[2]);
    s[1] = lt0 ? s[0] : (lt1 ? v : s[1]);
    ix[1] = lt0 ? ix[0] : (lt1 ? id : ix[1]);
    s[0] = lt0 ? v : s[0];
    ix[0] = lt0 ? id : ix[0];
}

// 32 queries per wave, 4 waves per workgroup, FOUR workgroups per CU (<=128 VGPRs): four waves
// per SIMD take turns on the matrix pipe, so one wave's top-3 epilogue / LDS round trips / barrier
// waits are covered by the other three.  The train tile goes HBM/L2 -> LDS with global_load_lds
// (no staging VGPRs, no ds_write pass); the LDS image is therefore lane-linear [32 rows][32 x 16 B]
// and bank conflicts are removed by XOR-swizzling the 16-byte chunk index with (row & 15) on the
// SOURCE address and on the fragment read (ds_read_b128: 16-lane groups hit 16 distinct slots).
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int kTileFloats = kTileT * kDim;            // 4096 floats = 16 KiB, linear
constexpr int kLdsFloats = 2 * kTileFloats + 2 * kTileT;

// Train tile -> LDS: 4 buffer_load_dwordx4 ... lds per wave (1 KiB each).  Addressing through a buffer
// descriptor keeps the per-lane state to ONE 32-bit offset register and makes rows past nt read as
// zeros in hardware (their ||t||^2 is +inf, so they never become candidates).
template <int W>
__device__ __forceinline__ void stage_tile(__amdgpu_buffer_rsrc_t trs, int row_bytes, int lane_off, const float* __restrict__ tn,
                                           int nt, int tile, float* __restrict__ tile_buf, float* __restrict__ tn_buf, int wave) {
    constexpr int PIECES = 16 / W;                                  // 1 KiB LDS-DMA pieces per wave per tile
    const int soff = tile * kTileT * row_bytes;
#pragma unroll
    for (int n = 0; n < PIECES; ++n) {
        float* dst = tile_buf + (wave * PIECES + n) * 256;          // wave-uniform, 1 KiB per instruction
        // row r = 2*(wave*PIECES + n) + h holds source chunk p ^ (r & 15) = (p ^ (r0 & 15)) ^ 2n at position p
        __builtin_amdgcn_raw_ptr_buffer_load_lds(trs, (lptr_t)dst, 16, (lane_off ^ (32 * n)) + 2 * n * row_bytes, soff, 0, 0);
    }
    if (threadIdx.x < kTileT) {
        const int row = tile * kTileT + threadIdx.x;
        tn_buf[threadIdx.x] = row < nt ? tn[row] : kInf;
    }
}

constexpr int kRecRows = 4;                       // a candidate record names this many adjacent train rows (one accumulator-register quad)
constexpr int kKeyBits = 8;                       // 6 bits tile-in-substream + 2 bits accumulator-register QUAD
constexpr int kKeyMask = (1 << kKeyBits) - 1;
constexpr int kSubTiles = 1 << (kKeyBits - 2);    // 64 tiles per substream
static_assert(kSubTiles == kSubTilesHost, "make_plan's copy");
constexpr int kKeyInf = 0x7F800000;               // +inf: larger than every finite non-negative score key

// decode 3 packed keys into (truncated score, train index) records
template <bool WIDE = false>
__device__ __forceinline__ void flush_keys(int k0, int k1, int k2, int sub_t0, int h, float* __restrict__ cs,
                                           int* __restrict__ ci) {
    const int ks[3] = {k0, k1, k2};
    if constexpr (WIDE) {
        // one 12-byte store per array (4-byte aligned vector types: global_store_dwordx3): a lane's record is private
        // and the lanes of a wave are 336+ B apart, so every store instruction is 64 separate requests — three dword
        // stores cost three times as much.  (Only where registers are plentiful: the 96-bit register tuples tipped the
        // split body, which sits at the 256-VGPR limit, into 49 spills.)
        f32x3 sc;
        i32x3 id;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int seq = ks[r] & kKeyMask;
            sc[r] = ks[r] == kKeyInf ? kInf : __int_as_float(ks[r] & ~kKeyMask);
            id[r] = ks[r] == kKeyInf ? -1 : (sub_t0 + (seq >> 2)) * kTileT + 8 * (seq & 3) + 4 * h;   // registers 4m..4m+3 = rows 8m + 4h + 0..3
        }
        *reinterpret_cast<f32x3*>(cs) = sc;
        *reinterpret_cast<i32x3*>(ci) = id;
    } else {
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int seq = ks[r] & kKeyMask;
            const bool empty = ks[r] == kKeyInf;
            cs[r] = empty ? kInf : __int_as_float(ks[r] & ~kKeyMask);
            ci[r] = empty ? -1 : (sub_t0 + (seq >> 2)) * kTileT + 8 * (seq & 3) + 4 * h;
        }
    }
}

// ABL != 0 are dev-only timing ablations (results are WRONG): bit0 skip the top-3 epilogue, bit1 skip the
// per-tile barrier, bit2 skip re-staging, bit3 constant accumulator init, bit4 s_setprio around the MFMA run.
template <int ABL, int W>
__global__ __launch_bounds__(64 * W, 4) void knn_filter_kernel(
    const float* __restrict__ Q, int64_t ldq, int nq, const float* __restrict__ T, int64_t ldt, int nt,
    const float* __restrict__ tn, int tiles, int64_t units, int smax, int nsub, float* __restrict__ cand_s,
    int* __restrict__ cand_i, long long* __restrict__ trace) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (trace && threadIdx.x == 0) {
        trace[4 * blockIdx.x + 0] = wall_clock64();
        trace[4 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg(0xF804);
        trace[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg(0xF814);
    }
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);   // provably uniform: no waterfall loops
    const int j = lane & 31;
    const int h = lane >> 5;
    const int hm = h ^ (j & 15);                 // swizzled position of chunk (2c + h) is (2c) ^ hm
    const int G = gridDim.x;
    const int64_t u_end = unit_begin(units, G, blockIdx.x + 1);
    int64_t u = unit_begin(units, G, blockIdx.x);
    float* const tnb = smem + 2 * kTileFloats;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;   // LDS byte offset of smem
    const int row_bytes = (int)ldt * 4;
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)T, 0, nt * row_bytes, 0x00020000);
    // staging offset of this lane for n = 0: row r0 = wave*8 + h, byte position of chunk (p ^ (r0 & 15))
    constexpr int PIECES = 16 / W;
    const int lane_off = (wave * 2 * PIECES + h) * row_bytes + (((lane & 31) ^ ((wave * 2 * PIECES + h) & 15)) << 4);

    while (u < u_end) {
        const int rb = (int)(u / tiles);
        const int t_begin = (int)(u - (int64_t)rb * tiles);
        const int t_end = (int)min((int64_t)tiles, t_begin + (u_end - u));
        const int slot = blockIdx.x - block_of_unit(units, G, (int64_t)rb * tiles);
        const int qrow = rb * (W * 32) + wave * 32 + j;
        const bool qok = qrow < nq;

        __syncthreads();   // every wave is done with both buffers of the previous segment
        stage_tile<W>(trs, row_bytes, lane_off, tn, nt, t_begin, smem, tnb, wave);

        float bq[64];
        {
            const float* src = Q + (int64_t)(qok ? qrow : 0) * ldq + 4 * h;
#pragma unroll
            for (int c = 0; c < 16; ++c) {
                float4 v = *reinterpret_cast<const float4*>(src + 8 * c);
                if (!qok) v = make_float4(0.f, 0.f, 0.f, 0.f);
                bq[4 * c + 0] = -2.f * v.x;
                bq[4 * c + 1] = -2.f * v.y;
                bq[4 * c + 2] = -2.f * v.z;
                bq[4 * c + 3] = -2.f * v.w;
            }
        }
        // ||q||^2 of this lane's query (its own 64 k's + the other half-wave's), folded into the score by one
        // extra MFMA step per tile (A = 1 on the h=0 half, B = ||q||^2 there) so that s ~ d^2 >= 0.
        float qn = 0.f;
#pragma unroll
        for (int c = 0; c < 64; ++c) qn = fmaf(0.25f * bq[c], bq[c], qn);
        qn += __shfl_xor(qn, 32, 64);
        const float aug_a = h == 0 ? 1.f : 0.f;
        const float aug_b = h == 0 ? qn : 0.f;

        // Running top-3 as PACKED KEYS, one per QUAD of adjacent trains: (min(score_r .. score_r+3) bits & ~255) |
        // (tile_in_substream << 2 | r/4).  Scores are non-negative up to rounding noise, so signed-integer order ==
        // float order and one insertion is 3 v_min_f32 + v_and_or + v_min_i32 + 2 v_med3_i32 (7 VALU per four scores)
        // instead of 13 compare/select ops per score; the 8 dropped mantissa bits (2^-15 relative) are covered by the
        // refine kernel's slack.  A substream is 64 tiles.
        int k0 = kKeyInf, k1 = kKeyInf, k2 = kKeyInf;
        int sub = 0, sub_t0 = t_begin;
        const int64_t obase = ((int64_t)qrow * (2 * smax * nsub) + (int64_t)slot * nsub * 2 + h) * 3;
        __syncthreads();   // tile t_begin landed (the barrier's release waits for the LDS-DMA: vmcnt(0))

        for (int t = t_begin; t < t_end; ++t) {
            if (t - sub_t0 == kSubTiles) {
                if (qok) flush_keys(k0, k1, k2, sub_t0, h, cand_s + obase + 6 * sub, cand_i + obase + 6 * sub);
                k0 = k1 = k2 = kKeyInf;
                ++sub;
                sub_t0 = t;
            }
            const int cur = (t - t_begin) & 1;
            if (t + 1 < t_end && !(ABL & 4)) stage_tile<W>(trs, row_bytes, lane_off, tn, nt, t + 1, smem + (cur ^ 1) * kTileFloats, tnb + (cur ^ 1) * kTileT, wave);

            f32x16 acc;
            if (ABL & 8) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 1.f;
            } else {
                const float* tnp = tnb + cur * kTileT + 4 * h;
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const float4 v = *reinterpret_cast<const float4*>(tnp + 8 * b);
                    acc[4 * b + 0] = v.x;
                    acc[4 * b + 1] = v.y;
                    acc[4 * b + 2] = v.z;
                    acc[4 * b + 3] = v.w;
                }
            }
            // A fragments: hand-placed ds_read_b128 two chunks (8 MFMAs = 512 pipe cycles) ahead of use.
            // hipcc sinks such reads next to their consumer (and then blocks the in-order wave on the LDS
            // round trip every 4 MFMAs), so the reads and their counted waits are inline asm:
            //   issue r(c+2); s_waitcnt lgkmcnt(2) => r(c) has landed, r(c+1), r(c+2) stay in flight.
            const unsigned abase = lds0 + (unsigned)(((ABL & 4) ? 0 : cur) * kTileFloats + j * kDim) * 4u + ((unsigned)hm << 4);
            f32x4 af[3];
            if (ABL & 16) __builtin_amdgcn_s_setprio(1);
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
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c % 3][0], bq[4 * c + 0], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c % 3][1], bq[4 * c + 1], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c % 3][2], bq[4 * c + 2], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(af[c % 3][3], bq[4 * c + 3], acc, 0, 0, 0);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(aug_a, aug_b, acc, 0, 0, 0);
            if (ABL & 16) __builtin_amdgcn_s_setprio(0);
            const int seq0 = (t - sub_t0) << 2;
            if (ABL & 1) {
                k0 = min(k0, __float_as_int(acc[0]) + __float_as_int(acc[15]));
            } else
#pragma unroll
            for (int r = 0; r < 16; r += 4) {                              // one key per QUAD of adjacent trains (see key_insert4)
                const int key = (min(min(__float_as_int(acc[r]), __float_as_int(acc[r + 1])), min(__float_as_int(acc[r + 2]), __float_as_int(acc[r + 3]))) & ~kKeyMask) | (seq0 + (r >> 2));
                const int lo = min(key, k0), hi = max(key, k0);            // (lo, hi) = sorted (key, k0)
                const int m1 = max(min(key, k1), min(max(key, k1), k0));   // med3(key, k0, k1)
                k2 = max(min(key, k1), min(max(key, k1), k2));             // med3(key, k1, k2)
                k1 = m1;
                k0 = lo;
                (void)hi;
            }
            if (!(ABL & 2)) __syncthreads();
        }

        if (qok) {
            flush_keys(k0, k1, k2, sub_t0, h, cand_s + obase + 6 * sub, cand_i + obase + 6 * sub);
            for (int e = sub + 1; e < nsub; ++e)       // unused substreams of this slot: empty
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    cand_s[obase + 6 * e + r] = kInf;
                    cand_i[obase + 6 * e + r] = -1;
                }
        }
        u += t_end - t_begin;
    }
    if (trace && threadIdx.x == 0) trace[4 * blockIdx.x + 1] = wall_clock64();
}

// ---------------------------------------------------------------- split-bf16 filter
// The filter only has to RANK (the refine kernel re-evaluates and certifies), so it need not run on
// the 157 TFLOP/s fp32 MFMA: every float is split exactly into hi + mid + delta, hi = bf16(x),
// mid = bf16(x - hi), |delta| <= 2^-16 |x|, and q.t ~ qh.th + qh.tm + qm.th on
// v_mfma_f32_32x32x16_bf16 (2.5 PFLOP/s; products of bf16 are exact in the fp32 accumulator).
// 25 MFMAs x 32 cycles per 32x32 tile instead of 65 x 64.  Neglected terms are bounded by
// 3.05 * 2^-16 |q||t|; the refine kernel's slack accounts for it (kEpsSplit).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// c + sum of eight fp16 products (v_dot2_f32_f16 x 4: exact products, fp32 accumulation)
__device__ __forceinline__ float dot8_f16(const uint4& a, const uint4& b, float c) {
    c = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a.x), __builtin_bit_cast(f16x2, b.x), c, false);
    c = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a.y), __builtin_bit_cast(f16x2, b.y), c, false);
    c = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a.z), __builtin_bit_cast(f16x2, b.z), c, false);
    c = __builtin_amdgcn_fdot2(__builtin_bit_cast(f16x2, a.w), __builtin_bit_cast(f16x2, b.w), c, false);
    return c;
}

__device__ __forceinline__ unsigned bf16_rn_bits(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}

// Filter arithmetic modes, decided ON THE DEVICE from what knn_prep_kernel saw (no host round trip):
//   kModeHalfExact  every value (Q pre-scaled by -2) is exactly a normal fp16 or zero (real SIFT descriptors are
//                   integers 0..255): ONE v_mfma_f32_32x32x16_f16 product, dot products exact.
//   kModeHalf       values within fp16 range and ||t||max >= 1/2: one fp16 product; rounding each operand to 11 bits
//                   perturbs the score by <= 2^-11 (|q|+|t|)^2 (+ an absolute term for values below the fp16 normal
//                   range, which the matrix pipe may flush) — the refine kernel's slack (kEpsHalf*) covers it.
//   kModeSplit      anything else (huge / tiny magnitudes): bf16 hi+mid split, three products, full fp32 range.
constexpr int kModeHalfExact = 0, kModeHalf = 1, kModeSplit = 2;
constexpr int kFlagHalfInexact = 2, kFlagRangeBad = 4, kFlagNotU8 = 8, kFlagSomeU8 = 16;   // (NotU8: some value is not an integer 0 .. 255; SomeU8: some chunk of a real row exists in the byte image only)
constexpr int kFlagQ8 = 32;            // the pair's rows were QUANTISED to 8 bits (float data on the integer body): byte image + residual norms, no fp16 image

// Lane exchanges without an address register: lane ^ M by DPP quad_perm (M = 1, 2) or ds_swizzle bit mode
// (M = 4, 8, 16); __shfl_xor compiles to ds_bpermute and keeps one address VGPR alive per distinct pattern.
template <int M>
__device__ __forceinline__ int lane_xor(int v) {
    if constexpr (M == 1) return __builtin_amdgcn_update_dpp(0, v, 0xB1 /*quad_perm [1,0,3,2]*/, 0xF, 0xF, false);
    else if constexpr (M == 2) return __builtin_amdgcn_update_dpp(0, v, 0x4E /*quad_perm [2,3,0,1]*/, 0xF, 0xF, false);
    else if constexpr (M == 4 || M == 8 || M == 16) return __builtin_amdgcn_ds_swizzle(v, 0x1F | (M << 10));
    else return __shfl_xor(v, M, 64);
}
template <int M>
__device__ __forceinline__ float lane_xor(float v) { return __int_as_float(lane_xor<M>(__float_as_int(v))); }
// Wave-uniform mode from the per-block flags and per-block max ||t||^2 (256 entries each).
__device__ __forceinline__ int knn_filter_mode(const int* __restrict__ flags, const float* __restrict__ bmax, int lane,
                                               float* tmax_out = nullptr) {
    const int fl = flags[lane] | flags[lane + 64] | flags[lane + 128] | flags[lane + 192];
    float tmax = fmaxf(fmaxf(bmax[lane], bmax[lane + 64]), fmaxf(bmax[lane + 128], bmax[lane + 192]));
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) tmax = fmaxf(tmax, __shfl_xor(tmax, m, 64));
    if (tmax_out) *tmax_out = tmax;
    if (__any((fl & kFlagRangeBad) != 0)) return kModeSplit;
    if (!__any((fl & kFlagHalfInexact) != 0)) return kModeHalfExact;
    return tmax >= 0.25f ? kModeHalf : kModeSplit;
}

// The arithmetic mode of a BATCH of pairs (one launch, one body): the most general mode any of its pairs needs — the
// modes are nested (fp16-exact data are fp16-representable data are split-representable data), and the refine kernel
// prices its slack with the same batch mode, so every pair is certified against the arithmetic that actually ran.
constexpr int kMinfoPairMode = 0, kMinfoBatchMode = kMaxBatch, kMinfoTmax = 16, kMinfoTerr = 24, kMinfoQmax = 32, kMinfoBase = 40, kMinfoI8 = 48,
              kMinfoQ8 = 49,      // 1: the integer body runs on QUANTISED data (some pair of the batch is float data): refine_q8_body certifies
              kMinfoSub8 = 50,    // tiles per substream of the integer body for this launch set (written by the prep launch's partition workgroup)
              kMinfoTicket = 51,  // knn_split_images_kernel's repair: arrival ticket — the workgroup that draws the LAST one reduces the rewritten words
                                  // (zeroed by the prep launch)
              kMinfoQ8S = 56, kMinfoQ8Lo = 64,   // per pair: the quantisation grid x ~ lo + s k, k = 0 .. 255 (float bits; written by the prep launch)
              kMinfoWords = 72;   // written by knn_split_images_kernel (block 0); kMinfoI8: 1 = the exact-integer body runs, kMinfoBase: its per-pair score offset
__device__ __forceinline__ int knn_batch_mode(const int* __restrict__ flags, const float* __restrict__ bmax, int n_pairs, int lane) {
    int mode = kModeHalfExact;
    for (int b = 0; b < n_pairs; ++b) mode = max(mode, knn_filter_mode(flags + b * kNormBlocks, bmax + b * kNormBlocks, lane));
    return mode;
}

// ---- fragment-order images of the q4 filter
constexpr int kFragBytes = 1024;                            // one MFMA operand fragment of a 32-row tile: 64 lanes x 16 B
constexpr int kInitFragBytes = 512;                         // the accumulator-init fragment: 64 lanes x 8 B (K = 8 bf16 MFMA)
constexpr int kTileFragBytes = 8 * kFragBytes + kInitFragBytes;   // 8 k-steps of 16, then the init fragment

// The accumulator init ||t||^2 + ||q||^2 as ONE v_mfma_f32_32x32x8_bf16 (K = 8: 16 pipe cycles, half a product MFMA): a
// float32 x >= 0 is split EXACTLY into three bf16 pieces hi + mid + lo (8 + 8 + 8 significant bits; every residual is exact
// in float32); the k-slots 0 .. 3 (lanes with h = 0) and 4 .. 7 (h = 1) carry
//     train side  {hi, mid, lo, 1 | 1, 1, 0, 0}        query side  {1, 1, 1, hi | mid, lo, 0, 0}
// so the product sums the six pieces in the fp32 accumulator.  +inf (padded train rows) is {inf, 0, 0}: no inf - inf.
// Pieces below bf16's normal range (x < 2^-110) may be flushed by the matrix pipe: an absolute error < 2^-120, far below
// the certificate's slack (which is relative to (|q| + |t|max)^2 >= 2^-28 in every mode).
__device__ __forceinline__ uint2 frag_init_operand(float x, bool query_side, int h) {
    unsigned hi = bf16_rn_bits(x), mid = 0, lo = 0;
    if (x < kInf) {
        const float r1 = x - __uint_as_float(hi << 16);
        mid = bf16_rn_bits(r1);
        lo = bf16_rn_bits(r1 - __uint_as_float(mid << 16));
    }
    const unsigned one = 0x3F80u;
    if (query_side) return h == 0 ? make_uint2(one | (one << 16), one | (hi << 16)) : make_uint2(mid | (lo << 16), 0u);
    return h == 0 ? make_uint2(hi | (mid << 16), lo | (one << 16)) : make_uint2(one | (one << 16), 0u);
}

// ---- exact-integer (i8 MFMA) body: images and scores
// Real SIFT descriptors are integers 0 .. 255 (sfm.py:246-252 -> :259-260).  For such data the filter runs on
// v_mfma_i32_32x32x32_i8 — half the MFMAs per distance of the fp16 product — with i32 accumulation, i.e. EXACT scores:
//     stored bytes   train  a = t - 128  (= t ^ 0x80)      query  b = 127 - q  (= q ^ 0x7F),   both in [-128, 127]
//     q - t = -(a + b + 1)   =>   d^2 = sum (a + b + 1)^2 = w_t + c_q - 128 + 2 sum a b,   w_t = sum (a + 1)^2 = |t - 127|^2,
//                                                                                       c_q = sum (b + 1)^2 = |q - 128|^2
// The factor 2 cannot ride on an i8 operand, so the accumulator holds HALF the score:  acc = C_t + sum a b  with
// C_t = floor(w_t / 2) - base  (base: the pair's mid-range of floor(w_t / 2), so that C_t fits the init product below), and
//     d^2 = c_q - 128 + (w_t & 1) + 2 (acc + base):   the filter ranks by acc, which is the exact order up to the parity bit.
// The refine kernel's certificate is therefore an INTEGER one with slack 1 (+ 2 for float32 square roots that collide):
// no epsilon, no error budget, no assumption about the matrix pipe's rounding.
// Init product (one MFMA per tile, K = 32): query side {1, -128 x 31}, train side {C & 127, digits d_k} with
// sum d_k = -(C >> 7), d_k in [-128, 127]:  C in [-503 936, 508 031].  Pairs whose floor(w_t / 2) spread exceeds that range
// (possible only when all-127 and all-0 / all-255 rows meet in one image) fall back to the fp16 body, as do non-u8 data.
constexpr int kI8Groups = 8;                                // 32-query groups per wave
constexpr int kI8Rows = 4 * kI8Groups * 32;                 // 1024 queries per workgroup (row block of the i8 partition)
constexpr int kI8SubTiles = 128;                            // tiles per substream: the key's low 8 bits are (tile inside it) << 1 | register half
constexpr int kI8TileBytes = 5 * 1024;                      // train image per 32 rows: 4 k-steps of 32 ([64 lanes][16 B]) + the init fragment
constexpr int kI8QTileBytes = 4 * 1024;                     // query image per 32 rows
constexpr int kI8CMax = 508031, kI8CMin = -503936;          // range of the init product
static_assert(kI8SubTiles == kI8SubTilesHost, "make_plan's copy");
constexpr int kKeyEmptyI = 0x7FFFFF00;                      // i8 keys >= this are empty slots / the first tile's pretend "previous tile"

// ---- 8-bit QUANTISED float data on the integer body ("q8")
// Float descriptors that are not u8 integers can still be FILTERED on v_mfma_i32_32x32x32_i8: x ~ lo + s k, k = 0 .. 255, one
// grid (lo, s) per image pair.  With q^ = lo + s k_q, t^ = lo + s k_t the integer body returns D = sum (k_q - k_t)^2 EXACTLY, and
//     | ||q - t|| - s sqrt(D) |  <=  ||q - q^|| + ||t - t^||                                   (triangle inequality)
// whatever rounding, clipping or grid produced the bytes: the prep pass MEASURES the residual norms per row, the refine
// kernel (refine_q8_body) selects / certifies with them and re-evaluates the survivors in the reference's float32 arithmetic.
// The grid is therefore a matter of speed only.  It comes from a SAMPLE — sixteen rows of Q and sixteen of T, evenly spaced;
// every workgroup of the pair reads the same 16 KiB (L2 hits) and derives the same grid: no extra pass over the data, no
// grid-wide reduction.  Values outside the sampled range saturate (their error is part of the measured residual).
constexpr int kQ8SubTiles = 32;                             // tiles per substream when the launch set runs on quantised data
static_assert(kQ8SubTiles == kQ8SubTilesHost, "make_plan's copy");
struct Q8Grid {
    float lo, s, inv;
    int kind;        // 0: every sampled value is a u8 integer (exact-integer path, decided per chunk); 1: quantise; 2: not quantisable (16-bit bodies)
};
__device__ __forceinline__ Q8Grid q8_sample_grid(const float* __restrict__ Q, int64_t ldq, int nq, const float* __restrict__ T, int64_t ldt, int nt,
                                                 float* __restrict__ red /*LDS [20]*/) {
    const int sr = threadIdx.x >> 4, c = threadIdx.x & 15;             // 256 threads: 16 rows x 16 chunks of 8 floats, of Q and of T
    const float* qs = Q + (int64_t)(((int64_t)sr * nq) >> 4) * ldq + 8 * c;
    const float* ts = T + (int64_t)(((int64_t)sr * nt) >> 4) * ldt + 8 * c;
    const float4 v0 = *reinterpret_cast<const float4*>(qs), v1 = *reinterpret_cast<const float4*>(qs + 4);
    const float4 v2 = *reinterpret_cast<const float4*>(ts), v3 = *reinterpret_cast<const float4*>(ts + 4);
    const float in[16] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w, v2.x, v2.y, v2.z, v2.w, v3.x, v3.y, v3.z, v3.w};
    float mn = kInf, mx = -kInf, sm = 0.f, sq = 0.f;
    bool bad = false, nonu8 = false;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        mn = fminf(mn, in[e]);
        mx = fmaxf(mx, in[e]);
        sm += in[e];
        sq = fmaf(in[e], in[e], sq);
        bad = bad || !(fabsf(in[e]) < kInf);                            // NaN / inf
        nonu8 = nonu8 || !(in[e] >= 0.f && in[e] <= 255.f && in[e] == floorf(in[e]));
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        mn = fminf(mn, __shfl_xor(mn, m, 64));
        mx = fmaxf(mx, __shfl_xor(mx, m, 64));
        sm += __shfl_xor(sm, m, 64);
        sq += __shfl_xor(sq, m, 64);
    }
    const int fl = (__any(bad) ? 1 : 0) | (__any(nonu8) ? 2 : 0);
    __syncthreads();                                                    // (a caller may loop: the previous round's reads are done)
    if ((threadIdx.x & 63) == 0) {
        red[5 * (threadIdx.x >> 6)] = mn;
        red[5 * (threadIdx.x >> 6) + 1] = mx;
        red[5 * (threadIdx.x >> 6) + 2] = __int_as_float(fl);
        red[5 * (threadIdx.x >> 6) + 3] = sm;
        red[5 * (threadIdx.x >> 6) + 4] = sq;
    }
    __syncthreads();
    int flw = 0;
    mn = red[0]; mx = red[1]; sm = 0.f; sq = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        mn = fminf(mn, red[5 * w]);
        mx = fmaxf(mx, red[5 * w + 1]);
        flw |= __float_as_int(red[5 * w + 2]);
        sm += red[5 * w + 3];
        sq += red[5 * w + 4];
    }
    Q8Grid g{0.f, 1.f, 1.f, 0};
    if (flw & 1) { g.kind = 2; return g; }
    if (!(flw & 2)) return g;                                           // u8 integers as far as the sample goes
    const float s = (mx - mn) * (1.f / 255.f), mabs = fmaxf(fabsf(mn), fabsf(mx));
    // (a range below the float32 spacing of the values, or absurd magnitudes: the 16-bit bodies' business)
    if (!(s >= 1e-12f && s <= 1e30f && s >= 1.52587890625e-05f * mabs)) { g.kind = 2; return g; }
    // COMPACT SUPPORT only: 256 levels over the range are worth it when the range is a few standard deviations (uniform data:
    // 3.5; a Gaussian's 4096-value sample: 7 and its tails would clip).  Heavy-tailed data stay with the 16-bit bodies — a
    // matter of speed; a wrong guess is caught by the measured residuals (knn_split_images_kernel) and repaired.
    {
        const float mean = sm * (1.f / 4096.f), var = fmaxf(sq * (1.f / 4096.f) - mean * mean, 0.f);
        if (!((mx - mn) * (mx - mn) <= 25.f * var)) { g.kind = 2; return g; }
    }
    g.lo = mn; g.s = s; g.inv = 1.f / s; g.kind = 1;
    return g;
}

// The fp16 image of one 8-element chunk of a row (sc = -2 for query rows): the packed values, the chunk's share of
// ||fp16(row) - row||^2 (the certificate's operand-rounding term) and the exactness / range flags.  Shared by the prep pass and
// by the repair of pairs that were quantised in vain (knn_split_images_kernel).
__device__ __forceinline__ float fp16_chunk(const float (&in)[8], float sc, unsigned (&fw)[4], unsigned& flags) {
    fw[0] = fw[1] = fw[2] = fw[3] = 0u;
    float err2 = 0.f;
    // The per-element tests — range (also NaN / inf), fp16-exactness, below fp16's normal range — are folded over the
    // lane's eight elements on bit patterns (the kernel is bound by its vector-ALU instruction count, not by the bytes):
    //   amax = max |e| bits;   umin = min (|e| bits - 1)  (0 wraps to 0xFFFFFFFF: "nonzero and below 2^-14" is ONE unsigned
    //   compare);   dor = OR of the residuals' bits (any bit but the sign: inexact)
    unsigned amax = 0u, umin = 0xFFFFFFFFu, dor = 0u;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float ev = sc * in[e];
        const _Float16 hv = (_Float16)ev;                            // round to nearest even
        fw[e >> 1] |= (unsigned)__builtin_bit_cast(unsigned short, hv) << (16 * (e & 1));
        const float dv = ev - (float)hv;
        err2 = fmaf(dv, dv, err2);
        const unsigned ab = __float_as_uint(ev) & 0x7FFFFFFFu;
        amax = max(amax, ab);
        umin = min(umin, ab - 1u);
        dor |= __float_as_uint(dv);
    }
    if (amax > 0x476A6000u /*60000.f*/) flags |= kFlagRangeBad;      // (NaN and inf patterns are larger still)
    const bool has_sub = umin < 0x38800000u - 1u;                     // some element is nonzero and below 2^-14
    if ((dor & 0x7FFFFFFFu) != 0u || has_sub) flags |= kFlagHalfInexact;
    if (has_sub) {
        // below fp16's normal range the matrix pipe may flush the operand to zero: the whole element is the error then
        // (|e - hv| <= |e| holds for the rounded subnormal too, so this bounds both behaviours).  Rare: redo the sum.
        err2 = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float ev = sc * in[e];
            const float dv = fabsf(ev) < 6.103515625e-5f ? ev : ev - (float)__builtin_bit_cast(_Float16, (unsigned short)(fw[e >> 1] >> (16 * (e & 1))));
            err2 = fmaf(dv, dv, err2);
        }
    }
    return err2;
}

// One pass over Q and T: rows → the fp16 image (Q pre-scaled by -2, exact), fp32 squared norms, per-block max of
// ||t||^2 and exactness / range flags, zero the rescan counter.  Rows >= n of the padded images are zero-filled.
// Image layout: [3][n_pad][128] 16-bit: bf16 hi, bf16 mid, fp16; the two bf16 planes are only needed by the split
// arithmetic (values outside fp16's range) and are written by knn_split_images_kernel, which runs when the flags say so:
// the common case moves 15 MB per 10k x 10k pair instead of 25.
// Batched: grid = (blocks + 1, B); column b works on pair b (its workspace arrays sit at b * stride).
constexpr int kPrepThreads = 256;      // one wave per SIMD, 56 registers: a prep workgroup of the NEXT launch set fits beside the q4 filter's waves
__global__ __launch_bounds__(kPrepThreads) void knn_prep_kernel(BatchPtrs P, int64_t ldq, int nq, int nq_pad,
                                                       int64_t ldt, int nt, int nt_pad,
                                                       unsigned short* __restrict__ qsplit, float* __restrict__ qn,
                                                       unsigned short* __restrict__ tsplit, float* __restrict__ tn,
                                                       float* __restrict__ bmax, int* __restrict__ midflag,
                                                       float* __restrict__ qerr /*[s_qn] per pair*/, float* __restrict__ bmaxerr /*[kNormBlocks] per pair*/,
                                                       float* __restrict__ bqmax /*[kNormBlocks] per pair: per-block max of ||q||^2*/,
                                                       int64_t s_qsplit, int64_t s_tsplit, int64_t s_qn, int64_t s_tn,
                                                       unsigned char* __restrict__ qfrag /*null: row-major images only (LDS-ring filter)*/,
                                                       unsigned char* __restrict__ tfrag, int64_t s_qfrag, int64_t s_tfrag,
                                                       int* __restrict__ zero, int nzero,
                                                       int64_t units, int tiles, int G, int seg_cost, int n_rb,
                                                       int64_t* __restrict__ wg_begin, int* __restrict__ rb_first,
                                                       int* __restrict__ rb_last, int* __restrict__ wg_sbase,
                                                       // the exact-integer body (null / 0 when not planned): byte images, integer norms, its partition
                                                       unsigned char* __restrict__ qi8, unsigned char* __restrict__ ti8, int64_t s_qi8, int64_t s_ti8,
                                                       int* __restrict__ wq /*[s_qn] per pair: |q - 128|^2*/, int* __restrict__ wt /*[s_tn] per pair: |t - 127|^2*/,
                                                       int* __restrict__ bwmin /*[kNormBlocks] per pair*/, int* __restrict__ bwmax,
                                                       unsigned short* __restrict__ rmq /*[s_qn] per pair: chunks of a row that exist in the byte image only*/,
                                                       unsigned short* __restrict__ rmt /*[s_tn]*/,
                                                       int64_t units8, int G8, int n_rb8, int64_t* __restrict__ wg_begin8,
                                                       int* __restrict__ rb_first8, int* __restrict__ rb_last8, int* __restrict__ wg_sbase8,
                                                       int q8 /*1: float pairs are quantised for the integer body*/, int* __restrict__ minfo) {
    constexpr int kPrepWaves = kPrepThreads / 64, kPrepRows = kPrepThreads / 16;
    __shared__ float wmax[kPrepWaves];
    __shared__ float q8red[20];
    const int pb = blockIdx.y;
    if (blockIdx.x >= gridDim.x - 2) {                     // the two extra workgroups (of column 0): partition tables, nothing else
        if (pb == 0) {
            if (blockIdx.x == gridDim.x - 2) fill_partition_tables(make_partition(units, tiles, G, seg_cost), n_rb, wg_begin, rb_first, rb_last, wg_sbase);
            else if (qi8) {
                // tiles per substream of the integer body: short streams when some pair of the batch will be quantised (the same
                // sample, the same rule as that pair's own workgroups below)
                int sub8 = kI8SubTiles;
                if (q8)
                    for (int b = 0; b < (int)gridDim.y; ++b)
                        if (q8_sample_grid(P.q[b], ldq, nq, P.t[b], ldt, nt, q8red).kind == 1) sub8 = kQ8SubTiles;
                if (threadIdx.x == 0) { minfo[kMinfoSub8] = sub8; minfo[kMinfoTicket] = 0; }
                fill_partition_tables(make_partition(units8, tiles, G8, seg_cost), n_rb8, wg_begin8, rb_first8, rb_last8, wg_sbase8, sub8);
            }
        }
        return;
    }
    const bool do8 = qi8 != nullptr;
    if (do8) { qi8 += pb * s_qi8; ti8 += pb * s_ti8; wq += pb * s_qn; wt += pb * s_tn; bwmin += pb * kNormBlocks; bwmax += pb * kNormBlocks; rmq += pb * s_qn; rmt += pb * s_tn; }
    int w8min = INT_MAX, w8max = INT_MIN;
    const float* __restrict__ Q = P.q[pb];
    const float* __restrict__ T = P.t[pb];
    int* __restrict__ stats = P.stats[pb];
    qsplit += pb * s_qsplit; tsplit += pb * s_tsplit; qn += pb * s_qn; tn += pb * s_tn;
    bmax += pb * kNormBlocks; midflag += pb * kNormBlocks;
    qerr += pb * s_qn; bmaxerr += pb * kNormBlocks; bqmax += pb * kNormBlocks;
    zero += pb * nzero;
    const bool frag = qfrag != nullptr;
    if (frag) { qfrag += pb * s_qfrag; tfrag += pb * s_tfrag; }
    const int nblk = gridDim.x - 2;
    __shared__ int wmid[kPrepWaves];
    Q8Grid g8{0.f, 1.f, 1.f, 0};
    if (do8 && q8) g8 = q8_sample_grid(Q, ldq, nq, T, ldt, nt, q8red);
    const bool q8pair = g8.kind == 1;                        // (uniform over the pair's workgroups)
    if (do8 && blockIdx.x == 0 && threadIdx.x == 0) {
        minfo[kMinfoQ8S + pb] = __float_as_int(g8.s);
        minfo[kMinfoQ8Lo + pb] = __float_as_int(g8.lo);
    }
    // SIXTEEN lanes per row, one 16-byte chunk (8 elements) each: a lane reads 32 contiguous bytes of its row (a wave = 4
    // rows x 512 B) and its eight fp16 values ARE one chunk of the images — row-major: 16 B at row * 256 + 16 c; fragment
    // order: 16 B at fragment c >> 1, lane 32 (c & 1) + row % 32 of the row's tile, so the four adjacent rows of a wave
    // store 64 contiguous bytes per chunk.  No transposition pass (through LDS, between two barriers, it cost 7 us of a
    // 46 us launch), no shared memory in the loop.
    const int c = threadIdx.x & 15;
    float mx = 0.f, mxe = 0.f, mxq = 0.f;
    unsigned flags = 0;
    const int rows = nq_pad + nt_pad;
    // A workgroup takes kPrepRows consecutive rows per trip; the rows of up to kPrepAhead trips are requested before the first
    // is worked on (the body is a dependent chain and a workgroup has only a few trips).
    constexpr int kPrepAhead = 3;
    const int row0 = blockIdx.x * kPrepRows + (threadIdx.x >> 4), rstep = nblk * kPrepRows;
    for (int rbase = row0; rbase < rows; rbase += kPrepAhead * rstep) {
    float4 vin[kPrepAhead][2];
#pragma unroll
    for (int k = 0; k < kPrepAhead; ++k) {
        const int row = rbase + k * rstep;
        const bool isq = row < nq_pad;
        const int r = isq ? row : row - nq_pad;
        vin[k][0] = vin[k][1] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < rows && r < (isq ? nq : nt)) {
            const float* src = (isq ? Q + (int64_t)r * ldq : T + (int64_t)r * ldt) + 8 * c;
            vin[k][0] = *reinterpret_cast<const float4*>(src);
            vin[k][1] = *reinterpret_cast<const float4*>(src + 4);
        }
    }
#pragma unroll
    for (int k = 0; k < kPrepAhead; ++k) {
        const int row = rbase + k * rstep;
        if (row >= rows) break;
        const bool isq = row < nq_pad;
        const int r = isq ? row : row - nq_pad;
        const int n = isq ? nq : nt, npad = isq ? nq_pad : nt_pad;
        const float in[8] = {vin[k][0].x, vin[k][0].y, vin[k][0].z, vin[k][0].w, vin[k][1].x, vin[k][1].y, vin[k][1].z, vin[k][1].w};
        // ||row||^2: the lane's eight squares as a balanced tree, then the 16 lanes of the row by xor 8, 4, 2, 1 — a fixed
        // shape 8 roundings deep (square, 3 + 4 adds): the refine kernel's certificate allows 9 for this norm (kEps*)
        float s = ((in[0] * in[0] + in[1] * in[1]) + (in[2] * in[2] + in[3] * in[3])) + ((in[4] * in[4] + in[5] * in[5]) + (in[6] * in[6] + in[7] * in[7]));
        s += lane_xor<8>(s); s += lane_xor<4>(s); s += lane_xor<2>(s); s += lane_xor<1>(s);
        const float sc = isq ? -2.f : 1.f;
        const bool real = r < n;
        // Which image does this lane's chunk (8 elements) need?  u8 integers -> the byte image of the exact-integer body (8 B);
        // anything else -> the fp16 image (16 B).  Only that one is written: u8 data costs 2.8 MB of image per 10k x 10k pair
        // instead of 8.1, float data 5.3.  A batch that mixes both kinds runs the 16-bit bodies and needs the fp16 image of its
        // u8 chunks after all: knn_split_images_kernel converts those from the byte image (`rowmask`: one bit per chunk of a row).
        // Rows past the end are zero in BOTH images.
        bool chunk8 = false;
        unsigned lo = 0u, hi = 0u;
        float err2q = 0.f;                                               // ||row - (lo + s k)||^2 of a quantised row (this lane's eight elements)
        if (q8pair) {
            // k = the saturating, rounding conversion of (x - lo) / s; the residual is measured against lo + s k as the float32
            // numbers they are (one fma: within 2^-24 max(|lo|, |hi|) of the real value — refine_q8_body's E carries that term)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                lo = __builtin_amdgcn_cvt_pk_u8_f32((in[e] - g8.lo) * g8.inv, e, lo);
                hi = __builtin_amdgcn_cvt_pk_u8_f32((in[4 + e] - g8.lo) * g8.inv, e, hi);
            }
            if (real) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float kf = (float)(((e < 4 ? lo : hi) >> (8 * (e & 3))) & 0xFFu);
                    const float dv = in[e] - fmaf(g8.s, kf, g8.lo);
                    err2q = fmaf(dv, dv, err2q);
                }
                flags |= kFlagQ8;
            }
            chunk8 = true;
        } else if (do8) {
            // saturating conversion; "every value an integer 0 .. 255" = the conversion was exact (-0 counts as 0; NaN / inf /
            // fractions / out-of-range values leave a nonzero difference)
            // (float data: the first element of a chunk settles it for the whole wave — three instructions instead of thirty)
            lo = __builtin_amdgcn_cvt_pk_u8_f32(in[0], 0, 0u);
            if (__any(((__float_as_uint(in[0] - (float)(lo & 0xFFu))) & 0x7FFFFFFFu) == 0u)) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (e) lo = __builtin_amdgcn_cvt_pk_u8_f32(in[e], e, lo);
                    hi = __builtin_amdgcn_cvt_pk_u8_f32(in[4 + e], e, hi);
                }
                unsigned nu = 0u;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    nu |= __float_as_uint(in[e] - (float)((lo >> (8 * e)) & 0xFFu));
                    nu |= __float_as_uint(in[4 + e] - (float)((hi >> (8 * e)) & 0xFFu));
                }
                chunk8 = (nu & 0x7FFFFFFFu) == 0u;
            }
            if (!chunk8) flags |= kFlagNotU8;
            else if (real) flags |= kFlagSomeU8;
        }
        float err2 = 0.f;                                                // ||fp16(row) - row||^2: the certificate's operand-rounding term
        const float nrm = (isq || real) ? s : kInf;                      // padded train rows can never be candidates
        if (!__all(chunk8 && real)) {                                    // (wave-uniform: four rows of all-u8 chunks skip the 16-bit work)
        unsigned fw[4];                                                  // the eight fp16 values
        err2 = fp16_chunk(in, sc, fw, flags);
        err2 += lane_xor<8>(err2); err2 += lane_xor<4>(err2); err2 += lane_xor<2>(err2); err2 += lane_xor<1>(err2);
        if (!(err2 < kInf)) err2 = 0.f;                                  // (out-of-range data: the split arithmetic runs, this term is unused)
        const uint4 packed = make_uint4(fw[0], fw[1], fw[2], fw[3]);
        if (frag) {
            // FRAGMENT ORDER (knn_filter_q4_kernel): [32-row tile][9 fragments][64 lanes][16 B]; fragment f < 8 is k-step f of
            // v_mfma_f32_32x32x16_f16 (lane 32 h + j holds elements 16 f + 8 h .. + 7 of row j of the tile: chunk c = 2 f + h),
            // then the 512-byte accumulator-init fragment (frag_init_operand: 8 B per lane)
            if (!(chunk8 && real))
                *reinterpret_cast<uint4*>((isq ? qfrag : tfrag) + (int64_t)(r >> 5) * kTileFragBytes + (((c >> 1) * 64 + (c & 1) * 32 + (r & 31)) << 4)) = packed;
        } else {
            *reinterpret_cast<uint4*>((isq ? qsplit : tsplit) + (2 * (int64_t)npad + r) * kDim + 8 * c) = packed;   // row-major fp16 plane (LDS-ring filter, its refine screens)
        }
        }
        if (q8pair) {
            err2q += lane_xor<8>(err2q); err2q += lane_xor<4>(err2q); err2q += lane_xor<2>(err2q); err2q += lane_xor<1>(err2q);
            err2 = err2q < kInf ? err2q : kInf;                          // (NaN / inf data: an infinite slack — everything is re-evaluated exactly)
        }
        // (train image only: the query side of the init product is the same for every query of the pair — ||q||^2max, see
        // knn_filter_q4_kernel — and is formed by the filter itself)
        if (frag && c < 2 && !isq) *reinterpret_cast<uint2*>(tfrag + (int64_t)(r >> 5) * kTileFragBytes + 8 * kFragBytes + ((c * 32 + (r & 31)) << 3)) = frag_init_operand(nrm, false, c);
        if (c == 0) (isq ? qn : tn)[r] = nrm;
        if (do8) {
            const unsigned long long m8 = __ballot(chunk8 && real);      // bit 16 k + c: chunk c of the wave's row k is in the byte image only
            if (c == 0) (isq ? rmq : rmt)[r] = (unsigned short)(m8 >> (threadIdx.x & 48));
        }
        if (do8 && __any(chunk8)) {
            const unsigned flip = isq ? 0x7F7F7F7Fu : 0x80808080u;      // b = 127 - q,  a = t - 128
            const unsigned x0 = real ? lo ^ flip : 0u, x1 = real ? hi ^ flip : 0u;   // rows past the end: zero bytes (masked in the filter: they never make a key)
            // sum (x + 1)^2 over the row = |t - 127|^2 resp. |q - 128|^2 (meaningful only if the whole row is u8: otherwise the pair never runs the integer body)
            int w8 = __builtin_amdgcn_sdot4((int)x0, (int)x0, 8, false);
            w8 = __builtin_amdgcn_sdot4((int)x1, (int)x1, w8, false);
            w8 = __builtin_amdgcn_sdot4((int)x0, 0x02020202, w8, false);
            w8 = __builtin_amdgcn_sdot4((int)x1, 0x02020202, w8, false);
            w8 += lane_xor<8>(w8); w8 += lane_xor<4>(w8); w8 += lane_xor<2>(w8); w8 += lane_xor<1>(w8);
            // fragment order: [32-row tile][k-step f = element / 32][lane 32 h + row % 32][16 B], h = (element / 16) & 1; this lane's
            // eight elements 8 c .. 8 c + 7 are half of the chunk of lane 32 ((c >> 1) & 1) + row % 32 in fragment c >> 2
            if (chunk8) {
                unsigned char* img = isq ? qi8 + (int64_t)(r >> 5) * kI8QTileBytes : ti8 + (int64_t)(r >> 5) * kI8TileBytes;
                *reinterpret_cast<uint2*>(img + (c >> 2) * 1024 + ((((c >> 1) & 1) * 32 + (r & 31)) << 4) + ((c & 1) << 3)) = make_uint2(x0, x1);
            }
            if (c == 0) (isq ? wq : wt)[r] = w8;
            if (!isq && real) { w8min = min(w8min, w8); w8max = max(w8max, w8); }
        }
        if (isq) {
            if (c == 0) qerr[r] = err2;
            mxq = fmaxf(mxq, s);
        } else {
            mx = fmaxf(mx, s);
            mxe = fmaxf(mxe, err2);
        }
    }
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        mx = fmaxf(mx, __shfl_xor(mx, m, 64));
        mxe = fmaxf(mxe, __shfl_xor(mxe, m, 64));
        mxq = fmaxf(mxq, __shfl_xor(mxq, m, 64));
    }
    int wfl = 0;
#pragma unroll
    for (int b = 1; b <= kFlagQ8; b <<= 1) wfl |= __any((flags & b) != 0) ? b : 0;
    __shared__ float wmaxe[kPrepWaves], wmaxq[kPrepWaves];
    __shared__ int w8lo[kPrepWaves], w8hi[kPrepWaves];
    if (do8) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            w8min = min(w8min, __shfl_xor(w8min, m, 64));
            w8max = max(w8max, __shfl_xor(w8max, m, 64));
        }
    }
    if ((threadIdx.x & 63) == 0) {
        wmax[threadIdx.x >> 6] = mx;
        wmaxe[threadIdx.x >> 6] = mxe;
        wmaxq[threadIdx.x >> 6] = mxq;
        wmid[threadIdx.x >> 6] = wfl;
        w8lo[threadIdx.x >> 6] = w8min;
        w8hi[threadIdx.x >> 6] = w8max;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float bm = wmax[0], bme = wmaxe[0], bmq = wmaxq[0];
        int fl = wmid[0], lo8 = w8lo[0], hi8 = w8hi[0];
        for (int w = 1; w < kPrepWaves; ++w) {
            bm = fmaxf(bm, wmax[w]);
            bme = fmaxf(bme, wmaxe[w]);
            bmq = fmaxf(bmq, wmaxq[w]);
            fl |= wmid[w];
            lo8 = min(lo8, w8lo[w]);
            hi8 = max(hi8, w8hi[w]);
        }
        bmax[blockIdx.x] = bm;
        bmaxerr[blockIdx.x] = bme;
        bqmax[blockIdx.x] = bmq;
        midflag[blockIdx.x] = fl;
        if (do8) { bwmin[blockIdx.x] = lo8; bwmax[blockIdx.x] = hi8; }
        if (blockIdx.x == 0 && stats) stats[0] = 0;   // rescanned-query counter (refine kernel)
        if (blockIdx.x == 1 || nblk == 1)
            for (int i = 0; i < nzero; ++i) zero[i] = 0;   // Lowe-ratio survivor counters of the fused match call
    }
}

// Train side of the i8 init product for C in [kI8CMin, kI8CMax]: slot 0 (byte 0 of the h = 0 lanes; query side 1) holds C & 127,
// the other 31 slots (query side -128) hold digits d_k in [-128, 127] with sum d_k = -(C >> 7), dealt greedily in slot order.
__device__ __forceinline__ uint4 frag_init_i8(int C, int h) {
    int S = -(C >> 7);                                         // what slots 1 .. 31 must add up to
    if (h == 1) S -= max(-15 * 128, min(15 * 127, S));         // (slots 1 .. 15 live in the h = 0 lanes)
    unsigned wds[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        int d;
        if (h == 0 && k == 0) d = C & 127;
        else { d = max(-128, min(127, S)); S -= d; }
        wds[k >> 2] |= (unsigned)(d & 0xFF) << (8 * (k & 3));
    }
    return make_uint4(wds[0], wds[1], wds[2], wds[3]);
}

// The bf16 (hi, mid) planes of the images, for the split arithmetic only: every workgroup derives the batch's mode from
// the flags the prep launch left (as the filter will) and returns at once unless it is the split mode.
// (A small grid — one 512-thread workgroup per CU, every one loops over the pairs: launching the waves of a prep-sized
// grid only to have them return cost 7 us.)
constexpr int kSplitThreads = 512;
__global__ __launch_bounds__(kSplitThreads) void knn_split_images_kernel(BatchPtrs P, int B, int64_t ldq, int nq, int nq_pad, int64_t ldt, int nt,
                                                                        int nt_pad, unsigned short* __restrict__ qsplit0,
                                                                        unsigned short* __restrict__ tsplit0, int64_t s_qsplit, int64_t s_tsplit,
                                                                        int* __restrict__ midflag, const float* __restrict__ bmax,
                                                                        int force_mode, unsigned char* __restrict__ qhm0 /*null: row-major planes*/,
                                                                        unsigned char* __restrict__ thm0, int64_t s_qhm, int64_t s_thm,
                                                                        float* __restrict__ bmaxerr, const float* __restrict__ bqmax, int* __restrict__ minfo,
                                                                        // exact-integer body (ti8 == null: not planned)
                                                                        unsigned char* __restrict__ ti8, int64_t s_ti8, const int* __restrict__ wt, int64_t s_tn,
                                                                        const int* __restrict__ bwmin, const int* __restrict__ bwmax,
                                                                        // ... and what the repair of a mixed batch needs (see below)
                                                                        const unsigned char* __restrict__ qi8, int64_t s_qi8, int64_t s_qn,
                                                                        const unsigned short* __restrict__ rmq, const unsigned short* __restrict__ rmt,
                                                                        unsigned char* __restrict__ qfrag, unsigned char* __restrict__ tfrag, int64_t s_qfrag, int64_t s_tfrag,
                                                                        float* __restrict__ qerr /*[B][s_qn]: rewritten for pairs repaired below*/,
                                                                        int* __restrict__ midflag_r, float* __restrict__ bmaxerr_r /*[B][kNormBlocks]: the repair's words*/,
                                                                        int delay_wg, long long delay_ticks /*test hook (sfm_debug_knn_split_delay): workgroup
                                                                        `delay_wg` starts `delay_ticks` of the 100 MHz clock late; -1: none*/) {
    if ((int)blockIdx.x == delay_wg) {                                    // a workgroup that is dispatched after the others have finished
        const long long t0 = wall_clock64();
        while (wall_clock64() - t0 < delay_ticks) __builtin_amdgcn_s_sleep(8);
    }
    // The batch's arithmetic mode, reduced ONCE for the launch set: wave b of every workgroup reduces pair b's 2 x 256 flag
    // words (the eight pairs in parallel: one round trip; as a loop over the pairs inside every filter workgroup this was
    // 8-10 us of dependent loads at the head of the filter launch), workgroup 0 leaves the result — per pair the mode,
    // ||t||max and the largest fp16 residual, and the batch's mode — in `minfo` for the filter and refine kernels.
    __shared__ int smode[kMaxBatch], s8ok[kMaxBatch], s8base[kMaxBatch], s8some[kMaxBatch], sq8[kMaxBatch];
    auto reduce_modes = [&](bool leave /*this workgroup leaves the per-pair results in minfo*/) {
        const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
        static_assert(kSplitThreads / 64 >= kMaxBatch, "one wave per pair");
        if (wave < B) {
            float tmax;
            const int m = knn_filter_mode(midflag + wave * kNormBlocks, bmax + wave * kNormBlocks, lane, &tmax);
            const float* be = bmaxerr + wave * kNormBlocks;
            const float* bq = bqmax + wave * kNormBlocks;
            float te = fmaxf(fmaxf(be[lane], be[lane + 64]), fmaxf(be[lane + 128], be[lane + 192]));
            float qm = fmaxf(fmaxf(bq[lane], bq[lane + 64]), fmaxf(bq[lane + 128], bq[lane + 192]));
#pragma unroll
            for (int sh = 32; sh >= 1; sh >>= 1) {
                te = fmaxf(te, __shfl_xor(te, sh, 64));
                qm = fmaxf(qm, __shfl_xor(qm, sh, 64));
            }
            if (ti8) {
                // exact-integer body: every value a u8 integer, and floor(w_t / 2) of the pair's train rows within the init product's range
                const int* fl = midflag + wave * kNormBlocks;
                const int* lo = bwmin + wave * kNormBlocks;
                const int* hi = bwmax + wave * kNormBlocks;
                const int fl4 = fl[lane] | fl[lane + 64] | fl[lane + 128] | fl[lane + 192];
                const bool u8 = !__any((fl4 & kFlagNotU8) != 0);
                const bool some = __any((fl4 & kFlagSomeU8) != 0);
                const bool q8p = __any((fl4 & kFlagQ8) != 0);              // the prep pass quantised this pair (float data)
                int wl = min(min(lo[lane], lo[lane + 64]), min(lo[lane + 128], lo[lane + 192]));
                int wh = max(max(hi[lane], hi[lane + 64]), max(hi[lane + 128], hi[lane + 192]));
#pragma unroll
                for (int sh = 32; sh >= 1; sh >>= 1) {
                    wl = min(wl, __shfl_xor(wl, sh, 64));
                    wh = max(wh, __shfl_xor(wh, sh, 64));
                }
                const int cl = wl >> 1, ch = wh >> 1;                      // (wl <= wh: nt >= 1 here)
                const int base = max(ch - kI8CMax, min(cl + (ch - cl) / 2, cl - kI8CMin));   // mid-range, nudged so that both ends fit when they can
                // a QUANTISED pair runs the integer body only if the grid fitted: the largest train residual (it enters every query's
                // slack) within 1.6 x what rounding alone leaves, s sqrt(128 / 12).  Beyond that — clipped tails the sample did not
                // show — the 16-bit bodies are the faster exact path: the pair is repaired below.
                const float s8 = __int_as_float(minfo[kMinfoQ8S + wave]);
                // ... and only if no float32 distance of the pair can overflow: sum (q - t)^2 <= (||q|| + ||t||)^2 < FLT_MAX (all
                // of them +inf would tie by index — the bound R below assumes finite values; the 16-bit bodies' business, as before)
                const bool fit = !q8p || (te <= 27.4f * s8 * s8 && tmax < 2.5e37f && qm < 2.5e37f);
                if (lane == 0) {
                    s8ok[wave] = (u8 && fit && wl <= wh && ch - base <= kI8CMax && cl - base >= kI8CMin) ? 1 : 0;
                    s8base[wave] = base;
                    s8some[wave] = some ? 1 : 0;
                    sq8[wave] = q8p ? 1 : 0;
                }
            }
            if (lane == 0) {
                smode[wave] = m;
                if (leave) {
                    minfo[kMinfoPairMode + wave] = m;
                    minfo[kMinfoTmax + wave] = __float_as_int(tmax);
                    minfo[kMinfoTerr + wave] = __float_as_int(te);
                    minfo[kMinfoQmax + wave] = __float_as_int(qm);
                }
            }
        }
        __syncthreads();
    };
    reduce_modes(blockIdx.x == 0);
    // The exact-integer body runs iff EVERY pair of the batch qualifies (one launch, one body).
    bool i8 = ti8 != nullptr;
    for (int b = 0; b < B && i8; ++b) i8 = s8ok[b] != 0;
    bool q8any = false, repaired = false;
    for (int b = 0; b < B && ti8; ++b) q8any = q8any || sq8[b] != 0;
    __shared__ int sq8r[kMaxBatch];                                    // pairs repaired below (their byte image must not be read back as values)
    if (threadIdx.x < kMaxBatch) sq8r[threadIdx.x] = (ti8 && q8any && !i8 && threadIdx.x < B) ? sq8[threadIdx.x] : 0;
    __syncthreads();
    if (ti8 && q8any && !i8) {
        // REPAIR: some pair was quantised by the prep pass (byte image only) but the batch runs a 16-bit body after all — the
        // grid did not fit, the init product's range is exceeded, or another pair of the batch is not integer-body material.
        // Those pairs get their fp16 image, residuals and flags now, from the original floats.  Rows are dealt exactly as in the
        // prep launch (workgroup x of kNormBlocks, sixteen lanes per row), so workgroup x simply REWRITES the pair's per-block
        // words x.  Rare by construction (the sample rule), so it only has to be right.
        __shared__ int rfl[kSplitThreads / 64];
        __shared__ float rme[kSplitThreads / 64];
        const int c = threadIdx.x & 15, rows = nq_pad + nt_pad;
        for (int pb = 0; pb < B; ++pb) {
            if (!sq8[pb]) continue;                                      // (uniform)
            const float* __restrict__ Q = P.q[pb];
            const float* __restrict__ T = P.t[pb];
            unsigned flags = kFlagNotU8;
            float mxe = 0.f;
            if (threadIdx.x < kPrepThreads)
                for (int row = blockIdx.x * (kPrepThreads / 16) + (threadIdx.x >> 4); row < rows; row += kNormBlocks * (kPrepThreads / 16)) {
                    const bool isq = row < nq_pad;
                    const int r = isq ? row : row - nq_pad;
                    const bool real = r < (isq ? nq : nt);
                    float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
                    if (real) {
                        const float* src = (isq ? Q + (int64_t)r * ldq : T + (int64_t)r * ldt) + 8 * c;
                        v0 = *reinterpret_cast<const float4*>(src);
                        v1 = *reinterpret_cast<const float4*>(src + 4);
                    }
                    const float in[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
                    unsigned fw[4];
                    float err2 = fp16_chunk(in, isq ? -2.f : 1.f, fw, flags);
                    err2 += lane_xor<8>(err2); err2 += lane_xor<4>(err2); err2 += lane_xor<2>(err2); err2 += lane_xor<1>(err2);
                    if (!(err2 < kInf)) err2 = 0.f;
                    *reinterpret_cast<uint4*>((isq ? qfrag + pb * s_qfrag : tfrag + pb * s_tfrag) + (int64_t)(r >> 5) * kTileFragBytes +
                                              (((c >> 1) * 64 + (c & 1) * 32 + (r & 31)) << 4)) = make_uint4(fw[0], fw[1], fw[2], fw[3]);
                    if (isq) { if (c == 0) qerr[pb * s_qn + r] = err2; }
                    else mxe = fmaxf(mxe, err2);
                }
            int wfl = 0;
#pragma unroll
            for (int b = 1; b <= kFlagQ8; b <<= 1) wfl |= __any((flags & b) != 0) ? b : 0;
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) mxe = fmaxf(mxe, __shfl_xor(mxe, m, 64));
            __syncthreads();
            if ((threadIdx.x & 63) == 0) { rfl[threadIdx.x >> 6] = wfl; rme[threadIdx.x >> 6] = mxe; }
            __syncthreads();
            if (threadIdx.x == 0) {
                int fl = 0;
                float me = 0.f;
                for (int w = 0; w < kPrepThreads / 64; ++w) { fl |= rfl[w]; me = fmaxf(me, rme[w]); }
                // (kFlagNotU8, no kFlagQ8: a 16-bit pair from here on.)  Into the SHADOW words: midflag / bmaxerr are what every workgroup
                // of this launch decides "repair or not" from, and a workgroup dispatched late — beside other launch sets' kernels the
                // grid is not co-resident — must still find what the first one found (with in-place rewrites it could see
                // no quantised pair any more, skip the repair, never draw a ticket, and leave `minfo` stale)
                midflag_r[pb * kNormBlocks + blockIdx.x] = fl;
                bmaxerr_r[pb * kNormBlocks + blockIdx.x] = me;
            }
        }
        // No grid-wide barrier (round 4 had a spin barrier here: it assumed every workgroup of this launch co-resident — false beside
        // other launch sets' kernels or on a partitioned device — and trapped on time-out).  "Last one out": every workgroup
        // publishes its words (release fence) and draws a ticket; the one that draws the LAST ticket sees them all (acquire
        // fence) and reduces the modes again, alone.  Nobody waits for anybody, so the others cannot know the outcome: they do below
        // whatever ANY outcome could need — the u8 chunks' fp16 image and the bf16 planes of every pair — and leave `minfo` alone.
        __shared__ int last_s;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");               // every lane: its own image / residual stores have left
        __syncthreads();
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (the flag must not overtake the write-back: MI355X_MICROARCH.md, compiler hazard)
            last_s = atomicAdd(&minfo[kMinfoTicket], 1) == (int)gridDim.x - 1 ? 1 : 0;
        }
        __syncthreads();
        repaired = true;
        i8 = false;
        q8any = false;
        if (last_s) {                                                    // (uniform)
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");           // the other workgroups' words, not this CU's cached copies
            // every workgroup has decided (a ticket is drawn after the decision): the repaired pairs' words take their place now
            for (int e = threadIdx.x; e < B * kNormBlocks; e += kSplitThreads)
                if (sq8[e / kNormBlocks]) { midflag[e] = midflag_r[e]; bmaxerr[e] = bmaxerr_r[e]; }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            __syncthreads();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            reduce_modes(true);
            if (threadIdx.x == 0) {
                int m = kModeHalfExact;
                for (int b = 0; b < B; ++b) m = max(m, smode[b]);
                minfo[kMinfoBatchMode] = m;
                minfo[kMinfoI8] = 0;
                minfo[kMinfoQ8] = 0;
            }
        }
    }
    int mode = kModeHalfExact;
    for (int b = 0; b < B; ++b) mode = max(mode, smode[b]);
    // (repaired: workgroup 0's view of the modes is the one from BEFORE the repair — the last workgroup out has written them)
    if (!repaired && blockIdx.x == 0 && threadIdx.x == 0) minfo[kMinfoBatchMode] = mode;
    // The integer body's init fragments (frag_init_i8: the digits of floor(w_t / 2) - base) need the pair's base, so they are
    // written here, not by the prep pass.
    if (!repaired && blockIdx.x == 0 && threadIdx.x < kMaxBatch + 1) {
        if (threadIdx.x == kMaxBatch) { minfo[kMinfoI8] = i8 ? 1 : 0; minfo[kMinfoQ8] = (i8 && q8any) ? 1 : 0; }
        else if (threadIdx.x < B && ti8) minfo[kMinfoBase + threadIdx.x] = s8base[threadIdx.x];
    }
    if (i8) {
        // one flat loop over (pair, row, half-wave): every load of the launch is in flight at once (a loop over the pairs was a
        // dependent round trip per pair: 9 us for a batch of 8)
        const int per_pair = 2 * nt_pad;
        for (int e = blockIdx.x * kSplitThreads + threadIdx.x; e < B * per_pair; e += gridDim.x * kSplitThreads) {
            const int pb = e / per_pair, x = e - pb * per_pair;
            const int r = x >> 1, h = x & 1;
            const int wv = r < nt ? wt[pb * s_tn + r] : 0;
            *reinterpret_cast<uint4*>(ti8 + pb * s_ti8 + (int64_t)(r >> 5) * kI8TileBytes + 4 * 1024 + ((h * 32 + (r & 31)) << 4)) =
                frag_init_i8(r < nt ? (wv >> 1) - s8base[pb] : kI8CMax, h);
        }
        return;
    }
    if (ti8) {
        // A 16-bit body runs although the integer body was planned: chunks the prep pass found to be u8 integers exist in the
        // byte image only (rmq / rmt: one bit per 8-element chunk of a row).  Convert them: t = a ^ 0x80, q = b ^ 0x7F are exact in
        // fp16, and so is -2 q.  Nothing to do for pure float data (no such chunk: one flag word per pair says so).
        bool some = false;
        for (int b = 0; b < B; ++b) some = some || s8some[b] != 0;
        if (some) {
            const int rows = nq_pad + nt_pad;
            for (int64_t e = (int64_t)blockIdx.x * kSplitThreads + threadIdx.x; e < (int64_t)B * rows * 16; e += (int64_t)gridDim.x * kSplitThreads) {
                const int c = (int)(e & 15);
                const int64_t rr = e >> 4;
                const int pb = (int)(rr / rows), row = (int)(rr - (int64_t)pb * rows);
                const bool isq = row < nq_pad;
                const int r = isq ? row : row - nq_pad;
                const unsigned m = isq ? rmq[pb * s_qn + r] : rmt[pb * s_tn + r];
                if (!((m >> c) & 1u) || sq8r[pb]) continue;              // (a quantised pair's bytes are not its values: repaired above)
                const unsigned char* src = (isq ? qi8 + pb * s_qi8 + (int64_t)(r >> 5) * kI8QTileBytes : ti8 + pb * s_ti8 + (int64_t)(r >> 5) * kI8TileBytes) +
                                           (c >> 2) * 1024 + ((((c >> 1) & 1) * 32 + (r & 31)) << 4) + ((c & 1) << 3);
                const uint2 x = *reinterpret_cast<const uint2*>(src);
                unsigned fb[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const unsigned byte = ((k < 4 ? x.x : x.y) >> (8 * (k & 3))) & 0xFFu;
                    const float v = isq ? -2.f * (float)(byte ^ 0x7Fu) : (float)(byte ^ 0x80u);
                    fb[k] = (unsigned)__builtin_bit_cast(unsigned short, (_Float16)v);
                }
                unsigned char* dst = (isq ? qfrag + pb * s_qfrag : tfrag + pb * s_tfrag) + (int64_t)(r >> 5) * kTileFragBytes + (((c >> 1) * 64 + (c & 1) * 32 + (r & 31)) << 4);
                *reinterpret_cast<uint4*>(dst) = make_uint4(fb[0] | (fb[1] << 16), fb[2] | (fb[3] << 16), fb[4] | (fb[5] << 16), fb[6] | (fb[7] << 16));
            }
        }
    }
    if (force_mode >= 0) mode = force_mode;
    if (mode != kModeSplit && !repaired) return;                 // (repaired: the mode is not known yet — see above)
    const bool frag = qhm0 != nullptr;
    const int l = threadIdx.x & 31;
    const int rows = nq_pad + nt_pad;
    constexpr int kRowsPerPass = kSplitThreads / 32;
    for (int pb = 0; pb < B; ++pb) {
    const float* __restrict__ Q = P.q[pb];
    const float* __restrict__ T = P.t[pb];
    unsigned short* __restrict__ qsplit = qsplit0 + pb * s_qsplit;
    unsigned short* __restrict__ tsplit = tsplit0 + pb * s_tsplit;
    for (int row = blockIdx.x * kRowsPerPass + (threadIdx.x >> 5); row < rows; row += gridDim.x * kRowsPerPass) {
        const bool isq = row < nq_pad;
        const int r = isq ? row : row - nq_pad;
        const int n = isq ? nq : nt, npad = isq ? nq_pad : nt_pad;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < n) v = *reinterpret_cast<const float4*>((isq ? Q + (int64_t)r * ldq : T + (int64_t)r * ldt) + 4 * l);
        const float sc = isq ? -2.f : 1.f;
        const float e[4] = {sc * v.x, sc * v.y, sc * v.z, sc * v.w};
        unsigned hb[4], mb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            hb[k] = bf16_rn_bits(e[k]);
            mb[k] = bf16_rn_bits(e[k] - __uint_as_float(hb[k] << 16));   // x - hi is exact in fp32
        }
        const uint2 hp = make_uint2(hb[0] | (hb[1] << 16), hb[2] | (hb[3] << 16)), mp = make_uint2(mb[0] | (mb[1] << 16), mb[2] | (mb[3] << 16));
        if (frag) {      // fragment order (q4 filter): per 32-row tile 8 hi fragments then 8 mid fragments, 1 KiB each (see knn_prep_kernel)
            unsigned char* fimg = (isq ? qhm0 + pb * s_qhm : thm0 + pb * s_thm) + (int64_t)(r >> 5) * (16 * kFragBytes);
            const int c = l >> 1;
            const int off = (((c >> 1) * 64 + (c & 1) * 32 + (r & 31)) << 4) + ((l & 1) << 3);
            *reinterpret_cast<uint2*>(fimg + off) = hp;
            *reinterpret_cast<uint2*>(fimg + 8 * kFragBytes + off) = mp;
        } else {
            unsigned short* img = isq ? qsplit : tsplit;
            *reinterpret_cast<uint2*>(img + (int64_t)r * kDim + 4 * l) = hp;
            *reinterpret_cast<uint2*>(img + ((int64_t)npad + r) * kDim + 4 * l) = mp;
        }
    }
    }
}

// ---------------------------------------------------------------- 16-bit filter, two query groups per wave
// LDS tile image: hi rows [32][256 B] at +0, mid rows (split) / the next tile's rows (fp16) at +8 KiB, 16-byte chunks
// XOR-swizzled with (row & 15).  Structured for the matrix pipe's sweet spot of TWO waves per SIMD (a pure 16-bit MFMA
// chain sustains 2.1 PFLOP/s at <= 2 waves/SIMD but 1.5 at 4 — scripts/ubench):
//   * a wave owns 64 queries (two 32-column groups, 128 VGPRs of query fragments), so every train fragment read
//     from LDS feeds 6 MFMAs and the two groups' accumulators form two independent dependency chains;
//   * software pipelining inside the wave: while tile t runs on the matrix pipe, the packed-key epilogue of
//     tile t-1 (its accumulators are kept) is interleaved between the MFMAs on the vector pipe.
// The pack is ONE v_and_or_b32 only if the mask sits in a VGPR and the sequence number in an SGPR (gfx9 VOP3 takes a
// single scalar operand and no literal; left alone hipcc keeps both scalar and emits v_and + v_or).
// One key per QUAD of adjacent trains (accumulator registers 4m..4m+3 = rows 8m + 4h + 0..3): min first, then one
// insertion — 7 VALU per four scores instead of 16 (the loop is instruction-issue bound: ~8 slots per 32-cycle MFMA
// per SIMD, shared by two waves).  A candidate record is therefore a row quad; the refine kernel evaluates all four
// rows exactly, and a discarded quad has ALL scores >= the stream's 3rd-best quad minimum, so the certificate is
// unchanged.
template <int W>
__device__ __forceinline__ void key_insert4(const f32x16& a, int r0, int seq0 /*wave-uniform*/, int vmask /*VGPR holding ~kKeyMask*/,
                                            int& k0, int& k1, int& k2) {
    static_assert(W % 4 == 0, "whole register quads");
#pragma unroll
    for (int r = r0; r < r0 + W; r += 4) {
        int key;
        // min on the BIT PATTERNS (v_min3_i32 + v_min_i32): fminf would add a canonicalising v_max_f32 per operand.
        // Scores are >= 0 up to rounding noise; a negative one still wins against every non-negative one, and among
        // negative ones (noise) any is a valid record score (such a stream never certifies, see the refine kernel).
        const int m = min(min(__float_as_int(a[r]), __float_as_int(a[r + 1])), min(__float_as_int(a[r + 2]), __float_as_int(a[r + 3])));
        asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(m), "v"(vmask), "s"(seq0 + (r >> 2)));
        const int lo = min(key, k0);
        const int m1 = max(min(key, k1), min(max(key, k1), k0));
        k2 = max(min(key, k1), min(max(key, k1), k2));
        k1 = m1;
        k0 = lo;
    }
}

// LDS ring of the pipelined filter: 3 tile images (16 KiB each) + 3 x 64 floats of ||t||^2.
constexpr int kRing = 3;
constexpr int kRingLdsBytes = kRing * kTileFloats * 4 + kRing * 256;
constexpr int kQScratchBytes = 4 * 4096;            // per-wave transposition slabs of the 4-wave filter's query-fragment prologue

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// Wait until at most `keep` of this wave's vector-memory operations are still in flight (keep is wave-uniform).
__device__ __forceinline__ void wait_vm_keep(int keep) {
    switch (keep) {
        case 0: wait_vmcnt<0>(); break;
        case 1: wait_vmcnt<1>(); break;
        case 2: wait_vmcnt<2>(); break;
        case 3: wait_vmcnt<3>(); break;
        case 4: wait_vmcnt<4>(); break;
        default: wait_vmcnt<5>(); break;
    }
}

template <int ABL, int W, bool KMID>
__device__ __forceinline__ void filter_split2_body(
    float* smem, const unsigned short* __restrict__ qsplit, const float* __restrict__ qnorm, int nq, int nq_pad,
    const unsigned short* __restrict__ tsplit, int nt_pad, const float* __restrict__ tn, int tiles, int64_t units,
    int smax, int nsub, float* __restrict__ cand_s, int* __restrict__ cand_i, const int64_t* __restrict__ wg_begin,
    const int* __restrict__ rb_first, int n_rb1, int64_t s_qsplit, int64_t s_tsplit, int64_t s_qn, int64_t s_tn, int64_t s_cand,
    long long* __restrict__ trace) {
    // All pointers are pair 0's; pair b of the batch sits at + b * stride.  A segment (the part of this workgroup's range
    // inside one query row block) belongs to one pair: its images, norms and candidate arrays are selected per segment.
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31;
    const int h = lane >> 5;
    const int hm = h ^ (j & 15);
    // XCD-aware order: consecutive workgroup ids are dealt round-robin to the 8 XCDs (each with its own L2), while
    // consecutive ranges of the partition share query row blocks and neighbouring train tiles.  Physical workgroup b
    // takes range (b % 8) * (G / 8) + b / 8, so each XCD works on ONE contiguous eighth of the (row block, tile) space:
    // its L2 then holds an eighth of the query image instead of all of it.
    const int G = gridDim.x;
    const int bid = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int64_t u_end = wg_begin[bid + 1];
    int64_t u = wg_begin[bid];
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) float*)smem;
    const unsigned lds_tn = lds0 + kRing * kTileFloats * 4;
    const int mid_off = nt_pad * 256;
    const int img_off = KMID ? 0 : 2 * mid_off;                 // single-product modes read the fp16 image
    __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)tsplit, 0, 3 * mid_off, 0x00020000);      // (re-made per segment:
    __amdgpu_buffer_rsrc_t tnrs = __builtin_amdgcn_make_buffer_rsrc((void*)tn, 0, nt_pad * 4, 0x00020000);          //  the pair's own arrays)
    const unsigned short* __restrict__ qsplit0 = qsplit;
    const float* __restrict__ qnorm0 = qnorm;
    float* __restrict__ cand_s0 = cand_s;
    int* __restrict__ cand_i0 = cand_i;
    constexpr int PIECES = 16 / W;
    const int p0 = wave * PIECES;
    const int r0 = 4 * (p0 & 7) + (lane >> 4);
    const int lane_off = r0 * 256 + (((lane & 15) ^ (r0 & 15)) << 4);
    const bool stage_pieces = true;
    const int my_vm = (stage_pieces ? PIECES : 0) + (wave == 0 ? 1 : 0);   // VMEM ops this wave issues per staged tile

    // Train tile `tile` → ring slot `buf`, entirely by LDS-DMA (hi/mid images 1 KiB per piece; ||t||^2 as one dword
    // piece from wave 0: padded rows hold +inf).  No VGPR destinations, so nothing here makes hipcc wait.
    // Single-product body: a ring slot holds TWO consecutive tiles of the fp16 image (the second where the split body
    // keeps its mid rows), `tile` is the first of the pair, and there is one barrier per PAIR of tiles.
    auto stage = [&](int tile, int buf) {
        const int soff = __builtin_amdgcn_readfirstlane(KMID ? tile * kTileT * 256 + (p0 >= 8 ? mid_off : 0)
                                                             : (tile + (p0 >= 8 ? 1 : 0)) * kTileT * 256 + img_off);
        const unsigned dst0 = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)buf * (kTileFloats * 4) + (unsigned)p0 * 1024u);
        if (stage_pieces)
#pragma unroll
            for (int n = 0; n < PIECES; ++n)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(trs, (lptr_t)(size_t)(dst0 + n * 1024), 16,
                                                         (lane_off ^ (64 * n)) + 4 * n * 256, soff, 0, 0);
        if (wave == 0) {
            const unsigned dtn = __builtin_amdgcn_readfirstlane(lds_tn + (unsigned)buf * 256u);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(tnrs, (lptr_t)(size_t)dtn, 4, lane * 4,
                                                     __builtin_amdgcn_readfirstlane(tile * kTileT * 4), 0, 0);
        }
    };

    while (u < u_end) {
        const int rb = (int)(u / tiles);
        const int t_begin = (int)(u - (int64_t)rb * tiles);
        const int t_end = (int)min((int64_t)tiles, t_begin + (u_end - u));
        const int slot = bid - rb_first[rb];                       // (rb is the GLOBAL row block: the tables span the batch)
        const int pb = n_rb1 > 0 ? rb / n_rb1 : 0;                 // pair of the batch this row block belongs to
        const int rbl = rb - pb * n_rb1;                           // row block inside the pair
        qsplit = qsplit0 + pb * s_qsplit;
        qnorm = qnorm0 + pb * s_qn;
        cand_s = cand_s0 + pb * s_cand;
        cand_i = cand_i0 + pb * s_cand;
        trs = __builtin_amdgcn_make_buffer_rsrc((void*)(tsplit + pb * s_tsplit), 0, 3 * mid_off, 0x00020000);
        tnrs = __builtin_amdgcn_make_buffer_rsrc((void*)(tn + pb * s_tn), 0, nt_pad * 4, 0x00020000);
        const int qrow0 = rbl * (W * 64) + wave * 64 + j;         // group g adds 32*g
        const bool qok[2] = {qrow0 < nq, qrow0 + 32 < nq};

        __syncthreads();                                           // previous segment fully consumed, nothing in flight
        constexpr int kPerSlot = KMID ? 1 : 2;                      // tiles per ring slot
        // Query fragments.  A lane needs 16 x 16 B of ITS query row (B operand: column j, k-chunk 2st+h), i.e. a
        // wave-level load touches 32 rows x 32 B — 64 separate requests per instruction, and issuing the 16 of them
        // took 4 us of a 6.6 us prologue.  Single-product body with 4-wave workgroups: load the wave's rows COALESCED
        // (1 KiB = 4 whole rows per instruction) and transpose through a 4 KiB per-wave LDS slab behind the ring (16 rows
        // at a time, same XOR swizzle as the tile image) while the first tiles' DMA — issued first — is in flight.
        constexpr bool kCoalescedQ = !KMID && W == 4;
        stage(t_begin, 0);
        if (t_begin + kPerSlot < t_end && !(ABL & 4)) stage(t_begin + kPerSlot, 1);

        uint4 bh[2][8], bm[2][8];
        float qn[2];
        if constexpr (kCoalescedQ) {
            const unsigned short* img = qsplit + (2 * (int64_t)nq_pad + rbl * (W * 64) + wave * 64) * kDim;   // fp16 image, this wave's 64 rows
            const int lr = lane >> 4, lc = lane & 15;                // row within a 4-row load, 16-byte chunk
#pragma unroll
            for (int g = 0; g < 2; ++g)
#pragma unroll
                for (int n = 0; n < 8; ++n)
                    bh[g][n] = *reinterpret_cast<const uint4*>(img + (int64_t)(32 * g + 4 * n + lr) * kDim + 8 * lc);
            // (plain LDS accesses: the wait they imply covers the tile DMA issued above, which has to land anyway)
            char* const scratch = reinterpret_cast<char*>(smem) + kRingLdsBytes + wave * 4096;
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                uint4 got[8];
#pragma unroll
                for (int s16 = 0; s16 < 2; ++s16) {                  // rows 16*s16 .. +16 of the group
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        const int r = 4 * n + lr;                    // row inside the slab
                        *reinterpret_cast<uint4*>(scratch + r * 256 + ((lc ^ r) << 4)) = bh[g][4 * s16 + n];
                    }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    if ((j >> 4) == s16)                             // the 32 lanes whose query row is in this slab
#pragma unroll
                        for (int st = 0; st < 8; ++st)
                            got[st] = *reinterpret_cast<const uint4*>(scratch + (j & 15) * 256 + ((((2 * st + h) ^ (j & 15))) << 4));
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                }
#pragma unroll
                for (int st = 0; st < 8; ++st) bh[g][st] = got[st];
                qn[g] = qok[g] ? qnorm[qrow0 + 32 * g] : 0.f;
            }
        } else {
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                const int qr = qok[g] ? qrow0 + 32 * g : 0;
                const unsigned short* sh = qsplit + ((KMID ? 0 : 2 * (int64_t)nq_pad) + qr) * kDim + 8 * h;
                const unsigned short* sm = qsplit + ((int64_t)nq_pad + qr) * kDim + 8 * h;
#pragma unroll
                for (int st = 0; st < 8; ++st) {
                    bh[g][st] = *reinterpret_cast<const uint4*>(sh + 16 * st);
                    if (KMID) bm[g][st] = *reinterpret_cast<const uint4*>(sm + 16 * st);
                }
                qn[g] = qok[g] ? qnorm[qrow0 + 32 * g] : 0.f;
            }
        }
        const float b_aug[2] = {h ? qn[0] : 1.f, h ? qn[1] : 1.f};
        int vmask;
        asm volatile("v_mov_b32 %0, %1" : "=v"(vmask) : "s"(~kKeyMask));

        int ka[2] = {kKeyInf, kKeyInf}, kb[2] = {kKeyInf, kKeyInf}, kc[2] = {kKeyInf, kKeyInf};
        int sub = 0, sub_t0 = t_begin;
        auto flush = [&](int sb, int st0) {                        // candidate records of both query groups
#pragma unroll
            for (int g = 0; g < 2; ++g)
                if (qok[g]) {
                    const int64_t ob = ((int64_t)(qrow0 + 32 * g) * (2 * smax * nsub) + (int64_t)slot * nsub * 2 + h) * 3 + 6 * sb;
                    flush_keys<!KMID>(ka[g], kb[g], kc[g], st0, h, cand_s + ob, cand_i + ob);
                }
            ka[0] = kb[0] = kc[0] = ka[1] = kb[1] = kc[1] = kKeyInf;
        };
        wait_vmcnt<0>();                                           // first tile(s) + query fragments landed
        asm volatile("" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (trace && threadIdx.x == 0) trace[8192 + 4 * blockIdx.x + 0] += wall_clock64() - trace[4 * blockIdx.x];   // dev: prologue(s)

        f32x16 accA[2], accB[2];
        // the first tile's "previous tile" is +inf everywhere: its insertions leave the (+inf) keys untouched, so the
        // epilogue needs no have-a-previous-tile branch (which cost two register copies per quad at the join)
#pragma unroll
        for (int g = 0; g < 2; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) accB[g][r] = kInf;
        // one tile: MFMAs of tile t into `cur`, packed-key inserts of tile t-1 from `prev` interleaved.
        // Every LDS access in here is inline asm: an ordinary load would make hipcc drain the in-flight LDS-DMA.
        // Single-product body: the loop is bound by a wave's own instruction stream, and the fragment addresses cost 10
        // VALU + a dozen SALU per tile (ring slot = (tile / 2) % 3, then 7 v_xor for the swizzle).  Instead: eight address
        // registers (swizzle applied once), the tile's half of the slot as the instructions' immediate offset (it is fixed
        // in each of the two unrolled copies), and the slot advanced by one add per register every SECOND tile.
        constexpr bool kFastAddr = !KMID && ABL == 0;
        unsigned fa[8] = {}, ta = 0;
        int rbuf = 0;
        if constexpr (kFastAddr) {
#pragma unroll
            for (int st = 0; st < 8; ++st) fa[st] = lds0 + (((unsigned)j * 256u + ((unsigned)hm << 4)) ^ (32u * st));
            ta = lds_tn + 4u * j;
        }
        auto tile = [&](f32x16(&cur)[2], f32x16(&prev)[2], int t, bool have_prev, auto half_c) {
            constexpr int kHalf = KMID ? 0 : decltype(half_c)::value;   // which tile of the slot's pair (fixed per unrolled copy)
            if (have_prev && (t - 1) - sub_t0 == kSubTiles) {
                flush(sub, sub_t0);
                wait_vmcnt<0>();                                   // stores are counted in vmcnt too: restart the count
                ++sub;
                sub_t0 = t - 1;
            }
            const int rel = t - t_begin;
            const int buf = kFastAddr ? rbuf : (KMID ? rel : rel >> 1) % kRing;
            const int half = kHalf;
            if (half == 0 && t + 2 * kPerSlot < t_end && !(ABL & 4)) stage(t + 2 * kPerSlot, buf + 2 >= kRing ? buf + 2 - kRing : buf + 2);
            const unsigned abase = lds0 + (unsigned)(((ABL & 4) ? 0 : buf) * kTileFloats) * 4u + (unsigned)half * 8192u +
                                   (unsigned)j * 256u + ((unsigned)hm << 4);
            const unsigned tnad = lds_tn + (unsigned)((ABL & 4) ? 0 : buf) * 256u + (unsigned)half * 128u + 4u * j;
            float tnj;
            u32x4 ah[KMID ? 2 : 8], am[2];
            if constexpr (kFastAddr) {
                if constexpr (kHalf == 0) asm volatile("ds_read_b32 %0, %1" : "=v"(tnj) : "v"(ta));
                else asm volatile("ds_read_b32 %0, %1 offset:128" : "=v"(tnj) : "v"(ta));
            } else {
                asm volatile("ds_read_b32 %0, %1" : "=v"(tnj) : "v"(tnad));
            }
            if (KMID) {
                asm volatile("ds_read_b128 %0, %1" : "=v"(ah[0]) : "v"(abase));
                asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(am[0]) : "v"(abase));
                asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(tnj));
            } else {
                // single-product body: a k-step is only 2 MFMAs (64 pipe cycles), less than the LDS latency, so the
                // whole A fragment of the tile (8 x 16 B per lane) is requested up front; LDS returns in order
#pragma unroll
                for (int st = 0; st < 8; ++st) {
                    if constexpr (!kFastAddr) asm volatile("ds_read_b128 %0, %1" : "=v"(ah[st]) : "v"(abase ^ (32u * st)));
                    else if constexpr (kHalf == 0) asm volatile("ds_read_b128 %0, %1" : "=v"(ah[st]) : "v"(fa[st]));
                    else asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(ah[st]) : "v"(fa[st]));
                }
                asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(tnj));
                if constexpr (kFastAddr && kHalf == 1) {            // the slot's second tile has issued its reads: on to the next slot
                    const int nb = rbuf == kRing - 1 ? 0 : rbuf + 1;
                    const int delta = nb == 0 ? -(kRing - 1) * (kTileFloats * 4) : kTileFloats * 4;
#pragma unroll
                    for (int st = 0; st < 8; ++st) fa[st] += (unsigned)delta;
                    ta += (unsigned)(nb == 0 ? -(kRing - 1) * 256 : 256);
                    rbuf = nb;
                }
            }
            // accumulator init ||t||^2 + ||q||^2 as one fp32 MFMA (A = [||t||^2, 1], B = [1; ||q||^2], C = 0): the loop is
            // bound by VALU issue (the matrix pipe idles more than half the time), so 32 v_add per tile cost more
            // than 2 x 64 matrix-pipe cycles
            {
                const float a_aug = h ? 1.f : tnj;
                const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int g = 0; g < 2; ++g) cur[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_aug, b_aug[g], zero, 0, 0, 0);
            }
            const int seq0 = __builtin_amdgcn_readfirstlane(max((t - 1) - sub_t0, 0) << 2);   // (first tile: the +inf previous tile)
#pragma unroll
            for (int st = 0; st < 8; ++st) {
                if constexpr (KMID) {
                    if (st + 1 < 8) {
                        const unsigned ad = abase ^ (32u * (st + 1));
                        asm volatile("ds_read_b128 %0, %1" : "=v"(ah[(st + 1) & 1]) : "v"(ad));
                        asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(am[(st + 1) & 1]) : "v"(ad));
                        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah[st & 1]), "+v"(am[st & 1]));
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[st & 1]), "+v"(am[st & 1]));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const bf16x8 Ah = __builtin_bit_cast(bf16x8, ah[st & 1]), Am = __builtin_bit_cast(bf16x8, am[st & 1]);
#pragma unroll
                    for (int g = 0; g < 2; ++g) {
                        const bf16x8 Bh = __builtin_bit_cast(bf16x8, bh[g][st]), Bm = __builtin_bit_cast(bf16x8, bm[g][st]);
                        cur[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bh, cur[g], 0, 0, 0);
                        cur[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah, Bm, cur[g], 0, 0, 0);
                        cur[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Am, Bh, cur[g], 0, 0, 0);
                    }
                } else {
                    switch (st) {                        // wait for fragment st only: 7 - st later ones stay in flight
                        case 0: asm volatile("s_waitcnt lgkmcnt(7)" : "+v"(ah[0])); break;
                        case 1: asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(ah[1 % (KMID ? 2 : 8)])); break;
                        case 2: asm volatile("s_waitcnt lgkmcnt(5)" : "+v"(ah[2 % (KMID ? 2 : 8)])); break;
                        case 3: asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ah[3 % (KMID ? 2 : 8)])); break;
                        case 4: asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(ah[4 % (KMID ? 2 : 8)])); break;
                        case 5: asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(ah[5 % (KMID ? 2 : 8)])); break;
                        case 6: asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(ah[6 % (KMID ? 2 : 8)])); break;
                        default: asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ah[7 % (KMID ? 2 : 8)])); break;
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int g = 0; g < 2; ++g)
                        cur[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[st % (KMID ? 2 : 8)]),
                                                                       __builtin_bit_cast(f16x8, bh[g][st]), cur[g], 0, 0, 0);
                }
                if (have_prev && (ABL & 1)) {    // dev ablation: keep the MFMAs alive with one op per k-step
                    ka[0] = min(ka[0], __float_as_int(prev[0][2 * st]) + __float_as_int(prev[0][2 * st + 1]));
                    ka[1] = min(ka[1], __float_as_int(prev[1][2 * st]) + __float_as_int(prev[1][2 * st + 1]));
                } else {                         // a register quad of each group every second k-step, beside the MFMAs
                    if (st & 1) {                // one register quad of each group every second k-step
                        key_insert4<4>(prev[0], 4 * (st >> 1), seq0, vmask, ka[0], kb[0], kc[0]);
                        key_insert4<4>(prev[1], 4 * (st >> 1), seq0, vmask, ka[1], kb[1], kc[1]);
                    }
                }
            }
            // Tile t+1 must have landed for every wave before anyone reads it; tile t+2 (just issued) stays in flight
            // across the barrier: counted vmcnt + raw s_barrier (a __syncthreads() would drain the DMA queue).
            // (single-product body: only after the second tile of a pair; then the next PAIR must have landed and the
            // one after it, issued while this pair's first tile ran, stays in flight)
            if (t + 1 < t_end && (KMID || half == 1) && !(ABL & 2)) {
                if (t + 1 + kPerSlot < t_end) wait_vm_keep(my_vm);
                else wait_vmcnt<0>();
                asm volatile("" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
            }
        };

        // Two copies of the tile body with the accumulator sets swapping roles (tile being computed / previous tile,
        // whose epilogue runs inside tile()): the loop is VALU-issue bound and a rotating copy would cost 32 v_mov per tile.
        for (int t = t_begin; t < t_end; t += 2) {
            tile(accA, accB, t, t > t_begin, std::integral_constant<int, 0>{});
            if (t + 1 < t_end) tile(accB, accA, t + 1, true, std::integral_constant<int, 1>{});
        }
        if (trace && threadIdx.x == 0) trace[8192 + 4 * blockIdx.x + 1] = wall_clock64();   // dev: tile loop done
        if (t_end > t_begin) {                             // epilogue of the last tile
            const int tl = t_end - 1;
            if (tl - sub_t0 == kSubTiles) {
                flush(sub, sub_t0);
                ++sub;
                sub_t0 = tl;
            }
            if ((t_end - t_begin) & 1) {
                key_insert4<16>(accA[0], 0, __builtin_amdgcn_readfirstlane((tl - sub_t0) << 2), vmask, ka[0], kb[0], kc[0]);
                key_insert4<16>(accA[1], 0, __builtin_amdgcn_readfirstlane((tl - sub_t0) << 2), vmask, ka[1], kb[1], kc[1]);
            } else {
                key_insert4<16>(accB[0], 0, __builtin_amdgcn_readfirstlane((tl - sub_t0) << 2), vmask, ka[0], kb[0], kc[0]);
                key_insert4<16>(accB[1], 0, __builtin_amdgcn_readfirstlane((tl - sub_t0) << 2), vmask, ka[1], kb[1], kc[1]);
            }
        }
        flush(sub, sub_t0);
#pragma unroll
        for (int g = 0; g < 2; ++g)
            if (qok[g]) {
                const int64_t ob = ((int64_t)(qrow0 + 32 * g) * (2 * smax * nsub) + (int64_t)slot * nsub * 2 + h) * 3;
                for (int e = sub + 1; e < nsub; ++e)
#pragma unroll
                    for (int r = 0; r < 3; ++r) {
                        cand_s[ob + 6 * e + r] = kInf;
                        cand_i[ob + 6 * e + r] = -1;
                    }
            }
        wait_vmcnt<0>();
        u += t_end - t_begin;
    }
}

// One launch, two bodies: KMID = false is the single-product fp16 body (kModeHalfExact / kModeHalf, chosen on the
// device by knn_filter_mode), KMID = true the three-product bf16 split.  The branch is taken once per workgroup, so
// only the chosen body's instructions are ever fetched.
template <int ABL, int W>
__global__ __launch_bounds__(64 * W, 2) void knn_filter_split2_kernel(
    const unsigned short* __restrict__ qsplit, const float* __restrict__ qnorm, int nq, int nq_pad,
    const unsigned short* __restrict__ tsplit, int nt, int nt_pad, const float* __restrict__ tn, int tiles, int64_t units,
    int smax, int nsub, const int* __restrict__ midflag, const float* __restrict__ bmax, int force_mode,
    float* __restrict__ cand_s, int* __restrict__ cand_i, const int64_t* __restrict__ wg_begin, const int* __restrict__ rb_first,
    int n_rb1, int n_pairs, int64_t s_qsplit, int64_t s_tsplit, int64_t s_qn, int64_t s_tn, int64_t s_cand, int* __restrict__ minfo,
    const float* __restrict__ bmaxerr, long long* __restrict__ trace) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if (trace && threadIdx.x == 0) {
        trace[4 * blockIdx.x + 0] = wall_clock64();
        trace[4 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg(0xF804);
        trace[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg(0xF814);
        trace[8192 + 4 * blockIdx.x + 2] = clock64();
    }
    const bool need_mid = (force_mode >= 0 ? force_mode : minfo[kMinfoBatchMode]) == kModeSplit;   // (reduced by knn_split_images_kernel)
    if (need_mid)
        filter_split2_body<ABL, W, true>(smem, qsplit, qnorm, nq, nq_pad, tsplit, nt_pad, tn, tiles, units, smax, nsub, cand_s, cand_i, wg_begin, rb_first,
                                         n_rb1, s_qsplit, s_tsplit, s_qn, s_tn, s_cand, trace);
    else
        filter_split2_body<ABL, W, false>(smem, qsplit, qnorm, nq, nq_pad, tsplit, nt_pad, tn, tiles, units, smax, nsub, cand_s, cand_i, wg_begin, rb_first,
                                          n_rb1, s_qsplit, s_tsplit, s_qn, s_tn, s_cand, trace);
    if (trace && threadIdx.x == 0) {
        trace[4 * blockIdx.x + 1] = wall_clock64();
        trace[8192 + 4 * blockIdx.x + 3] = clock64();
    }
}

// ---------------------------------------------------------------- q4 filter: fragments streamed L2 -> registers, no LDS
// Round 3.  The LDS-ring kernel above runs its tile loop at ~80 % of the matrix pipe's rate but spends 20 % of the pipe on
// the fp32 accumulator-init MFMA, needs two co-resident waves per SIMD to cover its barriers / LDS-DMA issue / ds_reads
// (which then contend for the one pipe), and pays a 5 us prologue per segment for the LDS transposition of the query
// fragments.  This kernel removes all of that:
//   * ONE wave per SIMD (a 256-thread workgroup per CU, 512 registers per lane): a wave owns FOUR 32-query groups whose B
//     fragments (4 x 8 k-steps x 4 registers + the init operand = 130 registers) live in ACCUMULATION registers — MFMA
//     A/B operands may be AGPRs on gfx950, the vector ALU never touches them — while the four accumulators (64), the
//     tile's shared init values (16), the packed keys (12) and the train fragments sit in VGPRs.  Every train fragment
//     feeds four MFMAs.
//   * NO LDS, no barrier, no LDS-DMA: the prep pass stores both images in FRAGMENT ORDER ([32-row tile][9][64 lanes][16 B]),
//     so an MFMA A operand of a whole tile is ONE coalesced 1 KiB buffer_load_dwordx4 straight into the registers the MFMA
//     reads.  A ring of kQ4Ring tiles of fragments (27 loads in flight per wave) replaces the LDS ring; a fragment's
//     registers are refilled for tile t + kQ4Ring right after its last MFMA of tile t.  The four waves of a workgroup walk
//     the same tiles (L1 / L2 hits) but nothing synchronises them.  (An LDS-DMA piece costs its wave 60-185 issue cycles
//     and there is no partner wave to hide them behind: the same loop fed through the LDS ring runs 10 % slower —
//     scripts/ubench/filter_q4.hip, V = 3 against V = 4.)
//   * accumulator init as ONE K = 8 bf16 MFMA per TILE on exact bf16 triples (frag_init_operand): the scores carry the
//     pair's largest ||q||^2 instead of the query's own (same offset for all of a query's records: rankings, thresholds and
//     third-best bounds are untouched; the refine kernel works in that frame), so the init value depends on the train
//     row only and the four groups' chains take it as the C operand of their first product: 33 MFMAs per tile, not 36.
//   * the accumulators are single-buffered: a tile is two phases — the chains of groups 0, 1 with the packed-key epilogue
//     of groups 2, 3 (previous tile) as fillers between the MFMAs, then the chains of 2, 3 with the epilogue of 0, 1 —
//     ~3 VALU per MFMA gap, below the ~5 a wave alone on its SIMD can hide per 32-cycle MFMA.
// In isolation (scripts/ubench/filter_q4.hip) the loop sustains 1520-1560 algorithmic TFLOP/s against 1610 for the same
// MFMA stream with operands in registers and 1220 for the LDS-ring loop: what is left is the power-limited clock
// (~1.75 GHz under a dense MFMA stream).
// Candidate records and the partition are those of the LDS-ring kernel (a row block is 512 queries, 256 workgroups) except
// that the streams of a row block are numbered DENSELY (fill_partition_tables: nothing is written for substreams that do
// not exist); the refine kernel is shared, a flag tells it the score frame and the stream numbering.
// MFMAs are inline asm (B operands constrained to "a"): hipcc pads nothing around them.  The hazards that matter —
// an MFMA's D read by a VALU (8-pass: 12 wait states) — are covered by construction: an accumulator is first read by the
// epilogue at least two later MFMAs (>= 16 passes) after the MFMA that completed it, and the segment tail carries explicit
// s_nops.
constexpr int kQ4Rows = 4 * 4 * 32;                         // queries per workgroup: 4 waves x 4 groups x 32
constexpr int kQ4Ring = 3;                                  // tiles of train fragments in registers (fp16 bodies)

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#define SFM_MFMA_F16(acc, a, b) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b))
#define SFM_MFMA_BF16(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b))
#define SFM_MFMA_BF16_INIT(acc, a, b) asm volatile("v_mfma_f32_32x32x8_bf16 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "a"(b))
#define SFM_MFMA_BF16_REINIT(acc, a, b) asm volatile("v_mfma_f32_32x32x8_bf16 %0, %1, %2, 0" : "+v"(acc) : "v"(a), "a"(b))   // in-loop: pins the register block (see SFM_MFMA_I8_REINIT)
// first product of a chain: D = A B + C with C a DIFFERENT register block (the tile's shared init values); D is early-clobber —
// an MFMA's D may coincide with its C exactly or not at all
#define SFM_MFMA_F16_C(acc, a, b, c) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "a"(b), "v"(c))
#define SFM_MFMA_BF16_C(acc, a, b, c) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "a"(b), "v"(c))

__device__ __forceinline__ void key_insert_quad(const f32x16& a, int r, int seq /*wave-uniform*/, int vmask, int& k0, int& k1, int& k2) {
    const int m = min(min(__float_as_int(a[r]), __float_as_int(a[r + 1])), min(__float_as_int(a[r + 2]), __float_as_int(a[r + 3])));
    int key;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(m), "v"(vmask), "s"(seq));
    const int lo = min(key, k0);
    const int m1 = max(min(key, k1), min(max(key, k1), k0));
    k2 = max(min(key, k1), min(max(key, k1), k2));
    k1 = m1;
    k0 = lo;
}

__device__ __forceinline__ int key_make(const f32x16& a, int r, int seq /*wave-uniform*/, int vmask) {
    const int m = min(min(__float_as_int(a[r]), __float_as_int(a[r + 1])), min(__float_as_int(a[r + 2]), __float_as_int(a[r + 3])));
    int key;
    asm("v_and_or_b32 %0, %1, %2, %3" : "=v"(key) : "v"(m), "v"(vmask), "s"(seq));
    return key;
}
__device__ __forceinline__ void key_put(int key, int& k0, int& k1, int& k2) {
    const int lo = min(key, k0);
    const int m1 = max(min(key, k1), min(max(key, k1), k0));
    k2 = max(min(key, k1), min(max(key, k1), k2));
    k1 = m1;
    k0 = lo;
}

// KMID = false: fp16 single product, 4 groups per wave in one pass.  KMID = true: bf16 hi/mid split (three products), two
// passes of 2 groups (their hi + mid B fragments fill the same 128 AGPRs), ring of 2 tiles x 17 fragments.
// ABL != 0: dev-only timing ablations (results are WRONG): bit0 no fragment refills in the loop, bit1 no packed-key epilogue.
template <bool KMID, int ABL>
__device__ __forceinline__ void filter_q4_body(
    const unsigned char* __restrict__ qfrag, const unsigned char* __restrict__ tfrag, const unsigned char* __restrict__ qhm,
    const unsigned char* __restrict__ thm, int nq, int nq_pad, int tiles, int smax, int nsub, float* __restrict__ cand_s0,
    int* __restrict__ cand_i0, const int64_t* __restrict__ wg_begin, const int* __restrict__ rb_first, int n_rb1, int64_t s_qfrag,
    int64_t s_tfrag, int64_t s_qhm, int64_t s_thm, int64_t s_cand, const int* __restrict__ minfo, const int* __restrict__ wg_sbase) {
    constexpr int NG = KMID ? 2 : 4;                          // groups per pass
    constexpr int NPASS = KMID ? 2 : 1;
    constexpr int P = NG / 2;                                 // groups per phase
    constexpr int D = KMID ? 2 : kQ4Ring;                     // ring depth in tiles
    constexpr int NF = KMID ? 16 : 8;                         // 16-byte fragments per tile: [hi 0..7][mid 8..15] / [fp16 0..7]; + the init fragment
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int G = gridDim.x;
    const int bid = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;   // XCD-aware order (see the LDS-ring kernel)
    const int64_t u_end = wg_begin[bid + 1];
    int64_t u = wg_begin[bid];
    const int voff = lane * 16, voffi = lane * 8;
    const int qtiles = nq_pad >> 5;
    int vmask;
    asm volatile("v_mov_b32 %0, %1" : "=v"(vmask) : "s"(~kKeyMask));

    while (u < u_end) {
        const int rb = __builtin_amdgcn_readfirstlane((int)(u / tiles));      // (the 64-bit division runs on the vector ALU: say that it is uniform)
        const int t_begin = __builtin_amdgcn_readfirstlane((int)(u - (int64_t)rb * tiles));
        const int t_end = __builtin_amdgcn_readfirstlane((int)min((int64_t)tiles, t_begin + (u_end - u)));
        const int sbase = __builtin_amdgcn_readfirstlane(u == wg_begin[bid] ? wg_sbase[bid] : 0);   // number of this segment's first substream in its row block (compact streams)
        const int pb = __builtin_amdgcn_readfirstlane(n_rb1 > 0 ? rb / n_rb1 : 0);
        const int rbl = rb - pb * n_rb1;
        float* __restrict__ cand_s = cand_s0 + pb * s_cand;
        int* __restrict__ cand_i = cand_i0 + pb * s_cand;
        const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)(tfrag + pb * s_tfrag), 0, tiles * kTileFragBytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc((void*)(qfrag + pb * s_qfrag), 0, qtiles * kTileFragBytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t thrs = __builtin_amdgcn_make_buffer_rsrc((void*)(KMID ? thm + pb * s_thm : tfrag), 0, KMID ? tiles * 16 * kFragBytes : 0, 0x00020000);
        const __amdgpu_buffer_rsrc_t qhrs = __builtin_amdgcn_make_buffer_rsrc((void*)(KMID ? qhm + pb * s_qhm : qfrag), 0, KMID ? qtiles * 16 * kFragBytes : 0, 0x00020000);

#pragma unroll 1
        for (int pass = 0; pass < NPASS; ++pass) {
            const int qg0 = rbl * (kQ4Rows / 32) + wave * 4 + pass * NG;        // first 32-query tile of this pass
            const int qrow0 = qg0 * 32 + j;                                    // group g adds 32 g
            bool qok[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) qok[g] = qrow0 + 32 * g < nq;

            // ---- train fragments of the first D tiles, then the query fragments
            i32x4 fr[D][NF];
            i32x2 fri[D];
            auto load_frag = [&](int s, int f, int tile_, bool live, bool in_loop = false) {
                // fragment f (f == NF: the init fragment) of `tile` -> ring slot s (live = false: the ring runs past the segment's
                // end; the segment's last tile is fetched again — L1 / L2 hits — and never used)
                if ((ABL & 1) && in_loop) return;
                const int tile = (ABL & 4) ? t_begin : live ? tile_ : t_end - 1;      // (ABL & 4: every refill re-reads the segment's first tile — L1 hits)
                if (f == NF) {
                    fri[s] = __builtin_amdgcn_raw_buffer_load_b64(trs, voffi, __builtin_amdgcn_readfirstlane(tile * kTileFragBytes + 8 * kFragBytes), 0);
                } else if constexpr (KMID) {
                    fr[s][f] = __builtin_amdgcn_raw_buffer_load_b128(thrs, voff, __builtin_amdgcn_readfirstlane(tile * (16 * kFragBytes) + f * kFragBytes), 0);
                } else {
                    fr[s][f] = __builtin_amdgcn_raw_buffer_load_b128(trs, voff, __builtin_amdgcn_readfirstlane(tile * kTileFragBytes + f * kFragBytes), 0);
                }
            };
#pragma unroll
            for (int d = 0; d < D; ++d) {
                load_frag(d, NF, t_begin + d, t_begin + d < t_end);           // (init fragment first: it is consumed first)
#pragma unroll
                for (int f = 0; f < NF; ++f) load_frag(d, f, t_begin + d, t_begin + d < t_end);
            }

            __builtin_amdgcn_sched_barrier(0);
            // the query fragments go straight into accumulation registers (hipcc makes the loads' destinations the AGPRs the
            // tied "a" constraints below ask for); ALL of them are requested before the first is waited for
            u32x4 bq[NG][KMID ? 16 : 8];
            u32x2 bi;                                                          // the init product's query side: ||q||^2max of the pair
            {
                i32x4 tmp[NG][KMID ? 16 : 8];
#pragma unroll
                for (int g = 0; g < NG; ++g) {
#pragma unroll
                    for (int f = 0; f < (KMID ? 16 : 8); ++f)
                        tmp[g][f] = KMID ? __builtin_amdgcn_raw_buffer_load_b128(qhrs, voff, __builtin_amdgcn_readfirstlane((qg0 + g) * (16 * kFragBytes) + f * kFragBytes), 0)
                                         : __builtin_amdgcn_raw_buffer_load_b128(qrs, voff, __builtin_amdgcn_readfirstlane((qg0 + g) * kTileFragBytes + f * kFragBytes), 0);
                }
                const uint2 qm = frag_init_operand(__int_as_float(minfo[kMinfoQmax + pb]), true, h);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < NG; ++g) {
#pragma unroll
                    for (int f = 0; f < (KMID ? 16 : 8); ++f) asm volatile("" : "=a"(bq[g][f]) : "0"(__builtin_bit_cast(u32x4, tmp[g][f])));
                }
                asm volatile("" : "=a"(bi) : "0"(u32x2{qm.x, qm.y}));
            }
            asm volatile("s_nop 4");                                           // v_accvgpr_write -> MFMA operand

            int k0[NG], k1[NG], k2[NG];
#pragma unroll
            for (int g = 0; g < NG; ++g) k0[g] = k1[g] = k2[g] = kKeyInf;
            int sub = 0, sub_t0 = t_begin;
            auto flush = [&](int sb, int st0) {
                // (cold: once per 64 tiles.  The record addresses are formed HERE from an opaque copy of the query row: left to
                // the optimiser their loop-invariant parts are hoisted and pin registers across the tile loop — the kernel must
                // stay within 240 VGPRs + 144 AGPRs = 384 registers so that a 128-register refine wave of another launch set
                // fits beside a filter wave on the same SIMD)
                int qr = qrow0;
                asm volatile("" : "+v"(qr));
#pragma unroll
                for (int g = 0; g < NG; ++g) {
                    if (qok[g]) {
                        const int64_t ob = ((int64_t)(qr + 32 * g) * (2 * smax * nsub) + (int64_t)(sbase + sb) * 2 + h) * 3;
                        flush_keys<true>(k0[g], k1[g], k2[g], st0, h, cand_s + ob, cand_i + ob);
                    }
                    k0[g] = k1[g] = k2[g] = kKeyInf;
                }
            };
            f32x16 acc[NG];
            // the first tile's "previous tile" (phase A runs the epilogue of the second half's groups) is +inf everywhere: its
            // insertions leave the keys untouched
#pragma unroll
            for (int g = P; g < NG; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][r] = kInf;
            // ||t||^2 + ||q||^2max of the tile about to run: ONE init MFMA per tile, shared by all groups as the C operand of
            // their chains' first products.  It is issued a phase ahead (right after the last chain of the previous tile
            // has read the previous values), so nothing ever waits for it; the segment's first one is followed by nops.
            f32x16 cinit;
            SFM_MFMA_BF16_INIT(cinit, __builtin_bit_cast(u32x2, fri[0]), bi);
            asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");

            // k-step st of group g: one fp16 product, or the three bf16 products hi.hi + hi.mid + mid.hi
            auto tile = [&](int t, auto slot_c) {
                constexpr int S = decltype(slot_c)::value;
                const bool more = t + D < t_end;
                const int seq_prev = __builtin_amdgcn_readfirstlane(max((t - 1) - sub_t0, 0) << 2);
                // ---- phase A: chains of the first half's groups; epilogue of the second half's groups (previous tile)
#pragma unroll
                for (int st = 0; st < 8; ++st) {
                    __builtin_amdgcn_sched_barrier(0);
                    const u32x4 a = __builtin_bit_cast(u32x4, fr[S][st]);
                    if constexpr (KMID) {
                        const u32x4 am = __builtin_bit_cast(u32x4, fr[S][8 + st]);
                        if (st == 0) SFM_MFMA_BF16_C(acc[0], a, bq[0][st], cinit);
                        else SFM_MFMA_BF16(acc[0], a, bq[0][st]);
                        SFM_MFMA_BF16(acc[0], a, bq[0][8 + st]);
                        SFM_MFMA_BF16(acc[0], am, bq[0][st]);
                        if (st & 1) key_insert_quad(acc[1], 4 * (st >> 1), seq_prev + (st >> 1), vmask, k0[1], k1[1], k2[1]);
                    } else {
                        // the epilogue's six VALU per quad are dealt to BOTH gaps (three after each MFMA): a wave alone on its SIMD
                        // hides ~5 single-issue instructions per 32-cycle MFMA, and none behind an MFMA it is still waiting to issue
                        const int g = 2 + (st >> 2);
                        if (st == 0) SFM_MFMA_F16_C(acc[0], a, bq[0][st], cinit);
                        else SFM_MFMA_F16(acc[0], a, bq[0][st]);
                        int key = 0;
                        if (!(ABL & 2)) key = key_make(acc[g], 4 * (st & 3), seq_prev + (st & 3), vmask);
                        __builtin_amdgcn_sched_barrier(0);
                        if (st == 0) SFM_MFMA_F16_C(acc[1], a, bq[1][st], cinit);
                        else SFM_MFMA_F16(acc[1], a, bq[1][st]);
                        if (ABL & 2) {
                            if (st == 0) k0[2] = min(k0[2], __float_as_int(acc[2][0]) + __float_as_int(acc[3][5]));
                        } else
                            key_put(key, k0[g], k1[g], k2[g]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
                // ---- substream boundary: every group's keys now cover exactly the tiles before t
                if (t - sub_t0 == kSubTiles) {
                    flush(sub, sub_t0);
                    ++sub;
                    sub_t0 = t;
                }
                const int seq_cur = __builtin_amdgcn_readfirstlane((t - sub_t0) << 2);
                // ---- phase B: chains of the second half's groups; epilogue of the first half's (this tile); the fragments die
                // one by one and are refilled for tile t + D
                __builtin_amdgcn_sched_barrier(0);
                load_frag(S, NF, t + D, more, true);
#pragma unroll
                for (int st = 0; st < 8; ++st) {
                    __builtin_amdgcn_sched_barrier(0);
                    const u32x4 a = __builtin_bit_cast(u32x4, fr[S][st]);
                    if constexpr (KMID) {
                        const u32x4 am = __builtin_bit_cast(u32x4, fr[S][8 + st]);
                        if (st == 0) SFM_MFMA_BF16_C(acc[1], a, bq[1][st], cinit);
                        else SFM_MFMA_BF16(acc[1], a, bq[1][st]);
                        SFM_MFMA_BF16(acc[1], a, bq[1][8 + st]);
                        SFM_MFMA_BF16(acc[1], am, bq[1][st]);
                        if (st == 1) SFM_MFMA_BF16_REINIT(cinit, __builtin_bit_cast(u32x2, fri[(S + 1) % D]), bi);   // the NEXT tile's
                    } else {
                        const int g = st >> 2;
                        if (st == 0) SFM_MFMA_F16_C(acc[2], a, bq[2][st], cinit);
                        else SFM_MFMA_F16(acc[2], a, bq[2][st]);
                        int key = 0;
                        if (!(ABL & 2)) key = key_make(acc[g], 4 * (st & 3), seq_cur + (st & 3), vmask);
                        __builtin_amdgcn_sched_barrier(0);
                        if (st == 0) SFM_MFMA_F16_C(acc[3], a, bq[3][st], cinit);
                        else SFM_MFMA_F16(acc[3], a, bq[3][st]);
                        if (st == 1) SFM_MFMA_BF16_REINIT(cinit, __builtin_bit_cast(u32x2, fri[(S + 1) % D]), bi);   // the NEXT tile's (two MFMAs after the last reader)
                        __builtin_amdgcn_sched_barrier(0);
                        load_frag(S, st, t + D, more, true);
                        if (ABL & 2) {
                            if (st == 0) k0[0] = min(k0[0], __float_as_int(acc[0][0]) + __float_as_int(acc[1][5]));
                        } else
                            key_put(key, k0[g], k1[g], k2[g]);
                    }
                    if constexpr (KMID) {
                        __builtin_amdgcn_sched_barrier(0);
                        load_frag(S, st, t + D, more, true);
                        load_frag(S, 8 + st, t + D, more, true);
                        if (st & 1) key_insert_quad(acc[0], 4 * (st >> 1), seq_cur + (st >> 1), vmask, k0[0], k1[0], k2[0]);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            // Whole rounds of D tiles first, the remainder after the loop: with conditional tiles INSIDE the loop hipcc's
            // waitcnt pass has to merge "only the first tile ran" into the back edge and makes every round's first tile wait for
            // ALL outstanding loads (vmcnt(8) .. (0) instead of (26) .. (18)): the ring's prefetch distance would be zero there.
            int t = t_begin;
            for (; t + D <= t_end; t += D) {
                tile(t, std::integral_constant<int, 0>{});
                tile(t + 1, std::integral_constant<int, 1>{});
                if constexpr (D > 2) tile(t + 2, std::integral_constant<int, (D > 2 ? 2 : 0)>{});
            }
            if (t < t_end) {
                tile(t, std::integral_constant<int, 0>{});
                if constexpr (D > 2)
                    if (t + 1 < t_end) tile(t + 1, std::integral_constant<int, 1>{});
            }
            // epilogue of the last tile's second-half groups (the MFMAs that completed them were the last instructions issued)
            asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
            if (t_end > t_begin) {
                const int seq = __builtin_amdgcn_readfirstlane(((t_end - 1) - sub_t0) << 2);
#pragma unroll
                for (int g = P; g < NG; ++g)
#pragma unroll
                    for (int r = 0; r < 16; r += 4) key_insert_quad(acc[g], r, seq + (r >> 2), vmask, k0[g], k1[g], k2[g]);
            }
            flush(sub, sub_t0);
        }
        u += t_end - t_begin;
    }
}


// ---------------------------------------------------------------- exact-integer body: v_mfma_i32_32x32x32_i8
// The third body of the q4 kernel (same launch: 256 workgroups of 4 waves, one wave per SIMD), chosen on the device when
// every value of the batch is a u8 integer (minfo[kMinfoI8]).  Differences from the fp16 body:
//   * 4 product MFMAs per 32 x 32 tile and group instead of 8, so the loop would be bound by everything that is NOT an MFMA:
//     a wave owns EIGHT 32-query groups (B fragments: 8 x 4 x 4 = 128 AGPRs), every train fragment (one 1 KiB load) feeds 8
//     MFMAs — with 4 groups the four waves' fragment loads (20 KiB per tile through the CU's 64 B/clk L1 path) cost 34 of a
//     tile-group's 128 pipe cycles (scripts/ubench/filter_i8.hip, profiles/r04_ubench_filter_i8_skeleton.txt);
//   * a candidate record is HALF of a lane's 16 accumulator registers — registers 8 e .. 8 e + 7 = the 8 train rows
//     16 e + 8 (r >> 2) + 4 h + (r & 3) of a tile — so the epilogue is 2 x (4 min3 / min + 1 pack + 3 insert) = 16 VALU per group
//     and tile, FOUR per MFMA gap (quads: six, more than a wave alone on its SIMD hides; whole-lane records: three, but the
//     refine kernel then reads 16 rows per record — its L1 traffic doubled and it lost more than the filter gained);
//     the key's low 8 bits are (tile inside a 128-tile substream) << 1 | e;
//   * i32 accumulation: scores are exact (up to the parity bit, see the images' comment), keys hold the whole score
//     (acc << 8 | tile), the accumulator init is one i8 MFMA per tile shared by the 8 groups (C operand of their first products);
//   * records are the three packed keys per stream and half-wave, ONE 16-byte slot per query ([row block][stream][h][1024
//     queries][3 keys + pad]: a store instruction writes 512 contiguous bytes per half-wave) — see the flush below.
// Row blocks are 1024 queries; the partition tables are those of the i8 plan (wg_begin8 ...).
typedef int i32x16 __attribute__((ext_vector_type(16)));
#define SFM_MFMA_I8(acc, a, b) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "a"(b))
#define SFM_MFMA_I8_C(acc, a, b, c) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %3" : "=&v"(acc) : "v"(a), "a"(b), "v"(c))
#define SFM_MFMA_I8_INIT(acc, a, b) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "=v"(acc) : "v"(a), "a"(b))
// The in-loop form declares the old values read ("+v") although the instruction does not read them: the tile's init values
// must keep ONE register block for the whole loop.  With "=v" the old block is dead after the last chain's first MFMA has
// been ISSUED, and hipcc hands its registers to the epilogue's temporaries — written by the vector ALU while that MFMA is still
// reading them as SrcC (no hardware interlock: results of a whole query group went wrong, found by tests/test_gpu_knn_i8.py).
#define SFM_MFMA_I8_REINIT(acc, a, b) asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, 0" : "+v"(acc) : "v"(a), "a"(b))

__device__ __forceinline__ int key_pack_i8(int m, int seq /*wave-uniform*/) {
    int key;
    asm("v_lshl_or_b32 %0, %1, 8, %2" : "=v"(key) : "v"(m), "s"(seq));
    return key;
}
__device__ __forceinline__ int imin3(int a, int b, int c) { return min(min(a, b), c); }   // v_min3_i32

template <int ABL>
__device__ __forceinline__ void filter_i8_body(
    const unsigned char* __restrict__ qi8, const unsigned char* __restrict__ ti8, int nq, int nq_pad, int nt, int tiles, int nstr /*stream slots per row block*/,
    int* __restrict__ keys0, int* __restrict__ sttab0, const int64_t* __restrict__ wg_begin, int n_rb1, int64_t s_qi8, int64_t s_ti8, int64_t s_keys,
    const int* __restrict__ wg_sbase, int G, int sub_tiles /*tiles per substream (wave-uniform): kI8SubTiles, or kQ8SubTiles on quantised data*/) {
    constexpr int NG = kI8Groups, P = NG / 2, D = 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int bid = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;   // XCD-aware order
    const int64_t u_end = wg_begin[bid + 1];
    int64_t u = wg_begin[bid];
    const int voff = lane * 16;
    const int qtiles = nq_pad >> 5;

    while (u < u_end) {
        const int rb = __builtin_amdgcn_readfirstlane((int)(u / tiles));
        const int t_begin = __builtin_amdgcn_readfirstlane((int)(u - (int64_t)rb * tiles));
        const int t_end = __builtin_amdgcn_readfirstlane((int)min((int64_t)tiles, t_begin + (u_end - u)));
        const int sbase = __builtin_amdgcn_readfirstlane(u == wg_begin[bid] ? wg_sbase[bid] : 0);
        const int pb = __builtin_amdgcn_readfirstlane(n_rb1 > 0 ? rb / n_rb1 : 0);
        const int rbl = rb - pb * n_rb1;
        int* __restrict__ keys = keys0 + pb * s_keys;
        int* __restrict__ sttab = sttab0 + (int64_t)rb * nstr * 2;               // (global row block: the table spans the batch)
        const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc((void*)(ti8 + pb * s_ti8), 0, tiles * kI8TileBytes, 0x00020000);
        const __amdgpu_buffer_rsrc_t qrs = __builtin_amdgcn_make_buffer_rsrc((void*)(qi8 + pb * s_qi8), 0, qtiles * kI8QTileBytes, 0x00020000);
        const int qg0 = rbl * (kI8Rows / 32) + wave * NG;                      // first 32-query tile of this wave
        const int qloc0 = wave * NG * 32 + j;                                  // query inside the row block; group g adds 32 g
        bool qok[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) qok[g] = qg0 * 32 + j + 32 * g < nq;

        // ---- train fragments of the first D tiles (init fragment first: it is consumed first), then the query fragments
        i32x4 fr[D][5];
        auto load_frag = [&](int s_, int f, int tile_, bool live, bool in_loop = false) {
            if ((ABL & 1) && in_loop) return;
            const int tile = live ? tile_ : t_end - 1;
            fr[s_][f] = __builtin_amdgcn_raw_buffer_load_b128(trs, voff, __builtin_amdgcn_readfirstlane(tile * kI8TileBytes + f * 1024), 0);
        };
#pragma unroll
        for (int d = 0; d < D; ++d) {
            load_frag(d, 4, t_begin + d, t_begin + d < t_end);
#pragma unroll
            for (int f = 0; f < 4; ++f) load_frag(d, f, t_begin + d, t_begin + d < t_end);
        }
        __builtin_amdgcn_sched_barrier(0);
        u32x4 bq[NG][4], bi;
        {
            i32x4 tmp[NG][4];
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int f = 0; f < 4; ++f)
                    tmp[g][f] = __builtin_amdgcn_raw_buffer_load_b128(qrs, voff, __builtin_amdgcn_readfirstlane((qg0 + g) * kI8QTileBytes + f * 1024), 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < NG; ++g)
#pragma unroll
                for (int f = 0; f < 4; ++f) asm volatile("" : "=a"(bq[g][f]) : "0"(__builtin_bit_cast(u32x4, tmp[g][f])));
            // query side of the init product: slot 0 (byte 0 of the h = 0 lanes) = 1, the other 31 slots = -128
            asm volatile("" : "=a"(bi) : "0"(u32x4{h ? 0x80808080u : 0x80808001u, 0x80808080u, 0x80808080u, 0x80808080u}));
        }
        asm volatile("s_nop 4");                                               // v_accvgpr_write -> MFMA operand

        int k0[NG], k1[NG], k2[NG];
#pragma unroll
        for (int g = 0; g < NG; ++g) k0[g] = k1[g] = k2[g] = INT_MAX;
        int sub = 0, sub_t0 = t_begin;
        auto flush = [&](int sb, int st0, int st1) {
            // cold: once per substream.  keys[((rbl * nstr + stream) * 2 + h) * 1024 + query-in-block][4]: the stream's three keys
            // of a query as ONE 16-byte store (the fourth word is padding) — a flush is 8 stores per lane instead of 24, and the
            // fragment wait that follows has to sit their completion out (stores and loads share vmcnt): 87.2 -> 84.0 us per batch
            // of 8 with 32-tile substreams (measured with a one-store-per-group build before the layout was changed)
            int ql = qloc0;
            asm volatile("" : "+v"(ql));
            const int64_t ob = ((int64_t)(rbl * nstr + sbase + sb) * 2 + h) * kI8Rows * 4;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (qok[g]) *reinterpret_cast<int4*>(keys + ob + (int64_t)(ql + 32 * g) * 4) = make_int4(k0[g], k1[g], k2[g], INT_MAX);
                k0[g] = k1[g] = k2[g] = INT_MAX;
            }
            if (threadIdx.x == 0) {                                            // the stream's tile range, for the refine kernel's decode / rescan
                sttab[2 * (sbase + sb)] = st0;
                sttab[2 * (sbase + sb) + 1] = st1 - st0;
            }
        };
        i32x16 acc[NG], cinit;
        // the first tile's "previous tile" (phase A runs the epilogue of groups P .. NG-1): values whose keys are >= kKeyEmptyI
#pragma unroll
        for (int g = P; g < NG; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[g][r] = 0x7FFFFF;
        SFM_MFMA_I8_INIT(cinit, __builtin_bit_cast(u32x4, fr[0][4]), bi);
        asm volatile("s_nop 15\n\ts_nop 7" ::: "memory");

        // Rows past the last train (the train image is padded to whole tiles) must never make a key: a record's score has to be
        // the score of a REAL row of the record (the refine kernel's selection argument counts distinct real rows).  The last,
        // partial tile's accumulators are masked before their epilogue — a wave-uniform branch taken once per segment at most.
        const int nrem = nt & 31;                                             // real rows of the partial tile (0: none is partial)
        auto mask_tail = [&](i32x16& a) {
#pragma unroll
            for (int r = 0; r < 16; ++r) a[r] = 8 * (r >> 2) + 4 * h + (r & 3) >= nrem ? 0x7FFFFF : a[r];
        };
        // one group's epilogue (16 VALU) in four pieces, one per MFMA gap of a k-step.  The two 8-register records' min trees are
        // interleaved (a wave alone on its SIMD stalls on every dependent vector-ALU pair: independent neighbours fill the slots)
        int m0 = 0, m1 = 0, n0 = 0, n1 = 0, key = 0, key2 = 0;
        auto epi = [&](int piece, i32x16& a, int seq /*(tile in substream) << 1*/, int& e0, int& e1, int& e2, bool partial) {
            if (ABL & 2) { if (piece == 0) e0 = min(e0, a[0] + a[15]); return; }
            if (piece == 0 && partial) mask_tail(a);
            if (piece == 0) { m0 = imin3(a[0], a[1], a[2]); n0 = imin3(a[8], a[9], a[10]); m1 = imin3(a[3], a[4], a[5]); n1 = imin3(a[11], a[12], a[13]); }
            else if (piece == 1) { m0 = imin3(m0, m1, a[6]); n0 = imin3(n0, n1, a[14]); m0 = min(m0, a[7]); n0 = min(n0, a[15]); }
            else if (piece == 2) { key = key_pack_i8(m0, seq); key2 = key_pack_i8(n0, seq | 1); key_put(key, e0, e1, e2); }
            else key_put(key2, e0, e1, e2);
        };
        auto tile = [&](int t, auto slot_c) {
            constexpr int S = decltype(slot_c)::value;
            const bool more = t + D < t_end;
            const int seq_prev = __builtin_amdgcn_readfirstlane(max((t - 1) - sub_t0, 0) << 1);
            const bool part_cur = nrem != 0 && t + 1 == tiles;      // (wave-uniform) t is the train image's last, partial tile; its second-half groups' epilogue is the tail's
            // ---- phase A: chains of groups 0 .. 3; k-step st carries the epilogue of group P + st (previous tile)
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const u32x4 a = __builtin_bit_cast(u32x4, fr[S][st]);
#pragma unroll
                for (int g = 0; g < P; ++g) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (st == 0) SFM_MFMA_I8_C(acc[g], a, bq[g][st], cinit);
                    else SFM_MFMA_I8(acc[g], a, bq[g][st]);
                    epi(g, acc[P + st], seq_prev, k0[P + st], k1[P + st], k2[P + st], false);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // ---- substream boundary: every group's keys now cover exactly the tiles before t
            if (t - sub_t0 == sub_tiles) {
                flush(sub, sub_t0, t);
                ++sub;
                sub_t0 = t;
            }
            const int seq_cur = __builtin_amdgcn_readfirstlane((t - sub_t0) << 1);
            // ---- phase B: chains of groups 4 .. 7; epilogue of groups 0 .. 3 (this tile); the fragments die one by one and are
            // refilled for tile t + D; the NEXT tile's init MFMA two MFMAs after the last reader of the current values
            __builtin_amdgcn_sched_barrier(0);
            load_frag(S, 4, t + D, more, true);
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                const u32x4 a = __builtin_bit_cast(u32x4, fr[S][st]);
#pragma unroll
                for (int g = 0; g < P; ++g) {
                    __builtin_amdgcn_sched_barrier(0);
                    if (st == 0) SFM_MFMA_I8_C(acc[P + g], a, bq[P + g][st], cinit);
                    else SFM_MFMA_I8(acc[P + g], a, bq[P + g][st]);
                    if (st == 1 && g == 1) SFM_MFMA_I8_REINIT(cinit, __builtin_bit_cast(u32x4, fr[(S + 1) % D][4]), bi);
                    epi(g, acc[st], seq_cur, k0[st], k1[st], k2[st], part_cur);
                    __builtin_amdgcn_sched_barrier(0);
                }
                load_frag(S, st, t + D, more, true);
            }
        };
        // whole rounds of D tiles, the remainder after the loop (see the fp16 body: hipcc's waitcnt merge at the back edge)
        int t = t_begin;
        for (; t + D <= t_end; t += D) {
            tile(t, std::integral_constant<int, 0>{});
            tile(t + 1, std::integral_constant<int, 1>{});
        }
        if (t < t_end) tile(t, std::integral_constant<int, 0>{});
        // epilogue of the last tile's groups P .. NG-1 (the MFMAs that completed them were the last instructions issued)
        asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
        if (t_end > t_begin) {
            const int seq = __builtin_amdgcn_readfirstlane(((t_end - 1) - sub_t0) << 1);
#pragma unroll
            for (int g = P; g < NG; ++g)
#pragma unroll
                for (int piece = 0; piece < 4; ++piece) epi(piece, acc[g], seq, k0[g], k1[g], k2[g], nrem != 0 && t_end == tiles);
        }
        flush(sub, sub_t0, t_end);
        u += t_end - t_begin;
    }
}

template <int ABL>
__global__ __launch_bounds__(256, 1) void knn_filter_q4_kernel(
    const unsigned char* __restrict__ qfrag, const unsigned char* __restrict__ tfrag, const unsigned char* __restrict__ qhm,
    const unsigned char* __restrict__ thm, int nq, int nq_pad, int tiles, int smax, int nsub, const int* __restrict__ midflag,
    const float* __restrict__ bmax, int force_mode, float* __restrict__ cand_s, int* __restrict__ cand_i,
    const int64_t* __restrict__ wg_begin, const int* __restrict__ rb_first, int n_rb1, int n_pairs, int64_t s_qfrag, int64_t s_tfrag,
    int64_t s_qhm, int64_t s_thm, int64_t s_cand, int* __restrict__ minfo, const float* __restrict__ bmaxerr,
    const int* __restrict__ wg_sbase, long long* __restrict__ trace,
    // the exact-integer body (qi8 == null: not planned): byte images, key records, its own partition (1024-query row blocks)
    const unsigned char* __restrict__ qi8, const unsigned char* __restrict__ ti8, int64_t s_qi8, int64_t s_ti8, int* __restrict__ keys8,
    int* __restrict__ sttab8, int64_t s_keys8, int nstr8, const int64_t* __restrict__ wg_begin8, int n_rb1_8, const int* __restrict__ wg_sbase8, int G8,
    int nt8) {
    if (trace && threadIdx.x == 0) {
        trace[4 * blockIdx.x + 0] = wall_clock64();
        trace[4 * blockIdx.x + 2] = __builtin_amdgcn_s_getreg(0xF804);
        trace[4 * blockIdx.x + 3] = __builtin_amdgcn_s_getreg(0xF814);
        trace[8192 + 4 * blockIdx.x + 2] = clock64();
    }
    if (qi8 && minfo[kMinfoI8]) {                          // (uniform: one scalar load)
        if ((int)blockIdx.x < G8) filter_i8_body<ABL>(qi8, ti8, nq, nq_pad, nt8, tiles, nstr8, keys8, sttab8, wg_begin8, n_rb1_8, s_qi8, s_ti8, s_keys8, wg_sbase8, G8, minfo[kMinfoSub8]);
        if (trace && threadIdx.x == 0) {
            trace[4 * blockIdx.x + 1] = wall_clock64();
            trace[8192 + 4 * blockIdx.x + 3] = clock64();
        }
        return;
    }
    const bool need_mid = (force_mode >= 0 ? force_mode : minfo[kMinfoBatchMode]) == kModeSplit;   // (reduced by knn_split_images_kernel)
    if (need_mid)
        filter_q4_body<true, 0>(qfrag, tfrag, qhm, thm, nq, nq_pad, tiles, smax, nsub, cand_s, cand_i, wg_begin, rb_first, n_rb1, s_qfrag, s_tfrag, s_qhm,
                             s_thm, s_cand, minfo, wg_sbase);
    else
        filter_q4_body<false, ABL>(qfrag, tfrag, qhm, thm, nq, nq_pad, tiles, smax, nsub, cand_s, cand_i, wg_begin, rb_first, n_rb1, s_qfrag, s_tfrag, s_qhm,
                              s_thm, s_cand, minfo, wg_sbase);
    if (trace && threadIdx.x == 0) {
        trace[4 * blockIdx.x + 1] = wall_clock64();
        trace[8192 + 4 * blockIdx.x + 3] = clock64();
    }
}

// ---------------------------------------------------------------- exact direct-form distance
// Reference arithmetic (OpenCV normL2Sqr_, SSE2 path): two 4-lane accumulators over blocks of 8,
// mul and add separately rounded; lanes summed as (d0+d1) then ((s0+s1)+s2)+s3.
// Compiled with -ffp-contract=off so none of this fuses.  The evaluators below (lane pair / quad per train) keep that
// order: which hardware lane owns which accumulator lane is free, the order of the 16 adds per accumulator is not.

// value of the next lane inside an aligned pair (valid on even lanes) / of lane 3 on lane 2 of a quad
__device__ __forceinline__ float lane_odd_neighbour(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xF5 /*quad_perm [1,1,3,3]*/, 0xF, 0xF, false));
}

__device__ __forceinline__ int imed3(int a, int b, int c) { return max(min(a, b), min(max(a, b), c)); }   // v_med3_i32

struct Best2 {
    float d[2];     // sqrtf distance
    int i[2];
};

__device__ __forceinline__ bool key_less(float da, int ia, float db, int ib) {
    return da < db || (da == db && ia < ib);
}

__device__ __forceinline__ void best2_insert(Best2& b, float d, int i) {
    if (key_less(d, i, b.d[0], b.i[0])) {
        b.d[1] = b.d[0]; b.i[1] = b.i[0];
        b.d[0] = d; b.i[0] = i;
    } else if (key_less(d, i, b.d[1], b.i[1])) {
        b.d[1] = d; b.i[1] = i;
    }
}

// Upper bound of the exact d^2 behind a float32 distance d = RN(sqrt(d^2)): d^2 <= (d / (1 - 2^-24))^2 < d*d * (1 + 2^-22)
// (the certificate needs d2^2 from above; carrying the squares next to the distances cost a third of the merge network).
__device__ __forceinline__ double dsq_upper(float d) { return (double)d * (double)d * (1.0 + 2.384185791015625e-07); }

// ---------------------------------------------------------------- refine
// Slack coefficients of the refine kernel's certificate: |filter score - exact d^2| <= eps0 = c (|q|+|t|max)^2 + op, with c =
//   rounding: fp32 arithmetic around the product — ||t||^2, ||q||^2 in the prep pass (9 roundings deep, relative to the
//             norm), ||q||^2 again in this kernel (13), the screen's v_dot2 chain (17 u of 2|q||t|) and its two adds, the
//             direct-form sums the answer is defined by (23 u d^2) and one ulp of sqrtf in d^2 terms (8 u, the tie margin
//             of hidden rows): <= 75 u (|q|+|t|)^2 -> 200 u.  (fp32-MFMA filter: its 64-MFMA chain too -> 600 u.)
//   MFMA:     the accumulations inside the 16-bit MFMA chain.  One v_mfma_f32_32x32x16_{f16,bf16} returns c + sum a_k b_k
//             within E * 2^-24 (|c| + sum |a_k b_k|); measured on gfx950 (sfm_selftest_mfma_accumulation, 3.4e9 samples
//             per regime): E <= 2.0 for operands of similar magnitude — the filter's regime — and <= 7.1 for exponents
//             spread over 2^16 with cancellation; the certificate assumes E = 16 (tests/test_gpu_knn.py holds the device
//             to E <= 8).  A chain is 8 product MFMAs + the init (24 + 1 in the split mode), each with
//             |c| + sum |a b| <= 1.5 N^2: 9 * 16 * 1.5 = 216 -> 256 u (split: 600 -> 640 u).  (Round 1 assumed 1600 u.)
//   split:    2 * 3.05 * 2^-16 / 4  neglected (mid.mid, delta) product terms, |q||t| <= N^2/4
//   half:     op is bounded from the data (see the kernel); without the measured residuals the worst case
//             2 * (2^-10 + 2^-22) / 4 (both operands rounded to 11 bits) plus, ABSOLUTE, 2 * 2^-14 * sqrt(128) * (|q|+|t|max)
//             for elements below the fp16 normal range.
// The packed-key TRUNCATION of a record's score (kKeyBits low mantissa bits cleared: record score in (s (1 - 2^-15), s]) is
// not part of eps0 — it is one-sided and relative to the score, so it only widens the record threshold by 2^-15 m2 (see thr).
constexpr float kU = 5.9604645e-8f;
constexpr float kEpsF32 = 600.f * kU;
static_assert(kKeyBits == 8, "kKeyTrunc = 2^(kKeyBits-23)");
constexpr float kKeyTrunc = 3.0517578e-5f * 1.002f;
constexpr float kEpsRound = 200.f * kU;                      // fp32 rounding around the product (budget: docs/knn.md "Error budget of the certificate")
constexpr float kChainHalf = 256.f * kU, kChainSplit = 640.f * kU;   // MFMA chain at E = 16 units per MFMA; x chain_scale (mfma_chain_scale)
constexpr float kEpsSplitOp = 2.33e-5f;
constexpr float kEpsHalfOp = 4.8840e-4f;
constexpr float kEpsHalfAbs = 1.3811e-3f;
constexpr int kModeF32 = 3;   // fp32-MFMA filter (host-selected)

__device__ __forceinline__ void best2_insert_unique(Best2& b, float d, int i) {
    if (i == b.i[0] || i == b.i[1]) return;          // the same train seen twice (a stream's own top-3 are rescanned)
    best2_insert(b, d, i);
}

template <int M>
__device__ __forceinline__ void best2_exchange_step(Best2& b) {
    Best2 o;
    o.d[0] = lane_xor<M>(b.d[0]); o.i[0] = lane_xor<M>(b.i[0]);
    o.d[1] = lane_xor<M>(b.d[1]); o.i[1] = lane_xor<M>(b.i[1]);
    best2_insert_unique(b, o.d[0], o.i[0]);
    best2_insert_unique(b, o.d[1], o.i[1]);
}

template <int WIDTH>
__device__ __forceinline__ void best2_group_reduce(Best2& b) {      // all lanes of a WIDTH-lane group end up equal
    if constexpr (WIDTH >= 64) best2_exchange_step<32>(b);
    if constexpr (WIDTH >= 32) best2_exchange_step<16>(b);
    if constexpr (WIDTH >= 16) best2_exchange_step<8>(b);
    if constexpr (WIDTH >= 8) best2_exchange_step<4>(b);
    best2_exchange_step<2>(b);
    best2_exchange_step<1>(b);
}

// Exact d^2 with FOUR lanes per train (a quad).  Lane j loads the float4 at elements 16n + 4j (n = 0..7), so a quad
// reads 64 contiguous bytes per instruction (a lane-per-row or 8-lane layout touches 2-4x as many cache lines per
// instruction, and the L1 tag rate is what bounds these kernels).  Element 16n + 4j + c belongs to accumulator lane
// a = 4(j&1) + c of the reference's 2x4-lane order at step i = 2n + (j>>1): the running sums alternate between the
// quad's lower and upper lane pair, acc <- swap_pairs(acc) + x, 16 steps, each add rounded exactly as the
// reference's.  All four lanes execute every step; a pair's "off" steps produce values that are overwritten before
// anyone reads them.  Result valid on lane j == 2.
__device__ __forceinline__ float quad_swap(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E /*quad_perm [2,3,0,1]*/, 0xF, 0xF, false));
}

// The quad evaluation for R trains at once: all 8R row fetches are issued before the first add (the kernel is bound by
// vector-ALU issue and memory round trips, not arithmetic).  NULLABLE: trow[r] == nullptr skips row r.  Differences and
// squares two elements per instruction (v_pk_add_f32 / v_pk_mul_f32: each half rounds exactly like the scalar op); the
// running sums stay scalar (their DPP operand is per instruction).
template <int R, bool NULLABLE>
__device__ __forceinline__ void exact_l2sq_quad_rows(const float* __restrict__ qrow /*LDS*/, const float* const (&trow)[R], int j,
                                                     float (&out)[R]) {
    float4 t[R][8];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int n = 0; n < 8; ++n)
            t[r][n] = (!NULLABLE || trow[r]) ? *reinterpret_cast<const float4*>(trow[r] + 16 * n + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        const float4 qv = *reinterpret_cast<const float4*>(qrow + 16 * n + 4 * j);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const f32x2 d0 = f32x2{qv.x, qv.y} - f32x2{t[r][n].x, t[r][n].y}, d1 = f32x2{qv.z, qv.w} - f32x2{t[r][n].z, t[r][n].w};
            const f32x2 x0 = d0 * d0, x1 = d1 * d1;
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                acc[r].x = quad_swap(acc[r].x) + x0.x;
                acc[r].y = quad_swap(acc[r].y) + x0.y;
                acc[r].z = quad_swap(acc[r].z) + x1.x;
                acc[r].w = quad_swap(acc[r].w) + x1.y;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const float s0 = acc[r].x + lane_odd_neighbour(acc[r].x), s1 = acc[r].y + lane_odd_neighbour(acc[r].y);
        const float s2 = acc[r].z + lane_odd_neighbour(acc[r].z), s3 = acc[r].w + lane_odd_neighbour(acc[r].w);
        out[r] = ((s0 + s1) + s2) + s3;
    }
}

// The refine kernel's merge over a query's 16 lanes: only lane 2 of every quad ever inserts evaluated rows (the other lanes
// hold what all 16 agreed on earlier, a subset of lane 2's), so the two in-quad exchange steps are one broadcast of lane 2.
__device__ __forceinline__ void best2_reduce16_from_quad_lane2(Best2& b) {
    auto bc = [](int v) { return __builtin_amdgcn_update_dpp(0, v, 0xAA /*quad_perm [2,2,2,2]*/, 0xF, 0xF, false); };
    b.d[0] = __int_as_float(bc(__float_as_int(b.d[0]))); b.i[0] = bc(b.i[0]);
    b.d[1] = __int_as_float(bc(__float_as_int(b.d[1]))); b.i[1] = bc(b.i[1]);
    best2_exchange_step<8>(b);
    best2_exchange_step<4>(b);
}

constexpr int kRatioBlock = 1024; // queries per ratio workgroup (256 threads x 4 consecutive queries)
constexpr int kRefQ = 16;        // queries per refine workgroup: 16 lanes each
constexpr int kQbStride = 144;   // integer bodies: bytes between the queries' byte rows in LDS (128 + 16: the four queries of a wave read four different bank groups;
                                 // at 128 queries 0 / 2 and 1 / 3 collided: 9.8e5 bank conflicts per launch, profiles/r05_knn_pmc.md)
static_assert(kRefQ * kQbStride <= kRefQ * 128 * 2, "the byte rows live in the fp16 body's qhalf / qrows arrays");
constexpr int kS1 = 6;           // candidate records per lane fetched up front by sweep 1 (96 per query)
constexpr int kRescanRows = 1;   // trains a quad has in flight during a rescan (32 VGPRs each)
constexpr int kRefItems = 512;   // rescan work list (query, stream); more → 16 candidate slots per query at a time (<= 96)
constexpr int kPreRows = 2;      // trains a quad has in flight during the rescan's fp16 prefilter (16 VGPRs each)
constexpr int kHotPre = 2;       // rows a lane quad has in flight in the hot-path fp16 screen (16 VGPRs each)
constexpr int kRecCap = 112;     // record list per query
constexpr int kQualCap = 144;    // exact-evaluation list per query (a chunk of 16 records adds up to 64 rows)


// ---------------------------------------------------------------- refine, exact-integer body
// Sixteen lanes per query as above, but everything is integer and every wave is on its own (no workgroup barrier):
//   records   three packed keys (acc << 8 | tile-in-substream << 1 | e) per stream; a record = the 8 train rows
//             16 e + 8 (r >> 2) + 4 h + (r & 3), r = 0 .. 7, of a tile (half of a lane's accumulator registers in the filter)
//   select    a(2) = the second smallest record score; a row of the exact top-2 (ties and float32 square roots that collide
//             included) sits in a record with score <= a(2) + 1:  two distinct rows with d^2 <= U = c_q - 128 + 2 (a(2) + base) + 1
//             exist, so d2^2 <= U, and a row with d^2 <= d2^2 + 2 has acc <= a(2) + 1.5
//   evaluate  the listed records' rows exactly: d^2 = c_q - 128 + w_t + 2 sum a b (v_dot4_i32_i8 on the byte images), a lane per row;
//             ordered by (sqrtf((float) d^2), index) — the reference compares float32 distances, lower index wins ties
//   certify   a stream discards only rows with acc >= its third key's: d^2 >= c_q - 128 + 2 (a3 + base); a stream with that
//             bound > d2^2 + 2 hides nothing (d^2 < 2^23: at most two consecutive integers share a float32 square root)
//   rescan    any other full stream: its rows are evaluated by the query's 16 lanes (duplicates / exact ties only)
struct Best2I {
    float d[2];
    int i[2];
    int s[2];      // exact d^2
};
__device__ __forceinline__ void best2i_insert_unique(Best2I& b, float d, int i, int s) {
    if (i == b.i[0] || i == b.i[1]) return;
    if (key_less(d, i, b.d[0], b.i[0])) {
        b.d[1] = b.d[0]; b.i[1] = b.i[0]; b.s[1] = b.s[0];
        b.d[0] = d; b.i[0] = i; b.s[0] = s;
    } else if (key_less(d, i, b.d[1], b.i[1])) {
        b.d[1] = d; b.i[1] = i; b.s[1] = s;
    }
}
template <int M>
__device__ __forceinline__ void best2i_exchange_step(Best2I& b) {
    const float d0 = lane_xor<M>(b.d[0]), d1 = lane_xor<M>(b.d[1]);
    const int i0 = lane_xor<M>(b.i[0]), i1 = lane_xor<M>(b.i[1]), s0 = lane_xor<M>(b.s[0]), s1 = lane_xor<M>(b.s[1]);
    best2i_insert_unique(b, d0, i0, s0);
    best2i_insert_unique(b, d1, i1, s1);
}
constexpr int kRecCapI8 = 56;      // (key, slot) pairs per query in the record list (the `rec` array: 112 ints per query)
// Key slots of a query (integer bodies): pair p = stream * 2 + half-wave holds the stream's three keys as one int4 at
// kq[p * kI8Rows * 4 .. + 2] (kq: the row block's array + 4 * query-in-block).  Slot c = 3 p + j names key j of pair p.
constexpr int kS1P = kS1 / 3;      // pairs per lane fetched up front (kS1 keys)
static_assert(kS1 % 3 == 0, "whole pairs");
__device__ __forceinline__ int key_slot(const int* __restrict__ kq, int c) { return kq[(int64_t)(c / 3) * (kI8Rows * 4) + (c % 3)]; }


__device__ __forceinline__ void refine_i8_body(
    const BatchPtrs& P, int B, int nq, int nt, int tiles, const int* __restrict__ minfo, const unsigned char* __restrict__ qi8, const unsigned char* __restrict__ ti8,
    int64_t s_qi8, int64_t s_ti8, const int* __restrict__ wq, const int* __restrict__ wt, int64_t s_qn, int64_t s_tn, const int* __restrict__ keys,
    const int* __restrict__ sttab, int64_t s_keys, int nstr, const int* __restrict__ rb_last8, int n_rb1, int G8, double ratio,
    int* __restrict__ ratio_counts, int ratio_stride, unsigned char* __restrict__ qb /*LDS [kRefQ][128]*/, int* __restrict__ recl /*LDS [kRefQ][2 * kRecCapI8]*/,
    int* __restrict__ scratch /*LDS [512]: the rescan's hand-over*/) {
    const int n_wg = (nq + kRefQ - 1) / kRefQ, n_tot = B * n_wg, wg_chunk = (n_tot + 7) >> 3;
    const int bidt = n_tot >= 64 ? (int)(blockIdx.x & 7) * wg_chunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (bidt >= n_tot) return;
    const int pb = bidt / n_wg, bid = bidt - pb * n_wg;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 4, sl = lane & 15;
    const int ql = wave * 4 + sub;
    const int q = bid * kRefQ + ql;
    const bool valid = q < nq;
    const int qc = valid ? q : 0;
    const int rbl = (bid * kRefQ) / kI8Rows, rb = pb * n_rb1 + rbl;
    const int qloc = qc - rbl * kI8Rows;
    const int NP = rb_last8[B * n_rb1 + rb] * 2, NC = 3 * NP; // live pairs (stream, half-wave) of the row block, key slots
    const int base = minfo[kMinfoBase + pb];
    qi8 += pb * s_qi8; ti8 += pb * s_ti8; wq += pb * s_qn; wt += pb * s_tn; keys += pb * s_keys;
    const int* __restrict__ kq = keys + ((int64_t)rbl * nstr * 2 * kI8Rows + qloc) * 4;  // pair p of this query: the int4 at kq[p * kI8Rows * 4]
    const int* __restrict__ stt = sttab + (int64_t)rb * nstr * 2;
    int* __restrict__ stats = P.stats[pb];
#ifdef SFM_DEBUG_NREC
    if (bid == 0 && threadIdx.x == 0 && stats) { stats[3] = 4; }
#else
    if (bid == 0 && threadIdx.x == 0 && stats) { stats[1] = G8; stats[2] = 2 * nstr; stats[3] = 4; }
#endif

    // the query's bytes -> LDS in element order (chunk 2 f + h of the fragment image = elements 32 f + 16 h .. + 15); wave-local
    if (sl < 8) *reinterpret_cast<uint4*>(qb + ql * kQbStride + 16 * sl) =
        *reinterpret_cast<const uint4*>(qi8 + (int64_t)(qc >> 5) * kI8QTileBytes + (sl >> 1) * 1024 + (((sl & 1) * 32 + (qc & 31)) << 4));
    const int cq = wq[qc] - 128;
    int kv[kS1];                                                 // kv[3 m + j]: key j of pair sl + 16 m (slot 3 (sl + 16 m) + j)
#pragma unroll
    for (int m = 0; m < kS1P; ++m) {
        const int pp = sl + 16 * m;
        const int4 x = (valid && pp < NP) ? *reinterpret_cast<const int4*>(kq + (int64_t)pp * (kI8Rows * 4)) : make_int4(INT_MAX, INT_MAX, INT_MAX, INT_MAX);
        kv[3 * m] = x.x; kv[3 * m + 1] = x.y; kv[3 * m + 2] = x.z;
    }
    auto slot_of = [&](int k) { return 3 * (sl + 16 * (k / 3)) + k % 3; };
    int a1 = INT_MAX, a2 = INT_MAX, atau = INT_MAX;
#pragma unroll
    for (int k = 0; k < kS1; ++k) {
        a2 = imed3(a1, a2, kv[k]);
        a1 = min(a1, kv[k]);
        if (k % 3 == 2) atau = min(atau, kv[k]);
    }
    if (valid)
        for (int pp = sl + 16 * kS1P; pp < NP; pp += 16) {
            const int4 x = *reinterpret_cast<const int4*>(kq + (int64_t)pp * (kI8Rows * 4));
            atau = min(atau, x.z);
            a2 = imed3(a1, a2, x.x); a1 = min(a1, x.x);
            a2 = imed3(a1, a2, x.y); a1 = min(a1, x.y);
            a2 = imed3(a1, a2, x.z); a1 = min(a1, x.z);
        }
    auto fold = [&](int o1, int o2, int ot) {
        const int hi = max(a1, o1);
        a1 = min(a1, o1);
        a2 = min(hi, min(a2, o2));
        atau = min(atau, ot);
    };
    fold(lane_xor<8>(a1), lane_xor<8>(a2), lane_xor<8>(atau));
    fold(lane_xor<4>(a1), lane_xor<4>(a2), lane_xor<4>(atau));
    fold(lane_xor<2>(a1), lane_xor<2>(a2), lane_xor<2>(atau));
    fold(lane_xor<1>(a1), lane_xor<1>(a2), lane_xor<1>(atau));
    // records with score <= a(2) + 1 (every record when there are fewer than two)
    const int thr = a2 >= kKeyEmptyI ? kKeyEmptyI - 1 : (int)((((unsigned)(a2 >> 8) + 1u) << 8) | 0xFFu);

    Best2I b;
    b.d[0] = b.d[1] = kInf; b.i[0] = b.i[1] = INT_MAX; b.s[0] = b.s[1] = INT_MAX;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // one row per lane: register r = sl & 7 of record (tile, half-wave hh, register half e) -> exact d^2, inserted under (sqrtf, index)
    auto eval_row = [&](int tile, int hh, int e, bool live, int qslot, int cqv, Best2I& acc2) {
        const int r = sl & 7;
        const int jr = 16 * e + 8 * (r >> 2) + 4 * hh + (r & 3);
        const int row = tile * kTileT + jr;
        const unsigned char* tp = ti8 + (int64_t)tile * kI8TileBytes + (jr << 4);
        uint4 tv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) tv[c] = *reinterpret_cast<const uint4*>(tp + (c >> 1) * 1024 + (c & 1) * 512);
        const int w = wt[row];
        int s0 = 0, s1 = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint4 qv = *reinterpret_cast<const uint4*>(qb + qslot * kQbStride + 16 * c);
            s0 = __builtin_amdgcn_sdot4((int)tv[c].x, (int)qv.x, s0, false);
            s1 = __builtin_amdgcn_sdot4((int)tv[c].y, (int)qv.y, s1, false);
            s0 = __builtin_amdgcn_sdot4((int)tv[c].z, (int)qv.z, s0, false);
            s1 = __builtin_amdgcn_sdot4((int)tv[c].w, (int)qv.w, s1, false);
        }
        const int d2 = cqv + w + 2 * (s0 + s1);
        if (live && row < nt) best2i_insert_unique(acc2, sqrtf((float)d2), row, d2);
    };
    auto row_scan = [&](int v, int& total) {
        v += __builtin_amdgcn_update_dpp(0, v, 0x111 /*row_shr:1*/, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x112 /*row_shr:2*/, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x114 /*row_shr:4*/, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x118 /*row_shr:8*/, 0xF, 0xF, true);
        total = __builtin_amdgcn_ds_swizzle(v, 0x10 | (0x0F << 5));
        return v;
    };
    auto wave_max = [&](int v) {
        v = max(v, lane_xor<16>(v));
        return max(v, __shfl_xor(v, 32, 64));
    };
    int* __restrict__ myrec = recl + ql * (2 * kRecCapI8);
    int nrec = 0;
    auto process = [&]() {                                       // evaluate the listed records (wave-uniform trip count)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int nmax = wave_max(nrec);
#pragma unroll 1
        for (int e0 = 0; e0 < nmax; e0 += 2) {                   // two 8-row records per pass: lanes 0-7 / 8-15 of the query
            const int e = e0 + (sl >> 3);
            const bool live = e < nrec;
            const int key = live ? myrec[2 * e] : 0, c = live ? myrec[2 * e + 1] : 0;
            const int tile = min(stt[2 * (c / 6)] + ((key & 0xFF) >> 1), tiles - 1);      // (idle lanes: slot 0 -> a real tile, result unused)
            eval_row(tile, (c / 3) & 1, key & 1, live, ql, cq, b);
        }
        __builtin_amdgcn_wave_barrier();
        nrec = 0;
    };
    {   // the chunks still in registers: one prefix sum when the list holds them all (always, but for exact ties en masse)
        bool take[kS1];
        int mine = 0;
#pragma unroll
        for (int k = 0; k < kS1; ++k) {
            take[k] = kv[k] <= thr;                              // (empty / invalid slots: INT_MAX or >= kKeyEmptyI > thr)
            mine += take[k] ? 1 : 0;
        }
        int total;
        int at = row_scan(mine, total) - mine;
        if (!__any(total > kRecCapI8)) {
#pragma unroll
            for (int k = 0; k < kS1; ++k)
                if (take[k]) { myrec[2 * at] = kv[k]; myrec[2 * at + 1] = slot_of(k); ++at; }
            nrec = total;
        } else {
#pragma unroll 1
            for (int k = 0; k < kS1; ++k) {
                if (__any(nrec > kRecCapI8 - 16)) process();
                int kvk = kv[0];
#pragma unroll
                for (int kk = 1; kk < kS1; ++kk) kvk = k == kk ? kv[kk] : kvk;
                const bool tk = kvk <= thr;
                int tot;
                const int a0 = row_scan(tk ? 1 : 0, tot) - (tk ? 1 : 0);
                if (tk) { myrec[2 * (nrec + a0)] = kvk; myrec[2 * (nrec + a0) + 1] = slot_of(k); }
                nrec += tot;
            }
        }
    }
    for (int p0 = 16 * kS1P; p0 < NP; p0 += 16) {                // (wave-uniform) more than 32 pairs (96 key slots) per query
#pragma unroll 1
        for (int j = 0; j < 3; ++j) {
            if (__any(nrec > kRecCapI8 - 16)) process();
            const int pp = p0 + sl;
            const int x = (valid && pp < NP) ? kq[(int64_t)pp * (kI8Rows * 4) + j] : INT_MAX;
            const bool take = x <= thr;
            int total;
            const int at = row_scan(take ? 1 : 0, total) - (take ? 1 : 0);
            if (take) { myrec[2 * (nrec + at)] = x; myrec[2 * (nrec + at) + 1] = 3 * pp + j; }
            nrec += total;
        }
    }
#ifdef SFM_DEBUG_NREC
    if (sl == 0 && stats && valid) { atomicAdd(stats + 1, nrec); atomicAdd(stats + 2, NC); }
#endif
    process();
    auto reduce16 = [&]() {
        best2i_exchange_step<8>(b);
        best2i_exchange_step<4>(b);
        best2i_exchange_step<2>(b);
        best2i_exchange_step<1>(b);
    };
    reduce16();
    // certificate (all-integer)
    const long long lim = b.i[1] != INT_MAX ? (long long)b.s[1] + 2 : LLONG_MAX;
    const bool open = valid && atau < kKeyEmptyI && (long long)cq + 2 * ((long long)(atau >> 8) + base) <= lim;
    // Rescan (rare: exact ties / a third record within 2 of the second neighbour): the WORKGROUP works for one open query at a
    // time — its 256 lanes take sixteen tiles of an uncertified stream per trip.  (A lone query's 16 lanes walking 128 tiles one
    // by one, a dependent load round trip each, was the kernel's tail: 48 us for one query in 10^4.)  Everything before this
    // point is wave-local; this barrier is the kernel's only one.
    if (__syncthreads_or(open ? 1 : 0)) {
        // scratch: [0 .. 15] open flag, [16 ..] per query {cq, lim lo, lim hi, qloc}, [128 ..] the waves' partial results
        if (sl == 0) {
            scratch[ql] = open ? 1 : 0;
            scratch[16 + 4 * ql + 0] = cq;
            scratch[16 + 4 * ql + 1] = (int)(unsigned)lim;
            scratch[16 + 4 * ql + 2] = (int)(lim >> 32);
            scratch[16 + 4 * ql + 3] = qloc;
        }
        __syncthreads();
        for (int w = 0; w < kRefQ; ++w) {                          // (uniform)
            if (!scratch[w]) continue;
            const int cq_w = scratch[16 + 4 * w];
            const long long lim_w = (long long)(((unsigned long long)(unsigned)scratch[16 + 4 * w + 2] << 32) | (unsigned)scratch[16 + 4 * w + 1]);
            const int* __restrict__ kw = kq + 4 * (scratch[16 + 4 * w + 3] - qloc);  // the open query's key slots
            Best2I pbst;
            pbst.d[0] = pbst.d[1] = kInf; pbst.i[0] = pbst.i[1] = INT_MAX; pbst.s[0] = pbst.s[1] = INT_MAX;
            for (int c3 = 2; c3 < NC; c3 += 3) {
                const int key3 = key_slot(kw, c3);                   // (uniform)
                if (!(key3 < kKeyEmptyI && (long long)cq_w + 2 * ((long long)(key3 >> 8) + base) <= lim_w)) continue;
                const int t0 = stt[2 * (c3 / 6)], len = stt[2 * (c3 / 6) + 1];
                for (int tt = ql; tt < len; tt += kRefQ) eval_row(t0 + tt, (c3 / 3) & 1, sl >> 3, true, w, cq_w, pbst);   // a tile's 16 rows of this half-wave per 16 lanes
            }
            best2i_exchange_step<32>(pbst);
            best2i_exchange_step<16>(pbst);
            if (sub == 0) {                                        // lanes 0 .. 15 of every wave: the wave's partial for lane sl
                int* o = scratch + 128 + (wave * 16 + sl) * 6;
                o[0] = __float_as_int(pbst.d[0]); o[1] = pbst.i[0]; o[2] = pbst.s[0];
                o[3] = __float_as_int(pbst.d[1]); o[4] = pbst.i[1]; o[5] = pbst.s[1];
            }
            __syncthreads();
            if (ql == w) {                                         // the owner merges (rows evaluated twice are skipped by index)
                for (int x = 0; x < 4; ++x) {
                    const int* o = scratch + 128 + (x * 16 + sl) * 6;
                    best2i_insert_unique(b, __int_as_float(o[0]), o[1], o[2]);
                    best2i_insert_unique(b, __int_as_float(o[3]), o[4], o[5]);
                }
            }
            __syncthreads();
        }
        if (__ballot(open)) reduce16();
    }

    int* __restrict__ idx_out = P.idx[pb];
    float* __restrict__ dist_out = P.dist[pb];
    unsigned char* __restrict__ ratio_mask = P.mask[pb];
    if (valid && sl == 0) {
        *reinterpret_cast<int2*>(idx_out + 2 * (int64_t)q) = make_int2(b.i[0] == INT_MAX ? -1 : b.i[0], b.i[1] == INT_MAX ? -1 : b.i[1]);
        *reinterpret_cast<float2*>(dist_out + 2 * (int64_t)q) = make_float2(b.d[0], b.d[1]);
        if (open && stats) atomicAdd(stats, 1);
    }
    if (ratio_counts) {                                          // (already offset to this pair by the caller)
        const bool pass = valid && sl == 0 && b.i[1] != INT_MAX && (double)b.d[0] < ratio * (double)b.d[1];
        if (ratio_mask && valid && sl == 0) ratio_mask[q] = pass ? 1 : 0;
        const int n = __popcll(__ballot(pass));
        if (lane == 0) ratio_counts[bid * 4 + wave] = n;           // one plain store per wave (see kRatioSub)
    }
}

// ---------------------------------------------------------------- refine, integer body on QUANTISED float data ("q8")
// The records are the integer body's (exact D = sum (k_q - k_t)^2 of the 8-bit images); what they rank is the QUANTISED
// distance d^ = s sqrt(D).  With E = ||q - q^|| + max_t ||t - t^|| (residual norms measured by the prep pass, rounded up) every
// row satisfies | d - d^ | <= E for the real distance d = ||q - t||, and the float32 direct-form value the reference returns,
// dc = sqrtf(fl(sum (q - t)^2)), satisfies | dc - d | <= rho d + eta (rho = 2^-19: <= 14 roundings of 2^-24 in the sum and the
// square root, generously; eta = 1e-17 for sums that underflow).  In units of s (e = E / s, ...):
//   select    two distinct real rows have D <= U = cq + 2 (a(2) + base) + 1 (the two smallest record keys), so the second
//             smallest dc is <= R = (sqrt(U) + e)(1 + rho) + eta; a row of the answer has dc <= R, hence
//             D <= Dlim(R) = ((R + eta)(1 + 2 rho) + e)^2.  Records whose lower bound of D is above that are skipped.
//   evaluate  the listed records' rows in integers (a lane per row); those with D <= Dlim(R) in the reference's float32
//             arithmetic (a lane quad per row, OpenCV's accumulation order), ordered by (dc, index)
//   certify   a stream discards only rows with acc >= its third key's: D >= Dlow = cq + 2 (a3 + base), so
//             dc >= (sqrt(Dlow) - e)(1 - rho) - eta; a stream whose bound exceeds the answer's dc2 hides nothing
//   rescan    any other full stream: the WAVE evaluates its rows in integers, 64 per trip, and the open query's lanes
//             re-evaluate those with D <= Dlim(dc2) exactly.  No workgroup barrier anywhere.
// u8 pairs inside a quantised batch are the case s = 1, lo = 0, E ~ 0.
// Everything below is float32 arithmetic in units of s on numbers <= 2^23; kQ8Rho = 4e-6 carries rho (1.9e-6) AND the roundings
// of these few operations (each <= 6e-8 relative, a dozen of them): every bound is pushed outwards by it at every use.
constexpr float kQ8Rho = 4e-6f, kQ8Eta = 1e-17f;

__device__ __forceinline__ void refine_q8_body(
    const BatchPtrs& P, int B, int64_t ldq, int nq, int64_t ldt, int nt, int tiles, const int* __restrict__ minfo, const unsigned char* __restrict__ qi8,
    const unsigned char* __restrict__ ti8, int64_t s_qi8, int64_t s_ti8, const int* __restrict__ wq, const int* __restrict__ wt, int64_t s_qn, int64_t s_tn,
    const float* __restrict__ qerr, const int* __restrict__ keys, const int* __restrict__ sttab, int64_t s_keys, int nstr, const int* __restrict__ rb_last8,
    int n_rb1, int G8, double ratio, int* __restrict__ ratio_counts, float* __restrict__ qrows /*LDS [kRefQ][128]*/,
    unsigned char* __restrict__ qb /*LDS [kRefQ][128]*/, int* __restrict__ recl /*LDS [kRefQ][2 * kRecCapI8]*/, int* __restrict__ quall /*LDS [kRefQ][kQualCap]*/,
    long long* __restrict__ trace) {
    const int n_wg = (nq + kRefQ - 1) / kRefQ, n_tot = B * n_wg, wg_chunk = (n_tot + 7) >> 3;
    const int bidt = n_tot >= 64 ? (int)(blockIdx.x & 7) * wg_chunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (bidt >= n_tot) return;
    if (trace && threadIdx.x == 0) trace[16 * bidt + 0] = wall_clock64();   // dev diagnostics
    const int pb = bidt / n_wg, bid = bidt - pb * n_wg;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sub = lane >> 4, sl = lane & 15;
    const int ql = wave * 4 + sub;
    const int q = bid * kRefQ + ql;
    const bool valid = q < nq;
    const int qc = valid ? q : 0;
    const int rbl = (bid * kRefQ) / kI8Rows, rb = pb * n_rb1 + rbl;
    const int qloc = qc - rbl * kI8Rows;
    const int NP = rb_last8[B * n_rb1 + rb] * 2, NC = 3 * NP; // live pairs (stream, half-wave) of the row block, key slots
    const int base = minfo[kMinfoBase + pb];
    const float* __restrict__ Q = P.q[pb];
    const float* __restrict__ T = P.t[pb];
    qi8 += pb * s_qi8; ti8 += pb * s_ti8; wq += pb * s_qn; wt += pb * s_tn; keys += pb * s_keys; qerr += pb * s_qn;
    const int* __restrict__ kq = keys + ((int64_t)rbl * nstr * 2 * kI8Rows + qloc) * 4;  // pair p of this query: the int4 at kq[p * kI8Rows * 4]
    const int* __restrict__ stt = sttab + (int64_t)rb * nstr * 2;
    int* __restrict__ stats = P.stats[pb];
    if (bid == 0 && threadIdx.x == 0 && stats) { stats[1] = G8; stats[2] = 2 * nstr; stats[3] = 5; }

    // the query's bytes (element order) and its float32 row -> LDS; wave-local
    if (sl < 8) *reinterpret_cast<uint4*>(qb + ql * kQbStride + 16 * sl) =
        *reinterpret_cast<const uint4*>(qi8 + (int64_t)(qc >> 5) * kI8QTileBytes + (sl >> 1) * 1024 + (((sl & 1) * 32 + (qc & 31)) << 4));
    {
        const float* src = Q + (int64_t)qc * ldq + 8 * sl;
        *reinterpret_cast<float4*>(qrows + ql * kDim + 8 * sl) = *reinterpret_cast<const float4*>(src);
        *reinterpret_cast<float4*>(qrows + ql * kDim + 8 * sl + 4) = *reinterpret_cast<const float4*>(src + 4);
    }
    const int cq = wq[qc] - 128;
    const float qe2 = qerr[qc];
    int kv[kS1], st0[kS1P];                                      // kv[3 m + j]: key j of pair sl + 16 m; st0[m]: the first tile of that pair's
#pragma unroll                                                   // stream (requested together: a record's tile needs no dependent table lookup)
    for (int m = 0; m < kS1P; ++m) {
        const int pp = sl + 16 * m;
        const bool in = valid && pp < NP;
        const int4 x = in ? *reinterpret_cast<const int4*>(kq + (int64_t)pp * (kI8Rows * 4)) : make_int4(INT_MAX, INT_MAX, INT_MAX, INT_MAX);
        kv[3 * m] = x.x; kv[3 * m + 1] = x.y; kv[3 * m + 2] = x.z;
        st0[m] = in ? stt[2 * (pp >> 1)] : 0;
    }
    auto slot_of = [&](int k) { return 3 * (sl + 16 * (k / 3)) + k % 3; };
    // the pair's grid and this query's slack, in units of s
    const float sf = __int_as_float(minfo[kMinfoQ8S + pb]), lof = __int_as_float(minfo[kMinfoQ8Lo + pb]);
    const float te2 = __int_as_float(minfo[kMinfoTerr + pb]);
    const float inv_s = 1.f / sf;
    const float eta_s = kQ8Eta * inv_s;
    float e_s;
    {
        const float mabs = fmaxf(fabsf(lof), fabsf(lof + 255.f * sf));
        // residual norms: float32 sums (<= 12 roundings) of differences each within 2^-24 mabs of the real one
        const float E = (sqrtf(qe2) + sqrtf(te2)) * (1.f + 8e-6f) + 24.f * 5.9604645e-08f * mabs * (1.f + 1e-6f);
        e_s = E * inv_s * (1.f + 1e-6f);
        if (!(e_s < 1e30f)) e_s = kInf;                          // (NaN / inf residuals: nothing is ever certified or skipped)
    }
    // largest D a row whose float32 distance is <= x (in units of s) can have
    auto dlim_of = [&](float x) -> float {
        const float r = (x + eta_s) * (1.f + 2.f * kQ8Rho) + e_s;
        return r * r * (1.f + 1e-6f) + 2.f;
    };
    int a1 = INT_MAX, a2 = INT_MAX, atau = INT_MAX;
#pragma unroll
    for (int k = 0; k < kS1; ++k) {
        a2 = imed3(a1, a2, kv[k]);
        a1 = min(a1, kv[k]);
        if (k % 3 == 2) atau = min(atau, kv[k]);
    }
    if (valid)
        for (int pp = sl + 16 * kS1P; pp < NP; pp += 16) {
            const int4 x = *reinterpret_cast<const int4*>(kq + (int64_t)pp * (kI8Rows * 4));
            atau = min(atau, x.z);
            a2 = imed3(a1, a2, x.x); a1 = min(a1, x.x);
            a2 = imed3(a1, a2, x.y); a1 = min(a1, x.y);
            a2 = imed3(a1, a2, x.z); a1 = min(a1, x.z);
        }
    auto fold = [&](int o1, int o2, int ot) {
        const int hi = max(a1, o1);
        a1 = min(a1, o1);
        a2 = min(hi, min(a2, o2));
        atau = min(atau, ot);
    };
    fold(lane_xor<8>(a1), lane_xor<8>(a2), lane_xor<8>(atau));
    fold(lane_xor<4>(a1), lane_xor<4>(a2), lane_xor<4>(atau));
    fold(lane_xor<2>(a1), lane_xor<2>(a2), lane_xor<2>(atau));
    fold(lane_xor<1>(a1), lane_xor<1>(a2), lane_xor<1>(atau));
    // record threshold (keys) and row threshold (D) from the second smallest record key; every record when there are fewer than two
    int thr = kKeyEmptyI - 1;
    int dlim = INT_MAX;                                          // rows with D <= dlim are evaluated in float32
    if (a2 < kKeyEmptyI) {
        const long long U = (long long)cq + 2 * ((long long)(a2 >> 8) + base) + 1;
        const float R = (sqrtf((float)(U > 0 ? U : 0)) * (1.f + 2e-7f) + e_s) * (1.f + kQ8Rho) + eta_s;
        const float dl = dlim_of(R);
        if (dl < 1.6e7f) {                                       // (D < 2^23 for every row: a larger bound limits nothing)
            dlim = (int)dl;
            // a row with D <= dlim has acc <= al (D >= cq + 2 (acc + base)); integer arithmetic from here on
            const long long al = (((long long)dlim - cq) >> 1) - base + 1;
            if (al < 8388606) thr = (int)((((unsigned)(int)al) << 8) | 0xFFu);
        }
        if (thr < a2) thr = a2;                                  // (never below the two records the bound came from)
    }
    if (trace && threadIdx.x == 0) trace[16 * bidt + 1] = wall_clock64();

    Best2 b;
    b.d[0] = b.d[1] = kInf; b.i[0] = b.i[1] = INT_MAX;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    // exact D of one row per lane: register r of record (tile, half-wave hh, register half e) against the bytes of query slot `qslot`
    auto int_row = [&](int tile, int hh, int e, int r, int qslot, int cqv, int& row) -> int {
        const int jr = 16 * e + 8 * (r >> 2) + 4 * hh + (r & 3);
        row = tile * kTileT + jr;
        const unsigned char* tp = ti8 + (int64_t)tile * kI8TileBytes + (jr << 4);
        uint4 tv[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) tv[c] = *reinterpret_cast<const uint4*>(tp + (c >> 1) * 1024 + (c & 1) * 512);
        const int w = wt[row];
        int s0 = 0, s1 = 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const uint4 qv = *reinterpret_cast<const uint4*>(qb + qslot * kQbStride + 16 * c);
            s0 = __builtin_amdgcn_sdot4((int)tv[c].x, (int)qv.x, s0, false);
            s1 = __builtin_amdgcn_sdot4((int)tv[c].y, (int)qv.y, s1, false);
            s0 = __builtin_amdgcn_sdot4((int)tv[c].z, (int)qv.z, s0, false);
            s1 = __builtin_amdgcn_sdot4((int)tv[c].w, (int)qv.w, s1, false);
        }
        return cqv + w + 2 * (s0 + s1);
    };
    auto row_scan = [&](int v, int& total) {
        v += __builtin_amdgcn_update_dpp(0, v, 0x111 /*row_shr:1*/, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x112 /*row_shr:2*/, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x114 /*row_shr:4*/, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x118 /*row_shr:8*/, 0xF, 0xF, true);
        total = __builtin_amdgcn_ds_swizzle(v, 0x10 | (0x0F << 5));
        return v;
    };
    auto wave_max = [&](int v) {
        v = max(v, lane_xor<16>(v));
        return max(v, __shfl_xor(v, 32, 64));
    };
    int* __restrict__ myrec = recl + ql * (2 * kRecCapI8);
    int* __restrict__ myqual = quall + ql * kQualCap;
    auto rec_info = [](int key, int t0, int c) { return ((t0 + ((key & 0xFF) >> 1)) << 1) | ((c / 3) & 1); };   // a record's tile << 1 | half-wave
    int nrec = 0, nqual = 0;
    // float32 evaluation of the rows on the query's list: a lane quad per row, four rows per pass (wave-uniform trip count)
    auto eval_list = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int nmax = wave_max(nqual);
#pragma unroll 1
        for (int e0 = 0; e0 < nmax; e0 += 4) {
            const int e = e0 + (sl >> 2);
            const int row = e < nqual ? myqual[e] : -1;
            const float* const trow[1] = {row >= 0 ? T + (int64_t)row * ldt : nullptr};
            float out[1];
            exact_l2sq_quad_rows<1, true>(qrows + ql * kDim, trow, sl & 3, out);
            if ((sl & 3) == 2 && row >= 0) best2_insert_unique(b, sqrtf(out[0]), row);
        }
        best2_reduce16_from_quad_lane2(b);
        __builtin_amdgcn_wave_barrier();
        nqual = 0;
    };
    auto process = [&]() {                                       // integer evaluation of the listed records: two 8-row records per pass
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const int nmax = wave_max(nrec);
#pragma unroll 1
        for (int e0 = 0; e0 < nmax; e0 += 2) {
            if (__any(nqual > kQualCap - 16)) eval_list();
            const int e = e0 + (sl >> 3);
            const bool live = e < nrec;
            const int key = live ? myrec[2 * e] : 0, inf = live ? myrec[2 * e + 1] : 0;      // (idle lanes: tile 0, result unused)
            int row;
            const int D = int_row(min(inf >> 1, tiles - 1), inf & 1, key & 1, sl & 7, ql, cq, row);
            const bool take = live && row < nt && D <= dlim;
            int total;
            const int at = row_scan(take ? 1 : 0, total) - (take ? 1 : 0);
            if (take) myqual[nqual + at] = row;
            nqual += total;
        }
        __builtin_amdgcn_wave_barrier();
        nrec = 0;
    };
    {   // the chunks still in registers: one prefix sum when the list holds them all
        bool take[kS1];
        int mine = 0;
#pragma unroll
        for (int k = 0; k < kS1; ++k) {
            take[k] = kv[k] <= thr;                              // (empty / invalid slots: INT_MAX or >= kKeyEmptyI > thr)
            mine += take[k] ? 1 : 0;
        }
        int total;
        int at = row_scan(mine, total) - mine;
        if (!__any(total > kRecCapI8)) {
#pragma unroll
            for (int k = 0; k < kS1; ++k)
                if (take[k]) { myrec[2 * at] = kv[k]; myrec[2 * at + 1] = rec_info(kv[k], st0[k / 3], slot_of(k)); ++at; }
            nrec = total;
        } else {
#pragma unroll 1
            for (int k = 0; k < kS1; ++k) {
                if (__any(nrec > kRecCapI8 - 16)) process();
                int kvk = kv[0], stk = st0[0];
#pragma unroll
                for (int kk = 1; kk < kS1; ++kk) { kvk = k == kk ? kv[kk] : kvk; stk = k == kk ? st0[kk / 3] : stk; }
                const bool tk = kvk <= thr;
                int tot;
                const int a0 = row_scan(tk ? 1 : 0, tot) - (tk ? 1 : 0);
                if (tk) { myrec[2 * (nrec + a0)] = kvk; myrec[2 * (nrec + a0) + 1] = rec_info(kvk, stk, slot_of(k)); }
                nrec += tot;
            }
        }
    }
    for (int p0 = 16 * kS1P; p0 < NP; p0 += 16) {                // (wave-uniform) more than 32 pairs (96 key slots) per query
        const int pp = p0 + sl;
        const bool in = valid && pp < NP;
        const int t0x = in ? stt[2 * (pp >> 1)] : 0;
#pragma unroll 1
        for (int j = 0; j < 3; ++j) {
            if (__any(nrec > kRecCapI8 - 16)) process();
            const int x = in ? kq[(int64_t)pp * (kI8Rows * 4) + j] : INT_MAX;
            const bool take = x <= thr;
            int total;
            const int at = row_scan(take ? 1 : 0, total) - (take ? 1 : 0);
            if (take) { myrec[2 * (nrec + at)] = x; myrec[2 * (nrec + at) + 1] = rec_info(x, t0x, 3 * pp + j); }
            nrec += total;
        }
    }
    if (trace && threadIdx.x == 0) { trace[16 * bidt + 2] = wall_clock64(); trace[16 * bidt + 13] = nrec; }
    process();
    if (trace && threadIdx.x == 0) { trace[16 * bidt + 3] = wall_clock64(); trace[16 * bidt + 14] = nqual; }
    eval_list();
    if (trace && threadIdx.x == 0) trace[16 * bidt + 4] = wall_clock64();

    // certificate: does the stream with third key `key3` provably hide nothing closer than the answer's second distance?
    auto hides_nothing = [&](int key3, float d2c) -> bool {
        if (key3 >= kKeyEmptyI) return true;                       // fewer than three records: the stream discarded nothing
        const long long dlow = (long long)cq + 2 * ((long long)(key3 >> 8) + base);
        const float lowb = (sqrtf((float)(dlow > 0 ? dlow : 0)) * (1.f - 2e-7f) - e_s) * (1.f - kQ8Rho) - eta_s;
        return lowb > d2c * inv_s * (1.f + 1e-6f);                 // (d2c = +inf while fewer than two rows are known, e_s = +inf: false)
    };
    const bool open = valid && !hides_nothing(atau, b.d[1]);
    // Rescan (rare): the WAVE works for one open query at a time; 64 lanes take four tiles of an uncertified stream per trip
    // (integer D of one row each), the rows within the bound join the open query's list and are re-evaluated by its own lanes.
    unsigned long long openm = __ballot(open);
    if (trace && threadIdx.x == 0) trace[16 * bidt + 6] = openm != 0;
    while (openm) {                                                // (wave-uniform)
        const int wl = (int)__builtin_ctzll(openm) >> 4;           // query slot inside the wave
        openm &= ~(0xFFFFull << (16 * wl));
        const int src = 16 * wl;
        const int cq_w = __shfl(cq, src, 64), qloc_w = __shfl(qloc, src, 64);
        const int* __restrict__ kw = kq + 4 * (qloc_w - qloc);     // the open query's key slots
        const bool mine = sub == wl;
        // Which streams are open?  The owner's lanes still HOLD the third keys of its first 16 kS1P pairs (kv[3 m + 2] of pair sl + 16 m):
        // one ballot per m lists the pairs whose bound fails at the CURRENT second distance — a superset of what fails later, the
        // distance only shrinks — and the walk visits those alone, in ascending pair order as before.  (Round 5 re-read every
        // stream's third key from memory, one dependent load per pair: 32-64 round trips, most of the +17 us a rescanning wave cost.)
        unsigned long long failm = 0;                              // bit pp: pair pp (< 16 kS1P) is open at the current bound
#pragma unroll
        for (int m = 0; m < kS1P; ++m)
            failm |= ((__ballot(mine && !hides_nothing(kv[3 * m + 2], b.d[1])) >> src) & 0xFFFFull) << (16 * m);
        for (int c3 = 2; c3 < NC; c3 += 3) {
            int key3;
            if (c3 < 3 * 16 * kS1P) {
                if (!failm) { c3 = 3 * 16 * kS1P - 1; continue; }  // (the pairs beyond the registers, if any, follow)
                const int pp = (int)__builtin_ctzll(failm);
                failm &= failm - 1;
                c3 = 3 * pp + 2;
                int held = kv[2];
#pragma unroll
                for (int m = 1; m < kS1P; ++m) held = (pp >> 4) == m ? kv[3 * m + 2] : held;
                key3 = __shfl(held, src + (pp & 15), 64);          // (uniform)
            } else {
                key3 = key_slot(kw, c3);                           // (uniform)
            }
            // the owner's lanes decide (their slack, their current second distance); everyone follows
            const bool hid = hides_nothing(key3, b.d[1]);
            const unsigned long long vote = __ballot(mine && !hid);
            if (!vote) continue;
            // D bound of rows that can still enter the answer, from the owner's current second distance
            int dl_o = INT_MAX;
            {
                const float dl = dlim_of(b.d[1] * inv_s * (1.f + 1e-6f));
                if (dl < 1.6e7f) dl_o = (int)dl;
            }
            const int dl_w = __shfl(dl_o, src, 64);
            const int t0 = stt[2 * (c3 / 6)], len = stt[2 * (c3 / 6) + 1], hh = (c3 / 3) & 1;
            for (int tt = 0; tt < len; tt += 4) {
                if (__any(nqual > kQualCap - 64)) eval_list();
                const int tl = tt + (lane >> 4);
                const bool live = tl < len;
                int row;
                const int D = int_row(live ? t0 + tl : t0, hh, (lane >> 3) & 1, lane & 7, wave * 4 + wl, cq_w, row);
                const bool take = live && row < nt && D <= dl_w;
                const unsigned long long tm = __ballot(take);
                const int nq_w = __shfl(nqual, src, 64);             // (the owner's list length)
                if (take) quall[(wave * 4 + wl) * kQualCap + nq_w + __popcll(tm & ((1ull << lane) - 1ull))] = row;
                if (mine) nqual += __popcll(tm);
            }
        }
        eval_list();
    }

    int* __restrict__ idx_out = P.idx[pb];
    float* __restrict__ dist_out = P.dist[pb];
    unsigned char* __restrict__ ratio_mask = P.mask[pb];
    if (valid && sl == 0) {
        *reinterpret_cast<int2*>(idx_out + 2 * (int64_t)q) = make_int2(b.i[0] == INT_MAX ? -1 : b.i[0], b.i[1] == INT_MAX ? -1 : b.i[1]);
        *reinterpret_cast<float2*>(dist_out + 2 * (int64_t)q) = make_float2(b.d[0], b.d[1]);
        if (open && stats) atomicAdd(stats, 1);
    }
    if (ratio_counts) {                                          // (already offset to this pair by the caller)
        const bool pass = valid && sl == 0 && b.i[1] != INT_MAX && (double)b.d[0] < ratio * (double)b.d[1];
        if (ratio_mask && valid && sl == 0) ratio_mask[q] = pass ? 1 : 0;
        const int n = __popcll(__ballot(pass));
        if (lane == 0) ratio_counts[bid * 4 + wave] = n;           // one plain store per wave (see kRatioSub)
    }
    if (trace && threadIdx.x == 0) trace[16 * bidt + 5] = wall_clock64();
}

// Sixteen lanes per query, sixteen queries per workgroup (one resident round for 10^4 queries).
//   sweep 1  two smallest filter scores over the query's candidate records and each stream's 3rd best
//   sweep 2  candidates within 4*eps of the 2nd smallest score are compacted and evaluated exactly → (d1, d2)
//   certify  per stream: the filter score includes |q|^2 (s ~ d^2) and a stream discards only trains with s >= its
//            3rd best s3, hence exact d^2 >= s3 - eps; a stream with d2^2 + eps < s3 provably hides nothing.
//   rescan   every other FULL stream (typically none; a handful for ~0.1 % of the queries; all of them for degenerate
//            train sets) has its <= 512 trains evaluated exactly by the whole workgroup and merged into the query's
//            answer.  After that the answer equals a full direct-form scan.
// FRAG: the fp16 train image is the q4 filter's FRAGMENT-ORDER image (`thalf` = tfrag: [32-row tile][9][64 lanes][16 B], tile
// stride kTileFragBytes) — what is contiguous there is a 16-byte chunk of ADJACENT ROWS, so the screens give a lane one row
// (a quad = the four adjacent rows of one record: 64 contiguous bytes per load instruction, as before) and every lane walks
// all 16 chunks of its row; there is no row-major fp16 copy of T at all.
template <bool FRAG>
__global__ __launch_bounds__(256, 4) void knn_refine_kernel(
    BatchPtrs P, int B, int64_t ldq, int nq, int64_t ldt, int nt,
    const float* __restrict__ cand_s, const int* __restrict__ cand_i, int rows_per_block, int tiles, int64_t units,
    int G, int smax, int nsub, int force_mode, const int* __restrict__ midflag, const float* __restrict__ bmax,
    const int* __restrict__ minfo /*null: derive from the flags (fp32-MFMA filter)*/, const float* __restrict__ qerr, int64_t s_qn,
    const unsigned short* __restrict__ thalf /*fp16 image of T, or null*/, const float* __restrict__ tn,
    const int64_t* __restrict__ wg_begin, const int* __restrict__ rb_first, const int* __restrict__ rb_last,
    int n_rb1, int64_t s_cand, int64_t s_tsplit, int64_t s_tn, double ratio,
    int* __restrict__ ratio_counts /*null unless fused with the Lowe ratio*/, int ratio_stride, float chain_scale,
    int qoff /*1: the filter's scores carry ||q||^2max of the pair instead of the query's own ||q||^2 (q4 filter)*/, long long* __restrict__ trace,
    // the exact-integer body (qi8 == null: not planned)
    const unsigned char* __restrict__ qi8, const unsigned char* __restrict__ ti8, int64_t s_qi8, int64_t s_ti8, const int* __restrict__ wq8,
    const int* __restrict__ wt8, const int* __restrict__ keys8, const int* __restrict__ sttab8, int64_t s_keys8, int nstr8,
    const int* __restrict__ rb_last8, int n_rb1_8, int G8) {
    // XCD-aware order (see the filter): physical workgroup b takes query block (b % 8) * chunk + b / 8, so the queries an
    // XCD refines are (roughly) those whose candidate records its own filter workgroups wrote.  Batched: the query
    // blocks of all pairs form one sequence, pair after pair.
    const int n_wg = (nq + kRefQ - 1) / kRefQ, n_tot = B * n_wg, wg_chunk = (n_tot + 7) >> 3;
    const int bidt = n_tot >= 64 ? (int)(blockIdx.x & 7) * wg_chunk + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if (bidt >= n_tot) return;
    const int pb = bidt / n_wg, bid = bidt - pb * n_wg;
    const float* __restrict__ Q = P.q[pb];
    const float* __restrict__ T = P.t[pb];
    int* __restrict__ idx_out = P.idx[pb];
    float* __restrict__ dist_out = P.dist[pb];
    int* __restrict__ stats = P.stats[pb];
    unsigned char* __restrict__ ratio_mask = P.mask[pb];
    cand_s += pb * s_cand; cand_i += pb * s_cand;
    const int* __restrict__ midflag0 = midflag;
    const float* __restrict__ bmax0 = bmax;
    midflag += pb * kNormBlocks; bmax += pb * kNormBlocks;
    if (thalf) thalf += pb * s_tsplit;
    tn += pb * s_tn;
    if (ratio_counts) ratio_counts += pb * ratio_stride;

    __shared__ __attribute__((aligned(16))) float qrows[kRefQ][kDim];
    __shared__ __attribute__((aligned(16))) _Float16 qhalf[kRefQ][kDim];   // fp16(-2 q): the screens' operand (as the filter's)
    __shared__ int qual[kRefQ][kQualCap];                // rows awaiting the exact evaluation
    __shared__ int rec[kRefQ][kRecCap];                  // records (first row of four) within the threshold
    __shared__ int items[kRefItems];                     // (query slot << 20) | stream
    __shared__ double q_lim[kRefQ];
    __shared__ float q_qq[kRefQ];
    __shared__ int cnt_lds[kRefQ];
    __shared__ int surv[kSubTiles * 16];
    __shared__ int nitem, nsurv;
    __shared__ Best2 wbest[4];
    if constexpr (FRAG)
    if (qi8 && minfo && minfo[kMinfoI8] && minfo[kMinfoQ8]) {   // (uniform) the integer body ran on quantised float data
        refine_q8_body(P, B, ldq, nq, ldt, nt, tiles, minfo, qi8, ti8, s_qi8, s_ti8, wq8, wt8, s_qn, s_tn, qerr, keys8, sttab8, s_keys8, nstr8, rb_last8, n_rb1_8, G8,
                       ratio, ratio_counts, &qrows[0][0], reinterpret_cast<unsigned char*>(&qhalf[0][0]), &rec[0][0], &qual[0][0], trace);
        return;
    }
    if constexpr (FRAG)
    if (qi8 && minfo && minfo[kMinfoI8]) {                 // (uniform) the exact-integer body ran: its records, its certificate
        refine_i8_body(P, B, nq, nt, tiles, minfo, qi8, ti8, s_qi8, s_ti8, wq8, wt8, s_qn, s_tn, keys8, sttab8, s_keys8, nstr8, rb_last8, n_rb1_8, G8, ratio,
                       ratio_counts, ratio_stride, reinterpret_cast<unsigned char*>(&qrows[0][0]), &rec[0][0], items);
        static_assert(kRefItems >= 128 + 64 * 6, "the i8 body's rescan scratch");
        return;
    }
    if (trace && threadIdx.x == 0) trace[16 * bidt + 0] = wall_clock64();   // dev diagnostics
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int sub = lane >> 4, sl = lane & 15;           // query slot inside the wave, lane inside the query
    const int ql = wave * 4 + sub;                       // query slot inside the workgroup
    const int q = bid * kRefQ + ql;
    const bool valid = q < nq;

    // streams of this workgroup's query row block = filter blocks that touched it (contiguous slots from 0);
    // the queries of a workgroup share the row block (rows_per_block is a multiple of 16)
    const int rb = pb * n_rb1 + (bid * kRefQ) / rows_per_block;      // GLOBAL row block (the partition tables span the batch)
    const int fb = rb_first[rb];
    const int lb = rb_last[rb];
    // (q4 filter: compact streams — the row block's live substreams, numbered densely; else every slot's nsub substreams)
    const int NC = qoff ? 2 * rb_last[B * n_rb1 + rb] * 3 : 2 * (lb - fb + 1) * nsub * 3;
    for (int e = threadIdx.x; e < kRefQ * 32; e += 256) {
        const int row = bid * kRefQ + (e >> 5);
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < nq) v = *reinterpret_cast<const float4*>(Q + (int64_t)row * ldq + 4 * (e & 31));
        *reinterpret_cast<float4*>(&qrows[e >> 5][4 * (e & 31)]) = v;
        const f16x2 h0 = {(_Float16)(-2.f * v.x), (_Float16)(-2.f * v.y)}, h1 = {(_Float16)(-2.f * v.z), (_Float16)(-2.f * v.w)};
        *reinterpret_cast<uint2*>(&qhalf[e >> 5][4 * (e & 31)]) = make_uint2(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1));
    }
    if (threadIdx.x == 0) nitem = nsurv = 0;

    // Slack that dominates: rounding of the filter's dot-product chain, of ||q||^2, ||t||^2, of the direct-form
    // sums (<= 24u*d^2) and the final sqrtf merge (8u*d^2) — 600u*(|q|+|t|max)^2, u = 2^-24 — plus what the
    // filter arithmetic that ran loses on top (packed keys; 16-bit operands): see kEps*.
    float tmax;                                                       // this pair's ||t||max (the slack is relative to it) ...
    int mode;                                                         // ... and the arithmetic the batch's filter launch ran
    if (minfo) {                                                      // (scalar loads of what filter block 0 reduced)
        tmax = __int_as_float(minfo[kMinfoTmax + pb]);
        mode = minfo[kMinfoBatchMode];
    } else {
        mode = knn_filter_mode(midflag, bmax, lane, &tmax);
        if (B > 1) mode = knn_batch_mode(midflag0, bmax0, B, lane);
    }
    if (force_mode >= 0) mode = force_mode;
    const float kEpsExact = kEpsRound + kChainHalf * chain_scale;      // (chain_scale: 1 unless this device's MFMA self-test exceeded E = 8)
    const float eps_coef = mode == kModeHalfExact ? kEpsExact : mode == kModeHalf ? kEpsExact + kEpsHalfOp
                           : mode == kModeSplit ? kEpsRound + kChainSplit * chain_scale + kEpsSplitOp : kEpsF32;

    const float* cs = cand_s + (int64_t)(valid ? q : 0) * (2 * smax * 3);
    const int* ci = cand_i + (int64_t)(valid ? q : 0) * (2 * smax * 3);
    if (bid == 0 && threadIdx.x == 0 && stats) {
        stats[1] = G; stats[2] = 2 * smax; stats[3] = mode;
    }

    // Sweep 1: two smallest scores and the smallest 3rd-best, on the scores' BIT PATTERNS (scores are >= 0 up to rounding
    // noise, so integer order = float order and min / med3 are single instructions without canonicalisation; among
    // negative noise values any order is as good as another: they differ by far less than eps).
    float s1v[kS1];
    int i1v[kS1];                                        // the records' row ids ride along: sweep 2 takes its first 16 * kS1
#pragma unroll                                           // records from these registers instead of a second, dependent round trip
    for (int k = 0; k < kS1; ++k) {                      // (NC: three scalar loads issued at the top of the kernel, behind the
        const int c = sl + 16 * k;                       // query rows' vector loads; slots past NC were never written by this
        s1v[k] = (valid && c < NC) ? cs[c] : kInf;       // launch — reading them would be HBM traffic for nothing)
        i1v[k] = (valid && c < NC) ? ci[c] : -1;
    }
    const float qe2 = (qerr && valid) ? qerr[pb * s_qn + q] : 0.f;   // (issued with the record loads: not a round trip of its own)
    __syncthreads();                                     // query rows in LDS
    if (trace && threadIdx.x == 0) trace[16 * bidt + 1] = wall_clock64();
    float qq;
    {
        const float4 a = *reinterpret_cast<const float4*>(&qrows[ql][8 * sl]), c4 = *reinterpret_cast<const float4*>(&qrows[ql][8 * sl + 4]);
        qq = a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w + c4.x * c4.x + c4.y * c4.y + c4.z * c4.z + c4.w * c4.w;
    }
    qq += lane_xor<8>(qq); qq += lane_xor<4>(qq); qq += lane_xor<2>(qq); qq += lane_xor<1>(qq);
    // The q4 filter scores with s = ||t||^2 + ||q||^2max - 2 q.t — the same offset for every query of the pair, so that ONE
    // init MFMA serves all the query groups of a wave.  Everything below that compares filter scores with each other is
    // untouched (a query's records all carry the same offset); the screens add the same number the filter did (qadd), the
    // certificate moves the exact d2^2 into the scores' frame (qadd - qq, in double), and the slack is priced on the
    // magnitudes the filter's accumulators actually held: N = ||q||max + ||t||max.
    const float qadd = (qoff && minfo) ? __int_as_float(minfo[kMinfoQmax + pb]) : qq;
    const float nsum = sqrtf(qadd) + sqrtf(tmax);
    if (sl == 0) q_qq[ql] = qadd;                         // (the rescan's operand)
    // Operand rounding of the fp16 single product, bounded from the data instead of the worst case 2^-10 (|q|+|t|)^2 / 2:
    // the filter scores with Q' = fp16(-2 q) and T' = fp16(t), so its product is off by at most
    // |dQ'| |t| + |Q'| |dT| + |dQ'| |dT| (Cauchy-Schwarz) with the residual norms |dQ'| of THIS query and max |dT| of the
    // pair's train rows, both measured by the prep pass (2-3 x smaller than the worst case on ordinary data: the threshold
    // window, the rows to screen and the rescans shrink with it; exactly 0 for fp16-exact descriptors).
    float eps;
    if (mode == kModeHalf && minfo && qerr) {
        const float dq = sqrtf(qe2), dt = sqrtf(__int_as_float(minfo[kMinfoTerr + pb]));
        const float op = dq * sqrtf(tmax) + 2.f * sqrtf(qq) * dt + dq * dt;
        eps = 1.01f * (kEpsExact * nsum * nsum + 1.001f * op);
    } else {
        eps = 1.01f * (eps_coef * nsum * nsum + (mode == kModeHalf ? kEpsHalfAbs * nsum : 0.f));
    }
    int a1 = kKeyInf, a2 = kKeyInf, atau;
    static_assert(kS1 == 6, "the 3rd-best selection below is written for six records per lane");
    {
#pragma unroll
        for (int k = 0; k < kS1; ++k) {
            const int x = __float_as_int(s1v[k]);
            a2 = imed3(a1, a2, x);         // (a1 <= a2): the second smallest of the three
            a1 = min(a1, x);
        }
        // record c = sl + 16 k is a stream's 3rd best iff c % 3 == 2 iff (sl % 3 + k) % 3 == 2: two of a lane's six
        const int r3 = sl % 3;
        const int t0 = __float_as_int(r3 == 0 ? s1v[2] : r3 == 1 ? s1v[1] : s1v[0]);
        const int t1 = __float_as_int(r3 == 0 ? s1v[5] : r3 == 1 ? s1v[4] : s1v[3]);
        atau = min(t0, t1);
    }
    if (valid)
        for (int c = sl + 16 * kS1; c < NC; c += 16) {
            const int x = __float_as_int(cs[c]);
            if (c % 3 == 2) atau = min(atau, x);
            a2 = imed3(a1, a2, x);
            a1 = min(a1, x);
        }
    auto fold = [&](int o1, int o2, int ot) {
        const int hi = max(a1, o1);
        a1 = min(a1, o1);
        a2 = min(hi, min(a2, o2));
        atau = min(atau, ot);
    };
    fold(lane_xor<8>(a1), lane_xor<8>(a2), lane_xor<8>(atau));
    fold(lane_xor<4>(a1), lane_xor<4>(a2), lane_xor<4>(atau));
    fold(lane_xor<2>(a1), lane_xor<2>(a2), lane_xor<2>(atau));
    fold(lane_xor<1>(a1), lane_xor<1>(a2), lane_xor<1>(atau));
    const float m2 = __int_as_float(a2), tau = __int_as_float(atau);
    // Record scores are truncated (never raised): the rows behind the two best records have s <= m2 (1 + 2^-15), hence
    // d^2 <= U = m2 (1 + 2^-15) + eps.  A record with score > U + 1.25 eps holds only rows with s > U + 1.25 eps, i.e.
    // d^2 > U + eps/4: farther than both by more than eps/4 >= 100u (|q|+|t|)^2, which also separates the float32 square
    // roots (one ulp of sqrtf is < 4u (|q|+|t|)^2 in d^2 terms) — it cannot be in the exact top-2, ties included.  A
    // stream's certificate needs no such allowance: what it discarded has s >= its third record's (truncated) score.
    const float thr = (m2 + kKeyTrunc * fabsf(m2)) + 2.25f * eps;
    if (trace && threadIdx.x == 0) trace[16 * bidt + 2] = wall_clock64();

    // Sweep 2.  The kernel is bound by vector-ALU ISSUE (four waves per SIMD, ~1000 instructions each), so every stage
    // is written for instruction count:
    //   compaction  records within `thr` -> the query's record list: one prefix sum over the lanes' counts (DPP row
    //               shifts: a query's 16 lanes are one DPP row), not a ballot per chunk;
    //   screen      (fp16 modes) a record names four adjacent rows, usually only the one that produced its minimum is
    //               near the threshold: rows are screened against the fp16 train image (L2-resident: the filter just
    //               streamed it; half the bytes) with v_dot2_f32_f16 on fp16(-2q), the filter's own operand:
    //               s' = ||t||^2 + ||q||^2 - 2 q^.t^ has |s' - d^2| <= eps, and a row of the exact top-2 has
    //               d^2 <= m2 + eps, hence s' <= m2 + 2 eps < thr.  A lane quad per pair of adjacent rows (512 contiguous
    //               bytes), survivors appended through an LDS counter;
    //   evaluate    survivors with the reference arithmetic, a lane quad per row.
    Best2 b;
    b.d[0] = b.d[1] = kInf; b.i[0] = b.i[1] = INT_MAX;
    const int jq = sl & 3, qd = sl >> 2;                  // lane inside the quad, quad inside the query's 16 lanes
    int cnt = 0;                                          // entries of rows[ql] (uniform inside a query's 16 lanes)
    int nrec = 0;                                         // entries of rec[ql]
    auto wave_max = [&](int v) {
        v = max(v, lane_xor<16>(v));
        return max(v, __shfl_xor(v, 32, 64));
    };
    // Exact evaluation of listed trains: FROM_REC = false the row list (screen / rescan survivors), true the record list
    // expanded (modes without a screen).  A row index past the last train (a record's tail) is clamped: evaluating an
    // extra REAL row exactly is always harmless.
    auto evaluate = [&](auto from_rec) {
        constexpr bool kFromRec = decltype(from_rec)::value;
        const int n = kFromRec ? kRecRows * nrec : cnt;
        const int nmax = wave_max(n);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll 1
        for (int e0 = 0; e0 < nmax; e0 += 4) {
            const int e = e0 + qd;
            int tid = 0;                                  // idle quads re-read train 0 (nt >= 1 here)
            if (e < n) tid = min(kFromRec ? rec[ql][e >> 2] + (e & 3) : qual[ql][e], nt - 1);
            const float* const rows[1] = {T + (int64_t)tid * (int)ldt};
            float dsq[1];
            exact_l2sq_quad_rows<1, false>(qrows[ql], rows, jq, dsq);
            if (jq == 2 && e < n) best2_insert_unique(b, sqrtf(dsq[0]), tid);
        }
        __builtin_amdgcn_wave_barrier();
        if (kFromRec) nrec = 0; else cnt = 0;
    };
    const bool use_half = thalf && (mode == kModeHalf || mode == kModeHalfExact);
    auto screen = [&]() {
        const int nrow = kRecRows * nrec, nmax = wave_max(nrow);
        const bool tight = nmax > kQualCap;               // (wave-uniform) the row list might not hold every survivor
        if (sl == 0) cnt_lds[ql] = 0;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        static_assert((kHotPre == 2 || kHotPre == 4) && kRecRows == 4, "a quad screens half a record or a whole one (adjacent rows) per pass");
        if constexpr (FRAG) {
#pragma unroll 1
            for (int e0 = 0; e0 < nmax; e0 += 16) {           // a record per quad, a row per lane
                if (tight) {                                  // degenerate inputs only
                    __builtin_amdgcn_wave_barrier();
                    cnt = cnt_lds[ql];
                    if (__any(cnt > kQualCap - 16)) {
                        evaluate(std::false_type{});
                        if (sl == 0) cnt_lds[ql] = 0;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                const int e = e0 + 4 * qd;
                const bool live = e < nrow;
                const int row = live ? rec[ql][e >> 2] + jq : 0;      // (< the padded image's rows: a record never leaves its tile)
                const unsigned char* fp = reinterpret_cast<const unsigned char*>(thalf) + (int64_t)(row >> 5) * kTileFragBytes + ((row & 31) << 4);
                float tnv = tn[row];                                  // (padded rows: +inf, never pass)
                float d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                for (int hf = 0; hf < 2; ++hf) {                      // 8 chunks (32 VGPRs) in flight at a time (unrolled: 138 spills at the 128-VGPR bound)
                    uint4 hv[8];
#pragma unroll
                    for (int k = 0; k < 8; ++k) hv[k] = *reinterpret_cast<const uint4*>(fp + (4 * hf + (k >> 1)) * kFragBytes + (k & 1) * 512);
#pragma unroll
                    for (int k = 0; k < 8; ++k)
                        d[k & 3] = dot8_f16(hv[k], *reinterpret_cast<const uint4*>(&qhalf[ql][8 * (8 * hf + k)]), d[k & 3]);
                }
                asm volatile("" : "+v"(tnv));
                const float sp = (tnv + qadd) + ((d[0] + d[2]) + (d[1] + d[3]));
                if (live && !(thr < sp)) qual[ql][atomicAdd(&cnt_lds[ql], 1)] = row;
            }
        } else {
#pragma unroll 1
            for (int e0 = 0; e0 < nmax; e0 += 4 * kHotPre) {
                if (tight) {                                  // degenerate inputs only
                    __builtin_amdgcn_wave_barrier();
                    cnt = cnt_lds[ql];
                    if (__any(cnt > kQualCap - 4 * kHotPre)) {
                        evaluate(std::false_type{});
                        if (sl == 0) cnt_lds[ql] = 0;
                        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                        __builtin_amdgcn_wave_barrier();
                    }
                }
                const int e = e0 + kHotPre * qd;              // rows e .. e + kHotPre - 1 of the list: rows (e & 3) ... of record e >> 2
                const bool live = e < nrow;
                const int row0 = live ? rec[ql][e >> 2] + (e & 3) : 0;   // (< the padded image's rows: a record never leaves its tile)
                const unsigned short* hp = thalf + (int64_t)row0 * kDim + 8 * jq;
                uint4 hv[kHotPre][4];
    #pragma unroll
                for (int r = 0; r < kHotPre; ++r)
    #pragma unroll
                    for (int n = 0; n < 4; ++n) hv[r][n] = *reinterpret_cast<const uint4*>(hp + r * kDim + 32 * n);
                float tnv = tn[row0 + (jq & (kHotPre - 1))];   // (padded rows: +inf, never pass)
                float dot[kHotPre];
    #pragma unroll
                for (int r = 0; r < kHotPre; ++r) dot[r] = 0.f;
    #pragma unroll
                for (int n = 0; n < 4; ++n) {
                    const uint4 qh = *reinterpret_cast<const uint4*>(&qhalf[ql][32 * n + 8 * jq]);
    #pragma unroll
                    for (int r = 0; r < kHotPre; ++r) dot[r] = dot8_f16(hv[r][n], qh, dot[r]);
                }
    #pragma unroll
                for (int r = 0; r < kHotPre; ++r) {
                    dot[r] += lane_xor<2>(dot[r]);
                    dot[r] += lane_xor<1>(dot[r]);
                }
                asm volatile("" : "+v"(tnv));                  // (keeps the load up here: sunk into the branch below it is a second round trip)
                // lane j of the quad finishes row j of its kHotPre
                float mine = dot[0];
    #pragma unroll
                for (int r = 1; r < kHotPre; ++r) mine = (jq & (kHotPre - 1)) == r ? dot[r] : mine;
                const float sp = (tnv + qadd) + mine;
                if (jq < kHotPre && live && !(thr < sp)) qual[ql][atomicAdd(&cnt_lds[ql], 1)] = row0 + jq;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        cnt = cnt_lds[ql];
        nrec = 0;
        if (trace && threadIdx.x == 0 && !trace[16 * bidt + 12]) { trace[16 * bidt + 12] = wall_clock64(); trace[16 * bidt + 14] = cnt; }
    };
#define SFM_REFINE_PROCESS()                    \
    do {                                        \
        if (use_half) {                         \
            screen();                           \
            evaluate(std::false_type{});        \
        } else {                                \
            evaluate(std::true_type{});         \
        }                                       \
    } while (0)                                           /* (wave-uniform branches) */
    // inclusive prefix sum over the 16 lanes of a query (one DPP row) and its total
    auto row_scan = [&](int v, int& total) {
        v += __builtin_amdgcn_update_dpp(0, v, 0x111 /*row_shr:1*/, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x112 /*row_shr:2*/, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x114 /*row_shr:4*/, 0xF, 0xF, true);
        v += __builtin_amdgcn_update_dpp(0, v, 0x118 /*row_shr:8*/, 0xF, 0xF, true);
        total = __builtin_amdgcn_ds_swizzle(v, 0x10 | (0x0F << 5));   // lane 15 of the row: (lane & 0x10) | 0x0F
        return v;
    };
    {   // the chunks still in registers from sweep 1: at most 96 records, the list holds them all
        bool take[kS1];
        int mine = 0;
#pragma unroll
        for (int k = 0; k < kS1; ++k) {
            take[k] = i1v[k] >= 0 && s1v[k] <= thr;
            mine += take[k] ? 1 : 0;
        }
        int total;
        int at = row_scan(mine, total) - mine;
#pragma unroll
        for (int k = 0; k < kS1; ++k)
            if (take[k]) rec[ql][at++] = i1v[k];
        nrec = total;
    }
    static_assert(kRecCap >= 16 * kS1 + 16, "the register round plus one more chunk");
    for (int k0 = kS1; k0 * 16 < NC; ++k0) {              // (wave-uniform) further chunks — more streams than 32 per query — are re-read
        if (__any(nrec > kRecCap - 16)) SFM_REFINE_PROCESS();   // the list might not take another chunk: work it off first
        const int c = sl + 16 * k0;
        const float sc = (valid && c < NC) ? cs[c] : kInf;
        const int id = (valid && c < NC) ? ci[c] : -1;                // (empty records: s = +inf, id = -1)
        const bool take = id >= 0 && sc <= thr;
        int total;
        const int at = row_scan(take ? 1 : 0, total) - (take ? 1 : 0);
        if (take) rec[ql][nrec + at] = id;
        nrec += total;
    }
    if (trace && threadIdx.x == 0) { trace[16 * bidt + 11] = wall_clock64(); trace[16 * bidt + 13] = kRecRows * nrec; }
    SFM_REFINE_PROCESS();                                 // the hot site
#undef SFM_REFINE_PROCESS
    if (trace && threadIdx.x == 0) trace[16 * bidt + 15] = wall_clock64();
    double lim = 0.0;
    bool rescanned = false;
    for (int round = 0;; ++round) {
        if (round == 1) evaluate(std::false_type{});      // rescan survivors (cold site)
        best2_reduce16_from_quad_lane2(b);
        if (trace && threadIdx.x == 0 && round == 0) trace[16 * bidt + 3] = wall_clock64();
        if (round == 1) break;
        // Certificate.  (s3 < 0 can only be rounding noise: such a stream never certifies.)
        lim = b.i[1] != INT_MAX ? dsq_upper(b.d[1]) + (double)eps + ((double)qadd - (double)qq) : (double)kInf;   // (in the scores' frame)
        const bool open = valid && tau < kInf && !(lim < (double)tau);    // some stream could not be certified
        if (!__syncthreads_or(open ? 1 : 0)) break;

        // ---- rescan of the streams that could not be certified (rare; everything below is cold code) ----
        if (open) {
            if (sl == 0) {
                q_lim[ql] = lim;
                cnt_lds[ql] = 0;
            }
            for (int c = sl; c < NC; c += 16)
                if (c % 3 == 2) {
                    const float s3 = cs[c];
                    if (s3 < kInf && !(lim < (double)s3)) {
                        const int p = atomicAdd(&nitem, 1);
                        if (p < kRefItems) items[p] = (ql << 20) | (c / 3);
                    }
                }
            rescanned = true;
        }
        __syncthreads();
        const int n_items = nitem;
        // more uncertified streams than the list holds (degenerate train sets): 16 candidate slots per query at a time
        const int n_rounds = n_items <= kRefItems ? 1 : (NC + 15) / 16;
        for (int k0 = 0; k0 < n_rounds; ++k0) {
            if (n_items > kRefItems) {
                __syncthreads();
                if (threadIdx.x == 0) nitem = 0;
                __syncthreads();
                const int c = sl + 16 * k0;
                if (open && c < NC && c % 3 == 2) {
                    const float s3 = cs[c];
                    if (s3 < kInf && !(lim < (double)s3)) items[atomicAdd(&nitem, 1)] = (ql << 20) | (c / 3);   // <= 96 per round
                }
                __syncthreads();
            }
            const int n_it = min(nitem, kRefItems);
            for (int it = 0; it < n_it; ++it) {
                const int w = items[it] >> 20, sid = items[it] & 0xFFFFF;
                int slot = sid / (2 * nsub), sb = (sid % (2 * nsub)) >> 1;
                const int h = sid & 1;
                if (qoff) {                                   // compact streams: walk the row block's blocks to the one that owns substream sid >> 1
                    sb = sid >> 1;
                    for (slot = 0; fb + slot < lb; ++slot) {
                        const int64_t a0 = max(wg_begin[fb + slot], (int64_t)rb * tiles), a1 = min(wg_begin[fb + slot + 1], (int64_t)(rb + 1) * tiles);
                        const int nl = (int)((a1 - a0 + kSubTiles - 1) / kSubTiles);
                        if (sb < nl) break;
                        sb -= nl;
                    }
                }
                const int wg = fb + slot;
                const int64_t u0 = max(wg_begin[wg], (int64_t)rb * tiles);
                const int64_t u1 = min(wg_begin[wg + 1], (int64_t)(rb + 1) * tiles);
                const int t_begin = (int)(u0 - (int64_t)rb * tiles) + kSubTiles * sb;
                const int t_end = min((int)(u1 - (int64_t)rb * tiles), t_begin + kSubTiles);
                const int ntr = (t_end - t_begin) * 16;
                auto train_of = [&](int i) { return (t_begin + (i >> 4)) * kTileT + (i & 3) + 8 * ((i >> 2) & 3) + 4 * h; };
                if (trace && threadIdx.x == 0 && !trace[16 * bidt + 6]) trace[16 * bidt + 6] = wall_clock64();
                // Which of the stream's trains need the exact arithmetic?  In the fp16 modes the fp16 image (L2-resident:
                // the filter just streamed it; half the bytes of the fp32 rows in HBM) gives s' = ||t||^2 + ||q||^2 - 2 q.t^
                // with |s' - d^2| <= eps (only t is rounded here, the filter rounds both operands), so only trains with
                // s' <= d2^2 + eps can enter the top-2.  Otherwise: all of them.
                if (use_half && FRAG) {
                    const double lim_w = q_lim[w];
                    const float qq_w = q_qq[w];
                    for (int i = threadIdx.x; i < ntr; i += 256) {     // a lane per train; four consecutive i are four adjacent rows
                        const int tr = train_of(i);                   // (< the padded image's rows; padded rows: ||t||^2 = +inf)
                        const unsigned char* fp = reinterpret_cast<const unsigned char*>(thalf) + (int64_t)(tr >> 5) * kTileFragBytes + ((tr & 31) << 4);
                        float d[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
                        for (int hf = 0; hf < 2; ++hf) {
                            uint4 hv[8];
#pragma unroll
                            for (int k = 0; k < 8; ++k) hv[k] = *reinterpret_cast<const uint4*>(fp + (4 * hf + (k >> 1)) * kFragBytes + (k & 1) * 512);
#pragma unroll
                            for (int k = 0; k < 8; ++k)
                                d[k & 3] = dot8_f16(hv[k], *reinterpret_cast<const uint4*>(&qhalf[w][8 * (8 * hf + k)]), d[k & 3]);
                        }
                        const float sp = (tn[tr] + qq_w) + ((d[0] + d[2]) + (d[1] + d[3]));
                        if (tr < nt && !(lim_w < (double)sp)) surv[atomicAdd(&nsurv, 1)] = tr;
                    }
                } else if (use_half) {
                    const double lim_w = q_lim[w];
                    const float qq_w = q_qq[w];
                    const int jq = threadIdx.x & 3;
                    for (int i0 = threadIdx.x >> 2; i0 < ntr; i0 += 64 * kPreRows) {   // a quad per train
                        uint4 hv[kPreRows][4];
                        int tr[kPreRows];
#pragma unroll
                        for (int r = 0; r < kPreRows; ++r) {
                            const int i = i0 + 64 * r;
                            tr[r] = i < ntr ? train_of(i) : nt;
#pragma unroll
                            for (int n = 0; n < 4; ++n)
                                hv[r][n] = tr[r] < nt ? *reinterpret_cast<const uint4*>(thalf + (int64_t)tr[r] * kDim + 32 * n + 8 * jq)
                                                      : make_uint4(0u, 0u, 0u, 0u);
                        }
#pragma unroll
                        for (int r = 0; r < kPreRows; ++r) {
                            float dot = 0.f;
#pragma unroll
                            for (int n = 0; n < 4; ++n)
                                dot = dot8_f16(hv[r][n], *reinterpret_cast<const uint4*>(&qhalf[w][32 * n + 8 * jq]), dot);
                            dot += lane_xor<2>(dot);
                            dot += lane_xor<1>(dot);
                            if (jq == 0 && tr[r] < nt) {
                                const float sp = (tn[tr[r]] + qq_w) + dot;
                                if (!(lim_w < (double)sp)) surv[atomicAdd(&nsurv, 1)] = tr[r];
                            }
                        }
                    }
                } else {
                    for (int i = threadIdx.x; i < ntr; i += 256) {
                        const int t = train_of(i);
                        if (t < nt) surv[atomicAdd(&nsurv, 1)] = t;
                    }
                }
                __syncthreads();
                const int ns = nsurv;
                if (trace && threadIdx.x == 0 && !trace[16 * bidt + 7]) { trace[16 * bidt + 7] = wall_clock64(); trace[16 * bidt + 10] = ns; }
                const int have = cnt_lds[w];
                if (have + ns <= kQualCap) {
                    // the usual case, a few survivors: queue them for the owning query's next (hot) evaluation
                    if (threadIdx.x < ns) qual[w][have + threadIdx.x] = surv[threadIdx.x];
                    __syncthreads();
                    if (threadIdx.x == 0) {
                        cnt_lds[w] = have + ns;
                        nsurv = 0;
                    }
                } else {
                    // many survivors: evaluate them here with the whole workgroup, a quad per train
                    Best2 pb;
                    pb.d[0] = pb.d[1] = kInf; pb.i[0] = pb.i[1] = INT_MAX;
                    for (int i0 = threadIdx.x >> 2; i0 < ns; i0 += 64 * kRescanRows) {
                        const float* rows[kRescanRows];
                        int tr[kRescanRows];
#pragma unroll
                        for (int r = 0; r < kRescanRows; ++r) {
                            const int i = i0 + 64 * r;
                            tr[r] = i < ns ? surv[i] : -1;
                            rows[r] = tr[r] >= 0 ? T + (int64_t)tr[r] * ldt : nullptr;
                        }
                        float dsq[kRescanRows];
                        exact_l2sq_quad_rows<kRescanRows, true>(qrows[w], rows, threadIdx.x & 3, dsq);
#pragma unroll
                        for (int r = 0; r < kRescanRows; ++r)
                            if ((threadIdx.x & 3) == 2 && rows[r]) best2_insert(pb, sqrtf(dsq[r]), tr[r]);
                    }
                    best2_group_reduce<64>(pb);
                    if (lane == 0) wbest[wave] = pb;
                    __syncthreads();
                    if (ql == w) {
#pragma unroll
                        for (int x = 0; x < 4; ++x) {
                            best2_insert_unique(b, wbest[x].d[0], wbest[x].i[0]);
                            best2_insert_unique(b, wbest[x].d[1], wbest[x].i[1]);
                        }
                    }
                    if (threadIdx.x == 0) nsurv = 0;
                }
                __syncthreads();
            }
        }
        if (open) cnt = cnt_lds[ql];                       // survivors queued for this query → round 1
        if (trace && threadIdx.x == 0) trace[16 * bidt + 8] = wall_clock64();
    }

    int tq = threadIdx.x;
    asm volatile("" : "+v"(tq));                          // output addresses are formed HERE (hoisted to the top they are spilled)
    const int qo = bid * kRefQ + (tq >> 4);
    if (valid && sl == 0) {
        *reinterpret_cast<int2*>(idx_out + 2 * (int64_t)qo) = make_int2(b.i[0] == INT_MAX ? -1 : b.i[0], b.i[1] == INT_MAX ? -1 : b.i[1]);
        *reinterpret_cast<float2*>(dist_out + 2 * (int64_t)qo) = make_float2(b.d[0], b.d[1]);
        if (rescanned && stats) atomicAdd(stats, 1);
    }
    if (ratio_counts) {
        // fused sfm_match_l2_f32: the Lowe test of sfm.py:264 here, survivors counted per WAVE (4 queries) with one plain
        // store each, so the match list needs the ordered scatter pass only.  (Round 3 added them to one counter per 1024
        // queries with atomics: on data with survivors — 30 % of the queries of a SIFT-like pair — the ~14 000 atomics of a batch
        // landed on half a dozen cache lines and took 45 us, more than the rest of the kernel.)
        const bool pass = valid && sl == 0 && b.i[1] != INT_MAX && (double)b.d[0] < ratio * (double)b.d[1];
        if (ratio_mask && valid && sl == 0) ratio_mask[qo] = pass ? 1 : 0;
        const int n = __popcll(__ballot(pass));
        if (lane == 0) ratio_counts[bid * 4 + (threadIdx.x >> 6)] = n;
    }
    if (trace && threadIdx.x == 0) {
        trace[16 * bidt + 4] = wall_clock64();
        trace[16 * bidt + 5] = __builtin_amdgcn_s_getreg(0xF804);
    }
}

// ---------------------------------------------------------------- ratio test + ordered compaction
// `m.distance < 0.70 * n.distance` (sfm.py:264): float32 distances promoted to double.
// Two multi-workgroup passes (a single workgroup is limited by one CU's ~10 B/clk load path):
//   count:   1024 queries per workgroup → pass bits (optionally the mask) + one count per workgroup
//   scatter: every workgroup sums the counts of its predecessors (a few dozen ints), re-derives its bits and writes
//            its survivors at the right offset → ascending queryIdx order, no atomics, deterministic.

__device__ __forceinline__ unsigned ratio_bits(const int* __restrict__ idx, const float* __restrict__ dist, int nq, int qb,
                                               double ratio, int (&ti)[4]) {
    unsigned pass = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int q = qb + k;
        ti[k] = -1;
        if (q < nq) {
            const int2 ii = *reinterpret_cast<const int2*>(idx + 2 * q);
            const float2 dd = *reinterpret_cast<const float2*>(dist + 2 * q);
            ti[k] = ii.x;
            pass |= ((ii.y >= 0) && ((double)dd.x < ratio * (double)dd.y) ? 1u : 0u) << k;
        }
    }
    return pass;
}

__global__ __launch_bounds__(256) void ratio_count_kernel(const int* __restrict__ idx, const float* __restrict__ dist, int nq,
                                                          double ratio, int* __restrict__ block_count,
                                                          unsigned char* __restrict__ mask) {
    __shared__ int wsum[4];
    const int qb = blockIdx.x * kRatioBlock + threadIdx.x * 4;
    int ti[4];
    const unsigned pass = ratio_bits(idx, dist, nq, qb, ratio, ti);
    if (mask)
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (qb + k < nq) mask[qb + k] = (pass >> k) & 1u;
    int c = __popc(pass);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) c += __shfl_xor(c, m, 64);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = c;
    __syncthreads();
    if (threadIdx.x == 0) block_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

constexpr int kRatioSub = 4 * (kRatioBlock / kRefQ);     // survivor counts per 1024 queries when the refine kernel wrote them: one per wave (4 queries)
__device__ __forceinline__ void ratio_scatter_body(const int* __restrict__ idx, const float* __restrict__ dist, int nq,
                                                   double ratio, const int* __restrict__ block_count,
                                                   int* __restrict__ out_q, int* __restrict__ out_t,
                                                   int* __restrict__ out_count, int per_block = 1) {
    __shared__ int wsum[4];
    __shared__ int base_s;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // exclusive prefix of the preceding workgroups' counts
    int part = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x * per_block; b += 256) part += block_count[b];
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) part += __shfl_xor(part, m, 64);
    if (lane == 0) wsum[wave] = part;
    __syncthreads();
    if (threadIdx.x == 0) base_s = wsum[0] + wsum[1] + wsum[2] + wsum[3];
    __syncthreads();
    const int base = base_s;
    __syncthreads();

    const int qb = blockIdx.x * kRatioBlock + threadIdx.x * 4;
    int ti[4];
    const unsigned pass = ratio_bits(idx, dist, nq, qb, ratio, ti);
    const int mine = __popc(pass);
    int incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int o = __shfl_up(incl, d, 64);
        if (lane >= d) incl += o;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    int woff = 0;
    for (int w = 0; w < wave; ++w) woff += wsum[w];
    int pos = base + woff + incl - mine;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (pass & (1u << k)) {
            out_q[pos] = qb + k;
            out_t[pos] = ti[k];
            ++pos;
        }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) *out_count = pos;   // last thread of the last workgroup: total
}

__global__ __launch_bounds__(256) void ratio_scatter_kernel(const int* __restrict__ idx, const float* __restrict__ dist, int nq,
                                                            double ratio, const int* __restrict__ block_count,
                                                            int* __restrict__ out_q, int* __restrict__ out_t,
                                                            int* __restrict__ out_count) {
    ratio_scatter_body(idx, dist, nq, ratio, block_count, out_q, out_t, out_count);
}

// Batched form: grid = (blocks, B), column b scatters pair b.
__global__ __launch_bounds__(256) void ratio_scatter_batch_kernel(BatchPtrs P, int nq, double ratio, const int* __restrict__ block_count,
                                                                  int count_stride) {
    const int pb = blockIdx.y;
    ratio_scatter_body(P.idx[pb], P.dist[pb], nq, ratio, block_count + pb * count_stride, P.out_q[pb], P.out_t[pb], P.out_count[pb], kRatioSub);
}

__global__ __launch_bounds__(256) void gather_matches_kernel(const float2* __restrict__ kp0, const float2* __restrict__ kp1,
                                                             const int* __restrict__ oq, const int* __restrict__ ot,
                                                             const int* __restrict__ count, int64_t capacity,
                                                             float2* __restrict__ p0, float2* __restrict__ p1) {
    int64_t n = *count;
    if (n > capacity) n = capacity;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        p0[i] = kp0[oq[i]];
        p1[i] = kp1[ot[i]];
    }
}

__global__ void knn_fill_empty_kernel(int* __restrict__ idx, float* __restrict__ dist, int64_t n2, int* __restrict__ stats) {
    const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (i < n2) {
        idx[i] = -1;
        dist[i] = kInf;
    }
    if (i == 0 && stats) stats[0] = stats[1] = stats[2] = stats[3] = 0;
}

long long* g_trace = nullptr;   // dev diagnostics only
int g_split_delay_wg = -1;      // test hook: see knn_split_images_kernel
long long g_split_delay_ticks = 0;

struct KnnWs {
    unsigned short* qsplit;       // per-pair arrays: pair b at base + b * stride (elements)
    unsigned short* tsplit;
    float* qn;
    float* tn;
    float* bmax;                  // [B][kNormBlocks]
    float* qerr;                  // [B][s_qn]: ||fp16(-2 q) - (-2 q)||^2 per query; bmaxerr [B][kNormBlocks]: per-block max of ||fp16(t) - t||^2
    float* bmaxerr;
    float* bqmax;                 // [B][kNormBlocks]: per-block max of ||q||^2 (the q4 filter's common score offset)
    int* midflag;
    int* minfo;                   // [kMinfoWords] modes / ||t||max reduced by filter block 0 for the refine kernel
    int64_t* wg_begin;            // partition tables over the whole batch (filled by the prep / norms launch)
    int* rb_first;
    int* rb_last;
    int* wg_sbase;                // [G] number of the first substream of a block's first segment (compact streams, q4)
    float* cand_s;
    int* cand_i;
    unsigned char* qfrag;         // q4 filter: fragment-order fp16 images (+ init fragments), kTileFragBytes per 32 rows
    unsigned char* tfrag;
    unsigned char* qhm;           // ... and the bf16 hi / mid planes in fragment order (split arithmetic only), 16 KiB per 32 rows
    unsigned char* thm;
    int64_t s_qsplit, s_tsplit, s_qn, s_tn, s_cand, s_qfrag, s_tfrag, s_qhm, s_thm;
    // exact-integer body (kFilterAuto): byte images, integer norms, key records, its partition tables
    unsigned char* qi8;
    unsigned char* ti8;
    int* wq;                      // [B][s_qn]  |q - 128|^2
    int* wt;                      // [B][s_tn]  |t - 127|^2
    int* bwmin;                   // [B][kNormBlocks] per-block min / max of wt over the real train rows
    int* bwmax;
    unsigned short* rmq;          // [B][s_qn] / [B][s_tn]: bit c = chunk c (8 elements) of the row exists in the byte image only
    unsigned short* rmt;
    int* midflag_r;               // [B][kNormBlocks] x 2: the repair's rewritten flag / residual words, copied over midflag / bmaxerr by the last
    float* bmaxerr_r;             // workgroup out — what the repair DECISION reads is never written while a workgroup may still be deciding
    int* keys8;                   // [B][row blocks][stream slots][2][1024 queries][4] packed keys (three per stream and half-wave, one 16-byte slot)
    int* sttab8;                  // [row blocks of the batch][stream slots][2]: first tile, tile count of a stream
    int64_t* wg_begin8;
    int* rb_first8;
    int* rb_last8;
    int* wg_sbase8;
    int64_t s_qi8, s_ti8, s_keys8;
    size_t bytes;
};

KnnWs carve_ws(void* ws, int64_t nq, int64_t nt, const Plan& p, const Plan* p8) {
    (void)nt;
    sfm::Carver c(ws);
    KnnWs w{};
    const size_t B = (size_t)p.B;
    w.s_tn = (int64_t)p.tiles * kTileT;
    w.s_qn = p.nq_pad;
    w.s_qsplit = p.q4 ? 0 : (int64_t)p.nq_pad * kDim * 3;   // (q4: the query images exist in fragment order only)
    w.s_tsplit = p.q4 ? 0 : (int64_t)p.tiles * kTileT * kDim * 3;
    w.s_cand = (int64_t)nq * 2 * p.smax * p.nsub * 3;
    w.bmax = c.take<float>(B * kNormBlocks);
    w.midflag = c.take<int>(B * kNormBlocks);
    w.minfo = c.take<int>(kMinfoWords);
    w.wg_begin = c.take<int64_t>((size_t)p.G + 1);
    w.rb_first = c.take<int>((size_t)p.n_rb);
    w.rb_last = c.take<int>(2 * (size_t)p.n_rb);          // [n_rb] last block, then [n_rb] live substreams of the row block
    w.wg_sbase = c.take<int>((size_t)p.G + 1);
    w.tn = c.take<float>(B * (size_t)w.s_tn);
    w.qn = c.take<float>(B * (size_t)w.s_qn);
    w.qerr = c.take<float>(B * (size_t)w.s_qn);
    w.bmaxerr = c.take<float>(B * kNormBlocks);
    w.bqmax = c.take<float>(B * kNormBlocks);
    w.qsplit = c.take<unsigned short>(B * (size_t)w.s_qsplit);
    w.tsplit = c.take<unsigned short>(B * (size_t)w.s_tsplit);
    w.cand_s = c.take<float>(B * (size_t)w.s_cand);
    w.cand_i = c.take<int>(B * (size_t)w.s_cand);
    w.s_qfrag = p.q4 ? (int64_t)(p.nq_pad / 32) * kTileFragBytes : 0;
    w.s_tfrag = p.q4 ? (int64_t)p.tiles * kTileFragBytes : 0;
    w.s_qhm = p.q4 ? (int64_t)(p.nq_pad / 32) * 16 * kFragBytes : 0;
    w.s_thm = p.q4 ? (int64_t)p.tiles * 16 * kFragBytes : 0;
    w.qfrag = c.take<unsigned char>(B * (size_t)w.s_qfrag);
    w.tfrag = c.take<unsigned char>(B * (size_t)w.s_tfrag);
    w.qhm = c.take<unsigned char>(B * (size_t)w.s_qhm);
    w.thm = c.take<unsigned char>(B * (size_t)w.s_thm);
    if (p8) {
        w.s_qi8 = (int64_t)(p.nq_pad / 32) * kI8QTileBytes;
        w.s_ti8 = (int64_t)p.tiles * kI8TileBytes;
        w.s_keys8 = (int64_t)p8->n_rb1 * p8->smax * p8->nsub * 2 * kI8Rows * 4;    // [row block][stream slot][half-wave][1024 queries][3 keys + pad]
        w.qi8 = c.take<unsigned char>(B * (size_t)w.s_qi8);
        w.ti8 = c.take<unsigned char>(B * (size_t)w.s_ti8);
        w.wq = c.take<int>(B * (size_t)w.s_qn);
        w.wt = c.take<int>(B * (size_t)w.s_tn);
        w.bwmin = c.take<int>(B * kNormBlocks);
        w.bwmax = c.take<int>(B * kNormBlocks);
        w.rmq = c.take<unsigned short>(B * (size_t)w.s_qn);
        w.rmt = c.take<unsigned short>(B * (size_t)w.s_tn);
        w.midflag_r = c.take<int>(B * kNormBlocks);
        w.bmaxerr_r = c.take<float>(B * kNormBlocks);
        w.keys8 = c.take<int>(B * (size_t)w.s_keys8);
        w.sttab8 = c.take<int>((size_t)p8->n_rb * p8->smax * p8->nsub * 2);
        w.wg_begin8 = c.take<int64_t>((size_t)p8->G + 1);
        w.rb_first8 = c.take<int>((size_t)p8->n_rb);
        w.rb_last8 = c.take<int>(2 * (size_t)p8->n_rb);
        w.wg_sbase8 = c.take<int>((size_t)p8->G + 1);
    }
    w.bytes = c.used();
    return w;
}

}  // namespace

// ---------------------------------------------------------------- self-test: accumulation error of the 16-bit MFMA
// The certificate's chain term rests on a property of the matrix pipe that no manual states: how far
// v_mfma_f32_32x32x16_{f16,bf16} is from the exact c + sum_k a_k b_k.  This kernel measures it: operands come from a hash
// (so a lane can recompute the row and column of every element it holds), the exact value is formed in fp64 (products
// of 16-bit floats are exact, 17 terms), and the error is reported in units of 2^-24 (|c| + sum |a_k b_k|), maximum
// over every output element of every trial, for regimes from "equal exponents" to "2^16 spread with cancellation".
namespace {
struct MfmaRegime { int ea, sa, eb, sb, ec, sc, signed_; };

__device__ inline uint32_t st_hash(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t h = a * 0x9E3779B1u ^ (b + 0x7F4A7C15u) * 0x85EBCA77u ^ (c + 0x165667B1u) * 0xC2B2AE3Du;
    h ^= h >> 15; h *= 0x2C1B3C6Du; h ^= h >> 12; h *= 0x297A2D39u; h ^= h >> 15;
    return h;
}
__device__ inline float st_gen(uint32_t h, int e0, int span, bool sg, int mant_bits) {     // 2^[e0, e0+span), mant_bits of mantissa
    const int e = e0 + (int)((h >> 11) % (uint32_t)span);
    const float m = 1.f + (float)(h & ((1u << mant_bits) - 1u)) / (float)(1u << mant_bits);
    return ldexpf(m, e) * ((sg && (h >> 31)) ? -1.f : 1.f);
}

typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
template <int KIND>       // 0: v_mfma_f32_32x32x16_f16, 1: ..._32x32x16_bf16, 2: v_mfma_f32_32x32x8_bf16 (the accumulator-init MFMA)
__global__ __launch_bounds__(256) void mfma_selftest_kernel(MfmaRegime rg, int trials, uint32_t seed, double* __restrict__ maxerr) {
    const int lane = threadIdx.x & 63, wid = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int rc = lane & 31, h = lane >> 5;
    constexpr bool BF16 = KIND != 0;
    constexpr int KK = KIND == 2 ? 8 : 16, PER = KK / 2;      // k-slots of the instruction, per lane
    constexpr int kMant = BF16 ? 7 : 10;
    double worst = 0;
    for (int t = 0; t < trials; ++t) {
        const uint32_t s = seed + 7919u * (uint32_t)(wid * trials + t);
        float av[8], bv[8];
        for (int e = 0; e < PER; ++e) {
            av[e] = st_gen(st_hash(s, rc, PER * h + e), rg.ea, rg.sa, rg.signed_, kMant);               // A[row rc][k = PER h + e]
            bv[e] = st_gen(st_hash(s ^ 0xABCDu, rc, PER * h + e), rg.eb, rg.sb, rg.signed_, kMant);     // B[k][col rc]
        }
        f32x16 C, D;
        for (int r = 0; r < 16; ++r) C[r] = st_gen(st_hash(s ^ 0x1234u, 8 * (r >> 2) + 4 * h + (r & 3), rc), rg.ec, rg.sc, rg.signed_, 23);
        if constexpr (KIND == 2) {
            typedef short s16x4 __attribute__((ext_vector_type(4)));
            bf16x4 A, B;
            for (int e = 0; e < 4; ++e) { A[e] = (__bf16)av[e]; B[e] = (__bf16)bv[e]; }
            D = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(__builtin_bit_cast(s16x4, A), __builtin_bit_cast(s16x4, B), C, 0, 0, 0);
        } else if constexpr (KIND == 1) {
            bf16x8 A, B;
            for (int e = 0; e < 8; ++e) { A[e] = (__bf16)av[e]; B[e] = (__bf16)bv[e]; }
            D = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B, C, 0, 0, 0);
        } else {
            f16x8 A, B;
            for (int e = 0; e < 8; ++e) { A[e] = (_Float16)av[e]; B[e] = (_Float16)bv[e]; }
            D = __builtin_amdgcn_mfma_f32_32x32x16_f16(A, B, C, 0, 0, 0);
        }
        for (int r = 0; r < 16; ++r) {
            const int row = 8 * (r >> 2) + 4 * h + (r & 3);
            double ref = (double)C[r], mag = fabs((double)C[r]);
            for (int kk = 0; kk < KK; ++kk) {
                const double a = (double)st_gen(st_hash(s, row, kk), rg.ea, rg.sa, rg.signed_, kMant);
                const double b = (double)st_gen(st_hash(s ^ 0xABCDu, rc, kk), rg.eb, rg.sb, rg.signed_, kMant);
                ref += a * b;
                mag += fabs(a * b);
            }
            const double err = fabs((double)D[r] - ref) / (mag * 5.9604644775390625e-08);
            if (err > worst) worst = err;
        }
    }
    for (int m = 32; m >= 1; m >>= 1) {
        const double o = __shfl_xor(worst, m, 64);
        if (o > worst) worst = o;
    }
    if (lane == 0) maxerr[wid] = worst;
}
}  // namespace

extern "C" int sfm_selftest_mfma_accumulation(int use_bf16, int trials_per_wave, double* regime_max_host /*[7]*/, void* ws, size_t ws_bytes,
                                              void* stream_) {
    constexpr int kBlocks = 1024;
    SFM_CHECK_ARG(regime_max_host && trials_per_wave >= 1, "sfm_selftest_mfma_accumulation: bad argument");
    if (!ws || ws_bytes < sizeof(double) * kBlocks * 4 + 256) {
        sfm::set_error("sfm_selftest_mfma_accumulation: workspace too small (need %zu bytes)", sizeof(double) * kBlocks * 4 + 256);
        return SFM_ERR_WORKSPACE;
    }
    hipStream_t stream = sfm::as_stream(stream_);
    double* d = reinterpret_cast<double*>(sfm::align_up((size_t)(uintptr_t)ws, 256));
    static const MfmaRegime regs[7] = {
        {0, 1, 0, 1, 4, 1, 0},         // equal exponents, positive
        {-3, 6, -3, 6, 0, 8, 1},       // spread exponents, signed (cancellation)
        {-8, 16, -8, 16, -8, 24, 1},   // wide spread
        {0, 1, 0, 1, 20, 1, 1},        // c dominates
        {6, 2, 6, 2, -10, 4, 1},       // products dominate
        {-14, 4, 0, 4, -10, 8, 1},     // small operands (down to fp16's smallest normals)
        {-1, 2, -1, 2, 5, 3, 0},       // the KNN filter's own regime: operands in [0.5, 2), accumulator 32 .. 256, positive
    };
    std::vector<double> host((size_t)kBlocks * 4);
    for (int r = 0; r < 7; ++r) {
        if (use_bf16 == 2) hipLaunchKernelGGL(mfma_selftest_kernel<2>, dim3(kBlocks), dim3(256), 0, stream, regs[r], trials_per_wave, 17u + 4096u * r, d);
        else if (use_bf16) hipLaunchKernelGGL(mfma_selftest_kernel<1>, dim3(kBlocks), dim3(256), 0, stream, regs[r], trials_per_wave, 17u + 4096u * r, d);
        else hipLaunchKernelGGL(mfma_selftest_kernel<0>, dim3(kBlocks), dim3(256), 0, stream, regs[r], trials_per_wave, 17u + 4096u * r, d);
        SFM_CHECK_LAUNCH();
        SFM_CHECK_HIP(hipMemcpyAsync(host.data(), d, sizeof(double) * host.size(), hipMemcpyDeviceToHost, stream));
        SFM_CHECK_HIP(sfm::stream_sync(stream));
        double w = 0;
        for (double v : host) w = std::max(w, v);
        regime_max_host[r] = w;
    }
    return SFM_OK;
}

// ---------------------------------------------------------------- once per device: is the certificate's MFMA assumption met HERE?
// The chain term of the certificate assumes E = 16 units per MFMA, twice the E <= 8 this library requires of a device (measured
// on gfx950: <= 2.0 in the filter's regime, <= 7.1 adversarially).  The first 16-bit KNN call on a device runs the self-test
// (three MFMA kinds x seven regimes, a few ms, ONE synchronisation of a private stream) and keeps the largest E: if it exceeds
// 8, the chain term of every later call is scaled by E_measured / 8 — the certificate widens, results stay exact, rescans get
// more frequent — so an unknown matrix pipe (other silicon, firmware) degrades speed, never correctness.
namespace {
struct ChainCal {
    std::once_flag once;
    float scale = -1.f;
    double worst = 0;
};
ChainCal g_chain_cal[16];
}  // namespace

static float mfma_chain_scale() {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) {
        sfm::set_error("sfm_knn2_l2_f32: no usable HIP device");
        return -1.f;
    }
    ChainCal& c = g_chain_cal[dev];
    std::call_once(c.once, [&] {
        // test hook (tests/test_gpu_knn.py): SFM_KNN_ASSUME_E=<units> can only RAISE the measured figure — the self-test always runs and
        // worst = max(measured, assumed), so no setting of the environment makes the library less conservative than this device
        // requires (an assumed 8 used to replace a measured 20)
        const char* assume = getenv("SFM_KNN_ASSUME_E");
        double worst = assume && atof(assume) >= 8.0 ? atof(assume) : 0.0;
        {
            void* ws = nullptr;
            hipStream_t st = nullptr;
            if (hipMalloc(&ws, 32768 + 512) != hipSuccess || hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) {
                if (ws) (void)hipFree(ws);
                return;                                           // scale stays -1: the call fails loudly
            }
            bool ok = true;
            for (int kind = 0; kind < 3 && ok; ++kind) {
                double r[7];
                ok = sfm_selftest_mfma_accumulation(kind, 4, r, ws, 32768 + 512, st) == SFM_OK;
                for (double v : r) worst = std::max(worst, v);
            }
            (void)hipStreamDestroy(st);
            (void)hipFree(ws);
            if (!ok || !(worst < 1e6)) return;
        }
        c.worst = worst;
        c.scale = worst <= 8.0 ? 1.f : (float)(worst / 8.0);
    });
    if (c.scale < 0.f) sfm::set_error("sfm_knn2_l2_f32: the MFMA accumulation self-test could not run on device %d", dev);
    return c.scale;
}

extern "C" int sfm_knn_mfma_selftest_result(double* worst_units, float* chain_scale) {
    const float s = mfma_chain_scale();
    if (s < 0.f) return SFM_ERR_DEVICE;
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (worst_units) *worst_units = g_chain_cal[dev].worst;
    if (chain_scale) *chain_scale = s;
    return SFM_OK;
}

extern "C" int sfm_debug_knn_split_delay(int workgroup, int microseconds) {
    g_split_delay_wg = workgroup;
    g_split_delay_ticks = (long long)(microseconds < 0 ? 0 : microseconds > 100000 ? 100000 : microseconds) * 100;
    return SFM_OK;
}

extern "C" int sfm_debug_set_trace(void* dev_buf) {
    g_trace = static_cast<long long*>(dev_buf);
    return SFM_OK;
}

namespace {
constexpr int64_t kMaxTrainRows = 4000000;                     // fragment offsets are 32-bit: 16 KiB per 32 rows in the split images

static_assert(kFilterHalf == 5 && kFilterNoQuant == 6, "include/sfm_hip.h SFM_KNN_FILTER_*");
bool filter_ok(int filter) { return filter >= kFilterAuto && filter <= kFilterNoQuant; }

size_t knn_ws_bytes(int64_t nq, int64_t nt, int dim, int B, int filter) {
    if (nq < 0 || nt < 0 || nt > kMaxTrainRows || dim != kDim || B < 1 || B > kMaxBatch || !filter_ok(filter)) return 0;
    if (B > 1 && filter == kFilterF32) return 0;               // the fp32-MFMA variant is single-pair
    const Plan p = make_plan(nq, nt, B, filter);
    if (!p.i8) return carve_ws(nullptr, nq, nt, p, nullptr).bytes + 256;
    const Plan p8 = make_plan(nq, nt, B, p.q8 ? kFilterI8PlanQ8 : kFilterI8Plan);
    return carve_ws(nullptr, nq, nt, p, &p8).bytes + 256;
}

size_t ratio_ws_bytes(int64_t nq, int B) {
    return sfm::align_up((size_t)B * (size_t)((nq + kRatioBlock - 1) / kRatioBlock * kRatioSub + 1) * sizeof(int), 256) + 256;
}

// KNN of B equally shaped pairs in ONE set of launches (prep, filter, refine), optionally fused with the Lowe-ratio
// survivor count (ratio_counts != null: ratio_stride ints per pair, one per 1024 queries).
int knn_batch_impl(int B, const BatchPtrs& P, int64_t nq, int64_t ldq, int64_t nt, int64_t ldt, int dim, int filter, void* ws, size_t ws_bytes,
                   void* stream_, double ratio, int* ratio_counts, int ratio_stride) {
    SFM_CHECK_ARG(dim == kDim, "sfm_knn2_l2_f32: dim must be 128 (got %d)", dim);
    SFM_CHECK_ARG(B >= 1 && B <= kMaxBatch, "sfm_match_batch_l2_f32: 1 <= batch <= %d (got %d)", kMaxBatch, B);
    SFM_CHECK_ARG(filter_ok(filter), "sfm_knn2_l2_f32: filter must be 0 (auto), 1 (fp32 MFMA), 2 (bf16 split pinned), 3 / 4 (LDS-ring kernel, auto / split), 5 (16-bit only), 6 (auto, never quantised) (got %d)", filter);
    SFM_CHECK_ARG(nq >= 0 && nt >= 0 && nq < INT_MAX / 256 && nt <= kMaxTrainRows, "sfm_knn2_l2_f32: bad sizes nq=%lld nt=%lld (nt <= %lld)",
                  (long long)nq, (long long)nt, (long long)kMaxTrainRows);
    if (nq == 0) return SFM_OK;
    for (int b = 0; b < B; ++b) {
        SFM_CHECK_ARG(P.q[b] && P.idx[b] && P.dist[b] && (P.t[b] || nt == 0), "sfm_knn2_l2_f32: null pointer");
        SFM_CHECK_ARG(((uintptr_t)P.q[b] & 15) == 0 && ((uintptr_t)P.t[b] & 15) == 0, "sfm_knn2_l2_f32: q/t must be 16-byte aligned");
        SFM_CHECK_ARG(((uintptr_t)P.idx[b] & 7) == 0 && ((uintptr_t)P.dist[b] & 7) == 0, "sfm_knn2_l2_f32: idx/dist must be 8-byte aligned");
    }
    SFM_CHECK_ARG(ldq >= dim && ldt >= dim && ldq % 4 == 0 && ldt % 4 == 0, "sfm_knn2_l2_f32: ldq/ldt must be >= dim and multiples of 4");
    hipStream_t stream = sfm::as_stream(stream_);
    if (nt == 0) {   // no train rows: OpenCV leaves idx = -1 (and emits no DMatch)
        for (int b = 0; b < B; ++b)
            hipLaunchKernelGGL(knn_fill_empty_kernel, dim3((unsigned)((2 * nq + 255) / 256)), dim3(256), 0, stream, P.idx[b], P.dist[b], 2 * nq, P.stats[b]);
        SFM_CHECK_LAUNCH();
        return SFM_OK;
    }
    SFM_CHECK_ARG(B == 1 || filter != kFilterF32, "sfm_match_batch_l2_f32: the fp32-MFMA filter variant is single-pair");
    const Plan p = make_plan(nq, nt, B, filter);
    Plan p8{};
    if (p.i8) p8 = make_plan(nq, nt, B, p.q8 ? kFilterI8PlanQ8 : kFilterI8Plan);
    const size_t need = knn_ws_bytes(nq, nt, dim, B, filter);
    if (!ws || ws_bytes < need) {
        sfm::set_error("sfm_knn2_l2_f32: workspace too small (%zu < %zu)", ws_bytes, need);
        return SFM_ERR_WORKSPACE;
    }
    // carve from a 256-aligned base inside the caller's buffer
    char* base = reinterpret_cast<char*>(sfm::align_up((size_t)(uintptr_t)ws, 256));
    const KnnWs w = carve_ws(base, nq, nt, p, p.i8 ? &p8 : nullptr);
    const int nstr8 = p.i8 ? p8.smax * p8.nsub : 0;

    const dim3 grid((unsigned)p.G);
    const int prof_reps = p.split ? sfm::prof_repeat() : 1;
    static const int abl = dev_env_int("SFM_KNN_ABL", 0);       // timing ablations (WRONG results): dev builds only
    if (p.split) {
        hipLaunchKernelGGL(knn_prep_kernel, dim3(kNormBlocks + 2, (unsigned)B), dim3(kPrepThreads), 0, stream, P, ldq, (int)nq, p.nq_pad, ldt, (int)nt,
                           p.tiles * kTileT, w.qsplit, w.qn, w.tsplit, w.tn, w.bmax, w.midflag, w.qerr, w.bmaxerr, w.bqmax, w.s_qsplit, w.s_tsplit, w.s_qn, w.s_tn,
                           p.q4 ? w.qfrag : nullptr, w.tfrag, w.s_qfrag, w.s_tfrag,
                           ratio_counts, 0 /*(the refine kernel writes every count: nothing to zero)*/,
                           p.units, p.tiles, p.G, p.seg_cost, p.n_rb, w.wg_begin, w.rb_first, w.rb_last, p.q4 ? w.wg_sbase : nullptr,
                           w.qi8, w.ti8, w.s_qi8, w.s_ti8, w.wq, w.wt, w.bwmin, w.bwmax, w.rmq, w.rmt, p8.units, p8.G, p8.n_rb, w.wg_begin8, w.rb_first8,
                           w.rb_last8, w.wg_sbase8, p.q8, w.minfo);
        SFM_CHECK_LAUNCH();
        hipLaunchKernelGGL(knn_split_images_kernel, dim3(kNormBlocks), dim3(kSplitThreads), 0, stream, P, B, ldq, (int)nq, p.nq_pad, ldt, (int)nt,
                           p.tiles * kTileT, w.qsplit, w.tsplit, w.s_qsplit, w.s_tsplit, w.midflag, w.bmax, p.force_mode,
                           p.q4 ? w.qhm : nullptr, w.thm, w.s_qhm, w.s_thm, w.bmaxerr, w.bqmax, w.minfo, w.ti8, w.s_ti8, w.wt, w.s_tn, w.bwmin, w.bwmax,
                           w.qi8, w.s_qi8, w.s_qn, w.rmq, w.rmt, w.qfrag, w.tfrag, w.s_qfrag, w.s_tfrag, w.qerr, w.midflag_r, w.bmaxerr_r, g_split_delay_wg, g_split_delay_ticks);
        SFM_CHECK_LAUNCH();
        sfm::prof_begin(sfm::kProfKnnFilter, stream);
#define SFM_LAUNCH_SPLIT2(A, WV)                                                                                        \
    hipLaunchKernelGGL((knn_filter_split2_kernel<A, WV>), grid, dim3(64 * WV), kRingLdsBytes + (WV == 4 ? kQScratchBytes : 0), stream, w.qsplit, w.qn,    \
                       (int)nq, p.nq_pad, w.tsplit, (int)nt, p.tiles * kTileT, w.tn, p.tiles, p.units, p.smax, p.nsub,    \
                       w.midflag, w.bmax, p.force_mode, w.cand_s, w.cand_i, w.wg_begin, w.rb_first, p.n_rb1, B, w.s_qsplit, w.s_tsplit,    \
                       w.s_qn, w.s_tn, w.s_cand, w.minfo, w.bmaxerr, g_trace)
        // sfm_profile_enable(n > 1): the filter is launched n times back-to-back inside ONE event pair (idempotent: same
        // inputs, same candidate records), so the ~7 us an event pair adds to a single launch is amortised
        for (int rep = 0; rep < prof_reps; ++rep) {
        if (p.q4) {
#define SFM_LAUNCH_Q4(A)                                                                                                                         \
    hipLaunchKernelGGL(knn_filter_q4_kernel<A>, grid, dim3(256), 0, stream, w.qfrag, w.tfrag, w.qhm, w.thm, (int)nq, p.nq_pad, p.tiles, p.smax, p.nsub, \
                       w.midflag, w.bmax, p.force_mode, w.cand_s, w.cand_i, w.wg_begin, w.rb_first, p.n_rb1, B, w.s_qfrag, w.s_tfrag, w.s_qhm,         \
                       w.s_thm, w.s_cand, w.minfo, w.bmaxerr, w.wg_sbase, g_trace, w.qi8, w.ti8, w.s_qi8, w.s_ti8, w.keys8, w.sttab8, w.s_keys8, nstr8,         \
                       w.wg_begin8, p8.n_rb1, w.wg_sbase8, p8.G, (int)nt)
            if (abl == 1) SFM_LAUNCH_Q4(1); else if (abl == 2) SFM_LAUNCH_Q4(2); else if (abl == 3) SFM_LAUNCH_Q4(3); else if (abl == 4) SFM_LAUNCH_Q4(4); else SFM_LAUNCH_Q4(0);
#undef SFM_LAUNCH_Q4
        } else if (p.waves == 4) {
            if (abl == 1) SFM_LAUNCH_SPLIT2(1, 4); else if (abl == 2) SFM_LAUNCH_SPLIT2(2, 4); else if (abl == 4) SFM_LAUNCH_SPLIT2(4, 4);
            else if (abl == 6) SFM_LAUNCH_SPLIT2(6, 4); else if (abl == 7) SFM_LAUNCH_SPLIT2(7, 4); else SFM_LAUNCH_SPLIT2(0, 4);
        } else if (p.waves == 16) {
            if (abl == 1) SFM_LAUNCH_SPLIT2(1, 16); else if (abl == 7) SFM_LAUNCH_SPLIT2(7, 16); else SFM_LAUNCH_SPLIT2(0, 16);
        } else {
            if (abl == 1) SFM_LAUNCH_SPLIT2(1, 8); else if (abl == 7) SFM_LAUNCH_SPLIT2(7, 8); else SFM_LAUNCH_SPLIT2(0, 8);
        }
        }
#undef SFM_LAUNCH_SPLIT2
    } else {
    const float* q = P.q[0];
    const float* t = P.t[0];
    hipLaunchKernelGGL(knn_norms_kernel, dim3(kNormBlocks + 1), dim3(256), 0, stream, t, ldt, (int)nt, w.tn, w.bmax,
                       P.stats[0], ratio_counts, 0, p.units, p.tiles, p.G, p.seg_cost, p.n_rb, w.wg_begin, w.rb_first, w.rb_last);
    SFM_CHECK_LAUNCH();
    sfm::prof_begin(sfm::kProfKnnFilter, stream);
#define SFM_LAUNCH_FILTER(A, WV)                                                                                     \
    hipLaunchKernelGGL((knn_filter_kernel<A, WV>), grid, dim3(64 * WV), kLdsFloats * sizeof(float), stream, q, ldq,    \
                       (int)nq, t, ldt, (int)nt, w.tn, p.tiles, p.units, p.smax, p.nsub, w.cand_s, w.cand_i, g_trace)
    if (p.waves == 4) {
        switch (abl) {
            case 1: SFM_LAUNCH_FILTER(1, 4); break;
            case 7: SFM_LAUNCH_FILTER(7, 4); break;
            default: SFM_LAUNCH_FILTER(0, 4); break;
        }
    } else if (p.waves == 8) {
        switch (abl) {
            case 1: SFM_LAUNCH_FILTER(1, 8); break;
            case 7: SFM_LAUNCH_FILTER(7, 8); break;
            default: SFM_LAUNCH_FILTER(0, 8); break;
        }
    } else {
        switch (abl) {
            case 1: SFM_LAUNCH_FILTER(1, 16); break;
            case 7: SFM_LAUNCH_FILTER(7, 16); break;
            default: SFM_LAUNCH_FILTER(0, 16); break;
        }
    }
#undef SFM_LAUNCH_FILTER
    }
    sfm::prof_end(sfm::kProfKnnFilter, stream, prof_reps);
    SFM_CHECK_LAUNCH();
    const int force_mode = !p.split ? kModeF32 : p.force_mode;
    float chain_scale = 1.f;
    if (p.split) {
        chain_scale = mfma_chain_scale();
        if (chain_scale < 0.f) return SFM_ERR_DEVICE;
    }
    const int64_t refine_wgs = (int64_t)B * ((nq + kRefQ - 1) / kRefQ), refine_grid = refine_wgs >= 64 ? 8 * ((refine_wgs + 7) / 8) : refine_wgs;
    sfm::prof_begin(sfm::kProfKnnRefine, stream);
#define SFM_LAUNCH_REFINE(FRAG, THALF, S_THALF)                                                                                          \
    hipLaunchKernelGGL(knn_refine_kernel<FRAG>, dim3((unsigned)refine_grid), dim3(256), 0, stream, P, B, ldq, (int)nq, ldt, (int)nt, w.cand_s,    \
                       w.cand_i, p.rows_per_block, p.tiles, p.units, p.G, p.smax * p.nsub, p.nsub, force_mode, w.midflag, w.bmax,               \
                       p.split ? w.minfo : nullptr, p.split ? w.qerr : nullptr, w.s_qn, THALF, w.tn, w.wg_begin, w.rb_first, w.rb_last, p.n_rb1, \
                       w.s_cand, S_THALF, w.s_tn, ratio, ratio_counts, ratio_stride, chain_scale, p.q4 ? 1 : 0, g_trace ? g_trace + 16384 : nullptr,    \
                       w.qi8, w.ti8, w.s_qi8, w.s_ti8, w.wq, w.wt, w.keys8, w.sttab8, w.s_keys8, nstr8, w.rb_last8, p8.n_rb1, p8.G)
    if (p.q4) SFM_LAUNCH_REFINE(true, reinterpret_cast<const unsigned short*>(w.tfrag), w.s_tfrag / 2);      // (stride in 16-bit elements)
    else SFM_LAUNCH_REFINE(false, p.split ? w.tsplit + (size_t)2 * p.tiles * kTileT * kDim : nullptr, w.s_tsplit);
#undef SFM_LAUNCH_REFINE
    sfm::prof_end(sfm::kProfKnnRefine, stream);
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}

BatchPtrs single_pair(const float* q, const float* t, int32_t* idx, float* dist, int32_t* stats, int32_t* out_q, int32_t* out_t,
                      int32_t* out_count, uint8_t* mask) {
    BatchPtrs P{};
    P.q[0] = q; P.t[0] = t; P.idx[0] = idx; P.dist[0] = dist; P.stats[0] = stats;
    P.out_q[0] = out_q; P.out_t[0] = out_t; P.out_count[0] = out_count; P.mask[0] = mask;
    return P;
}
}  // namespace

extern "C" size_t sfm_knn2_l2_f32_ws_bytes(int64_t nq, int64_t nt, int dim, int filter) { return knn_ws_bytes(nq, nt, dim, 1, filter); }

extern "C" int sfm_knn2_l2_f32(const float* q, int64_t nq, int64_t ldq, const float* t, int64_t nt, int64_t ldt,
                               int dim, int filter, int32_t* idx, float* dist, int32_t* stats, void* ws, size_t ws_bytes,
                               void* stream_) {
    return knn_batch_impl(1, single_pair(q, t, idx, dist, stats, nullptr, nullptr, nullptr, nullptr), nq, ldq, nt, ldt, dim, filter, ws, ws_bytes,
                          stream_, 0.0, nullptr, 0);
}

extern "C" size_t sfm_ratio_compact_ws_bytes(int64_t nq) {
    if (nq < 0) return 0;
    return sfm::align_up((size_t)((nq + kRatioBlock - 1) / kRatioBlock + 1) * sizeof(int), 256) + 256;
}

extern "C" int sfm_ratio_compact(const int32_t* idx, const float* dist, int64_t nq, double ratio, int32_t* out_q,
                                 int32_t* out_t, int32_t* out_count, uint8_t* mask, void* ws, size_t ws_bytes,
                                 void* stream_) {
    SFM_CHECK_ARG(nq >= 0 && nq < INT_MAX, "sfm_ratio_compact: bad nq");
    SFM_CHECK_ARG(out_count && (nq == 0 || (idx && dist && out_q && out_t)), "sfm_ratio_compact: null pointer");
    SFM_CHECK_ARG(((uintptr_t)idx & 7) == 0 && ((uintptr_t)dist & 7) == 0, "sfm_ratio_compact: idx/dist must be 8-byte aligned");
    hipStream_t stream = sfm::as_stream(stream_);
    if (nq == 0) {
        SFM_CHECK_HIP(hipMemsetAsync(out_count, 0, sizeof(int32_t), stream));
        return SFM_OK;
    }
    if (!ws || ws_bytes < sfm_ratio_compact_ws_bytes(nq)) {
        sfm::set_error("sfm_ratio_compact: workspace too small (%zu < %zu)", ws_bytes, sfm_ratio_compact_ws_bytes(nq));
        return SFM_ERR_WORKSPACE;
    }
    int* counts = reinterpret_cast<int*>(sfm::align_up((size_t)(uintptr_t)ws, 256));
    const unsigned blocks = (unsigned)((nq + kRatioBlock - 1) / kRatioBlock);
    hipLaunchKernelGGL(ratio_count_kernel, dim3(blocks), dim3(256), 0, stream, idx, dist, (int)nq, ratio, counts, mask);
    SFM_CHECK_LAUNCH();
    hipLaunchKernelGGL(ratio_scatter_kernel, dim3(blocks), dim3(256), 0, stream, idx, dist, (int)nq, ratio, counts, out_q,
                       out_t, out_count);
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}

extern "C" size_t sfm_match_batch_l2_f32_ws_bytes(int64_t nq, int64_t nt, int dim, int batch, int filter) {
    const size_t k = knn_ws_bytes(nq, nt, dim, batch, filter);
    return k ? k + ratio_ws_bytes(nq, batch) : 0;
}

extern "C" size_t sfm_match_l2_f32_ws_bytes(int64_t nq, int64_t nt, int dim, int filter) { return sfm_match_batch_l2_f32_ws_bytes(nq, nt, dim, 1, filter); }

namespace {
int match_batch_impl(int B, const BatchPtrs& P, int64_t nq, int64_t ldq, int64_t nt, int64_t ldt, int dim, int filter, double ratio, void* ws,
                     size_t ws_bytes, void* stream_) {
    SFM_CHECK_ARG(B >= 1 && B <= kMaxBatch, "sfm_match_batch_l2_f32: 1 <= batch <= %d (got %d)", kMaxBatch, B);
    SFM_CHECK_ARG(nq >= 0 && nq < INT_MAX / 256, "sfm_match_l2_f32: bad nq");
    for (int b = 0; b < B; ++b) SFM_CHECK_ARG(P.out_count[b] && (nq == 0 || (P.out_q[b] && P.out_t[b])), "sfm_match_l2_f32: null pointer");
    hipStream_t stream = sfm::as_stream(stream_);
    if (nq == 0 || nt == 0) {   // no neighbours, no matches
        for (int b = 0; b < B; ++b) {
            SFM_CHECK_HIP(hipMemsetAsync(P.out_count[b], 0, sizeof(int32_t), stream));
            if (P.mask[b] && nq > 0) SFM_CHECK_HIP(hipMemsetAsync(P.mask[b], 0, (size_t)nq, stream));
        }
        return knn_batch_impl(B, P, nq, ldq, nt, ldt, dim, filter, ws, ws_bytes, stream_, 0.0, nullptr, 0);
    }
    const size_t rbytes = ratio_ws_bytes(nq, B);
    const size_t need = sfm_match_batch_l2_f32_ws_bytes(nq, nt, dim, B, filter);
    if (!ws || ws_bytes < need || need == 0) {
        SFM_CHECK_ARG(dim == kDim, "sfm_match_l2_f32: dim must be 128 (got %d)", dim);
        SFM_CHECK_ARG(filter_ok(filter), "sfm_match_l2_f32: bad filter variant %d", filter);
        SFM_CHECK_ARG(nt <= kMaxTrainRows, "sfm_match_l2_f32: nt <= %lld", (long long)kMaxTrainRows);
        SFM_CHECK_ARG(B == 1 || filter != kFilterF32, "sfm_match_batch_l2_f32: the fp32-MFMA filter variant is single-pair");
        sfm::set_error("sfm_match_l2_f32: workspace too small (%zu < %zu)", ws_bytes, need);
        return SFM_ERR_WORKSPACE;
    }
    int* counts = reinterpret_cast<int*>(sfm::align_up((size_t)(uintptr_t)ws, 256));
    const unsigned blocks = (unsigned)((nq + kRatioBlock - 1) / kRatioBlock);
    const int stride = (int)blocks * kRatioSub + 1;            // per-wave survivor counts, written (every one of them) by the refine kernel
    const int rc = knn_batch_impl(B, P, nq, ldq, nt, ldt, dim, filter, static_cast<char*>(ws) + rbytes, ws_bytes - rbytes, stream_, ratio, counts, stride);
    if (rc != SFM_OK) return rc;
    hipLaunchKernelGGL(ratio_scatter_batch_kernel, dim3(blocks, (unsigned)B), dim3(256), 0, stream, P, (int)nq, ratio, counts, stride);
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}
}  // namespace

extern "C" int sfm_match_l2_f32(const float* q, int64_t nq, int64_t ldq, const float* t, int64_t nt, int64_t ldt, int dim, int filter,
                                double ratio, int32_t* idx, float* dist, int32_t* out_q, int32_t* out_t,
                                int32_t* out_count, uint8_t* mask, int32_t* stats, void* ws, size_t ws_bytes,
                                void* stream_) {
    SFM_CHECK_ARG(out_count && (nq == 0 || (out_q && out_t)), "sfm_match_l2_f32: null pointer");
    return match_batch_impl(1, single_pair(q, t, idx, dist, stats, out_q, out_t, out_count, mask), nq, ldq, nt, ldt, dim, filter, ratio, ws, ws_bytes, stream_);
}

// `batch` (<= 8) image pairs of one shape in ONE set of launches: the unit space of the filter is batch x row blocks x
// tiles under a single partition, so a workgroup's prologue, the launch ramp and the kernel boundaries are paid once
// per batch instead of once per pair.  Pointer arrays are HOST arrays of device pointers (mask / stats entries may be NULL).
extern "C" int sfm_match_batch_l2_f32(int batch, const float* const* q, int64_t nq, int64_t ldq, const float* const* t, int64_t nt,
                                      int64_t ldt, int dim, int filter, double ratio, int32_t* const* idx, float* const* dist,
                                      int32_t* const* out_q, int32_t* const* out_t, int32_t* const* out_count, uint8_t* const* mask,
                                      int32_t* const* stats, void* ws, size_t ws_bytes, void* stream_) {
    SFM_CHECK_ARG(batch >= 1 && batch <= kMaxBatch, "sfm_match_batch_l2_f32: 1 <= batch <= %d (got %d)", kMaxBatch, batch);
    SFM_CHECK_ARG(q && t && idx && dist && out_q && out_t && out_count, "sfm_match_batch_l2_f32: null pointer array");
    BatchPtrs P{};
    for (int b = 0; b < batch; ++b) {
        P.q[b] = q[b]; P.t[b] = t[b]; P.idx[b] = idx[b]; P.dist[b] = dist[b];
        P.out_q[b] = out_q[b]; P.out_t[b] = out_t[b]; P.out_count[b] = out_count[b];
        P.mask[b] = mask ? mask[b] : nullptr;
        P.stats[b] = stats ? stats[b] : nullptr;
    }
    return match_batch_impl(batch, P, nq, ldq, nt, ldt, dim, filter, ratio, ws, ws_bytes, stream_);
}

extern "C" int sfm_gather_matches(const float* kp0, const float* kp1, const int32_t* out_q, const int32_t* out_t,
                                  const int32_t* count, int64_t capacity, float* pts0, float* pts1, void* stream_) {
    SFM_CHECK_ARG(kp0 && kp1 && out_q && out_t && count && pts0 && pts1 && capacity >= 0, "sfm_gather_matches: bad argument");
    if (capacity == 0) return SFM_OK;
    unsigned blocks = (unsigned)((capacity + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(gather_matches_kernel, dim3(blocks), dim3(256), 0, sfm::as_stream(stream_),
                       reinterpret_cast<const float2*>(kp0), reinterpret_cast<const float2*>(kp1), out_q, out_t, count,
                       capacity, reinterpret_cast<float2*>(pts0), reinterpret_cast<float2*>(pts1));
    SFM_CHECK_LAUNCH();
    return SFM_OK;
}
