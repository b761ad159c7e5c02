// nwknn_kernels.hip -- gfx950 (MI355X, CDNA4) kernels of the brute-force kNN ground-truth hot path.
//
// What this replaces in the reference (paths relative to /root/reference):
//   neighborhoodwatch/cu_knn.py:243-245   knn(dataset.astype(f32), query.astype(f32), k)   (pylibraft, sqeuclidean)
//   neighborhoodwatch/cu_knn.py:246-248   cuvs brute_force build/search, metric="cosine"
//   neighborhoodwatch/cu_knn.py:256-257   1 - torch.matmul(q, b.T); torch.topk(.., largest=False)
//   neighborhoodwatch/cu_knn.py:211,235   columnar -> row-major copies          (prep kernel writes the MFMA-ready layout)
//   neighborhoodwatch/cu_knn.py:273       indices + batch_offset                (finalize kernel adds the shard offset)
//   neighborhoodwatch/merge.py:73-101     [Q,2k] argsort fold of partial lists  (merge kernel)
//
// Design (DESIGN.md has the full story):
//   * distance tile = f32 MFMA GEMM (v_mfma_f32_32x32x2_f32): 128 queries x 128 base rows per workgroup,
//     K streamed in 32-wide slabs through double-buffered LDS filled by LDS-DMA (global_load_lds_dwordx4).
//     One accumulator carries the whole K loop in ascending k, so every dot product is bit-for-bit the
//     k-ordered fmaf chain of oracle/nwknn_oracle.c.
//   * the Q x B distance matrix never exists: the epilogue turns each accumulator into a 64-bit key
//     (sortable(distance) << 32 | base row) and keeps only keys below the query's current k-th key (tau);
//     survivors are appended to a per-query candidate list with half-wave aggregated atomics.
//   * a compaction kernel sorts each query's list (bitonic, LDS), keeps k, and tightens tau; the host walks the
//     base in geometrically growing chunks so that tau is already tight when the big chunks run.
//   * all integer/key work is exact: 64-bit key order == (distance, index) order, lower index wins ties.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "nwknn_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;
typedef unsigned int u32;

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 sortable_u32(float f)
{
    u32 u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float unsortable_f32(u32 s)
{
    u32 u = (s & 0x80000000u) ? (s & 0x7FFFFFFFu) : ~s;
    return __uint_as_float(u);
}

// canonical epilogue: must stay expression-for-expression equal to nwo_epilogue() in oracle/nwknn_oracle.c
template <int METRIC>
__device__ __forceinline__ float distance_epilogue(float dot, float qn, float bn)
{
    if (METRIC == NWKNN_METRIC_SQEUCLIDEAN) return fmaxf(__fmaf_rn(-2.0f, dot, __fadd_rn(qn, bn)), 0.0f);
    if (METRIC == NWKNN_METRIC_ONE_MINUS_DOT) return __fsub_rn(1.0f, dot);
    // sqrtf() lowers to the correctly rounded sequence; __fsqrt_rn is a bare v_sqrt_f32 (1 ulp) on gfx950
    return __fsub_rn(1.0f, __fdiv_rn(dot, __fmul_rn(sqrtf(qn), sqrtf(bn))));
}

// make a pointer provably wave-uniform (SGPR pair) so that global_load_lds takes the saddr + 32-bit voffset form
__device__ __forceinline__ const char* uniform_ptr(const char* q)
{
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)q);
    const unsigned hi = __builtin_amdgcn_readfirstlane((unsigned)((uintptr_t)q >> 32));
    return (const char*)(((uintptr_t)hi << 32) | (uintptr_t)lo);
}

__device__ __forceinline__ u64 fmix64(u64 z)
{
    z ^= z >> 30; z *= 0xBF58476D1CE4E5B9ull;
    z ^= z >> 27; z *= 0x94D049BB133111EBull;
    z ^= z >> 31;
    return z;
}

// ------------------------------------------------------------------------------------------------
// synthetic generator: value(seed,row,col) on a 2^-23 grid in [-1,1)  (oracle nwo_gen_value)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nwk_gen_kernel(u64 seed, long long row0, long long rows, int d, float* __restrict__ out)
{
    const long long total = rows * (long long)d;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / d;
        const int c = (int)(i - r * d);
        const u64 rk = fmix64(seed + (u64)(row0 + r) * 0x9E3779B97F4A7C15ull);
        const u64 u = fmix64(rk + (u64)(c + 1) * 0xD1342543DE82EF95ull);
        const int v = (int)(u >> 40) - (1 << 23);
        out[i] = (float)v * (1.0f / 8388608.0f);
    }
}

// ------------------------------------------------------------------------------------------------
// prep: raw row-major f32 [rows][d]  ->  MFMA-ready layout [rows_pad][dp] + squared norms
//   * dp = d rounded up to 32 (zero padded: fma(0,0,acc) leaves every chain unchanged)
//   * inside every aligned group of 8 k's the order is [0 2 4 6 | 1 3 5 7]: one ds_read_b128 then hands
//     lanes 0-31 the even k's and lanes 32-63 the odd k's of four consecutive v_mfma_f32_32x32x2_f32, which
//     therefore consume k in ascending order
//   * norm = the same ascending fmaf chain the MFMA performs for q.q (one lane owns one row)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nwk_prep_kernel(const float* __restrict__ raw, long long rows, int d, long long rows_pad,
                                                        int dp, float* __restrict__ prepared, float* __restrict__ norms)
{
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows_pad) return;
    float* dst = prepared + r * dp;
    if (r >= rows) {
        for (int k = 0; k < dp; k += 4) *(f32x4*)(dst + k) = (f32x4){0.f, 0.f, 0.f, 0.f};
        norms[r] = 0.0f;
        return;
    }
    const float* src = raw + r * (long long)d;
    float acc = 0.0f;
    const bool vec = ((d & 3) == 0) && ((((uintptr_t)raw) & 15) == 0);
    for (int k0 = 0; k0 < dp; k0 += 8) {
        float x[8];
        if (vec && k0 + 8 <= d) {
            const f32x4 a = *(const f32x4*)(src + k0), b = *(const f32x4*)(src + k0 + 4);
            x[0] = a[0]; x[1] = a[1]; x[2] = a[2]; x[3] = a[3]; x[4] = b[0]; x[5] = b[1]; x[6] = b[2]; x[7] = b[3];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) x[j] = (k0 + j < d) ? src[k0 + j] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) acc = __fmaf_rn(x[j], x[j], acc);
        *(f32x4*)(dst + k0) = (f32x4){x[0], x[2], x[4], x[6]};
        *(f32x4*)(dst + k0 + 4) = (f32x4){x[1], x[3], x[5], x[7]};
    }
    norms[r] = acc;
}

// ------------------------------------------------------------------------------------------------
// the distance-tile + filter kernel
// ------------------------------------------------------------------------------------------------
#define TQ 128
#define TB 128
#define BK 32
#define STAGE_BYTES 32768          // A tile 16 KiB + B tile 16 KiB
#define B_TILE_OFF 16384

// LDS image of one operand tile: 128 rows x 32 floats (128 B rows = eight 16-B chunks); logical chunk c of row r
// lives in slot c ^ ((r >> 1) & 7).  ds_read_b128 by 32 consecutive rows is then conflict-free (every 16-lane
// service group touches 16 distinct 16-B slots of the 256-B bank row).  The LDS-DMA writes linearly
// (base + lane*16), so the swizzle is applied to the per-lane SOURCE address (guide rule 21).

struct TileCoord { int qt, bt, ok; };

__device__ __forceinline__ TileCoord tile_from_block(const NwkTileParams& p)
{
    // XCD-aware mapping: blocks b and b+8 share an XCD (and its L2); give each XCD 8x8 super-tiles so that the
    // 64 co-resident workgroups of an XCD share 8 query tiles and 8 base tiles.
    const int bid = blockIdx.x;
    const int xcd = bid & 7, local = bid >> 3;
    const long long super = (long long)(local >> 6) * 8 + xcd;
    const int within = local & 63;
    const int sq = (int)(super / p.n_super_b), sb = (int)(super % p.n_super_b);
    const int qi = sq * 8 + (within >> 3), bi = sb * 8 + (within & 7);
    TileCoord t;
    t.ok = (qi < p.n_qtiles) && (bi < p.n_btiles);
    t.qt = t.ok ? (p.qtile_list ? p.qtile_list[qi] : qi) : 0;
    t.bt = p.bt0 + bi;
    return t;
}

// ------------------------------------------------------------------------------------------------
// shared epilogue: 2x2 accumulators of one wave -> distances -> 64-bit keys -> filter against tau -> candidate lists.
// s_qn / s_tau hold the tile's 128 query norms / thresholds in LDS.  For one 32-row block the 16 rows a lane owns are
// rows rb+8j .. rb+8j+3 (j = 0..3, rb = block + 4*half), so norms and thresholds come in with 12 ds_read_b128 issued
// back to back and the 32 elements that follow are pure VALU (the first version interleaved two LDS round trips and a
// branch per element: ~60k cycles per tile; this form: a few thousand).
// ------------------------------------------------------------------------------------------------
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));

template <int METRIC, bool DENSE>
__device__ __forceinline__ void tile_epilogue(const NwkTileParams& p, const f32x16 (&acc)[2][2], const float* s_qn, const u64* s_tau,
                                              const float (&bnv)[2], const u32 (&colv)[2], const bool (&colok)[2], long long q0, int wr,
                                              int h, int c32)
{
    const int cap = p.cap;
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int rb = wr * 64 + m * 32 + 4 * h;
        f32x4 qn4[4];
        u64x2 tl[4], th[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            qn4[j] = *(const f32x4*)(s_qn + rb + 8 * j);
            if (!DENSE) {
                tl[j] = *(const u64x2*)(s_tau + rb + 8 * j);
                th[j] = *(const u64x2*)(s_tau + rb + 8 * j + 2);
            }
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            u64 key[16];
            bool pass[16];
            bool any = false;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float dist = distance_epilogue<METRIC>(acc[m][n][r], qn4[r >> 2][r & 3], bnv[n]);
                const u32 sd = (METRIC == NWKNN_METRIC_SQEUCLIDEAN) ? (__float_as_uint(dist) | 0x80000000u)   // dist >= +0, never NaN
                                                                     : sortable_u32(dist);
                key[r] = ((u64)sd << 32) | colv[n];
                if (!DENSE) {
                    const u64 tau = (r & 2) ? th[r >> 2][r & 1] : tl[r >> 2][r & 1];
                    pass[r] = (key[r] < tau) & colok[n];
                    any |= pass[r];
                }
            }
            if (DENSE) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = rb + (r & 3) + 8 * (r >> 2);
                    const long long slot = (long long)(colv[n] - (u32)p.dense_col0);
                    p.buf[(q0 + row) * cap + slot] = colok[n] ? key[r] : ~0ull;
                }
            } else if (__ballot(any)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const u64 mk = __ballot(pass[r]);
                    if (mk) {                                        // wave-uniform
                        const int row = rb + (r & 3) + 8 * (r >> 2);
                        const u32 mh = h ? (u32)(mk >> 32) : (u32)mk; // this half-wave's survivors: all for ONE query row
                        const int leader = mh ? (__ffs(mh) - 1) : 0;
                        u32 base = 0;
                        if (mh && c32 == leader) base = atomicAdd(p.cnt + q0 + row, (u32)__popc(mh));
                        base = __shfl(base, leader + (h << 5));
                        if (pass[r]) {
                            const u32 pos = base + __popc(mh & ((1u << c32) - 1u));
                            if (pos < (u32)cap) p.buf[(q0 + row) * cap + pos] = key[r];
                        }
                    }
                }
            }
        }
    }
}

// DIAG (timing experiments only, results are wrong when != 0): 1 = no LDS fragment reads in the loop, 2 = no DMA after
// slab 0, 4 = no barriers.  Never selected by the product (variant 0..4 pass DIAG = 0).
template <int METRIC, bool DENSE, int DIAG = 0>
__global__ __launch_bounds__(256, 2) void nwk_tile_kernel(const NwkTileParams p)
{
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES];

    unsigned long long st0 = 0, st1 = 0, st2 = 0, rt0 = 0;
    if (p.dbg) { st0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    const TileCoord tc = tile_from_block(p);
    if (!tc.ok) return;
    // De-phase the two workgroups that share a CU.  The first 512 blocks of a launch become resident together, two per
    // CU (blocks 256..511 are the second round); with equal tile times both would reach their epilogues -- where they
    // issue no MFMA -- at the same moment for the whole launch.  Holding the second round back by about half a tile
    // once keeps one of the two in its K loop at all times.  Placement is not guaranteed: a wrong guess costs nothing
    // but the delay.
    if (p.stagger > 0 && blockIdx.x >= 256 && blockIdx.x < 512) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        while (__builtin_amdgcn_s_memtime() - t0 < (unsigned long long)p.stagger) __builtin_amdgcn_s_sleep(32);
    }

    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63;
    const int wr = w >> 1, wc = w & 1;
    const int h = l >> 5, c32 = l & 31;
    const long long q0 = (long long)tc.qt * TQ;
    const long long b0 = (long long)tc.bt * TB;
    const int dp = p.dp;
    const int nk = dp / BK;

    // ---- LDS-DMA sources: wave w fills rows [w*32, w*32+32) of both operand tiles, 8 rows per instruction.
    // Address = wave-uniform tile base (SGPRs) + 32-bit per-lane byte offset: the saddr form of global_load_lds
    // (half the address payload of a 64-bit vaddr and no 64-bit VALU add per DMA).
    unsigned voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (w * 4 + i) * 8 + (l >> 3);
        const int chunk = (l & 7) ^ ((row >> 1) & 7);
        voff[i] = (unsigned)(((DIAG & 16) ? 0 : row) * dp + chunk * 4) * 4u;
    }
    // one buffer descriptor per operand tile (wave-uniform), slab offset in an SGPR: buffer_load_dwordx4 ... offen lds
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.qp + ((DIAG & 16) ? 0 : q0) * dp), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bp + ((DIAG & 16) ? 0 : b0) * dp), 0, 0x7FFFFFFF, 0x00020000);
    const int wave_lds = __builtin_amdgcn_readfirstlane(w * 4096);   // wave-uniform LDS byte offset of this wave's 4 KiB strip

    auto issue_stage = [&](int stage, int kt) {
        char* base = smem + stage * STAGE_BYTES + wave_lds;
        const int soff = kt * (BK * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)(base + i * 1024), 16, voff[i], soff, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if ((DIAG & 8) && kt > 0) continue;                                    // DIAG 8: half the DMA instructions
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)(base + B_TILE_OFF + i * 1024), 16, voff[i], soff, 0, 0);
        }
    };

    // ---- fragment read addresses: lane (h, c32) reads 16 B of row (blk*32 + c32), logical chunk 2g+h
    int offA[2], offB[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int ra = wr * 64 + m * 32 + c32;
        const int rb = wc * 64 + m * 32 + c32;
        offA[m] = ra * 128 + ((h ^ ((ra >> 1) & 7)) << 4);
        offB[m] = B_TILE_OFF + rb * 128 + ((h ^ ((rb >> 1) & 7)) << 4);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    // column side of the epilogue, fetched early
    float bnv[2];
    u32 colv[2];
    bool colok[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const long long col = b0 + wc * 64 + n * 32 + c32;
        colv[n] = (u32)col;
        colok[n] = col < p.nb_valid;
        bnv[n] = p.bn[col];
    }
    // row side (norm + threshold of query row q0+tid), in flight under the K loop
    float my_qn = 0.f;
    u64 my_tau = ~0ull;
    if (tid < TQ) {
        my_qn = p.qn[q0 + tid];
        if (!DENSE) my_tau = p.tau[q0 + tid];
    }

    issue_stage(0, 0);
    if (p.dbg) st1 = __builtin_amdgcn_s_memtime();
    f32x4 fa[2], fb[2];
    if (DIAG & 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int m = 0; m < 2; ++m) { fa[m] = *(const f32x4*)(smem + offA[m]); fb[m] = *(const f32x4*)(smem + offB[m]); }
    }
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            if (!(DIAG & 2)) issue_stage(cur ^ 1, kt + 1);
            if (DIAG & 8) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");     // the 8 loads of stage `cur` have landed
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!(DIAG & 4)) __builtin_amdgcn_s_barrier();
        const char* sb = smem + cur * STAGE_BYTES;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                if (DIAG & 1) {
                    a[m] = fa[m]; b[m] = fb[m];
                    asm volatile("" : "+v"(a[m]), "+v"(b[m]));       // keep the MFMAs from being merged
                } else {
                    a[m] = *(const f32x4*)(sb + (offA[m] ^ (g << 5)));
                    b[m] = *(const f32x4*)(sb + (offB[m] ^ (g << 5)));
                }
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][t], b[n][t], acc[m][n], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(DIAG & 4)) __builtin_amdgcn_s_barrier();                // everyone is done reading stage `cur`
    }

    if (p.dbg) st2 = __builtin_amdgcn_s_memtime();
    // ---- epilogue: accumulators -> keys -> filter -> candidate lists
    float* s_qn = (float*)smem;
    u64* s_tau = (u64*)(smem + 1024);
    if (tid < TQ) {
        s_qn[tid] = my_qn;
        if (!DENSE) s_tau[tid] = my_tau;
    }
    __syncthreads();
    tile_epilogue<METRIC, DENSE>(p, acc, s_qn, s_tau, bnv, colv, colok, q0, wr, h, c32);
    if (p.dbg && tid == 0) {
        unsigned long long* o = p.dbg + (size_t)blockIdx.x * 8;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_amdgcn_s_memtime(); o[4] = rt0; o[5] = __builtin_amdgcn_s_memrealtime();
        o[6] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); o[7] = 1;      // HW_REG_XCC_ID (id 20), bits [3:0]
    }
}

// ------------------------------------------------------------------------------------------------
// pipelined form of the tile kernel: same tile, same 2 x 32 KiB stages, but
//   * fragment registers are double buffered: the ds_reads of 8-k group g+1 are issued BEFORE the 16 MFMAs of group g;
//   * ONE barrier per slab, placed before the MFMAs of the slab's last group: by then this wave has issued (and
//     waited for) all its LDS reads of slab s, and slab s+1's DMA has landed -- so the same barrier publishes slab s+1
//     (RAW) and releases slab s's buffer (WAR); the DMA of slab s+2 and the first fragment reads of slab s+1 go out
//     right behind it and their latency hides under the 16 MFMAs of the last group.
// A wave alone on its SIMD therefore never waits on LDS latency, only on barrier skew.
// ------------------------------------------------------------------------------------------------
#define EPI_OFF (2 * STAGE_BYTES)
#define EPI_BYTES 1536

template <int METRIC, bool DENSE>
__global__ __launch_bounds__(256, 2) void nwk_tile_kernel_pipe(const NwkTileParams p)
{
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES + EPI_BYTES];

    const TileCoord tc = tile_from_block(p);
    if (!tc.ok) return;

    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63;
    const int wr = w >> 1, wc = w & 1;
    const int h = l >> 5, c32 = l & 31;
    const long long q0 = (long long)tc.qt * TQ;
    const long long b0 = (long long)tc.bt * TB;
    const int dp = p.dp;
    const int nk = dp / BK;

    unsigned voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (w * 4 + i) * 8 + (l >> 3);
        voff[i] = (unsigned)(row * dp + (((l & 7) ^ ((row >> 1) & 7)) << 2)) * 4u;
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.qp + q0 * dp), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bp + b0 * dp), 0, 0x7FFFFFFF, 0x00020000);
    const int wave_lds = __builtin_amdgcn_readfirstlane(w * 4096);
    auto issue_stage = [&](int stage, int kt) {
        char* base = smem + stage * STAGE_BYTES + wave_lds;
        const int soff = kt * (BK * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)(base + i * 1024), 16, voff[i], soff, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)(base + B_TILE_OFF + i * 1024), 16, voff[i], soff, 0, 0);
    };

    int offA[2], offB[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int ra = wr * 64 + m * 32 + c32;
        const int rb = wc * 64 + m * 32 + c32;
        offA[m] = ra * 128 + ((h ^ ((ra >> 1) & 7)) << 4);
        offB[m] = B_TILE_OFF + rb * 128 + ((h ^ ((rb >> 1) & 7)) << 4);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    float bnv[2];
    u32 colv[2];
    bool colok[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const long long col = b0 + wc * 64 + n * 32 + c32;
        colv[n] = (u32)col;
        colok[n] = col < p.nb_valid;
        bnv[n] = p.bn[col];
    }
    float my_qn = 0.f;
    u64 my_tau = ~0ull;
    if (tid < TQ) {
        my_qn = p.qn[q0 + tid];
        if (!DENSE) my_tau = p.tau[q0 + tid];
    }

    f32x4 fa[2][2], fb[2][2];                       // [register set][m or n]
    auto read_frags = [&](int set, int stage, int g) {
        const char* sb = smem + stage * STAGE_BYTES;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            fa[set][m] = *(const f32x4*)(sb + (offA[m] ^ (g << 5)));
            fb[set][m] = *(const f32x4*)(sb + (offB[m] ^ (g << 5)));
        }
    };
    // half a group: 8 MFMAs = k-pairs t0, t0+1 of the register set (ascending k is preserved: t = 0,1 then 2,3)
    auto mma8 = [&](int set, int t0) {
#pragma unroll
        for (int t = t0; t < t0 + 2; ++t)
#pragma unroll
            for (int m = 0; m < 2; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[set][m][t], fb[set][n][t], acc[m][n], 0, 0, 0);
    };
    // one group: first 8 MFMAs, then the ds_reads of the NEXT group into the other register set, then the last 8 MFMAs.
    // The reads have 8 MFMAs (>= 512 cycles) to land before their first use, so hipcc's lgkmcnt(0) there is free.
#define NWK_GROUP(SET, NEXT_STAGE, NEXT_G)            \
    do {                                              \
        __builtin_amdgcn_sched_barrier(0);            \
        mma8(SET, 0);                                 \
        __builtin_amdgcn_sched_barrier(0);            \
        read_frags((SET) ^ 1, NEXT_STAGE, NEXT_G);    \
        __builtin_amdgcn_sched_barrier(0);            \
        mma8(SET, 2);                                 \
        __builtin_amdgcn_sched_barrier(0);            \
    } while (0)

    issue_stage(0, 0);
    if (nk > 1) {
        issue_stage(1, 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    read_frags(0, 0, 0);

    for (int kt = 0; kt + 1 < nk; ++kt) {                 // every slab but the last ends in a slab boundary
        const int cur = kt & 1;
        NWK_GROUP(0, cur, 1);
        NWK_GROUP(1, cur, 2);
        NWK_GROUP(0, cur, 3);
        // slab boundary: every LDS read of slab kt by this wave is complete, slab kt+1 (issued a slab ago) has landed
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) issue_stage(cur, kt + 2);
        NWK_GROUP(1, cur ^ 1, 0);
    }
    {
        const int cur = (nk - 1) & 1;                     // last slab: nothing left to prefetch after its group 3
        NWK_GROUP(0, cur, 1);
        NWK_GROUP(1, cur, 2);
        NWK_GROUP(0, cur, 3);
        __builtin_amdgcn_sched_barrier(0);
        mma8(1, 0);
        mma8(1, 2);
    }
#undef NWK_GROUP

    // ---- epilogue (own LDS region: no wave can still be reading it)
    float* s_qn = (float*)(smem + EPI_OFF);
    u64* s_tau = (u64*)(smem + EPI_OFF + 512);
    if (tid < TQ) {
        s_qn[tid] = my_qn;
        if (!DENSE) s_tau[tid] = my_tau;
    }
    __syncthreads();
    tile_epilogue<METRIC, DENSE>(p, acc, s_qn, s_tau, bnv, colv, colok, q0, wr, h, c32);
}

// ------------------------------------------------------------------------------------------------
// ring form of the tile kernel: K slabs of BKT floats stream through an NST-deep LDS ring with ONE barrier per slab.
// Slab s+NST-1 is issued right after the barrier of iteration s (its buffer was last read in iteration s-1, and every
// wave has finished iteration s-1 before it passes that barrier), so no second barrier is needed; the counted vmcnt
// leaves NST-2 slabs in flight across the barrier.  Smaller slabs (BKT=16: 16 KiB per stage) allow 3 workgroups per CU.
// ------------------------------------------------------------------------------------------------
template <int N> __device__ __forceinline__ void wait_vmcnt()
{
    static_assert(N >= 0 && N <= 16, "vmcnt immediate");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else static_assert(N == 0, "add the immediate");
}

template <int METRIC, bool DENSE, int BKT, int NST, int WPS>
__global__ __launch_bounds__(256, WPS) void nwk_tile_kernel_ring(const NwkTileParams p)
{
    constexpr int ROWB = BKT * 4;                 // bytes per LDS row
    constexpr int CH = BKT / 4;                   // 16-B chunks per row
    constexpr int RPB = 256 / ROWB;               // rows per 256-B bank row
    constexpr int RSH = (RPB == 2) ? 1 : (RPB == 4) ? 2 : 3;
    constexpr int OPB = TQ * ROWB;                // bytes of one operand tile per stage
    constexpr int STG = 2 * OPB;
    constexpr int RPI = 1024 / ROWB;              // rows filled by one DMA wave-instruction
    constexpr int IPW = (TQ / RPI) / 4;           // DMA instructions per wave per operand per slab
    constexpr int LPS = 2 * IPW;                  // loads per slab per wave
    constexpr int NG = BKT / 8;                   // 8-k groups per slab
    static_assert(NST >= 3 && NST <= 4, "ring depth");
    __shared__ __attribute__((aligned(16))) char smem[NST * STG];

    const TileCoord tc = tile_from_block(p);
    if (!tc.ok) return;

    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63;
    const int wr = w >> 1, wc = w & 1;
    const int h = l >> 5, c32 = l & 31;
    const long long q0 = (long long)tc.qt * TQ;
    const long long b0 = (long long)tc.bt * TB;
    const int dp = p.dp;
    const int nk = dp / BKT;

    unsigned voff[4];                             // IPW <= 4 (fixed bound: a template-dependent bound here breaks host-side stub emission in hipcc 7.2)
#pragma unroll
    for (int i = 0; i < IPW; ++i) {
        const int row = (w * IPW + i) * RPI + (l / CH);
        const int chunk = (l % CH) ^ ((row >> RSH) & (CH - 1));
        voff[i] = (unsigned)(row * dp + chunk * 4) * 4u;
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.qp + q0 * dp), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bp + b0 * dp), 0, 0x7FFFFFFF, 0x00020000);
    const int wave_lds = __builtin_amdgcn_readfirstlane(w * IPW * 1024);

    auto issue_slab = [&](int kt) {
        char* base = smem + (kt % NST) * STG + wave_lds;
        const int soff = kt * (BKT * 4);
#pragma unroll
        for (int i = 0; i < IPW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)(base + i * 1024), 16, voff[i], soff, 0, 0);
#pragma unroll
        for (int i = 0; i < IPW; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)(base + OPB + i * 1024), 16, voff[i], soff, 0, 0);
    };

    int offA[2], offB[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int ra = wr * 64 + m * 32 + c32;
        const int rb = wc * 64 + m * 32 + c32;
        offA[m] = ra * ROWB + ((h ^ ((ra >> RSH) & (CH - 1))) << 4);
        offB[m] = OPB + rb * ROWB + ((h ^ ((rb >> RSH) & (CH - 1))) << 4);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    float bnv[2];
    u32 colv[2];
    bool colok[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const long long col = b0 + wc * 64 + n * 32 + c32;
        colv[n] = (u32)col;
        colok[n] = col < p.nb_valid;
        bnv[n] = p.bn[col];
    }
    float my_qn = 0.f;
    u64 my_tau = ~0ull;
    if (tid < TQ) {
        my_qn = p.qn[q0 + tid];
        if (!DENSE) my_tau = p.tau[q0 + tid];
    }

#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < nk) issue_slab(s);

    for (int kt = 0; kt < nk; ++kt) {
        const int ahead = nk - 1 - kt;                      // slabs after this one that exist
        if (ahead >= NST - 2) wait_vmcnt<(NST - 2) * LPS>();
        else if (NST == 4 && ahead == 1) wait_vmcnt<LPS>();
        else wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (kt + NST - 1 < nk) issue_slab(kt + NST - 1);
        const char* sb = smem + (kt % NST) * STG;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            f32x4 a[2], b[2];
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                a[m] = *(const f32x4*)(sb + (offA[m] ^ (g << 5)));
                b[m] = *(const f32x4*)(sb + (offB[m] ^ (g << 5)));
            }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][t], b[n][t], acc[m][n], 0, 0, 0);
        }
    }

    // ---- epilogue
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                       // every wave is done with the ring before it is reused
    float* s_qn = (float*)smem;
    u64* s_tau = (u64*)(smem + 1024);
    if (tid < TQ) {
        s_qn[tid] = my_qn;
        if (!DENSE) s_tau[tid] = my_tau;
    }
    __syncthreads();
    tile_epilogue<METRIC, DENSE>(p, acc, s_qn, s_tau, bnv, colv, colok, q0, wr, h, c32);
}

// ------------------------------------------------------------------------------------------------
// persistent form of the tile kernel (the default): 2 workgroups per CU stay resident and walk their XCD's
// super-tiles; the K-slab stream never drains between tiles -- while the last slab of tile t is multiplied, slab 0 of
// tile t+1 (and its norms / thresholds) is already in flight, and the epilogue of tile t runs under that DMA.
// Every load is an LDS-DMA, so all vmcnt bookkeeping is explicit (no compiler-inserted drains).
//   LDS: 2 x 32 KiB operand stages | 2 x 2 KiB tile metadata {qn[128] f32, bn[128] f32, tau[128] u64}
// ------------------------------------------------------------------------------------------------
#define META_BYTES 2048
#define META_OFF (2 * STAGE_BYTES)

struct PTile { long long q0, b0; int ok; };

__device__ __forceinline__ PTile ptile_next(const NwkTileParams& p, long long& super, long long n_super, int slot)
{
    // super-tile = SQ x SB tiles (SQ*SB = 64 = workgroups per XCD); this workgroup owns position `slot` of each
    PTile t; t.ok = 0; t.q0 = 0; t.b0 = 0;
    const int SB = 1 << p.sb_shift, SQ = 64 >> p.sb_shift;
    for (; super < n_super; super += 8) {
        const int sq = (int)(super / p.n_super_b), sb = (int)(super - (long long)sq * p.n_super_b);
        const int qi = sq * SQ + (slot >> p.sb_shift), bi = sb * SB + (slot & (SB - 1));
        if (qi < p.n_qtiles && bi < p.n_btiles) {
            const int qt = p.qtile_list ? p.qtile_list[qi] : qi;
            t.q0 = (long long)qt * TQ; t.b0 = (long long)(p.bt0 + bi) * TB; t.ok = 1;
            super += 8;
            return t;
        }
    }
    return t;
}

template <int METRIC, bool DENSE>
__global__ __launch_bounds__(256, 2) void nwk_tile_kernel_persistent(const NwkTileParams p)
{
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE_BYTES + 2 * META_BYTES];

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), l = tid & 63;
    const int wr = w >> 1, wc = w & 1;
    const int h = l >> 5, c32 = l & 31;
    const int dp = p.dp;
    const int nk = dp / BK;

    const int xcd = blockIdx.x & 7, slot = (blockIdx.x >> 3) & 63;
    long long super = xcd;                                      // this XCD group walks super-tiles xcd, xcd+8, ...
    const long long n_super = (long long)p.n_super_q * p.n_super_b;

    // per-lane DMA byte offsets inside a tile: wave w fills rows [32w, 32w+32), 8 rows per instruction
    unsigned voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (w * 4 + i) * 8 + (l >> 3);
        voff[i] = (unsigned)(row * dp + (((l & 7) ^ ((row >> 1) & 7)) << 2)) * 4u;
    }
    const int wave_lds = w * 4096;

    auto issue_slab = [&](int stage, const PTile& t, int kt) {
        char* base = smem + stage * STAGE_BYTES + wave_lds;
        const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.qp + t.q0 * dp), 0, 0x7FFFFFFF, 0x00020000);
        const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bp + t.b0 * dp), 0, 0x7FFFFFFF, 0x00020000);
        const int soff = kt * (BK * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)(base + i * 1024), 16, voff[i], soff, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)(base + B_TILE_OFF + i * 1024), 16, voff[i], soff, 0, 0);
    };
    // metadata: 2 DMA instructions per wave (256 B each): waves 0/1 -> qn halves, waves 2/3 -> bn halves; every wave -> a quarter of tau
    auto issue_meta = [&](int mb, const PTile& t) {
        char* base = smem + META_OFF + mb * META_BYTES;
        const float* src = (w < 2) ? (p.qn + t.q0 + w * 64) : (p.bn + t.b0 + (w - 2) * 64);
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(src + l), (LDS_AS void*)(base + w * 256), 4, 0, 0);
        const u32* tsrc = (const u32*)(p.tau + t.q0) + w * 64;
        __builtin_amdgcn_global_load_lds((const GLOBAL_AS void*)(tsrc + l), (LDS_AS void*)(base + 1024 + w * 256), 4, 0, 0);
    };

    int offA[2], offB[2];
#pragma unroll
    for (int m = 0; m < 2; ++m) {
        const int ra = wr * 64 + m * 32 + c32;
        const int rb = wc * 64 + m * 32 + c32;
        offA[m] = ra * 128 + ((h ^ ((ra >> 1) & 7)) << 4);
        offB[m] = B_TILE_OFF + rb * 128 + ((h ^ ((rb >> 1) & 7)) << 4);
    }

    PTile cur = ptile_next(p, super, n_super, slot);
    if (!cur.ok) return;
    int stage = 0, mb = 0;
    issue_meta(mb, cur);
    issue_slab(stage, cur, 0);

    for (;;) {
        const PTile nxt = ptile_next(p, super, n_super, slot);

        f32x16 acc[2][2];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

        for (int kt = 0; kt < nk; ++kt) {
            if (kt + 1 < nk) {
                issue_slab(stage ^ 1, cur, kt + 1);
                asm volatile("s_waitcnt vmcnt(8)" ::: "memory");      // everything older than the 8 just issued has landed
            } else if (nxt.ok) {
                issue_meta(mb ^ 1, nxt);
                issue_slab(stage ^ 1, nxt, 0);
                asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();
            const char* sb = smem + stage * STAGE_BYTES;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 a[2], b[2];
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    a[m] = *(const f32x4*)(sb + (offA[m] ^ (g << 5)));
                    b[m] = *(const f32x4*)(sb + (offB[m] ^ (g << 5)));
                }
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int m = 0; m < 2; ++m)
#pragma unroll
                        for (int n = 0; n < 2; ++n)
                            acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][t], b[n][t], acc[m][n], 0, 0, 0);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                              // stage may be refilled from here on
            stage ^= 1;
        }

        // ---- epilogue of `cur` (its metadata landed before the first slab's barrier)
        {
            const char* meta = smem + META_OFF + mb * META_BYTES;
            const float* s_qn = (const float*)meta;
            const float* s_bn = (const float*)(meta + 512);
            const u64* s_tau = (const u64*)(meta + 1024);
            float bnv[2];
            u32 colv[2];
            bool colok[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const int cl = wc * 64 + n * 32 + c32;
                const long long col = cur.b0 + cl;
                colv[n] = (u32)col;
                colok[n] = col < p.nb_valid;
                bnv[n] = s_bn[cl];
            }
            tile_epilogue<METRIC, DENSE>(p, acc, s_qn, s_tau, bnv, colv, colok, cur.q0, wr, h, c32);
        }
        if (!nxt.ok) break;
        // the other metadata buffer is refilled during the NEXT tile's last slab; with a single slab per tile (d <= 32)
        // that is its first instruction, so every wave must be out of this epilogue first
        __builtin_amdgcn_s_barrier();
        cur = nxt;
        mb ^= 1;
    }
}

// ------------------------------------------------------------------------------------------------
// compaction: sort one query's candidate list, (optionally) drop duplicate keys, keep k, tighten tau
//   dynamic LDS: [cap] u64 keys | [2][cap] u32 scan buffers | 16 B scratch   (one array: keeps the base 16-B aligned)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nwk_compact_kernel(const NwkCompactParams p)
{
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    u64* s = (u64*)dyn;
    u32* scan0 = (u32*)(dyn + (size_t)p.cap * 8);
    u32* scan1 = scan0 + p.cap;
    u32* misc = scan1 + p.cap;

    const int tid = threadIdx.x;
    long long q;
    if (p.qtile_list) q = (long long)p.qtile_list[blockIdx.x / TQ] * TQ + (blockIdx.x % TQ);
    else q = blockIdx.x;

    u32 n = p.dense_count > 0 ? (u32)p.dense_count : p.cnt[q];
    const bool overflow = n > (u32)p.cap;
    if (overflow) n = p.cap;
    // nothing appended since the last compaction: the list is already the sorted top-k
    if (!overflow && !p.dedupe && p.dense_count == 0 && n <= (u32)p.k && p.tau[q] != ~0ull) return;

    u32 np2 = 2;
    while (np2 < n) np2 <<= 1;
    u64* list = p.buf + q * p.cap;
    for (u32 i = tid; i < np2; i += 256) s[i] = (i < n) ? list[i] : ~0ull;
    if (tid == 0) misc[0] = 0;
    __syncthreads();
    for (u32 k2 = 2; k2 <= np2; k2 <<= 1) {
        for (u32 j = k2 >> 1; j > 0; j >>= 1) {
            for (u32 i = tid; i < np2; i += 256) {
                const u32 ixj = i ^ j;
                if (ixj > i) {
                    const u64 a = s[i], b = s[ixj];
                    const bool up = (i & k2) == 0;
                    if ((a > b) == up) { s[i] = b; s[ixj] = a; }
                }
            }
            __syncthreads();
        }
    }
    const u32 k = (u32)p.k;
    if (p.dedupe) {
        // a rescan can re-find entries that are already listed: keep the first occurrence of every key
        for (u32 i = tid; i < np2; i += 256) scan0[i] = (s[i] != ~0ull && (i == 0 || s[i] != s[i - 1])) ? 1u : 0u;
        __syncthreads();
        u32* src = scan0;
        u32* dst = scan1;
        for (u32 off = 1; off < np2; off <<= 1) {                  // inclusive Hillis-Steele scan, ping-pong
            for (u32 i = tid; i < np2; i += 256) dst[i] = src[i] + (i >= off ? src[i - off] : 0u);
            __syncthreads();
            u32* t = src; src = dst; dst = t;
        }
        const u32 nvalid = src[np2 - 1];
        for (u32 i = tid; i < np2; i += 256) {
            const bool keep = s[i] != ~0ull && (i == 0 || s[i] != s[i - 1]);
            const u32 at = src[i] - 1;
            if (keep && at < k) {
                list[at] = s[i];
                if (at == k - 1) p.tau[q] = s[i];
            }
        }
        if (tid == 0) {
            p.cnt[q] = nvalid < k ? nvalid : k;
            if (nvalid < k) p.tau[q] = ~0ull;
        }
    } else {
        // sentinels (padding columns of the dense first chunk) sort last: find where they start
        for (u32 i = tid; i < np2; i += 256)
            if (s[i] != ~0ull && (i + 1 == np2 || s[i + 1] == ~0ull)) misc[0] = i + 1;
        __syncthreads();
        const u32 nvalid = misc[0];
        const u32 keep = nvalid < k ? nvalid : k;
        for (u32 i = tid; i < keep; i += 256) list[i] = s[i];
        if (tid == 0) {
            p.cnt[q] = keep;
            p.tau[q] = (nvalid >= k) ? s[k - 1] : ~0ull;
        }
    }
    if (tid == 0 && overflow) { p.ovf[q] = 1u; atomicOr(p.ovf_any, 1u); }
}

// ------------------------------------------------------------------------------------------------
// finalize: keys -> (float32 distance, int64 index + offset)   (cu_knn.py:273 adds the tile offset the same way)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nwk_finalize_kernel(const u64* __restrict__ buf, const u32* __restrict__ cnt, int cap, int k,
                                                            long long nq, long long index_offset, float* __restrict__ out_d,
                                                            long long* __restrict__ out_i)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nq * k) return;
    const long long q = i / k;
    const int j = (int)(i - q * k);
    if ((u32)j < cnt[q]) {
        const u64 key = buf[q * cap + j];
        out_d[i] = unsortable_f32((u32)(key >> 32));
        out_i[i] = (long long)(u32)key + index_offset;
    } else {
        out_d[i] = __builtin_inff();
        out_i[i] = -1;
    }
}

// ------------------------------------------------------------------------------------------------
// merge of n_lists partial top-k lists [n_lists][Q][k_in] by (distance, index)   (merge.py:73-101)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nwk_merge_kernel(const float* __restrict__ dists, const long long* __restrict__ idx, int n_lists,
                                                         long long nq, int k_in, int k_out, float* __restrict__ out_d,
                                                         long long* __restrict__ out_i, int np2)
{
    extern __shared__ __attribute__((aligned(16))) char dyn[];
    u32* sd = (u32*)dyn;                            // sortable distance
    long long* si = (long long*)(dyn + (size_t)np2 * 4);      // np2 is a power of two >= 2: 8-B aligned
    const int tid = threadIdx.x;
    const long long q = blockIdx.x;
    const int w = n_lists * k_in;
    for (int i = tid; i < np2; i += 256) {
        if (i < w) {
            const int lst = i / k_in, j = i - lst * k_in;
            const long long at = ((long long)lst * nq + q) * k_in + j;
            const long long id = idx[at];
            sd[i] = id < 0 ? 0xFFFFFFFFu : sortable_u32(dists[at]);
            si[i] = id < 0 ? 0x7FFFFFFFFFFFFFFFll : id;
        } else { sd[i] = 0xFFFFFFFFu; si[i] = 0x7FFFFFFFFFFFFFFFll; }
    }
    __syncthreads();
    for (int k2 = 2; k2 <= np2; k2 <<= 1) {
        for (int j = k2 >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < np2; i += 256) {
                const int ixj = i ^ j;
                if (ixj > i) {
                    const u32 da = sd[i], db = sd[ixj];
                    const long long ia = si[i], ib = si[ixj];
                    const bool gt = (da > db) || (da == db && ia > ib);
                    const bool up = (i & k2) == 0;
                    if (gt == up) { sd[i] = db; sd[ixj] = da; si[i] = ib; si[ixj] = ia; }
                }
            }
            __syncthreads();
        }
    }
    for (int j = tid; j < k_out; j += 256) {
        const bool ok = j < np2 && si[j] != 0x7FFFFFFFFFFFFFFFll;
        out_d[q * k_out + j] = ok ? unsortable_f32(sd[j]) : __builtin_inff();
        out_i[q * k_out + j] = ok ? si[j] : -1;
    }
}

// ------------------------------------------------------------------------------------------------
// launchers (host side, called from nwknn_api.cpp)
// ------------------------------------------------------------------------------------------------
extern "C++" {

hipError_t nwk_launch_gen(hipStream_t st, uint64_t seed, long long row0, long long rows, int d, float* out)
{
    const long long total = rows * (long long)d;
    if (total <= 0) return hipSuccess;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(nwk_gen_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (u64)seed, row0, rows, d, out);
    return hipGetLastError();
}

hipError_t nwk_launch_prep(hipStream_t st, const float* raw, long long rows, int d, long long rows_pad, int dp, float* prepared,
                           float* norms)
{
    if (rows_pad <= 0) return hipSuccess;
    const long long blocks = (rows_pad + 255) / 256;
    hipLaunchKernelGGL(nwk_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, st, raw, rows, d, rows_pad, dp, prepared, norms);
    return hipGetLastError();
}

template <int METRIC>
static hipError_t launch_tile_metric(hipStream_t st, const NwkTileParams& p, bool dense, unsigned grid, int variant)
{
#define NWK_LAUNCH(K) do { if (dense) hipLaunchKernelGGL((K<METRIC, true>), dim3(grid), dim3(256), 0, st, p); \
                           else hipLaunchKernelGGL((K<METRIC, false>), dim3(grid), dim3(256), 0, st, p); } while (0)
#define NWK_LAUNCH_RING(BKT, NST, WPS) do { if (dense) hipLaunchKernelGGL((nwk_tile_kernel_ring<METRIC, true, BKT, NST, WPS>), dim3(grid), dim3(256), 0, st, p); \
                           else hipLaunchKernelGGL((nwk_tile_kernel_ring<METRIC, false, BKT, NST, WPS>), dim3(grid), dim3(256), 0, st, p); } while (0)
    switch (variant) {
    case 1: NWK_LAUNCH(nwk_tile_kernel_persistent); break;
    case 2: NWK_LAUNCH_RING(16, 3, 3); break;
    case 3: NWK_LAUNCH_RING(16, 4, 2); break;
    case 4: NWK_LAUNCH_RING(32, 3, 1); break;
    case 5: NWK_LAUNCH(nwk_tile_kernel_pipe); break;
    case 11: hipLaunchKernelGGL((nwk_tile_kernel<METRIC, false, 1>), dim3(grid), dim3(256), 0, st, p); break;
    case 12: hipLaunchKernelGGL((nwk_tile_kernel<METRIC, false, 2>), dim3(grid), dim3(256), 0, st, p); break;
    case 14: hipLaunchKernelGGL((nwk_tile_kernel<METRIC, false, 4>), dim3(grid), dim3(256), 0, st, p); break;
    case 17: hipLaunchKernelGGL((nwk_tile_kernel<METRIC, false, 7>), dim3(grid), dim3(256), 0, st, p); break;
    case 15: hipLaunchKernelGGL((nwk_tile_kernel<METRIC, false, 5>), dim3(grid), dim3(256), 0, st, p); break;
    case 18: hipLaunchKernelGGL((nwk_tile_kernel<METRIC, false, 8>), dim3(grid), dim3(256), 0, st, p); break;
    case 19: hipLaunchKernelGGL((nwk_tile_kernel<METRIC, false, 16>), dim3(grid), dim3(256), 0, st, p); break;
    default: NWK_LAUNCH(nwk_tile_kernel); break;
    }
#undef NWK_LAUNCH
#undef NWK_LAUNCH_RING
    return hipGetLastError();
}

static int pow2_ceil_shift(int x) { int s = 0; while ((1 << s) < x) ++s; return s; }

hipError_t nwk_launch_tile(hipStream_t st, NwkTileParams p, int metric, bool dense, int variant)
{
    if (p.n_qtiles <= 0 || p.n_btiles <= 0) return hipSuccess;
    if (variant >= 2 && (p.dp % 32) != 0) variant = 0;
    if (variant >= 10 && dense) variant = 0;
    long long grid;
    if (variant == 1) {
        // super-tile shape SQ x SB with SQ*SB = 64: 8x8 when both sides are large, stretched when one side is short
        int sb_shift = 3;
        if (p.n_btiles < 8) sb_shift = pow2_ceil_shift(p.n_btiles);
        if (p.n_qtiles < (64 >> sb_shift)) sb_shift = 6 - pow2_ceil_shift(p.n_qtiles);
        p.sb_shift = sb_shift;
        const int SB = 1 << sb_shift, SQ = 64 >> sb_shift;
        p.n_super_q = (p.n_qtiles + SQ - 1) / SQ;
        p.n_super_b = (p.n_btiles + SB - 1) / SB;
        const long long supers = (long long)p.n_super_q * p.n_super_b;
        (void)supers;
        grid = 512;                                                 // 8 XCD groups x 64 slots: every (group, slot) must exist
    } else {
        const long long n_super_q = (p.n_qtiles + 7) / 8;
        p.n_super_b = (p.n_btiles + 7) / 8;
        long long supers = n_super_q * p.n_super_b;
        supers = (supers + 7) / 8 * 8;
        grid = supers * 64;
    }
    if (grid > 0x7FFFFFFFll) return hipErrorInvalidValue;
    switch (metric) {
    case NWKNN_METRIC_SQEUCLIDEAN: return launch_tile_metric<NWKNN_METRIC_SQEUCLIDEAN>(st, p, dense, (unsigned)grid, variant);
    case NWKNN_METRIC_ONE_MINUS_DOT: return launch_tile_metric<NWKNN_METRIC_ONE_MINUS_DOT>(st, p, dense, (unsigned)grid, variant);
    case NWKNN_METRIC_COSINE: return launch_tile_metric<NWKNN_METRIC_COSINE>(st, p, dense, (unsigned)grid, variant);
    default: return hipErrorInvalidValue;
    }
}

hipError_t nwk_launch_compact(hipStream_t st, const NwkCompactParams& p, long long n_queries)
{
    if (n_queries <= 0) return hipSuccess;
    const size_t lds = (size_t)p.cap * 16 + 16;
    hipLaunchKernelGGL(nwk_compact_kernel, dim3((unsigned)n_queries), dim3(256), lds, st, p);
    return hipGetLastError();
}

hipError_t nwk_launch_finalize(hipStream_t st, const u64* buf, const u32* cnt, int cap, int k, long long nq, long long index_offset,
                               float* out_d, long long* out_i)
{
    const long long total = nq * k;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(nwk_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, buf, cnt, cap, k, nq, index_offset,
                       out_d, out_i);
    return hipGetLastError();
}

hipError_t nwk_launch_merge(hipStream_t st, const float* dists, const long long* idx, int n_lists, long long nq, int k_in, int k_out,
                            float* out_d, long long* out_i)
{
    if (nq <= 0) return hipSuccess;
    int np2 = 2;
    while (np2 < n_lists * k_in) np2 <<= 1;
    const size_t lds = (size_t)np2 * 12;
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(nwk_merge_kernel, dim3((unsigned)nq), dim3(256), lds, st, dists, idx, n_lists, nq, k_in, k_out, out_d, out_i, np2);
    return hipGetLastError();
}

}  // extern "C++"
