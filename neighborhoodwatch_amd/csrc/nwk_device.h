// nwk_device.h -- gfx950 (MI355X, CDNA4) kernels of the brute-force kNN ground-truth hot path: shared device code.
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
//   * distance tile = f32 MFMA GEMM (v_mfma_f32_32x32x2_f32): 256 (or 128) queries x 128 base rows per workgroup, three (two)
//     workgroups per CU, K streamed in 16- (32-)float slabs through double-buffered LDS filled by LDS-DMA
//     (buffer_load_dwordx4 ... lds).  One accumulator carries the whole K loop in ascending k, so every dot product is
//     bit-for-bit the k-ordered fmaf chain of oracle/nwknn_oracle.c.
//   * the Q x B distance matrix never exists: the epilogue turns each accumulator into a 64-bit key
//     (sortable(distance) << 32 | base row) and keeps only keys below the query's current k-th key (tau);
//     survivors are appended to a per-query candidate list (fp32 kernel: half-wave aggregated atomics; fp16 kernels: staged in an
//     LDS list and moved out once per strip).
//   * a compaction kernel sorts each query's list (bitonic, LDS), keeps k, and tightens tau; the host walks the
//     base in geometrically growing chunks so that tau is already tight when the big chunks run.
//   * all integer/key work is exact: 64-bit key order == (distance, index) order, lower index wins ties.
//   * option "prescreen": the filtered chunks run on the fp16 MFMA kernel (v_mfma_f32_16x16x32_f16) over scaled fp16 copies under a
//     proven additive error bound, and nwk_rescore_kernel replaces every listed pair by its canonical fp32 key -- the same bits
//     out, several times faster; fp16 INPUT data takes the same route by default ("exact fp16 mode"), MaxSim has its own
//     pre-screen / prune / rescore kernels.
//
// This header holds what every kernel file shares: vector types, the order-preserving key map, the canonical epilogue, the
// XCD-aware block -> tile mapping, and the filter epilogue of the 32x32-accumulator kernels.  The kernels live in
//   nwk_prep.hip       seeded generator, layout/norm preparation, max|x| and the scaled fp16 copies of the pre-screen
//   nwk_tile_f32.hip   fp32 distance tile (v_mfma_f32_32x32x2_f32) fused with the running top-k filter
//   nwk_tile_f16x.h    fp16 distance tile, ping-pong form (v_mfma_f32_16x16x32_f16; instantiated by nwk_tile_f16x_pre.hip / _raw.hip): the
//                      default from 768 halves per row on
//   nwk_tile_f16p.hip  fp16 distance tile, streaming form: the default below
//   nwk_tile_f16.hip   fp16 distance tile, burst form (the build with a dense mode; rows beyond 2048 halves), direct and pre-screen (PRE = 1)
//   nwk_select.hip     rescoring of pre-screened pairs, list compaction, finalize, partial-list merge
//   nwk_maxsim.hip     MaxSim late interaction (extension)
// each with its host-side launchers (declared in nwknn_internal.h, called from nwknn_api.cpp).
#pragma once
#ifndef NWK_MFMA_SNAKE
#define NWK_MFMA_SNAKE 1      // issue order of a k-step's MFMAs (nwk_tile_f16y.h); 0 = n ascending under every m, for A/B builds
#endif
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "nwknn_internal.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned long long u64;
typedef unsigned int u32;

#define GLOBAL_AS __attribute__((address_space(1)))
#define LDS_AS __attribute__((address_space(3)))

// In-kernel cycle stamps go to NwkTileParams::dbg when it is set.  Only a developer build of the library (-DNWKNN_DEV, tools/mk_ab_lib.sh)
// has the option that sets it ("dbg_ptr") and the DIAG timing ablations, whose results are wrong by construction; the shipped library
// compiles none of the stamp code (the pointer is a constant null: every `if (NWK_DBG(p))` folds away).  tests/test_isa_budget.py holds
// the register budget of the tile kernels against the compiler's metadata, so a toolchain change that moves it is seen on the CPU.
#if defined(NWKNN_DEV) || defined(NWK_KEEP_DBG)      // NWK_KEEP_DBG: build-time A/B of the register allocation with the never-taken stamp branches left in (tools/mk_f32_variants.sh)
#define NWK_DBG(p) ((p).dbg)
#else
#define NWK_DBG(p) ((unsigned long long*)nullptr)
#endif

// ------------------------------------------------------------------------------------------------
// small device helpers
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32 sortable_u32(float f)
{
    u32 u = (f != f) ? 0x7FC00000u : __float_as_uint(f);      // every NaN ranks as +qNaN (last): sign/payload differ between CPUs and GPUs
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

// The same value from STAGED operands: for cosine the tile kernels keep the correctly rounded roots -- sqrtf(qn) per query row in LDS
// (stage_query_side), sqrtf(bn) per column in a register, each computed once per tile -- so that a block's exact pass is one multiply, one
// correctly rounded division and one subtraction per element instead of two more root sequences: the same operations on the same values
// as the canonical expression above, hence the same bits.  (Not for the pre-screen kernels, PRE = 1: their keys are estimates.)
template <int METRIC>
__device__ __forceinline__ float distance_from_staged(float dot, float qs, float bs)
{
    if (METRIC == NWKNN_METRIC_COSINE) return __fsub_rn(1.0f, __fdiv_rn(dot, __fmul_rn(qs, bs)));
    return distance_epilogue<METRIC>(dot, qs, bs);
}
template <int METRIC, int PRE>
__device__ __forceinline__ float staged_norm(float n)
{
    return (METRIC == NWKNN_METRIC_COSINE && !PRE) ? sqrtf(n) : n;
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
// the distance-tile + filter kernel
// ------------------------------------------------------------------------------------------------
// One workgroup (4 waves) computes a TM x 128 tile of the distance matrix over the whole d:
//   * waves sit 2 x 2; a wave owns (TM/2) x 64 outputs = MT x 2 accumulators of v_mfma_f32_32x32x2_f32 (MT = TM/64);
//   * K streams in BKT-wide slabs through two LDS stages filled by LDS-DMA (buffer_load_dwordx4 ... offen lds: one
//     wave-uniform descriptor per operand tile, the slab offset in an SGPR, a 32-bit per-lane offset -- measured 3 %
//     faster than global_load_lds with 64-bit per-lane addresses, the DMA instruction stream is what the MFMAs compete with);
//   * counted s_waitcnt vmcnt(LPS) + raw s_barrier keep the next slab's DMA in flight across the barrier.
// Two shapes are built:  <128, 32> (64 KiB LDS, 145 VGPRs)  and  <256, 16> (48 KiB LDS): same barriers per MFMA, but
// 25 % fewer DMA instructions and fragment reads per MFMA and half as many tile prologues/epilogues/dispatches.
//
// LDS image of an operand tile: rows of ROWB = 4*BKT bytes, CH = BKT/4 chunks of 16 B; logical chunk c of row r is stored
// in slot c ^ ((r >> RSH) & (CH-1)) with 2^RSH rows per 256-B bank row.  ds_read_b128 by 32 consecutive rows is then
// conflict-free.  The DMA writes LDS linearly (base + lane*16), so the swizzle is applied to the per-lane SOURCE
// offset (guide rule 21) and again on the read address.

#define TB 128                     // base rows per tile

struct TileCoord { int qt, bt, ok; };

__device__ __forceinline__ TileCoord tile_from_block(const NwkTileParams& p)
{
    // XCD-aware mapping: blocks b and b+8 share an XCD (and its L2; checked: tools/xcd_map.hip); each XCD works through super-tiles
    // of 64 tiles (SQ query tiles x SB base tiles, SQ*SB = 64, SQ = 1 << p.sq_shift) so that its co-resident workgroups share operands.
    // In steady state the 64 (fp16: 32) resident blocks of an XCD are consecutive block numbers whose start times -- hence K
    // positions -- are spread evenly over one tile time, and the L2 holds only the last 2-3 "slab times" of the XCD's operand
    // stream: a slab fetched by one block is still there for the block started right behind it, not for one started eight later.
    // So what is shared through L2 is decided by the walk order: sharers must be at most ~2 block numbers apart.  With the base
    // tile index fastest (8 x 8) the query tile is chained through 8 consecutive blocks and the base tile is not shared at all;
    // with the query index fastest and 2 x 32, the base tile is shared by a pair (distance 1) and the query tile by 32 blocks
    // (distance 2): 38 % less fabric traffic on the fp32 kernel (tools/traffic_sweep.sh) -- but 0.27 % slower, and the traffic is
    // served by the Infinity Cache either way, so 8 x 8 / base fastest stays the default (p.walk_q_fastest selects the other walk).
    const int bid = blockIdx.x;
    const int xcd = bid & 7, local = bid >> 3;
    const long long super = (long long)(local >> 6) * 8 + xcd;
    const int within = local & 63;
    const int sb_shift = 6 - p.sq_shift;
    const int sq = (int)(super / p.n_super_b), sb = (int)(super % p.n_super_b);
    int qi = (sq << p.sq_shift) + (within >> sb_shift), bi = (sb << sb_shift) + (within & ((1 << sb_shift) - 1));
    if (p.walk_q_fastest) {
        qi = (sq << p.sq_shift) + (within & ((1 << p.sq_shift) - 1));
        bi = (sb << sb_shift) + (within >> p.sq_shift);
    }
    TileCoord t;
    t.ok = (qi < p.n_qtiles) && (bi < p.n_btiles);
    t.qt = t.ok ? (p.qtile_list ? p.qtile_list[qi] : qi) : 0;
    t.bt = p.bt0 + bi;
    return t;
}

// ------------------------------------------------------------------------------------------------
// epilogue: MT x 2 accumulators of one wave -> distances -> 64-bit keys -> filter against tau -> candidate lists.
// s_qn / s_tau hold the tile's query norms / thresholds in LDS.  For one 32-row block the 16 rows a lane owns are
// rows rb+8j .. rb+8j+3 (j = 0..3, rb = block + 4*half), so norms and thresholds come in with 12 ds_read_b128 issued
// back to back and the 32 elements that follow are pure VALU (the first version interleaved two LDS round trips and a
// branch per element: ~60k cycles per tile; this form: a few thousand).
// ------------------------------------------------------------------------------------------------
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Pre-filter.  key < tau requires dist <= tau_d (tau_d = distance part of tau).  For the two additive metrics that is a
// lower bound on the raw dot product that splits into a row part and a column part:
//     sqeuclidean   : qn + bn - 2 dot <= tau_d   <=>  dot >= (qn - tau_d)/2 + bn/2
//     one_minus_dot : 1 - dot <= tau_d           <=>  dot >= 1 - tau_d
// so the common case costs one add and one compare per accumulator.  The row part is lowered by a margin that covers
// every rounding of the canonical fp32 evaluation (a few ulps of the operands), which makes the test a strict superset of
// the exact one; whatever passes it is re-evaluated exactly.
// Cosine splits MULTIPLICATIVELY:  1 - dot / (sqrt(qn) sqrt(bn)) <= tau_d   <=>  dot >= (1 - tau_d) sqrt(qn) * sqrt(bn),  a row term times a
// column term -- one multiply and one compare per accumulator.  (Rounds 1-4 pre-tested on the distance itself: two correctly rounded square
// roots and a division per accumulator, 38 spilled registers at three workgroups per CU.)  The canonical evaluation rounds five times
// (two roots, their product, the quotient, the subtraction: < 6e-7 relative on T = 1 - tau_d and on tau_d); the row term is lowered by
// 2e-6 (|T| + |tau_d| + 1) and then by 4e-6 of its magnitude, which also covers the bare v_sqrt_f32 (1 ulp) used for the two terms and the
// product in the test.  A row or column whose squared norm is outside [2^-80, 2^80] (zero rows, Inf, NaN) gets a term that makes every
// comparison come out "maybe" (-inf, or NaN: the test is !(acc < thr)), so the canonical arithmetic decides those pairs, NaN and all.
__device__ __forceinline__ bool cosine_norm_ok(float n);
template <int METRIC>
__device__ __forceinline__ float prefilter_row(float qn, u64 tau)
{
    if (tau == ~0ull) return -__builtin_inff();                      // fewer than k listed: everything may pass
    const float td = unsortable_f32((u32)(tau >> 32));
    if (METRIC == NWKNN_METRIC_SQEUCLIDEAN) return 0.5f * (qn - td) - 1.0e-6f * (fabsf(qn) + fabsf(td)) - 1.0e-30f;
    if (METRIC == NWKNN_METRIC_ONE_MINUS_DOT) return (1.0f - td) - 1.0e-6f * (1.0f + fabsf(td));
    if (!cosine_norm_ok(qn)) return -__builtin_inff();
    const float t = 1.0f - td;
    const float tm = t - 2.0e-6f * (fabsf(t) + fabsf(td) + 1.0f);
    const float sq = __builtin_amdgcn_sqrtf(qn);
    return tm * sq - 4.0e-6f * fabsf(tm) * sq;                        // times prefilter_col = sqrt(bn): a lower bound of T sqrt(qn) sqrt(bn) whatever its sign
}
template <int METRIC>
__device__ __forceinline__ float prefilter_col(float bn)
{
    if (METRIC == NWKNN_METRIC_SQEUCLIDEAN) return 0.5f * bn - 1.0e-6f * fabsf(bn);
    if (METRIC == NWKNN_METRIC_COSINE) return cosine_norm_ok(bn) ? __builtin_amdgcn_sqrtf(bn) : __builtin_nanf("");
    return 0.0f;
}

// Pre-screen thresholds (fp16 pre-screen of the exact fp32 search).  The accumulator holds  S * dot16  with S = pre_scale and
// |dot16 - dot32| <= E(q,b) = eps/2 (qn + bn) + aq sqrt(bn) + ab sqrt(qn) + cst   (DESIGN.md "pre-screen bound"), so every pair
// whose CANONICAL fp32 distance is <= tau_d satisfies  S*dot16 >= S * (row(q) + col(b))  with
//     sqeuclidean   : row = (qn - tau_d)/2 - eps/2 qn - ab sqrt(qn) - cst,   col = bn/2 - eps/2 bn - aq sqrt(bn)
//     one_minus_dot : row = (1 - tau_d)    - eps/2 qn - ab sqrt(qn) - cst,   col =        - eps/2 bn - aq sqrt(bn)
// each lowered by a few ulps of its operands for the fp32 evaluation of these very expressions (the square roots are bare
// v_sqrt_f32, 1 ulp: the a_q / a_b constants carry 1 % of slack).
// Cosine: the copies are NORMALISED rows (x / |x| * 2^14), the accumulator is S * cos(q,b) up to the bound in pre_cst (eps_half, aq, ab
// are zero), and the test is the one_minus_dot one.  A row whose squared norm is outside [2^-80, 2^80] (zero rows included) has an
// all-zero copy and a threshold of -inf: everything involving it is listed and the canonical arithmetic decides.
__device__ __forceinline__ bool cosine_norm_ok(float n) { return n >= 0x1p-80f && n <= 0x1p80f; }

template <int METRIC>
__device__ __forceinline__ float prescreen_row(float qn, u64 tau, const NwkTileParams& p)
{
    if (tau == ~0ull) return -__builtin_inff();
    if (METRIC == NWKNN_METRIC_COSINE && !cosine_norm_ok(qn)) return -__builtin_inff();
    const float td = unsortable_f32((u32)(tau >> 32));
    const float err = p.pre_eps_half * qn + p.pre_ab * __builtin_amdgcn_sqrtf(qn) + p.pre_cst;
    float r;
    if (METRIC == NWKNN_METRIC_SQEUCLIDEAN) r = 0.5f * (qn - td) - err - 2.0e-6f * (fabsf(qn) + fabsf(td)) - 1.0e-30f;
    else r = (1.0f - td) - err - 2.0e-6f * (1.0f + fabsf(td) + err);
    return r * p.pre_scale;
}
template <int METRIC>
__device__ __forceinline__ float prescreen_col(float bn, const NwkTileParams& p)
{
    if (METRIC == NWKNN_METRIC_COSINE) return cosine_norm_ok(bn) ? 0.0f : -__builtin_inff();
    const float err = p.pre_eps_half * bn + p.pre_aq * __builtin_amdgcn_sqrtf(bn);
    float c;
    if (METRIC == NWKNN_METRIC_SQEUCLIDEAN) c = 0.5f * bn - err - 2.0e-6f * fabsf(bn);
    else c = -err - 2.0e-6f * err;
    return c * p.pre_scale;
}

// one register of survivors (a half-wave's lanes hold 32 columns of ONE query row): one list-length atomic per half-wave, positions by rank
__device__ __forceinline__ void append_half_wave(const NwkTileParams& p, bool pass, u64 key, long long qrow, int h, int c32)
{
    const u64 mk = __ballot(pass);
    if (!mk) return;                                                 // wave-uniform
    const u32 mh = h ? (u32)(mk >> 32) : (u32)mk;
    const int leader = mh ? (__ffs(mh) - 1) : 0;
    u32 base = 0;
    if (mh && c32 == leader) base = atomicAdd(p.cnt + qrow, (u32)__popc(mh));
    base = __shfl(base, leader + (h << 5));
    if (pass) {
        const u32 pos = base + __popc(mh & ((1u << c32) - 1u));
        if (pos < (u32)p.cap) p.buf[qrow * p.cap + pos] = key;
    }
}

template <int METRIC, bool DENSE, int MT, int PRE = 0>
__device__ __forceinline__ void tile_epilogue(const NwkTileParams& p, const f32x16 (&acc)[MT][2], const float* s_qn, const u64* s_tau,
                                              const float* s_pre, const float (&bnv)[2], const u32 (&colv)[2], const bool (&colok)[2],
                                              long long q0, int wrow0, int h, int c32)
{
    const int cap = p.cap;
    float cpre[2], bs[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        cpre[n] = colok[n] ? (PRE ? prescreen_col<METRIC>(bnv[n], p) : prefilter_col<METRIC>(bnv[n])) : __builtin_inff();   // padding columns never pass
        bs[n] = staged_norm<METRIC, PRE>(bnv[n]);                    // cosine: sqrtf(bn), once per tile (s_qn holds sqrtf(qn): stage_query_side)
    }
    // PRE: acc = S*dot16 is tested against the scaled row + column terms of the pre-screen bound (carrying the column term in
    // the accumulators from the start saves one add per element but keeps 128 registers live through the prologue: it spilled)
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int rb = wrow0 + m * 32 + 4 * h;
        if (DENSE) {
            f32x4 qn4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) qn4[j] = *(const f32x4*)(s_qn + rb + 8 * j);
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float dist = distance_from_staged<METRIC>(acc[m][n][r], qn4[r >> 2][r & 3], bs[n]);
                    const int row = rb + (r & 3) + 8 * (r >> 2);
                    const u32 sd = (METRIC == NWKNN_METRIC_SQEUCLIDEAN) ? (__float_as_uint(dist) | 0x80000000u) : sortable_u32(dist);
                    const long long slot = (long long)(colv[n] - (u32)p.dense_col0);
                    p.buf[(q0 + row) * cap + slot] = colok[n] ? (((u64)sd << 32) | colv[n]) : ~0ull;
                }
            continue;
        }
        f32x4 pre4[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) pre4[j] = *(const f32x4*)(s_pre + rb + 8 * j);
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            bool maybe = false;
            if (METRIC == NWKNN_METRIC_COSINE && !PRE) {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    maybe |= !(acc[m][n][r] < pre4[r >> 2][r & 3] * cpre[n]);   // NaN (a degenerate row or column) falls through
                maybe &= colok[n];                                   // (+inf times a negative row term would pass a padding column)
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    maybe |= !(acc[m][n][r] < pre4[r >> 2][r & 3] + cpre[n]);   // NaN falls through
            }
            if (!__ballot(maybe)) continue;                        // the common case: nothing in this 32x32 block can survive
            // exact pass over the block, four rows (one j) at a time to keep the register footprint small: norms and
            // thresholds by ds_read_b128, straight-line VALU (distance -> key -> 64-bit compare); only rows with a survivor
            // reach the atomics
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if constexpr (METRIC == NWKNN_METRIC_COSINE && !PRE) {
                    // cosine: two rows at a time -- the correctly rounded division's temporaries come on top of the 128 accumulators, and four
                    // keys + four thresholds beside them were what the three-per-CU build spilled (10 -> see tests/test_isa_budget.py)
                    const float bsn = sqrtf(bnv[n]);                 // a flagged block is rare: the root is taken here, not held through the tile
#pragma unroll
                    for (int i0 = 0; i0 < 4; i0 += 2) {
                        const f32x2 qn2 = *(const f32x2*)(s_qn + rb + 8 * j + i0);
                        const u64x2 t2 = *(const u64x2*)(s_tau + rb + 8 * j + i0);
                        u64 key[2];
                        bool pass[2];
#pragma unroll
                        for (int i = 0; i < 2; ++i) {
                            const float dist = distance_from_staged<METRIC>(acc[m][n][4 * j + i0 + i], qn2[i], bsn);
                            key[i] = ((u64)sortable_u32(dist) << 32) | colv[n];
                            pass[i] = (key[i] < t2[i]) & colok[n];
                        }
#pragma unroll
                        for (int i = 0; i < 2; ++i) append_half_wave(p, pass[i], key[i], q0 + rb + i0 + i + 8 * j, h, c32);
                    }
                    continue;
                }
                const f32x4 qn4 = *(const f32x4*)(s_qn + rb + 8 * j);
                const u64x2 tl = *(const u64x2*)(s_tau + rb + 8 * j);
                const u64x2 th = *(const u64x2*)(s_tau + rb + 8 * j + 2);
                u64 key[4];
                bool pass[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (PRE) {     // pre-screen: the additive test IS the filter; the key carries the fp16 estimate of the distance,
                                   // from which the rescore kernel derives bounds that let it skip most of the listed pairs
                        const float sdot = acc[m][n][4 * j + i];                               // S * dot16
                        const float est = (METRIC == NWKNN_METRIC_SQEUCLIDEAN) ? __fmaf_rn(-2.0f * p.pre_inv_scale, sdot, qn4[i] + bnv[n])
                                                                               : 1.0f - sdot * p.pre_inv_scale;
                        key[i] = ((u64)__float_as_uint(est) << 32) | colv[n];
                        pass[i] = !(acc[m][n][4 * j + i] < pre4[j][i] + cpre[n]) & colok[n];
                        continue;
                    }
                    const float dist = distance_from_staged<METRIC>(acc[m][n][4 * j + i], qn4[i], bs[n]);
                    const u32 sd = (METRIC == NWKNN_METRIC_SQEUCLIDEAN) ? (__float_as_uint(dist) | 0x80000000u)   // dist >= +0, never NaN
                                                                         : sortable_u32(dist);
                    key[i] = ((u64)sd << 32) | colv[n];
                    const u64 tau = (i & 2) ? th[i & 1] : tl[i & 1];
                    pass[i] = (key[i] < tau) & colok[n];           // exact (distance, index) order
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const u64 mk = __ballot(pass[i]);
                    if (mk) {                                    // wave-uniform
                        const int row = rb + i + 8 * j;
                        const u32 mh = h ? (u32)(mk >> 32) : (u32)mk; // this half-wave's survivors: all for ONE query row
                        const int leader = mh ? (__ffs(mh) - 1) : 0;
                        u32 base = 0;
                        if (mh && c32 == leader) base = atomicAdd(p.cnt + q0 + row, (u32)__popc(mh));
                        base = __shfl(base, leader + (h << 5));
                        if (pass[i]) {
                            const u32 pos = base + __popc(mh & ((1u << c32) - 1u));
                            if (pos < (u32)cap) p.buf[(q0 + row) * cap + pos] = key[i];
                        }
                    }
                }
            }
        }
    }
}

// staging of the tile's query-side epilogue inputs in LDS: norms | 64-bit thresholds | pre-filter row terms
template <int METRIC, bool DENSE, int PRE = 0>
__device__ __forceinline__ void stage_query_side(char* lds, int rows, int tid, float my_qn, u64 my_tau, const NwkTileParams& p, long long q0)
{
    float* s_qn = (float*)lds;
    u64* s_tau = (u64*)(lds + rows * 4);
    float* s_pre = (float*)(lds + rows * 12);
    if (tid < rows) {
        s_qn[tid] = staged_norm<METRIC, PRE>(my_qn);                  // cosine (direct filter): the correctly rounded root, see distance_from_staged
        if (!DENSE) {
            s_tau[tid] = my_tau;
            float pre = PRE ? prescreen_row<METRIC>(my_qn, my_tau, p) : prefilter_row<METRIC>(my_qn, my_tau);
            // padding query rows (all zero) must list nothing: with one_minus_dot / cosine they tie with every base row, which
            // the pre-screen cannot break by index and would list wholesale.  (The direct cosine test is on the distance: -inf.)
            if (p.nq_valid > 0 && q0 + tid >= p.nq_valid)
                pre = __builtin_inff();
            s_pre[tid] = pre;
        }
    }
}

// DIAG (timing experiments only, results are wrong when != 0): 1 = no LDS fragment reads in the loop, 2 = no DMA after
// slab 0, 4 = no barriers, 8 = no B-operand DMA after slab 0.  Instantiated under -DNWKNN_DEV only; the product launches DIAG = 0.
