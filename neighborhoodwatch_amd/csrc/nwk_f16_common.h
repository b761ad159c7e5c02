// nwk_f16_common.h -- shared by the fp16 tile kernels (nwk_tile_f16.hip, nwk_tile_f16p.hip): tile geometry, the filter epilogue of
// the 16x16x32 accumulator layout and the LDS staging of survivors.
#pragma once

#include "nwk_device.h"

#define H_TM 256
#define H_TN 256
#define H_BK 64                       // halves per slab = 128-B rows
#define H_STAGE (2 * 256 * 128)       // A 32 KiB + B 32 KiB

// Accumulator layout of v_mfma_f32_16x16x32_f16: lane l holds D[4*(l/16) + i][l%16], i = 0..3  ->  column (base row) = l%16,
// four consecutive query rows; a wave's 128 x 64 tile is acc[8][4] (16-row blocks x 16-column blocks).
// Survivor staging (SC > 0).  A survivor used to cost its quarter-wave a returning global atomic on the query's list length -- a
// 1-3k cycle round trip, one after the other for every (column block, row) of the wave's tile that had one, with the other
// seven waves waiting at the next barrier.  At 10M-row scale a 256 x 256 tile holds one or two survivors (3.5 % of the tile
// kernel); on a 1M-row shard (C2, or one of eight GPUs on C3) it holds a dozen and the round trips were 10-20 %.  Now a survivor
// goes to a small list in LDS ({key, query row of the tile}; one ds_add_rtn per ballot) and the workgroup moves the list out
// once per strip, all lanes at once, one round trip in total (flush_survivors).  An entry that finds the LDS list full takes the
// old direct route, so the capacity is a speed matter only.
#define H_SC 1024                     // LDS survivor slots per workgroup (12 KiB)
#define H_DENSE_BLOCK 24              // a wave's 16 x 64 block with more survivors than this bypasses the LDS list (see tile_epilogue16)

__device__ __forceinline__ void append_direct(const NwkTileParams& p, long long qrow, u64 key)
{
#if defined(H_DIAG_ATOMIC) && H_DIAG_ATOMIC == 1     // timing ablations (tools/mk_variant_lib.sh; wrong results): no list-length atomic at all ...
    const u32 pos = (u32)key & 511u;
#elif defined(H_DIAG_ATOMIC) && H_DIAG_ATOMIC == 2   // ... or one whose counters lie 8 KiB apart (a cache line each) instead of 32 to a line
    const u32 pos = atomicAdd((u32*)(p.buf + qrow * p.cap), 1u) & 511u;
#else
    const u32 pos = atomicAdd(p.cnt + qrow, 1u);
#endif
    if (pos < (u32)p.cap) p.buf[qrow * p.cap + pos] = key;
}

// every lane of the workgroup calls this after a barrier that follows the last append; the caller resets *s_cnt behind another barrier.
// s_rc (optional): H_TM words of LDS that nothing else uses during the call.  With it, a list of H_FLUSH_BY_ROW entries or more goes
// out with ONE list-length atomic per query row instead of one per entry: a device-scope atomic on p.cnt is resolved beyond L2, ~5 G/s for
// the whole chip, and in the early chunks of a walk (loose thresholds, hundreds of survivors per query and chunk) the tile kernel
// waited for nothing else -- C2 pre-screened: 0.8 ms for a chunk whose MFMAs take 0.02 ms (tools/chunk_trace.sh).
#define H_FLUSH_BY_ROW 96
__device__ __forceinline__ void flush_survivors(const NwkTileParams& p, LDS_AS const u32* s_cnt, LDS_AS const u64* s_key, LDS_AS u32* s_row,
                                                long long q0, int tid, int nthreads, LDS_AS u32* s_rc = nullptr)
{
    u32 n = *s_cnt;
    if (n > H_SC) n = H_SC;
    if (s_rc == nullptr || n < H_FLUSH_BY_ROW) {                     // workgroup-uniform
        for (u32 e = (u32)tid; e < n; e += (u32)nthreads) append_direct(p, q0 + s_row[e], s_key[e]);
        return;
    }
    for (int r = tid; r < H_TM; r += nthreads) s_rc[r] = 0;
    __syncthreads();
    for (u32 e = (u32)tid; e < n; e += (u32)nthreads) {              // rank of the entry among those of its row, kept in the entry's own word
        const u32 r = s_row[e];
        s_row[e] = r | (__hip_atomic_fetch_add(s_rc + r, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) << 8);
    }
    __syncthreads();
    for (int r = tid; r < H_TM; r += nthreads) {
        const u32 c = s_rc[r];
        if (c) s_rc[r] = atomicAdd(p.cnt + q0 + r, c);
    }
    __syncthreads();
    for (u32 e = (u32)tid; e < n; e += (u32)nthreads) {
        const u32 r = s_row[e] & 255u, pos = s_rc[r] + (s_row[e] >> 8);
        if (pos < (u32)p.cap) p.buf[(q0 + r) * p.cap + pos] = s_key[e];
    }
}

// Pre-screen, first pass over a finished tile (round 2): which of the wave's eight 16-row blocks can hold a survivor at all?
// The exact test of a block (below) costs an LDS round trip for its four row terms, ~33 VALU and a branch, eight times in a row with
// nothing to hide them behind -- ~3.4k cycles per tile while the matrix pipes idle.  A pair passes when acc >= row(q) + col(b); with
// pmin[m] = min of the block's four row terms (the query tile is the same for the whole strip: eight registers, loaded once) and
// cmin = min of the lane's four column terms,  fl(pmin + cmin) <= fl(row + col)  for every element of the block (rounded addition
// is monotone), so "max of the block's 16 accumulators >= pmin + cmin" holds whenever any of them passes: 8 v_max3, one add and
// one compare per block, no LDS, no branch.  The blocks it flags (a few percent) go through the exact test as before.
// H_FIRSTPASS (build-time variant, tools/mk_variant_lib.sh): 0 = the form above (shipped); 1 = the lane's own column term per column block
// (prescreen_block_mask_cols: 16 VALU per block instead of 10, ~10x fewer flagged blocks).  Round 5 re-measured round 3's finding with
// today's kernels, on the C3 shape (100k x 10M, d = 1024, same-box A/B, profiles/r05ad_firstpass_cols.log): ping-pong form 1 347 -> 1 317 TF
// (-2.2 %), two-issuing-waves form 1 335 -> 1 325 (-0.8 %), d = 128: -1.2 % -- although the ablations of the same day price everything
// behind the first pass at 5.4-7 % of the tile kernel (first pass only: 1 385 / 1 421 TF; no epilogue at all: 1 409 / 1 460;
// profiles/r05ac_epilogue_share_c3.log).  So that cost is not the exact tests of flagged-but-empty blocks: it is the APPENDS of real
// survivors (~420 per query and chunk whatever the kernel does: ~650 wave-cycles each), which no first pass can thin out.
#ifndef H_FIRSTPASS
#define H_FIRSTPASS 0
#endif
__device__ __forceinline__ u32 prescreen_block_mask_cols(const f32x4 (&acc)[8][4], const float (&pmin)[8], const float (&cpre)[4])
{
    u32 fired = 0;
    // An element passes the exact test when acc >= fl(row + col).  Then, in real arithmetic, acc - col >= row - u |row + col| >= pmin - u T, and the
    // rounded d = fl(acc - col) >= (acc - col) - u |acc - col|; with T <= max(|pmin|, |d|) + |col| and |acc - col| <= max(|pmin|, |d|) (+ lower order)
    // the test below -- the block's largest d against pmin lowered by 8 u (|pmin| + |d| + max |col|) -- fires whenever any element passes.
    const float cslack = 4.8e-7f * fmaxf(fmaxf(fabsf(cpre[0]), fabsf(cpre[1])), fmaxf(fabsf(cpre[2]), fabsf(cpre[3])));
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        float d[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const float mx = fmaxf(fmaxf(acc[m][n][0], acc[m][n][1]), fmaxf(acc[m][n][2], acc[m][n][3]));
            d[n] = mx - cpre[n];                                     // rounded subtraction is monotone: d[n] >= fl(acc - cpre[n]) for the block's elements
        }
        const float dm = fmaxf(fmaxf(d[0], d[1]), fmaxf(d[2], d[3]));
        const float thr = pmin[m] - (4.8e-7f * (fabsf(pmin[m]) + fabsf(dm)) + cslack);
        const bool f = !(dm < thr);                                  // NaN (inf - inf: an infinite column term of a padding column meets ...) flags the block
        if (__ballot(f)) fired |= 1u << m;
    }
    return fired;
}

__device__ __forceinline__ u32 prescreen_block_mask(const f32x4 (&acc)[8][4], const float (&pmin)[8], float cmin)
{
    u32 fired = 0;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        float mx = fmaxf(acc[m][0][0], acc[m][0][1]);
        mx = fmaxf(fmaxf(mx, acc[m][0][2]), acc[m][0][3]);
#pragma unroll
        for (int n = 1; n < 4; ++n) {
            mx = fmaxf(fmaxf(mx, acc[m][n][0]), acc[m][n][1]);
            mx = fmaxf(fmaxf(mx, acc[m][n][2]), acc[m][n][3]);
        }
        const bool f = !(mx < pmin[m] + cmin);                       // NaN (an infinite row term meeting an infinite column term) flags the block
        if (__ballot(f)) fired |= 1u << m;
    }
    return fired;
}

// Filtered pre-screen epilogue, the flagged blocks of a tile, COMPACT form (the ping-pong kernel).  tile_epilogue16 below unrolls over the
// wave's eight 16-row blocks: eight copies of the exact test and of the sixteen append sequences, ~50 KB of code per call site, and the
// ping-pong kernel has four call sites -- 228 KB of code, 5 spilled registers.  Here only the exact test is unrolled (in place, row terms
// in registers: no LDS round trip per flagged block); the blocks that hold a survivor are walked by a scalar loop over the set bits of a
// mask, a block's sixteen accumulators copied into fixed registers first (the fragment registers are dead during an epilogue), so there is
// ONE copy of the append code per call site: ~45 KB, no spills, no scratch, and the file compiles in half the time.  Speed of the
// compact form by itself: the same within +-1 % (C3 shard +0.6 %, C4 slice +2.4 %, 32k x 4M -1.1 % / +0.2 % on two boxes) -- the
// suspicion that the flagged blocks' cost was instruction fetch (a 64 KB instruction cache per two CUs) did not hold: a build in which
// the flagged blocks are never visited is 17 % faster with either form, so that time is the visits themselves, above all the appends
// (1.8 flagged blocks per wave-tile on C3, the slowest of a group's four waves decides; in-kernel stamps: ~2 400 cycles of epilogue per
// tile and wave).  Same tests, same keys as tile_epilogue16<METRIC, false, 1, SC>.
// Survivors go to a list of the WAVE's own (H_SCW slots of the workgroup's H_SC; its length a scalar the wave carries through the strip):
// no reservation to make, so no returning LDS atomic, no cross-lane shuffle of its result and no s_waitcnt the compiler puts in front of
// an LDS atomic while LDS-DMA is in flight -- ~500 cycles per (column block, row) register with a survivor before, on the critical path
// of a tile-boundary phase (the slowest of a group's four waves decides).  flush_wave_lists empties the eight lists at the end of the strip.
#define H_SCW (H_SC / 8)
template <int METRIC>
__device__ __forceinline__ void tile_epilogue16_flagged(const NwkTileParams& p, const f32x4 (&acc)[8][4], const float* s_qn,
                                                        const float (&bnv)[4], const u32 (&colv)[4], const bool (&colok)[4], const float (&cpre)[4],
                                                        long long q0, int wrow0, int g, int c16, LDS_AS u64* w_key, LDS_AS u32* w_row, u32& wcnt,
                                                        const f32x4 (&rowterm)[8], u32 block_mask)
{
    // the exact test of every flagged block, in place: "does any of the block's 16 x 64 pairs pass?" (1.8 blocks per wave-tile on C3, one
    // in eight of them holds a survivor).  Row terms come in registers (the first pass loaded them for its minima).
    u32 real = 0;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        if (!((block_mask >> m) & 1u)) continue;                     // wave-uniform
        float mx = -__builtin_inff();                                // operands are finite by construction (see tile_epilogue16)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const f32x4 dlt = acc[m][n] - (rowterm[m] + cpre[n]);
            mx = fmaxf(fmaxf(mx, dlt[0]), dlt[1]);
            mx = fmaxf(fmaxf(mx, dlt[2]), dlt[3]);
        }
        if (__ballot(!(mx < 0.0f))) real |= 1u << m;
    }
#if defined(H_DIAG_FLAGGED) && H_DIAG_FLAGGED == 1                  // timing ablation (wrong results): exact tests only, nothing is ever listed
    if (p.nb_valid >= 0) real = 0;
#endif
    // the blocks that hold a survivor: ONE copy of the append code, the block's accumulators and row terms copied into fixed registers
#pragma clang loop unroll(disable)
    while (real) {
        const int m = __builtin_ctz(real);
        real &= real - 1u;
        f32x4 a[4], pre4;
        switch (m) {
            // (the empty asm keeps the cases apart: merged into one load at a variable index, they would send the accumulators to scratch)
#define H_TAKE_BLOCK(M) a[0] = acc[M][0]; a[1] = acc[M][1]; a[2] = acc[M][2]; a[3] = acc[M][3]; pre4 = rowterm[M]; asm volatile("; block " #M : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(pre4)); break
            case 0: H_TAKE_BLOCK(0);
            case 1: H_TAKE_BLOCK(1);
            case 2: H_TAKE_BLOCK(2);
            case 3: H_TAKE_BLOCK(3);
            case 4: H_TAKE_BLOCK(4);
            case 5: H_TAKE_BLOCK(5);
            case 6: H_TAKE_BLOCK(6);
            default: H_TAKE_BLOCK(7);
#undef H_TAKE_BLOCK
        }
        const int rb = wrow0 + m * 16 + 4 * g;                       // this lane's four query rows: rb .. rb+3
        const f32x4 qn4 = *(const f32x4*)(s_qn + rb);
        // (Tried, round 5: the block's sixteen pass masks first -- packed adds, sixteen compares into scalar registers, no branch in between --
        // then one scalar test per mask, instead of the column-block test below: -4 % at d = 128, -2 % on C2, -1.8 % on the C3 shape,
        // profiles/r05ae_append_variant.log.)
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            {   // the block's test once more per column block: a block with a survivor usually has it in one of the four
                const f32x4 dlt = a[n] - (pre4 + cpre[n]);
                const float mxn = fmaxf(fmaxf(dlt[0], dlt[1]), fmaxf(dlt[2], dlt[3]));
                if (!__ballot(!(mxn < 0.0f))) continue;              // wave-uniform
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float sdot = a[n][i];                          // S * dot16
                const bool pass = !(sdot < pre4[i] + cpre[n]) & colok[n];
                const u64 mk = __ballot(pass);
                if (!mk) continue;                                   // wave-uniform
                const float est = (METRIC == NWKNN_METRIC_SQEUCLIDEAN) ? __fmaf_rn(-2.0f * p.pre_inv_scale, sdot, qn4[i] + bnv[n])
                                                                       : 1.0f - sdot * p.pre_inv_scale;
                const u64 key = ((u64)__float_as_uint(est) << 32) | colv[n];
                const int lane = (g << 4) | c16;
                if (pass) {
                    const u32 pos = wcnt + (u32)__popcll(mk & ((1ull << lane) - 1ull));
                    if (pos < (u32)H_SCW) { w_key[pos] = key; w_row[pos] = (u32)(rb + i); }
                    else append_direct(p, q0 + rb + i, key);
                }
                wcnt += (u32)__popcll(mk);                           // wave-uniform
            }
        }
    }
}

// every lane of the workgroup calls this behind a barrier that follows the waves' last appends and their stores of s_wcnt[wave]
__device__ __forceinline__ void flush_wave_lists(const NwkTileParams& p, LDS_AS const u32* s_wcnt, LDS_AS const u64* s_key, LDS_AS const u32* s_row,
                                                 long long q0, int tid, int nthreads)
{
#if defined(H_DIAG_FLAGGED) && H_DIAG_FLAGGED == 2                  // timing ablation (wrong results): the lists are filled but never emptied
    if (p.nb_valid >= 0) return;
#endif
    for (int e = tid; e < H_SC; e += nthreads) {
        const u32 n = s_wcnt[e / H_SCW];
        if ((u32)(e % H_SCW) < (n < (u32)H_SCW ? n : (u32)H_SCW)) append_direct(p, q0 + s_row[e], s_key[e]);
    }
}

template <int METRIC, bool DENSE, int PRE, int SC = 0, int AGG = 0>
__device__ __forceinline__ void tile_epilogue16(const NwkTileParams& p, const f32x4 (&acc)[8][4], const float* s_qn, const u64* s_tau,
                                                const float* s_pre, const float (&bnv)[4], const u32 (&colv)[4], const bool (&colok)[4],
                                                const float (&cpre)[4], long long q0, int wrow0, int g, int c16,
                                                LDS_AS u32* s_cnt = nullptr, LDS_AS u64* s_key = nullptr, LDS_AS u32* s_row = nullptr,
                                                u32 block_mask = 0xFFu)
{
    const int cap = p.cap;
    float bs[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) bs[n] = staged_norm<METRIC, PRE>(bnv[n]);   // cosine, direct filter: sqrtf(bn) once per tile (nwk_device.h, distance_from_staged)
#pragma unroll
    for (int m = 0; m < 8; ++m) {
        if (!((block_mask >> m) & 1u)) continue;                     // wave-uniform: prescreen_block_mask cleared this block
        const int rb = wrow0 + m * 16 + 4 * g;                       // this lane's four query rows: rb .. rb+3
        if (DENSE) {
            const f32x4 qn4 = *(const f32x4*)(s_qn + rb);
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    u32 hi;
                    if (PRE) {                                     // exact fp16 mode: every slot carries the estimate and is rescored
                        const float sdot = acc[m][n][i];
                        const float est = (METRIC == NWKNN_METRIC_SQEUCLIDEAN) ? __fmaf_rn(-2.0f * p.pre_inv_scale, sdot, qn4[i] + bnv[n])
                                                                               : 1.0f - sdot * p.pre_inv_scale;
                        hi = __float_as_uint(est);
                    } else {
                        const float dist = distance_from_staged<METRIC>(acc[m][n][i], qn4[i], bs[n]);
                        hi = (METRIC == NWKNN_METRIC_SQEUCLIDEAN) ? (__float_as_uint(dist) | 0x80000000u) : sortable_u32(dist);
                    }
                    const long long slot = (long long)(colv[n] - (u32)p.dense_col0);
                    p.buf[(q0 + rb + i) * cap + slot] = colok[n] ? (((u64)hi << 32) | colv[n]) : ~0ull;
                }
            continue;
        }
        const f32x4 pre4 = *(const f32x4*)(s_pre + rb);
        bool maybe = false;
        if (METRIC == NWKNN_METRIC_COSINE && !PRE) {              // the multiplicative split of prefilter_row / prefilter_col (nwk_device.h)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                bool mn = false;
#pragma unroll
                for (int i = 0; i < 4; ++i) mn |= !(acc[m][n][i] < pre4[i] * cpre[n]);
                maybe |= mn & colok[n];
            }
        } else if (PRE) {
            // pre-screen operands are finite by construction, so "some acc >= threshold" can be a running maximum of the
            // differences (v_max3), with no per-element compare into SGPRs
            float mx = -__builtin_inff();
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const f32x4 dlt = acc[m][n] - (pre4 + cpre[n]);
                mx = fmaxf(fmaxf(mx, dlt[0]), dlt[1]);
                mx = fmaxf(fmaxf(mx, dlt[2]), dlt[3]);
            }
            maybe = !(mx < 0.0f);
        } else {
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) maybe |= !(acc[m][n][i] < pre4[i] + cpre[n]);   // NaN falls through
        }
        if (!__ballot(maybe)) continue;                            // the common case: nothing in this 16 x 64 strip can survive
        const f32x4 qn4 = *(const f32x4*)(s_qn + rb);
        const u64x2 tl = *(const u64x2*)(s_tau + rb);
        const u64x2 th = *(const u64x2*)(s_tau + rb + 2);
        if (PRE && SC > 0 && AGG) {
            // (AGG: the streaming form, which has the registers for it.)  The block's 16 pass masks first, then ONE reservation for all of its survivors: in the LDS list (one ds_add_rtn per block), or
            // -- a block of a loose-threshold chunk, more survivors than the list is sized for -- straight in the queries' lists with one
            // list-length atomic per query ROW (the row's survivors of all four column blocks counted together).
            // (the lane keeps its own 16 pass bits in one register; the wave masks are re-made from them where needed, not held)
            u32 bits = 0, total = 0;
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const bool ps = !(acc[m][n][i] < pre4[i] + cpre[n]) & colok[n];
                    bits |= ps ? (1u << (4 * n + i)) : 0u;
                    total += (u32)__popcll(__ballot(ps));
                }
            if (!total) continue;
            const int lane = (g << 4) | c16;
            auto key_of = [&](int n, int i) {                          // see tile_epilogue: the key carries the fp16 estimate
                const float sdot = acc[m][n][i];                       // S * dot16 (this kernel's accumulators start at zero)
                const float est = (METRIC == NWKNN_METRIC_SQEUCLIDEAN) ? __fmaf_rn(-2.0f * p.pre_inv_scale, sdot, qn4[i] + bnv[n])
                                                                       : 1.0f - sdot * p.pre_inv_scale;
                return ((u64)__float_as_uint(est) << 32) | colv[n];
            };
            if (total > H_DENSE_BLOCK) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    u32 mq[4], c = 0;
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        mq[n] = (u32)(__ballot((bits >> (4 * n + i)) & 1u) >> (16 * g)) & 0xFFFFu;
                        c += (u32)__popc(mq[n]);
                    }
                    u32 base = 0;
                    if (c && c16 == 0) base = atomicAdd(p.cnt + q0 + rb + i, c);
                    base = __shfl(base, g << 4);
#pragma unroll
                    for (int n = 0; n < 4; ++n) {
                        if ((bits >> (4 * n + i)) & 1u) {
                            const u32 pos = base + (u32)__popc(mq[n] & ((1u << c16) - 1u));
                            if (pos < (u32)cap) p.buf[(q0 + rb + i) * cap + pos] = key_of(n, i);
                        }
                        base += (u32)__popc(mq[n]);
                    }
                }
            } else {
                u32 base = 0;
                if (lane == 0) base = __hip_atomic_fetch_add(s_cnt, total, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                base = __builtin_amdgcn_readfirstlane(base);
#pragma unroll
                for (int n = 0; n < 4; ++n)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const bool ps = (bits >> (4 * n + i)) & 1u;
                        const u64 mk = __ballot(ps);
                        if (ps) {
                            const u32 pos = base + (u32)__popcll(mk & ((1ull << lane) - 1ull));
                            if (pos < (u32)SC) { s_key[pos] = key_of(n, i); s_row[pos] = (u32)(rb + i); }
                            else append_direct(p, q0 + rb + i, key_of(n, i));
                        }
                        base += (u32)__popcll(mk);
                    }
            }
            continue;
        }
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            u64 key[4];
            bool pass[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (PRE) {                                         // see tile_epilogue: the key carries the fp16 estimate
                    const float sdot = acc[m][n][i];                 // S * dot16 (this kernel's accumulators start at zero)
                    const float est = (METRIC == NWKNN_METRIC_SQEUCLIDEAN) ? __fmaf_rn(-2.0f * p.pre_inv_scale, sdot, qn4[i] + bnv[n])
                                                                           : 1.0f - sdot * p.pre_inv_scale;
                    key[i] = ((u64)__float_as_uint(est) << 32) | colv[n];
                    pass[i] = !(acc[m][n][i] < pre4[i] + cpre[n]) & colok[n];
                    continue;
                }
                const float dist = distance_from_staged<METRIC>(acc[m][n][i], qn4[i], bs[n]);
                const u32 sd = (METRIC == NWKNN_METRIC_SQEUCLIDEAN) ? (__float_as_uint(dist) | 0x80000000u) : sortable_u32(dist);
                key[i] = ((u64)sd << 32) | colv[n];
                const u64 tau = (i & 2) ? th[i & 1] : tl[i & 1];
                pass[i] = (key[i] < tau) & colok[n];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u64 mk = __ballot(pass[i]);
                if (SC > 0) {
                    if (mk) {                                        // wave-uniform: one LDS atomic for the wave's survivors of this register
                        const int lane = (g << 4) | c16;
                        const int leader = __ffsll((long long)mk) - 1;
                        u32 base = 0;
                        if (lane == leader) base = __hip_atomic_fetch_add(s_cnt, (u32)__popcll(mk), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        base = __shfl(base, leader);
                        if (pass[i]) {
                            const u32 pos = base + (u32)__popcll(mk & ((1ull << lane) - 1ull));
                            if (pos < (u32)SC) { s_key[pos] = key[i]; s_row[pos] = (u32)(rb + i); }
                            else append_direct(p, q0 + rb + i, key[i]);
                        }
                    }
                    continue;
                }
                if (mk) {                                            // wave-uniform
                    const u32 mq = (u32)(mk >> (16 * g)) & 0xFFFFu;   // this quarter-wave's survivors: all for ONE query row
                    const int leader = mq ? (__ffs(mq) - 1) : 0;
                    u32 base = 0;
                    if (mq && c16 == leader) base = atomicAdd(p.cnt + q0 + rb + i, (u32)__popc(mq));
                    base = __shfl(base, leader + (g << 4));
                    if (pass[i]) {
                        const u32 pos = base + __popc(mq & ((1u << c16) - 1u));
                        if (pos < (u32)cap) p.buf[(q0 + rb + i) * cap + pos] = key[i];
                    }
                }
            }
        }
    }
}

