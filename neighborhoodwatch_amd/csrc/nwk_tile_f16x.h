// nwk_tile_f16x.h -- fp16 distance tile + filter, "ping-pong" form (option f16_mfma = 4): the kernel template, compiled by
// nwk_tile_f16x_pre.hip and nwk_tile_f16x_raw.hip.  (gfx950 only; shared pieces: nwk_device.h, nwk_f16_common.h)
//
// Same workgroup tile as the burst and the streaming form (256 queries x 256 base rows, 8 waves 2 x 4, a wave owns 128 x 64 = 8 x 4
// accumulators of v_mfma_f32_16x16x32_f16), same LDS image (two 64 KiB stages of 128-B rows, 16-B chunk c of row r in slot
// c ^ ((r >> 1) & 7)), same filter epilogue, same accumulation order per output element (ascending k) -- so the same bits.  What changes
// is WHO does WHAT WHEN.  In the other two forms all eight waves run one program: every phase each of them issues LDS-DMA, reads
// fragments and runs MFMAs, and the two waves that share a SIMD reach their DMA issue, their LDS reads, their MFMAs and the barrier
// together -- the matrix pipe idles while both are in their memory instructions and is fought over when both have operands (PMC, round 2:
// MFMA busy 55 % of SIMD cycles; a bare MFMA loop of this shape runs 1.6-1.7 PF on random data where the kernel ran 1.16-1.2).
//
// Here the two waves of a SIMD alternate roles, slab by slab (a slab = 64 halves of K = two k-steps = 64 MFMAs per wave):
//
//     group 0 = waves 0-3 (query rows   0-127 of the tile, one wave per SIMD)      group 1 = waves 4-7 (query rows 128-255)
//     phase 2s   : group 0 LOADS slab s                                             group 1 COMPUTES slab s-1
//     phase 2s+1 : group 0 COMPUTES slab s                                          group 1 LOADS slab s
//
//   LOAD    = issue this group's share of the LDS-DMA stream, (at a tile boundary: the previous tile's epilogue), then bring ALL of the
//             slab's fragments into registers: 16 A + 8 B ds_read_b128 (96 VGPRs), s_waitcnt, s_barrier;
//   COMPUTE = 64 MFMAs back to back on registers -- no LDS access, no vector memory, no wait in between -- then s_barrier.
// So at any time exactly one wave per SIMD owns the matrix pipe and has everything it needs in registers, while its partner's DMA issue,
// LDS round trips and epilogue VALU hide behind it (MI355X_MICROARCH.md, "Two waves per SIMD").  Two barriers per slab (the streaming
// form: four).
//
// LDS-DMA shares (one instruction = 8 rows x 128 B = 1 KiB; a stage = A rows 0-255 | B rows 0-255 of one slab).  A piece of LDS may be
// refilled once its last reader has finished with it, and the stream is dealt out so that (i) both groups issue about the same number of
// instructions per LOAD phase (the issuing wave sits in the texture unit's queue ~70 cycles per instruction: the L2 -> L1 path moves
// 64 B per clock and CU, so a slab's 64 KiB are ~1000 cycles of it however they are issued), and (ii) NO barrier ever waits for data in
// flight -- a wave waits for its own DMA in its LOAD phase, behind the partner's MFMAs:
//     group 0, LOAD of slab s (phase 2s):    A rows 128-255 of slab s+1, the LOWER half of ITS OWN 64 B rows (wave (0,c): rows 64c..64c+31)
//                                            of slab s+1 -> stage (s+1)&1 (4 + 4 per wave), and once per tile the next tile's column norms
//     group 1, LOAD of slab s (phase 2s+1):  A rows 0-127 of slab s+2, and -- once ITS OWN reads of slab s have landed -- the UPPER half of
//                                            its own B rows (wave (1,c): rows 64c+32..64c+63) of slab s+2 -> stage s&1   (4 + 4 per wave)
//   B rows 64c..64c+63 are read by exactly two waves, (0,c) in phase 2s and (1,c) in phase 2s+1: when (1,c)'s reads have landed the rows are
//   dead, and (1,c) itself may refill them without asking anyone; what a wave brought for itself it may read after its own s_waitcnt.
// Waits (a wave can only wait for its own DMA; what OTHER waves read of it is published by the barrier that follows the wait):
//     group 0, LOAD of slab s:  s_waitcnt vmcnt(0) FIRST -- its share of slab s, requested a whole slab period ago, nothing younger is
//              outstanding -- then all 24 fragment reads (upper A and the upper B half: group 1 waited for them a phase ago; its own lower
//              B half: the wait just made), then the DMA of slab s+1, under whose queue time the reads return;
//     group 1, LOAD of slab s:  all fragment reads, upper A of slab s+2 behind them, s_waitcnt lgkmcnt(0), its B half of slab s+2, then
//              s_waitcnt vmcnt(8): all but the eight just issued, i.e. its share of slab s+1 (requested a slab period ago).
//   Past the end of the strip group 1 still issues (source clamped, into pieces nobody reads again) so that its count stays a constant;
//   group 0 simply skips.  Every piece has at least a slab period to arrive, two slabs' worth of DMA are in flight.
// Epilogue inputs as in the streaming form: query norms | thresholds | row terms staged once per strip, column norms by DMA.
// Survivors are staged in the LDS list and flushed once at the end of the strip (an entry that finds the list full goes out
// directly); the first slab of a tile takes a zero C operand instead of cleared accumulators.
#include <type_traits>

#include "nwk_f16_common.h"

#define X_BN_BUFS 4                   // column-norm buffers (tile & 3)
#ifndef X_DIAG
#define X_DIAG 0
#endif
// build-time variants (tools/mk_variant_lib.sh), defaults = what measured fastest
#ifndef X_COMPACT
#define X_COMPACT 1                   // 1: the flagged blocks of a pre-screen epilogue go through ONE copy of the exact test (tile_epilogue16_flagged)
#endif
#ifndef X_PRIO
#define X_PRIO 1                      // 1: a wave raises its priority for its COMPUTE phase (s_setprio), so its MFMAs win the issue arbitration
#endif                                //    against the partner's LOAD-phase instructions whichever of the two is the older wave

// (Tried: the `nt` cache-policy bit on the LDS-DMA loads of the base rows, or of the query rows, so that the streamed operand does not push the
// re-used one out of L2: fabric bytes per launch 49 -> 82 GB / 66 GB and 8 % SLOWER -- a line marked non-temporal leaves L2 before the other
// workgroups of the super-tile have read it.  Bound for anything of the kind: with every operand L2-resident, X_DIAG=32, the kernel is 4.3 % faster.)

template <int METRIC, int PRE>
__global__ __launch_bounds__(512, 2) void nwk_tile_kernel_f16x(const NwkTileParams p)
{
    constexpr int SC = H_SC;
    constexpr int OFF_EPI = 2 * H_STAGE, OFF_BN = OFF_EPI + H_TM * 16, OFF_TAB = OFF_BN + X_BN_BUFS * H_TN * 4,
                  OFF_KEY = OFF_TAB + (PRE ? X_BN_BUFS * H_TN * 4 : 0), OFF_ROW = OFF_KEY + SC * 8,
                  OFF_CNT = OFF_ROW + SC * 4;
    __shared__ __attribute__((aligned(16))) char smem[OFF_CNT + 32];    // PRE: eight list lengths (one per wave), else one
    char* const epi = smem + OFF_EPI;
    LDS_AS u64* const s_key = (LDS_AS u64*)(smem + OFF_KEY);
    LDS_AS u32* const s_row = (LDS_AS u32*)(smem + OFF_ROW);
    LDS_AS u32* const s_cnt = (LDS_AS u32*)(smem + OFF_CNT);

    const TileCoord tc = tile_from_block(p);
    if (!tc.ok) return;
    // timing ablations (wrong results by construction): 1 no fragment reads, 2 no LDS-DMA inside the loop, 4 no MFMAs, 8 no epilogue, 16 DMA
    // waits one slab late (what a third LDS stage would allow), 32 all operands L2-resident, 128 the epilogue's first pass only (256: the same with the rest compiled in but never taken).  A
    // BUILD-time constant, 0 in every library the package builds (tools/mk_variant_lib.sh makes the ablated ones, under ab/)
    constexpr int diag = X_DIAG;

    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63;
    const int wr = w >> 2, wc = w & 3;
    const int g = l >> 4, c16 = l & 15;
    const long long q0 = (long long)tc.qt * H_TM;
    const int T = p.tiles_per_wg > 1 ? p.tiles_per_wg : 1;
    const int bt_first = p.bt0 + (tc.bt - p.bt0) * T;             // tc.bt numbers the strip
    int n_here = p.bt0 + p.n_btiles_real - bt_first;
    if (n_here > T) n_here = T;
    const long long b_first = (long long)bt_first * H_TN;
    const int dp = p.dp;
    const int nk = dp / H_BK;
    const int S = n_here * nk;                                    // slabs of the strip
    const _Float16* qp = (const _Float16*)p.qp;
    const _Float16* bp = (const _Float16*)p.bp;

    // ---- LDS-DMA ----------------------------------------------------------------------------------------------------------------------
    // piece pi of an operand = rows 8 pi .. 8 pi + 7.  Wave u (0..3) of a group brings pieces u + 4 j: they share the parity of pi, hence
    // the swizzle term ((row >> 1) & 7) = (4 (pi & 1) + l / 16) & 7 of row 8 pi + l / 8: one per-lane offset for all of them.
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int grp = wu >> 2, u = wu & 3;
    const unsigned voff0 = (unsigned)((l >> 3) * dp * 2 + (((l & 7) ^ ((l >> 4) & 7)) << 4));               // pieces of even pi
    const unsigned voff1 = (unsigned)((l >> 3) * dp * 2 + (((l & 7) ^ ((4 + (l >> 4)) & 7)) << 4));         // odd pi
    const unsigned voffA = (u & 1) ? voff1 : voff0;               // this wave's A pieces u + 4 j share the parity of u
    // (diag & 32: every workgroup streams the same few tiles -- 4 query tiles, one strip of base tiles -- so that nothing comes from beyond L2)
    const __amdgpu_buffer_rsrc_t rsA =
        __builtin_amdgcn_make_buffer_rsrc((void*)(qp + ((diag & 32) ? (long long)(tc.qt & 3) * H_TM : q0) * dp), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(bp + ((diag & 32) ? 0ll : b_first) * dp), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsN = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bn + b_first), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc((void*)((PRE ? p.ctab : p.bn) + b_first), 0, 0x7FFFFFFF, 0x00020000);
    auto issue_A1 = [&](int half, int stage, int k, int j) __attribute__((always_inline)) {      // piece j (0..3) of this wave's share of an A half
        const int pi = 16 * half + u + 4 * j;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)(smem + stage * H_STAGE + pi * 1024), 16, voffA, k * 128 + pi * 8 * dp * 2, 0, 0);
    };
    auto issue_B1 = [&](int half, int stage, int t, int k, int j) __attribute__((always_inline)) {
        const int pi = 8 * u + 4 * half + j;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)(smem + stage * H_STAGE + 32768 + pi * 1024), 16, (j & 1) ? voff1 : voff0,
                                                 k * 128 + (t * H_TN + pi * 8) * dp * 2, 0, 0);
    };
    auto issue_A = [&](int half, int stage, int k) {              // half 0: rows 0-127 (group 1 issues), 1: rows 128-255 (group 0)
#pragma unroll
        for (int j = 0; j < 4; ++j) issue_A1(half, stage, k, j);
    };
    auto issue_Bhalf = [&](int half, int stage, int t, int k) {   // the wave's OWN columns: B rows 64 u + 32 half .. + 31 = pieces 8 u + 4 half + j
#pragma unroll
        for (int j = 0; j < 4; ++j) issue_B1(half, stage, t, k, j);
    };
    auto issue_N = [&](int t) {                                   // wave u brings the 64 column norms of columns 64 u .. of tile t
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsN, (LDS_AS void*)(smem + OFF_BN + (t & (X_BN_BUFS - 1)) * H_TN * 4 + u * 256), 4, (unsigned)(l * 4),
                                                 (t * H_TN + u * 64) * 4, 0, 0);
        if (PRE)                                                    // ... and their column terms of this search (the epilogue's first pass tests against them;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsT, (LDS_AS void*)(smem + OFF_TAB + (t & (X_BN_BUFS - 1)) * H_TN * 4 + u * 256), 4,      // the norms serve the estimates of survivors only)
                                                     (unsigned)(l * 4), (t * H_TN + u * 64) * 4, 0, 0);
    };

    // ---- fragments --------------------------------------------------------------------------------------------------------------------
    const int swz = (c16 >> 1) & 7;
    const int offA0 = (wr * 128 + c16) * 128 + ((g ^ swz) << 4);
    const int offB0 = 32768 + (wc * 64 + c16) * 128 + ((g ^ swz) << 4);
    f32x4 acc[8][4];
    f16x8 fa[2][8], fb[2][4];                                     // [k-step][row block] / [k-step][column block] of ONE slab
    auto read_A = [&](int stage) {
        if (diag & 1) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int m = 0; m < 8; ++m) asm volatile("" : "+v"(fa[ks][m]));
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int m = 0; m < 8; ++m) fa[ks][m] = *(const f16x8*)(smem + stage * H_STAGE + (offA0 ^ (ks << 6)) + m * 2048);
    };
    auto read_B = [&](int stage) {
        if (diag & 1) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int n = 0; n < 4; ++n) asm volatile("" : "+v"(fb[ks][n]));
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int n = 0; n < 4; ++n) fb[ks][n] = *(const f16x8*)(smem + stage * H_STAGE + (offB0 ^ (ks << 6)) + n * 2048);
    };
    // 64 MFMAs on registers; FIRST: the tile's first slab starts from a zero C.  (Accumulators in the AGPR half of the register file --
    // inline-asm MFMAs with "a" constraints, 128 + 128 registers -- were tried to take the C / D traffic off the VGPR ports the partner's LDS
    // returns use: the K loop alone ran 4.5 % SLOWER, the whole kernel 7 %.)
    auto compute = [&](auto first_tag) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
        if (X_PRIO) __builtin_amdgcn_s_setprio(2);
        if (diag & 4) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {                        // keep the fragments (and so the reads) alive
#pragma unroll
                for (int m = 0; m < 8; ++m) asm volatile("" :: "v"(fa[ks][m]));
#pragma unroll
                for (int n = 0; n < 4; ++n) asm volatile("" :: "v"(fb[ks][n]));
            }
            if (FIRST) {
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n) acc[m][n] = zero;
            }
            return;
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = NWK_MFMA_SNAKE && (m & 1) ? 3 - j : j;      // snaked issue order: see nwk_tile_f16y.h
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[ks][m], fb[ks][n], (FIRST && ks == 0) ? zero : acc[m][n], 0, 0, 0);
                }
        if (X_PRIO) __builtin_amdgcn_s_setprio(0);
    };

    // ---- epilogue of tile t (this wave's 128 x 64 part) ------------------------------------------------------------------------
#define X_FENCE() __builtin_amdgcn_sched_barrier(0)
    // (Tried and dropped, twice: issuing a tile-boundary phase's LDS-DMA instructions one at a time between the blocks of the first-pass test, so
    // that their ~70 cycles each in the queue hide under the epilogue's vector work -- 2.5 % SLOWER on C3 and the C4 slice with the unrolled
    // epilogue (13 spilled registers instead of 5), 1 % slower with the compact one.)
    u32 wcnt = 0;                                                   // PRE: length of this wave's survivor list (a scalar)
    auto epilogue = [&](int t) __attribute__((always_inline)) {
        if (diag & 8) {                                             // keep the accumulators (and so the MFMAs) alive
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) asm volatile("" :: "v"(acc[m][n]));
            return;
        }
        // (Raising this wave's issue priority for the epilogue -- the partner's MFMAs have slack in a boundary phase -- changes nothing: +-0.2 %.)
        const long long col0 = b_first + (long long)t * H_TN + wc * 64 + c16;
        const float* s_bn = (const float*)(smem + OFF_BN + (t & (X_BN_BUFS - 1)) * H_TN * 4);
        float bnv[4], cpre[4];
        u32 colv[4];
        bool colok[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const long long col = col0 + n * 16;
            bnv[n] = s_bn[wc * 64 + n * 16 + c16];
            colv[n] = (u32)col;
            colok[n] = col < p.nb_valid;
            // PRE: the column term comes from the table of this search (prescreen_col evaluated once per base row by nwk_col_terms_kernel instead
            // of per tile and wave here: ~15 vector instructions and a square root per column block; +inf past the last row)
            if (PRE) cpre[n] = ((const float*)(smem + OFF_TAB + (t & (X_BN_BUFS - 1)) * H_TN * 4))[wc * 64 + n * 16 + c16];
            else cpre[n] = colok[n] ? prefilter_col<METRIC>(bnv[n]) : __builtin_inff();
        }
        int g_here = g, c_here = c16;                               // opaque per tile (see nwk_tile_kernel_f16s)
        asm volatile("" : "+v"(g_here));
        asm volatile("" : "+v"(c_here));
        u32 blocks = 0xFFu;
        f32x4 rowterm[8];                                           // PRE: this lane's row terms, kept for the exact tests of the flagged blocks
        if (PRE) {                                                  // smallest row term of each 16-row block: recomputed per tile (8 LDS reads)
            const float* s_pre = (const float*)(epi + H_TM * 12);   // rather than kept in 8 registers across the K loops
            float pmin[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                rowterm[m] = *(const f32x4*)(s_pre + wr * 128 + m * 16 + 4 * g_here);
                pmin[m] = fminf(fminf(rowterm[m][0], rowterm[m][1]), fminf(rowterm[m][2], rowterm[m][3]));
            }
            // (Tried: the lane's own column term per column block in this first pass instead of the smallest of its four -- ~10x fewer
            // flagged blocks, 16 instead of 10 VALU per block: 2 % SLOWER on C3, 1 % on the C4 slice; and a column term that costs two
            // instructions instead of prescreen_col's ~15 (timing experiment, what a per-row table brought by DMA would save): +1 %; a second
            // look at the flagged blocks only -- the same test per column block, ~17 VALU, before the exact test's LDS round trip: 0.9 % SLOWER
            // on 32k x 4M, 1.8 % on 100k x 1M.)
            if (H_FIRSTPASS == 1) blocks = prescreen_block_mask_cols(acc, pmin, cpre);
            else blocks = prescreen_block_mask(acc, pmin, fminf(fminf(cpre[0], cpre[1]), fminf(cpre[2], cpre[3])));
#ifdef NWKNN_DEV
            if (NWK_DBG(p) && (p.diag & 64)) {                      // dev census: how selective are the epilogue's tests?  (tools/f16x_census.py)
                float rmin = pmin[0], mx = -__builtin_inff();
#pragma unroll
                for (int m = 1; m < 8; ++m) rmin = fminf(rmin, pmin[m]);
#pragma unroll
                for (int m = 0; m < 8; ++m)
#pragma unroll
                    for (int n = 0; n < 4; ++n)
#pragma unroll
                        for (int i = 0; i < 4; ++i) mx = fmaxf(mx, acc[m][n][i]);
                const bool coarse = __ballot(!(mx < rmin + fminf(fminf(cpre[0], cpre[1]), fminf(cpre[2], cpre[3])))) != 0;
                if (l == 0) {
                    unsigned long long* c = NWK_DBG(p) + (size_t)(1u << 23);      // behind the stamp area
                    atomicAdd(c + 0, 1ull); atomicAdd(c + 1, blocks ? 1ull : 0ull); atomicAdd(c + 2, coarse ? 1ull : 0ull);
                    atomicAdd(c + 3, (unsigned long long)__popc(blocks));
                }
            }
#endif
        }
        if (diag & 128) { asm volatile("" :: "s"(blocks)); blocks = 0; }   // first pass only: no block ever takes the exact test
        if ((diag & 256) && p.nb_valid >= 0) blocks = 0;            // the same, with the exact test still compiled in (never taken at run time)
        if (blocks) {
            if (PRE && X_COMPACT)
                tile_epilogue16_flagged<METRIC>(p, acc, (const float*)epi, bnv, colv, colok, cpre, q0, wr * 128, g_here, c_here, s_key + wu * H_SCW,
                                                s_row + wu * H_SCW, wcnt, rowterm, blocks);
            else
                tile_epilogue16<METRIC, false, PRE, SC>(p, acc, (const float*)epi, (const u64*)(epi + H_TM * 4), (const float*)(epi + H_TM * 12), bnv,
                                                        colv, colok, cpre, q0, wr * 128, g_here, c_here, s_cnt, s_key, s_row, blocks);
        }
    };


    // developer builds: per-wave cycle sums of the pieces of a phase (tools/f16x_stamps.py).  Slots: 0 DMA issue, 1 epilogue, 2 fragment
    // reads until they have landed, 3 wait at the barrier that ends the LOAD phase, 4 MFMA issue, 5 wait for this wave's DMA, 6 wait at
    // the barrier that ends the COMPUTE phase.  The product compiles none of it.
#ifdef NWKNN_DEV
    unsigned long long tsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const bool stamp_on = NWK_DBG(p) != nullptr && !(p.diag & 64);    // diag 64: the buffer takes the census counters instead
    unsigned long long tlast = stamp_on ? __builtin_amdgcn_s_memtime() : 0;
#define X_STAMP(I)                                                     \
    do {                                                               \
        if (stamp_on) {                                                \
            X_FENCE();                                                 \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
            tsum[I] += now_ - tlast;                                   \
            tlast = now_;                                              \
            X_FENCE();                                                 \
        }                                                              \
    } while (0)
#else
#define X_STAMP(I) do { } while (0)
#endif
#define X_BARRIER(WAIT, SLOT_WAIT, SLOT_BAR)               \
    do {                                                   \
        X_FENCE();                                         \
        asm volatile(WAIT ::: "memory");                   \
        X_STAMP(SLOT_WAIT);                                \
        __builtin_amdgcn_s_barrier();                      \
        X_STAMP(SLOT_BAR);                                 \
        X_FENCE();                                         \
    } while (0)

    if (grp == 0) {
        // =========================================================== group 0 ===========================================================
        float my_qn = p.qn[q0 + tid];                             // tid < 256 here: one query row per thread
        u64 my_tau = p.tau[q0 + tid];
        X_FENCE();
        issue_A(1, 0, 0); issue_Bhalf(0, 0, 0, 0); issue_N(0);    // this group's share of slab 0 -> stage 0
        X_FENCE();
        stage_query_side<METRIC, false, PRE>(epi, H_TM, tid, my_qn, my_tau, p, q0);
        if (tid == 0) *s_cnt = 0;
        X_BARRIER("s_waitcnt vmcnt(0) lgkmcnt(0)", 7, 7);         // [pre] slab 0 complete (group 1 waited for its part), epilogue inputs staged
        int sigma = 0;
        // one LOAD phase: A fragments, this group's DMA for the NEXT slab, the wait for its DMA of THIS slab, B fragments
        auto load = [&](int t, int kt, bool with_epilogue) __attribute__((always_inline)) {
            const int st = sigma & 1;
            if (with_epilogue) { epilogue(t - 1); X_STAMP(1); X_FENCE(); }
            // this group's share of slab sigma was requested a whole slab period ago: wait for it FIRST (nothing younger is outstanding), put
            // all 24 fragment reads in flight, and let them return under the ~70 cycles each of the DMA instructions sits in the queue.
            // (The other order -- A reads, DMA, vmcnt(9), B reads -- measured 0.5 % slower.)  The column norms are requested once per tile,
            // with its first slab (round 3's first version re-requested them with every slab to keep a count constant: 15 of 16 instructions
            // for nothing).
            if (!(diag & 16)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            X_STAMP(5);
            X_FENCE();
            read_A(st);
            read_B(st);
            X_FENCE();
            if (sigma + 1 < S && !(diag & 2)) {                     // wave-uniform
                const bool wrap = kt + 1 == nk;
                const int t2 = wrap ? t + 1 : t, k2 = wrap ? 0 : kt + 1;
                issue_A(1, st ^ 1, k2); issue_Bhalf(0, st ^ 1, t2, k2);
                if (wrap) issue_N(t2);
            }
            X_STAMP(0);
            X_BARRIER("s_waitcnt lgkmcnt(0)", 2, 3);
        };
        for (int t = 0; t < n_here; ++t) {
            load(t, 0, t > 0);                                      // first slab of tile t: + the previous tile's epilogue, COMPUTE from a zero C
            compute(std::true_type{});
            X_STAMP(4);
            X_BARRIER("s_nop 0", 6, 6);
            ++sigma;
            for (int kt = 1; kt < nk; ++kt) {
                load(t, kt, false);
                compute(std::false_type{});
                X_STAMP(4);
                X_BARRIER("s_nop 0", 6, 6);
                ++sigma;
            }
        }
        epilogue(n_here - 1);                           // while group 1 computes the strip's last slab
    } else {
        // =========================================================== group 1 ===========================================================
        int t2 = 0, k2 = 0;                                         // (tile, k-slab) of slab sigma + 2, kept incrementally
        auto advance2 = [&]() { if (++k2 == nk) { k2 = 0; ++t2; } };
        X_FENCE();
        issue_A(0, 0, 0); issue_Bhalf(1, 0, 0, 0);                  // upper A and this wave's B half of slab 0 -> stage 0
        advance2();                                                 // -> slab 1 (a one-slab strip: slab 0 again, a copy nobody reads)
        {
            const bool has1 = S > 1;
            issue_A(0, 1, has1 ? k2 : 0); issue_Bhalf(1, 1, has1 ? t2 : 0, has1 ? k2 : 0);
        }
        advance2();                                                 // -> slab 2
        X_BARRIER("s_waitcnt vmcnt(8)", 7, 7);                      // [pre] slab 0's part has landed
        X_BARRIER("s_nop 0", 7, 7);                                 // end of phase 0 (group 0 loaded slab 0; nothing to compute yet)
        int sigma = 0;
        auto load = [&](int t, bool with_epilogue) __attribute__((always_inline)) {
            const int st = sigma & 1;
            const bool real = sigma + 2 < S;                        // (wave-uniform) nothing is requested past the end of the strip
            const int tt = t2, kk = k2;
            if (with_epilogue) { epilogue(t - 1); X_STAMP(1); X_FENCE(); }
            read_A(st);
            read_B(st);
            X_FENCE();
            if (!(diag & 2) && real) issue_A(0, st, kk);            // upper A of slab sigma + 2 (group 0 read this stage's a phase ago)
            X_STAMP(0);
            X_FENCE();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's reads have landed: its B rows of slab sigma are dead
            X_STAMP(2);
            X_FENCE();
            if (!(diag & 2) && real) issue_Bhalf(1, st, tt, kk);
            advance2();
            if ((diag & 2) || !real) X_BARRIER("s_waitcnt vmcnt(0)", 5, 3);      // the strip's last two slabs: this group's share of slab sigma + 1 is the youngest there is
            else if (diag & 16) X_BARRIER("s_waitcnt vmcnt(16)", 5, 3);      // timing experiment (WRONG results)
            else X_BARRIER("s_waitcnt vmcnt(8)", 5, 3);
        };
        for (int t = 0; t < n_here; ++t) {
            load(t, t > 0);
            compute(std::true_type{});
            X_STAMP(4);
            ++sigma;
            if (sigma < S) X_BARRIER("s_nop 0", 6, 6);
            for (int kt = 1; kt < nk; ++kt) {
                load(t, false);
                compute(std::false_type{});
                X_STAMP(4);
                ++sigma;
                if (sigma < S) X_BARRIER("s_nop 0", 6, 6);
            }
        }
        epilogue(n_here - 1);
        X_STAMP(1);
    }
#ifdef NWKNN_DEV
    if (stamp_on && l == 0) {
        unsigned long long* o = NWK_DBG(p) + ((size_t)blockIdx.x * 8 + w) * 8;
#pragma unroll
        for (int i = 0; i < 7; ++i) o[i] = tsum[i];
        o[7] = (unsigned long long)S | ((tsum[7] & 0xFFFFFFull) << 16);           // slabs | cycles from kernel entry to the end of the [pre] phase
    }
#endif
#undef X_BARRIER
#undef X_STAMP
#undef X_FENCE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // no LDS-DMA may still be in flight when the workgroup leaves the CU
    if (PRE && X_COMPACT) {
        if (l == 0) s_cnt[wu] = wcnt;
        __syncthreads();
        flush_wave_lists(p, s_cnt, s_key, s_row, q0, tid, 512);
#ifdef NWKNN_DEV
        if (stamp_on && l == 0)                                     // the tail of the workgroup: last stamp of the strip -> the lists' stores issued
            NWK_DBG(p)[((size_t)blockIdx.x * 8 + w) * 8 + 7] |= ((__builtin_amdgcn_s_memtime() - tlast) & 0xFFFFFFull) << 40;
#endif
    } else {
        __syncthreads();
        flush_survivors(p, s_cnt, s_key, s_row, q0, tid, 512);
    }
}

// ------------------------------------------------------------------------------------------------
// launcher (host side, called from nwk_launch_tile_f16 in nwk_tile_f16.hip).  The kernel is instantiated in two translation units,
// nwk_tile_f16x_pre.hip (PRE = 1: pre-screen) and nwk_tile_f16x_raw.hip (PRE = 0: raw fp16 sums), so that the two halves of the ~60 s
// this template costs hipcc compile side by side; each defines X_PRE_INSTANCE before including this file.
// ------------------------------------------------------------------------------------------------
extern "C++" {

#if X_PRE_INSTANCE
// the pre-screen's column term of every base row for one search (it depends on the search's scale and error constants): what the ping-pong
// kernel's epilogue used to evaluate per tile, wave and column block
template <int METRIC>
__global__ __launch_bounds__(256) void nwk_col_terms_kernel(const NwkTileParams p, long long rows_pad, float* out)
{
    for (long long b = (long long)blockIdx.x * 256 + threadIdx.x; b < rows_pad; b += (long long)gridDim.x * 256)
        out[b] = b < p.nb_valid ? prescreen_col<METRIC>(p.bn[b], p) : __builtin_inff();
}

hipError_t nwk_launch_col_terms(hipStream_t st, const NwkTileParams& p, int metric, long long rows_pad, float* out)
{
    if (rows_pad <= 0) return hipSuccess;
    long long blocks = (rows_pad + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (metric == NWKNN_METRIC_SQEUCLIDEAN) hipLaunchKernelGGL(nwk_col_terms_kernel<NWKNN_METRIC_SQEUCLIDEAN>, dim3((unsigned)blocks), dim3(256), 0, st, p, rows_pad, out);
    else if (metric == NWKNN_METRIC_ONE_MINUS_DOT) hipLaunchKernelGGL(nwk_col_terms_kernel<NWKNN_METRIC_ONE_MINUS_DOT>, dim3((unsigned)blocks), dim3(256), 0, st, p, rows_pad, out);
    else if (metric == NWKNN_METRIC_COSINE) hipLaunchKernelGGL(nwk_col_terms_kernel<NWKNN_METRIC_COSINE>, dim3((unsigned)blocks), dim3(256), 0, st, p, rows_pad, out);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

hipError_t nwk_launch_tile_f16x_pre(hipStream_t st, const NwkTileParams& p, int metric, unsigned grid)
#else
hipError_t nwk_launch_tile_f16x_raw(hipStream_t st, const NwkTileParams& p, int metric, unsigned grid)
#endif
{
#define NWK_X(M) hipLaunchKernelGGL((nwk_tile_kernel_f16x<M, X_PRE_INSTANCE>), dim3(grid), dim3(512), 0, st, p)
    if (metric == NWKNN_METRIC_SQEUCLIDEAN) NWK_X(NWKNN_METRIC_SQEUCLIDEAN);
    else if (metric == NWKNN_METRIC_ONE_MINUS_DOT) NWK_X(NWKNN_METRIC_ONE_MINUS_DOT);
    else if (metric == NWKNN_METRIC_COSINE) NWK_X(NWKNN_METRIC_COSINE);
    else return hipErrorInvalidValue;
#undef NWK_X
    return hipGetLastError();
}

}  // extern "C++"
