// nwk_tile_f16y.h -- fp16 distance tile + filter, "two issuing waves" form (option f16_mfma = 5): the kernel template, compiled by
// nwk_tile_f16y_pre.hip and nwk_tile_f16y_raw.hip.  (gfx950 only; shared pieces: nwk_device.h, nwk_f16_common.h)
//
// Same workgroup tile as the other forms (256 queries x 256 base rows, 8 waves 2 x 4, a wave owns 128 x 64 = 8 x 4 accumulators of
// v_mfma_f32_16x16x32_f16), same LDS image (two 64 KiB stages of 128-B rows, 16-B chunk c of row r in slot c ^ ((r >> 1) & 7)), same
// filter epilogue, same accumulation order per output element (ascending k) -- so the same bits.  What changes, again, is who does what
// when.  The ping-pong form (nwk_tile_f16x.h) lets ONE wave per SIMD own the matrix pipe at any time; round 5's probe
// (tools/f16y_probe.hip, profiles/r05a_*) shows what that costs: every non-MFMA instruction of the computing wave is a bubble nobody
// fills, and on operands that toggle nothing (no power limit) the structure reaches 81 % of the bare MFMA rate where two waves that BOTH
// issue MFMAs reach 88 % with the LDS-DMA stream in place.  This form is that arm:
//
//   * all eight waves run ONE program; every wave issues its eighth of a slab's LDS-DMA, reads its own fragments and issues its own MFMAs;
//   * ONE barrier per slab, placed in the MIDDLE of the slab's work.  Iteration s (between barriers M(s-1) and M(s)):
//         phase X:  32 MFMAs of (slab s-1, k-step 1) from fragment set 1   ||  8 LDS-DMA instructions of slab s+1 -> stage (s+1) & 1
//                                                                          ||  12 fragment reads of (slab s, k-step 0) -> set 0
//         phase Y:  32 MFMAs of (slab s,   k-step 0) from fragment set 0   ||  12 fragment reads of (slab s, k-step 1) -> set 1
//         s_waitcnt lgkmcnt(0) vmcnt(0);  s_barrier = M(s)
//     M(s) certifies (a) every wave's share of slab s+1 has landed (each waited for its own; requested right after M(s-1)) -> its fragments
//     may be read after M(s); (b) every wave holds ALL its fragments of slab s in registers (k-step 0 consumed, k-step 1 in set 1) -> stage
//     s & 1 may be refilled with slab s+2 right after M(s).  A wave arrives at the barrier with 32 MFMAs' worth of operands in registers:
//     nothing drains, the pipe restarts at once;
//   * a PRIORITY LADDER inside the interval.  The SIMD's arbiter prefers the older wave: without help waves 0-3 issue their 64 MFMAs in the
//     first 57 % of the interval and sit in the barrier for 875 of 2 048 cycles while their partners run alone (stamps in the probe) -- the
//     single-wave regime again.  So a wave lowers its own priority as it advances: the interval's 64 MFMAs are four chunks of 16 at
//     s_setprio 3, 2, 1, 0; ties go to the older wave, the order becomes A0 B0 A1 B1 A2 B2 A3 B3, and only the last chunk runs alone
//     (283 cycles in the barrier instead of 875);
//   * the instruction stream of a chunk is pinned (sched_group_barrier): per four MFMAs one LDS-DMA instruction and two fragment reads,
//     so that a wave's memory instructions fall into its partner's MFMA slots and the texture path sees an even stream.
// (Tried at the tile boundaries and dropped: STAGGERING the two groups there -- waves 0-3 at priorities 3, 2, waves 4-7 at 1, 0 in the boundary
// copies of the phases, so that each group's epilogue runs under the other's MFMAs instead of side by side with it: -0.4 % at d = 128, +0.5 %
// at 384, -0.1 % at 768, +0.2 % at 1024, same-box A/B, profiles/r05s_f16y_stagger_ab.log -- nothing.)
// A tile boundary (first slab of a tile): phase X carries no fragment reads -- both fragment sets are dead once its MFMAs are issued and
// the epilogue of the finished tile takes their registers -- then the epilogue, then the reads of set 0 and phase Y from a zero C.
//
// LDS-DMA shares (one instruction = 8 rows x 128 B = 1 KiB; a stage = A rows 0-255 | B rows 0-255 of one slab): wave w brings A pieces
// w + 8 j and B pieces w + 8 j (j = 0..3): all of one parity, hence ONE per-lane offset for the eight of them.  Column norms (and, PRE, the
// column terms) of a tile ride the same stream once per tile, brought by waves 0-3 with the tile's first slab.  Every wait is vmcnt(0): a
// wave requests a slab right behind a barrier and waits for it in front of the next one, a whole interval later.
#include <type_traits>

#include "nwk_f16_common.h"

#define Y_BN_BUFS 4                   // column-norm buffers (tile & 3)
#ifndef Y_DIAG
#define Y_DIAG 0                      // build-time timing ablations (tools/mk_variant_lib.sh; WRONG results): 8 = no epilogue, 128 = first pass only
#endif

template <int METRIC, int PRE>
__global__ __launch_bounds__(512, 2) void nwk_tile_kernel_f16y(const NwkTileParams p)
{
    constexpr int SC = H_SC;
    constexpr int OFF_EPI = 2 * H_STAGE, OFF_BN = OFF_EPI + H_TM * 16, OFF_TAB = OFF_BN + Y_BN_BUFS * H_TN * 4,
                  OFF_KEY = OFF_TAB + (PRE ? Y_BN_BUFS * H_TN * 4 : 0), OFF_ROW = OFF_KEY + SC * 8,
                  OFF_CNT = OFF_ROW + SC * 4;
    __shared__ __attribute__((aligned(16))) char smem[OFF_CNT + 32];    // PRE: eight list lengths (one per wave), else one
    char* const epi = smem + OFF_EPI;
    LDS_AS u64* const s_key = (LDS_AS u64*)(smem + OFF_KEY);
    LDS_AS u32* const s_row = (LDS_AS u32*)(smem + OFF_ROW);
    LDS_AS u32* const s_cnt = (LDS_AS u32*)(smem + OFF_CNT);

    const TileCoord tc = tile_from_block(p);
    if (!tc.ok) return;

    // Register budget: 128 accumulators + two fragment sets of 48 leave 32 registers for everything else, and a spilled LDS address in the K
    // loop is reloaded through scratch -- the compiler then waits vmcnt(0) for it, i.e. for the LDS-DMA in flight (the first build did, five
    // times per slab).  So what lives across the loop is kept to six vector registers: the lane number, the per-lane DMA offset and four LDS
    // addresses; everything wave-uniform is scalar, everything the epilogue needs is recomputed per tile from the lane number.
    const int tid = threadIdx.x;
    const int l = tid & 63;
    const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wu >> 2, wc = wu & 3;
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
    // piece pi of an operand = rows 8 pi .. 8 pi + 7; the swizzle term of row 8 pi + l / 8 is (4 (pi & 1) + l / 16) & 7: wave wu's pieces
    // wu + 8 j share the parity of wu, so one per-lane offset serves its four A and four B pieces.
    const unsigned voff = (unsigned)((l >> 3) * dp * 2 + (((l & 7) ^ ((4 * (wu & 1) + (l >> 4)) & 7)) << 4));
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(qp + q0 * dp), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(bp + b_first * dp), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsN = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bn + b_first), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsT = __builtin_amdgcn_make_buffer_rsrc((void*)((PRE ? p.ctab : p.bn) + b_first), 0, 0x7FFFFFFF, 0x00020000);
    auto issue_A1 = [&](int stage, int k, int j) __attribute__((always_inline)) {
        const int pi = wu + 8 * j;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)(smem + stage * H_STAGE + pi * 1024), 16, voff, k * 128 + pi * 8 * dp * 2, 0, 0);
    };
    auto issue_B1 = [&](int stage, int t, int k, int j) __attribute__((always_inline)) {
        const int pi = wu + 8 * j;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)(smem + stage * H_STAGE + 32768 + pi * 1024), 16, voff,
                                                 k * 128 + (t * H_TN + pi * 8) * dp * 2, 0, 0);
    };
    // half 0: this wave's four A pieces; half 1: its four B pieces (a chunk of 16 MFMAs carries one half)
    auto issue_half = [&](int half, int stage, int t, int k) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (half == 0) issue_A1(stage, k, j);
            else issue_B1(stage, t, k, j);
        }
    };
    auto issue_N = [&](int t) __attribute__((always_inline)) {    // waves 0-3: the 64 column norms of columns 64 u .. of tile t (+ PRE: their column terms)
        if (wu < 4) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsN, (LDS_AS void*)(smem + OFF_BN + (t & (Y_BN_BUFS - 1)) * H_TN * 4 + wu * 256), 4, (unsigned)(l * 4),
                                                     (t * H_TN + wu * 64) * 4, 0, 0);
            if (PRE)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsT, (LDS_AS void*)(smem + OFF_TAB + (t & (Y_BN_BUFS - 1)) * H_TN * 4 + wu * 256), 4,
                                                         (unsigned)(l * 4), (t * H_TN + wu * 64) * 4, 0, 0);
        }
    };

    // ---- fragments --------------------------------------------------------------------------------------------------------------------
    // LDS addresses of this lane's fragments in the CURRENT stage: [k-step]; the other stage is bit 16 away, so the four registers are
    // flipped once per slab (every read of an iteration goes to the stage of the slab phase Y works on) and the blocks are immediates
    int adrA[2], adrB[2];
    {
        const int g = l >> 4, c16 = l & 15;
        const int swz = (c16 >> 1) & 7;
        adrA[0] = (wr * 128 + c16) * 128 + ((g ^ swz) << 4);
        adrB[0] = 32768 + (wc * 64 + c16) * 128 + ((g ^ swz) << 4);
        adrA[1] = adrA[0] ^ 64; adrB[1] = adrB[0] ^ 64;
    }
    auto flip_stage = [&]() __attribute__((always_inline)) { adrA[0] ^= H_STAGE; adrA[1] ^= H_STAGE; adrB[0] ^= H_STAGE; adrB[1] ^= H_STAGE; };
    f32x4 acc[8][4];
    f16x8 fa[2][8], fb[2][4];                                     // [set = k-step][row block] / [set][column block]
    auto read_B = [&](int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < 4; ++n) fb[ks][n] = *(const f16x8*)(smem + adrB[ks] + n * 2048);
    };
    auto read_A = [&](int ks, int m0, int m1) __attribute__((always_inline)) {
#pragma unroll
        for (int m = m0; m < m1; ++m) fa[ks][m] = *(const f16x8*)(smem + adrA[ks] + m * 2048);
    };
    auto mfma_rows = [&](int ks, int m0, int m1, auto first_tag) __attribute__((always_inline)) {
        constexpr bool FIRST = decltype(first_tag)::value;        // the tile's first k-step starts from a zero C
        const f32x4 zero = {0.0f, 0.0f, 0.0f, 0.0f};
        // Issue order: A fragment m held over four MFMAs, the B fragments walked 0..3 and back 3..0 -- every two consecutive MFMAs share one
        // operand register.  The order is worth clock: bare loop on random operands (tools/f16y_probe.hip, profiles/r05an_mfma_order.log)
        // 2 012 TF with n always ascending, 2 049 TF snaked, 1 948 TF when both operands change every time.
#pragma unroll
        for (int m = m0; m < m1; ++m)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = NWK_MFMA_SNAKE && (m & 1) ? 3 - j : j;
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[ks][m], fb[ks][n], FIRST ? zero : acc[m][n], 0, 0, 0);
            }
    };
    // the instruction stream of one chunk of 16 MFMAs: four groups of {4 MFMAs, at most one LDS-DMA instruction, two fragment reads}
    auto pattern = [&](bool with_dma) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            if (with_dma) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
    };
#define Y_FENCE() __builtin_amdgcn_sched_barrier(0)
#define Y_PRIO(P) do { Y_FENCE(); __builtin_amdgcn_s_setprio(P); Y_FENCE(); } while (0)

    // ---- epilogue of tile t (this wave's 128 x 64 part) ------------------------------------------------------------------------
    u32 wcnt = 0;                                                   // PRE: length of this wave's survivor list (a scalar)
    auto epilogue = [&](int t) __attribute__((always_inline)) {
        if (Y_DIAG & 8) {                                           // keep the accumulators (and so the MFMAs) alive
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < 4; ++n) asm volatile("" :: "v"(acc[m][n]));
            return;
        }
        int l_here = l;                                             // opaque per tile: nothing derived from it is hoisted out of the strip loop
        asm volatile("" : "+v"(l_here));
        const int g = l_here >> 4, c16 = l_here & 15;
        const long long col0 = b_first + (long long)t * H_TN + wc * 64 + c16;
        const float* s_bn = (const float*)(smem + OFF_BN + (t & (Y_BN_BUFS - 1)) * H_TN * 4);
        float bnv[4], cpre[4];
        u32 colv[4];
        bool colok[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            const long long col = col0 + n * 16;
            bnv[n] = s_bn[wc * 64 + n * 16 + c16];
            colv[n] = (u32)col;
            colok[n] = col < p.nb_valid;
            if (PRE) cpre[n] = ((const float*)(smem + OFF_TAB + (t & (Y_BN_BUFS - 1)) * H_TN * 4))[wc * 64 + n * 16 + c16];
            else cpre[n] = colok[n] ? prefilter_col<METRIC>(bnv[n]) : __builtin_inff();
        }
        const int g_here = g, c_here = c16;
        u32 blocks = 0xFFu;
        f32x4 rowterm[8];                                           // PRE: this lane's row terms, kept for the exact tests of the flagged blocks
        if (PRE) {
            const float* s_pre = (const float*)(epi + H_TM * 12);
            float pmin[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                rowterm[m] = *(const f32x4*)(s_pre + wr * 128 + m * 16 + 4 * g_here);
                pmin[m] = fminf(fminf(rowterm[m][0], rowterm[m][1]), fminf(rowterm[m][2], rowterm[m][3]));
            }
            if (H_FIRSTPASS == 1) blocks = prescreen_block_mask_cols(acc, pmin, cpre);
            else blocks = prescreen_block_mask(acc, pmin, fminf(fminf(cpre[0], cpre[1]), fminf(cpre[2], cpre[3])));
        }
        if (Y_DIAG & 128) { asm volatile("" :: "s"(blocks)); blocks = 0; }   // first pass only: no block ever takes the exact test
        if (blocks) {
            if (PRE)
                tile_epilogue16_flagged<METRIC>(p, acc, (const float*)epi, bnv, colv, colok, cpre, q0, wr * 128, g_here, c_here, s_key + wu * H_SCW,
                                                s_row + wu * H_SCW, wcnt, rowterm, blocks);
            else
                tile_epilogue16<METRIC, false, PRE, SC>(p, acc, (const float*)epi, (const u64*)(epi + H_TM * 4), (const float*)(epi + H_TM * 12), bnv,
                                                        colv, colok, cpre, q0, wr * 128, g_here, c_here, s_cnt, s_key, s_row, blocks);
        }
    };

    // ---- the phases -------------------------------------------------------------------------------------------------------------
    // (tile, k-slab) of the slab that is requested next, kept incrementally and CLAMPED to the strip's last slab: the request behind the last
    // barrier asks for that slab once more, into the stage nobody reads again -- 1 / S more LDS-DMA instead of a branch in the pinned
    // instruction stream of every phase X (a scalar branch ends the basic block, and sched_group_barrier interleaves inside one).
    int t2 = 0, k2 = 0, nxt = 0;
    auto advance2 = [&]() __attribute__((always_inline)) {
        const bool step = nxt + 1 < S;                              // wave-uniform; s_cselect, no branch
        const bool wrap = k2 + 1 == nk;
        nxt += step ? 1 : 0;
        t2 += (step && wrap) ? 1 : 0;
        k2 = step ? (wrap ? 0 : k2 + 1) : k2;
    };
    // phase X of an iteration: the MFMAs of (previous slab, k-step 1) from set 1; the request of the next slab into the other stage;
    // WITH_READS: set 0 <- (this slab, k-step 0)
    auto phase_x = [&](int st, auto reads_tag) __attribute__((always_inline)) {
        constexpr bool WITH_READS = decltype(reads_tag)::value;
        const int tt = t2, kk = k2;
        Y_PRIO(3);
        issue_half(0, st ^ 1, tt, kk);
        if (WITH_READS) { read_B(0); read_A(0, 0, 4); }
        mfma_rows(1, 0, 4, std::false_type{});
        pattern(true);
        Y_PRIO(2);
        issue_half(1, st ^ 1, tt, kk);
        if (WITH_READS) read_A(0, 4, 8);
        mfma_rows(1, 4, 8, std::false_type{});
        pattern(true);
        Y_FENCE();
        advance2();
    };
    // phase Y: the MFMAs of (this slab, k-step 0) from set 0; set 1 <- (this slab, k-step 1); then the barrier M
    auto phase_y = [&](auto first_tag) __attribute__((always_inline)) {
        Y_PRIO(1);
        read_B(1); read_A(1, 0, 4);
        mfma_rows(0, 0, 4, first_tag);
        pattern(false);
        Y_PRIO(0);
        read_A(1, 4, 8);
        mfma_rows(0, 4, 8, first_tag);
        pattern(false);
        Y_FENCE();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this wave's share of the next slab has landed; its fragments of this slab are in registers
        __builtin_amdgcn_s_barrier();
        Y_FENCE();
        flip_stage();
    };

    // ---- prologue -----------------------------------------------------------------------------------------------------------------
    {
        float my_qn = 0.0f;
        u64 my_tau = 0;
        if (tid < H_TM) { my_qn = p.qn[q0 + tid]; my_tau = p.tau[q0 + tid]; }      // one query row per thread of waves 0-3
        Y_FENCE();
        issue_half(0, 0, 0, 0); issue_half(1, 0, 0, 0); issue_N(0);               // slab 0 -> stage 0, the first tile's column norms
        advance2();
        Y_FENCE();
        stage_query_side<METRIC, false, PRE>(epi, H_TM, tid, my_qn, my_tau, p, q0);
        if (tid == 0) *s_cnt = 0;
        Y_FENCE();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                               // slab 0 complete, epilogue inputs staged
        Y_FENCE();
        issue_half(0, 1, t2, k2); issue_half(1, 1, t2, k2);                         // slab 1 -> stage 1 (a one-slab strip: slab 0 again, a copy nobody reads)
        advance2();
        read_B(0); read_A(0, 0, 8);                                                 // set 0 <- (slab 0, k-step 0)
        Y_FENCE();
    }

    // ---- the strip ------------------------------------------------------------------------------------------------------------------
    int st = 0;                                                     // stage of the slab phase Y works on (scalar)
    for (int t = 0; t < n_here; ++t) {
        if (t > 0) {
            phase_x(st, std::false_type{});                         // the previous tile's last k-step; both fragment sets are dead behind it
            __builtin_amdgcn_s_setprio(0);
            issue_N(t);                                             // this tile's column norms (and terms): needed by its epilogue, a tile time from here
            epilogue(t - 1);
            Y_FENCE();
            read_B(0); read_A(0, 0, 8);                             // set 0 <- (first slab of tile t, k-step 0)
            Y_FENCE();
        }
        phase_y(std::true_type{});                                  // from a zero C
        st ^= 1;
        for (int kt = 1; kt < nk; ++kt) {
            phase_x(st, std::true_type{});
            phase_y(std::false_type{});
            st ^= 1;
        }
    }
    // the strip's last k-step (nothing left to request or to read), then its last epilogue
    Y_PRIO(3);
    mfma_rows(1, 0, 8, std::false_type{});
    Y_PRIO(0);
    epilogue(n_here - 1);
#undef Y_PRIO
#undef Y_FENCE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // no LDS-DMA may still be in flight when the workgroup leaves the CU
    if (PRE) {
        if (l == 0) s_cnt[wu] = wcnt;
        __syncthreads();
        flush_wave_lists(p, s_cnt, s_key, s_row, q0, tid, 512);
    } else {
        __syncthreads();
        flush_survivors(p, s_cnt, s_key, s_row, q0, tid, 512);
    }
}

// ------------------------------------------------------------------------------------------------
// launcher (host side, called from nwk_launch_tile_f16 in nwk_tile_f16.hip); two translation units as for the ping-pong form
// ------------------------------------------------------------------------------------------------
extern "C++" {

#if Y_PRE_INSTANCE
hipError_t nwk_launch_tile_f16y_pre(hipStream_t st, const NwkTileParams& p, int metric, unsigned grid)
#else
hipError_t nwk_launch_tile_f16y_raw(hipStream_t st, const NwkTileParams& p, int metric, unsigned grid)
#endif
{
#define NWK_Y(M) hipLaunchKernelGGL((nwk_tile_kernel_f16y<M, Y_PRE_INSTANCE>), dim3(grid), dim3(512), 0, st, p)
    if (metric == NWKNN_METRIC_SQEUCLIDEAN) NWK_Y(NWKNN_METRIC_SQEUCLIDEAN);
    else if (metric == NWKNN_METRIC_ONE_MINUS_DOT) NWK_Y(NWKNN_METRIC_ONE_MINUS_DOT);
    else if (metric == NWKNN_METRIC_COSINE) NWK_Y(NWKNN_METRIC_COSINE);
    else return hipErrorInvalidValue;
#undef NWK_Y
    return hipGetLastError();
}

}  // extern "C++"
