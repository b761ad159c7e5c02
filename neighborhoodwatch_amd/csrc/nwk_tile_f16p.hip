// nwk_tile_f16p.hip -- fp16 distance tile + filter, "streaming" form (option f16_mfma = 3).  (gfx950 only; shared pieces: nwk_device.h,
// nwk_f16_common.h)
//
// Same workgroup tile as nwk_tile_kernel_f16s (256 queries x 256 base rows, 8 waves 2 x 4, a wave owns 128 x 64 = 8 x 4 accumulators of
// v_mfma_f32_16x16x32_f16), same LDS image (two 64 KiB stages of 128-B rows, 16-B chunk c of row r in slot c ^ ((r >> 1) & 7)), same
// filter epilogue.  What changes is WHEN the operands move.  The f16s kernel requests a whole 64 KiB slab in one burst of 64 LDS-DMA
// instructions behind its one barrier per slab and waits for it one slab period later: the DMA queue of the CU sees a burst, then
// nothing, and every slab has exactly one slab period to arrive.  Here a slab is cut into the pieces a phase consumes, and a piece's
// LDS space is refilled -- with the slab two ahead -- in the very phase after its last read:
//
//   phase j of slab s (16 MFMAs per wave: query-row blocks 2j, 2j+1  x  4 column blocks  x  2 k-steps)
//     start:  s_waitcnt lgkmcnt(0) [+ counted vmcnt, phases 2 and 3]; s_barrier
//             -> every wave has finished the LDS reads it issued in the previous phase, and what those waits cover has landed
//     then:   LDS-DMA for slab s+2:   j=0: A-quarter 0      j=1: A-quarter 1, B-quarters 0,1
//                                     j=2: A-quarter 2, B-quarters 2,3      j=3: A-quarter 3, the tile's column norms
//             (A-quarter j = the 32 query rows 32j..32j+31 of both wave rows, B-quarter c = base rows 64c..64c+63: 8 KiB each, one
//             DMA instruction per wave)
//     then:   fragment reads for the NEXT phase (A-quarter j+1; in phase 0 also B k-step 1 of this slab, in phase 3 A-quarter 0 and B
//             k-step 0 of the next slab) under this phase's MFMAs.
//
// The B fragments of a slab live in registers for the whole slab, so B's LDS space is free after phase 0; every A-quarter is read one
// phase before it is used.  Every piece therefore has 5-7 phases (1.25-1.75 slab periods) to arrive instead of 4, the CU issues 1-3 DMA
// instructions per wave and phase instead of 8 in one, and -- because the stream of slabs runs on across the tiles of a strip -- a
// tile boundary costs the epilogue and nothing else: no drain, no refill burst, no first-slab round trip.  The price is four barriers
// per slab instead of one (measured on the f16s kernel: 1.6 %).
//
// Counted waits.  Per wave the DMA instructions are issued in one fixed order, slab after slab:  A0 | A1 B0 B1 | A2 B2 B3 | A3 N
// (9 per slab; the strip's first two slabs are issued in the same order up front).  Phase 3 of slab s reads A-quarter 0 and B of slab
// s+1, the youngest of which (B3 of s+1) is followed by A3, N of slab s+1 and the 7 instructions of slab s+2 issued so far: vmcnt(9).
// Phase 2 reads A-quarter 3 of slab s; behind it: N(s), the 9 of slab s+1, A0 A1 B0 B1 of slab s+2: vmcnt(14).  Everything else a
// phase reads is older than what the previous counted wait covered.  Instructions past the end of the strip are still issued (source
// clamped to the last slab, destination a piece nobody reads again) so that the counts are constants; other memory instructions
// (the rare direct append of a survivor that found the LDS list full) only make the waits stricter.
#include "nwk_f16_common.h"

#define P_BN_BUFS 4                   // column-norm buffers (tile & 3): the DMA for tile t+2 may run while tile t is in its epilogue

template <int METRIC, int PRE, int AGG>
__global__ __launch_bounds__(512, 2) void nwk_tile_kernel_f16p(const NwkTileParams p)
{
    constexpr int SC = H_SC;
    constexpr int OFF_EPI = 2 * H_STAGE, OFF_BN = OFF_EPI + H_TM * 16, OFF_KEY = OFF_BN + P_BN_BUFS * H_TN * 4, OFF_ROW = OFF_KEY + SC * 8,
                  OFF_CNT = OFF_ROW + SC * 4;
    constexpr int OFF_RC = OFF_CNT + 16;                           // AGG: row counters of the aggregated flush (flush_survivors)
    __shared__ __attribute__((aligned(16))) char smem[OFF_RC + (AGG ? H_TM * 4 : 0)];
    char* const epi = smem + OFF_EPI;
    LDS_AS u64* const s_key = (LDS_AS u64*)(smem + OFF_KEY);
    LDS_AS u32* const s_row = (LDS_AS u32*)(smem + OFF_ROW);
    LDS_AS u32* const s_cnt = (LDS_AS u32*)(smem + OFF_CNT);
    LDS_AS u32* const s_rc = AGG ? (LDS_AS u32*)(smem + OFF_RC) : nullptr;

    const TileCoord tc = tile_from_block(p);
    if (!tc.ok) return;

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
    const int NS = n_here * nk;                                   // slabs of the strip
    const _Float16* qp = (const _Float16*)p.qp;
    const _Float16* bp = (const _Float16*)p.bp;

    // ---- LDS-DMA ----------------------------------------------------------------------------------------------------------------------
    // one instruction = 8 rows x 128 B.  A-quarter j: wave w brings rows 128 (w / 4) + 32 j + 8 (w % 4) + [0, 8); B-quarter c: rows
    // 64 c + 8 w + [0, 8).  In both the swizzle term ((row >> 1) & 7) of row (... + l / 8) is (4 (w & 1) + l / 16) & 7: one lane offset.
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const unsigned voff = (unsigned)((l >> 3) * dp * 2 + (((l & 7) ^ ((4 * (wu & 1) + (l >> 4)) & 7)) << 4));
    const int rowA = 128 * (wu >> 2) + 8 * (wu & 3), rowB = 8 * wu;      // + 32 j / + 64 c
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(qp + q0 * dp), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(bp + b_first * dp), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsN = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bn + b_first), 0, 0x7FFFFFFF, 0x00020000);
    // the slab two ahead of the one being computed, clamped to the strip's last slab: (tile, k-slab) and its stage
    int t2 = 0, k2 = 0, st2 = 0;
    auto advance2 = [&]() {
        st2 ^= 1;
        if (t2 * nk + k2 + 1 < NS) { if (++k2 == nk) { k2 = 0; ++t2; } }
    };
    auto issue_A = [&](int j) {
        char* dst = smem + st2 * H_STAGE + (rowA + 32 * j) * 128;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)dst, 16, voff, k2 * 128 + (rowA + 32 * j) * dp * 2, 0, 0);
    };
    auto issue_B = [&](int c) {
        char* dst = smem + st2 * H_STAGE + 32768 + (rowB + 64 * c) * 128;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)dst, 16, voff, k2 * 128 + (t2 * H_TN + rowB + 64 * c) * dp * 2, 0, 0);
    };
    auto issue_N = [&]() {                                        // this wave's 64 column norms of tile t2 (both wave rows bring the same bytes)
        char* dst = smem + OFF_BN + (t2 & (P_BN_BUFS - 1)) * H_TN * 4 + (wu & 3) * 256;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsN, (LDS_AS void*)dst, 4, (unsigned)(l * 4), (t2 * H_TN + (wu & 3) * 64) * 4, 0, 0);
    };
    auto issue_slab = [&]() { issue_A(0); issue_A(1); issue_B(0); issue_B(1); issue_A(2); issue_B(2); issue_B(3); issue_A(3); issue_N(); };

    // ---- fragments ------------------------------------------------------------------------------------------------------------------------
    const int swz = (c16 >> 1) & 7;
    const int offA0 = (wr * 128 + c16) * 128 + ((g ^ swz) << 4);
    const int offB0 = 32768 + (wc * 64 + c16) * 128 + ((g ^ swz) << 4);
    f32x4 acc[8][4];
    f16x8 fa[2][4], fb[2][4];                                     // fa[buf][2 ks + mb]: row blocks 2j+mb of one phase; fb[ks][n]
    auto read_A = [&](int buf, int stage, int j) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
                fa[buf][2 * ks + mb] = *(const f16x8*)(smem + stage * H_STAGE + (offA0 ^ (ks << 6)) + (2 * j + mb) * 2048);
    };
    auto read_B = [&](int ks, int stage) {
#pragma unroll
        for (int n = 0; n < 4; ++n) fb[ks][n] = *(const f16x8*)(smem + stage * H_STAGE + (offB0 ^ (ks << 6)) + n * 2048);
    };
    auto mma = [&](int buf, int j, int ks, int mb) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int n = NWK_MFMA_SNAKE && (mb & 1) ? 3 - i : i;             // snaked issue order (nwk_tile_f16y.h)
            acc[2 * j + mb][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[buf][2 * ks + mb], fb[ks][n], acc[2 * j + mb][n], 0, 0, 0);
        }
    };
    auto zero_acc = [&]() {
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[m][n][i] = 0.0f;
    };

    // ---- strip prologue ---------------------------------------------------------------------------------------------------------------
    float my_qn = 0.f;
    u64 my_tau = ~0ull;
    if (tid < H_TM) { my_qn = p.qn[q0 + tid]; my_tau = p.tau[q0 + tid]; }
    __builtin_amdgcn_sched_barrier(0);
    issue_slab();                                                 // slab 0 -> stage 0
    advance2();
    issue_slab();                                                 // slab 1 -> stage 1
    advance2();
    __builtin_amdgcn_sched_barrier(0);
    stage_query_side<METRIC, false, PRE>(epi, H_TM, tid, my_qn, my_tau, p, q0);
    if (tid == 0) *s_cnt = 0;
    zero_acc();
    asm volatile("s_waitcnt vmcnt(11) lgkmcnt(0)" ::: "memory");  // A-quarter 0 and B of slab 0 have landed (behind B3: A3, N and slab 1's nine)
    __builtin_amdgcn_s_barrier();
    float pmin[8];                                                // pre-screen: smallest row term of each of this wave's 16-row blocks (prescreen_block_mask)
    if (PRE) {
        const float* s_pre = (const float*)(epi + H_TM * 12);
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const f32x4 r4 = *(const f32x4*)(s_pre + wr * 128 + m * 16 + 4 * g);
            pmin[m] = fminf(fminf(r4[0], r4[1]), fminf(r4[2], r4[3]));
        }
    }
    read_A(0, 0, 0);
    read_B(0, 0);

#define P_HEAD(WAIT)                                       \
    do {                                                   \
        asm volatile(WAIT ::: "memory");                   \
        __builtin_amdgcn_s_barrier();                      \
        __builtin_amdgcn_sched_barrier(0);                 \
    } while (0)
#define P_FENCE() __builtin_amdgcn_sched_barrier(0)

    int t = 0, kt = 0;
    for (int S = 0; S < NS; ++S) {
        const int st = S & 1;
        const bool tile_end = (kt == nk - 1);                     // wave-uniform
        // ---- phase 0: row blocks 0,1 (fragments in fa[0]) ---------------------------------------------------------------------
        P_HEAD("s_waitcnt lgkmcnt(0)");
        issue_A(0);
        P_FENCE(); read_B(1, st);                                 // k-step 1 of B is used in this very phase: first thing behind the barrier
        P_FENCE(); mma(0, 0, 0, 0); P_FENCE();
        read_A(1, st, 1);
        P_FENCE(); mma(0, 0, 0, 1); mma(0, 0, 1, 0); mma(0, 0, 1, 1); P_FENCE();
        // ---- phase 1: row blocks 2,3 (fa[1]) ----------------------------------------------------------------------------------------
        P_HEAD("s_waitcnt lgkmcnt(0)");
        issue_A(1); issue_B(0); issue_B(1);
        P_FENCE(); mma(1, 1, 0, 0); P_FENCE();
        read_A(0, st, 2);
        P_FENCE(); mma(1, 1, 0, 1); mma(1, 1, 1, 0); mma(1, 1, 1, 1); P_FENCE();
        // ---- phase 2: row blocks 4,5 (fa[0]) ----------------------------------------------------------------------------------------
        P_HEAD("s_waitcnt vmcnt(14) lgkmcnt(0)");
        issue_A(2); issue_B(2); issue_B(3);
        P_FENCE(); mma(0, 2, 0, 0); P_FENCE();
        read_A(1, st, 3);
        P_FENCE(); mma(0, 2, 0, 1); mma(0, 2, 1, 0); mma(0, 2, 1, 1); P_FENCE();
        // ---- phase 3: row blocks 6,7 (fa[1]); the next slab's first fragments unless the tile ends here ----------------------------
        P_HEAD("s_waitcnt vmcnt(9) lgkmcnt(0)");
        issue_A(3); issue_N();
        advance2();
        P_FENCE(); mma(1, 3, 0, 0); mma(1, 3, 0, 1); P_FENCE();
        if (!tile_end) { read_A(0, st ^ 1, 0); read_B(0, st ^ 1); }
        P_FENCE(); mma(1, 3, 1, 0); mma(1, 3, 1, 1); P_FENCE();
        if (!tile_end) { ++kt; continue; }

        // ---- tile boundary: epilogue of tile t; the DMA stream keeps running ---------------------------------------------------
        {
            const long long col0 = b_first + (long long)t * H_TN + wc * 64 + c16;
            const float* s_bn = (const float*)(smem + OFF_BN + (t & (P_BN_BUFS - 1)) * H_TN * 4);
            float bnv[4], cpre[4];
            u32 colv[4];
            bool colok[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                const long long col = col0 + n * 16;
                bnv[n] = s_bn[wc * 64 + n * 16 + c16];             // brought by this wave's own DMA, covered by the phase-3 wait above
                colv[n] = (u32)col;
                colok[n] = col < p.nb_valid;
                const float c = PRE ? prescreen_col<METRIC>(bnv[n], p) : prefilter_col<METRIC>(bnv[n]);
                cpre[n] = colok[n] ? c : __builtin_inff();
            }
            int g_here = g, c_here = c16;                           // opaque per tile (see nwk_tile_kernel_f16s)
            asm volatile("" : "+v"(g_here));
            asm volatile("" : "+v"(c_here));
            u32 blocks = 0xFFu;
            if (PRE) blocks = prescreen_block_mask(acc, pmin, fminf(fminf(cpre[0], cpre[1]), fminf(cpre[2], cpre[3])));
            if (blocks)
                tile_epilogue16<METRIC, false, PRE, SC, AGG>(p, acc, (const float*)epi, (const u64*)(epi + H_TM * 4), (const float*)(epi + H_TM * 12), bnv,
                                                        colv, colok, cpre, q0, wr * 128, g_here, c_here, s_cnt, s_key, s_row, blocks);
        }
        ++t; kt = 0;
        if (S + 1 < NS) {
            zero_acc();
            read_A(0, st ^ 1, 0);                                  // published by the phase-3 barrier of this slab
            read_B(0, st ^ 1);
            // the staged survivor list is past half full (early, small chunks): empty it now.  Every wave sees the same count only
            // behind a barrier that follows the last append, so the check itself sits behind one.
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const u32 staged = __builtin_amdgcn_readfirstlane(*s_cnt);
            if (staged > (u32)(SC / 2)) {
                flush_survivors(p, s_cnt, s_key, s_row, q0, tid, 512, s_rc);
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if (tid == 0) *s_cnt = 0;
            }
        }
    }
#undef P_HEAD
#undef P_FENCE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // no LDS-DMA may still be in flight when the workgroup leaves the CU
    __syncthreads();
    flush_survivors(p, s_cnt, s_key, s_row, q0, tid, 512, s_rc);
}

// ------------------------------------------------------------------------------------------------
// launcher (host side, called from nwk_launch_tile_prescreen / nwk_launch_tile_f16 in nwk_tile_f16.hip)
// ------------------------------------------------------------------------------------------------
extern "C++" {

hipError_t nwk_launch_tile_f16p(hipStream_t st, const NwkTileParams& p, int metric, int pre, unsigned grid)
{
#define NWK_P(M)                                                                                                  \
    do {                                                                                                           \
        if (pre && p.agg) hipLaunchKernelGGL((nwk_tile_kernel_f16p<M, 1, 1>), dim3(grid), dim3(512), 0, st, p);    \
        else if (pre) hipLaunchKernelGGL((nwk_tile_kernel_f16p<M, 1, 0>), dim3(grid), dim3(512), 0, st, p);        \
        else hipLaunchKernelGGL((nwk_tile_kernel_f16p<M, 0, 0>), dim3(grid), dim3(512), 0, st, p);                 \
    } while (0)
    if (metric == NWKNN_METRIC_SQEUCLIDEAN) NWK_P(NWKNN_METRIC_SQEUCLIDEAN);
    else if (metric == NWKNN_METRIC_ONE_MINUS_DOT) NWK_P(NWKNN_METRIC_ONE_MINUS_DOT);
    else if (metric == NWKNN_METRIC_COSINE) NWK_P(NWKNN_METRIC_COSINE);
    else return hipErrorInvalidValue;
#undef NWK_P
    return hipGetLastError();
}

}  // extern "C++"
