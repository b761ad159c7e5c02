// nwk_tile_f16.hip -- fp16 distance tile + filter kernels (K1h): 32x32x16, 16x16x32 and the two-workgroup form; PRE = 1 is the pre-screen of the exact search  (gfx950 only; shared pieces and the overall design: nwk_device.h)
#include "nwk_device.h"
#include "nwk_f16_common.h"

// ------------------------------------------------------------------------------------------------
// fp16 distance tile: 256 queries x 256 base rows per workgroup of 8 waves (2 x 4; a wave owns 128 x 64 = 4 x 2
// accumulators of v_mfma_f32_32x32x16_f16, fp32 accumulate), K in 64-half (128-B) slabs through two 64 KiB LDS stages.
// fp16 MFMA is 16x the fp32 rate, so the operand stream is the problem: per DMA instruction this shape has
// 256*256/(256+256) = 128 cycles of MFMA work, the most a 160 KiB LDS allows with 2 stages.  Same LDS image, swizzle,
// DMA form, epilogue and candidate-list logic as the fp32 kernel.  The accumulation order inside one MFMA (16 k's) is
// not specified by the ISA, so this path is checked against the float64 oracle with a tolerance, not bit for bit.
// ------------------------------------------------------------------------------------------------

template <int METRIC, bool DENSE, int DIAG = 0, int PRE = 0>
__global__ __launch_bounds__(512, 2) void nwk_tile_kernel_f16(const NwkTileParams p)
{
    __shared__ __attribute__((aligned(16))) char smem[2 * H_STAGE + H_TM * 16];
    char* const epi = smem + 2 * H_STAGE;     // query norms | thresholds | pre-filter terms of this tile, written once up front

    unsigned long long st0 = 0, st1 = 0, st2 = 0, rt0 = 0;
    if (p.dbg) { st0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    const TileCoord tc = tile_from_block(p);
    if (!tc.ok) return;

    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63;
    const int wr = w >> 2, wc = w & 3;
    const int h = l >> 5, c32 = l & 31;
    const long long q0 = (long long)tc.qt * H_TM;
    const long long b0 = (long long)tc.bt * H_TN;
    const int dp = p.dp;                          // halves per prepared row
    const int nk = dp / H_BK;
    const _Float16* qp = (const _Float16*)p.qp;
    const _Float16* bp = (const _Float16*)p.bp;

    // DMA: 32 wave-instructions per operand per slab (8 rows each); wave w issues rows [32w, 32w+32) of both tiles
    unsigned voff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = (w * 4 + i) * 8 + (l >> 3);
        voff[i] = (unsigned)(row * dp * 2 + (((l & 7) ^ ((row >> 1) & 7)) << 4));
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(qp + q0 * dp), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(bp + b0 * dp), 0, 0x7FFFFFFF, 0x00020000);
    const int lds_w = __builtin_amdgcn_readfirstlane(w * 4096);
    auto issue_stage = [&](int stage, int kt) {
        char* base = smem + stage * H_STAGE + lds_w;
        const int soff = kt * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)(base + i * 1024), 16, voff[i], soff, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)(base + 32768 + i * 1024), 16, voff[i], soff, 0, 0);
    };

    int offA[4], offB[2];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
        const int ra = wr * 128 + m * 32 + c32;
        offA[m] = ra * 128 + ((h ^ ((ra >> 1) & 7)) << 4);
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int rb = wc * 64 + n * 32 + c32;
        offB[n] = 32768 + rb * 128 + ((h ^ ((rb >> 1) & 7)) << 4);
    }

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
    f32x16 acc[4][2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;
    // query-side epilogue inputs: the loads go out here, the first two slabs' DMA right behind them (below), and only then
    // are the values used and staged in their own LDS region -- the first barrier of the K loop publishes them
    float my_qn = 0.f;
    u64 my_tau = ~0ull;
    if (tid < H_TM) {
        my_qn = p.qn[q0 + tid];
        if (!DENSE) my_tau = p.tau[q0 + tid];
    }

    // Software pipeline: fragment registers are double buffered (set S feeds the MFMAs of k-step s while the ds_reads of
    // k-step s+1 land in set S^1), and there is ONE barrier per slab, placed before the slab's last k-step: by then this
    // wave has completed every LDS read of slab kt and slab kt+1's DMA (issued a slab earlier) has landed, so the same
    // barrier publishes slab kt+1 and releases slab kt's buffer; the DMA of slab kt+2 and the first fragment reads of
    // slab kt+1 go out right behind it, under the 8 MFMAs of the last k-step.
    f16x8 fa[2][4], fb[2][2];
    auto read_frags = [&](int set, int stage, int ks) {
        const char* sb = smem + stage * H_STAGE;
#pragma unroll
        for (int m = 0; m < 4; ++m) fa[set][m] = *(const f16x8*)(sb + (offA[m] ^ (ks << 5)));
#pragma unroll
        for (int n = 0; n < 2; ++n) fb[set][n] = *(const f16x8*)(sb + (offB[n] ^ (ks << 5)));
    };
    auto mma = [&](int set, int m0, int m1) {
#pragma unroll
        for (int m = m0; m < m1; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n)
                acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[set][m], fb[set][n], acc[m][n], 0, 0, 0);
    };
#define NWK_HSTEP(SET, NEXT_STAGE, NEXT_KS)              \
    do {                                                 \
        __builtin_amdgcn_sched_barrier(0);               \
        mma(SET, 0, 1);                                  \
        __builtin_amdgcn_sched_barrier(0);               \
        if (!(DIAG & 1)) read_frags((SET) ^ 1, NEXT_STAGE, NEXT_KS); \
        __builtin_amdgcn_sched_barrier(0);               \
        mma(SET, 1, 4);                                  \
        __builtin_amdgcn_sched_barrier(0);               \
    } while (0)

    __builtin_amdgcn_sched_barrier(0);
    issue_stage(0, 0);
    if (nk > 1) issue_stage(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    stage_query_side<METRIC, DENSE, PRE>(epi, H_TM, tid, my_qn, my_tau, p, q0);
    if (nk > 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");       // slab 0 landed; my epilogue-input ds_writes are done
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (p.dbg) st1 = __builtin_amdgcn_s_memtime();
    read_frags(0, 0, 0);
    if (DIAG & 1) read_frags(1, 0, 1);

    for (int kt = 0; kt + 1 < nk; ++kt) {
        const int cur = kt & 1;
        NWK_HSTEP(0, cur, 1);
        NWK_HSTEP(1, cur, 2);
        NWK_HSTEP(0, cur, 3);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        if (!(DIAG & 4)) __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk && !(DIAG & 2)) issue_stage(cur, kt + 2);
        NWK_HSTEP(1, cur ^ 1, 0);
    }
    {
        const int cur = (nk - 1) & 1;
        NWK_HSTEP(0, cur, 1);
        NWK_HSTEP(1, cur, 2);
        NWK_HSTEP(0, cur, 3);
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 0, 4);
    }
#undef NWK_HSTEP

    if (p.dbg) st2 = __builtin_amdgcn_s_memtime();
    if (!(DIAG & 16))
        tile_epilogue<METRIC, DENSE, 4, PRE>(p, acc, (const float*)epi, (const u64*)(epi + H_TM * 4), (const float*)(epi + H_TM * 12), bnv, colv,
                                        colok, q0, wr * 128, h, c32);
    else {                                                   // DIAG 16: no epilogue (keep the accumulators alive)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) asm volatile("" :: "v"(acc[m][n]));
    }
    if (p.dbg && tid == 0) {
        unsigned long long* o = p.dbg + (size_t)blockIdx.x * 8;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_amdgcn_s_memtime(); o[4] = rt0; o[5] = __builtin_amdgcn_s_memrealtime();
        o[6] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)); o[7] = 1;
    }
}

// ------------------------------------------------------------------------------------------------
// fp16 distance tile, v_mfma_f32_16x16x32_f16 form.  Same workgroup tile (256 x 256, 8 waves 2 x 4, a wave owns 128 x 64),
// same LDS image, DMA and barrier scheme as nwk_tile_kernel_f16; the wave's tile is 8 x 4 accumulators of 16 x 16.  Where the
// chip holds its clock down under fp16 MFMA load this shape sustains a higher clock at the same cycles per FLOP
// (MI355X_MICROARCH.md, DVFS give-back item 7).  Per 32-deep k-step a wave reads 8 A + 4 B fragments (ds_read_b128: lane l
// takes row l%16, 16-B chunk l/16 -- conflict-free on the existing swizzle) for 32 MFMAs, i.e. the same LDS bytes per FLOP.
// Software pipeline in four phases per 64-half slab, 16 MFMAs each: the A fragments of rows 0-63 / 64-127 of the wave's tile
// alternate between two register buffers, the B fragments of the two k-steps between two sets; one barrier per slab before
// its last phase, as in the 32x32 kernel.
// Accumulator layout: lane l holds D[4*(l/16) + i][l%16], i = 0..3  ->  column (base row) = l%16, four consecutive query rows.
// ------------------------------------------------------------------------------------------------
template <int METRIC, bool DENSE, int PRE = 0>
__global__ __launch_bounds__(512, 2) void nwk_tile_kernel_f16s(const NwkTileParams p)
{
    constexpr int SC = DENSE ? 0 : H_SC;                        // survivor staging (dense chunks list every key in its own slot)
    __shared__ __attribute__((aligned(16))) char smem[2 * H_STAGE + H_TM * 16 + SC * 12 + 16];
    char* const epi = smem + 2 * H_STAGE;
    LDS_AS u64* const s_key = (LDS_AS u64*)(smem + 2 * H_STAGE + H_TM * 16);
    LDS_AS u32* const s_row = (LDS_AS u32*)(smem + 2 * H_STAGE + H_TM * 16 + SC * 8);
    LDS_AS u32* const s_cnt = (LDS_AS u32*)(smem + 2 * H_STAGE + H_TM * 16 + SC * 12);

    unsigned long long st0 = 0, st1 = 0, st2 = 0, rt0 = 0;       // dev stamps (p.dbg): see tools/stamps2.py
    if (p.dbg) { st0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    const TileCoord tc = tile_from_block(p);
    if (!tc.ok) return;

    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63;
    const int wr = w >> 2, wc = w & 3;
    const int g = l >> 4, c16 = l & 15;
    const long long q0 = (long long)tc.qt * H_TM;
    // A workgroup walks up to p.tiles_per_wg consecutive base tiles of its query tile ("strip"): the next tile's first two slabs are
    // requested before the current tile's epilogue, so from the second tile on the DMA round trip, the query-side staging and the
    // dispatch gap (~4k of a d=1024 tile's ~59k cycles) are hidden or gone.  tc.bt numbers the strip, not the tile.
    const int T = p.tiles_per_wg > 1 ? p.tiles_per_wg : 1;
    const int bt_first = p.bt0 + (tc.bt - p.bt0) * T;
    int n_here = p.bt0 + p.n_btiles_real - bt_first;
    if (n_here > T) n_here = T;
    long long b0 = (long long)bt_first * H_TN;
    const int dp = p.dp;
    const int nk = dp / H_BK;
    const _Float16* qp = (const _Float16*)p.qp;
    const _Float16* bp = (const _Float16*)p.bp;

    // DMA: a wave-instruction brings 8 rows x 128 B; wave w owns rows [32w, 32w+32) of both operand tiles.  The swizzle term
    // ((row >> 1) & 7) of row = 32w + 8i + l/8 is ((i & 1) * 4 + l/16) & 7: two per-lane offsets (i even / odd) serve all four
    // instructions, the row origins go into the scalar offset next to the slab position.
    unsigned voff2[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) voff2[i] = (unsigned)((l >> 3) * dp * 2 + (((l & 7) ^ ((i * 4 + (l >> 4)) & 7)) << 4));
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int row_org = wu * 32 * dp * 2, row_step = 8 * dp * 2;                 // bytes (SGPRs)
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(qp + q0 * dp), 0, 0x7FFFFFFF, 0x00020000);
    __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(bp + b0 * dp), 0, 0x7FFFFFFF, 0x00020000);
    const int lds_w = wu * 4096;
    auto issue_stage = [&](int stage, int kt) {
        char* base = smem + stage * H_STAGE + lds_w;
        const int soff = kt * 128 + row_org;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)(base + i * 1024), 16, voff2[i & 1], soff + i * row_step, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)(base + 32768 + i * 1024), 16, voff2[i & 1], soff + i * row_step, 0, 0);
    };

    // fragment addresses: row = block*16 + l%16, chunk = 4*ks + l/16, slot = chunk ^ ((row >> 1) & 7); block*16 and the
    // wave's row origin are multiples of 16, so the swizzle term depends on l%16 only and blocks are +2048-byte immediates
    const int swz = (c16 >> 1) & 7;
    const int offA0 = (wr * 128 + c16) * 128 + ((g ^ swz) << 4);
    const int offB0 = 32768 + (wc * 64 + c16) * 128 + ((g ^ swz) << 4);

    // column-side epilogue inputs are (re)computed where they are used, so that nothing but the accumulators, the fragment
    // buffers and a few addresses lives across the K loop (256 VGPRs at two waves per SIMD)
    f32x4 acc[8][4];
    float my_qn = 0.f;                                   // loads out now, used after the first DMAs have been issued
    u64 my_tau = ~0ull;
    if (tid < H_TM) {
        my_qn = p.qn[q0 + tid];
        if (!DENSE) my_tau = p.tau[q0 + tid];
    }

    f16x8 fa[2][4], fb[2][4];
    auto read_A = [&](int buf, int stage, int ks, int mh) {
        const char* sb = smem + stage * H_STAGE + (offA0 ^ (ks << 6)) + mh * 4 * 2048;
#pragma unroll
        for (int j = 0; j < 4; ++j) fa[buf][j] = *(const f16x8*)(sb + j * 2048);
    };
    auto read_B = [&](int set, int stage, int ks) {
        const char* sb = smem + stage * H_STAGE + (offB0 ^ (ks << 6));
#pragma unroll
        for (int n = 0; n < 4; ++n) fb[set][n] = *(const f16x8*)(sb + n * 2048);
    };
    auto mma = [&](int abuf, int bset, int mh, int j0, int j1) {
#pragma unroll
        for (int j = j0; j < j1; ++j)
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[mh * 4 + j][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[abuf][j], fb[bset][n], acc[mh * 4 + j][n], 0, 0, 0);
    };
#define NWK_SPHASE(ABUF, BSET, MH, LOADS)                \
    do {                                                 \
        __builtin_amdgcn_sched_barrier(0);               \
        mma(ABUF, BSET, MH, 0, 1);                       \
        __builtin_amdgcn_sched_barrier(0);               \
        LOADS;                                           \
        __builtin_amdgcn_sched_barrier(0);               \
        mma(ABUF, BSET, MH, 1, 4);                       \
        __builtin_amdgcn_sched_barrier(0);               \
    } while (0)

    __builtin_amdgcn_sched_barrier(0);
    issue_stage(0, 0);
    if (nk > 1) issue_stage(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    stage_query_side<METRIC, DENSE, PRE>(epi, H_TM, tid, my_qn, my_tau, p, q0);
    if (SC > 0 && tid == 0) *s_cnt = 0;                  // published by the first barrier of the K loop, long before any append
    int par = 0;                                         // slab kt of the current tile lives in stage (kt & 1) ^ par
  for (int t = 0; t < n_here; ++t) {
    const long long col0 = b0 + wc * 64 + c16;
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[m][n][i] = 0.0f;
    // first tile: only the two slab requests are outstanding; later tiles: the previous epilogue's stores follow them in the counter
    if (t == 0 && nk > 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (p.dbg && t == 0) st1 = __builtin_amdgcn_s_memtime();
    if (SC > 0 && t > 0) {                               // the staged list is past half full (early, small chunks): empty it now
        const u32 staged = __builtin_amdgcn_readfirstlane(*s_cnt);
        if (staged > (u32)(SC / 2)) {
            flush_survivors(p, s_cnt, s_key, s_row, q0, tid, 512);
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (tid == 0) *s_cnt = 0;                    // the K loop's barriers order this before the next append
        }
    }
    read_A(0, par, 0, 0);
    read_B(0, par, 0);

    for (int kt = 0; kt + 1 < nk; ++kt) {
        const int cur = (kt & 1) ^ par;
        NWK_SPHASE(0, 0, 0, read_A(1, cur, 0, 1));
        NWK_SPHASE(1, 0, 1, read_A(0, cur, 1, 0); read_B(1, cur, 1));
        NWK_SPHASE(0, 1, 0, read_A(1, cur, 1, 1));
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (kt + 2 < nk) issue_stage(cur, kt + 2);
        NWK_SPHASE(1, 1, 1, read_A(0, cur ^ 1, 0, 0); read_B(0, cur ^ 1, 0));
    }
    float bnv[4];                                        // column norms: requested under the last slab's MFMAs
#pragma unroll
    for (int n = 0; n < 4; ++n) bnv[n] = p.bn[col0 + n * 16];
    {
        const int cur = ((nk - 1) & 1) ^ par;
        NWK_SPHASE(0, 0, 0, read_A(1, cur, 0, 1));
        NWK_SPHASE(1, 0, 1, read_A(0, cur, 1, 0); read_B(1, cur, 1));
        NWK_SPHASE(0, 1, 0, read_A(1, cur, 1, 1));
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 1, 1, 0, 4);
    }
    if (p.dbg) st2 = __builtin_amdgcn_s_memtime();
    if (t + 1 < n_here) {                                // wave-uniform: request the next tile's first slabs before this tile's epilogue
        __builtin_amdgcn_s_barrier();                    // every wave has read its last fragments: both stages are free
        b0 += H_TN;
        rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(bp + b0 * dp), 0, 0x7FFFFFFF, 0x00020000);
        par ^= nk & 1;
        __builtin_amdgcn_sched_barrier(0);
        issue_stage(par, 0);
        if (nk > 1) issue_stage(par ^ 1, 1);
        __builtin_amdgcn_sched_barrier(0);
    }

    float cpre[4];
    u32 colv[4];
    bool colok[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const long long col = col0 + n * 16;
        colv[n] = (u32)col;
        colok[n] = col < p.nb_valid;
        const float c = PRE ? prescreen_col<METRIC>(bnv[n], p) : prefilter_col<METRIC>(bnv[n]);
        cpre[n] = colok[n] ? c : __builtin_inff();
    }
    int g_here = g;                                      // opaque per tile: otherwise the epilogue's 32 list-row addresses are hoisted
    asm volatile("" : "+v"(g_here));                     // out of the strip loop and live (spilled) through every K loop
    int c_here = c16;
    asm volatile("" : "+v"(c_here));
    tile_epilogue16<METRIC, DENSE, PRE, SC>(p, acc, (const float*)epi, (const u64*)(epi + H_TM * 4), (const float*)(epi + H_TM * 12), bnv, colv,
                                            colok, cpre, q0, wr * 128, g_here, c_here, s_cnt, s_key, s_row);
  }
#undef NWK_SPHASE
    if (SC > 0) {                                        // end of the strip: the staged survivors go to the queries' lists
        __syncthreads();
        flush_survivors(p, s_cnt, s_key, s_row, q0, tid, 512);
    }
    if (p.dbg && l == 0) {          // wave 0 writes its own stamps; every wave folds its end times in (the tile ends with its last wave)
        unsigned long long* o = p.dbg + (size_t)blockIdx.x * 8;
        const unsigned long long t3 = __builtin_amdgcn_s_memtime(), r3 = __builtin_amdgcn_s_memrealtime();
        if (w == 0) {
            o[0] = st0; o[1] = st1; o[2] = st2; o[4] = rt0;
            o[6] = (unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32      // HW_REG_XCC_ID[3:0]
                 | (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));           // HW_REG_HW_ID
            o[7] = 1;
        }
        atomicMax(o + 3, t3);
        atomicMax(o + 5, r3);
        unsigned long long* o2 = p.dbg + (size_t)gridDim.x * 8 + (size_t)blockIdx.x * 16 + w * 2;     // per wave: end of its K loop, end of its epilogue
        o2[0] = st2; o2[1] = t3;
    }
}

// ------------------------------------------------------------------------------------------------
// fp16 distance tile, "two workgroups per CU" form (option f16_mfma = 2).  The 8-wave kernels above keep ONE workgroup per CU:
// its prologue (first DMA round trip), its epilogue and the dispatch gap to the next workgroup leave the MFMA pipes idle for
// ~25 % of a d=1024 tile (tools/stamps2.py), and both waves of a SIMD stall at the same barriers.  Here a workgroup is 4 waves
// (2 x 2, a wave still owns 128 x 64 = 8 x 4 accumulators of v_mfma_f32_16x16x32_f16) on a 256 x 128 tile with 32-half slabs
// (64-B LDS rows, one k-step per slab) in a THREE-stage ring of 24 KiB: 76 KiB per workgroup, so two INDEPENDENT workgroups
// share a CU, one wave of each per SIMD, and one's barriers / prologue / epilogue overlap the other's MFMAs.
//   LDS image: row r, 16-B chunk c at slot c ^ f(r), f(r) = (-(r >> 2)) & 3: ds_read_b128 by (row l%16, chunk l/16) is
//   conflict-free (each 16-lane service group covers all 64 banks once).
//   Ring: the DMA of slab t+3 goes into the stage slab t is leaving, right behind the barrier in the middle of slab t (every wave
//   has then finished reading it); that barrier also publishes slab t+1, which was issued two slabs earlier (vmcnt(6): 6 DMA
//   instructions per wave and slab: 4 x A, 2 x B).
// ------------------------------------------------------------------------------------------------
#define Q_TM 256
#define Q_TN 128
#define Q_BK 32
#define Q_STAGE (Q_TM * 64 + Q_TN * 64)        // 24 KiB
#define Q_NST 3

template <int METRIC, bool DENSE, int PRE = 0>
__global__ __launch_bounds__(256, 2) void nwk_tile_kernel_f16q(const NwkTileParams p)
{
    __shared__ __attribute__((aligned(16))) char smem[Q_NST * Q_STAGE + Q_TM * 16];
    char* const epi = smem + Q_NST * Q_STAGE;

    const TileCoord tc = tile_from_block(p);
    if (!tc.ok) return;

    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63;
    const int wr = w >> 1, wc = w & 1;
    const int g = l >> 4, c16 = l & 15;
    const long long q0 = (long long)tc.qt * Q_TM;
    const long long b0 = (long long)tc.bt * Q_TN;
    const int dp = p.dp;
    const int nk = dp / Q_BK;                                // even: dp is a multiple of 64
    const _Float16* qp = (const _Float16*)p.qp;
    const _Float16* bp = (const _Float16*)p.bp;

    // DMA: one wave-instruction = 16 rows x 64 B (4 lanes per row).  Wave w brings A rows [64w, 64w+64) and B rows [32w, 32w+32).
    // The swizzle term f(row) depends on (row >> 2) & 3 = (l >> 4) & 3 only (row origins are multiples of 16), so ONE per-lane
    // offset serves every instruction and the row origin goes into the scalar offset with the slab position.
    const unsigned voff = (unsigned)((l >> 2) * dp * 2 + (((l & 3) ^ ((0 - (l >> 4)) & 3)) << 4));
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(qp + q0 * dp), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(bp + b0 * dp), 0, 0x7FFFFFFF, 0x00020000);
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int lds_wA = wu * 4096;
    const int lds_wB = Q_TM * 64 + wu * 2048;
    const int rowsA = wu * 64 * dp * 2, rowsB = wu * 32 * dp * 2, step16 = 16 * dp * 2;      // byte offsets of row origins (SGPRs)
    auto issue_slab = [&](int kt) {
        char* st = smem + (kt % Q_NST) * Q_STAGE;
        const int soff = kt * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)(st + lds_wA + i * 1024), 16, voff, soff + rowsA + i * step16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)(st + lds_wB + i * 1024), 16, voff, soff + rowsB + i * step16, 0, 0);
    };

    const int swz = (0 - (c16 >> 2)) & 3;
    const int offA0 = (wr * 128 + c16) * 64 + ((g ^ swz) << 4);
    const int offB0 = Q_TM * 64 + (wc * 64 + c16) * 64 + ((g ^ swz) << 4);
    const long long col0 = b0 + wc * 64 + c16;

    f32x4 acc[8][4];
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int i = 0; i < 4; ++i) acc[m][n][i] = 0.0f;

    float my_qn = p.qn[q0 + tid];                            // 256 threads = 256 query rows
    u64 my_tau = ~0ull;
    if (!DENSE) my_tau = p.tau[q0 + tid];

    f16x8 fa[2][4], fb[2][4];
    auto read_A = [&](int buf, int kt, int mh) {
        const char* sb = smem + (kt % Q_NST) * Q_STAGE + offA0 + mh * 4 * 1024;
#pragma unroll
        for (int j = 0; j < 4; ++j) fa[buf][j] = *(const f16x8*)(sb + j * 1024);
    };
    auto read_B = [&](int set, int kt) {
        const char* sb = smem + (kt % Q_NST) * Q_STAGE + offB0;
#pragma unroll
        for (int n = 0; n < 4; ++n) fb[set][n] = *(const f16x8*)(sb + n * 1024);
    };
    auto mma = [&](int abuf, int bset, int mh, int j0, int j1) {
#pragma unroll
        for (int j = j0; j < j1; ++j)
#pragma unroll
            for (int n = 0; n < 4; ++n)
                acc[mh * 4 + j][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[abuf][j], fb[bset][n], acc[mh * 4 + j][n], 0, 0, 0);
    };
#define NWK_QPHASE(ABUF, BSET, MH, LOADS)                \
    do {                                                 \
        __builtin_amdgcn_sched_barrier(0);               \
        mma(ABUF, BSET, MH, 0, 1);                       \
        __builtin_amdgcn_sched_barrier(0);               \
        LOADS;                                           \
        __builtin_amdgcn_sched_barrier(0);               \
        mma(ABUF, BSET, MH, 1, 4);                       \
        __builtin_amdgcn_sched_barrier(0);               \
    } while (0)
    // one slab: rows 0-63 of the wave's tile, barrier + restage, rows 64-127 (during which the next slab's first fragments load).
    // No MFMA sits inside a runtime branch (that would make the compiler copy accumulators): the last slab is peeled.
#define NWK_QSLAB(KT, BSET)                                                                          \
    do {                                                                                             \
        NWK_QPHASE(0, BSET, 0, read_A(1, (KT), 1));                                                  \
        if ((KT) + 2 < nk) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");               \
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                             \
        __builtin_amdgcn_s_barrier();                                                                \
        if ((KT) + 3 < nk) issue_slab((KT) + 3);                                                     \
        NWK_QPHASE(1, BSET, 1, read_A(0, (KT) + 1, 0); read_B((BSET) ^ 1, (KT) + 1));                \
    } while (0)

    __builtin_amdgcn_sched_barrier(0);
    issue_slab(0);
    issue_slab(1);                                           // nk >= 2
    if (nk > 2) issue_slab(2);
    __builtin_amdgcn_sched_barrier(0);
    stage_query_side<METRIC, DENSE, PRE>(epi, Q_TM, tid, my_qn, my_tau, p, q0);
    if (nk > 2) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");     // slab 0 landed; my epilogue-input ds_writes are done
    else asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    read_A(0, 0, 0);
    read_B(0, 0);

    for (int kt = 0; kt + 2 < nk; kt += 2) {
        NWK_QSLAB(kt, 0);
        NWK_QSLAB(kt + 1, 1);
    }
    NWK_QSLAB(nk - 2, 0);
    {                                                        // last slab (odd index: B set 1), nothing left to stage or prefetch
        NWK_QPHASE(0, 1, 0, read_A(1, nk - 1, 1));
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 1, 1, 0, 4);
    }
#undef NWK_QSLAB
#undef NWK_QPHASE

    float bnv[4], cpre[4];
    u32 colv[4];
    bool colok[4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const long long col = col0 + n * 16;
        colv[n] = (u32)col;
        colok[n] = col < p.nb_valid;
        bnv[n] = p.bn[col];
    }
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        const float c = PRE ? prescreen_col<METRIC>(bnv[n], p) : prefilter_col<METRIC>(bnv[n]);
        cpre[n] = colok[n] ? c : __builtin_inff();
    }
    tile_epilogue16<METRIC, DENSE, PRE>(p, acc, (const float*)epi, (const u64*)(epi + Q_TM * 4), (const float*)(epi + Q_TM * 12), bnv, colv,
                                        colok, cpre, q0, wr * 128, g, c16);
}


// ------------------------------------------------------------------------------------------------
// launchers (host side, called from nwknn_api.cpp)
// ------------------------------------------------------------------------------------------------
extern "C++" {

template <int METRIC>
static hipError_t launch_tile_f16_metric(hipStream_t st, const NwkTileParams& p, bool dense, unsigned grid, int diag)
{
    if (p.f16_shape == 2 && diag == 0) {                   // 4-wave workgroups on 256 x 128 tiles, two per CU
        if (dense) hipLaunchKernelGGL((nwk_tile_kernel_f16q<METRIC, true, 0>), dim3(grid), dim3(256), 0, st, p);
        else hipLaunchKernelGGL((nwk_tile_kernel_f16q<METRIC, false, 0>), dim3(grid), dim3(256), 0, st, p);
        return hipGetLastError();
    }
    if (p.f16_shape == 1 && diag == 0) {                   // v_mfma_f32_16x16x32_f16 form (the caller has set up strips)
        if (dense) hipLaunchKernelGGL((nwk_tile_kernel_f16s<METRIC, true, 0>), dim3(grid), dim3(512), 0, st, p);
        else hipLaunchKernelGGL((nwk_tile_kernel_f16s<METRIC, false, 0>), dim3(grid), dim3(512), 0, st, p);
        return hipGetLastError();
    }
    if (dense) { hipLaunchKernelGGL((nwk_tile_kernel_f16<METRIC, true, 0>), dim3(grid), dim3(512), 0, st, p); return hipGetLastError(); }
    switch (diag) {              // dev-only timing ablations (wrong results by construction)
    case 1: hipLaunchKernelGGL((nwk_tile_kernel_f16<METRIC, false, 1>), dim3(grid), dim3(512), 0, st, p); break;
    case 2: hipLaunchKernelGGL((nwk_tile_kernel_f16<METRIC, false, 2>), dim3(grid), dim3(512), 0, st, p); break;
    case 4: hipLaunchKernelGGL((nwk_tile_kernel_f16<METRIC, false, 4>), dim3(grid), dim3(512), 0, st, p); break;
    case 7: hipLaunchKernelGGL((nwk_tile_kernel_f16<METRIC, false, 7>), dim3(grid), dim3(512), 0, st, p); break;
    case 16: hipLaunchKernelGGL((nwk_tile_kernel_f16<METRIC, false, 16>), dim3(grid), dim3(512), 0, st, p); break;
    case 23: hipLaunchKernelGGL((nwk_tile_kernel_f16<METRIC, false, 23>), dim3(grid), dim3(512), 0, st, p); break;
    default: hipLaunchKernelGGL((nwk_tile_kernel_f16<METRIC, false, 0>), dim3(grid), dim3(512), 0, st, p); break;
    }
    return hipGetLastError();
}

// fp16 path: p.n_qtiles counts 256-row query tiles; p.n_btiles / p.bt0 count base tiles of 256 rows (f16_shape 0, 1) or 128 rows
// (f16_shape 2); p.qp/p.bp point at halves.
// the 16x16x32 kernel walks strips of p.tiles_per_wg base tiles per workgroup: p.n_btiles becomes the number of strips
static void setup_strips(NwkTileParams& p, bool dense, int diag)
{
    p.n_btiles_real = p.n_btiles;
    if ((p.f16_shape == 1 || p.f16_shape == 3) && diag == 0 && !dense && p.tiles_per_wg > 1) p.n_btiles = (p.n_btiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
    else p.tiles_per_wg = 1;
}

hipError_t nwk_launch_tile_f16(hipStream_t st, NwkTileParams p, int metric, bool dense, int diag)
{
    if (p.n_qtiles <= 0 || p.n_btiles <= 0) return hipSuccess;
    if (p.sq_shift < 0 || p.sq_shift > 6) return hipErrorInvalidValue;
    if (p.f16_shape == 3 && (dense || diag != 0)) p.f16_shape = 1;               // the streaming form has no dense / ablation builds
    setup_strips(p, dense, diag);
    const int SQ = 1 << p.sq_shift, SB = 64 >> p.sq_shift;
    const long long n_super_q = (p.n_qtiles + SQ - 1) / SQ;
    p.n_super_b = (p.n_btiles + SB - 1) / SB;
    long long supers = n_super_q * p.n_super_b;
    supers = (supers + 7) / 8 * 8;
    const long long grid = supers * 64;
    if (grid > 0x7FFFFFFFll) return hipErrorInvalidValue;
    if (p.f16_shape == 3) return nwk_launch_tile_f16p(st, p, metric, 0, (unsigned)grid);
    switch (metric) {
    case NWKNN_METRIC_SQEUCLIDEAN: return launch_tile_f16_metric<NWKNN_METRIC_SQEUCLIDEAN>(st, p, dense, (unsigned)grid, diag);
    case NWKNN_METRIC_ONE_MINUS_DOT: return launch_tile_f16_metric<NWKNN_METRIC_ONE_MINUS_DOT>(st, p, dense, (unsigned)grid, diag);
    case NWKNN_METRIC_COSINE: return launch_tile_f16_metric<NWKNN_METRIC_COSINE>(st, p, dense, (unsigned)grid, diag);
    default: return hipErrorInvalidValue;
    }
}

// pre-screen launch of the fp16 tile kernel: p.qp/p.bp are the scaled fp16 copies, tiles are 256 x 256, never dense
hipError_t nwk_launch_tile_prescreen(hipStream_t st, NwkTileParams p, int metric, bool dense)
{
    if (p.n_qtiles <= 0 || p.n_btiles <= 0) return hipSuccess;
    if (p.sq_shift < 0 || p.sq_shift > 6) return hipErrorInvalidValue;
    if (metric == NWKNN_METRIC_COSINE && p.f16_shape == 0) p.f16_shape = 1;       // cosine has 16x16x32 builds only
    if (dense) p.f16_shape = 1;                                                    // dense + estimate keys (exact fp16 mode): 16x16x32 build
    setup_strips(p, dense, 0);
    const int SQ = 1 << p.sq_shift, SB = 64 >> p.sq_shift;
    const long long n_super_q = (p.n_qtiles + SQ - 1) / SQ;
    p.n_super_b = (p.n_btiles + SB - 1) / SB;
    long long supers = n_super_q * p.n_super_b;
    supers = (supers + 7) / 8 * 8;
    const long long grid = supers * 64;
    if (grid > 0x7FFFFFFFll) return hipErrorInvalidValue;
    if (!dense && p.f16_shape == 3) return nwk_launch_tile_f16p(st, p, metric, 1, (unsigned)grid);
    if (dense) {
        if (metric == NWKNN_METRIC_SQEUCLIDEAN)
            hipLaunchKernelGGL((nwk_tile_kernel_f16s<NWKNN_METRIC_SQEUCLIDEAN, true, 1>), dim3((unsigned)grid), dim3(512), 0, st, p);
        else if (metric == NWKNN_METRIC_ONE_MINUS_DOT)
            hipLaunchKernelGGL((nwk_tile_kernel_f16s<NWKNN_METRIC_ONE_MINUS_DOT, true, 1>), dim3((unsigned)grid), dim3(512), 0, st, p);
        else if (metric == NWKNN_METRIC_COSINE)
            hipLaunchKernelGGL((nwk_tile_kernel_f16s<NWKNN_METRIC_COSINE, true, 1>), dim3((unsigned)grid), dim3(512), 0, st, p);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (p.f16_shape == 2) {
        if (metric == NWKNN_METRIC_SQEUCLIDEAN)
            hipLaunchKernelGGL((nwk_tile_kernel_f16q<NWKNN_METRIC_SQEUCLIDEAN, false, 1>), dim3((unsigned)grid), dim3(256), 0, st, p);
        else if (metric == NWKNN_METRIC_ONE_MINUS_DOT)
            hipLaunchKernelGGL((nwk_tile_kernel_f16q<NWKNN_METRIC_ONE_MINUS_DOT, false, 1>), dim3((unsigned)grid), dim3(256), 0, st, p);
        else if (metric == NWKNN_METRIC_COSINE)
            hipLaunchKernelGGL((nwk_tile_kernel_f16q<NWKNN_METRIC_COSINE, false, 1>), dim3((unsigned)grid), dim3(256), 0, st, p);
        else return hipErrorInvalidValue;
        return hipGetLastError();
    }
    if (metric == NWKNN_METRIC_COSINE) {                   // cosine: 16x16x32 kernel only (f16_mfma 0 falls back to it)
        hipLaunchKernelGGL((nwk_tile_kernel_f16s<NWKNN_METRIC_COSINE, false, 1>), dim3((unsigned)grid), dim3(512), 0, st, p);
        return hipGetLastError();
    }
    if (metric == NWKNN_METRIC_SQEUCLIDEAN) {
        if (p.f16_shape == 1) hipLaunchKernelGGL((nwk_tile_kernel_f16s<NWKNN_METRIC_SQEUCLIDEAN, false, 1>), dim3((unsigned)grid), dim3(512), 0, st, p);
        else hipLaunchKernelGGL((nwk_tile_kernel_f16<NWKNN_METRIC_SQEUCLIDEAN, false, 0, 1>), dim3((unsigned)grid), dim3(512), 0, st, p);
    } else if (metric == NWKNN_METRIC_ONE_MINUS_DOT) {
        if (p.f16_shape == 1) hipLaunchKernelGGL((nwk_tile_kernel_f16s<NWKNN_METRIC_ONE_MINUS_DOT, false, 1>), dim3((unsigned)grid), dim3(512), 0, st, p);
        else hipLaunchKernelGGL((nwk_tile_kernel_f16<NWKNN_METRIC_ONE_MINUS_DOT, false, 0, 1>), dim3((unsigned)grid), dim3(512), 0, st, p);
    }
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // extern "C++"
