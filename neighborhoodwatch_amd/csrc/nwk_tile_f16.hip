// nwk_tile_f16.hip -- fp16 distance tile + filter kernel (K1h), burst form: v_mfma_f32_16x16x32_f16, one burst of LDS-DMA per slab; the only
// fp16 build with a dense mode.  PRE = 1 is the pre-screen of the exact search.  Launcher of all fp16 tile kernels.  (gfx950 only; shared
// pieces and the overall design: nwk_device.h, nwk_f16_common.h; the streaming form: nwk_tile_f16p.hip)
#include "nwk_device.h"
#include "nwk_f16_common.h"

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

    unsigned long long st0 = 0, st1 = 0, st2 = 0, rt0 = 0;       // dev stamps (NWK_DBG(p)): see tools/stamps2.py
    if (NWK_DBG(p)) { st0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
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
    if (NWK_DBG(p) && t == 0) st1 = __builtin_amdgcn_s_memtime();
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
    if (NWK_DBG(p)) st2 = __builtin_amdgcn_s_memtime();
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
    if (NWK_DBG(p) && l == 0) {          // wave 0 writes its own stamps; every wave folds its end times in (the tile ends with its last wave)
        unsigned long long* o = NWK_DBG(p) + (size_t)blockIdx.x * 8;
        const unsigned long long t3 = __builtin_amdgcn_s_memtime(), r3 = __builtin_amdgcn_s_memrealtime();
        if (w == 0) {
            o[0] = st0; o[1] = st1; o[2] = st2; o[4] = rt0;
            o[6] = (unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32      // HW_REG_XCC_ID[3:0]
                 | (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));           // HW_REG_HW_ID
            o[7] = 1;
        }
        atomicMax(o + 3, t3);
        atomicMax(o + 5, r3);
        unsigned long long* o2 = NWK_DBG(p) + (size_t)gridDim.x * 8 + (size_t)blockIdx.x * 16 + w * 2;     // per wave: end of its K loop, end of its epilogue
        o2[0] = st2; o2[1] = t3;
    }
}

// ------------------------------------------------------------------------------------------------
// launchers (host side, called from nwknn_api.cpp)
// ------------------------------------------------------------------------------------------------
extern "C++" {

// the burst kernel's and the streaming kernel's workgroups walk strips of p.tiles_per_wg base tiles: p.n_btiles becomes the number of strips
static void setup_strips(NwkTileParams& p, bool dense)
{
    p.n_btiles_real = p.n_btiles;
    if (!dense && p.tiles_per_wg > 1) p.n_btiles = (p.n_btiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
    else p.tiles_per_wg = 1;
}

template <int PRE>
static hipError_t launch_f16s(hipStream_t st, const NwkTileParams& p, int metric, bool dense, unsigned grid)
{
#define NWK_S(M)                                                                                                  \
    do {                                                                                                           \
        if (dense) hipLaunchKernelGGL((nwk_tile_kernel_f16s<M, true, PRE>), dim3(grid), dim3(512), 0, st, p);      \
        else hipLaunchKernelGGL((nwk_tile_kernel_f16s<M, false, PRE>), dim3(grid), dim3(512), 0, st, p);           \
    } while (0)
    if (metric == NWKNN_METRIC_SQEUCLIDEAN) NWK_S(NWKNN_METRIC_SQEUCLIDEAN);
    else if (metric == NWKNN_METRIC_ONE_MINUS_DOT) NWK_S(NWKNN_METRIC_ONE_MINUS_DOT);
    else if (metric == NWKNN_METRIC_COSINE) NWK_S(NWKNN_METRIC_COSINE);
    else return hipErrorInvalidValue;
#undef NWK_S
    return hipGetLastError();
}

// fp16 tile launch.  p.n_qtiles counts 256-row query tiles; p.n_btiles / p.bt0 count 256-row base tiles; p.qp/p.bp point at halves.
// pre = 0: raw fp16 sums against the thresholds (compute path with option f16_exact = 0, or inputs holding Inf/NaN);
// pre = 1: the conservative pre-screen (p.qp/p.bp are the scaled copies, or the native fp16 rows in exact fp16 mode), keys carry estimates.
// p.f16_shape: 1 = burst form (nwk_tile_kernel_f16s; the only build with a dense mode), 3 = streaming form (nwk_tile_kernel_f16p),
// 4 = ping-pong form (nwk_tile_kernel_f16x), 5 = two-issuing-waves form (nwk_tile_kernel_f16y).
hipError_t nwk_launch_tile_f16(hipStream_t st, NwkTileParams p, int metric, bool dense, int pre)
{
    if (p.n_qtiles <= 0 || p.n_btiles <= 0) return hipSuccess;
    if (p.sq_shift < 0 || p.sq_shift > 6) return hipErrorInvalidValue;
    if (dense) p.f16_shape = 1;
    setup_strips(p, dense);
    const int SQ = 1 << p.sq_shift, SB = 64 >> p.sq_shift;
    const long long n_super_q = (p.n_qtiles + SQ - 1) / SQ;
    p.n_super_b = (p.n_btiles + SB - 1) / SB;
    long long supers = n_super_q * p.n_super_b;
    supers = (supers + 7) / 8 * 8;
    const long long grid = supers * 64;
    if (grid > 0x7FFFFFFFll) return hipErrorInvalidValue;
    if (p.f16_shape == 3) return nwk_launch_tile_f16p(st, p, metric, pre, (unsigned)grid);
    if (p.f16_shape == 4) return pre ? nwk_launch_tile_f16x_pre(st, p, metric, (unsigned)grid) : nwk_launch_tile_f16x_raw(st, p, metric, (unsigned)grid);
    if (p.f16_shape == 5) return pre ? nwk_launch_tile_f16y_pre(st, p, metric, (unsigned)grid) : nwk_launch_tile_f16y_raw(st, p, metric, (unsigned)grid);
    if (p.f16_shape != 1) return hipErrorInvalidValue;
    return pre ? launch_f16s<1>(st, p, metric, dense, (unsigned)grid) : launch_f16s<0>(st, p, metric, dense, (unsigned)grid);
}

}  // extern "C++"
