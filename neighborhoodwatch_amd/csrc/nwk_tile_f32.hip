// nwk_tile_f32.hip -- fp32 distance tile + filter kernel (K1)  (gfx950 only; shared pieces and the overall design: nwk_device.h)
#include "nwk_device.h"

template <int METRIC, bool DENSE, int TM, int BKT, int DIAG = 0, int OCC = 2>
__global__ __launch_bounds__(256, OCC) void nwk_tile_kernel(const NwkTileParams p)
{
    constexpr int MT = TM / 64;                   // 32-row accumulator blocks per wave
    constexpr int ROWB = BKT * 4;                 // bytes per LDS row
    constexpr int CH = BKT / 4;                   // 16-B chunks per row
    constexpr int RSH = (ROWB == 128) ? 1 : 2;    // log2(rows per 256-B bank row)
    constexpr int A_BYTES = TM * ROWB, B_BYTES = TB * ROWB;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int RPI = 1024 / ROWB;              // rows filled by one DMA wave-instruction
    constexpr int IA = TM / RPI / 4, IB = TB / RPI / 4;   // DMA instructions per wave per slab (A, B)
    constexpr int LPS = IA + IB;
    constexpr int NG = BKT / 8;                   // 8-k groups per slab
    static_assert((TM == 128 && BKT == 32) || (TM == 256 && BKT == 16), "built shapes");
    static_assert(2 * STAGE >= TM * 16, "epilogue staging reuses the operand stages");
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    unsigned long long st0 = 0, st1 = 0, st2 = 0, rt0 = 0;
    if (NWK_DBG(p)) { st0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
    const TileCoord tc = tile_from_block(p);
    if (!tc.ok) return;

    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63;
    const int wr = w >> 1, wc = w & 1;
    const int h = l >> 5, c32 = l & 31;
    const long long q0 = (long long)tc.qt * TM;
    const long long b0 = (long long)tc.bt * TB;
    const int dp = p.dp;
    const int nk = dp / BKT;

    // ---- LDS-DMA sources.  Wave w fills rows [w*IA*RPI, (w+1)*IA*RPI) of the A tile and [w*IB*RPI, ...) of the B tile; instruction i of
    // a wave brings rows (w*I+i)*RPI + l/CH.  With 64-B rows (RPI = 16) the swizzle term ((row >> 2) & 3) = (l/CH >> 2) & 3 is the same
    // for every instruction and wave, so ONE per-lane offset serves all of them and the row block goes into the instruction's scalar
    // offset (7 VGPRs less than an offset per instruction: the 168-register shape no longer depends on what else the compiler sees).
    // With 128-B rows (RPI = 8) the term depends on the parity of i: two per-lane offsets.
    // (two scalars, not an array: a lambda that captures an array whose bound depends on a template parameter makes this clang drop the
    // kernel's host-side stub without a diagnostic)
    constexpr bool TWO = ROWB != 64;
    const unsigned voff0 = (unsigned)((l / CH) * dp + (((l % CH) ^ (((l / CH) >> RSH) & (CH - 1))) << 2)) * 4u;
    const unsigned voff1 = TWO ? (unsigned)((l / CH) * dp + (((l % CH) ^ (((RPI + l / CH) >> RSH) & (CH - 1))) << 2)) * 4u : voff0;
    static_assert((RPI >> RSH) % CH == 0 || TWO, "swizzle term independent of the instruction index");
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.qp + q0 * dp), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bp + b0 * dp), 0, 0x7FFFFFFF, 0x00020000);
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int ldsA = wu * IA * 1024;
    const int ldsB = A_BYTES + wu * IB * 1024;
    const int rowA = wu * IA * RPI * dp * 4, rowB = wu * IB * RPI * dp * 4;       // byte offset of this wave's first row block (scalar)

    auto issue_stage = [&](int stage, int kt) {
        char* base = smem + stage * STAGE;
        const int soff = kt * ROWB;
#pragma unroll
        for (int i = 0; i < IA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)(base + ldsA + i * 1024), 16, (i & 1) ? voff1 : voff0, soff + rowA + i * RPI * dp * 4, 0, 0);
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            if ((DIAG & 8) && kt > 0) continue;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)(base + ldsB + i * 1024), 16, (i & 1) ? voff1 : voff0, soff + rowB + i * RPI * dp * 4, 0, 0);
        }
    };

    // ---- fragment read addresses: lane (h, c32) reads 16 B of row (block*32 + c32), logical chunk 2g+h
    int offA[MT], offB[2];
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        const int ra = wr * (TM / 2) + m * 32 + c32;
        offA[m] = ra * ROWB + ((h ^ ((ra >> RSH) & (CH - 1))) << 4);
    }
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        const int rb = wc * 64 + n * 32 + c32;
        offB[n] = A_BYTES + rb * ROWB + ((h ^ ((rb >> RSH) & (CH - 1))) << 4);
    }

    f32x16 acc[MT][2];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    // column side (norm of base row) and row side (norm + threshold of query row q0+tid) of the epilogue: in flight under the K loop
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
    if (tid < TM) {
        my_qn = p.qn[q0 + tid];
        if (!DENSE) my_tau = p.tau[q0 + tid];
    }

    issue_stage(0, 0);
    if (NWK_DBG(p)) st1 = __builtin_amdgcn_s_memtime();
    f32x4 fa[MT], fb[2];
    if (DIAG & 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int m = 0; m < MT; ++m) fa[m] = *(const f32x4*)(smem + offA[m]);
#pragma unroll
        for (int n = 0; n < 2; ++n) fb[n] = *(const f32x4*)(smem + offB[n]);
    }
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            if (!(DIAG & 2)) issue_stage(cur ^ 1, kt + 1);
            // everything older than the LPS loads just issued -- i.e. stage `cur` -- has landed
            if ((DIAG & 8) && kt > 0) { if (IA == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); }
            else if (LPS == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (!(DIAG & 4)) __builtin_amdgcn_s_barrier();
        const char* sb = smem + cur * STAGE;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            f32x4 a[MT], b[2];
#pragma unroll
            for (int m = 0; m < MT; ++m) {
                if (DIAG & 1) { a[m] = fa[m]; asm volatile("" : "+v"(a[m])); }
                else a[m] = *(const f32x4*)(sb + (offA[m] ^ (g << 5)));
            }
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                if (DIAG & 1) { b[n] = fb[n]; asm volatile("" : "+v"(b[n])); }
                else b[n] = *(const f32x4*)(sb + (offB[n] ^ (g << 5)));
            }
            // one accumulator per output carries the whole K loop in ascending k: t = 0..3 are k-pairs 8g+2t, 8g+2t+1
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][t], b[n][t], acc[m][n], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (!(DIAG & 4)) __builtin_amdgcn_s_barrier();                // everyone is done reading stage `cur`
    }

    if (NWK_DBG(p)) st2 = __builtin_amdgcn_s_memtime();
    // ---- epilogue: accumulators -> keys -> filter -> candidate lists (staging reuses the idle operand stages)
    stage_query_side<METRIC, DENSE>(smem, TM, tid, my_qn, my_tau, p, q0);
    __syncthreads();
    tile_epilogue<METRIC, DENSE, MT>(p, acc, (const float*)smem, (const u64*)(smem + TM * 4), (const float*)(smem + TM * 12), bnv, colv, colok,
                                     q0, wr * (TM / 2), h, c32);
    if (NWK_DBG(p) && tid == 0) {
        unsigned long long* o = NWK_DBG(p) + (size_t)blockIdx.x * 8;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_amdgcn_s_memtime(); o[4] = rt0; o[5] = __builtin_amdgcn_s_memrealtime();
        o[6] = (unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32      // HW_REG_XCC_ID[3:0]
             | (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));           // HW_REG_HW_ID
        o[7] = 1;
    }
}


// ------------------------------------------------------------------------------------------------
// launchers (host side, called from nwknn_api.cpp)
// ------------------------------------------------------------------------------------------------
extern "C++" {

template <int METRIC, int TM, int BKT>
static hipError_t launch_tile_shape(hipStream_t st, const NwkTileParams& p, bool dense, unsigned grid, int diag, int occ)
{
    // <256,16>: 48 KiB of LDS per workgroup, so THREE fit a CU if the kernel stays within 168 VGPRs (it does, with 5 spilled
    // outside the K loop): the third workgroup's MFMAs fill what the barriers, DMA issue and tile turnarounds of the other two
    // leave idle -- C3: 144.3 -> 148.6 TF (+3.0 %, same-box A/B).  <128,32> needs 64 KiB: two per CU.  (A 128 x 128 tile on
    // 16-float slabs fits FOUR per CU -- 112 VGPRs, 32 KiB -- and is slower: 145.4 TF, a third more operand bytes per FLOP.)
    if constexpr (TM == 256) {
        if (!dense && diag == 0 && occ >= 3) {
            hipLaunchKernelGGL((nwk_tile_kernel<METRIC, false, TM, BKT, 0, 3>), dim3(grid), dim3(256), 0, st, p);
            return hipGetLastError();
        }
    }
    if (dense) { hipLaunchKernelGGL((nwk_tile_kernel<METRIC, true, TM, BKT, 0>), dim3(grid), dim3(256), 0, st, p); return hipGetLastError(); }
#ifdef NWKNN_DEV
    switch (diag) {              // timing ablations (wrong results by construction): developer builds only
    case 1: hipLaunchKernelGGL((nwk_tile_kernel<METRIC, false, TM, BKT, 1>), dim3(grid), dim3(256), 0, st, p); return hipGetLastError();
    case 2: hipLaunchKernelGGL((nwk_tile_kernel<METRIC, false, TM, BKT, 2>), dim3(grid), dim3(256), 0, st, p); return hipGetLastError();
    case 4: hipLaunchKernelGGL((nwk_tile_kernel<METRIC, false, TM, BKT, 4>), dim3(grid), dim3(256), 0, st, p); return hipGetLastError();
    case 8: hipLaunchKernelGGL((nwk_tile_kernel<METRIC, false, TM, BKT, 8>), dim3(grid), dim3(256), 0, st, p); return hipGetLastError();
    default: break;
    }
#endif
    hipLaunchKernelGGL((nwk_tile_kernel<METRIC, false, TM, BKT, 0>), dim3(grid), dim3(256), 0, st, p);
    return hipGetLastError();
}

template <int METRIC>
static hipError_t launch_tile_metric(hipStream_t st, const NwkTileParams& p, bool dense, unsigned grid, int tile_rows, int diag, int occ)
{
    if (tile_rows == 256) return launch_tile_shape<METRIC, 256, 16>(st, p, dense, grid, diag, occ);
    return launch_tile_shape<METRIC, 128, 32>(st, p, dense, grid, diag, occ);
}

// p.n_qtiles counts query tiles of `tile_rows` (128 or 256) rows; p.n_btiles counts 128-row base tiles.
hipError_t nwk_launch_tile(hipStream_t st, NwkTileParams p, int metric, bool dense, int tile_rows, int diag, int occ)
{
    if (p.n_qtiles <= 0 || p.n_btiles <= 0) return hipSuccess;
    if (tile_rows != 128 && tile_rows != 256) return hipErrorInvalidValue;
    if (p.sq_shift < 0 || p.sq_shift > 6) return hipErrorInvalidValue;
    const int SQ = 1 << p.sq_shift, SB = 64 >> p.sq_shift;
    const long long n_super_q = (p.n_qtiles + SQ - 1) / SQ;
    p.n_super_b = (p.n_btiles + SB - 1) / SB;
    long long supers = n_super_q * p.n_super_b;
    supers = (supers + 7) / 8 * 8;
    const long long grid = supers * 64;
    if (grid > 0x7FFFFFFFll) return hipErrorInvalidValue;
    switch (metric) {
    case NWKNN_METRIC_SQEUCLIDEAN: return launch_tile_metric<NWKNN_METRIC_SQEUCLIDEAN>(st, p, dense, (unsigned)grid, tile_rows, diag, occ);
    case NWKNN_METRIC_ONE_MINUS_DOT: return launch_tile_metric<NWKNN_METRIC_ONE_MINUS_DOT>(st, p, dense, (unsigned)grid, tile_rows, diag, occ);
    case NWKNN_METRIC_COSINE: return launch_tile_metric<NWKNN_METRIC_COSINE>(st, p, dense, (unsigned)grid, tile_rows, diag, occ);
    default: return hipErrorInvalidValue;
    }
}

}  // extern "C++"
