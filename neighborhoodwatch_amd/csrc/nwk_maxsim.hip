// nwk_maxsim.hip -- MaxSim late interaction (extension, not in the reference): tile, pre-screen, prune and rescore kernels  (gfx950 only; shared pieces and the overall design: nwk_device.h)
#include "nwk_device.h"
#include <algorithm>

// ------------------------------------------------------------------------------------------------
// MaxSim late interaction (BASELINE config 5; NOT in the reference -- its `ck` is flat token-level kNN):
//     score(query, doc) = sum over the query's TQK tokens of  max over the doc's TDK tokens of  (q_token . d_token)
// and the k docs with the highest score per query.  One workgroup multiplies 256 doc-token rows (4 docs of 64 tokens)
// by 128 query-token columns (4 queries of 32 tokens) with the same LDS/DMA/MFMA loop as the kNN tile (d = 128 is
// 8 slabs of 16).  Orientation matters: doc tokens sit on the accumulator ROWS (registers), query tokens on the lanes,
// so the max over a doc's 64 tokens is 31 in-lane v_max + one cross-half exchange, and only the final sum over the 32
// query tokens crosses lanes (5 butterfly steps per (doc, query) pair; the transposed orientation needs 80 per pair).
// Each wave yields 2 docs x 2 queries = 4 scores per tile.  Scores are ranked through the same 64-bit keys as distances
// (key = sortable(-score) << 32 | doc), so list filtering, compaction and repair are shared with the kNN path.
// Canonical arithmetic: dot = ascending-k fmaf chain (fp32 MFMA), max exact, token sum = balanced pairwise tree in
// natural token order (what the xor-1,2,4,8,16 butterfly computes) -- restated by oracle nwo_maxsim, bit for bit.
// ------------------------------------------------------------------------------------------------
#define MS_TDK 64                  // tokens per doc
#define MS_TQK 32                  // tokens per query
#define MS_DOCS_PER_TILE 4
#define MS_QUERIES_PER_TILE 4

// The token sum without LDS exchanges (round 2).  __shfl_xor is a ds_bpermute round trip; six of them in a row per (doc, query) pair,
// four pairs per wave and tile, were as long as the fp16 tile's MFMAs.  One v_permlane32_swap gives every lane the max of the two
// half-waves, four DPP adds (quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror) produce lane for lane the values of the
// xor-1,2,4,8 butterfly (fp32 addition is commutative, and after each level the groups being mirrored are uniform), and the last
// level is lane 0's view of xor 16: row 0 + row 1.  The balanced pairwise tree of the canonical arithmetic, bit for bit.
template <int CTRL>
__device__ __forceinline__ float ms_dpp(float x)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, false));
}

// v: a lane's max over its 32 accumulator rows of one (doc, query-token column); lanes l and l ^ 32 hold the two halves of the doc's
// 64 tokens.  Returns the sum over the 32 query tokens (lanes 0..31) of the max over the doc's tokens, wave-uniform.
__device__ __forceinline__ float ms_token_sum(float v)
{
    const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    // (elements copied to scalars first: __builtin_bit_cast(float, sw[1]) on the vector element reads element 0 with this clang)
    const unsigned sw_lo = sw[0], sw_hi = sw[1];
    float s = fmaxf(__builtin_bit_cast(float, sw_lo), __builtin_bit_cast(float, sw_hi));     // [lo, lo] vs [hi, hi]: every lane holds max(lo, hi)
    s = __fadd_rn(s, ms_dpp<0xB1>(s));       // quad_perm [1,0,3,2]  == xor 1
    s = __fadd_rn(s, ms_dpp<0x4E>(s));       // quad_perm [2,3,0,1]  == xor 2
    s = __fadd_rn(s, ms_dpp<0x141>(s));      // row_half_mirror      == xor 4 (quads are uniform by now)
    s = __fadd_rn(s, ms_dpp<0x140>(s));      // row_mirror           == xor 8
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), 16));
    return __fadd_rn(r0, r1);                // == xor 16 as lane 0 sees it
}

template <bool DENSE>
__global__ __launch_bounds__(256, 2) void nwk_maxsim_kernel(const NwkTileParams p)
{
    constexpr int TM = 256, BKT = 16;
    constexpr int MT = TM / 64, ROWB = BKT * 4, CH = BKT / 4, RSH = 2;
    constexpr int A_BYTES = TM * ROWB, B_BYTES = TB * ROWB, STAGE = A_BYTES + B_BYTES;
    constexpr int RPI = 1024 / ROWB, IA = TM / RPI / 4, IB = TB / RPI / 4;
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const TileCoord tc = tile_from_block(p);         // tc.qt: group of 4 queries, tc.bt: tile of 4 docs
    if (!tc.ok) return;
    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63;
    const int wr = w >> 1, wc = w & 1;
    const int h = l >> 5, c32 = l & 31;
    const long long drow0 = (long long)tc.bt * TM;   // first doc-token row of the tile   (A operand, from p.bp)
    const long long qrow0 = (long long)tc.qt * TB;   // first query-token row of the tile (B operand, from p.qp)
    const int dp = p.dp;
    const int nk = dp / BKT;

    unsigned voffA[4], voffB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = (w * IA + i) * RPI + (l / CH), rb = (w * IB + i) * RPI + (l / CH);
        voffA[i] = (unsigned)(ra * dp + (((l % CH) ^ ((ra >> RSH) & (CH - 1))) << 2)) * 4u;
        voffB[i] = (unsigned)(rb * dp + (((l % CH) ^ ((rb >> RSH) & (CH - 1))) << 2)) * 4u;
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bp + drow0 * dp), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.qp + qrow0 * dp), 0, 0x7FFFFFFF, 0x00020000);
    const int ldsA = __builtin_amdgcn_readfirstlane(w * IA * 1024);
    const int ldsB = __builtin_amdgcn_readfirstlane(A_BYTES + w * IB * 1024);
    auto issue_stage = [&](int stage, int kt) {
        char* base = smem + stage * STAGE;
        const int soff = kt * ROWB;
#pragma unroll
        for (int i = 0; i < IA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)(base + ldsA + i * 1024), 16, voffA[i], soff, 0, 0);
#pragma unroll
        for (int i = 0; i < IB; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)(base + ldsB + i * 1024), 16, voffB[i], soff, 0, 0);
    };

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

    // thresholds of this wave's two queries (wave-uniform addresses), in flight under the K loop
    const long long q_first = (long long)tc.qt * MS_QUERIES_PER_TILE + 2 * wc;
    u64 tauq[2] = {~0ull, ~0ull};
    if (!DENSE) { tauq[0] = p.tau[q_first]; tauq[1] = p.tau[q_first + 1]; }

    issue_stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            issue_stage(cur ^ 1, kt + 1);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const char* sb = smem + cur * STAGE;
#pragma unroll
        for (int g = 0; g < BKT / 8; ++g) {
            f32x4 a[MT], b[2];
#pragma unroll
            for (int m = 0; m < MT; ++m) a[m] = *(const f32x4*)(sb + (offA[m] ^ (g << 5)));
#pragma unroll
            for (int n = 0; n < 2; ++n) b[n] = *(const f32x4*)(sb + (offB[n] ^ (g << 5)));
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m][t], b[n][t], acc[m][n], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    // ---- epilogue: 2 docs x 2 queries per wave (the four max trees, then the four token sums, interleaved by the scheduler)
    const int cap = p.cap;
    float tsum[2][2];
#pragma unroll
    for (int dd = 0; dd < 2; ++dd)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            float v = acc[2 * dd][n][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) v = fmaxf(v, acc[2 * dd][n][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) v = fmaxf(v, acc[2 * dd + 1][n][r]);
            tsum[dd][n] = v;                                  // the two half-waves hold the two halves of this doc's 64 token rows
        }
#pragma unroll
    for (int dd = 0; dd < 2; ++dd)
#pragma unroll
        for (int n = 0; n < 2; ++n) tsum[dd][n] = ms_token_sum(tsum[dd][n]);     // balanced pairwise tree over the query's tokens
#pragma unroll
    for (int dd = 0; dd < 2; ++dd) {
        const long long doc = (long long)tc.bt * MS_DOCS_PER_TILE + 2 * wr + dd;
        const bool doc_ok = doc < p.nb_valid;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const long long q = q_first + n;
            const u64 key = ((u64)sortable_u32(-tsum[dd][n]) << 32) | (u32)doc;
            if (l == 0) {
                if (DENSE) {
                    p.buf[q * cap + (doc - p.dense_col0)] = doc_ok ? key : ~0ull;
                } else if (doc_ok && key < tauq[n]) {
                    const u32 pos = atomicAdd(p.cnt + q, 1u);
                    if (pos < (u32)cap) p.buf[q * cap + pos] = key;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// MaxSim through the fp16 pre-screen (option "prescreen" before nwknn_maxsim_set_docs).  The score is a sum over the query's tokens of
// a max over the doc's tokens of dot products, so its fp16 estimate is off by at most
//     sum_i max_j E(q_i, d_j)  <=  R(q) + C(d),   R(q) = sum_i rowerr(q_i),   C(d) = 32 * max_j colerr(d_j)
// with E = rowerr + colerr the per-dot bound of the kNN pre-screen: additive in (query, doc), so the same test and the same
// bound-based pruning apply.  nwk_maxsim_pre_kernel is nwk_maxsim_kernel on the scaled fp16 copies (v_mfma_f32_32x32x16_f16: the same
// accumulator layout, hence the same in-lane max / butterfly sum), 32-half slabs of 64-B LDS rows (image of the f16q kernel), two
// workgroups per CU; it lists (query, doc) pairs with key = fp16 estimate of -score.  nwk_maxsim_rescore_kernel then recomputes
// the survivors with the canonical arithmetic of nwk_maxsim_kernel (ascending-k fmaf chains, exact max, pairwise tree sum).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nwk_maxsim_bounds_kernel(const float* __restrict__ norms, long long n_items, int tokens, float eps_half,
                                                                 float a_self, float cst, int is_doc, float* __restrict__ out)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_items) return;
    float s = 0.0f, m = 0.0f, ns = 0.0f, nm = 0.0f;
    for (int j = 0; j < tokens; ++j) {
        const float n = norms[i * tokens + j];
        const float e = eps_half * n + a_self * __builtin_amdgcn_sqrtf(n) + cst;
        s += e; m = fmaxf(m, e); ns += n; nm = fmaxf(nm, n);
    }
    // + the two fp32 token sums (canonical and estimate) disagreeing by a few ulps of sum|max dot| <= (sum qn + 32 max bn) / 2
    out[i] = is_doc ? (float)NWK_MS_TOKENS_PER_QUERY * (m * 1.0001f + 4.0e-6f * nm) : s * 1.0001f + 4.0e-6f * ns + 1.0e-30f;
}

__global__ __launch_bounds__(256, 2) void nwk_maxsim_pre_kernel(const NwkTileParams p)
{
    constexpr int TM = 256, TQ = 128, STAGE = (TM + TQ) * 64;        // 32-half slabs: 64-B rows, 24 KiB per stage
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const TileCoord tc = tile_from_block(p);         // tc.qt: group of 4 queries, tc.bt: tile of 4 docs
    if (!tc.ok) return;
    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63;
    const int wr = w >> 1, wc = w & 1;
    const int h = l >> 5, c32 = l & 31;
    const long long drow0 = (long long)tc.bt * TM;   // first doc-token row of the tile   (A operand, fp16 copy in p.bp)
    const long long qrow0 = (long long)tc.qt * TQ;   // first query-token row of the tile (B operand, fp16 copy in p.qp)
    const int dp = p.dp;                             // halves per row
    const int nk = dp / 32;
    const _Float16* docs = (const _Float16*)p.bp;
    const _Float16* qrys = (const _Float16*)p.qp;

    const unsigned voff = (unsigned)((l >> 2) * dp * 2 + (((l & 3) ^ ((0 - (l >> 4)) & 3)) << 4));
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(docs + drow0 * dp), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(qrys + qrow0 * dp), 0, 0x7FFFFFFF, 0x00020000);
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int rowsA = wu * 64 * dp * 2, rowsB = wu * 32 * dp * 2, step16 = 16 * dp * 2;
    auto issue_stage = [&](int stage, int kt) {
        char* st = smem + stage * STAGE;
        const int soff = kt * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)(st + wu * 4096 + i * 1024), 16, voff, soff + rowsA + i * step16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)(st + TM * 64 + wu * 2048 + i * 1024), 16, voff, soff + rowsB + i * step16, 0, 0);
    };

    const int swz = (0 - (c32 >> 2)) & 3;
    int offA[4], offB[2];
#pragma unroll
    for (int m = 0; m < 4; ++m) offA[m] = (wr * 128 + m * 32 + c32) * 64 + ((h ^ swz) << 4);
#pragma unroll
    for (int n = 0; n < 2; ++n) offB[n] = TM * 64 + (wc * 64 + n * 32 + c32) * 64 + ((h ^ swz) << 4);

    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.0f;

    const long long q_first = (long long)tc.qt * MS_QUERIES_PER_TILE + 2 * wc;
    const u64 tauq[2] = {p.tau[q_first], p.tau[q_first + 1]};
    const float rq[2] = {p.ms_rq[q_first], p.ms_rq[q_first + 1]};

    issue_stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            issue_stage(cur ^ 1, kt + 1);
            asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        const char* sb = smem + cur * STAGE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            f16x8 a[4], b[2];
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = *(const f16x8*)(sb + (offA[m] ^ (ks << 5)));
#pragma unroll
            for (int n = 0; n < 2; ++n) b[n] = *(const f16x8*)(sb + (offB[n] ^ (ks << 5)));
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m], b[n], acc[m][n], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    }

    const int cap = p.cap;
#pragma unroll
    for (int dd = 0; dd < 2; ++dd) {
        const long long doc = (long long)tc.bt * MS_DOCS_PER_TILE + 2 * wr + dd;
        const bool doc_ok = doc < p.nb_valid;
        const float cd = doc_ok ? p.ms_cd[doc] : 0.0f;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            float v = acc[2 * dd][n][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) v = fmaxf(v, acc[2 * dd][n][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) v = fmaxf(v, acc[2 * dd + 1][n][r]);
            v = fmaxf(v, __shfl_xor(v, 32));
            float sum = v;
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) sum = __fadd_rn(sum, __shfl_xor(sum, o));
            if (l == 0 && doc_ok) {
                const long long q = q_first + n;
                const float score = sum * p.pre_inv_scale;                         // fp16 estimate of the MaxSim score
                bool pass = true;
                if (tauq[n] != ~0ull) {
                    const float td = unsortable_f32((u32)(tauq[n] >> 32));          // = -(k-th best score so far)
                    const float err = rq[n] + cd;
                    pass = !(score < (-td) - err - 2.0e-6f * (fabsf(td) + err));
                }
                if (pass && q < p.nq_valid) {
                    const u32 pos = atomicAdd(p.cnt + q, 1u);
                    if (pos < (u32)cap) p.buf[q * cap + pos] = ((u64)__float_as_uint(-score) << 32) | (u32)doc;
                }
            }
        }
    }
}

// The pre-screen for d <= 128 (ColBERT: 128), "ring" form: the default.  A workgroup walks a strip of p.tiles_per_wg consecutive document
// tiles for ONE group of 4 queries; the query tokens go through LDS once, their fragments then stay in registers for the whole strip, and
// only document tokens stream.  Round 2's first strip kernel (two 16 KiB stages; removed in round 3 together with the 32x32x16 build of the
// ring) sat at 37 % of the fp16 MFMA rate although a tile is only 64 MFMAs per wave, for two reasons:
//   * its two 16 KiB stages let a document slab be requested ONE slab (16 MFMAs per wave, ~0.25 us) before it is waited for, with two
//     barriers per slab -- the LDS-DMA round trip is several times that, so every slab began with a stall;
//   * its epilogue ran 4 x 6 ds_bpermute exchanges one after the other, each behind an s_waitcnt lgkmcnt(0) (the butterfly sum over
//     the query's tokens, once per (doc, query) pair of the wave): 24 LDS round trips per tile, as long as the tile's MFMAs.
// Here the document stream runs through a ring of FOUR 16 KiB stages (a whole d = 128 tile ahead; 64 KiB per workgroup, still two
// workgroups per CU): slab S+3 is requested at the top of slab S into the stage slab S-1 just released, ONE barrier per slab,
// one counted wait (vmcnt(8): the requests of slabs S+1 and S+2 may still be in flight; requests past the end of the strip are
// still issued, source clamped to the last slab, so the count is a constant), and the stream runs on across the tiles of the strip.
// The first k-step of a tile takes a zero C operand instead of cleared accumulators; the per-document bounds of the strip are
// staged in LDS once (no register-destination load, hence no vmcnt(0), inside the loop).  The epilogue exchanges nothing through
// LDS: the cross-half max is one v_permlane32_swap, the sum over the 32 query tokens four DPP adds (quad_perm [1,0,3,2], [2,3,0,1],
// row_half_mirror, row_mirror: lane for lane the values of the xor-1,2,4,8 butterfly, fp32 addition being commutative) and two
// v_readlane for the last level, the four pairs of a wave interleaved.  Same estimates, same keys as the strip kernel.
#define MS_RING_MAX_STRIP 64

// The ring kernel is built on v_mfma_f32_16x16x32_f16.  Under this kernel the chip holds only ~1.7 GHz (PMC: profiles/
// r02e_c5_pmc_summary.json), and in that power-limited regime the 16x16x32 shape delivers more FLOP/s than 32x32x16 at equal cycles
// (MI355X_MICROARCH.md, DVFS give-back item 7; the kNN fp16 tile kernel gained 5 % from the same switch).  Same workgroup tile, same LDS
// image, ring, waits and barriers; a 32-half slab is ONE k-step: 8 row blocks x 4 column blocks = 32 MFMAs per wave and slab.
// Accumulator layout: lane l holds D[4 (l / 16) + i][l % 16] -- a doc's 64 token rows are 4 row blocks x 4 registers in-lane and the four
// 16-lane rows of the wave, so the max over a doc is 15 in-lane v_max, one v_permlane16_swap (rows 0|1, 2|3) and one
// v_permlane32_swap (halves); the sum over a query's 32 tokens is one in-lane add of its two column blocks and the four DPP adds over
// the 16 lanes of a row.  (The token sum's association differs from the canonical tree here -- block c | c + 16 first -- which an
// ESTIMATE may do: the bound carries 33 ulps of the sum of norms for exactly that.)
__device__ __forceinline__ float ms_rows_max(float v)      // max over the wave's four 16-lane rows, in every lane
{
    const auto a = __builtin_amdgcn_permlane16_swap(__builtin_bit_cast(unsigned, v), __builtin_bit_cast(unsigned, v), false, false);
    const unsigned a0 = a[0], a1 = a[1];                     // [r0 r0 r2 r2] | [r1 r1 r3 r3]   (scalars first: see ms_token_sum)
    const float m1 = fmaxf(__builtin_bit_cast(float, a0), __builtin_bit_cast(float, a1));
    const auto b = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, m1), __builtin_bit_cast(unsigned, m1), false, false);
    const unsigned b0 = b[0], b1 = b[1];
    return fmaxf(__builtin_bit_cast(float, b0), __builtin_bit_cast(float, b1));
}

// WAVES = 4: the 256 x 128 tile above (4 documents x 4 queries), two workgroups per CU.
// WAVES = 8 (round 5, the default): a 256 x 256 tile, 4 documents x EIGHT queries, one eight-wave workgroup per CU -- the same eight waves a CU
// held before (a wave still owns 128 document rows x 64 query-token columns = 2 documents x 2 queries, its 64 registers of query fragments and
// its 128 accumulators), but they now share ONE document stream and one ring: half the LDS-DMA instructions, L2 -> LDS bytes and ring space per
// FLOP (256 FLOP/B instead of 128), and a slab's barrier joins the two waves of every SIMD instead of leaving them to two unrelated workgroups.
// SPB = slabs per barrier (WAVES = 8 only; round 5).  One eight-wave workgroup has the CU's whole LDS to itself, so the ring can hold EIGHT
// 16 KiB slabs and the barrier joins the waves every second slab (64 MFMAs per wave between barriers instead of 32): a barrier costs the
// interval 10-12 % whatever it waits for, because the SIMD's arbiter prefers the older wave, which then sits in the barrier while its
// partner runs alone (tools/f16y_probe.hip) -- hence also the priority ladder inside the interval (s_setprio 3, 2, 1, 0 per 16 MFMAs).
// SPB = 2 needs an even number of k-slabs per tile (d = 64, 128: ColBERT's); other depths run SPB = 1.
template <int WAVES, int SPB = 1>
__global__ __launch_bounds__(64 * WAVES, 2) void nwk_maxsim_pre_ring16_kernel(const NwkTileParams p)
{
    constexpr int TM = 256, TQ = 32 * WAVES, NK = 4, SLAB = TM * 64, RING = 4 * SPB, B_SLAB = TQ * 64;      // 32-half slabs of 64-B rows
    static_assert(SPB == 1 || WAVES == 8, "the deep ring is the eight-wave workgroup's");
    constexpr int IA = 16 / WAVES;                   // LDS-DMA instructions (16 rows x 64 B each) a wave issues per document slab: 256 rows / WAVES / 16
    constexpr int OFF_CD = RING * SLAB;
    static_assert(NK * B_SLAB <= RING * SLAB, "the query tokens are staged through the ring's space");
    __shared__ __attribute__((aligned(16))) char smem[OFF_CD + MS_RING_MAX_STRIP * MS_DOCS_PER_TILE * 4];

    const TileCoord tc = tile_from_block(p);         // tc.qt: group of WAVES queries, tc.bt: STRIP of document tiles
    if (!tc.ok) return;
    const int tid = threadIdx.x;
    const int w = tid >> 6, l = tid & 63;
    const int wr = WAVES == 8 ? (w >> 2) : (w >> 1), wc = WAVES == 8 ? (w & 3) : (w & 1);
    const int g = l >> 4, c16 = l & 15;
    const int T = p.tiles_per_wg > 1 ? p.tiles_per_wg : 1;         // <= MS_RING_MAX_STRIP (launcher)
    const int bt_first = p.bt0 + (tc.bt - p.bt0) * T;
    int n_here = p.bt0 + p.n_btiles_real - bt_first;
    if (n_here > T) n_here = T;
    const long long qrow0 = (long long)tc.qt * TQ;
    const int dp = p.dp;                             // halves per row: 32 * nk, nk <= NK
    const int nk = dp / 32;
    const _Float16* docs = (const _Float16*)p.bp;
    const _Float16* qrys = (const _Float16*)p.qp;

    const unsigned voff = (unsigned)((l >> 2) * dp * 2 + (((l & 3) ^ ((0 - (l >> 4)) & 3)) << 4));      // the ring kernel's LDS image
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(qrys + qrow0 * dp), 0, 0x7FFFFFFF, 0x00020000);
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int rowsA = wu * (16 * IA) * dp * 2, rowsB = wu * 32 * dp * 2, step16 = 16 * dp * 2;
    // fragment of a 16-row block: lane l reads row l % 16, k = 8 (l / 16) .. + 7 = the row's 16-B chunk l / 16, stored in slot chunk ^ swizzle(row)
    const int swz = (0 - (c16 >> 2)) & 3;
    const int offA0 = (wr * 128 + c16) * 64 + ((g ^ swz) << 4);      // + 1024 m
    const int offB0 = (wc * 64 + c16) * 64 + ((g ^ swz) << 4);       // + 1024 n

    for (int kt = 0; kt < nk; ++kt)
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)(smem + kt * B_SLAB + wu * 2048 + i * 1024), 16, voff,
                                                     kt * 64 + rowsB + i * step16, 0, 0);
    const long long q_first = (long long)tc.qt * WAVES + 2 * wc;      // WAVES queries per tile (32 token columns each)
    const u64 tauq[2] = {p.tau[q_first], p.tau[q_first + 1]};
    const float rq[2] = {p.ms_rq[q_first], p.ms_rq[q_first + 1]};
    float cd_mine = 0.0f;
    {
        const long long doc = (long long)bt_first * MS_DOCS_PER_TILE + tid;
        if (tid < n_here * MS_DOCS_PER_TILE && doc < p.nb_valid) cd_mine = p.ms_cd[doc];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid < MS_RING_MAX_STRIP * MS_DOCS_PER_TILE) ((float*)(smem + OFF_CD))[tid] = cd_mine;
    __builtin_amdgcn_s_barrier();
    f16x8 bq[NK][4];
#pragma unroll
    for (int kt = 0; kt < NK; ++kt)
#pragma unroll
        for (int n = 0; n < 4; ++n)
            bq[kt][n] = (kt < nk) ? *(const f16x8*)(smem + kt * B_SLAB + offB0 + n * 1024) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    float ntd[2], atd[2];
    bool open_[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        open_[n] = tauq[n] == ~0ull;
        const float td = unsortable_f32((u32)(tauq[n] >> 32));
        ntd[n] = -td; atd[n] = fabsf(td);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(docs + (long long)bt_first * TM * dp), 0, 0x7FFFFFFF, 0x00020000);
    const int NS = n_here * nk;
    const int tile_bytes = TM * dp * 2;
    int it = 0, ik = 0, ist = 0;
    auto issue_next = [&]() {
        const int soff = it * tile_bytes + ik * 64 + rowsA;
#pragma unroll
        for (int i = 0; i < IA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)(smem + ist * SLAB + wu * (IA * 1024) + i * 1024), 16, voff, soff + i * step16, 0, 0);
        ist = (ist + 1) & (RING - 1);
        if (it * nk + ik + 1 < NS) { if (++ik == nk) { ik = 0; ++it; } }
    };
#pragma unroll
    for (int i = 0; i < 3 * SPB; ++i) issue_next();                 // three barrier intervals ahead

    const int cap = p.cap;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    int st = 0;
    for (int t = 0; t < n_here; ++t) {
        f32x4 acc[8][4];
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            if (kt >= nk) break;
            if (kt % SPB == 0) {
                // all but the requests of the two intervals behind this one (2 x SPB x IA instructions) have landed
                if (2 * SPB * IA == 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int i = 0; i < SPB; ++i) issue_next();
            }
            const char* sb = smem + st * SLAB;
            st = (st + 1) & (RING - 1);
            f16x8 a[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) a[m] = *(const f16x8*)(sb + offA0 + m * 1024);
            if (SPB == 2) { __builtin_amdgcn_sched_barrier(0); if (kt % 2 == 0) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(1); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = NWK_MFMA_SNAKE && (m & 1) ? 3 - j : j;          // snaked issue order (nwk_tile_f16y.h)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], bq[kt][n], kt == 0 ? zero4 : acc[m][n], 0, 0, 0);
                }
            if (SPB == 2) { __builtin_amdgcn_sched_barrier(0); if (kt % 2 == 0) __builtin_amdgcn_s_setprio(2); else __builtin_amdgcn_s_setprio(0); __builtin_amdgcn_sched_barrier(0); }
#pragma unroll
            for (int m = 4; m < 8; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = NWK_MFMA_SNAKE && (m & 1) ? 3 - j : j;
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], bq[kt][n], kt == 0 ? zero4 : acc[m][n], 0, 0, 0);
                }
        }
        // ---- epilogue of tile t: 2 docs (row blocks 4 dd .. 4 dd + 3) x 2 queries (column blocks 2 qn, 2 qn + 1) per wave
        const float* cd_s = (const float*)(smem + OFF_CD) + t * MS_DOCS_PER_TILE + 2 * wr;
        float vmx[2][4];
#pragma unroll
        for (int dd = 0; dd < 2; ++dd)
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                float v = fmaxf(acc[4 * dd][n][0], acc[4 * dd][n][1]);
                v = fmaxf(fmaxf(v, acc[4 * dd][n][2]), acc[4 * dd][n][3]);
#pragma unroll
                for (int m = 1; m < 4; ++m) {
                    v = fmaxf(fmaxf(v, acc[4 * dd + m][n][0]), acc[4 * dd + m][n][1]);
                    v = fmaxf(fmaxf(v, acc[4 * dd + m][n][2]), acc[4 * dd + m][n][3]);
                }
                vmx[dd][n] = v;
            }
#pragma unroll
        for (int dd = 0; dd < 2; ++dd)
#pragma unroll
            for (int n = 0; n < 4; ++n) vmx[dd][n] = ms_rows_max(vmx[dd][n]);
        float score[2][2];
#pragma unroll
        for (int dd = 0; dd < 2; ++dd)
#pragma unroll
            for (int qn = 0; qn < 2; ++qn) {
                float s = __fadd_rn(vmx[dd][2 * qn], vmx[dd][2 * qn + 1]);      // tokens c and c + 16
                s = __fadd_rn(s, ms_dpp<0xB1>(s));
                s = __fadd_rn(s, ms_dpp<0x4E>(s));
                s = __fadd_rn(s, ms_dpp<0x141>(s));
                s = __fadd_rn(s, ms_dpp<0x140>(s));
                score[dd][qn] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), 0)) * p.pre_inv_scale;
            }
#pragma unroll
        for (int dd = 0; dd < 2; ++dd) {
            const long long doc = (long long)(bt_first + t) * MS_DOCS_PER_TILE + 2 * wr + dd;
            const bool doc_ok = doc < p.nb_valid;
            const float cd = cd_s[dd];
#pragma unroll
            for (int n = 0; n < 2; ++n) {
                const long long q = q_first + n;
                const float err = rq[n] + cd;
                const bool pass = open_[n] || !(score[dd][n] < ntd[n] - err - 2.0e-6f * (atd[n] + err));
                if (pass && doc_ok && q < p.nq_valid && l == 0) {
                    const u32 pos = atomicAdd(p.cnt + q, 1u);
                    if (pos < (u32)cap) p.buf[q * cap + pos] = ((u64)__float_as_uint(-score[dd][n]) << 32) | (u32)doc;
                }
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA in flight when the workgroup leaves
}

// ------------------------------------------------------------------------------------------------
// The pre-screen for d <= 128, "pipelined" form (round 5; the default: option ms_form = 3).  Halving the document stream per FLOP
// (<8,1>) and halving the barriers (<8,2>) bought 1.4 % together: the ring kernels are bound by neither.  A tile is 4 096 MFMA cycles
// per SIMD and lives ~6 500: the rest is the EPILOGUE -- per wave and tile ~120 v_max, 16 lane swaps, 32 DPP adds, run by both waves of a
// SIMD at the same time with the matrix pipe idle.  So the epilogue moves UNDER the MFMAs:
//   * the ring's granule is a whole TILE (4 documents x all of d = 64 KiB; two stages), one barrier per tile: B(t) certifies that tile t
//     has landed and that tile t-1 has been read, tile t+1 is requested right behind it and has a whole tile time to arrive;
//   * a wave's work on a tile is split by DOCUMENT, not by k-slab: first all k-steps of its document 0 (row blocks 0-3: 16 nk MFMAs into
//     acc[0..3]), then all k-steps of document 1 (acc[4..7]) -- the same ascending-k chain per accumulator, hence the same estimates;
//   * software pipeline: while document 0 of tile t is being multiplied, the vector pipes reduce document 1 of tile t-1 (its accumulators
//     are not written again before the second half), and while document 1 of tile t is multiplied they reduce document 0 of tile t.  The
//     reductions are independent of the MFMAs in flight and the instruction stream is pinned as {1 MFMA, 2 VALU} (sched_group_barrier);
//     only the threshold test and the rare append sit behind each half.
// Same estimates, same keys, same lists as the ring kernels.  C5, same-box A/B (bench.py --workload c5): four-wave ring 0.482-0.485 of 2.5 PF
// (2 007-2 022 q/s pre-screened), <8,1> 0.487, <8,2> 0.493, this form 0.520-0.524 (2 149-2 162 q/s).  Tried on it and dropped, each +-0.3 %:
// s_setprio 1 / 0 for the two halves of a barrier interval; the next tile's LDS-DMA instructions spread through the MFMAs of document 0
// instead of issued right behind the barrier; strips of 32 instead of 64 tiles (16: -2 %, 8: -6 %).
// ------------------------------------------------------------------------------------------------
// NKT = k-slabs per tile (dp / 32 = 2 or 4: a template parameter so that a document's MFMAs are ONE basic block -- a run-time bound puts a branch
// behind every k-step, and sched_group_barrier interleaves inside a block only).
template <int NKT>
__global__ __launch_bounds__(512, 2) void nwk_maxsim_pre_tile8_kernel(const NwkTileParams p)
{
    constexpr int TM = 256, TQ = 256, NK = NKT, SLAB = TM * 64, STAGE = 4 * SLAB, B_SLAB = TQ * 64;      // 32-half slabs of 64-B rows; a stage = one tile
    constexpr int OFF_CD = 2 * STAGE;
    __shared__ __attribute__((aligned(16))) char smem[OFF_CD + MS_RING_MAX_STRIP * MS_DOCS_PER_TILE * 4];

    const TileCoord tc = tile_from_block(p);         // tc.qt: group of 8 queries, tc.bt: STRIP of document tiles
    if (!tc.ok) return;
    const int tid = threadIdx.x;
    const int l = tid & 63;
    const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wr = wu >> 2, wc = wu & 3;
    const int g = l >> 4, c16 = l & 15;
    const int T = p.tiles_per_wg > 1 ? p.tiles_per_wg : 1;         // <= MS_RING_MAX_STRIP (launcher)
    const int bt_first = p.bt0 + (tc.bt - p.bt0) * T;
    int n_here = p.bt0 + p.n_btiles_real - bt_first;
    if (n_here > T) n_here = T;
    const long long qrow0 = (long long)tc.qt * TQ;
    const int dp = p.dp;                             // halves per row = 32 NKT (the launcher picks the instance)
    constexpr int nk = NKT;
    const _Float16* docs = (const _Float16*)p.bp;
    const _Float16* qrys = (const _Float16*)p.qp;

    const unsigned voff = (unsigned)((l >> 2) * dp * 2 + (((l & 3) ^ ((0 - (l >> 4)) & 3)) << 4));      // the ring kernels' LDS image
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(qrys + qrow0 * dp), 0, 0x7FFFFFFF, 0x00020000);
    const int rowsA = wu * 32 * dp * 2, rowsB = wu * 32 * dp * 2, step16 = 16 * dp * 2;
    const int swz = (0 - (c16 >> 2)) & 3;
    const int offA0 = (wr * 128 + c16) * 64 + ((g ^ swz) << 4);      // + 1024 m + SLAB kt
    const int offB0 = (wc * 64 + c16) * 64 + ((g ^ swz) << 4);       // + 1024 n

    for (int kt = 0; kt < nk; ++kt)                                  // the 8 queries' tokens, through stage 0, once
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)(smem + kt * B_SLAB + wu * 2048 + i * 1024), 16, voff,
                                                     kt * 64 + rowsB + i * step16, 0, 0);
    const long long q_first = (long long)tc.qt * 8 + 2 * wc;
    const u64 tauq[2] = {p.tau[q_first], p.tau[q_first + 1]};
    const float rq[2] = {p.ms_rq[q_first], p.ms_rq[q_first + 1]};
    float cd_mine = 0.0f;
    {
        const long long doc = (long long)bt_first * MS_DOCS_PER_TILE + tid;
        if (tid < n_here * MS_DOCS_PER_TILE && doc < p.nb_valid) cd_mine = p.ms_cd[doc];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (tid < MS_RING_MAX_STRIP * MS_DOCS_PER_TILE) ((float*)(smem + OFF_CD))[tid] = cd_mine;
    __builtin_amdgcn_s_barrier();
    f16x8 bq[NK][4];
#pragma unroll
    for (int kt = 0; kt < NK; ++kt)
#pragma unroll
        for (int n = 0; n < 4; ++n)
            bq[kt][n] = (kt < nk) ? *(const f16x8*)(smem + kt * B_SLAB + offB0 + n * 1024) : f16x8{0, 0, 0, 0, 0, 0, 0, 0};
    float ntd[2], atd[2];
    bool open_[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
        open_[n] = tauq[n] == ~0ull;
        const float td = unsortable_f32((u32)(tauq[n] >> 32));
        ntd[n] = -td; atd[n] = fabsf(td);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                   // the query fragments are in registers: stage 0 is free

    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(docs + (long long)bt_first * TM * dp), 0, 0x7FFFFFFF, 0x00020000);
    const int tile_bytes = TM * dp * 2;
    // tile t of the strip -> stage t & 1: k-slab kt at + kt SLAB, this wave's 32 rows of each (2 instructions per slab, 2 nk per tile)
    auto issue_tile = [&](int t) __attribute__((always_inline)) {
        const int tt = t < n_here ? t : n_here - 1;                  // past the end: the last tile once more, into the stage nobody reads again (no branch)
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)(smem + (t & 1) * STAGE + kt * SLAB + wu * 2048 + i * 1024), 16, voff,
                                                         tt * tile_bytes + kt * 64 + rowsA + i * step16, 0, 0);
        }
    };
    issue_tile(0);

    const int cap = p.cap;
    const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
    f32x4 acc[8][4];
    // the MFMAs of one document (half dd of the wave's rows) of the tile in stage st
    auto mfma_doc = [&](int dd, int st) __attribute__((always_inline)) {
#pragma unroll
        for (int kt = 0; kt < NK; ++kt) {
            f16x8 a[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) a[m] = *(const f16x8*)(smem + st * STAGE + kt * SLAB + offA0 + (4 * dd + m) * 1024);
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = NWK_MFMA_SNAKE && (m & 1) ? 3 - j : j;          // snaked issue order (nwk_tile_f16y.h)
                    acc[4 * dd + m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[m], bq[kt][n], kt == 0 ? zero4 : acc[4 * dd + m][n], 0, 0, 0);
                }
        }
    };
    // reductions of one document: max over its 64 token rows per query-token column, sum over each query's 32 tokens -> two score estimates
    // (the document's bound term is read from LDS HERE, at the head of the pinned stream, so that the threshold test behind the phase finds it
    // in a register instead of starting an LDS round trip with both waves of the SIMD waiting)
    auto reduce_doc = [&](int t, int dd, float (&score)[2], float& cd) __attribute__((always_inline)) {
        cd = ((const float*)(smem + OFF_CD))[t * MS_DOCS_PER_TILE + 2 * wr + dd];
        float vmx[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            float v = fmaxf(acc[4 * dd][n][0], acc[4 * dd][n][1]);
            v = fmaxf(fmaxf(v, acc[4 * dd][n][2]), acc[4 * dd][n][3]);
#pragma unroll
            for (int m = 1; m < 4; ++m) {
                v = fmaxf(fmaxf(v, acc[4 * dd + m][n][0]), acc[4 * dd + m][n][1]);
                v = fmaxf(fmaxf(v, acc[4 * dd + m][n][2]), acc[4 * dd + m][n][3]);
            }
            vmx[n] = ms_rows_max(v);
        }
#pragma unroll
        for (int qn = 0; qn < 2; ++qn) {
            float s = __fadd_rn(vmx[2 * qn], vmx[2 * qn + 1]);      // tokens c and c + 16
            s = __fadd_rn(s, ms_dpp<0xB1>(s));
            s = __fadd_rn(s, ms_dpp<0x4E>(s));
            s = __fadd_rn(s, ms_dpp<0x141>(s));
            s = __fadd_rn(s, ms_dpp<0x140>(s));
            score[qn] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, s), 0)) * p.pre_inv_scale;
        }
    };
    // threshold test and append of one document's two estimates (tile t of the strip)
    auto list_doc = [&](int t, int dd, const float (&score)[2], float cd) __attribute__((always_inline)) {
        const long long doc = (long long)(bt_first + t) * MS_DOCS_PER_TILE + 2 * wr + dd;
        const bool doc_ok = doc < p.nb_valid;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const long long q = q_first + n;
            const float err = rq[n] + cd;
            const bool pass = open_[n] || !(score[n] < ntd[n] - err - 2.0e-6f * (atd[n] + err));
            if (pass && doc_ok && q < p.nq_valid && l == 0) {
                const u32 pos = atomicAdd(p.cnt + q, 1u);
                if (pos < (u32)cap) p.buf[q * cap + pos] = ((u64)__float_as_uint(-score[n]) << 32) | (u32)doc;
            }
        }
    };
    auto interleave = [&]() __attribute__((always_inline)) {          // {1 MFMA, 2 VALU} per MFMA of a document: the reductions ride in the MFMAs' shadow
#pragma unroll
        for (int i = 0; i < 16 * NKT; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
        }
    };
#define MS_FENCE() __builtin_amdgcn_sched_barrier(0)

    // tile 0, document 0: nothing to reduce yet
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                                     // B(0): tile 0 complete
    MS_FENCE();
    issue_tile(1);
    mfma_doc(0, 0);
    MS_FENCE();
    for (int t = 0; t < n_here; ++t) {
        float s0[2], s1[2], cd0, cd1;
        // document 1 of tile t  ||  reductions of document 0 of tile t
        mfma_doc(1, t & 1);
        reduce_doc(t, 0, s0, cd0);
        interleave();
        MS_FENCE();
        list_doc(t, 0, s0, cd0);
        MS_FENCE();
        if (t + 1 < n_here) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");  // this wave's part of tile t + 1 has landed (requested a tile time ago)
            __builtin_amdgcn_s_barrier();                             // B(t + 1): tile t + 1 complete, tile t read by everybody
            MS_FENCE();
            issue_tile(t + 2);
            // document 0 of tile t + 1  ||  reductions of document 1 of tile t
            mfma_doc(0, (t + 1) & 1);
            reduce_doc(t, 1, s1, cd1);
            interleave();
            MS_FENCE();
            list_doc(t, 1, s1, cd1);
            MS_FENCE();
        }
    }
    {
        float s1[2], cd1;                                             // the strip's last document 1: nothing left to hide it under
        reduce_doc(n_here - 1, 1, s1, cd1);
        list_doc(n_here - 1, 1, s1, cd1);
    }
#undef MS_FENCE
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // no LDS-DMA in flight when the workgroup leaves
}

// Rescoring in two launches.  (1) nwk_maxsim_prune_kernel, one wave per query: bounds est +- (R(q) + C(doc)) of the listed pairs,
// k-th smallest upper bound by bisection, pairs that cannot reach the top k become ~0, the slots of the others go to surv[q][.].
// (2) nwk_maxsim_rescore_kernel, grid (queries, slices): a wave takes every slices-th survivor of its query, two at a time (each
// query-token value fetched from LDS feeds two docs); lane j owns doc token j (64 lanes = 64 tokens), 32 fmaf chains per lane and doc
// (one per query token, ascending k over the prepared fp32 rows), max over the lanes by reduce-and-transpose, pairwise tree sum
// -- the arithmetic of nwk_maxsim_kernel, bit for bit.  Splitting a query's survivors over several waves matters: there are only
// ~2k queries but ~6k survivors per query, 4096 FMAs per lane each.
__global__ __launch_bounds__(256) void nwk_maxsim_prune_kernel(const NwkMsRescoreParams p)
{
    extern __shared__ __attribute__((aligned(16))) u32 ms_dyn[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const long long q = (long long)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * wpb + wv));
    u32* s_ub = ms_dyn + (size_t)wv * 2 * p.cap;
    u32* s_lb = s_ub + p.cap;
    const u32 start = p.cnt0[q];
    u32 end = p.cnt[q];
    if (end > (u32)p.cap) end = (u32)p.cap;
    u32* surv = p.surv + q * p.cap;
    if (start >= end) { if (lane == 0) p.nsurv[q] = 0; return; }
    u64* list = p.buf + q * p.cap;
    u32 nsurv = 0;
    if (p.prune && end >= (u32)p.k) {
        const float rq = p.rq[q];
        for (u32 e0 = 0; e0 < end; e0 += 64) {
            const u32 e = e0 + lane;
            if (e < end) {
                const u64 key = list[e];
                if (e < start) { s_ub[e] = (u32)(key >> 32); s_lb[e] = 0u; }
                else {
                    const float est = __uint_as_float((u32)(key >> 32));             // estimate of -score
                    const float err = (rq + p.cd[(u32)key]) * 1.0001f + 2.0e-6f * (fabsf(est) + 1.0f);
                    s_ub[e] = sortable_u32(est + err);
                    s_lb[e] = sortable_u32(est - err);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        u32 lo = 0u, hi = 0xFFFFFFFFu;
        while (lo < hi) {
            const u32 mid = lo + ((hi - lo) >> 1);
            u32 c = 0;
            for (u32 e0 = 0; e0 < end; e0 += 64) {
                const u32 e = e0 + lane;
                c += (u32)__popcll(__ballot(e < end && s_ub[e] <= mid));
            }
            if (c >= (u32)p.k) hi = mid; else lo = mid + 1u;
        }
        const u32 U = lo;
        for (u32 e0 = start; e0 < end; e0 += 64) {
            const u32 e = e0 + lane;
            const bool live = e < end;
            const bool keep = live && s_lb[e] <= U;
            const u64 mk = __ballot(keep);
            if (keep) surv[nsurv + (u32)__popcll(mk & ((1ull << lane) - 1ull))] = e;
            else if (live) list[e] = ~0ull;
            nsurv += (u32)__popcll(mk);
        }
    } else {
        nsurv = end - start;
        for (u32 i = lane; i < nsurv; i += 64) surv[i] = start + i;
    }
    if (lane == 0) {
        p.nsurv[q] = nsurv;
        if (NWK_DBG(p)) { atomicAdd(NWK_DBG(p), (unsigned long long)(end - start)); atomicAdd(NWK_DBG(p) + 1, (unsigned long long)nsurv); }
    }
}

// The pre-screen's invariant for one rescored (query, doc) pair, as in nwk_rescore_kernel: the listed estimate of -score against the
// canonical -score and the bound the pair was kept under (the expression of nwk_maxsim_prune_kernel).  Called by the lane that writes the key.
struct MsCheck {
    u32 n_chk = 0, n_bad = 0;
    float worst = 0.0f;
    __device__ __forceinline__ void pair(const NwkMsRescoreParams& p, float rq, u64 old_key, float neg_score)
    {
        const float est = __uint_as_float((u32)(old_key >> 32));
        const float err = (rq + p.cd[(u32)old_key]) * 1.0001f + 2.0e-6f * (fabsf(est) + 1.0f);
        const float dev = fabsf(neg_score - est);
        ++n_chk;
        if (!(dev <= err * p.chk_scale)) ++n_bad;
        const float ratio = dev / err;
        worst = fmaxf(worst, ratio == ratio ? ratio : __builtin_inff());
    }
    __device__ __forceinline__ void flush(const NwkMsRescoreParams& p) const
    {
        if (!p.chk || !n_chk) return;
        atomicAdd(p.chk, (u64)n_chk);
        if (n_bad) atomicAdd(p.chk + 1, (u64)n_bad);
        atomicMax((u32*)(p.chk + 2), __float_as_uint(worst));
    }
};

template <bool QLDS>      // the query's token rows are staged in LDS (a generic pointer would make every read a flat_load)
__global__ __launch_bounds__(256) void nwk_maxsim_rescore_kernel(const NwkMsRescoreParams p)
{
    extern __shared__ __attribute__((aligned(16))) u32 ms_dyn[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const long long q = (long long)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * wpb + wv));
    const u32 nsurv = p.nsurv[q];
    const u32 slice = blockIdx.y, slices = gridDim.y;
    if (slice * 2 >= nsurv) return;
    u64* list = p.buf + q * p.cap;
    const u32* surv = p.surv + q * p.cap;
    const float* __restrict__ qtok = p.qp + q * NWK_MS_TOKENS_PER_QUERY * p.dp;
    // the query's 32 token rows are read by every lane for every survivor: stage them in LDS once (uniform global loads do not
    // become scalar loads here -- the kernel also stores to global memory)
    float* s_q = (float*)ms_dyn + (size_t)wv * NWK_MS_TOKENS_PER_QUERY * p.dp;
    if (QLDS) {
        const int n4 = NWK_MS_TOKENS_PER_QUERY * p.dp / 4;
        for (int i = lane; i < n4; i += 64) *(f32x4*)(s_q + 4 * i) = *(const f32x4*)(qtok + 4 * i);
        __builtin_amdgcn_wave_barrier();
    }

    MsCheck chk;
    const float rq_mine = p.rq[q];
    for (u32 i = slice * 2; i < nsurv; i += slices * 2) {                               // survivors i and i+1 (wave-uniform)
        const u32 e0 = surv[i], e1 = (i + 1 < nsurv) ? surv[i + 1] : e0;
        const u64 old0 = list[e0], old1 = list[e1];
        const u32 doc0 = (u32)old0, doc1 = (u32)old1;
        const float* __restrict__ br0 = p.bp + ((long long)doc0 * NWK_MS_TOKENS_PER_DOC + lane) * p.dp;
        const float* __restrict__ br1 = p.bp + ((long long)doc1 * NWK_MS_TOKENS_PER_DOC + lane) * p.dp;
        float a0[NWK_MS_TOKENS_PER_QUERY], a1[NWK_MS_TOKENS_PER_QUERY];
#pragma unroll
        for (int t = 0; t < NWK_MS_TOKENS_PER_QUERY; ++t) { a0[t] = 0.0f; a1[t] = 0.0f; }
        // the doc rows are gathers from HBM (512-B stride between lanes): keep the loads two 8-k steps ahead of their use
        f32x4 pf[3][4];
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const int kk = (s2 * 8 < p.dp) ? s2 * 8 : 0;
            pf[s2][0] = *(const f32x4*)(br0 + kk); pf[s2][1] = *(const f32x4*)(br0 + kk + 4);
            pf[s2][2] = *(const f32x4*)(br1 + kk); pf[s2][3] = *(const f32x4*)(br1 + kk + 4);
        }
        for (int k0 = 0; k0 < p.dp; k0 += 8) {
            const f32x4 be0 = pf[0][0], bo0 = pf[0][1], be1 = pf[0][2], bo1 = pf[0][3];
            {
                const int kk = (k0 + 16 < p.dp) ? k0 + 16 : 0;                          // (past the end: a harmless reload of step 0)
                pf[2][0] = *(const f32x4*)(br0 + kk); pf[2][1] = *(const f32x4*)(br0 + kk + 4);
                pf[2][2] = *(const f32x4*)(br1 + kk); pf[2][3] = *(const f32x4*)(br1 + kk + 4);
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) { pf[0][c] = pf[1][c]; pf[1][c] = pf[2][c]; }
#pragma unroll
            for (int t = 0; t < NWK_MS_TOKENS_PER_QUERY; ++t) {
                f32x4 qe, qo;
                if (QLDS) { qe = *(const f32x4*)(s_q + t * p.dp + k0); qo = *(const f32x4*)(s_q + t * p.dp + k0 + 4); }
                else { qe = *(const f32x4*)(qtok + t * p.dp + k0); qo = *(const f32x4*)(qtok + t * p.dp + k0 + 4); }
#pragma unroll
                for (int j = 0; j < 4; ++j) {                                          // k ascending: k0+2j, k0+2j+1
                    a0[t] = __fmaf_rn(be0[j], qe[j], a0[t]);
                    a1[t] = __fmaf_rn(be1[j], qe[j], a1[t]);
                    a0[t] = __fmaf_rn(bo0[j], qo[j], a0[t]);
                    a1[t] = __fmaf_rn(bo1[j], qo[j], a1[t]);
                }
            }
        }
        // max over the doc's 64 tokens (= lanes) for each of the 32 query tokens, as a reduce-and-transpose: at stage s lanes whose
        // bit s is b keep the tokens whose bit s is b and hand the others to their xor-2^s partner: 16+8+4+2+1 exchanges instead of
        // 32 x 6, after which lane l holds token l & 31 reduced over its half-wave; one last exchange joins the halves.  (max is exact.)
#pragma unroll
        for (int sft = 0; sft < 5; ++sft) {
            const bool up = (lane >> sft) & 1;
#pragma unroll
            for (int j = 0; j < (16 >> sft); ++j) {
                const float l0 = a0[2 * j], h0 = a0[2 * j + 1], l1 = a1[2 * j], h1 = a1[2 * j + 1];
                a0[j] = fmaxf(up ? h0 : l0, __shfl_xor(up ? l0 : h0, 1 << sft));
                a1[j] = fmaxf(up ? h1 : l1, __shfl_xor(up ? l1 : h1, 1 << sft));
            }
        }
        float s0 = fmaxf(a0[0], __shfl_xor(a0[0], 32)), s1 = fmaxf(a1[0], __shfl_xor(a1[0], 32));
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {                                             // lane c = query token c & 31: the kernel's tree
            s0 = __fadd_rn(s0, __shfl_xor(s0, o));
            s1 = __fadd_rn(s1, __shfl_xor(s1, o));
        }
        if (lane == 0) {
            list[e0] = ((u64)sortable_u32(-s0) << 32) | doc0;
            chk.pair(p, rq_mine, old0, -s0);
            if (e1 != e0) { list[e1] = ((u64)sortable_u32(-s1) << 32) | doc1; chk.pair(p, rq_mine, old1, -s1); }
        }
    }
    if (lane == 0) chk.flush(p);
}


// The same rescoring on the matrix pipes.  Recomputing one (query, document) score is a 64 x 32 x d fp32 product -- the shape of
// v_mfma_f32_32x32x2_f32, whose result is the k-ordered fmaf chain, i.e. the very arithmetic nwk_maxsim_kernel defines as
// canonical (document tokens on the accumulator rows, query tokens on the lanes: the in-lane max and the butterfly sum below are
// that kernel's).  A workgroup = 4 waves on ONE query (its 32 token rows staged once in LDS, 528-B pitch: conflict-free
// ds_read_b128), each wave takes every (4 x slices)-th survivor; the document's token rows come straight from global memory in
// fragment shape (lane = row, 16 B = four k-steps of its half), four 8-k groups ahead of the MFMAs (two register sets).
// 128 MFMAs per pair instead of 4096 FMAs per lane: the kernel becomes a gather of 32 KiB per document.
__global__ __launch_bounds__(256) void nwk_maxsim_rescore_mfma_kernel(const NwkMsRescoreParams p)
{
    extern __shared__ __attribute__((aligned(16))) char msq_dyn[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const long long q = blockIdx.x;
    const u32 nsurv = p.nsurv[q];
    if (blockIdx.y * 4 >= nsurv) return;                             // workgroup-uniform
    const int dp = p.dp;                                             // multiple of 32
    const int pitch = dp * 4 + 16;
    const float* __restrict__ qtok = p.qp + q * NWK_MS_TOKENS_PER_QUERY * dp;
    const int per_row = dp / 4;
    for (int i = threadIdx.x; i < NWK_MS_TOKENS_PER_QUERY * per_row; i += 256) {
        const int row = i / per_row, c = i - row * per_row;
        *(f32x4*)(msq_dyn + row * pitch + c * 16) = *(const f32x4*)(qtok + row * dp + c * 4);
    }
    __syncthreads();
    const int h = lane >> 5, c32 = lane & 31;
    const char* sb = msq_dyn + c32 * pitch + h * 16;                 // + 32 g: chunk 2g + h of query token c32
    u64* list = p.buf + q * p.cap;
    const u32* surv = p.surv + q * p.cap;
    const int ngroups = dp / 8;                                      // multiple of 4
    MsCheck chk;
    const float rq_mine = p.rq[q];
    for (u32 i = blockIdx.y * 4 + wv; i < nsurv; i += gridDim.y * 4) {                   // wave-uniform
        const u32 e = surv[i];
        const u64 old = list[e];
        const u32 doc = (u32)old;
        const float* __restrict__ a0p = p.bp + ((long long)doc * NWK_MS_TOKENS_PER_DOC + c32) * dp + 4 * h;   // token rows 0..31
        const float* __restrict__ a1p = a0p + 32 * dp;                                                      // token rows 32..63
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; }
        f32x4 xa[4][2], xb[4][2];                                    // two sets of four groups x two row blocks
        auto load4 = [&](f32x4 (&dst)[4][2], int g0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) { dst[j][0] = *(const f32x4*)(a0p + 8 * (g0 + j)); dst[j][1] = *(const f32x4*)(a1p + 8 * (g0 + j)); }
        };
        auto mma4 = [&](const f32x4 (&src)[4][2], int g0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const f32x4 b = *(const f32x4*)(sb + 32 * (g0 + j));
#pragma unroll
                for (int t = 0; t < 4; ++t) {                          // k = 8 (g0 + j) + 2 t + h: ascending over the whole loop
                    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(src[j][0][t], b[t], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(src[j][1][t], b[t], acc1, 0, 0, 0);
                }
            }
        };
        load4(xa, 0);
        for (int g0 = 0; g0 < ngroups; g0 += 8) {
            if (g0 + 4 < ngroups) load4(xb, g0 + 4);
            mma4(xa, g0);
            if (g0 + 4 >= ngroups) break;
            if (g0 + 8 < ngroups) load4(xa, g0 + 8);
            mma4(xb, g0 + 4);
        }
        float v = acc0[0];
#pragma unroll
        for (int r = 1; r < 16; ++r) v = fmaxf(v, acc0[r]);
#pragma unroll
        for (int r = 0; r < 16; ++r) v = fmaxf(v, acc1[r]);
        const float sum = ms_token_sum(v);                           // max over the two half-waves, then the kernel's pairwise tree over the 32 query tokens
        if (lane == 0) { list[e] = ((u64)sortable_u32(-sum) << 32) | doc; chk.pair(p, rq_mine, old, -sum); }
    }
    if (lane == 0) chk.flush(p);
}

// ------------------------------------------------------------------------------------------------
// launchers (host side, called from nwknn_api.cpp)
// ------------------------------------------------------------------------------------------------
extern "C++" {

// MaxSim: p.n_qtiles counts groups of 4 queries (128 query-token rows), p.n_btiles tiles of 4 docs (256 doc-token rows)
hipError_t nwk_launch_maxsim(hipStream_t st, NwkTileParams p, bool dense)
{
    if (p.n_qtiles <= 0 || p.n_btiles <= 0) return hipSuccess;
    if (p.sq_shift < 0 || p.sq_shift > 6) return hipErrorInvalidValue;
    const int SQ = 1 << p.sq_shift, SB = 64 >> p.sq_shift;
    const long long n_super_q = (p.n_qtiles + SQ - 1) / SQ;
    p.n_super_b = (p.n_btiles + SB - 1) / SB;
    long long supers = n_super_q * p.n_super_b;
    supers = (supers + 7) / 8 * 8;
    const long long grid = supers * 64;
    if (grid > 0x7FFFFFFFll) return hipErrorInvalidValue;
    if (dense) hipLaunchKernelGGL((nwk_maxsim_kernel<true>), dim3((unsigned)grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL((nwk_maxsim_kernel<false>), dim3((unsigned)grid), dim3(256), 0, st, p);
    return hipGetLastError();
}

// MaxSim pre-screen: p.qp / p.bp are the scaled fp16 copies (dp halves), p.n_qtiles groups of 4 queries, p.n_btiles tiles of 4 docs
hipError_t nwk_launch_maxsim_pre(hipStream_t st, NwkTileParams p)
{
    if (p.n_qtiles <= 0 || p.n_btiles <= 0) return hipSuccess;
    if (p.sq_shift < 0 || p.sq_shift > 6) return hipErrorInvalidValue;
    const bool ring = p.dp <= 128 && p.tiles_per_wg != 1;       // d <= 128: query fragments fit the registers (tiles_per_wg = 1: the plain kernel)
    const bool wide = ring && p.f16_shape != 4;                 // 256 x 256 tiles of 8 queries (p.f16_shape = 4: option ms_waves, the four-wave build)
    if (wide) p.n_qtiles = (p.n_qtiles + 1) / 2;                // the query matrix is padded to 8 queries (nq_pad): the last tile's rows exist
    if (ring) {
        if (p.tiles_per_wg < 1) {                               // auto (option ms_strip = 0): the ring likes long strips (64: +3 % over 8 on C5)
            p.tiles_per_wg = MS_RING_MAX_STRIP;                 // as long as the launch still has a few workgroups for each of the 512 slots
            while (p.tiles_per_wg > 8 && (long long)((p.n_btiles + p.tiles_per_wg - 1) / p.tiles_per_wg) * p.n_qtiles < 2048) p.tiles_per_wg >>= 1;
        }
        if (p.tiles_per_wg > MS_RING_MAX_STRIP) p.tiles_per_wg = MS_RING_MAX_STRIP;
        p.n_btiles_real = p.n_btiles;
        p.n_btiles = (p.n_btiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
    }
    const int SQ = 1 << p.sq_shift, SB = 64 >> p.sq_shift;
    const long long n_super_q = (p.n_qtiles + SQ - 1) / SQ;
    p.n_super_b = (p.n_btiles + SB - 1) / SB;
    long long supers = n_super_q * p.n_super_b;
    supers = (supers + 7) / 8 * 8;
    const long long grid = supers * 64;
    if (grid > 0x7FFFFFFFll) return hipErrorInvalidValue;
    if (wide && p.agg == 3 && p.dp == 128) hipLaunchKernelGGL(nwk_maxsim_pre_tile8_kernel<4>, dim3((unsigned)grid), dim3(512), 0, st, p);      // p.agg: option ms_form
    else if (wide && p.agg == 3 && p.dp == 64) hipLaunchKernelGGL(nwk_maxsim_pre_tile8_kernel<2>, dim3((unsigned)grid), dim3(512), 0, st, p);
    else if (wide && p.agg == 2 && (p.dp / 32) % 2 == 0) hipLaunchKernelGGL((nwk_maxsim_pre_ring16_kernel<8, 2>), dim3((unsigned)grid), dim3(512), 0, st, p);
    else if (wide) hipLaunchKernelGGL((nwk_maxsim_pre_ring16_kernel<8, 1>), dim3((unsigned)grid), dim3(512), 0, st, p);
    else if (ring) hipLaunchKernelGGL((nwk_maxsim_pre_ring16_kernel<4, 1>), dim3((unsigned)grid), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(nwk_maxsim_pre_kernel, dim3((unsigned)grid), dim3(256), 0, st, p);
    return hipGetLastError();
}

hipError_t nwk_launch_maxsim_bounds(hipStream_t st, const float* norms, long long n_items, int tokens, float eps_half, float a_self, float cst,
                                    int is_doc, float* out)
{
    if (n_items <= 0) return hipSuccess;
    hipLaunchKernelGGL(nwk_maxsim_bounds_kernel, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, st, norms, n_items, tokens, eps_half, a_self,
                       cst, is_doc, out);
    return hipGetLastError();
}

hipError_t nwk_launch_maxsim_rescore(hipStream_t st, NwkMsRescoreParams p, long long nq_pad, int slices)
{
    if (nq_pad <= 0) return hipSuccess;
    {   // (1) prune: one wave per query
        const int waves = p.cap <= 1024 ? 4 : (p.cap <= 2048 ? 2 : 1);      // nq_pad is a multiple of 8
        const size_t lds = (size_t)waves * p.cap * 8 + 512;
        hipLaunchKernelGGL(nwk_maxsim_prune_kernel, dim3((unsigned)((nq_pad + waves - 1) / waves)), dim3(64 * waves), lds, st, p);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    // (2) rescore.  On the matrix pipes when the query's padded token rows fit LDS comfortably (d <= 256): workgroups of four
    //     waves per (query, slice)
    const size_t mfma_lds = (size_t)NWK_MS_TOKENS_PER_QUERY * ((size_t)p.dp * 4 + 16);
    if (mfma_lds <= 40 * 1024 && !p.rescore_valu) {
        const int s4 = std::max(1, (slices + 3) / 4);
        hipLaunchKernelGGL(nwk_maxsim_rescore_mfma_kernel, dim3((unsigned)nq_pad, (unsigned)s4), dim3(256), mfma_lds, st, p);
        return hipGetLastError();
    }
    //     else on the vector pipes: (query, slice) waves; the query's 32 token rows in LDS when they fit (d = 128: 16 KiB per wave)
    const size_t qbytes = (size_t)NWK_MS_TOKENS_PER_QUERY * p.dp * 4;
    const int waves = 2;
    p.q_in_lds = (waves * qbytes <= 64 * 1024) ? 1 : 0;
    const size_t lds = p.q_in_lds ? waves * qbytes : 16;
    if (p.q_in_lds)
        hipLaunchKernelGGL((nwk_maxsim_rescore_kernel<true>), dim3((unsigned)((nq_pad + waves - 1) / waves), (unsigned)slices), dim3(64 * waves), lds, st, p);
    else
        hipLaunchKernelGGL((nwk_maxsim_rescore_kernel<false>), dim3((unsigned)((nq_pad + waves - 1) / waves), (unsigned)slices), dim3(64 * waves), lds, st, p);
    return hipGetLastError();
}

}  // extern "C++"
