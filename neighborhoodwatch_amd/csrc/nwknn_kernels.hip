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
//   * option "prescreen": the filtered chunks run on the fp16 MFMA kernel (v_mfma_f32_16x16x32_f16) over scaled fp16 copies under a
//     proven additive error bound, and nwk_rescore_kernel replaces every listed pair by its canonical fp32 key -- the same bits
//     out, several times faster; fp16 INPUT data takes the same route by default ("exact fp16 mode"), MaxSim has its own
//     pre-screen / prune / rescore kernels.
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
// prep, tiled form (the default for row-major input): a lane-per-row kernel touches 64 different rows per load
// instruction, 16 B from each, i.e. four requests per 64-B granule (2 TB/s on fp32, 0.9 TB/s on fp16).  Here a block owns
// 256 rows and walks K in 64-byte steps: every global load and store instruction moves whole 64-B row segments (four lanes
// per row), the row <-> lane transposition goes through LDS (80-B row pitch: conflict-free for ds_read/write_b128), and the
// lane that owns a row still runs the norm chain in ascending k, so the results are bit-identical to nwk_prep_kernel.
// TIN = float | _Float16; OUT16 selects the fp16 layout.  Requires a 16-B aligned source and d*sizeof(TIN) % 16 == 0.
// ------------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <typename TIN, bool OUT16>
__global__ __launch_bounds__(256) void nwk_prep_tiled_kernel(const TIN* __restrict__ raw, long long rows, int d, long long rows_pad, int dp,
                                                              void* __restrict__ prepared, float* __restrict__ norms)
{
    constexpr int EIN = 16 / (int)sizeof(TIN);            // elements per 16-B piece of input
    constexpr int KS = 4 * EIN;                           // elements per step (64 B of input per row)
    constexpr int OES = OUT16 ? 2 : 4;                    // output element size
    constexpr int OB = KS * OES;                          // output bytes per row per step: 64 (f32->f32, f16->f16) or 32 (f32->f16)
    constexpr int OP = OB / 16;                           // 16-B output pieces per row per step
    constexpr int PITCH = 80;
    __shared__ __attribute__((aligned(16))) char tile_in[256 * PITCH];
    __shared__ __attribute__((aligned(16))) char tile_out[256 * PITCH];

    const int tid = threadIdx.x;
    const long long r0 = (long long)blockIdx.x * 256;
    const size_t in_row_bytes = (size_t)d * sizeof(TIN);
    const size_t out_row_bytes = (size_t)dp * OES;
    float acc = 0.0f;

    for (int k0 = 0; k0 < dp; k0 += KS) {
        // load: 4 lanes per row, 64 rows per pass
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            const int row = pass * 64 + (tid >> 2), seg = tid & 3;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (r0 + row < rows && k0 + seg * EIN < d)
                v = *(const u32x4*)((const char*)raw + (size_t)(r0 + row) * in_row_bytes + (size_t)k0 * sizeof(TIN) + seg * 16);
            *(u32x4*)(tile_in + row * PITCH + seg * 16) = v;
        }
        __syncthreads();
        // the lane that owns row `tid`: ascending-k norm chain + output layout for this step
        {
            float x[KS];
#pragma unroll
            for (int piece = 0; piece < 4; ++piece) {
                const u32x4 v = *(const u32x4*)(tile_in + tid * PITCH + piece * 16);
                if (sizeof(TIN) == 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[piece * EIN + j] = __uint_as_float(v[j]);
                } else {
                    const f16x8 hv = __builtin_bit_cast(f16x8, v);
#pragma unroll
                    for (int j = 0; j < 8; ++j) x[piece * EIN + j] = (float)hv[j];
                }
            }
            if (OUT16) {
                _Float16 hx[KS];
#pragma unroll
                for (int j = 0; j < KS; ++j) {
                    hx[j] = (_Float16)x[j];
                    const float xf = (float)hx[j];
                    acc = __fmaf_rn(xf, xf, acc);
                }
#pragma unroll
                for (int piece = 0; piece < OP; ++piece) {
                    f16x8 hv;
#pragma unroll
                    for (int j = 0; j < 8; ++j) hv[j] = hx[piece * 8 + j];
                    *(f16x8*)(tile_out + tid * PITCH + piece * 16) = hv;
                }
            } else {
#pragma unroll
                for (int j = 0; j < KS; ++j) acc = __fmaf_rn(x[j], x[j], acc);
#pragma unroll
                for (int g = 0; g < KS / 8; ++g) {            // [0 2 4 6 | 1 3 5 7] inside every group of 8
                    *(f32x4*)(tile_out + tid * PITCH + g * 32) = (f32x4){x[8 * g], x[8 * g + 2], x[8 * g + 4], x[8 * g + 6]};
                    *(f32x4*)(tile_out + tid * PITCH + g * 32 + 16) = (f32x4){x[8 * g + 1], x[8 * g + 3], x[8 * g + 5], x[8 * g + 7]};
                }
            }
        }
        __syncthreads();
        // store: OP lanes per row
#pragma unroll
        for (int pass = 0; pass < OP; ++pass) {
            const int idx = pass * 256 + tid;
            const int row = idx / OP, seg = idx % OP;
            if (r0 + row < rows_pad)
                *(u32x4*)((char*)prepared + (size_t)(r0 + row) * out_row_bytes + (size_t)k0 * OES + seg * 16) =
                    *(const u32x4*)(tile_out + row * PITCH + seg * 16);
        }
    }
    if (r0 + tid < rows_pad) norms[r0 + tid] = acc;
}

// ------------------------------------------------------------------------------------------------
// prep from an arbitrary strided source: element (r, k) at raw[r*rs + k*ks].  Used for COLUMN-major input (rs = 1,
// ks = rows of the staged slab): parquet stores one column per dimension (generate_dataset.py:219-262) and the reference
// pays a device transpose per tile for it (cu_knn.py:211,235).  One lane owns one row, so for every k a wave reads 64
// consecutive elements of column k -- the columnar layout is the coalesced one for this kernel -- and the norm chain and
// the prepared row come out of the same pass.  TIN = float / double / _Float16 (values are converted with one rounding,
// as ``.astype(np.float32)`` does at cu_knn.py:243); OUT16 selects the fp16 layout of the fp16 compute path.
// ------------------------------------------------------------------------------------------------
template <typename TIN, bool OUT16>
__global__ __launch_bounds__(256) void nwk_prep_strided_kernel(const TIN* __restrict__ raw, long long rs, long long ks, long long rows, int d,
                                                                long long rows_pad, int dp, void* __restrict__ prepared, float* __restrict__ norms)
{
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows_pad) return;
    const bool live = r < rows;
    const TIN* src = raw + r * rs;
    float acc = 0.0f;
    for (int k0 = 0; k0 < dp; k0 += 8) {
        float x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = (live && k0 + j < d) ? (float)src[(long long)(k0 + j) * ks] : 0.0f;
        if (OUT16) {
            f16x8 hv;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                hv[j] = (_Float16)x[j];
                const float xf = (float)hv[j];
                acc = __fmaf_rn(xf, xf, acc);
            }
            *(f16x8*)((_Float16*)prepared + r * dp + k0) = hv;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __fmaf_rn(x[j], x[j], acc);
            float* dst = (float*)prepared + r * dp + k0;
            *(f32x4*)(dst) = (f32x4){x[0], x[2], x[4], x[6]};
            *(f32x4*)(dst + 4) = (f32x4){x[1], x[3], x[5], x[7]};
        }
    }
    norms[r] = acc;
}

// ------------------------------------------------------------------------------------------------
// fp16 compute path (BASELINE config 4): generator and prep.
//   * synthetic f16 values: the top 11 bits of the same 64-bit draw, as a signed integer times 2^-10 -- exactly
//     representable in fp16, in [-1, 1)
//   * prep: raw row-major (f32 or f16) -> fp16 [rows_pad][dp], dp = d rounded up to 64 halves (one 128-B LDS row per
//     K slab), natural k order (lane half h of v_mfma_f32_32x32x16_f16 holds k = 8h..8h+7: one ds_read_b128);
//     f32 input is rounded to nearest-even; the norm is the fp32 fmaf chain over the ROUNDED values, so every term of
//     |q|^2 + |b|^2 - 2 q.b refers to the same vectors.
// ------------------------------------------------------------------------------------------------

__global__ __launch_bounds__(256) void nwk_gen_f16_kernel(u64 seed, long long row0, long long rows, int d, _Float16* __restrict__ out)
{
    const long long total = rows * (long long)d;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / d;
        const int c = (int)(i - r * d);
        const u64 rk = fmix64(seed + (u64)(row0 + r) * 0x9E3779B97F4A7C15ull);
        const u64 u = fmix64(rk + (u64)(c + 1) * 0xD1342543DE82EF95ull);
        const int v = (int)(u >> 53) - (1 << 10);
        out[i] = (_Float16)((float)v * (1.0f / 1024.0f));
    }
}

template <typename TIN>
__global__ __launch_bounds__(256) void nwk_prep_f16_kernel(const TIN* __restrict__ raw, long long rows, int d, long long rows_pad, int dp,
                                                            _Float16* __restrict__ prepared, float* __restrict__ norms)
{
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r >= rows_pad) return;
    _Float16* dst = prepared + r * dp;
    if (r >= rows) {
        const f16x8 z = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int k = 0; k < dp; k += 8) *(f16x8*)(dst + k) = z;
        norms[r] = 0.0f;
        return;
    }
    const TIN* src = raw + r * (long long)d;
    float acc = 0.0f;
    for (int k0 = 0; k0 < dp; k0 += 8) {
        f16x8 hv;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const _Float16 x = (k0 + j < d) ? (_Float16)src[k0 + j] : (_Float16)0;
            hv[j] = x;
            const float xf = (float)x;
            acc = __fmaf_rn(xf, xf, acc);
        }
        *(f16x8*)(dst + k0) = hv;
    }
    norms[r] = acc;
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
    // XCD-aware mapping: blocks b and b+8 share an XCD (and its L2); each XCD works through super-tiles of 64 tiles
    // (SQ query tiles x SB base tiles, SQ*SB = 64, SQ = 1 << p.sq_shift) so that its co-resident workgroups share operands.
    const int bid = blockIdx.x;
    const int xcd = bid & 7, local = bid >> 3;
    const long long super = (long long)(local >> 6) * 8 + xcd;
    const int within = local & 63;
    const int sb_shift = 6 - p.sq_shift;
    const int sq = (int)(super / p.n_super_b), sb = (int)(super % p.n_super_b);
    const int qi = (sq << p.sq_shift) + (within >> sb_shift), bi = (sb << sb_shift) + (within & ((1 << sb_shift) - 1));
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

// Pre-filter.  key < tau requires dist <= tau_d (tau_d = distance part of tau).  For the two additive metrics that is a
// lower bound on the raw dot product that splits into a row part and a column part:
//     sqeuclidean   : qn + bn - 2 dot <= tau_d   <=>  dot >= (qn - tau_d)/2 + bn/2
//     one_minus_dot : 1 - dot <= tau_d           <=>  dot >= 1 - tau_d
// so the common case costs one add and one compare per accumulator.  The row part is lowered by a margin that covers
// every rounding of the canonical fp32 evaluation (a few ulps of the operands), which makes the test a strict superset of
// the exact one; whatever passes it is re-evaluated exactly.  Cosine has no additive split and pre-tests on the distance.
template <int METRIC>
__device__ __forceinline__ float prefilter_row(float qn, u64 tau)
{
    if (tau == ~0ull) return -__builtin_inff();                      // fewer than k listed: everything may pass
    const float td = unsortable_f32((u32)(tau >> 32));
    if (METRIC == NWKNN_METRIC_SQEUCLIDEAN) return 0.5f * (qn - td) - 1.0e-6f * (fabsf(qn) + fabsf(td)) - 1.0e-30f;
    if (METRIC == NWKNN_METRIC_ONE_MINUS_DOT) return (1.0f - td) - 1.0e-6f * (1.0f + fabsf(td));
    return td;                                                        // cosine: compared against the distance itself
}
template <int METRIC>
__device__ __forceinline__ float prefilter_col(float bn)
{
    if (METRIC == NWKNN_METRIC_SQEUCLIDEAN) return 0.5f * bn - 1.0e-6f * fabsf(bn);
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

template <int METRIC, bool DENSE, int MT, int PRE = 0>
__device__ __forceinline__ void tile_epilogue(const NwkTileParams& p, const f32x16 (&acc)[MT][2], const float* s_qn, const u64* s_tau,
                                              const float* s_pre, const float (&bnv)[2], const u32 (&colv)[2], const bool (&colok)[2],
                                              long long q0, int wrow0, int h, int c32)
{
    const int cap = p.cap;
    float cpre[2];
#pragma unroll
    for (int n = 0; n < 2; ++n)
        cpre[n] = colok[n] ? (PRE ? prescreen_col<METRIC>(bnv[n], p) : prefilter_col<METRIC>(bnv[n])) : __builtin_inff();   // padding columns never pass
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
                    const float dist = distance_epilogue<METRIC>(acc[m][n][r], qn4[r >> 2][r & 3], bnv[n]);
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
                for (int r = 0; r < 16; ++r) {
                    const int row = rb + (r & 3) + 8 * (r >> 2);
                    const float dist = distance_epilogue<METRIC>(acc[m][n][r], s_qn[row], bnv[n]);
                    maybe |= !(dist > pre4[r >> 2][r & 3]);
                }
                maybe &= colok[n];
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
                    const float dist = distance_epilogue<METRIC>(acc[m][n][4 * j + i], qn4[i], bnv[n]);
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
        s_qn[tid] = my_qn;
        if (!DENSE) {
            s_tau[tid] = my_tau;
            float pre = PRE ? prescreen_row<METRIC>(my_qn, my_tau, p) : prefilter_row<METRIC>(my_qn, my_tau);
            // padding query rows (all zero) must list nothing: with one_minus_dot / cosine they tie with every base row, which
            // the pre-screen cannot break by index and would list wholesale.  (The direct cosine test is on the distance: -inf.)
            if (p.nq_valid > 0 && q0 + tid >= p.nq_valid)
                pre = (METRIC == NWKNN_METRIC_COSINE && !PRE) ? -__builtin_inff() : __builtin_inff();
            s_pre[tid] = pre;
        }
    }
}

// DIAG (timing experiments only, results are wrong when != 0): 1 = no LDS fragment reads in the loop, 2 = no DMA after
// slab 0, 4 = no barriers, 8 = no B-operand DMA after slab 0.  The product always launches DIAG = 0.
template <int METRIC, bool DENSE, int TM, int BKT, int DIAG = 0>
__global__ __launch_bounds__(256, 2) void nwk_tile_kernel(const NwkTileParams p)
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
    if (p.dbg) { st0 = __builtin_amdgcn_s_memtime(); rt0 = __builtin_amdgcn_s_memrealtime(); }
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

    // ---- LDS-DMA sources.  Wave w fills rows [w*IA*RPI, (w+1)*IA*RPI) of the A tile and [w*IB*RPI, ...) of the B tile.
    unsigned voffA[4], voffB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ra = (w * IA + i) * RPI + (l / CH), rb = (w * IB + i) * RPI + (l / CH);
        voffA[i] = (unsigned)(ra * dp + (((l % CH) ^ ((ra >> RSH) & (CH - 1))) << 2)) * 4u;
        voffB[i] = (unsigned)(rb * dp + (((l % CH) ^ ((rb >> RSH) & (CH - 1))) << 2)) * 4u;
    }
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)(p.qp + q0 * dp), 0, 0x7FFFFFFF, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bp + b0 * dp), 0, 0x7FFFFFFF, 0x00020000);
    const int ldsA = __builtin_amdgcn_readfirstlane(w * IA * 1024);
    const int ldsB = __builtin_amdgcn_readfirstlane(A_BYTES + w * IB * 1024);

    auto issue_stage = [&](int stage, int kt) {
        char* base = smem + stage * STAGE;
        const int soff = kt * ROWB;
#pragma unroll
        for (int i = 0; i < IA; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (LDS_AS void*)(base + ldsA + i * 1024), 16, voffA[i], soff, 0, 0);
#pragma unroll
        for (int i = 0; i < IB; ++i) {
            if ((DIAG & 8) && kt > 0) continue;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, (LDS_AS void*)(base + ldsB + i * 1024), 16, voffB[i], soff, 0, 0);
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
    if (p.dbg) st1 = __builtin_amdgcn_s_memtime();
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

    if (p.dbg) st2 = __builtin_amdgcn_s_memtime();
    // ---- epilogue: accumulators -> keys -> filter -> candidate lists (staging reuses the idle operand stages)
    stage_query_side<METRIC, DENSE>(smem, TM, tid, my_qn, my_tau, p, q0);
    __syncthreads();
    tile_epilogue<METRIC, DENSE, MT>(p, acc, (const float*)smem, (const u64*)(smem + TM * 4), (const float*)(smem + TM * 12), bnv, colv, colok,
                                     q0, wr * (TM / 2), h, c32);
    if (p.dbg && tid == 0) {
        unsigned long long* o = p.dbg + (size_t)blockIdx.x * 8;
        o[0] = st0; o[1] = st1; o[2] = st2; o[3] = __builtin_amdgcn_s_memtime(); o[4] = rt0; o[5] = __builtin_amdgcn_s_memrealtime();
        o[6] = (unsigned long long)__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) << 32      // HW_REG_XCC_ID[3:0]
             | (unsigned long long)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));           // HW_REG_HW_ID
        o[7] = 1;
    }
}

// ------------------------------------------------------------------------------------------------
// fp16 distance tile: 256 queries x 256 base rows per workgroup of 8 waves (2 x 4; a wave owns 128 x 64 = 4 x 2
// accumulators of v_mfma_f32_32x32x16_f16, fp32 accumulate), K in 64-half (128-B) slabs through two 64 KiB LDS stages.
// fp16 MFMA is 16x the fp32 rate, so the operand stream is the problem: per DMA instruction this shape has
// 256*256/(256+256) = 128 cycles of MFMA work, the most a 160 KiB LDS allows with 2 stages.  Same LDS image, swizzle,
// DMA form, epilogue and candidate-list logic as the fp32 kernel.  The accumulation order inside one MFMA (16 k's) is
// not specified by the ISA, so this path is checked against the float64 oracle with a tolerance, not bit for bit.
// ------------------------------------------------------------------------------------------------
#define H_TM 256
#define H_TN 256
#define H_BK 64                       // halves per slab = 128-B rows
#define H_STAGE (2 * 256 * 128)       // A 32 KiB + B 32 KiB

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
template <int METRIC, bool DENSE, int PRE>
__device__ __forceinline__ void tile_epilogue16(const NwkTileParams& p, const f32x4 (&acc)[8][4], const float* s_qn, const u64* s_tau,
                                                const float* s_pre, const float (&bnv)[4], const u32 (&colv)[4], const bool (&colok)[4],
                                                const float (&cpre)[4], long long q0, int wrow0, int g, int c16)
{
    const int cap = p.cap;
#pragma unroll
    for (int m = 0; m < 8; ++m) {
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
                        const float dist = distance_epilogue<METRIC>(acc[m][n][i], qn4[i], bnv[n]);
                        hi = (METRIC == NWKNN_METRIC_SQEUCLIDEAN) ? (__float_as_uint(dist) | 0x80000000u) : sortable_u32(dist);
                    }
                    const long long slot = (long long)(colv[n] - (u32)p.dense_col0);
                    p.buf[(q0 + rb + i) * cap + slot] = colok[n] ? (((u64)hi << 32) | colv[n]) : ~0ull;
                }
            continue;
        }
        const f32x4 pre4 = *(const f32x4*)(s_pre + rb);
        bool maybe = false;
        if (METRIC == NWKNN_METRIC_COSINE && !PRE) {
            const f32x4 qn4 = *(const f32x4*)(s_qn + rb);
#pragma unroll
            for (int n = 0; n < 4; ++n) {
                bool mn = false;
#pragma unroll
                for (int i = 0; i < 4; ++i) mn |= !(distance_epilogue<METRIC>(acc[m][n][i], qn4[i], bnv[n]) > pre4[i]);
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
                const float dist = distance_epilogue<METRIC>(acc[m][n][i], qn4[i], bnv[n]);
                const u32 sd = (METRIC == NWKNN_METRIC_SQEUCLIDEAN) ? (__float_as_uint(dist) | 0x80000000u) : sortable_u32(dist);
                key[i] = ((u64)sd << 32) | colv[n];
                const u64 tau = (i & 2) ? th[i & 1] : tl[i & 1];
                pass[i] = (key[i] < tau) & colok[n];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const u64 mk = __ballot(pass[i]);
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

template <int METRIC, bool DENSE, int PRE = 0>
__global__ __launch_bounds__(512, 2) void nwk_tile_kernel_f16s(const NwkTileParams p)
{
    __shared__ __attribute__((aligned(16))) char smem[2 * H_STAGE + H_TM * 16];
    char* const epi = smem + 2 * H_STAGE;

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
    tile_epilogue16<METRIC, DENSE, PRE>(p, acc, (const float*)epi, (const u64*)(epi + H_TM * 4), (const float*)(epi + H_TM * 12), bnv, colv,
                                        colok, cpre, q0, wr * 128, g_here, c_here);
  }
#undef NWK_SPHASE
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

    // ---- epilogue: 2 docs x 2 queries per wave
    const int cap = p.cap;
#pragma unroll
    for (int dd = 0; dd < 2; ++dd) {
        const long long doc = (long long)tc.bt * MS_DOCS_PER_TILE + 2 * wr + dd;
        const bool doc_ok = doc < p.nb_valid;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            float v = acc[2 * dd][n][0];
#pragma unroll
            for (int r = 1; r < 16; ++r) v = fmaxf(v, acc[2 * dd][n][r]);
#pragma unroll
            for (int r = 0; r < 16; ++r) v = fmaxf(v, acc[2 * dd + 1][n][r]);
            v = fmaxf(v, __shfl_xor(v, 32));                 // the two half-waves hold the other 32 token rows of this doc
            float sum = v;                                    // lane c32 = query token c32: balanced pairwise tree over tokens
#pragma unroll
            for (int o = 1; o < 32; o <<= 1) sum = __fadd_rn(sum, __shfl_xor(sum, o));
            const long long q = q_first + n;
            const u64 key = ((u64)sortable_u32(-sum) << 32) | (u32)doc;
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
        if (p.dbg) { atomicAdd(p.dbg, (unsigned long long)(end - start)); atomicAdd(p.dbg + 1, (unsigned long long)nsurv); }
    }
}

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

    for (u32 i = slice * 2; i < nsurv; i += slices * 2) {                               // survivors i and i+1 (wave-uniform)
        const u32 e0 = surv[i], e1 = (i + 1 < nsurv) ? surv[i + 1] : e0;
        const u32 doc0 = (u32)list[e0], doc1 = (u32)list[e1];
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
            if (e1 != e0) list[e1] = ((u64)sortable_u32(-s1) << 32) | doc1;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fp16 pre-screen support (exact fp32 search, option "prescreen"): the fp16 MFMA kernel above runs over SCALED fp16 copies
// of the prepared fp32 operands with the loosened additive filter (PRE = 1) and lists every pair that could be among the
// k nearest; the rescore kernel then replaces each listed pair by its canonical fp32 key, so compaction, tau and the final
// result are bit-for-bit those of the direct fp32 path.
//   * maxabs      : max |x| over a prepared fp32 matrix, as float bits through atomicMax (NaN/Inf show up as >= 0x7F800000)
//   * half_copy   : prepared fp32 [rows_pad][dp32] (k order [0 2 4 6 | 1 3 5 7] per group of 8) -> fp16 [rows_pad][dp16] in natural
//                   order, multiplied by a power of two that puts max |x| in [2^14, 2^15): no overflow, and whatever falls below
//                   the fp16 normal range is < 2^-28 of the largest element (the absolute term of the bound)
//   * rescore     : one wave per query, one lane per listed pair: ascending-k fmaf chain over the prepared fp32 rows (the same
//                   chain the fp32 MFMA accumulator performs), canonical epilogue, 64-bit key written back in place
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void nwk_maxabs_kernel(const float* __restrict__ x, long long n4, u32* __restrict__ out_bits)
{
    u32 m = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const u32x4 v = *(const u32x4*)(x + 4 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) { const u32 a = v[j] & 0x7FFFFFFFu; m = a > m ? a : m; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const u32 o = (u32)__shfl_xor((int)m, off); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out_bits, m);
}

// max |bits| over an fp16 matrix (n8 groups of 8 halves): >= 0x7C00 means Inf/NaN somewhere
__global__ __launch_bounds__(256) void nwk_maxabs_f16_kernel(const _Float16* __restrict__ x, long long n8, u32* __restrict__ out_bits)
{
    u32 m = 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
        const u32x4 v = *(const u32x4*)(x + 8 * i);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const u32 lo = v[j] & 0x7FFFu, hi = (v[j] >> 16) & 0x7FFFu;
            m = lo > m ? lo : m; m = hi > m ? hi : m;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const u32 o = (u32)__shfl_xor((int)m, off); m = o > m ? o : m; }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out_bits, m);
}

__global__ __launch_bounds__(256) void nwk_half_copy_kernel(const float* __restrict__ prepared, long long rows_pad, int dp32, int dp16,
                                                             float scale0, const float* __restrict__ norms, _Float16* __restrict__ out)
{
    const int g16 = dp16 >> 3;
    const long long total = rows_pad * g16;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / g16;
        const int k0 = (int)(i - r * g16) << 3;
        f16x8 hv = {0, 0, 0, 0, 0, 0, 0, 0};
        float scale = scale0;
        if (norms) {                                       // cosine: unit rows (degenerate norms: an all-zero copy, see cosine_norm_ok)
            const float n = norms[r];
            scale = cosine_norm_ok(n) ? scale0 * (1.0f / sqrtf(n)) : 0.0f;
        }
        if (k0 < dp32 && scale != 0.0f) {
            const f32x4 ev = *(const f32x4*)(prepared + r * dp32 + k0), od = *(const f32x4*)(prepared + r * dp32 + k0 + 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                hv[2 * j] = (_Float16)(ev[j] * scale);
                hv[2 * j + 1] = (_Float16)(od[j] * scale);
            }
        }
        *(f16x8*)(out + r * dp16 + k0) = hv;
    }
}

// fp16 rows -> scaled (cosine: unit-row) fp16 copy, natural k order on both sides (exact fp16 mode, cosine)
__global__ __launch_bounds__(256) void nwk_half_copy16_kernel(const _Float16* __restrict__ rows16, long long rows_pad, int dp, float scale0,
                                                               const float* __restrict__ norms, _Float16* __restrict__ out)
{
    const int g16 = dp >> 3;
    const long long total = rows_pad * g16;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long r = i / g16;
        float scale = scale0;
        if (norms) {
            const float n = norms[r];
            scale = cosine_norm_ok(n) ? scale0 * (1.0f / sqrtf(n)) : 0.0f;
        }
        const f16x8 v = *(const f16x8*)(rows16 + 8 * i);
        f16x8 hv;
#pragma unroll
        for (int j = 0; j < 8; ++j) hv[j] = (_Float16)((float)v[j] * scale);
        *(f16x8*)(out + 8 * i) = hv;
    }
}

// One wave per query (blockDim = 64 * waves, dynamic LDS = waves * cap * 8 bytes).  The keys the pre-screen listed carry the
// fp16 ESTIMATE of the distance; with the error bound e(q,b) the exact value lies in [est - e, est + e].  Phases:
//   1. bounds of every list entry into LDS (entries of the compacted prefix [0, cnt0) are exact: lb = ub = their distance);
//   2. U = k-th smallest UPPER bound (bisection over the sortable 32-bit space, ballot counts): at least k entries are <= U,
//      so the k-th key after this chunk is <= U and every candidate whose LOWER bound is > U can be dropped unrescored;
//   3. the survivors are packed and rescored 64 at a time: ascending-k fmaf chain over the prepared fp32 rows (the chain the
//      fp32 MFMA accumulator performs), canonical epilogue, 64-bit key written back in place; dropped entries become ~0
//      (the sentinel compaction already sorts last).
template <int METRIC, bool ROWS16>
__global__ __launch_bounds__(256) void nwk_rescore_kernel(const NwkRescoreParams p)
{
    extern __shared__ __attribute__((aligned(16))) u32 rs_dyn[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, wpb = blockDim.x >> 6;
    const long long q = (long long)__builtin_amdgcn_readfirstlane((int)(blockIdx.x * wpb + wv));
    u32* s_ub = rs_dyn + (size_t)wv * 2 * p.cap;          // later: the packed survivor slots
    u32* s_lb = s_ub + p.cap;
    const u32 start = p.cnt0[q];
    u32 end = p.cnt[q];
    if (end > (u32)p.cap) end = (u32)p.cap;
    if (start >= end) return;
    const float* __restrict__ qrow = ROWS16 ? nullptr : p.qp + q * p.dp;
    const _Float16* __restrict__ qrow16 = ROWS16 ? (const _Float16*)p.qp + q * p.dp : nullptr;      // exact fp16 mode: the rows ARE the data
    const float qn = p.qn[q];
    u64* list = p.buf + q * p.cap;

    u32 nsurv = 0;
    if (p.prune && end >= (u32)p.k) {
        const float row_err = p.eps_half * qn + p.ab * __builtin_amdgcn_sqrtf(qn) + p.cst;
        for (u32 e0 = 0; e0 < end; e0 += 64) {
            const u32 e = e0 + lane;
            if (e < end) {
                const u64 key = list[e];
                if (e < start) { s_ub[e] = (u32)(key >> 32); s_lb[e] = 0u; }
                else if (key == ~0ull) { s_ub[e] = 0xFFFFFFFFu; s_lb[e] = 0xFFFFFFFFu; }       // empty slot of a dense chunk
                else {
                    const float est = __uint_as_float((u32)(key >> 32));
                    const float bn = p.bn[(u32)key];
                    float err = row_err + p.eps_half * bn + p.aq * __builtin_amdgcn_sqrtf(bn);          // >= |dot16 - dot32|
                    if (METRIC == NWKNN_METRIC_SQEUCLIDEAN) err = 2.0f * err;
                    if (METRIC == NWKNN_METRIC_COSINE) err = err * 1.0001f + 2.0e-6f * (fabsf(est) + 1.0f);
                    else err = err * 1.0001f + 2.0e-6f * (fabsf(est) + qn + bn + 1.0f);  // roundings of est and of these lines
                    float ub = est + err;
                    if (METRIC == NWKNN_METRIC_SQEUCLIDEAN) ub = fmaxf(ub, 0.0f);       // the canonical distance is clamped at 0
                    u32 sub = sortable_u32(ub), slb = sortable_u32(est - err);
                    if (METRIC == NWKNN_METRIC_COSINE && !(cosine_norm_ok(qn) && cosine_norm_ok(bn))) {
                        sub = 0xFFC00000u; slb = 0u;                                   // no estimate for degenerate rows: [-inf, NaN rank]
                    }
                    s_ub[e] = sub;
                    s_lb[e] = slb;
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        u32 lo = 0u, hi = 0xFFFFFFFFu;
        while (lo < hi) {                                   // smallest x with  #{ub <= x} >= k   (wave-uniform)
            const u32 mid = lo + ((hi - lo) >> 1);
            u32 c = 0;
            for (u32 e0 = 0; e0 < end; e0 += 64) {
                const u32 e = e0 + lane;
                c += (u32)__popcll(__ballot(e < end && s_ub[e] <= mid));
            }
            if (c >= (u32)p.k) hi = mid; else lo = mid + 1u;
        }
        const u32 U = lo;
        __builtin_amdgcn_wave_barrier();
        for (u32 e0 = start; e0 < end; e0 += 64) {
            const u32 e = e0 + lane;
            const bool live = e < end;
            const bool keep = live && s_lb[e] <= U;
            const u64 mk = __ballot(keep);
            const u32 pos = nsurv + (u32)__popcll(mk & ((1ull << lane) - 1ull));
            __builtin_amdgcn_wave_barrier();
            if (keep) s_ub[pos] = e;                        // s_ub is free after the bisection: it now holds the packed survivor slots
            else if (live) list[e] = ~0ull;
            nsurv += (u32)__popcll(mk);
        }
        __builtin_amdgcn_wave_barrier();
    } else {
        nsurv = end - start;
    }
    const bool packed = p.prune && end >= (u32)p.k;

    for (u32 i0 = 0; i0 < nsurv; i0 += 64) {
        const u32 i = i0 + lane;
        const bool live = i < nsurv;
        const u32 e = live ? (packed ? s_ub[i] : start + i) : start;
        const u64 old = live ? list[e] : ~0ull;
        const bool valid = old != ~0ull;                               // dense chunks leave ~0 in the slots of padding columns
        const u32 col = valid ? (u32)old : 0u;
        float acc = 0.0f;
        if (ROWS16) {
            const _Float16* __restrict__ brow = (const _Float16*)p.bp + (long long)col * p.dp;
            for (int k0 = 0; k0 < p.dp; k0 += 8) {
                const f16x8 bv = *(const f16x8*)(brow + k0), qv = *(const f16x8*)(qrow16 + k0);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc = __fmaf_rn((float)qv[j], (float)bv[j], acc);       // natural order: ascending k
            }
        } else {
            const float* __restrict__ brow = p.bp + (long long)col * p.dp;
            for (int k0 = 0; k0 < p.dp; k0 += 8) {
                const f32x4 be = *(const f32x4*)(brow + k0), bo = *(const f32x4*)(brow + k0 + 4);
                const f32x4 qe = *(const f32x4*)(qrow + k0), qo = *(const f32x4*)(qrow + k0 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {                          // k = k0+2j, then k0+2j+1: ascending
                    acc = __fmaf_rn(qe[j], be[j], acc);
                    acc = __fmaf_rn(qo[j], bo[j], acc);
                }
            }
        }
        const float dist = distance_epilogue<METRIC>(acc, qn, p.bn[col]);
        if (valid) list[e] = ((u64)sortable_u32(dist) << 32) | col;
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
    if (p.qtile_list) q = (long long)p.qtile_list[blockIdx.x / p.qtile_rows] * p.qtile_rows + (blockIdx.x % p.qtile_rows);
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
            if (p.cnt0) p.cnt0[q] = nvalid < k ? nvalid : k;
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
            if (p.cnt0) p.cnt0[q] = keep;
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
                                                            long long* __restrict__ out_i, int negate)
{
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= nq * k) return;
    const long long q = i / k;
    const int j = (int)(i - q * k);
    if ((u32)j < cnt[q]) {
        const u64 key = buf[q * cap + j];
        const float v = unsortable_f32((u32)(key >> 32));
        out_d[i] = negate ? -v : v;                          // MaxSim ranks -score
        out_i[i] = (long long)(u32)key + index_offset;
    } else {
        out_d[i] = negate ? -__builtin_inff() : __builtin_inff();
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

hipError_t nwk_launch_gen_f16(hipStream_t st, uint64_t seed, long long row0, long long rows, int d, void* out)
{
    const long long total = rows * (long long)d;
    if (total <= 0) return hipSuccess;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(nwk_gen_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (u64)seed, row0, rows, d, (_Float16*)out);
    return hipGetLastError();
}

hipError_t nwk_launch_prep_f16(hipStream_t st, const void* raw, int raw_is_f16, long long rows, int d, long long rows_pad, int dp, void* prepared,
                               float* norms)
{
    if (rows_pad <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((rows_pad + 255) / 256);
    if ((((uintptr_t)raw) & 15) == 0 && (raw_is_f16 ? (d % 8 == 0) : (d % 4 == 0))) {          // tiled fast path
        if (raw_is_f16)
            hipLaunchKernelGGL((nwk_prep_tiled_kernel<_Float16, true>), dim3(blocks), dim3(256), 0, st, (const _Float16*)raw, rows, d, rows_pad, dp, prepared, norms);
        else
            hipLaunchKernelGGL((nwk_prep_tiled_kernel<float, true>), dim3(blocks), dim3(256), 0, st, (const float*)raw, rows, d, rows_pad, dp, prepared, norms);
        return hipGetLastError();
    }
    if (raw_is_f16)
        hipLaunchKernelGGL((nwk_prep_f16_kernel<_Float16>), dim3(blocks), dim3(256), 0, st, (const _Float16*)raw, rows, d, rows_pad, dp, (_Float16*)prepared, norms);
    else
        hipLaunchKernelGGL((nwk_prep_f16_kernel<float>), dim3(blocks), dim3(256), 0, st, (const float*)raw, rows, d, rows_pad, dp, (_Float16*)prepared, norms);
    return hipGetLastError();
}

// in_dtype: 0 f32, 1 f16, 2 f64 (NWKNN_DTYPE_*)
hipError_t nwk_launch_prep_strided(hipStream_t st, const void* raw, int in_dtype, long long rs, long long ks, long long rows, int d,
                                   long long rows_pad, int dp, int out_f16, void* prepared, float* norms)
{
    if (rows_pad <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((rows_pad + 255) / 256);
#define NWK_PS(T, O) hipLaunchKernelGGL((nwk_prep_strided_kernel<T, O>), dim3(blocks), dim3(256), 0, st, (const T*)raw, rs, ks, rows, d, rows_pad, dp, prepared, norms)
    if (in_dtype == 0) { if (out_f16) NWK_PS(float, true); else NWK_PS(float, false); }
    else if (in_dtype == 1) { if (out_f16) NWK_PS(_Float16, true); else return hipErrorInvalidValue; }
    else if (in_dtype == 2) { if (out_f16) NWK_PS(double, true); else NWK_PS(double, false); }
    else return hipErrorInvalidValue;
#undef NWK_PS
    return hipGetLastError();
}

hipError_t nwk_launch_prep(hipStream_t st, const float* raw, long long rows, int d, long long rows_pad, int dp, float* prepared,
                           float* norms)
{
    if (rows_pad <= 0) return hipSuccess;
    const long long blocks = (rows_pad + 255) / 256;
    if ((((uintptr_t)raw) & 15) == 0 && d % 4 == 0) {                                           // tiled fast path
        hipLaunchKernelGGL((nwk_prep_tiled_kernel<float, false>), dim3((unsigned)blocks), dim3(256), 0, st, raw, rows, d, rows_pad, dp, (void*)prepared, norms);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(nwk_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, st, raw, rows, d, rows_pad, dp, prepared, norms);
    return hipGetLastError();
}

template <int METRIC, int TM, int BKT>
static hipError_t launch_tile_shape(hipStream_t st, const NwkTileParams& p, bool dense, unsigned grid, int diag)
{
    if (dense) { hipLaunchKernelGGL((nwk_tile_kernel<METRIC, true, TM, BKT, 0>), dim3(grid), dim3(256), 0, st, p); return hipGetLastError(); }
    switch (diag) {              // dev-only timing ablations (wrong results by construction)
    case 1: hipLaunchKernelGGL((nwk_tile_kernel<METRIC, false, TM, BKT, 1>), dim3(grid), dim3(256), 0, st, p); break;
    case 2: hipLaunchKernelGGL((nwk_tile_kernel<METRIC, false, TM, BKT, 2>), dim3(grid), dim3(256), 0, st, p); break;
    case 4: hipLaunchKernelGGL((nwk_tile_kernel<METRIC, false, TM, BKT, 4>), dim3(grid), dim3(256), 0, st, p); break;
    case 8: hipLaunchKernelGGL((nwk_tile_kernel<METRIC, false, TM, BKT, 8>), dim3(grid), dim3(256), 0, st, p); break;
    default: hipLaunchKernelGGL((nwk_tile_kernel<METRIC, false, TM, BKT, 0>), dim3(grid), dim3(256), 0, st, p); break;
    }
    return hipGetLastError();
}

template <int METRIC>
static hipError_t launch_tile_metric(hipStream_t st, const NwkTileParams& p, bool dense, unsigned grid, int tile_rows, int diag)
{
    if (tile_rows == 256) return launch_tile_shape<METRIC, 256, 16>(st, p, dense, grid, diag);
    return launch_tile_shape<METRIC, 128, 32>(st, p, dense, grid, diag);
}

// p.n_qtiles counts query tiles of `tile_rows` (128 or 256) rows; p.n_btiles counts 128-row base tiles.
hipError_t nwk_launch_tile(hipStream_t st, NwkTileParams p, int metric, bool dense, int tile_rows, int diag)
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
    case NWKNN_METRIC_SQEUCLIDEAN: return launch_tile_metric<NWKNN_METRIC_SQEUCLIDEAN>(st, p, dense, (unsigned)grid, tile_rows, diag);
    case NWKNN_METRIC_ONE_MINUS_DOT: return launch_tile_metric<NWKNN_METRIC_ONE_MINUS_DOT>(st, p, dense, (unsigned)grid, tile_rows, diag);
    case NWKNN_METRIC_COSINE: return launch_tile_metric<NWKNN_METRIC_COSINE>(st, p, dense, (unsigned)grid, tile_rows, diag);
    default: return hipErrorInvalidValue;
    }
}

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
    if (p.f16_shape == 1 && diag == 0 && !dense && p.tiles_per_wg > 1) p.n_btiles = (p.n_btiles + p.tiles_per_wg - 1) / p.tiles_per_wg;
    else p.tiles_per_wg = 1;
}

hipError_t nwk_launch_tile_f16(hipStream_t st, NwkTileParams p, int metric, bool dense, int diag)
{
    if (p.n_qtiles <= 0 || p.n_btiles <= 0) return hipSuccess;
    if (p.sq_shift < 0 || p.sq_shift > 6) return hipErrorInvalidValue;
    setup_strips(p, dense, diag);
    const int SQ = 1 << p.sq_shift, SB = 64 >> p.sq_shift;
    const long long n_super_q = (p.n_qtiles + SQ - 1) / SQ;
    p.n_super_b = (p.n_btiles + SB - 1) / SB;
    long long supers = n_super_q * p.n_super_b;
    supers = (supers + 7) / 8 * 8;
    const long long grid = supers * 64;
    if (grid > 0x7FFFFFFFll) return hipErrorInvalidValue;
    switch (metric) {
    case NWKNN_METRIC_SQEUCLIDEAN: return launch_tile_f16_metric<NWKNN_METRIC_SQEUCLIDEAN>(st, p, dense, (unsigned)grid, diag);
    case NWKNN_METRIC_ONE_MINUS_DOT: return launch_tile_f16_metric<NWKNN_METRIC_ONE_MINUS_DOT>(st, p, dense, (unsigned)grid, diag);
    case NWKNN_METRIC_COSINE: return launch_tile_f16_metric<NWKNN_METRIC_COSINE>(st, p, dense, (unsigned)grid, diag);
    default: return hipErrorInvalidValue;
    }
}

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

hipError_t nwk_launch_rescore(hipStream_t st, const NwkRescoreParams& p, int metric, long long nq_pad)
{
    if (nq_pad <= 0) return hipSuccess;
    const int waves = p.cap <= 1024 ? 4 : (p.cap <= 2048 ? 2 : 1);         // nq_pad is a multiple of 256
    const unsigned grid = (unsigned)(nq_pad / waves);
    const size_t lds = (size_t)waves * p.cap * 8 + 512;        // + slack: predicated-off lanes may still address past `end`
#define NWK_RS(M)                                                                                                        \
    do {                                                                                                                  \
        if (p.rows16) hipLaunchKernelGGL((nwk_rescore_kernel<M, true>), dim3(grid), dim3(64 * waves), lds, st, p);        \
        else hipLaunchKernelGGL((nwk_rescore_kernel<M, false>), dim3(grid), dim3(64 * waves), lds, st, p);                \
    } while (0)
    if (metric == NWKNN_METRIC_SQEUCLIDEAN) NWK_RS(NWKNN_METRIC_SQEUCLIDEAN);
    else if (metric == NWKNN_METRIC_ONE_MINUS_DOT) NWK_RS(NWKNN_METRIC_ONE_MINUS_DOT);
    else if (metric == NWKNN_METRIC_COSINE) NWK_RS(NWKNN_METRIC_COSINE);
    else return hipErrorInvalidValue;
#undef NWK_RS
    return hipGetLastError();
}

hipError_t nwk_launch_maxabs(hipStream_t st, const float* x, long long n, unsigned int* out_bits)
{
    if (n <= 0) return hipSuccess;                              // n is a multiple of 4 (dp is a multiple of 32)
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(nwk_maxabs_kernel, dim3((unsigned)blocks), dim3(256), 0, st, x, n / 4, out_bits);
    return hipGetLastError();
}

hipError_t nwk_launch_half_copy16(hipStream_t st, const void* rows16, long long rows_pad, int dp, float scale, const float* norms, void* out16)
{
    const long long total = rows_pad * (dp / 8);
    if (total <= 0) return hipSuccess;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(nwk_half_copy16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const _Float16*)rows16, rows_pad, dp, scale, norms, (_Float16*)out16);
    return hipGetLastError();
}

hipError_t nwk_launch_maxabs_f16(hipStream_t st, const void* x, long long n, unsigned int* out_bits)
{
    if (n <= 0) return hipSuccess;                              // n is a multiple of 64 halves
    long long blocks = (n / 8 + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(nwk_maxabs_f16_kernel, dim3((unsigned)blocks), dim3(256), 0, st, (const _Float16*)x, n / 8, out_bits);
    return hipGetLastError();
}

hipError_t nwk_launch_half_copy(hipStream_t st, const float* prepared, long long rows_pad, int dp32, int dp16, float scale, const float* norms,
                                void* out16)
{
    const long long total = rows_pad * (dp16 / 8);
    if (total <= 0) return hipSuccess;
    long long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    hipLaunchKernelGGL(nwk_half_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, st, prepared, rows_pad, dp32, dp16, scale, norms, (_Float16*)out16);
    return hipGetLastError();
}

// MaxSim pre-screen: p.qp / p.bp are the scaled fp16 copies (dp halves), p.n_qtiles groups of 4 queries, p.n_btiles tiles of 4 docs
hipError_t nwk_launch_maxsim_pre(hipStream_t st, NwkTileParams p)
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
    hipLaunchKernelGGL(nwk_maxsim_pre_kernel, dim3((unsigned)grid), dim3(256), 0, st, p);
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
    // (2) rescore: (query, slice) waves; the query's 32 token rows in LDS when they fit (d = 128: 16 KiB per wave)
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

hipError_t nwk_launch_compact(hipStream_t st, const NwkCompactParams& p, long long n_queries)
{
    if (n_queries <= 0) return hipSuccess;
    const size_t lds = (size_t)p.cap * 16 + 16;
    hipLaunchKernelGGL(nwk_compact_kernel, dim3((unsigned)n_queries), dim3(256), lds, st, p);
    return hipGetLastError();
}

hipError_t nwk_launch_finalize(hipStream_t st, const u64* buf, const u32* cnt, int cap, int k, long long nq, long long index_offset,
                               float* out_d, long long* out_i, int negate)
{
    const long long total = nq * k;
    if (total <= 0) return hipSuccess;
    hipLaunchKernelGGL(nwk_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, buf, cnt, cap, k, nq, index_offset,
                       out_d, out_i, negate);
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
