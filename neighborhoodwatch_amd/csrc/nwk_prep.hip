// nwk_prep.hip -- seeded generator, layout + norm preparation, max|x| and the scaled fp16 copies of the pre-screen  (gfx950 only; shared pieces and the overall design: nwk_device.h)
#include "nwk_device.h"

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

// max |x| of everything a prep kernel lays out, folded into the layout pass (round 5; the pre-screen's scale needs it, and the separate
// reduction re-read the whole prepared base): per lane the largest |x| as float bits (non-negative floats order like their bit patterns;
// NaN / Inf show up as >= 0x7F800000, as in nwk_maxabs_kernel), one atomicMax per wave.
__device__ __forceinline__ void prep_fold_maxabs(u32 mbits, u32* __restrict__ out_bits)
{
    if (!out_bits) return;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const u32 other = (u32)__shfl_xor((int)mbits, o); mbits = other > mbits ? other : mbits; }
    if ((threadIdx.x & 63) == 0 && mbits) atomicMax(out_bits, mbits);
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
                                                        int dp, float* __restrict__ prepared, float* __restrict__ norms, u32* __restrict__ maxabs_bits)
{
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    u32 mbits = 0u;
    if (r < rows_pad && r >= rows) {
        float* dst = prepared + r * dp;
        for (int k = 0; k < dp; k += 4) *(f32x4*)(dst + k) = (f32x4){0.f, 0.f, 0.f, 0.f};
        norms[r] = 0.0f;
    } else if (r < rows) {
    float* dst = prepared + r * dp;
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
        for (int j = 0; j < 8; ++j) { acc = __fmaf_rn(x[j], x[j], acc); const u32 b = __float_as_uint(x[j]) & 0x7FFFFFFFu; mbits = b > mbits ? b : mbits; }
        *(f32x4*)(dst + k0) = (f32x4){x[0], x[2], x[4], x[6]};
        *(f32x4*)(dst + k0 + 4) = (f32x4){x[1], x[3], x[5], x[7]};
    }
    norms[r] = acc;
    }
    prep_fold_maxabs(mbits, maxabs_bits);
}

// ------------------------------------------------------------------------------------------------
// prep, tiled form (the default for row-major input): a lane-per-row kernel touches 64 different rows per load
// instruction, 16 B from each, i.e. four requests per 64-B granule (2 TB/s on fp32, 0.9 TB/s on fp16).  Here a block owns
// 256 rows and walks K in 128-byte steps: every global load and store instruction moves whole 128-B lines (eight lanes
// per row; round 1: 64-B segments), the row <-> lane transposition goes through LDS (144-B row pitch: conflict-free for ds_read/write_b128), and the
// lane that owns a row still runs the norm chain in ascending k, so the results are bit-identical to nwk_prep_kernel.
// TIN = float | _Float16; OUT16 selects the fp16 layout.  Requires a 16-B aligned source and d*sizeof(TIN) % 16 == 0.
// ------------------------------------------------------------------------------------------------
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <typename TIN, bool OUT16>
__global__ __launch_bounds__(256) void nwk_prep_tiled_kernel(const TIN* __restrict__ raw, long long rows, int d, long long rows_pad, int dp,
                                                              void* __restrict__ prepared, float* __restrict__ norms, u32* __restrict__ maxabs_bits)
{
    // round 2: a step moves 128 B of input per row (was 64): every global access is now a whole 128-B line (eight lanes per row), and
    // the next step's lines are requested one step ahead (registers) so that they fly under the LDS passes and the stores of this one
    constexpr int SEGS = 8;                               // 16-B pieces of input per row per step
    constexpr int EIN = 16 / (int)sizeof(TIN);            // elements per 16-B piece of input
    constexpr int KS = SEGS * EIN;                        // elements per step: 32 (float) / 64 (fp16)
    constexpr int OES = OUT16 ? 2 : 4;                    // output element size
    constexpr int OB = KS * OES;                          // output bytes per row per step: 128 (f32->f32, f16->f16) or 64 (f32->f16)
    constexpr int OP = OB / 16;                           // 16-B output pieces per row per step
    constexpr int PITCH = 144;                            // 128 + 16: lane-per-row ds_read_b128 / ds_write_b128 stay conflict-free
    constexpr int RPP = 256 / SEGS;                       // rows per load pass
    // round 4: ONE 36 KiB tile, rewritten in place by the lane that owns the row (it reads all of its row's pieces into registers before it
    // writes the permuted / converted ones back), so FOUR workgroups fit a CU instead of two: twice the loads in flight for a kernel whose
    // only business is HBM (PREP_ALIAS=0 builds the two-tile form for A/B: tools/mk_variant_lib.sh)
#ifndef PREP_ALIAS
#define PREP_ALIAS 1
#endif
#ifndef PREP_NT
#define PREP_NT 0                                         // build-time A/B: 1 = non-temporal stores of the prepared rows, 2 = non-temporal loads of the raw rows too
#endif
    __shared__ __attribute__((aligned(16))) char tile_in[256 * PITCH];
#if PREP_ALIAS
    char* const tile_out = tile_in;
#else
    __shared__ __attribute__((aligned(16))) char tile_out[256 * PITCH];
#endif

    const int tid = threadIdx.x;
    const long long r0 = (long long)blockIdx.x * 256;
    const size_t in_row_bytes = (size_t)d * sizeof(TIN);
    const size_t out_row_bytes = (size_t)dp * OES;
    float acc = 0.0f;
    u32 mbits = 0u;

    u32x4 v[SEGS];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int pass = 0; pass < SEGS; ++pass) {
            const int row = pass * RPP + (tid / SEGS), seg = tid % SEGS;
            v[pass] = (u32x4){0u, 0u, 0u, 0u};
            if (k0 < dp && r0 + row < rows && k0 + seg * EIN < d) {
                const u32x4* src = (const u32x4*)((const char*)raw + (size_t)(r0 + row) * in_row_bytes + (size_t)k0 * sizeof(TIN) + seg * 16);
                v[pass] = PREP_NT >= 2 ? __builtin_nontemporal_load(src) : *src;
            }
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < dp; k0 += KS) {
#pragma unroll
        for (int pass = 0; pass < SEGS; ++pass) {
            const int row = pass * RPP + (tid / SEGS), seg = tid % SEGS;
            *(u32x4*)(tile_in + row * PITCH + seg * 16) = v[pass];
        }
        __syncthreads();
        fetch(k0 + KS);                                   // next step's lines: requested now, consumed after this step's stores
        // the lane that owns row `tid`: ascending-k norm chain + output layout for this step
        {
            float x[KS];
#pragma unroll
            for (int piece = 0; piece < SEGS; ++piece) {
                const u32x4 pv = *(const u32x4*)(tile_in + tid * PITCH + piece * 16);
                if (sizeof(TIN) == 4) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) x[piece * EIN + j] = __uint_as_float(pv[j]);
                } else {
                    const f16x8 hv = __builtin_bit_cast(f16x8, pv);
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
                for (int j = 0; j < KS; ++j) { acc = __fmaf_rn(x[j], x[j], acc); const u32 b = __float_as_uint(x[j]) & 0x7FFFFFFFu; mbits = b > mbits ? b : mbits; }
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
            if (r0 + row < rows_pad) {
                u32x4* dst = (u32x4*)((char*)prepared + (size_t)(r0 + row) * out_row_bytes + (size_t)k0 * OES + seg * 16);
                const u32x4 val = *(const u32x4*)(tile_out + row * PITCH + seg * 16);
                if (PREP_NT >= 1) __builtin_nontemporal_store(val, dst); else *dst = val;
            }
        }
#if PREP_ALIAS
        __syncthreads();                                  // the tile has been read out: the next step may overwrite it
#endif
    }
    if (r0 + tid < rows_pad) norms[r0 + tid] = acc;
    if (!OUT16) prep_fold_maxabs(mbits, maxabs_bits);
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
                                                                long long rows_pad, int dp, void* __restrict__ prepared, float* __restrict__ norms,
                                                                u32* __restrict__ maxabs_bits)
{
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    u32 mbits = 0u;
    if (r < rows_pad) {
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
            for (int j = 0; j < 8; ++j) { acc = __fmaf_rn(x[j], x[j], acc); const u32 b = __float_as_uint(x[j]) & 0x7FFFFFFFu; mbits = b > mbits ? b : mbits; }
            float* dst = (float*)prepared + r * dp + k0;
            *(f32x4*)(dst) = (f32x4){x[0], x[2], x[4], x[6]};
            *(f32x4*)(dst + 4) = (f32x4){x[1], x[3], x[5], x[7]};
        }
    }
    norms[r] = acc;
    }
    if (!OUT16) prep_fold_maxabs(mbits, maxabs_bits);
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
    // ONE atomic per workgroup: the first version had every wave of up to 4 096 workgroups hit the same word -- 16k device-scope atomics on
    // one address, 0.19 ms for a 30 MB query matrix whose bytes take 10 us (profiles/r05m_c2_prescreen_chunk_trace.log)
    __shared__ u32 s_m[4];
    if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        const u32 a = s_m[0] > s_m[1] ? s_m[0] : s_m[1], b = s_m[2] > s_m[3] ? s_m[2] : s_m[3];
        const u32 mm = a > b ? a : b;
        if (mm) atomicMax(out_bits, mm);
    }
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
            hipLaunchKernelGGL((nwk_prep_tiled_kernel<_Float16, true>), dim3(blocks), dim3(256), 0, st, (const _Float16*)raw, rows, d, rows_pad, dp, prepared, norms, (u32*)nullptr);
        else
            hipLaunchKernelGGL((nwk_prep_tiled_kernel<float, true>), dim3(blocks), dim3(256), 0, st, (const float*)raw, rows, d, rows_pad, dp, prepared, norms, (u32*)nullptr);
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
                                   long long rows_pad, int dp, int out_f16, void* prepared, float* norms, unsigned int* maxabs_bits)
{
    if (rows_pad <= 0) return hipSuccess;
    const unsigned blocks = (unsigned)((rows_pad + 255) / 256);
#define NWK_PS(T, O) hipLaunchKernelGGL((nwk_prep_strided_kernel<T, O>), dim3(blocks), dim3(256), 0, st, (const T*)raw, rs, ks, rows, d, rows_pad, dp, prepared, norms, maxabs_bits)
    if (in_dtype == 0) { if (out_f16) NWK_PS(float, true); else NWK_PS(float, false); }
    else if (in_dtype == 1) { if (out_f16) NWK_PS(_Float16, true); else return hipErrorInvalidValue; }
    else if (in_dtype == 2) { if (out_f16) NWK_PS(double, true); else NWK_PS(double, false); }
    else return hipErrorInvalidValue;
#undef NWK_PS
    return hipGetLastError();
}

hipError_t nwk_launch_prep(hipStream_t st, const float* raw, long long rows, int d, long long rows_pad, int dp, float* prepared,
                           float* norms, unsigned int* maxabs_bits)
{
    if (rows_pad <= 0) return hipSuccess;
    const long long blocks = (rows_pad + 255) / 256;
    if ((((uintptr_t)raw) & 15) == 0 && d % 4 == 0) {                                           // tiled fast path
        hipLaunchKernelGGL((nwk_prep_tiled_kernel<float, false>), dim3((unsigned)blocks), dim3(256), 0, st, raw, rows, d, rows_pad, dp, (void*)prepared, norms, maxabs_bits);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(nwk_prep_kernel, dim3((unsigned)blocks), dim3(256), 0, st, raw, rows, d, rows_pad, dp, prepared, norms, maxabs_bits);
    return hipGetLastError();
}

hipError_t nwk_launch_maxabs(hipStream_t st, const float* x, long long n, unsigned int* out_bits)
{
    if (n <= 0) return hipSuccess;                              // n is a multiple of 4 (dp is a multiple of 32)
    long long blocks = (n / 4 + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
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

}  // extern "C++"
