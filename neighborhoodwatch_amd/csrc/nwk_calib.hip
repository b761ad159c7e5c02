// nwk_calib.hip -- device calibration: what a BARE MFMA loop delivers on THIS chip, on random operands  (gfx950 only)
//
// The nominal peaks (157.3 TF f32, 2.5 PF f16: MI355X_MICROARCH.md) assume 2.4 GHz.  Under dense MFMA work on random data the chip
// holds its clock well below that (power limit: the guide's "DVFS give-back"), by an amount that differs from device to device
// (+-6 %) and from MFMA shape to MFMA shape.  A tile kernel can therefore not reach the nominal peak whatever it does; what it can be
// held against is the rate of the same MFMA instruction issued back to back from registers -- no LDS, no LDS-DMA, no global memory, no
// epilogue -- at the same register tile per wave and the same number of waves per SIMD as the kernel it stands beside:
//   f16: 8 waves per workgroup (two per SIMD), a wave owns 128 x 64 outputs = 8 x 4 accumulators of v_mfma_f32_16x16x32_f16 and walks
//        a 64-half slab = 2 k-steps = 64 MFMAs per iteration over 16 A + 8 B fragments held in registers (nwk_tile_kernel_f16x's COMPUTE
//        phase, and nothing else);
//   f32: 4 waves per workgroup, three workgroups per CU, a wave owns 128 x 64 = 4 x 2 accumulators of v_mfma_f32_32x32x2_f32, 64 MFMAs
//        per iteration over 8 k-pairs of fragments (nwk_tile_kernel<.,false,256,16>'s slab).
// Operands are seeded random values (the generator of nwknn_generate), distinct per fragment, so the multiplier arrays toggle as they do
// on real data; on all-zero operands the same loop runs ~20 % faster (the guide's item 1) -- that is NOT the ceiling of a real search.
// bench.py prints achieved / this as `roofline.frac_of_bare_loop` next to achieved / nominal.
#include "nwk_device.h"

struct NwkCalibOut {
    unsigned long long cycles, realtime;      // s_memtime / s_memrealtime deltas around the loop of one wave (clock = cycles / realtime x 100 MHz)
    float sink;
};

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void nwk_mfma_bare_f16_kernel(const _Float16* __restrict__ rnd, int iters, NwkCalibOut* out)
{
    extern __shared__ char occupancy_ballast[];                      // dynamic LDS nobody touches: the launcher sizes it to cap the workgroups per CU
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    f16x8 fa[2][8], fb[2][4];
    const f16x8* src = (const f16x8*)rnd + ((size_t)(blockIdx.x * WAVES + w) % 64) * 64 * 24 + l;     // 24 fragments per lane, 64 distinct waves' worth
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
        for (int m = 0; m < 8; ++m) fa[ks][m] = src[(ks * 12 + m) * 64];
#pragma unroll
        for (int n = 0; n < 4; ++n) fb[ks][n] = src[(ks * 12 + 8 + n) * 64];
    }
    f32x4 acc[8][4];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = zero;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int n = NWK_MFMA_SNAKE && (m & 1) ? 3 - j : j;      // the tile kernels' issue order: the ceiling is the order's too (+1.8 % over n ascending)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[ks][m], fb[ks][n], acc[m][n], 0, 0, 0);
                }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    if (threadIdx.x == 0) { out[blockIdx.x].cycles = t1 - t0; out[blockIdx.x].realtime = r1 - r0; }
    if (s == 12345.678f) out[blockIdx.x].sink = s;                  // keeps the accumulators alive; never true in practice
}

__global__ __launch_bounds__(256, 3) void nwk_mfma_bare_f32_kernel(const float* __restrict__ rnd, int iters, NwkCalibOut* out)
{
    extern __shared__ char occupancy_ballast[];
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    float fa[4][4], fb[4][2];                                        // [k-pair][row block] / [k-pair][column block]: one 8-k group of a slab (24 registers, as in the kernel)
    const float* src = rnd + ((size_t)(blockIdx.x * 4 + w) % 64) * 64 * 24 + l;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int m = 0; m < 4; ++m) fa[t][m] = src[(t * 6 + m) * 64];
#pragma unroll
        for (int n = 0; n < 2; ++n) fb[t][n] = src[(t * 6 + 4 + n) * 64];
    }
    f32x16 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < 2; ++g)                                  // a 16-float slab = two 8-k groups = 64 MFMAs
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int m = 0; m < 4; ++m)
#pragma unroll
                    for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[t][m], fb[t][n], acc[m][n], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[m][n][r];
    if (threadIdx.x == 0) { out[blockIdx.x].cycles = t1 - t0; out[blockIdx.x].realtime = r1 - r0; }
    if (s == 12345.678f) out[blockIdx.x].sink = s;
}

// fabric / HBM probe: every lane streams 16-B loads over the working set, pass after pass; nothing is written but one word per workgroup
__global__ __launch_bounds__(256) void nwk_stream_read_kernel(const f32x4* __restrict__ src, long long n16, int passes, float* __restrict__ sink)
{
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    const long long stride = (long long)gridDim.x * 256;
    for (int p = 0; p < passes; ++p)
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n16; i += 4 * stride) {
            // four independent loads in flight per lane
            const f32x4 a = src[i];
            const f32x4 b = i + stride < n16 ? src[i + stride] : acc;
            const f32x4 c = i + 2 * stride < n16 ? src[i + 2 * stride] : acc;
            const f32x4 d = i + 3 * stride < n16 ? src[i + 3 * stride] : acc;
            acc += (a + b) + (c + d);
        }
    const float s = acc[0] + acc[1] + acc[2] + acc[3];
    if (s == 12345.678f) sink[blockIdx.x] = s;                       // never true on the zero-filled buffer: keeps the loads alive
}

extern "C++" {

hipError_t nwk_launch_stream_read(hipStream_t st, const void* src, long long bytes, int passes, unsigned grid, float* sink)
{
    hipLaunchKernelGGL(nwk_stream_read_kernel, dim3(grid), dim3(256), 0, st, (const f32x4*)src, bytes / 16, passes, sink);
    return hipGetLastError();
}

// One launch: `grid` workgroups x `iters` loop iterations.  FLOPs per launch = grid x waves x iters x 64 MFMAs x 2 x M x N x K.
// waves_per_simd: how many of these MFMA-only waves share a SIMD (f16: 1 = four-wave workgroups, one per CU; 2 = the eight-wave workgroup of
// the tile kernel; f32: 1..3 four-wave workgroups per CU), enforced with dynamic LDS nobody touches.
hipError_t nwk_launch_mfma_bare(hipStream_t st, int f16, int waves_per_simd, const void* rnd, int iters, unsigned grid, void* out)
{
    const size_t lds_total = 160 * 1024;
    if (f16) {
        if (waves_per_simd >= 2) hipLaunchKernelGGL((nwk_mfma_bare_f16_kernel<8>), dim3(grid), dim3(512), lds_total / 2 + 1024, st, (const _Float16*)rnd, iters, (NwkCalibOut*)out);
        else hipLaunchKernelGGL((nwk_mfma_bare_f16_kernel<4>), dim3(grid), dim3(256), lds_total / 2 + 1024, st, (const _Float16*)rnd, iters, (NwkCalibOut*)out);
    } else {
        const int occ = waves_per_simd < 1 ? 1 : (waves_per_simd > 3 ? 3 : waves_per_simd);
        const size_t lds = occ == 3 ? lds_total / 3 - 2048 : (occ == 2 ? lds_total / 2 - 2048 : lds_total / 2 + 1024);
        hipLaunchKernelGGL(nwk_mfma_bare_f32_kernel, dim3(grid), dim3(256), lds, st, (const float*)rnd, iters, (NwkCalibOut*)out);
    }
    return hipGetLastError();
}

}  // extern "C++"
