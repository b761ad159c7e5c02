// tools/lds_mfma_probe.hip -- is a bigger per-wave register tile the lever that is left for the fp16 tile kernel?
//
// The ping-pong kernel's waves hold 128 x 64 accumulators (128 VGPRs of a 256-register wave, two waves per SIMD) and read 24 fragments
// (16 A + 8 B ds_read_b128) per 64 MFMAs: 192 KiB of fragment reads per 64-half slab and CU.  DESIGN.md section 8 names the alternative:
// ONE wave per SIMD with 512 registers and a 128 x 128 accumulator tile (in AGPRs), 32 fragments per 128 MFMAs = -33 % fragment bytes per
// FLOP, at the price of the ping-pong overlap.  This probe prices the two register tiles in isolation: the kernel's LDS image (128-B rows,
// 16-B chunk c of row r in slot c ^ ((r >> 1) & 7)) filled ONCE with random halves, then N slabs of [fragment reads + MFMAs] over the
// same stage -- no LDS-DMA, no global memory, no epilogue, no barriers -- for
//   mode 0: 8 waves (2 per SIMD), wave tile 128 x 64 (8 x 4 accumulators of v_mfma_f32_16x16x32_f16): the shipped shape;
//   mode 1: 4 waves (1 per SIMD), wave tile 128 x 128 (8 x 8 accumulators): the alternative;
//   mode 2: 8 waves, 128 x 64, fragments read once and kept (MFMAs only): the bare loop, for scale.
// It prints TFLOP/s and the in-kernel clock per mode.  What it cannot show is how either tile overlaps with an LDS-DMA stream.
//   hipcc -O3 --offload-arch=gfx950 tools/lds_mfma_probe.hip -o ab/lds_mfma_probe && ab/lds_mfma_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct Out { unsigned long long cycles, realtime; float sink; };

#define STAGE 65536

template <int MODE>
__global__ __launch_bounds__(MODE == 1 ? 256 : 512, MODE == 1 ? 1 : 2) void probe(const _Float16* __restrict__ rnd, int slabs, Out* out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];      // one 64 KiB stage (A rows 0-255 | B rows 0-255 of a 64-half slab) + ballast
    constexpr int NT = MODE == 1 ? 256 : 512;
    constexpr int NB = MODE == 1 ? 8 : 4;                            // B column blocks per wave
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, c16 = l & 15;
    for (int i = tid; i < STAGE / 16; i += NT) ((f16x8*)smem)[i] = ((const f16x8*)rnd)[(blockIdx.x % 8) * (STAGE / 16) + i];
    __syncthreads();
    const int wr = MODE == 1 ? (w >> 1) : (w >> 2), wc = MODE == 1 ? (w & 1) : (w & 3);
    const int swz = (c16 >> 1) & 7;
    const int offA0 = (wr * 128 + c16) * 128 + ((g ^ swz) << 4);
    const int offB0 = 32768 + (wc * (MODE == 1 ? 128 : 64) + c16) * 128 + ((g ^ swz) << 4);
    f32x4 acc[8][NB];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[m][n] = zero;
    f16x8 fa[8], fb[NB];
    if (MODE == 2) {
#pragma unroll
        for (int m = 0; m < 8; ++m) fa[m] = *(const f16x8*)(smem + offA0 + m * 2048);
#pragma unroll
        for (int n = 0; n < NB; ++n) fb[n] = *(const f16x8*)(smem + offB0 + n * 2048);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < slabs; ++s) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (MODE != 2) {
#pragma unroll
                for (int m = 0; m < 8; ++m) fa[m] = *(const f16x8*)(smem + (offA0 ^ (ks << 6)) + m * 2048);
#pragma unroll
                for (int n = 0; n < NB; ++n) fb[n] = *(const f16x8*)(smem + (offB0 ^ (ks << 6)) + n * 2048);
            }
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[m], fb[n], acc[m][n], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < NB; ++n) sum += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    if (tid == 0) { out[blockIdx.x].cycles = t1 - t0; out[blockIdx.x].realtime = r1 - r0; }
    if (sum == 12345.678f) out[blockIdx.x].sink = sum;
}

// modes 3 / 4: the ping-pong phase structure (two groups of four waves alternate, one barrier per phase) with the fragment reads moved INTO the
// compute phase, software-pipelined one k-step ahead (two register buffers of 8 A + 4 B fragments), so that the other phase holds nothing but the
// LDS-DMA issue: mode 3 leaves it empty, mode 4 issues the wave's share of the next slab (8 x 1 KiB from an L2-resident buffer into the other
// stage) and waits for its own previous share first.  Data are random either way: what is measured is the rate of the structure.
template <int DMA>
__global__ __launch_bounds__(512, 2) void probe_pp(const _Float16* __restrict__ rnd, const _Float16* __restrict__ stream, int slabs, Out* out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];      // two 64 KiB stages
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, c16 = l & 15;
    for (int i = tid; i < 2 * STAGE / 16; i += 512) ((f16x8*)smem)[i] = ((const f16x8*)rnd)[(blockIdx.x % 4) * (2 * STAGE / 16) + i];
    __syncthreads();
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int grp = wu >> 2, wr = wu >> 2, wc = wu & 3;
    const int swz = (c16 >> 1) & 7;
    const int offA0 = (wr * 128 + c16) * 128 + ((g ^ swz) << 4);
    const int offB0 = 32768 + (wc * 64 + c16) * 128 + ((g ^ swz) << 4);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)stream, 0, 0x7FFFFFFF, 0x00020000);
    f32x4 acc[8][4];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = zero;
    f16x8 fa[2][8], fb[2][4];
    auto read_k = [&](int buf, int stage, int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 8; ++m) fa[buf][m] = *(const f16x8*)(smem + stage * STAGE + (offA0 ^ (ks << 6)) + m * 2048);
#pragma unroll
        for (int n = 0; n < 4; ++n) fb[buf][n] = *(const f16x8*)(smem + stage * STAGE + (offB0 ^ (ks << 6)) + n * 2048);
    };
    auto mfma_k = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[buf][m], fb[buf][n], acc[m][n], 0, 0, 0);
    };
    read_k(0, 0, 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (grp == 1) __builtin_amdgcn_s_barrier();                      // group 1 starts one phase later
    for (int s = 0; s < slabs; ++s) {
        const int st = s & 1;
        // ---- COMPUTE: reads of the next k-step in flight under this k-step's MFMAs
        __builtin_amdgcn_s_setprio(2);
        read_k(1, st, 1);
        mfma_k(0);
        read_k(0, st ^ 1, 0);
        mfma_k(1);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        // ---- the other phase: nothing (mode 3) or this wave's share of the LDS-DMA stream (mode 4)
        if (DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // its share of the slab the partner is about to read: requested a slab period ago
            const int soff = ((s * 8 + wu) & 127) * 65536;           // an 8 MiB window of the stream buffer: L2 / Infinity-Cache resident
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + (st ^ 1) * STAGE + (wu * 8 + i) * 1024), 16,
                                                         (unsigned)(l * 16), soff + i * 1024, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
    }
    if (grp == 0) __builtin_amdgcn_s_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) sum += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    if (tid == 0) { out[blockIdx.x].cycles = t1 - t0; out[blockIdx.x].realtime = r1 - r0; }
    if (sum == 12345.678f) out[blockIdx.x].sink = sum;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int MODE>
static int run(const _Float16* rnd, Out* out, int cus, int slabs, const char* what)
{
    const unsigned grid = (unsigned)cus * 4;
    const size_t lds = 84 * 1024;                                    // > half of 160 KiB: one workgroup per CU in every mode
    const int nt = MODE == 1 ? 256 : 512, waves = nt / 64, nb = MODE == 1 ? 8 : 4;
    const double flops = (double)grid * waves * slabs * 2.0 * 8 * nb * (2.0 * 16 * 16 * 32);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(nt), lds, 0, rnd, slabs, out);     // settle the clock
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 40;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe<MODE>, dim3(grid), dim3(nt), lds, 0, rnd, slabs, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Out> h(grid);
    CK(hipMemcpy(h.data(), out, grid * sizeof(Out), hipMemcpyDeviceToHost));
    std::vector<double> clk;
    for (auto& o : h) if (o.realtime) clk.push_back((double)o.cycles / (double)o.realtime * 0.1);
    std::sort(clk.begin(), clk.end());
    const double frag_kib = MODE == 2 ? 0.0 : (double)waves * 2 * (8 + nb) * 1.0;     // KiB of fragment reads per slab and CU
    printf("%-58s %8.1f TFLOP/s  clock %.3f GHz  fragment reads %5.0f KiB per slab and CU (%.0f B per MFMA)\n", what, flops * reps / (ms * 1e-3) / 1e12,
           clk.empty() ? 0.0 : clk[clk.size() / 2], frag_kib, MODE == 2 ? 0.0 : frag_kib * 1024.0 / (waves * 2.0 * 8 * nb));
    fflush(stdout);
    return 0;
}

template <int DMA>
static int run_pp(const _Float16* rnd, const _Float16* stream, Out* out, int cus, int slabs, const char* what)
{
    const unsigned grid = (unsigned)cus * 4;
    const size_t lds = 2 * STAGE + 1024;
    const double flops = (double)grid * 8 * slabs * 64.0 * (2.0 * 16 * 16 * 32);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 30; ++i) hipLaunchKernelGGL(probe_pp<DMA>, dim3(grid), dim3(512), lds, 0, rnd, stream, slabs, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 40;
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(probe_pp<DMA>, dim3(grid), dim3(512), lds, 0, rnd, stream, slabs, out);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Out> h(grid);
    CK(hipMemcpy(h.data(), out, grid * sizeof(Out), hipMemcpyDeviceToHost));
    std::vector<double> clk;
    for (auto& o : h) if (o.realtime) clk.push_back((double)o.cycles / (double)o.realtime * 0.1);
    std::sort(clk.begin(), clk.end());
    const double tf = flops * reps / (ms * 1e-3) / 1e12;
    printf("%-58s %8.1f TFLOP/s  clock %.3f GHz  = %.2f us per slab and CU\n", what, tf, clk.empty() ? 0.0 : clk[clk.size() / 2],
           2.0 * 256 * 256 * 64 * cus / (tf * 1e12) * 1e6);
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv)
{
    const int slabs = argc > 1 ? atoi(argv[1]) : 4000;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    _Float16* rnd; Out* out;
    const size_t n = 8 * STAGE / 2;
    CK(hipMalloc(&rnd, n * 2)); CK(hipMalloc(&out, (size_t)cus * 4 * sizeof(Out)));
    std::vector<_Float16> h(n);
    unsigned long long z = 0x6E77CA11ull;
    for (size_t i = 0; i < n; ++i) { z = z * 6364136223846793005ull + 1442695040888963407ull; h[i] = (_Float16)((float)(int)((z >> 40) & 2047) / 1024.0f - 1.0f); }
    CK(hipMemcpy(rnd, h.data(), n * 2, hipMemcpyHostToDevice));
    _Float16* stream;                                                // 8 MiB + slack of random halves: the LDS-DMA source of mode 4
    const size_t sn = (size_t)(128 * 65536 + 65536) / 2;
    CK(hipMalloc(&stream, sn * 2));
    for (size_t off = 0; off < sn; off += n) CK(hipMemcpy(stream + off, h.data(), std::min(n, sn - off) * 2, hipMemcpyHostToDevice));
    printf("%s, %d CUs, %d slabs of 64 halves per workgroup, one workgroup per CU, random operands\n", prop.gcnArchName, cus, slabs);
    for (int r = 0; r < 2; ++r) {
        if (run<2>(rnd, out, cus, slabs, "bare: 8 waves, 128 x 64 per wave, fragments kept in registers")) return 1;
        if (run<0>(rnd, out, cus, slabs, "shipped tile: 8 waves (2 / SIMD), 128 x 64 per wave, reads + MFMAs")) return 1;
        if (run<1>(rnd, out, cus, slabs, "alternative: 4 waves (1 / SIMD), 128 x 128 per wave, reads + MFMAs")) return 1;
        if (run_pp<0>(rnd, stream, out, cus, slabs, "ping-pong phases, reads pipelined INTO the compute phase, no DMA")) return 1;
        if (run_pp<1>(rnd, stream, out, cus, slabs, "... + the LDS-DMA stream issued in the other phase (64 KiB per slab)")) return 1;
    }
    return 0;
}
