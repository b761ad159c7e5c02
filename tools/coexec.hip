// Microbenchmark: do the f32 MFMA (v_mfma_f32_32x32x2_f32, 64 FLOP/clk/SIMD) and the f32 VALU (v_fma_f32 / v_pk_fma_f32,
// 64 FLOP/clk/SIMD) of one SIMD run at the same time when they come from two different waves?  If they do, an exact-f32
// distance tile could draw FLOPs from both pipes (the fmaf chain of the VALU is the same arithmetic as the MFMA's).
//   hipcc -O3 --offload-arch=gfx950 tools/coexec.hip -o gpurun_out/coexec && gpurun_out/coexec
// Workgroup = 512 threads = 8 waves = 2 per SIMD.  mode bit 0: waves 0-3 run MFMAs; bit 1: waves 4-7 run VALU FMAs
// (bit 2: packed form); a wave whose bit is off exits at once.  mode 8: all 8 waves MFMA; mode 9: all 8 waves VALU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ void mfma_body(int iters, float* out) {
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f - threadIdx.x * 1e-3f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            asm volatile("v_mfma_f32_32x32x2_f32 %0, %4, %5, %0\n\t"
                         "v_mfma_f32_32x32x2_f32 %1, %4, %5, %1\n\t"
                         "v_mfma_f32_32x32x2_f32 %2, %4, %5, %2\n\t"
                         "v_mfma_f32_32x32x2_f32 %3, %4, %5, %3\n\t"
                         : "+v"(c0), "+v"(c1), "+v"(c2), "+v"(c3) : "v"(a), "v"(b));
        }
    }
    float s = 0;
    for (int j = 0; j < 16; ++j) s += c0[j] + c1[j] + c2[j] + c3[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <bool PK>
__device__ __forceinline__ void valu_body(int iters, float* out) {
    // 32 independent accumulators (16 pairs for the packed form); one loop pass = 64 plain FMAs per lane either way.
    f32x2 c[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) c[j] = f32x2{0.f, 0.f};
    f32x2 a = {1.0f + threadIdx.x * 1e-3f, 0.999f}, b = {0.5f, 0.25f - threadIdx.x * 1e-3f};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                if (PK) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(c[j]) : "v"(a), "v"(b));
                else {
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c[j].x) : "v"(a.x), "v"(b.x));
                    asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c[j].y) : "v"(a.y), "v"(b.y));
                }
            }
        }
    }
    float s = 0;
    for (int j = 0; j < 16; ++j) s += c[j].x + c[j].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

__global__ __launch_bounds__(512, 2) void coexec(int mode, int it_mfma, int it_valu, float* out) {
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (mode == 8) { mfma_body(it_mfma, out); return; }
    if (mode == 9) { valu_body<false>(it_valu, out); return; }
    if (mode == 10) { valu_body<true>(it_valu, out); return; }
    if (w < 4) { if (mode & 1) mfma_body(it_mfma, out); }
    else if (mode & 2) { if (mode & 4) valu_body<true>(it_valu, out); else valu_body<false>(it_valu, out); }
}

int main() {
    hipDeviceProp_t p; CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    float* out; CHECK(hipMalloc(&out, sizeof(float) * 512 * cus));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    // per wave: MFMA pass = 16 MFMAs x 4096 FLOP = 65536 FLOP (ideal 1024 cycles); VALU pass = 64 FMA x 64 lanes x 2 = 8192 FLOP (ideal 128 cycles)
    const int it_m = 20000, it_v = 160000;
    struct { int mode; const char* name; } runs[] = {
        {1, "MFMA only, waves 0-3 (1 wave/SIMD)"}, {2, "v_fma only, waves 4-7 (1 wave/SIMD)"}, {6, "v_pk_fma only, waves 4-7"},
        {3, "MFMA waves 0-3 + v_fma waves 4-7"}, {7, "MFMA waves 0-3 + v_pk_fma waves 4-7"},
        {8, "MFMA, all 8 waves"}, {9, "v_fma, all 8 waves"}, {10, "v_pk_fma, all 8 waves"}};
    for (auto& r : runs) {
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(coexec, dim3(cus), dim3(512), 0, 0, r.mode, it_m, it_v, out);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
            float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
            if (rep == 0) continue;
            double fm = 0, fv = 0;
            const double wm = (r.mode == 8) ? 8 : ((r.mode < 8 && (r.mode & 1)) ? 4 : 0);
            const double wv = (r.mode == 9 || r.mode == 10) ? 8 : ((r.mode < 8 && (r.mode & 2)) ? 4 : 0);
            fm = wm * cus * (double)it_m * 65536.0; fv = wv * cus * (double)it_v * 8192.0;
            printf("%-44s %8.2f ms   MFMA %7.1f TF   VALU %7.1f TF   sum %7.1f TF\n", r.name, ms, fm / ms * 1e-9, fv / ms * 1e-9, (fm + fv) / ms * 1e-9);
        }
    }
    return 0;
}
