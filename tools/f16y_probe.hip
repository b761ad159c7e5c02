// tools/f16y_probe.hip -- which K-loop STRUCTURE holds the bare MFMA rate once the LDS-DMA stream and the barriers are in it?
//
// Round 4's probe (tools/lds_mfma_probe.hip) priced two things in isolation: eight free-running waves (two per SIMD, fragment reads + MFMAs,
// no barrier, no DMA) hold the bare rate; the ping-pong phase structure (one group of four waves computes while the other loads, two barriers
// per slab) loses 14 % before any DMA is issued and 8 % more with the DMA stream.  The loss of the ping-pong form is structural: at any time
// only ONE wave per SIMD feeds the matrix pipe, and every non-MFMA instruction that wave issues is a bubble nobody fills.
//
// This probe prices the arm that keeps two issuing waves per SIMD: all eight waves run ONE program over the SAME LDS image as the shipped
// kernel (two 64 KiB stages of 128-B rows, 16-B chunk c of row r in slot c ^ ((r >> 1) & 7)), with ONE barrier per slab placed in the MIDDLE
// of the slab's work ("mid-slab barrier"):
//
//     iteration s (between barrier M(s-1) and barrier M(s)), every wave:
//        phase X:  32 MFMAs of (slab s-1, k-step 1) from fragment set 1  ||  its 8 LDS-DMA instructions of slab s+1 -> stage (s+1)&1
//                                                                        ||  12 fragment reads of (slab s, k-step 0) -> set 0
//        phase Y:  32 MFMAs of (slab s,   k-step 0) from fragment set 0  ||  12 fragment reads of (slab s, k-step 1) -> set 1
//        s_waitcnt lgkmcnt(0) vmcnt(0);  s_barrier  = M(s)
//
//   M(s) certifies (a) every wave's share of slab s+1 has landed (each waited for its own before arriving; requested right after M(s-1), a
//   whole iteration earlier) -> its fragments may be read after M(s);  (b) every wave holds ALL its fragments of slab s in registers
//   (k-step 0 consumed, k-step 1 in set 1) -> stage s&1 may be refilled with slab s+2 right after M(s).
//   A wave arrives at the barrier with 32 MFMAs' worth of operands in registers: nothing drains, the pipe restarts at once.
//
// Modes (all: 256 x 256 workgroup tile, 8 waves 2 x 4, a wave 128 x 64 = 8 x 4 accumulators of v_mfma_f32_16x16x32_f16, random operands):
//   bare     fragments kept in registers, MFMAs only
//   free     reads + MFMAs, no barrier, no DMA                                   (round 4: = bare)
//   pp       ping-pong phases + DMA in the other phase                           (round 4's model of the shipped kernel)
//   mid0     mid-slab barrier, no DMA
//   mid1     mid-slab barrier + DMA (all 8 instructions in phase X), compiler's instruction order
//   mid3     ... explicit interleave (sched_group_barrier): per 4 MFMAs one DMA instruction and two fragment reads
//   mid5/7   mid1 / mid3 with s_setprio 1 around nothing (both waves equal) -- not built; priority makes no sense with one program
//   mid9     mid1 with the wait for the DMA moved BEHIND the barrier's certification (vmcnt(8) -- a third stage would allow it; WRONG data, timing only)
//   hipcc -O3 --offload-arch=gfx950 tools/f16y_probe.hip -o ab/f16y_probe && ab/f16y_probe [slabs]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
struct Out { unsigned long long cycles, realtime; float sink; unsigned wait[8]; };

#define STAGE 65536
#define LDS_AS __attribute__((address_space(3)))

// ---- reference points (as in lds_mfma_probe.hip) ------------------------------------------------------------------------------------
template <int MODE>   // 0 free-running reads + MFMAs, 2 bare (A fragment held over 4 MFMAs); bare in other issue orders: 3 snake, 4 B held over 8, 5 both operands change every time
__global__ __launch_bounds__(512, 2) void probe_free(const _Float16* __restrict__ rnd, int slabs, Out* out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, c16 = l & 15;
    for (int i = tid; i < STAGE / 16; i += 512) ((f16x8*)smem)[i] = ((const f16x8*)rnd)[(blockIdx.x % 8) * (STAGE / 16) + i];
    __syncthreads();
    const int wr = w >> 2, wc = w & 3;
    const int swz = (c16 >> 1) & 7;
    const int offA0 = (wr * 128 + c16) * 128 + ((g ^ swz) << 4);
    const int offB0 = 32768 + (wc * 64 + c16) * 128 + ((g ^ swz) << 4);
    f32x4 acc[8][4];
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) acc[m][n] = zero;
    f16x8 fa[8], fb[4];
    if (MODE >= 2) {
#pragma unroll
        for (int m = 0; m < 8; ++m) fa[m] = *(const f16x8*)(smem + offA0 + m * 2048);
#pragma unroll
        for (int n = 0; n < 4; ++n) fb[n] = *(const f16x8*)(smem + offB0 + n * 2048);
    }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int s = 0; s < slabs; ++s) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            if (MODE < 2) {
#pragma unroll
                for (int m = 0; m < 8; ++m) fa[m] = *(const f16x8*)(smem + (offA0 ^ (ks << 6)) + m * 2048);
#pragma unroll
                for (int n = 0; n < 4; ++n) fb[n] = *(const f16x8*)(smem + (offB0 ^ (ks << 6)) + n * 2048);
            }
#pragma unroll
            for (int i = 0; i < 32; ++i) {
                // issue order of the 32 MFMAs of a k-step: which operand registers two consecutive instructions share
                const int m = MODE == 4 ? ((i >> 3) & 1 ? 7 - (i & 7) : (i & 7)) : MODE == 5 ? (i & 7) : (i >> 2);
                const int n = MODE == 4 ? (i >> 3) : MODE == 5 ? ((i + (i >> 3)) & 3) : MODE == 3 ? ((i >> 2) & 1 ? 3 - (i & 3) : (i & 3)) : (i & 3);
                acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[m], fb[n], acc[m][n], 0, 0, 0);
                if (MODE >= 3) __builtin_amdgcn_sched_barrier(0);       // keep the order as written
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) sum += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    if (tid == 0) { out[blockIdx.x].cycles = t1 - t0; out[blockIdx.x].realtime = r1 - r0; }
    if (sum == 12345.678f) out[blockIdx.x].sink = sum;
}

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
    if (grp == 1) __builtin_amdgcn_s_barrier();
    for (int s = 0; s < slabs; ++s) {
        const int st = s & 1;
        __builtin_amdgcn_s_setprio(2);
        read_k(1, st, 1);
        mfma_k(0);
        read_k(0, st ^ 1, 0);
        mfma_k(1);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        if (DMA) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int soff = ((s * 8 + wu) & 127) * 65536;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(smem + (st ^ 1) * STAGE + (wu * 8 + i) * 1024), 16, (unsigned)(l * 16), soff + i * 1024, 0, 0);
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

// ---- the mid-slab-barrier arm ------------------------------------------------------------------------------------------------------
// V bit 0: the DMA stream; bit 1: explicit interleave; bit 3: wait vmcnt(8) instead of 0 (timing of a deeper ring; data WRONG);
// bit 4: DMA instructions split 4 + 4 over the two phases; bit 5: group 1 (waves 4-7) issues its DMA in phase Y, group 0 in phase X (staggered)
template <int V>
__global__ __launch_bounds__(512, 2) void probe_mid(const _Float16* __restrict__ rnd, const _Float16* __restrict__ stream, int slabs, Out* out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];      // two 64 KiB stages
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, c16 = l & 15;
    for (int i = tid; i < 2 * STAGE / 16; i += 512) ((f16x8*)smem)[i] = ((const f16x8*)rnd)[(blockIdx.x % 4) * (2 * STAGE / 16) + i];
    __syncthreads();
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int wr = wu >> 2, wc = wu & 3;
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
        for (int n = 0; n < 4; ++n) fb[buf][n] = *(const f16x8*)(smem + stage * STAGE + (offB0 ^ (ks << 6)) + n * 2048);
#pragma unroll
        for (int m = 0; m < 8; ++m) fa[buf][m] = *(const f16x8*)(smem + stage * STAGE + (offA0 ^ (ks << 6)) + m * 2048);
    };
    auto mfma_k = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[buf][m], fb[buf][n], acc[m][n], 0, 0, 0);
    };
    auto dma = [&](int stage, int s, int i0, int i1) __attribute__((always_inline)) {
        const int soff = ((s * 8 + wu) & 127) * 65536;
#pragma unroll
        for (int i = i0; i < i1; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(smem + stage * STAGE + (wu * 8 + i) * 1024), 16, (unsigned)(l * 16), soff + i * 1024, 0, 0);
    };
    constexpr bool DMA = V & 1, ILV = V & 2, LATE = V & 8, SPLIT = V & 16, STAG = V & 32, STAMP = V & 64;
    unsigned long long twait = 0;
    // prologue: slab 0 sits in stage 0 (filled above); set 0 <- (0, k0); MFMA k0 || set 1 <- (0, k1); M(0)
    read_k(0, 0, 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    read_k(1, 0, 1);
    mfma_k(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int s = 1; s <= slabs; ++s) {
        const int st = s & 1;
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase X
        if (DMA) {
            if (STAG) { if (wr == 0) dma(st ^ 1, s, 0, 8); }
            else dma(st ^ 1, s, 0, SPLIT ? 4 : 8);
        }
        read_k(0, st, 0);
        mfma_k(1);
        if (ILV) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // 4 MFMA
                if (DMA) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   // 1 VMEM read (the LDS-DMA)
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);   // 2 DS reads
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- phase Y
        if (DMA) {
            if (STAG) { if (wr == 1) dma(st ^ 1, s, 0, 8); }
            else if (SPLIT) dma(st ^ 1, s, 4, 8);
        }
        read_k(1, st, 1);
        mfma_k(0);
        if (ILV) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
                if (DMA && (SPLIT || STAG)) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if (LATE) asm volatile("s_waitcnt lgkmcnt(0) vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
        if (STAMP) {
            const unsigned long long b0 = __builtin_amdgcn_s_memtime();
            __builtin_amdgcn_s_barrier();
            twait += __builtin_amdgcn_s_memtime() - b0;
        } else
            __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    mfma_k(1);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) sum += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    if (tid == 0) { out[blockIdx.x].cycles = t1 - t0; out[blockIdx.x].realtime = r1 - r0; }
    if (STAMP && l == 0) out[blockIdx.x].wait[wu] = (unsigned)(twait / (unsigned long long)slabs);
    if (sum == 12345.678f) out[blockIdx.x].sink = sum;
}

// ---- the same schedule with a SOFT barrier: a monotone arrival counter in LDS instead of s_barrier -------------------------------------
// A wave posts its arrival at M(s) (ds_add, no return) and goes on with work that needs no certification -- the MFMAs of phase X, whose
// operands it holds -- and only then checks that all eight have arrived (ds_read + compare; retried if not), before it requests slab s+2 and
// reads slab s+1.  Nobody stops unless somebody is more than POLL MFMAs behind.  V bit 0: DMA; POLL = MFMAs of phase X issued before the poll.
template <int V, int POLL>
__global__ __launch_bounds__(512, 2) void probe_soft(const _Float16* __restrict__ rnd, const _Float16* __restrict__ stream, int slabs, Out* out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];      // two 64 KiB stages + the counter
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, c16 = l & 15;
    for (int i = tid; i < 2 * STAGE / 16; i += 512) ((f16x8*)smem)[i] = ((const f16x8*)rnd)[(blockIdx.x % 4) * (2 * STAGE / 16) + i];
    LDS_AS unsigned* const cnt = (LDS_AS unsigned*)(smem + 2 * STAGE);
    if (tid == 0) *cnt = 0;
    __syncthreads();
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int wr = wu >> 2, wc = wu & 3;
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
        for (int n = 0; n < 4; ++n) fb[buf][n] = *(const f16x8*)(smem + stage * STAGE + (offB0 ^ (ks << 6)) + n * 2048);
#pragma unroll
        for (int m = 0; m < 8; ++m) fa[buf][m] = *(const f16x8*)(smem + stage * STAGE + (offA0 ^ (ks << 6)) + m * 2048);
    };
    auto mfma_rows = [&](int buf, int m0, int m1) __attribute__((always_inline)) {
#pragma unroll
        for (int m = m0; m < m1; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[buf][m], fb[buf][n], acc[m][n], 0, 0, 0);
    };
    auto dma = [&](int stage, int s) __attribute__((always_inline)) {
        const int soff = ((s * 8 + wu) & 127) * 65536;
#pragma unroll
        for (int i = 0; i < 8; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(smem + stage * STAGE + (wu * 8 + i) * 1024), 16, (unsigned)(l * 16), soff + i * 1024, 0, 0);
    };
    constexpr bool DMA = V & 1;
    unsigned long long spins = 0;
    read_k(0, 0, 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    read_k(1, 0, 1);
    mfma_rows(0, 0, 8);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (l == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);          // post M(0)
    for (int s = 1; s <= slabs; ++s) {
        const int st = s & 1;
        __builtin_amdgcn_sched_barrier(0);
        mfma_rows(1, 0, POLL / 4);                                   // phase X, the part before the poll
        __builtin_amdgcn_sched_barrier(0);
        while (true) {                                               // all eight at M(s-1)?
            const unsigned c = __hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (__builtin_amdgcn_readfirstlane(c) >= 8u * (unsigned)s) break;
            ++spins;
            __builtin_amdgcn_s_sleep(1);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (DMA) dma(st ^ 1, s);
        read_k(0, st, 0);
        mfma_rows(1, POLL / 4, 8);
        __builtin_amdgcn_sched_barrier(0);
        read_k(1, st, 1);                                            // phase Y
        mfma_rows(0, 0, 8);
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
        if (l == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);      // post M(s)
    }
    mfma_rows(1, 0, 8);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) sum += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    if (tid == 0) { out[blockIdx.x].cycles = t1 - t0; out[blockIdx.x].realtime = r1 - r0; }
    if (l == 0) out[blockIdx.x].wait[wu] = (unsigned)(spins * 1000ull / (unsigned long long)slabs);       // failed polls per 1000 slabs
    if (sum == 12345.678f) out[blockIdx.x].sink = sum;
}

// ---- mid-slab barrier + a PRIORITY LADDER -------------------------------------------------------------------------------------------
// The stamps of probe_mid show why one barrier per slab costs 10-12 %: the SIMD's arbiter prefers the OLDER wave, so waves 0-3 issue their
// 64 MFMAs in the first ~57 % of the interval and sit in the barrier for the rest (875 of 2048 cycles), while their partners 4-7 run ALONE --
// the single-wave regime of the ping-pong form.  A wave whose priority FALLS as it advances through the interval hands the pipe to a
// partner that is behind: the interval's 64 MFMAs per wave are four chunks of 16 issued at s_setprio 3, 2, 1, 0.  Ties go to the older wave, so
// the order is A0 B0 A1 B1 A2 B2 A3 B3 and only the last chunk (1/8 of the interval) runs alone.
// LAD 0: no priorities (= mid19);  1: 3,2,1,0 in chunks of 16;  2: 1,1,0,0 (one level per phase);  3: chunks of 24,16,16,8 at 3,2,1,0
template <int DMA, int LAD>
__global__ __launch_bounds__(512, 2) void probe_ladder(const _Float16* __restrict__ rnd, const _Float16* __restrict__ stream, int slabs, Out* out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];      // two 64 KiB stages
    const int tid = threadIdx.x, w = tid >> 6, l = tid & 63, g = l >> 4, c16 = l & 15;
    for (int i = tid; i < 2 * STAGE / 16; i += 512) ((f16x8*)smem)[i] = ((const f16x8*)rnd)[(blockIdx.x % 4) * (2 * STAGE / 16) + i];
    __syncthreads();
    const int wu = __builtin_amdgcn_readfirstlane(w);
    const int wr = wu >> 2, wc = wu & 3;
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
    auto read_B = [&](int buf, int stage, int ks) __attribute__((always_inline)) {
#pragma unroll
        for (int n = 0; n < 4; ++n) fb[buf][n] = *(const f16x8*)(smem + stage * STAGE + (offB0 ^ (ks << 6)) + n * 2048);
    };
    auto read_A = [&](int buf, int stage, int ks, int m0, int m1) __attribute__((always_inline)) {
#pragma unroll
        for (int m = m0; m < m1; ++m) fa[buf][m] = *(const f16x8*)(smem + stage * STAGE + (offA0 ^ (ks << 6)) + m * 2048);
    };
    auto mfma_rows = [&](int buf, int m0, int m1) __attribute__((always_inline)) {
#pragma unroll
        for (int m = m0; m < m1; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[buf][m], fb[buf][n], acc[m][n], 0, 0, 0);
    };
    auto dma = [&](int stage, int s, int i0, int i1) __attribute__((always_inline)) {
        const int soff = ((s * 8 + wu) & 127) * 65536;
#pragma unroll
        for (int i = i0; i < i1; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (LDS_AS void*)(smem + stage * STAGE + (wu * 8 + i) * 1024), 16, (unsigned)(l * 16), soff + i * 1024, 0, 0);
    };
    // interleave of one chunk: NM MFMAs in groups of 4, each group followed by (at most) one DMA instruction and two fragment reads
    auto pattern = [&](int groups, bool with_dma) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < groups; ++i) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
            if (with_dma) __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        }
    };
#define PRIO(P) do { if (LAD) { __builtin_amdgcn_sched_barrier(0); __builtin_amdgcn_s_setprio(P); __builtin_amdgcn_sched_barrier(0); } } while (0)
    unsigned long long twait = 0;
    constexpr int p0 = LAD == 2 ? 1 : 3, p1 = LAD == 2 ? 1 : 2, p2 = LAD == 2 ? 0 : 1, p3 = 0;
    constexpr int c0 = LAD == 3 ? 6 : 4, c1 = 8;                     // rows of phase X in chunk 0 (chunk 1 = the rest); phase Y likewise with c2
    constexpr int c2 = LAD == 3 ? 6 : 4;                             // LAD 3: X = 24 + 8?  no: chunks 24,16,16,8 = X rows 0-5 | X rows 6-7 + Y rows 0-1 | Y rows 2-5 | Y rows 6-7
    read_B(0, 0, 0); read_A(0, 0, 0, 0, 8);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    read_B(1, 0, 1); read_A(1, 0, 1, 0, 8);
    mfma_rows(0, 0, 8);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int s = 1; s <= slabs; ++s) {
        const int st = s & 1;
        if (LAD == 4) {
            PRIO(3);
            if (DMA) dma(st ^ 1, s, 0, 2);
            read_B(0, st, 0); read_A(0, st, 0, 0, 2);
            mfma_rows(1, 0, 4);
            pattern(4, DMA);
            PRIO(2);
            if (DMA) dma(st ^ 1, s, 2, 4);
            read_A(0, st, 0, 2, 8);
            mfma_rows(1, 4, 8);
            pattern(4, DMA);
            PRIO(1);
            if (DMA) dma(st ^ 1, s, 4, 6);
            read_B(1, st, 1); read_A(1, st, 1, 0, 2);
            mfma_rows(0, 0, 4);
            pattern(4, DMA);
            PRIO(0);
            if (DMA) dma(st ^ 1, s, 6, 8);
            read_A(1, st, 1, 2, 8);
            mfma_rows(0, 4, 8);
            pattern(4, DMA);
        } else if (LAD != 3) {
            // chunk 0: X rows 0-3
            PRIO(p0);
            if (DMA) dma(st ^ 1, s, 0, 4);
            read_B(0, st, 0); read_A(0, st, 0, 0, 2);
            mfma_rows(1, 0, 4);
            pattern(4, DMA);
            // chunk 1: X rows 4-7
            PRIO(p1);
            if (DMA) dma(st ^ 1, s, 4, 8);
            read_A(0, st, 0, 2, 8);
            mfma_rows(1, 4, 8);
            pattern(4, DMA);
            // chunk 2: Y rows 0-3
            PRIO(p2);
            read_B(1, st, 1); read_A(1, st, 1, 0, 2);
            mfma_rows(0, 0, 4);
            pattern(4, false);
            // chunk 3: Y rows 4-7
            PRIO(p3);
            read_A(1, st, 1, 2, 8);
            mfma_rows(0, 4, 8);
            pattern(4, false);
        } else {
            PRIO(3);                                                  // 24: X rows 0-5
            if (DMA) dma(st ^ 1, s, 0, 6);
            read_B(0, st, 0); read_A(0, st, 0, 0, 8);
            mfma_rows(1, 0, 6);
            pattern(6, DMA);
            PRIO(2);                                                  // 16: X rows 6-7, Y rows 0-1
            if (DMA) dma(st ^ 1, s, 6, 8);
            mfma_rows(1, 6, 8);
            read_B(1, st, 1);
            read_A(1, st, 1, 0, 4);
            mfma_rows(0, 0, 2);
            pattern(4, DMA);
            PRIO(1);                                                  // 16: Y rows 2-5
            read_A(1, st, 1, 4, 8);
            mfma_rows(0, 2, 6);
            pattern(4, false);
            PRIO(0);                                                  // 8: Y rows 6-7
            mfma_rows(0, 6, 8);
        }
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt lgkmcnt(0) vmcnt(0)" ::: "memory");
        const unsigned long long b0 = __builtin_amdgcn_s_memtime();
        __builtin_amdgcn_s_barrier();
        twait += __builtin_amdgcn_s_memtime() - b0;
    }
#undef PRIO
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    mfma_rows(1, 0, 8);
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
#pragma unroll
    for (int m = 0; m < 8; ++m)
#pragma unroll
        for (int n = 0; n < 4; ++n) sum += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    if (tid == 0) { out[blockIdx.x].cycles = t1 - t0; out[blockIdx.x].realtime = r1 - r0; }
    if (l == 0) out[blockIdx.x].wait[wu] = (unsigned)(twait / (unsigned long long)slabs);
    if (sum == 12345.678f) out[blockIdx.x].sink = sum;
    (void)c0; (void)c1; (void)c2;
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <typename F>
static int timeit(F launch, Out* out, unsigned grid, int cus, int slabs, const char* what, bool stamps = false)
{
    const double flops = (double)grid * 8 * (slabs + 1) * 64.0 * (2.0 * 16 * 16 * 32);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 30; ++i) launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    const int reps = 40;
    for (int i = 0; i < reps; ++i) launch();
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<Out> h(grid);
    CK(hipMemcpy(h.data(), out, grid * sizeof(Out), hipMemcpyDeviceToHost));
    std::vector<double> clk;
    for (auto& o : h) if (o.realtime) clk.push_back((double)o.cycles / (double)o.realtime * 0.1);
    std::sort(clk.begin(), clk.end());
    const double tf = flops * reps / (ms * 1e-3) / 1e12;
    printf("%-78s %8.1f TFLOP/s  clock %.3f GHz  = %.3f us per slab and CU\n", what, tf, clk.empty() ? 0.0 : clk[clk.size() / 2],
           2.0 * 256 * 256 * 64 * cus / (tf * 1e12) * 1e6);
    if (stamps) {
        double wsum[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (auto& o : h) for (int i = 0; i < 8; ++i) wsum[i] += o.wait[i];
        printf("      per wave 0-7, mean over workgroups (mid: cycles per slab inside s_barrier; soft: failed polls per 1000 slabs):");
        for (int i = 0; i < 8; ++i) printf(" %.0f", wsum[i] / grid);
        printf("\n");
    }
    fflush(stdout);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
    return 0;
}

int main(int argc, char** argv)
{
    const int slabs = argc > 1 ? atoi(argv[1]) : 4000;
    const bool zeros = argc > 2 && atoi(argv[2]) == 1;
    const int keep_bits = argc > 3 ? atoi(argv[3]) : 10;             // explicit mantissa bits kept in the random halves (10 = all): how much of the power is the multiplier array's?
    const int only = argc > 4 ? atoi(argv[4]) : 0;                    // 1: only bare, pp and lad1/1                // all-zero operands toggle nothing: the chip keeps its clock, what is left is the structure
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    _Float16* rnd; Out* out;
    const size_t n = 8 * STAGE / 2;
    const unsigned grid = (unsigned)cus * 4;
    CK(hipMalloc(&rnd, n * 2)); CK(hipMalloc(&out, (size_t)grid * sizeof(Out)));
    std::vector<_Float16> h(n);
    unsigned long long z = 0x6E77CA11ull;
    for (size_t i = 0; i < n; ++i) { z = z * 6364136223846793005ull + 1442695040888963407ull; h[i] = zeros ? (_Float16)0.0f : (_Float16)((float)(int)((z >> 40) & 2047) / 1024.0f - 1.0f); }
    if (keep_bits < 10)
        for (size_t i = 0; i < n; ++i) { unsigned short b; memcpy(&b, &h[i], 2); b &= (unsigned short)(0xFFFFu << (10 - keep_bits)); memcpy(&h[i], &b, 2); }
    CK(hipMemcpy(rnd, h.data(), n * 2, hipMemcpyHostToDevice));
    _Float16* stream;
    const size_t sn = (size_t)(128 * 65536 + 65536) / 2;
    CK(hipMalloc(&stream, sn * 2));
    for (size_t off = 0; off < sn; off += n) CK(hipMemcpy(stream + off, h.data(), std::min(n, sn - off) * 2, hipMemcpyHostToDevice));
    printf("%s, %d CUs, %d slabs of 64 halves per workgroup, one workgroup per CU, %s operands\n", prop.gcnArchName, cus, slabs, zeros ? "ALL-ZERO" : "random");
    printf("explicit mantissa bits kept: %d\n", keep_bits);
    const size_t lds1 = 84 * 1024, lds2 = 2 * STAGE + 1024;
#define RUN_FREE(M, WHAT) if (timeit([&] { hipLaunchKernelGGL(probe_free<M>, dim3(grid), dim3(512), lds1, 0, rnd, slabs, out); }, out, grid, cus, slabs - 1, WHAT)) return 1
#define RUN_PP(D, WHAT) if (timeit([&] { hipLaunchKernelGGL(probe_pp<D>, dim3(grid), dim3(512), lds2, 0, rnd, stream, slabs, out); }, out, grid, cus, slabs - 1, WHAT)) return 1
#define RUN_MID(V, WHAT) if (timeit([&] { hipLaunchKernelGGL(probe_mid<V>, dim3(grid), dim3(512), lds2, 0, rnd, stream, slabs, out); }, out, grid, cus, slabs, WHAT, ((V) & 64) != 0)) return 1
#define RUN_LAD(D, L, WHAT) if (timeit([&] { hipLaunchKernelGGL((probe_ladder<D, L>), dim3(grid), dim3(512), lds2, 0, rnd, stream, slabs, out); }, out, grid, cus, slabs, WHAT, true)) return 1
    for (int r = 0; r < 1; ++r) {
        RUN_FREE(2, "bare: fragments kept in registers");
        RUN_FREE(3, "bare, snake order (one operand shared by every consecutive pair)");
        RUN_FREE(4, "bare, B held over 8 MFMAs, A snakes");
        RUN_FREE(5, "bare, both operands change with every MFMA");
        if (only) {
            RUN_PP(1, "pp: ping-pong phases + LDS-DMA in the other phase (round 4's model of the shipped kernel)");
            RUN_LAD(1, 0, "lad1/0: mid-slab barrier + DMA (4 + 4), no priorities");
            RUN_LAD(1, 1, "lad1/1: mid-slab barrier + DMA (4 + 4), priority 3,2,1,0 per chunk of 16");
            RUN_LAD(1, 4, "lad1/4: mid-slab barrier + DMA (2 + 2 + 2 + 2), priority 3,2,1,0 per chunk of 16");
            RUN_LAD(1, 2, "lad1/2: mid-slab barrier + DMA (4 + 4), priority 1,1,0,0");
            continue;
        }
        RUN_FREE(0, "free: reads + MFMAs, no barrier, no DMA");
        RUN_PP(1, "pp: ping-pong phases + LDS-DMA in the other phase (round 4's model of the shipped kernel)");
        RUN_MID(0, "mid0: mid-slab barrier, no DMA");
        RUN_MID(2, "mid2: mid-slab barrier, no DMA, explicit interleave");
        RUN_MID(1, "mid1: mid-slab barrier + DMA (8 in phase X), compiler order");
        RUN_MID(3, "mid3: mid-slab barrier + DMA (8 in phase X), explicit interleave");
        RUN_MID(17, "mid17: mid-slab barrier + DMA (4 + 4), compiler order");
        RUN_MID(19, "mid19: mid-slab barrier + DMA (4 + 4), explicit interleave");
        RUN_MID(66, "mid66: mid2 + stamps around the barrier");
        RUN_MID(67, "mid67: mid3 + stamps around the barrier");
#define RUN_SOFT(V, P, WHAT) if (timeit([&] { hipLaunchKernelGGL((probe_soft<V, P>), dim3(grid), dim3(512), lds2, 0, rnd, stream, slabs, out); }, out, grid, cus, slabs, WHAT, true)) return 1
        RUN_SOFT(0, 32, "soft0/32: ... polled after 32 MFMAs of phase X, no DMA");
        RUN_SOFT(1, 32, "soft1/32: counter, polled after 32 MFMAs, + DMA");
        RUN_LAD(0, 0, "lad0/0: mid-slab barrier, chunks of 16, no priorities, no DMA");
        RUN_LAD(0, 1, "lad0/1: ... priority 3,2,1,0 per chunk, no DMA");
        RUN_LAD(0, 2, "lad0/2: ... priority 1,1,0,0, no DMA");
        RUN_LAD(0, 3, "lad0/3: ... chunks of 24,16,16,8 at 3,2,1,0, no DMA");
        RUN_LAD(1, 0, "lad1/0: mid-slab barrier + DMA (4 + 4), no priorities");
        RUN_LAD(1, 1, "lad1/1: ... priority 3,2,1,0 per chunk of 16");
        RUN_LAD(1, 2, "lad1/2: ... priority 1,1,0,0");
        RUN_LAD(1, 3, "lad1/3: ... chunks of 24,16,16,8 at 3,2,1,0");
    }
    return 0;
}
