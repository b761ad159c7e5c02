// Is "blocks b and b+8 share an XCD" true for a long grid of unevenly long workgroups, and do an XCD's blocks start in index order?
//   hipcc -O3 --offload-arch=gfx950 tools/xcd_map.hip -o /tmp/xcd_map && /tmp/xcd_map
// Each workgroup (256 threads, 2 per CU by LDS) spins for a pseudo-random 40-80 us and records XCC_ID, CU id, start and end time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256, 2) void probe(unsigned long long* out, int base_ticks)
{
    __shared__ char pad[72 * 1024];                      // 2 workgroups per CU, like the fp32 tile kernel
    pad[threadIdx.x] = 0;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();     // 100 MHz
    const unsigned h = (blockIdx.x * 2654435761u) >> 20;                 // 0..4095
    const unsigned long long dur = base_ticks + (base_ticks * (unsigned long long)h >> 12);   // 1x .. 2x
    while (__builtin_amdgcn_s_memrealtime() - t0 < dur) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));      // HW_REG_XCC_ID[3:0]
        out[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11));      // HW_REG_HW_ID
        out[blockIdx.x * 4 + 2] = t0;
        out[blockIdx.x * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
}

int main()
{
    const int nb = 8 * 64 * 40;                          // 40 super-tiles of 64 blocks per XCD
    unsigned long long* d; CHECK(hipMalloc(&d, nb * 32));
    std::vector<unsigned long long> h(nb * 4);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(nb), dim3(256), 0, 0, d, 4000);          // 40-80 us
        CHECK(hipDeviceSynchronize());
    }
    CHECK(hipMemcpy(h.data(), d, nb * 32, hipMemcpyDeviceToHost));
    // 1. XCC of block b vs b % 8
    int hist[8][8] = {};
    for (int b = 0; b < nb; ++b) hist[b & 7][h[b * 4] & 7]++;
    int consistent = 0;
    for (int r = 0; r < 8; ++r) { int m = 0; for (int x = 0; x < 8; ++x) m = std::max(m, hist[r][x]); consistent += m; }
    printf("blocks whose XCC is the majority XCC of their (b %% 8) class: %d of %d (%.2f %%)\n", consistent, nb, 100.0 * consistent / nb);
    // 2. per class: are blocks started in local-index order, and how many super-tiles (64 consecutive local indices) are alive at once?
    double max_alive = 0, sum_alive = 0; long samples = 0; long inversions = 0, pairs = 0;
    for (int r = 0; r < 8; ++r) {
        std::vector<std::pair<unsigned long long, int>> ev;              // (start, local)
        for (int b = r; b < nb; b += 8) ev.push_back({h[b * 4 + 2], b >> 3});
        for (size_t i = 1; i < ev.size(); ++i) { pairs++; if (ev[i].first < ev[i - 1].first) inversions++; }
        // at each block's start: distinct super-tiles among blocks of this class alive at that time
        for (int b = r; b < nb; b += 8 * 7) {
            const unsigned long long t = h[b * 4 + 2];
            bool seen[64] = {}; int n = 0;
            for (int c = r; c < nb; c += 8) if (h[c * 4 + 2] <= t && h[c * 4 + 3] > t) { const int s = (c >> 3) >> 6; if (!seen[s]) { seen[s] = true; ++n; } }
            max_alive = std::max<double>(max_alive, n); sum_alive += n; samples++;
        }
    }
    printf("start-order inversions between consecutive local indices of a class: %ld of %ld; super-tiles alive at once per class: mean %.2f, max %.0f\n",
           inversions, pairs, sum_alive / samples, max_alive);
    return 0;
}
