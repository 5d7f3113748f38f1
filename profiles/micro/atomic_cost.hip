// Micro-benchmark: what does a wave-level global_atomic_add_f32 (no return) cost on gfx950 as a function of the lanes that
// take part and of the cache lines they touch?  (The training backward scatters gradients with it: train_density_bwd_kernel.)
// Every wave issues ITERS atomic instructions; per instruction the active lanes form G groups of L consecutive floats, every
// group at a pseudo-random 64-byte-aligned place of a BUF-byte buffer (1.5 MB: the size of the density-gradient images).
//   hipcc --offload-arch=gfx950 -O3 -o atomic_cost atomic_cost.hip && ./atomic_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned mix(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

// L = lanes per group (consecutive floats), active = number of active lanes (multiple of L), spread: 0 = groups at random lines,
// 1 = all groups in ONE random 4 KB window (neighbouring rows of one ray), 2 = every group at the same line (full contention)
__global__ void __launch_bounds__(256) k(float *buf, unsigned lines, int iters, int L, int active, int spread, int store)
{
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int g = lane / L, e = lane % L;
    for (int i = 0; i < iters; ++i) {
        unsigned h = mix(wave * 9781u + i * 6271u);
        unsigned line;
        if (spread == 0) line = mix(h + g * 77u) % lines;
        else if (spread == 1) line = ((h % (lines / 64)) * 64 + mix(h + g) % 64);
        else line = h % lines;
        float *p = buf + (size_t)line * 16 + e;
        if (lane < active) {
            if (store) *p = 1.0f;
            else atomicAdd(p, 1.0f);
        }
    }
}

int main()
{
    const size_t BUF = 3 * 512 * 1024;
    float *buf;
    hipMalloc(&buf, BUF + 4096);
    hipMemset(buf, 0, BUF + 4096);
    const unsigned lines = BUF / 64;
    const int blocks = 2048, iters = 400;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    struct Cfg { const char *name; int L, active, spread, store; };
    const Cfg cfgs[] = {
        {"64 lanes, 64 random lines (1 float each)", 1, 64, 0, 0},
        {"32 lanes, 32 random lines", 1, 32, 0, 0},
        {"16 lanes, 16 random lines", 1, 16, 0, 0},
        {" 4 lanes,  4 random lines", 1, 4, 0, 0},
        {" 1 lane", 1, 1, 0, 0},
        {"64 lanes = 32 pairs (8 B) at random lines   [density scatter]", 2, 64, 0, 0},
        {"64 lanes = 16 quads (16 B) at random lines  [gauge scatter]", 4, 64, 0, 0},
        {"64 lanes = 4 x 16 floats (whole lines) at random lines", 16, 64, 0, 0},
        {"16 lanes = one whole line", 16, 16, 0, 0},
        {"64 lanes = 256 contiguous bytes (4 lines)", 64, 64, 0, 0},
        {"64 lanes, 64 lines inside one 4 KB window", 1, 64, 1, 0},
        {"32 pairs inside one 4 KB window", 2, 64, 1, 0},
        {"16 quads inside one 4 KB window", 4, 64, 1, 0},
        {"64 lanes on one float (same address)", 1, 64, 2, 0},
        {"plain stores: 64 lanes, 64 random lines", 1, 64, 0, 1},
        {"plain stores: 4 x whole lines", 16, 64, 0, 1},
    };
    printf("%-66s %10s %12s %12s %12s\n", "pattern", "ms", "G instr/s", "G lane-op/s", "G groups/s");
    for (const Cfg &c : cfgs) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, buf, lines, 20, c.L, c.active, c.spread, c.store);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, buf, lines, iters, c.L, c.active, c.spread, c.store);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double instr = (double)blocks * 4 * iters;
        printf("%-66s %10.3f %12.2f %12.2f %12.2f\n", c.name, ms, instr / ms / 1e6, instr * c.active / ms / 1e6, instr * (c.active / c.L) / ms / 1e6);
    }
    return 0;
}
