// Micro-benchmark: does the SCOPE of a global float atomic change what it costs on gfx950?
//
// atomicAdd() is an agent-scope atomic: with eight XCDs whose L2s are not coherent with each other the compiler marks it sc1 and it is
// carried out past the L2 (21 G line-transactions/s for the whole device, profiles/r02_micro_atomic_cost.txt).  A workgroup-scope
// atomic carries no sc bit and is carried out in the L2 of the XCD the wave runs on -- which is only correct when every wave that
// adds to a buffer runs on the SAME XCD, e.g. one private copy of the buffer per XCD (HW_REG_XCC_ID picks it) that a later kernel
// sums.  This program measures both and CHECKS the private-copy scheme: every wave adds 1.0 at pseudo-random floats of copy[xcc]
// with workgroup-scope atomics and of a reference buffer with agent-scope atomics; after the kernel the sum of the eight copies must
// equal the reference in every float.
//   hipcc --offload-arch=gfx950 -O3 -o atomic_scope atomic_scope.hip && ./atomic_scope
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__device__ __forceinline__ unsigned mix(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 15u; }

// mode 0: agent scope into buf; 1: workgroup scope into copy[xcc]; 2: both (the check); 3: wavefront scope into copy[xcc]
// L lanes per group of consecutive floats, groups at random 64-byte lines
__global__ void __launch_bounds__(256) k(float *buf, float *copies, size_t copy_floats, unsigned lines, int iters, int L, int mode)
{
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int g = lane / L, e = lane % L;
    float *mine = copies + (size_t)xcc_id() * copy_floats;
    for (int i = 0; i < iters; ++i) {
        const unsigned h = mix(wave * 9781u + i * 6271u);
        const unsigned line = mix(h + g * 77u) % lines;
        const size_t o = (size_t)line * 16 + e;
        if (mode == 0 || mode == 2) atomicAdd(buf + o, 1.0f);
        if (mode == 1 || mode == 2) __hip_atomic_fetch_add(mine + o, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (mode == 3) __hip_atomic_fetch_add(mine + o, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
}

int main()
{
    // two buffer sizes: 1.5 MB (the density-gradient images: L2-resident) and 36 MB (the colour-gradient planes: beyond one XCD's 4 MB L2)
    for (size_t BUF : {(size_t)3 * 512 * 1024, (size_t)36 * 1024 * 1024}) {
        float *buf, *copies;
        hipMalloc(&buf, BUF);
        hipMalloc(&copies, 8 * BUF);
        const unsigned lines = BUF / 64;
        const size_t copy_floats = BUF / 4;
        const int blocks = 2048, iters = 400;
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        printf("buffer %.1f MB (x8 private copies)\n", BUF / 1048576.0);
        printf("  %-58s %10s %12s\n", "pattern", "ms", "G lines/s");
        struct Cfg { const char *name; int L, mode; };
        const Cfg cfgs[] = {
            {"agent scope, 64 lanes on 64 random lines", 1, 0},
            {"agent scope, 4 x whole lines", 16, 0},
            {"workgroup scope, 64 lanes on 64 random lines", 1, 1},
            {"workgroup scope, 16 quads", 4, 1},
            {"workgroup scope, 4 x whole lines", 16, 1},
            {"wavefront scope, 4 x whole lines", 16, 3},
        };
        for (const Cfg &c : cfgs) {
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, buf, copies, copy_floats, lines, 20, c.L, c.mode);
            hipEventRecord(e0);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, buf, copies, copy_floats, lines, iters, c.L, c.mode);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            const double instr = (double)blocks * 4 * iters;
            printf("  %-58s %10.3f %12.2f\n", c.name, ms, instr * (64 / c.L) / ms / 1e6);
        }
        // the check: private copies summed == agent-scope reference, float for float
        for (int L : {1, 16}) {
            hipMemset(buf, 0, BUF);
            hipMemset(copies, 0, 8 * BUF);
            hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, buf, copies, copy_floats, lines, 100, L, 2);
            hipDeviceSynchronize();
            std::vector<float> ref(copy_floats), cp(8 * copy_floats);
            hipMemcpy(ref.data(), buf, BUF, hipMemcpyDeviceToHost);
            hipMemcpy(cp.data(), copies, 8 * BUF, hipMemcpyDeviceToHost);
            size_t bad = 0, used[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            double total = 0;
            for (size_t i = 0; i < copy_floats; ++i) {
                float s = 0;
                for (int x = 0; x < 8; ++x) { s += cp[x * copy_floats + i]; used[x] += cp[x * copy_floats + i] != 0.0f; }
                bad += s != ref[i];
                total += ref[i];
            }
            printf("  check (L = %2d): %zu of %zu floats differ from the agent-scope reference; total %.0f (expected %.0f); floats touched per XCD copy:",
                   L, bad, copy_floats, total, (double)blocks * 4 * 100 * 64);
            for (int x = 0; x < 8; ++x) printf(" %zu", used[x]);
            printf("\n");
        }
        hipFree(buf); hipFree(copies);
    }
    return 0;
}
