// Micro-benchmark: cost of LDS float atomics (ds_add_f32, no return) on gfx950 against the number of active lanes and of address
// collisions, next to ds_add_u32 and to a plain read-add-write.  One 256-thread block per CU-slot, 4 blocks per CU.
//   hipcc --offload-arch=gfx950 -O3 -o lds_atomic_cost lds_atomic_cost.hip && ./lds_atomic_cost
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>     // 0 ds_add_f32, 1 ds_add_u32, 2 read-add-write (not atomic), 3 ds_write only
__global__ void __launch_bounds__(256) k(float *out, int iters, int active, int distinct)
{
    __shared__ float tile[4][2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *t = tile[wave];
    for (int e = lane; e < 2048; e += 64) t[e] = 0.0f;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // `distinct` different addresses among the active lanes (lane % distinct), spread over the banks
    const int slot = (lane % distinct) * 33 & 2047;
    for (int i = 0; i < iters; ++i) {
        const int a = (slot + i * 67) & 2047;
        if (lane < active) {
            if (OP == 0) (void)__hip_atomic_fetch_add(t + a, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            else if (OP == 1) (void)__hip_atomic_fetch_add(reinterpret_cast<unsigned *>(t) + a, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            else if (OP == 2) { t[a] = t[a] + 1.0f; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
            else t[a] = (float)i;
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float s = 0.0f;
    for (int e = lane; e < 2048; e += 64) s += t[e];
    if (s == 1.2345f) out[threadIdx.x] = s;
}

template <int OP>
static void run(const char *name, float *out, int active, int distinct)
{
    const int blocks = 1024, iters = 4000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, 100, active, distinct);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(k<OP>, dim3(blocks), dim3(256), 0, 0, out, iters, active, distinct);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    // per CU: blocks/256 blocks x 4 waves x iters instructions; cycles at 2.4 GHz
    const double per_cu = (double)blocks / 256 * 4 * iters;
    printf("%-14s active %2d distinct %2d : %8.3f ms  -> %6.1f cycles per wave instruction per CU (2.4 GHz)\n", name, active, distinct, ms, ms * 1e-3 * 2.4e9 / per_cu);
}

int main()
{
    float *out;
    (void)hipMalloc(&out, 4096);
    const int cfg[][2] = {{64, 64}, {32, 32}, {16, 16}, {4, 4}, {64, 16}, {64, 4}, {64, 1}, {16, 4}};
    for (auto &c : cfg) run<0>("ds_add_f32", out, c[0], c[1]);
    for (auto &c : cfg) run<1>("ds_add_u32", out, c[0], c[1]);
    run<2>("read-add-write", out, 64, 64);
    run<2>("read-add-write", out, 16, 16);
    run<3>("ds_write", out, 64, 64);
    return 0;
}
