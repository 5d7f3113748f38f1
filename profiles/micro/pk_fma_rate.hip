// Micro-benchmark: is v_pk_fma_f32 (two fp32 FMAs per lane) issued at the rate of v_fma_f32 on gfx950, i.e. does packing halve VALU time?
//   hipcc --offload-arch=gfx950 -O3 -o pk_fma_rate pk_fma_rate.hip && ./pk_fma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int PK>
__global__ void __launch_bounds__(256) k(float *out, int iters)
{
    const float w = 1.0001f, b = 0.5f;
    if (PK) {
        f32x2 x[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) x[j] = f32x2{threadIdx.x * 1e-3f + j, threadIdx.x * 2e-3f + j};
        const f32x2 ww = {w, w}, bb = {b, b};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] = __builtin_elementwise_fma(x[j], ww, bb);
        }
        float s = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += x[j].x + x[j].y;
        if (s == 1.2345f) out[threadIdx.x] = s;
    } else {
        float x[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) x[j] = threadIdx.x * 1e-3f + j;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int j = 0; j < 16; ++j) x[j] = fmaf(x[j], w, b);
        }
        float s = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) s += x[j];
        if (s == 1.2345f) out[threadIdx.x] = s;
    }
}

int main()
{
    float *out;
    (void)hipMalloc(&out, 4096);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int blocks = 256 * 8, iters = 20000;
    for (int pk = 0; pk < 2; ++pk) {
        if (pk) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, 10); else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, 10);
        (void)hipEventRecord(e0);
        if (pk) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters); else hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters);
        (void)hipEventRecord(e1);
        (void)hipEventSynchronize(e1);
        float ms = 0;
        (void)hipEventElapsedTime(&ms, e0, e1);
        const double fma = (double)blocks * 256 * iters * 8 * 16;       // fp32 FMAs (both variants do the same number)
        printf("%s: %8.3f ms, %.1f TFLOP/s fp32 (same FMA count; %d instructions per loop trip)\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ", ms, 2 * fma / ms / 1e9, pk ? 64 : 128);
    }
    return 0;
}
