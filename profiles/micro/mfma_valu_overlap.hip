// Micro-benchmark: do fp32 MFMA (v_mfma_f32_16x16x4_f32) and fp32 VALU (v_fma_f32) overlap on one SIMD of gfx950?
// Waves of a 512-thread block (2 per SIMD) run either an MFMA loop, a VALU loop, or one of each per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_overlap mfma_valu_overlap.hip && ./mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>   // per wave role: bit0 = even waves do MFMA, bit1 = odd waves do VALU, ... see main
__global__ void __launch_bounds__(512) k(float *out, int iters, int role_even, int role_odd)
{
    const int wave = threadIdx.x >> 6;
    const int role = (wave & 4) ? role_odd : role_even;      // waves 0-3 -> one per SIMD, waves 4-7 -> second wave per SIMD
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x0 = threadIdx.x * 1e-3f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const float w = 1.0001f, b = 0.5f;
    if (role == 1) {          // fp32 MFMA
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b, a3, 0, 0, 0);
            }
        }
    } else if (role == 2) {   // fp32 VALU
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                x0 = fmaf(x0, w, b); x1 = fmaf(x1, w, b); x2 = fmaf(x2, w, b); x3 = fmaf(x3, w, b);
                x4 = fmaf(x4, w, b); x5 = fmaf(x5, w, b); x6 = fmaf(x6, w, b); x7 = fmaf(x7, w, b);
            }
        }
    } else if (role == 4 || role == 5 || role == 6) {   // one stream: per f32 MFMA, NF = 2 / 4 / 6 independent fp32 VALU fmas
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                f32x4 &a = (u & 3) == 0 ? a0 : (u & 3) == 1 ? a1 : (u & 3) == 2 ? a2 : a3;
                a = __builtin_amdgcn_mfma_f32_16x16x4f32(w, b, a, 0, 0, 0);
                x0 = fmaf(x0, w, b); x1 = fmaf(x1, w, b);
                if (role >= 5) { x2 = fmaf(x2, w, b); x3 = fmaf(x3, w, b); }
                if (role >= 6) { x4 = fmaf(x4, w, b); x5 = fmaf(x5, w, b); }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (role == 7) {   // one stream: per bf16 MFMA 16x16x32, 2 fp32 VALU fmas
        bf16x8 av = {1, 2, 3, 4, 5, 6, 7, 8}, bv = {8, 7, 6, 5, 4, 3, 2, 1};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                f32x4 &a = (u & 3) == 0 ? a0 : (u & 3) == 1 ? a1 : (u & 3) == 2 ? a2 : a3;
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, a, 0, 0, 0);
                x0 = fmaf(x0, w, b); x1 = fmaf(x1, w, b);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else if (role == 3) {   // bf16 MFMA 16x16x32
        bf16x8 av = {1, 2, 3, 4, 5, 6, 7, 8}, bv = {8, 7, 6, 5, 4, 3, 2, 1};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, a3, 0, 0, 0);
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3] + x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}

int main()
{
    float *out;
    hipMalloc(&out, 256 * 512 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    const char *names[] = {"idle", "f32 MFMA 16x16x4 (32/iter)", "f32 VALU fma (128/iter)", "bf16 MFMA 16x16x32 (32/iter)", "32 x (f32 MFMA + 2 fma)", "32 x (f32 MFMA + 4 fma)", "32 x (f32 MFMA + 6 fma)", "32 x (bf16 MFMA + 2 fma)"};
    const int combos[][2] = {{1, 0}, {2, 0}, {3, 0}, {1, 1}, {2, 2}, {1, 2}, {3, 2}, {3, 3}, {1, 3}, {4, 0}, {5, 0}, {6, 0}, {7, 0}, {4, 4}, {6, 6}};
    for (auto &c : combos) {
        k<0><<<256, 512>>>(out, 100, c[0], c[1]);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        k<0><<<256, 512>>>(out, iters, c[0], c[1]);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        printf("wave A per SIMD: %-30s wave B per SIMD: %-30s %8.3f ms  (%.1f cycles per loop iteration at 2.4 GHz)\n", names[c[0]], names[c[1]], ms, ms * 1e-3 * 2.4e9 / iters);
    }
    return 0;
}
