// A stand-in for RCCL's all-gather kernel on ONE GPU (profiles/exp_exchange_contention.py): `wgs` workgroups of 256 threads that each hold their CU slot
// for `ns` nanoseconds (wall clock), with a register footprint that cannot share a CU with the level-3 render workgroup (12 waves x 168 registers = 504 of
// a SIMD's 512: a wave that needs more than 8 registers does not fit beside it -- RCCL's kernels need ~100).  What RCCL's kernel does to the render
// pipeline on a real node: it takes CUs only where a render workgroup has left, and keeps them while it waits for its peers.
//   hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o profiles/micro/libcu_holder.so profiles/micro/cu_holder.hip
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ void __launch_bounds__(256) cu_holder_kernel(long long ns, int *sink)
{
    // v127 written: the kernel is allocated >= 128 registers per lane (two waves per SIMD at most, none beside three 168-register waves)
    asm volatile("v_mov_b32 v127, 0" ::: "v127");
    const long long t0 = (long long)wall_clock64();          // 100 MHz constant clock: 10 ns per tick
    while (((long long)wall_clock64() - t0) * 10 < ns) __builtin_amdgcn_s_sleep(8);
    if (sink && threadIdx.x == 0 && ns < 0) sink[blockIdx.x] = 1;
}

extern "C" int cu_holder_launch(int wgs, long long ns, void *stream)
{
    hipLaunchKernelGGL(cu_holder_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, ns, (int *)nullptr);
    return (int)hipGetLastError();
}
