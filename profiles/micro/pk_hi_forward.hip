// Micro-benchmark: is the HIGH half of a v_pk_mul_f32 / v_pk_add_f32 result safe to read with a scalar (un-packed) VALU instruction right
// after it on gfx950?  Found while hunting a 1-in-50 000 nondeterminism of the InfoInv NGF_F_SPLIT_BF16 colour pass: the
// compiler's packed code for the sin / cos doubling chain (v_pk_mul_f32, v_pk_add_f32 with op_sel, v_pk_mov_b32, each followed by v_mov /
// v_add / v_mul reading ONE half of the pair) gave timing-dependent results; the same arithmetic as single instructions did not.
//   hipcc --offload-arch=gfx950 -O3 -o pk_hi_forward pk_hi_forward.hip && ./pk_hi_forward
// Each wave runs `iters` rounds of {packed op; dependent scalar op at distance GAP} on varying data and counts results that differ from
// the reference computed with un-packed instructions.  Half of the waves (odd wave ids) can run a bf16 MFMA loop instead, so that the
// packed chain of one wave meets the matrix instructions of its SIMD neighbour (mode 1), or fp32 MFMAs (mode 2).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef short bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int GAP>
__device__ __forceinline__ float chain(float a0, float a1, float b0, float b1)
{
    float r;
    // v[10:11] = (a0, a1), v[12:13] = (b0, b1); packed product -> v[14:15]; then scalar consumers of BOTH halves in the pattern of the
    // compiler's code: v_mov of the high half, v_add of the high half with itself, v_mul with the low half
    if constexpr (GAP == 0)
        asm volatile("v_mov_b32 v10, %1\n\tv_mov_b32 v11, %2\n\tv_mov_b32 v12, %3\n\tv_mov_b32 v13, %4\n\ts_nop 7\n\t"
                     "v_pk_mul_f32 v[14:15], v[10:11], v[12:13]\n\t"
                     "v_mov_b32 v16, v15\n\t"
                     "v_add_f32 v17, v15, v15\n\t"
                     "v_pk_add_f32 v[18:19], v[14:15], v[16:17] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "v_mul_f32 v20, v14, v17\n\t"
                     "v_add_f32 v20, v20, v18\n\t"
                     "v_add_f32 %0, v20, v19\n\t"
                     : "=v"(r) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20");
    else
        asm volatile("v_mov_b32 v10, %1\n\tv_mov_b32 v11, %2\n\tv_mov_b32 v12, %3\n\tv_mov_b32 v13, %4\n\ts_nop 7\n\t"
                     "v_pk_mul_f32 v[14:15], v[10:11], v[12:13]\n\ts_nop 1\n\t"
                     "v_mov_b32 v16, v15\n\t"
                     "v_add_f32 v17, v15, v15\n\ts_nop 1\n\t"
                     "v_pk_add_f32 v[18:19], v[14:15], v[16:17] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\ts_nop 1\n\t"
                     "v_mul_f32 v20, v14, v17\n\t"
                     "v_add_f32 v20, v20, v18\n\t"
                     "v_add_f32 %0, v20, v19\n\t"
                     : "=v"(r) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20");
    return r;
}

__device__ __forceinline__ float reference(float a0, float a1, float b0, float b1)
{
    // the same data flow with single instructions (volatile asm: no re-association, no packing)
    float p0, p1, m16, m17, s18, s19, t;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(a0), "v"(b0));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(a1), "v"(b1));
    m16 = p1;
    asm volatile("v_add_f32 %0, %1, %1" : "=v"(m17) : "v"(p1));
    // v_pk_add with op_sel:[0,1] op_sel_hi:[1,0], neg on src1: lo = src0.lo - src1.hi = p0 - m17 ; hi = src0.hi - src1.lo = p1 - m16
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(s18) : "v"(p0), "v"(m17));
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(s19) : "v"(p1), "v"(m16));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(p0), "v"(m17));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(t) : "v"(t), "v"(s18));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(t) : "v"(t), "v"(s19));
    return t;
}

// pattern 2: the compiler's v_pk_mov_b32 ... op_sel:[1,0] (dst = {src0.hi, src1.lo}) fed by a packed product, consumed by scalar ops, with
// two of the wave's OWN bf16 MFMAs issued right before (their results are not used by the chain)
template <int GAP>
__device__ __forceinline__ float chain2(float a0, float a1, float b0, float b1, bf16x8 fa, bf16x8 fb, f32x4 &acc)
{
    float r;
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb, fa, acc, 0, 0, 0);
    if constexpr (GAP == 0)
        asm volatile("v_mov_b32 v10, %1\n\tv_mov_b32 v11, %2\n\tv_mov_b32 v12, %3\n\tv_mov_b32 v13, %4\n\t"
                     "v_pk_mul_f32 v[14:15], v[10:11], v[12:13] op_sel_hi:[0,1]\n\t"
                     "v_pk_mov_b32 v[16:17], v[10:11], v[14:15] op_sel:[1,0]\n\t"
                     "v_add_f32 v18, v15, v15\n\t"
                     "v_pk_mul_f32 v[16:17], v[16:17], v[14:15]\n\t"
                     "v_mul_f32 v19, v14, v18\n\t"
                     "v_add_f32 v19, v19, v16\n\t"
                     "v_add_f32 %0, v19, v17\n\t"
                     : "=v"(r) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19");
    else
        asm volatile("v_mov_b32 v10, %1\n\tv_mov_b32 v11, %2\n\tv_mov_b32 v12, %3\n\tv_mov_b32 v13, %4\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                     "v_pk_mul_f32 v[14:15], v[10:11], v[12:13] op_sel_hi:[0,1]\n\ts_nop 1\n\t"
                     "v_pk_mov_b32 v[16:17], v[10:11], v[14:15] op_sel:[1,0]\n\ts_nop 1\n\t"
                     "v_add_f32 v18, v15, v15\n\t"
                     "v_pk_mul_f32 v[16:17], v[16:17], v[14:15]\n\ts_nop 1\n\t"
                     "v_mul_f32 v19, v14, v18\n\t"
                     "v_add_f32 v19, v19, v16\n\t"
                     "v_add_f32 %0, v19, v17\n\t"
                     : "=v"(r) : "v"(a0), "v"(a1), "v"(b0), "v"(b1) : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19");
    return r;
}
__device__ __forceinline__ float reference2(float a0, float a1, float b0, float b1)
{
    float p0, p1, m16, m17, d, q0, q1, t;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p0) : "v"(a0), "v"(b0));      // op_sel_hi:[0,1]: hi = src0.lo * src1.hi
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(p1) : "v"(a0), "v"(b1));
    m16 = a1; m17 = p0;                                                        // v_pk_mov op_sel:[1,0]: {src0.hi, src1.lo}
    asm volatile("v_add_f32 %0, %1, %1" : "=v"(d) : "v"(p1));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q0) : "v"(m16), "v"(p0));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(q1) : "v"(m17), "v"(p1));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(t) : "v"(p0), "v"(d));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(t) : "v"(t), "v"(q0));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(t) : "v"(t), "v"(q1));
    return t;
}

// pattern 3: WRITE after write.  A packed op writes v[14:15]; DIST single instructions later an un-packed v_mov writes v14 / v15 again
// (the compiler's code reuses the packed chain's temporaries as the bf16 fragment registers of the next split).  The final value must be the
// v_mov's.  Neighbour waves (odd wave ids) keep the matrix pipe busy.
template <int DIST>
__device__ __forceinline__ void waw(float a0, float a1, float b0, float b1, float x0, float x1, float &r0, float &r1)
{
    if constexpr (DIST == 0)
        asm volatile("v_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\tv_mov_b32 v12, %4\n\tv_mov_b32 v13, %5\n\t"
                     "v_pk_mul_f32 v[14:15], v[10:11], v[12:13]\n\t"
                     "v_pk_add_f32 v[14:15], v[14:15], v[12:13] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "v_mov_b32 v14, %6\n\tv_mov_b32 v15, %7\n\t"
                     "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                     "v_mov_b32 %0, v14\n\tv_mov_b32 %1, v15\n\t"
                     : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(x0), "v"(x1) : "v10", "v11", "v12", "v13", "v14", "v15");
    else
        asm volatile("v_mov_b32 v10, %2\n\tv_mov_b32 v11, %3\n\tv_mov_b32 v12, %4\n\tv_mov_b32 v13, %5\n\t"
                     "v_pk_mul_f32 v[14:15], v[10:11], v[12:13]\n\t"
                     "v_pk_add_f32 v[14:15], v[14:15], v[12:13] op_sel:[0,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]\n\t"
                     "v_add_f32 v16, v10, v11\n\tv_add_f32 v17, v12, v13\n\tv_mul_f32 v16, v16, v17\n\tv_add_f32 v17, v16, v10\n\t"
                     "v_cvt_pk_bf16_f32 v14, %6, %7\n\tv_cvt_pk_bf16_f32 v15, %7, %6\n\t"
                     "s_nop 7\n\ts_nop 7\n\ts_nop 7\n\ts_nop 7\n\t"
                     "v_mov_b32 %0, v14\n\tv_mov_b32 %1, v15\n\t"
                     : "=v"(r0), "=v"(r1) : "v"(a0), "v"(a1), "v"(b0), "v"(b1), "v"(x0), "v"(x1) : "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17");
}

template <int DIST>
__global__ void __launch_bounds__(512) k3(unsigned long long *bad, int iters, int neighbour)
{
    const int wave = threadIdx.x >> 6;
    unsigned long long mism = 0;
    if (neighbour && (wave & 1)) {
        f32x4 c = {0, 0, 0, 0};
        bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int u = 0; u < 8; ++u) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
        }
        if (c[0] == 123.456f) bad[1] = 1;
        return;
    }
    float x = 0.37f + 1e-3f * threadIdx.x + 1e-5f * blockIdx.x, y = 0.91f - 7e-4f * threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        float r0, r1;
        const float x0 = x * 3.0f + 1.0f, x1 = y * 5.0f - 2.0f;
        waw<DIST>(x, y, y * 0.5f + 0.1f, x * 0.25f + 0.3f, x0, x1, r0, r1);
        unsigned e0 = __float_as_uint(x0), e1 = __float_as_uint(x1);
        if (DIST) {       // expected: the two v_cvt_pk_bf16_f32 results
            const unsigned h0 = __float_as_uint((float)(__bf16)x0) >> 16, h1 = __float_as_uint((float)(__bf16)x1) >> 16;
            e0 = h0 | (h1 << 16); e1 = h1 | (h0 << 16);
        }
        if (__float_as_uint(r0) != e0 || __float_as_uint(r1) != e1) ++mism;
        x = x * 0.999f + 1e-4f * (i & 7); y = y * 1.0003f - 2e-4f * (i & 3);
        if (y > 4.0f) y -= 3.0f;
    }
    if (mism) atomicAdd(bad, mism);
}

template <int GAP>
__global__ void __launch_bounds__(512) k2(unsigned long long *bad, int iters)
{
    unsigned long long mism = 0;
    f32x4 acc = {0, 0, 0, 0};
    bf16x8 fa = {1, 2, 3, 4, 5, 6, 7, 8}, fb = {8, 7, 6, 5, 4, 3, 2, 1};
    float x = 0.37f + 1e-3f * threadIdx.x + 1e-5f * blockIdx.x, y = 0.91f - 7e-4f * threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        const float a0 = x, a1 = y, b0 = y * 0.5f + 0.1f, b1 = x * 0.25f + 0.3f;
        const float got = chain2<GAP>(a0, a1, b0, b1, fa, fb, acc), ref = reference2(a0, a1, b0, b1);
        if (__float_as_uint(got) != __float_as_uint(ref)) ++mism;
        x = x * 0.999f + 1e-4f * (i & 7); y = y * 1.0003f - 2e-4f * (i & 3);
        if (y > 4.0f) y -= 3.0f;
    }
    if (acc[0] == 123.456f) bad[1] = 1;
    if (mism) atomicAdd(bad, mism);
}

template <int GAP>
__global__ void __launch_bounds__(512) k(unsigned long long *bad, int iters, int neighbour)
{
    const int wave = threadIdx.x >> 6;
    unsigned long long mism = 0;
    if (neighbour && (wave & 1)) {      // the SIMD neighbour: matrix instructions only
        f32x4 c = {0, 0, 0, 0};
        bf16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
        for (int i = 0; i < iters; ++i) {
            if (neighbour == 1) {
#pragma unroll
                for (int u = 0; u < 8; ++u) c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
            } else {
#pragma unroll
                for (int u = 0; u < 8; ++u) c = __builtin_amdgcn_mfma_f32_16x16x4f32(1.0f + u, 0.5f, c, 0, 0, 0);
            }
        }
        if (c[0] == 123.456f) bad[1] = 1;
        return;
    }
    float x = 0.37f + 1e-3f * threadIdx.x + 1e-5f * blockIdx.x, y = 0.91f - 7e-4f * threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        const float a0 = x, a1 = y, b0 = y * 0.5f + 0.1f, b1 = x * 0.25f + 0.3f;
        const float got = chain<GAP>(a0, a1, b0, b1), ref = reference(a0, a1, b0, b1);
        if (__float_as_uint(got) != __float_as_uint(ref)) ++mism;
        x = x * 0.999f + 1e-4f * (i & 7); y = y * 1.0003f - 2e-4f * (i & 3);
        if (y > 4.0f) y -= 3.0f;
    }
    if (mism) atomicAdd(bad, mism);
}

int main()
{
    unsigned long long *bad;
    (void)hipMalloc(&bad, 16);
    for (int neighbour = 0; neighbour < 3; ++neighbour)
        for (int gap = 0; gap < 2; ++gap) {
            (void)hipMemset(bad, 0, 16);
            const int iters = 200000;
            if (gap == 0) k<0><<<1024, 512>>>(bad, iters, neighbour);
            else k<1><<<1024, 512>>>(bad, iters, neighbour);
            (void)hipDeviceSynchronize();
            unsigned long long h[2];
            (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
            const double total = 1024.0 * (neighbour ? 256 : 512) * iters;
            printf("neighbour wave: %-18s gap after packed ops: %s -> %llu mismatching results of %.3g\n",
                   neighbour == 0 ? "same packed chain" : neighbour == 1 ? "bf16 MFMA loop" : "fp32 MFMA loop", gap ? "s_nop 1" : "none   ", h[0], total);
        }
    for (int gap = 0; gap < 2; ++gap) {
        (void)hipMemset(bad, 0, 16);
        const int iters = 200000;
        if (gap == 0) k2<0><<<1024, 512>>>(bad, iters);
        else k2<1><<<1024, 512>>>(bad, iters);
        (void)hipDeviceSynchronize();
        unsigned long long h[2];
        (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
        printf("pattern 2 (own bf16 MFMAs, v_pk_mov_b32)    gap after packed ops: %s -> %llu mismatching results of %.3g\n", gap ? "s_nop 1" : "none   ", h[0],
               1024.0 * 512 * iters);
    }
    for (int neighbour = 0; neighbour < 2; ++neighbour)
        for (int dist = 0; dist < 2; ++dist) {
            (void)hipMemset(bad, 0, 16);
            const int iters = 200000;
            if (dist == 0) k3<0><<<1024, 512>>>(bad, iters, neighbour);
            else k3<1><<<1024, 512>>>(bad, iters, neighbour);
            (void)hipDeviceSynchronize();
            unsigned long long h[2];
            (void)hipMemcpy(h, bad, 16, hipMemcpyDeviceToHost);
            printf("pattern 3 (write after a packed write, %s, neighbour %s) -> %llu wrong final values of %.3g\n", dist ? "v_cvt_pk_bf16_f32 4 instructions later" : "v_mov right behind",
                   neighbour ? "bf16 MFMA loop" : "same chain", h[0], 1024.0 * (neighbour ? 256 : 512) * iters);
        }
    return 0;
}
