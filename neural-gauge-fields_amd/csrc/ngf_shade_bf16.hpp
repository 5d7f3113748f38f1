// ngf_shade_bf16.hpp -- NGF_F_SPLIT_BF16 (opt-in): the colour MLP's matrix products on v_mfma_f32_16x16x32_bf16 with every fp32
// operand split into three bf16 terms (x = hi + mid + lo, 3 x 8 significand bits) and fp32 accumulation.
//
// Why: on gfx950 v_mfma_f32_16x16x4_f32 runs on the SIMD's fp32 vector datapath (64 flop/clk/SIMD) and serialises with the VALU
// (profiles/micro/mfma_valu_overlap.hip); the bf16 instruction does 16 x the work per issue (16x16x32 in ~17 cycles vs 16x16x4 in
// ~33).  With the six products of order <= 2 (hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid) the dropped terms are below 2^-24 of the
// product, i.e. the result carries fp32-level error (measured against the fp32 path in tests/test_gpu_parity.py) while a pass
// costs 168 bf16 MFMAs (~2.8 k matrix cycles) instead of 208 fp32 MFMAs (~6.8 k) plus ~290 VALU instructions for the splits.
// Not bit-identical to the fp32-MFMA path (different summation tree): behind a flag, default off.
//
//   layer 1  K = 160 per sample = 144 plane features + 16 view inputs (no per-ray fold: the view inputs fill the k-padding),
//            lane (s, kq) supplies 40 of them as five 8-element B fragments: [feat plane 0 (12) | plane 1 (12) | plane 2 (12) | view (4)]
//   layer 2  K = 64: the lane's 16 ReLU'd accumulators as two B fragments
//   layer 3  64 -> 3 on the fp32 VALU as in mlp_tail16 (48 FMAs per lane: no matrix work worth splitting)
// A fragments (weights, pre-split on the host, layer 1 pre-composed with `basis`) live in LDS as [mt][k-block][part][lane][8 bf16].
#pragma once
#include "ngf_shade16.hpp"

namespace ngf {

typedef short bf16x8 __attribute__((ext_vector_type(8)));

struct MlpLayoutBf16 {                        // floats (a bf16x8 fragment = 4 floats)
    static constexpr int KB1 = 5, KB2 = 2;
    static constexpr int W1 = 0;                              // [4 mt][5 kb][3 parts][64 lanes][4]
    static constexpr int W2 = W1 + 4 * KB1 * 3 * 64 * 4;      // [4 mt][2 kb][3 parts][64 lanes][4]
    static constexpr int B1 = W2 + 4 * KB2 * 3 * 64 * 4;      // [4 kq][16] fp32, accumulator order
    static constexpr int B2 = B1 + 64;
    static constexpr int W3 = B2 + 64;                        // [3][4 kq][16] fp32
    static constexpr int B3 = W3 + 192;
    static constexpr int TOTAL = B3 + 4;
};

struct Split8 { bf16x8 h, m, l; };

// x = hi + mid + lo with hi = bf16(x), mid = bf16(x - hi), lo = bf16(x - hi - mid); the two subtractions are exact in fp32
__device__ __forceinline__ Split8 split8(const float x[8])
{
    Split8 s;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 hb = (__bf16)x[e];
        const float r1 = x[e] - (float)hb;
        const __bf16 mb = (__bf16)r1;
        const float r2 = r1 - (float)mb;
        const __bf16 lb = (__bf16)r2;
        s.h[e] = __builtin_bit_cast(short, hb);
        s.m[e] = __builtin_bit_cast(short, mb);
        s.l[e] = __builtin_bit_cast(short, lb);
    }
    return s;
}

#define NGF_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)

// acc[mt] += W[mt][kb] . x for the four unit tiles; w: the k-block's A fragments [mt stride MT_STRIDE][3 parts][64 lanes] (this lane's).
// The three fragments of tile mt+1 are read from LDS behind the six MFMAs of tile mt; sched_barrier keeps it that way (left alone
// hipcc reads all twelve 16-byte fragments of a k-block up front: 48 more live registers and spills).
struct AFrag { bf16x8 h, m, l; };
template <int MT_STRIDE>
__device__ __forceinline__ AFrag afrag_load(const float *w, int mt)
{
    AFrag a;
    a.h = *reinterpret_cast<const bf16x8 *>(w + mt * MT_STRIDE);
    a.m = *reinterpret_cast<const bf16x8 *>(w + mt * MT_STRIDE + 64 * 4);
    a.l = *reinterpret_cast<const bf16x8 *>(w + mt * MT_STRIDE + 2 * 64 * 4);
    return a;
}
// six products for TWO unit tiles, their dependent accumulator chains interleaved (a chain of six MFMAs on one accumulator pays the
// dependent-issue latency six times; two chains hide each other)
#ifdef NGF_EXP_NOPS      // experiment build (profiles/exp_determinism_builds.sh): 32 idle issue slots around every bf16 MFMA pair -- raises the rate of
                         // the packed-math nondeterminism of the InfoInv pass (pe_octave, ngf_infoinv.hpp) from 1 in 50 000 launches to 1 in 5
#define NGF_EXP_GAP() do { __builtin_amdgcn_sched_barrier(0); asm volatile("s_nop 15\n\ts_nop 15"); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define NGF_EXP_GAP() do { } while (0)
#endif
__device__ __forceinline__ void six_products2(const AFrag &a, const AFrag &b, const Split8 &x, f32x4 &c0, f32x4 &c1)
{
    NGF_EXP_GAP();
    c0 = NGF_MFMA_BF16(a.l, x.h, c0);  c1 = NGF_MFMA_BF16(b.l, x.h, c1);          // smallest terms first
    NGF_EXP_GAP();
    c0 = NGF_MFMA_BF16(a.h, x.l, c0);  c1 = NGF_MFMA_BF16(b.h, x.l, c1);
    NGF_EXP_GAP();
    c0 = NGF_MFMA_BF16(a.m, x.m, c0);  c1 = NGF_MFMA_BF16(b.m, x.m, c1);
    NGF_EXP_GAP();
    c0 = NGF_MFMA_BF16(a.m, x.h, c0);  c1 = NGF_MFMA_BF16(b.m, x.h, c1);
    NGF_EXP_GAP();
    c0 = NGF_MFMA_BF16(a.h, x.m, c0);  c1 = NGF_MFMA_BF16(b.h, x.m, c1);
    NGF_EXP_GAP();
    c0 = NGF_MFMA_BF16(a.h, x.h, c0);  c1 = NGF_MFMA_BF16(b.h, x.h, c1);
    NGF_EXP_GAP();
}
template <int MT_STRIDE>
__device__ __forceinline__ void kblock_bf16(const float *w, const Split8 &x, f32x4 acc[4])
{
    const AFrag a0 = afrag_load<MT_STRIDE>(w, 0), a1 = afrag_load<MT_STRIDE>(w, 1);
    __builtin_amdgcn_sched_barrier(0);
    const AFrag a2 = afrag_load<MT_STRIDE>(w, 2), a3 = afrag_load<MT_STRIDE>(w, 3);       // in flight behind the first twelve MFMAs
    __builtin_amdgcn_sched_barrier(0);
    six_products2(a0, a1, x, acc[0], acc[1]);
    __builtin_amdgcn_sched_barrier(0);
    six_products2(a2, a3, x, acc[2], acc[3]);
}

// layer 3 + sigmoid of mlp_tail16 on its own (c = layer-2 pre-activations in accumulator order)
__device__ __forceinline__ void mlp_layer3_16(const float *blob, int oW3, int oB3, int lane, const f32x4 c[4], float rgb[3])
{
    const int kq = lane >> 4;
    const f32x4 *w3 = reinterpret_cast<const f32x4 *>(blob + oW3 + kq * 16);      // 16-byte LDS reads with immediate offsets (see mlp_tail16)
    float hr[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) hr[k] = relu1(c[k >> 2][k & 3]);
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float s = 0.0f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 w = w3[ch * 16 + q];
#pragma unroll
            for (int e = 0; e < 4; ++e) s = fmaf(w[e], hr[4 * q + e], s);
        }
        s = s + __shfl_xor(s, 16);
        s = s + __shfl_xor(s, 32);
        s = s + blob[oB3 + ch];
        rgb[ch] = s;              // logit (see mlp_tail16)
    }
}

// v: the lane's four view inputs (entries kq*4 .. kq*4+3 of its sample's ray)
__device__ __forceinline__ void mlp_pass16_bf16(const RenderArgs &A, const float *blob, const float rec[kRecFloats], const f32x4 v, int lane, float rgb[3])
{
    using L = MlpLayoutBf16;
    blob = per_pass16(blob);
    const int kq = lane >> 4;
    constexpr int KB_STRIDE = 3 * 64 * 4, MT1 = L::KB1 * KB_STRIDE, MT2 = L::KB2 * KB_STRIDE;
    const float *w1 = blob + L::W1 + lane * 4, *w2 = blob + L::W2 + lane * 4;
    Gather16<48> g;
    float f0[12], f1[12], f2[12];
    gather16_issue<48, 0>(A, rec, kq, g);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = *reinterpret_cast<const f32x4 *>(blob + L::B1 + kq * 16 + mt * 4);
    mix16<48>(g, f0);
    __builtin_amdgcn_sched_barrier(0);
    gather16_issue<48, 1>(A, rec, kq, g);
    __builtin_amdgcn_sched_barrier(0);
    {   // k-block 0: features 0..7 of plane 0
        const Split8 x = split8(f0);
        kblock_bf16<MT1>(w1, x, acc);
    }
    __builtin_amdgcn_sched_barrier(0);
    mix16<48>(g, f1);
    __builtin_amdgcn_sched_barrier(0);
    gather16_issue<48, 2>(A, rec, kq, g);
    __builtin_amdgcn_sched_barrier(0);
    {   // k-blocks 1, 2: plane 0 [8..11] + plane 1 [0..3], plane 1 [4..11]
        const float xa[8] = {f0[8], f0[9], f0[10], f0[11], f1[0], f1[1], f1[2], f1[3]};
        kblock_bf16<MT1>(w1 + KB_STRIDE, split8(xa), acc);
        kblock_bf16<MT1>(w1 + 2 * KB_STRIDE, split8(f1 + 4), acc);
    }
    __builtin_amdgcn_sched_barrier(0);
    mix16<48>(g, f2);
    __builtin_amdgcn_sched_barrier(0);
    {   // k-blocks 3, 4: plane 2 [0..7], plane 2 [8..11] + the four view inputs
        kblock_bf16<MT1>(w1 + 3 * KB_STRIDE, split8(f2), acc);
        const float xb[8] = {f2[8], f2[9], f2[10], f2[11], v[0], v[1], v[2], v[3]};
        kblock_bf16<MT1>(w1 + 4 * KB_STRIDE, split8(xb), acc);
    }
    __builtin_amdgcn_sched_barrier(0);
    // layer 2: the lane's 16 hidden activations (tiles 0,1 | tiles 2,3) are its B fragments
    f32x4 c[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) c[mt] = *reinterpret_cast<const f32x4 *>(blob + L::B2 + kq * 16 + mt * 4);
    {
        float h[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) h[k] = relu1(acc[k >> 2][k & 3]);
        kblock_bf16<MT2>(w2, split8(h), c);
        kblock_bf16<MT2>(w2 + KB_STRIDE, split8(h + 8), c);
    }
    __builtin_amdgcn_sched_barrier(0);
    mlp_layer3_16(blob, L::W3, L::B3, lane, c, rgb);
}

// ---- NGF_F_BAKE_COLOR | NGF_F_SPLIT_BF16 (round 5, opt-in): level 3 with layer 2 as six bf16 products per fp32 product -------------------------
// What binds the level-3 frame is its 64 fp32 matrix instructions per pass (2048 matrix-pipe cycles) and its L1 traffic, not its vector
// instructions (profiles/r05_level3_ablations.txt).  Layer 1 is folded into the planes at level 3, so layer 2 IS the matrix work:
// here its 64 x 64 x 16 product runs as 48 v_mfma_f32_16x16x32_bf16 (~17 cycles each) on 3-term split operands -- the same technique, the same
// weight-fragment image and the same order of the six products as mlp_pass16_bf16's layer 2; fp32-level error, other last bits than the fp32 path.
__device__ __forceinline__ void mlp_pass16_baked_bf16(const RenderArgs &A, const float *blob, const float rec[kRecFloats], const f32x4 v, int lane,
                                                      float rgb[3], const float *pre = nullptr, const RecCells *gcells = nullptr)
{
    using L = MlpLayout16BakedBf16;
    blob = per_pass16(blob);
    const int kq = lane >> 4;
    f32x4 acc[4];
    baked16_layer1<L>(A, blob, rec, v, lane, acc, pre, gcells);
    constexpr int KB_STRIDE = 3 * 64 * 4, MT2 = 2 * KB_STRIDE;
    const float *w2 = blob + L::W2 + lane * 4;
    f32x4 c[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) c[mt] = *reinterpret_cast<const f32x4 *>(blob + L::B2 + kq * 16 + mt * 4);
    {
        float h[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) h[k] = relu1(acc[k >> 2][k & 3]);
        kblock_bf16<MT2>(w2, split8(h), c);
        kblock_bf16<MT2>(w2 + KB_STRIDE, split8(h + 8), c);
    }
    __builtin_amdgcn_sched_barrier(0);
    mlp_layer3_16(blob, L::W3, L::B3, lane, c, rgb);
}

// ---- the same pass in the register budget of 12 waves per CU (168) ------------------------------------------------------------------
// mlp_pass16_bf16 keeps a plane's four taps (48 registers) and four tiles' fragments (48) in flight and needs ~240 registers with the
// march state around it.  Here a cell's taps arrive in two ROWS of 24 registers -- the bilinear sum is accumulated in bil_mix's order
// (w00 v00, + w10 v10, + w01 v01, + w11 v11), so the features are the same bits -- and the A fragments come two tiles at a time.
// Same k-blocks, same order of the six products per accumulator: bit-identical colours.
struct GatherRow16 { f32x4 a[2][3]; };          // one row of a cell: two taps x 12 channels
__device__ __forceinline__ void gather_row16(const float *base, GatherRow16 &g)
{
    // base: the lane's channels [4kq, 4kq+4) of the row's first tap; +16 per quad of channels, +48 for the second tap
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int q = 0; q < 3; ++q) g.a[t][q] = *reinterpret_cast<const f32x4 *>(base + t * 48 + 16 * q);
}
template <int MT_STRIDE>
__device__ __forceinline__ void kblock_bf16_pairs16(const float *w, const Split8 &x, f32x4 acc[4])
{
    {
        const AFrag a0 = afrag_load<MT_STRIDE>(w, 0), a1 = afrag_load<MT_STRIDE>(w, 1);
        six_products2(a0, a1, x, acc[0], acc[1]);
    }
    __builtin_amdgcn_sched_barrier(0);
    {
        const AFrag a2 = afrag_load<MT_STRIDE>(w, 2), a3 = afrag_load<MT_STRIDE>(w, 3);
        six_products2(a2, a3, x, acc[2], acc[3]);
    }
    __builtin_amdgcn_sched_barrier(0);
}

__device__ __forceinline__ void mlp_pass16_bf16_rows(const RenderArgs &A, const float *blob, const float rec[kRecFloats], const f32x4 v, int lane, float rgb[3])
{
    using L = MlpLayoutBf16;
    blob = per_pass16(blob);
    const int kq = lane >> 4;
    constexpr int KB_STRIDE = 3 * 64 * 4, MT1 = L::KB1 * KB_STRIDE, MT2 = L::KB2 * KB_STRIDE;
    const float *w1 = blob + L::W1 + lane * 4, *w2 = blob + L::W2 + lane * 4;
    GatherRow16 g;
    Bil b = bil_setup(rec[2], rec[3], karg_tex(offsetof(RenderArgs, app) + (0) * sizeof(Tex)));
    gather_row16(karg_tex(offsetof(RenderArgs, app) + (0) * sizeof(Tex)).p + (size_t)b.idx * 48 + 4 * kq, g);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = *reinterpret_cast<const f32x4 *>(blob + L::B1 + kq * 16 + mt * 4);
    float left[4];
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        float f[12];
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) f[4 * q + e] = fmaf(b.w10, g.a[1][q][e], b.w00 * g.a[0][q][e]);
        __builtin_amdgcn_sched_barrier(0);
        gather_row16(karg_tex(offsetof(RenderArgs, app) + (p) * sizeof(Tex)).p + ((size_t)b.idx + karg_tex(offsetof(RenderArgs, app) + (p) * sizeof(Tex)).stride) * 48 + 4 * kq, g);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) f[4 * q + e] = fmaf(b.w11, g.a[1][q][e], fmaf(b.w01, g.a[0][q][e], f[4 * q + e]));
        __builtin_amdgcn_sched_barrier(0);
        if (p < 2) {                           // the next plane's first row travels behind this plane's MFMAs
            b = bil_setup(rec[4 + 2 * p], rec[5 + 2 * p], karg_tex(offsetof(RenderArgs, app) + (p + 1) * sizeof(Tex)));
            gather_row16(karg_tex(offsetof(RenderArgs, app) + (p + 1) * sizeof(Tex)).p + (size_t)b.idx * 48 + 4 * kq, g);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (p == 0) {                          // k-block 0: plane 0 [0..7]
            kblock_bf16_pairs16<MT1>(w1, split8(f), acc);
            left[0] = f[8]; left[1] = f[9]; left[2] = f[10]; left[3] = f[11];
        } else if (p == 1) {                   // k-blocks 1, 2: plane 0 [8..11] + plane 1 [0..3], plane 1 [4..11]
            const float xa[8] = {left[0], left[1], left[2], left[3], f[0], f[1], f[2], f[3]};
            kblock_bf16_pairs16<MT1>(w1 + KB_STRIDE, split8(xa), acc);
            kblock_bf16_pairs16<MT1>(w1 + 2 * KB_STRIDE, split8(f + 4), acc);
        } else {                               // k-blocks 3, 4: plane 2 [0..7], plane 2 [8..11] + the four view inputs
            kblock_bf16_pairs16<MT1>(w1 + 3 * KB_STRIDE, split8(f), acc);
            const float xb[8] = {f[8], f[9], f[10], f[11], v[0], v[1], v[2], v[3]};
            kblock_bf16_pairs16<MT1>(w1 + 4 * KB_STRIDE, split8(xb), acc);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    f32x4 c[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) c[mt] = *reinterpret_cast<const f32x4 *>(blob + L::B2 + kq * 16 + mt * 4);
    {
        float h[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) h[k] = relu1(acc[k >> 2][k & 3]);
        kblock_bf16_pairs16<MT2>(w2, split8(h), c);
        kblock_bf16_pairs16<MT2>(w2 + KB_STRIDE, split8(h + 8), c);
    }
    __builtin_amdgcn_sched_barrier(0);
    mlp_layer3_16(blob, L::W3, L::B3, lane, c, rgb);
}

}  // namespace ngf
