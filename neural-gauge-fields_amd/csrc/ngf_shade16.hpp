// ngf_shade16.hpp -- the colour MLP on v_mfma_f32_16x16x4_f32: 16 samples per pass, FOUR lanes per sample.
//
// Why a second shade formulation: with v_mfma_f32_32x32x2_f32 (ngf_render.hpp, two lanes per sample) a lane
// carries 32 accumulators and gathers 24 (baked: 32) channels per tap, which pins the fused kernel at
// ~220-250 VGPRs = 2 waves per SIMD, too few to overlap the march (TA / latency bound) with the shade
// (matrix-pipe bound).  Here lane (s = l&15, kq = l>>4) owns 16 accumulators and gathers 12 (baked: 16)
// channels per tap: the same matrix work per sample, about half the registers per lane.
//
//   v_mfma_f32_16x16x4_f32: lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15];
//                           lane l, register r holds D[i = 4*(l>>4) + r][j = l&15].
// As in the 32-wide form the product is evaluated transposed (rows = hidden units, columns = samples): the four
// accumulator tiles of a lane are hidden units n = mt*16 + 4*kq + r of ITS sample and feed layer 2 in place.
#pragma once
#include "ngf_device.hpp"

namespace ngf {

constexpr int kBatch16 = 16;

template <int APP>
struct MlpLayout16 {                      // floats
    static constexpr int QCH = APP / 4;           // colour channels per plane per lane (12)
    static constexpr int KT = 3 * QCH + 4;        // layer-1 k-steps, 4 inputs each (40)
    static constexpr int W1 = 0;                  // [4 mt][KT][64 lanes]
    static constexpr int W2 = W1 + 4 * KT * 64;   // [4 mt][16][64 lanes]
    static constexpr int B1 = W2 + 4 * 16 * 64;   // [4 kq][16]
    static constexpr int B2 = B1 + 64;
    static constexpr int W3 = B2 + 64;            // [3][4 kq][16]
    static constexpr int B3 = W3 + 192;
    static constexpr int TOTAL = B3 + 4;
};

struct MlpLayout16Baked {                 // NGF_F_BAKE_COLOR: only the view-input k-steps of layer 1 remain
    static constexpr int W1V = 0;                 // [4 mt][4][64 lanes]
    static constexpr int W2 = W1V + 4 * 4 * 64;
    static constexpr int B1 = W2 + 4 * 16 * 64;
    static constexpr int B2 = B1 + 64;
    static constexpr int W3 = B2 + 64;
    static constexpr int B3 = W3 + 192;
    static constexpr int TOTAL = B3 + 4;
};

__device__ __forceinline__ const float *per_pass16(const float *blob)
{
    int z = 0;
    asm volatile("" : "+v"(z));       // keeps the read-only LDS image from being hoisted out of the persistent loops
    return blob + z;
}

#define NGF_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// layers 2 and 3; acc[mt][r] = layer-1 pre-activation of hidden unit mt*16 + 4*kq + r
__device__ __forceinline__ void mlp_tail16(const float *blob, int oW2, int oB2, int oW3, int oB3, int lane, const f32x4 acc[4],
                                           float rgb[3])
{
    const int kq = lane >> 4;
    f32x4 c[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) c[mt] = *reinterpret_cast<const f32x4 *>(blob + oB2 + kq * 16 + mt * 4);
    const float *w2 = blob + oW2 + lane;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const float h = fmaxf(acc[t >> 2][t & 3], 0.0f);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) c[mt] = NGF_MFMA16(w2[(mt * 16 + t) * 64], h, c[mt]);
    }
    const float *w3 = blob + oW3 + kq * 16;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s = fmaf(w3[ch * 64 + k], fmaxf(c[k >> 2][k & 3], 0.0f), s);
        s = s + __shfl_xor(s, 16);
        s = s + __shfl_xor(s, 32);
        s = s + blob[oB3 + ch];
        rgb[ch] = 1.0f / (1.0f + expf(-s));
    }
}

// ---- faithful layer 1 (pre-composed with basis): 12 channels per tap per lane -----------------------------------
template <int APP>
struct Gather16 {                 // one plane: 4 taps x (APP/16) float4
    f32x4 raw[4][APP / 16];
    Bil b;
};

template <int APP, int P>
__device__ __forceinline__ void gather16_issue(const RenderArgs &A, const float rec[kRecFloats], int kq, Gather16<APP> &g)
{
    constexpr int NQ = APP / 16;
    const Tex &t = A.app[P];
    g.b = bil_setup(rec[2 + 2 * P], rec[3 + 2 * P], t);
    const f32x4 *t00 = reinterpret_cast<const f32x4 *>(t.p + (size_t)g.b.idx * APP + kq * (APP / 4));
    const f32x4 *t01 = t00 + (size_t)t.stride * (APP / 4);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        g.raw[0][q] = t00[q];
        g.raw[1][q] = t00[APP / 4 + q];
        g.raw[2][q] = t01[q];
        g.raw[3][q] = t01[APP / 4 + q];
    }
}

template <int APP, int P>
__device__ __forceinline__ void layer1_plane16(const float *blob, int lane, const Gather16<APP> &g, f32x4 acc[4])
{
    using L = MlpLayout16<APP>;
    constexpr int NQ = APP / 16;
    const float *w1 = blob + L::W1 + lane;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float f = bil_mix(g.b, g.raw[0][q][e], g.raw[1][q][e], g.raw[2][q][e], g.raw[3][q][e]);
            const int t = P * L::QCH + 4 * q + e;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = NGF_MFMA16(w1[(mt * L::KT + t) * 64], f, acc[mt]);
        }
}

template <int APP>
__device__ __forceinline__ void mlp_pass16(const RenderArgs &A, const float *blob, const float rec[kRecFloats], const float *vf,
                                           int lane, float rgb[3])
{
    using L = MlpLayout16<APP>;
    blob = per_pass16(blob);
    const int kq = lane >> 4;
    Gather16<APP> ga, gb;
    gather16_issue<APP, 0>(A, rec, kq, ga);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = *reinterpret_cast<const f32x4 *>(blob + L::B1 + kq * 16 + mt * 4);
    {   // view-direction inputs: lane-quarter kq supplies entries kq*4 .. kq*4+3
        const f32x4 v = *reinterpret_cast<const f32x4 *>(vf + kq * 4);
        const float *w1 = blob + L::W1 + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = NGF_MFMA16(w1[(mt * L::KT + 3 * L::QCH + j) * 64], v[j], acc[mt]);
    }
    __builtin_amdgcn_sched_barrier(0);
    gather16_issue<APP, 1>(A, rec, kq, gb);            // plane 1 in flight behind plane 0's MFMAs
    __builtin_amdgcn_sched_barrier(0);
    layer1_plane16<APP, 0>(blob, lane, ga, acc);
    __builtin_amdgcn_sched_barrier(0);
    gather16_issue<APP, 2>(A, rec, kq, ga);
    __builtin_amdgcn_sched_barrier(0);
    layer1_plane16<APP, 1>(blob, lane, gb, acc);
    __builtin_amdgcn_sched_barrier(0);
    layer1_plane16<APP, 2>(blob, lane, ga, acc);
    mlp_tail16(blob, L::W2, L::B2, L::W3, L::B3, lane, acc, rgb);
}

// ---- NGF_F_BAKE_COLOR: 64-channel layer-1 pre-activation planes, channel kq*16 + mt*4 + r ----------------------
template <int P>
__device__ __forceinline__ void baked16_issue(const RenderArgs &A, const float rec[kRecFloats], int kq, Gather16<64> &g)
{
    const Tex &t = A.app[P];
    g.b = bil_setup(rec[2 + 2 * P], rec[3 + 2 * P], t);
    const f32x4 *t00 = reinterpret_cast<const f32x4 *>(t.p + (size_t)g.b.idx * 64 + kq * 16);
    const f32x4 *t01 = t00 + (size_t)t.stride * 16;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        g.raw[0][q] = t00[q];
        g.raw[1][q] = t00[16 + q];
        g.raw[2][q] = t01[q];
        g.raw[3][q] = t01[16 + q];
    }
}

__device__ __forceinline__ void baked16_consume(const Gather16<64> &g, f32x4 acc[4])
{
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mt][e] += bil_mix(g.b, g.raw[0][mt][e], g.raw[1][mt][e], g.raw[2][mt][e], g.raw[3][mt][e]);
}

__device__ __forceinline__ void mlp_pass16_baked(const RenderArgs &A, const float *blob, const float rec[kRecFloats], const float *vf,
                                                 int lane, float rgb[3])
{
    using L = MlpLayout16Baked;
    blob = per_pass16(blob);
    const int kq = lane >> 4;
    Gather16<64> ga, gb;
    baked16_issue<0>(A, rec, kq, ga);
    __builtin_amdgcn_sched_barrier(0);
    baked16_issue<1>(A, rec, kq, gb);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = *reinterpret_cast<const f32x4 *>(blob + L::B1 + kq * 16 + mt * 4);
    {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(vf + kq * 4);
        const float *w1 = blob + L::W1V + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = NGF_MFMA16(w1[(mt * 4 + j) * 64], v[j], acc[mt]);
    }
    __builtin_amdgcn_sched_barrier(0);
    baked16_consume(ga, acc);
    __builtin_amdgcn_sched_barrier(0);
    baked16_issue<2>(A, rec, kq, ga);
    __builtin_amdgcn_sched_barrier(0);
    baked16_consume(gb, acc);
    __builtin_amdgcn_sched_barrier(0);
    baked16_consume(ga, acc);
    mlp_tail16(blob, L::W2, L::B2, L::W3, L::B3, lane, acc, rgb);
}

}  // namespace ngf
