// ngf_shade16.hpp -- the colour MLP on v_mfma_f32_16x16x4_f32: 16 samples per pass, FOUR lanes per sample.
//
// Why this formulation: with v_mfma_f32_32x32x2_f32 (round 1's first shade, two lanes per sample; removed in round 2 when InfoInv
// moved to the 16-sample form too) a lane carries 32 accumulators and gathers 24 (baked: 32) channels per tap, which pins the fused kernel at
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
// floats between the per-ray vectors of a tile's view-fold table (64 values each).  68, not 64: the lanes of a pass read the vector of THEIR sample's
// ray, and with a stride of 64 floats every ray's vector starts in the same LDS bank -- eight rays, eight-way conflicts on each of the pass's reads
// (19.6 % of the LDS-active cycles of the level-3 launch were bank conflicts, profiles/r04_triplane_R1_bdc_pmc.txt); a multiple of 4 keeps the
// 16-byte reads of levels 1-2 aligned.
constexpr int kFoldStride = 68;

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

// NGF_F_BAKE_COLOR | NGF_F_SPLIT_BF16 (round 5, opt-in): level 3 with LAYER 2 on the bf16 matrix pipe as 3-term split products (ngf_shade_bf16.hpp
// mlp_pass16_baked_bf16).  Same front part as MlpLayout16Baked (the view-input k-steps at offset 0), then W2 as bf16 A fragments
// [4 mt][2 k-blocks][3 parts][64 lanes][8 bf16] (= 4 floats per fragment), then the fp32 tables.
struct MlpLayout16BakedBf16 {
    static constexpr int W1V = 0;                 // [4 mt][4][64 lanes] fp32
    static constexpr int W2 = W1V + 4 * 4 * 64;   // bf16 fragments: 4 x 2 x 3 x 64 x 4 floats
    static constexpr int B1 = W2 + 4 * 2 * 3 * 64 * 4;
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

// the 4 view inputs lane-quarter kq supplies: entries kq*4 .. kq*4+3 of [d(3), sin(d_x, 2d_x, d_y, 2d_y, d_z, 2d_z), cos(same), 0]
__device__ __forceinline__ f32x4 view_entries16(const float d[3], int kq)
{
    f32x4 v;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int f = 4 * kq + e;                  // 0..15
        const int g = f < 3 ? 0 : (f < 9 ? f - 3 : (f < 15 ? f - 9 : 0));     // index into [d_x, 2d_x, d_y, 2d_y, d_z, 2d_z]
        const int dim = g >> 1;
        const float dd = dim == 0 ? d[0] : (dim == 1 ? d[1] : d[2]);
        float s, c;
        sincos_small((g & 1) ? dd * 2.0f : dd, s, c);
        const float raw = f == 0 ? d[0] : (f == 1 ? d[1] : d[2]);
        v[e] = f < 3 ? raw : (f < 9 ? s : (f < 15 ? c : 0.0f));
    }
    return v;
}

#define NGF_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// ReLU in ONE instruction.  fmaxf(x, 0.0f) compiles to v_max_f32 x, x, x (quieting a possible signalling NaN) + v_max_f32 0, x -- and
// LLVM folds v_med3_f32(x, 0, inf) back into the same pair.  The sign-magnitude order of IEEE floats makes the INTEGER maximum of the
// bit pattern with 0 the same function for every non-NaN x (negative floats, -0 included, are negative integers -> +0): one
// v_max_i32, bit-identical results, 32 instructions fewer per pass.
__device__ __forceinline__ float relu1(float x) { return __int_as_float(max(__float_as_int(x), 0)); }

// sigmoid (networks.py:31) with v_rcp_f32 (<= 1 ulp) instead of the IEEE division sequence (v_div_scale x2, v_rcp, 4 FMAs, v_div_fmas,
// v_div_fixup: ~10 instructions per channel): the colour moves by <= 1 ulp of a value in (0, 1)
__device__ __forceinline__ float sigmoid_rcp(float s) { return __builtin_amdgcn_rcpf(1.0f + expf(-s)); }

// layers 2 and 3; acc[mt][r] = layer-1 pre-activation of hidden unit mt*16 + 4*kq + r.  `rgb` receives the three pre-sigmoid sums (identical in
// the four lanes of a sample): every shade function of the library returns LOGITS since round 3 -- the fused kernel applies the sigmoid to ONE
// channel per lane (lane quarter kq takes channel kq and writes its own entry of the result list) instead of three in all 64 lanes.
__device__ __forceinline__ void mlp_tail16(const float *blob, int oW2, int oB2, int oW3, int oB3, int lane, const f32x4 acc[4],
                                           float rgb[3], unsigned long long *tk = nullptr)
{
    const int kq = lane >> 4;
    f32x4 c[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) c[mt] = *reinterpret_cast<const f32x4 *>(blob + oB2 + kq * 16 + mt * 4);
    const float *w2 = blob + oW2 + lane;
#ifndef NGF_EXP_RELU_INTERLEAVED
    // all sixteen ReLUs first, then 64 matrix instructions in a row: hipcc's own order -- four MFMAs, the next input's v_max_i32, four MFMAs ... --
    // switches the SIMD between its matrix and its vector pipe sixteen times per pass, and a switch costs the wave ~38 cycles
    // (profiles/r02_micro_mfma_valu_overlap.txt: 32 x (MFMA + 2 FMA) take 71 cycles per MFMA instead of 33)
    float h[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) h[t] = relu1(acc[t >> 2][t & 3]);
    __builtin_amdgcn_sched_barrier(0);
#ifdef NGF_EXP_NO_LAYER2      // TIMING EXPERIMENT (wrong pixels): a sixteenth of layer 2's matrix instructions -- what the matrix pipe's share of the pass costs
#pragma unroll
    for (int t = 0; t < 16; ++t) c[t & 3] = NGF_MFMA16(w2[((t & 3) * 16 + t) * 64], h[t], c[t & 3]);
#else
#pragma unroll
    for (int t = 0; t < 16; ++t)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) c[mt] = NGF_MFMA16(w2[(mt * 16 + t) * 64], h[t], c[mt]);
#endif
    __builtin_amdgcn_sched_barrier(0);
#else
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        const float h = relu1(acc[t >> 2][t & 3]);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) c[mt] = NGF_MFMA16(w2[(mt * 16 + t) * 64], h, c[mt]);
    }
#endif
    if (tk) { __builtin_amdgcn_sched_barrier(0); tk[4] = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0); }
    // layer-3 weights as twelve 16-byte LDS reads with immediate offsets (left to itself hipcc pairs the scalar reads into ds_read2_b32, whose 8-bit
    // offsets do not reach the image's offset: one v_add_u32 per pair, 24 vector instructions per pass)
    const f32x4 *w3 = reinterpret_cast<const f32x4 *>(blob + oW3 + kq * 16);
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
        rgb[ch] = s;              // the LOGIT: the caller applies sigmoid_rcp (the render kernel to one channel per lane quarter)
    }
}

// ---- faithful layer 1 (pre-composed with basis): 12 channels per tap per lane -----------------------------------
template <int APP>
struct Gather16 {                 // one plane: 4 taps x (APP/16) float4
    f32x4 raw[4][APP / 16];
    Bil b;
};

// PRE: the cell comes ready-made (bil_from_rec: the march computed it for the density fetch at the same coordinates on a plane of the same size)
template <int APP, int P>
__device__ __forceinline__ void gather16_issue(const RenderArgs &A, const float rec[kRecFloats], int kq, Gather16<APP> &g, const RecCells *pre = nullptr)
{
    constexpr int NQ = APP / 16;
    const Tex t = karg_tex(offsetof(RenderArgs, app) + P * sizeof(Tex));
    if (pre) g.b = bil_from_rec(pre->idx[P], pre->wx1[P], pre->wy1[P], (pre->bits >> (8 + P)) & 1);
    else g.b = bil_setup(rec[2 + 2 * P], rec[3 + 2 * P], t);
    // lane-quarter kq owns channels [16q + 4kq, 16q + 4kq + 4), q = 0..NQ-1: the four lanes of a sample read one
    // contiguous 64-byte piece per load instruction, so a wave-wide load touches 16 cache lines instead of ~32
    const f32x4 *t00 = tex_at<f32x4>(t.p, (uint32_t)g.b.idx * APP + 4u * kq);
    const f32x4 *t01 = tex_at<f32x4>(t.p, (uint32_t)(g.b.idx + t.stride) * APP + 4u * kq);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        g.raw[0][q] = t00[4 * q];
        g.raw[1][q] = t00[APP / 4 + 4 * q];
        g.raw[2][q] = t01[4 * q];
        g.raw[3][q] = t01[APP / 4 + 4 * q];
    }
}

template <int APP>
__device__ __forceinline__ void mix16(const Gather16<APP> &g, float feat[APP / 4])
{
#pragma unroll
    for (int q = 0; q < APP / 16; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) feat[4 * q + e] = bil_mix(g.b, g.raw[0][q][e], g.raw[1][q][e], g.raw[2][q][e], g.raw[3][q][e]);
}

template <int APP, int P>
__device__ __forceinline__ void layer1_plane16(const float *blob, int lane, const float feat[APP / 4], f32x4 acc[4])
{
    using L = MlpLayout16<APP>;
    const float *w1 = blob + L::W1 + lane;
#pragma unroll
    for (int j = 0; j < L::QCH; ++j) {
        const int t = P * L::QCH + j;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = NGF_MFMA16(w1[(mt * L::KT + t) * 64], feat[j], acc[mt]);
    }
}

// One gather buffer (48 VGPRs) + 12 interpolated features: plane p is interpolated into feat[], the buffer is
// re-issued for plane p+1 at once, and plane p's 48 MFMAs run while that gather is in flight.
#define NGF_TICK(i) do { if (tk) tk[i] = __builtin_readcyclecounter(); } while (0)

// Fold (ii) of SURVEY section 7: the view-direction part of layer 1 is constant per ray.  For small tiles the kernel evaluates
// b1 + W1[:, view] . view ONCE per ray per tile (view_fold16: the same 16 MFMAs a pass would spend on it) into an LDS table in
// accumulator order; a pass then starts from its sample's ray vector `pre` (4 broadcast ds_read_b128) instead of the bias and
// skips those 16 MFMAs and their operand reads.  Same products, same order: bit-identical.
template <int OW1V, int OB1, int KSTRIDE>     // OW1V: first view k-step; KSTRIDE: k-steps per unit tile in the image
__device__ __forceinline__ void view_fold16_regs(const float *blob, const f32x4 v, float *pre, int n_rays, int lane)
{
    const int kq = lane >> 4, s = lane & 15;
    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = *reinterpret_cast<const f32x4 *>(blob + OB1 + kq * 16 + mt * 4);
    const float *w1 = blob + OW1V * 64 + lane;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = NGF_MFMA16(w1[(mt * KSTRIDE + j) * 64], v[j], acc[mt]);
    if (s < n_rays) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<f32x4 *>(pre + s * kFoldStride + kq * 16 + mt * 4) = acc[mt];
    }
}

// the same with the view inputs of ray s read from the tile's LDS table (lane (s, kq) takes entries kq*4 .. kq*4+3)
template <int OW1V, int OB1, int KSTRIDE>
__device__ __forceinline__ void view_fold16(const float *blob, const float *vfeat, float *pre, int n_rays, int lane)
{
    const int kq = lane >> 4, s = lane & 15;
    const f32x4 v = *reinterpret_cast<const f32x4 *>(vfeat + (s < n_rays ? s : 0) * kViewFeat + kq * 4);
    view_fold16_regs<OW1V, OB1, KSTRIDE>(blob, v, pre, n_rays, lane);
}

template <int APP>
__device__ __forceinline__ void mlp_pass16(const RenderArgs &A, const float *blob, const float rec[kRecFloats], const f32x4 v,
                                           int lane, float rgb[3], unsigned long long *tk = nullptr, const float *pre = nullptr, const RecCells *cells = nullptr)
{
    using L = MlpLayout16<APP>;
    blob = per_pass16(blob);
    const int kq = lane >> 4;
    Gather16<APP> g;
    float feat[L::QCH];
    gather16_issue<APP, 0>(A, rec, kq, g, cells);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[4];
    if (pre) {      // per-ray b1 + W1[:, view] . view from the tile's table
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = *reinterpret_cast<const f32x4 *>(pre + kq * 16 + mt * 4);
    } else {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = *reinterpret_cast<const f32x4 *>(blob + L::B1 + kq * 16 + mt * 4);
        // view-direction inputs: lane-quarter kq supplies entries kq*4 .. kq*4+3 (v)
        const float *w1 = blob + L::W1 + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = NGF_MFMA16(w1[(mt * L::KT + 3 * L::QCH + j) * 64], v[j], acc[mt]);
    }
    __builtin_amdgcn_sched_barrier(0);
    NGF_TICK(0);                       // gather 0 issued + view MFMAs issued
    mix16<APP>(g, feat);
    __builtin_amdgcn_sched_barrier(0);
    NGF_TICK(1);                       // plane 0 arrived and interpolated
    gather16_issue<APP, 1>(A, rec, kq, g, cells);
    __builtin_amdgcn_sched_barrier(0);
    layer1_plane16<APP, 0>(blob, lane, feat, acc);
    __builtin_amdgcn_sched_barrier(0);
    mix16<APP>(g, feat);
    __builtin_amdgcn_sched_barrier(0);
    gather16_issue<APP, 2>(A, rec, kq, g, cells);
    __builtin_amdgcn_sched_barrier(0);
    layer1_plane16<APP, 1>(blob, lane, feat, acc);
    __builtin_amdgcn_sched_barrier(0);
    mix16<APP>(g, feat);
    __builtin_amdgcn_sched_barrier(0);
    layer1_plane16<APP, 2>(blob, lane, feat, acc);
    __builtin_amdgcn_sched_barrier(0);
    NGF_TICK(2);                       // all layer-1 MFMAs issued
    mlp_tail16(blob, L::W2, L::B2, L::W3, L::B3, lane, acc, rgb, tk);
    __builtin_amdgcn_sched_barrier(0);
    NGF_TICK(3);                       // layers 2-3 done
}


// ---- NGF_F_NO_FOLD: level 0 of SURVEY section 7 -- rgb_decoder exactly as written (networks.py:25-30) -----------------------------
//   g = basis(feat)                144 x 144, no bias / activation: its own MFMA stage (324 MFMAs per pass)
//   in = [g, view, PE(view)]       the view inputs enter per sample (no per-ray fold)
//   Linear(159 -> 64) + ReLU ...   layer 1 on the un-composed W1 (160 MFMAs), then layers 2 and 3 as usual
// The basis matrix (83 KB) does not fit LDS next to the layers: it is streamed from L2 as [k-step][3 tile groups][lane][4] (one
// 16-byte load feeds four MFMAs).  The 144 outputs land in accumulator order (unit mt*16 + 4*kq + r of the lane's own sample),
// which is the B operand order of the next layer, so layer 1 consumes them in place.
struct MlpLayout16NoFold {                 // LDS image (floats): layer 1 on [g(144) | view(16)] in accumulator order, then as MlpLayout16
    static constexpr int KT = 40;
    static constexpr int W1 = 0;                  // [4 mt][40][64]
    static constexpr int W2 = W1 + 4 * KT * 64;
    static constexpr int B1 = W2 + 4 * 16 * 64;
    static constexpr int B2 = B1 + 64;
    static constexpr int W3 = B2 + 64;
    static constexpr int B3 = W3 + 192;
    static constexpr int TOTAL = B3 + 4;
};
constexpr int kBasisPackFloats = 36 * 3 * 64 * 4;      // [36 k-steps][3 groups of 4 unit tiles (9 used)][64 lanes][4]

struct BasisPair { f32x4 a[2][3]; };              // A operands of two k-steps (9 unit tiles in 3 groups of 4)

template <int T0>
__device__ __forceinline__ void basis_load(const float *__restrict__ bpack, int lane, BasisPair &w)
{
    const f32x4 *wp = reinterpret_cast<const f32x4 *>(bpack) + (size_t)T0 * 3 * 64 + lane;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int gq = 0; gq < 3; ++gq) w.a[u][gq] = wp[(u * 3 + gq) * 64];
}

__device__ __forceinline__ void basis_mma(const BasisPair &w, float f0, float f1, f32x4 g[9])
{
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const float f = u ? f1 : f0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            g[e] = NGF_MFMA16(w.a[u][0][e], f, g[e]);
            g[4 + e] = NGF_MFMA16(w.a[u][1][e], f, g[4 + e]);
        }
        g[8] = NGF_MFMA16(w.a[u][2][0], f, g[8]);
    }
}

// the 12 k-steps of plane P: two k-steps of weights are in flight behind the 18 MFMAs of the previous two (sched_barrier pins
// the order; left alone hipcc hoists all 36 sixteen-byte loads of a plane and spills)
template <int P>
__device__ __forceinline__ void basis_plane16(const float *__restrict__ bpack, int lane, const float feat[12], f32x4 g[9])
{
    BasisPair w0, w1;
    basis_load<P * 12>(bpack, lane, w0);
    __builtin_amdgcn_sched_barrier(0);
    basis_load<P * 12 + 2>(bpack, lane, w1);
    __builtin_amdgcn_sched_barrier(0);
    basis_mma(w0, feat[0], feat[1], g);
    __builtin_amdgcn_sched_barrier(0);
    basis_load<P * 12 + 4>(bpack, lane, w0);
    __builtin_amdgcn_sched_barrier(0);
    basis_mma(w1, feat[2], feat[3], g);
    __builtin_amdgcn_sched_barrier(0);
    basis_load<P * 12 + 6>(bpack, lane, w1);
    __builtin_amdgcn_sched_barrier(0);
    basis_mma(w0, feat[4], feat[5], g);
    __builtin_amdgcn_sched_barrier(0);
    basis_load<P * 12 + 8>(bpack, lane, w0);
    __builtin_amdgcn_sched_barrier(0);
    basis_mma(w1, feat[6], feat[7], g);
    __builtin_amdgcn_sched_barrier(0);
    basis_load<P * 12 + 10>(bpack, lane, w1);
    __builtin_amdgcn_sched_barrier(0);
    basis_mma(w0, feat[8], feat[9], g);
    __builtin_amdgcn_sched_barrier(0);
    basis_mma(w1, feat[10], feat[11], g);
}

__device__ __forceinline__ void mlp_pass16_nofold(const RenderArgs &A, const float *blob, const float rec[kRecFloats], const f32x4 v, int lane,
                                                  float rgb[3])
{
    using L = MlpLayout16NoFold;
    blob = per_pass16(blob);
    const int kq = lane >> 4;
    Gather16<48> gth;
    float feat[12];
    f32x4 g[9];
#pragma unroll
    for (int mt = 0; mt < 9; ++mt) g[mt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    gather16_issue<48, 0>(A, rec, kq, gth);
    __builtin_amdgcn_sched_barrier(0);
    mix16<48>(gth, feat);
    __builtin_amdgcn_sched_barrier(0);
    gather16_issue<48, 1>(A, rec, kq, gth);
    __builtin_amdgcn_sched_barrier(0);
    basis_plane16<0>(A.basis_pack, lane, feat, g);
    __builtin_amdgcn_sched_barrier(0);
    mix16<48>(gth, feat);
    __builtin_amdgcn_sched_barrier(0);
    gather16_issue<48, 2>(A, rec, kq, gth);
    __builtin_amdgcn_sched_barrier(0);
    basis_plane16<1>(A.basis_pack, lane, feat, g);
    __builtin_amdgcn_sched_barrier(0);
    mix16<48>(gth, feat);
    __builtin_amdgcn_sched_barrier(0);
    basis_plane16<2>(A.basis_pack, lane, feat, g);
    __builtin_amdgcn_sched_barrier(0);
    // layer 1: k-step (mt_in, r) takes g[mt_in][r]; the last four k-steps take the lane's four view inputs
    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = *reinterpret_cast<const f32x4 *>(blob + L::B1 + kq * 16 + mt * 4);
    const float *w1 = blob + L::W1 + lane;
#pragma unroll
    for (int t = 0; t < 40; ++t) {
        const float x = t < 36 ? g[t >> 2][t & 3] : v[t & 3];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = NGF_MFMA16(w1[(mt * L::KT + t) * 64], x, acc[mt]);
        if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);      // keeps the LDS operand reads next to their MFMAs (register pressure)
    }
    mlp_tail16(blob, L::W2, L::B2, L::W3, L::B3, lane, acc, rgb);
}

// ---- NGF_F_BAKE_COLOR: 64-channel layer-1 pre-activation planes, channel = hidden unit mt*16 + 4*kq + r ----------------------
struct BakedHalf {                // half a plane: 4 taps x 2 float4 = accumulator tiles 2h, 2h+1
    f32x4 raw[4][2];
    Bil b;
    f32x2 wa, wb;                 // {w00, w10}, {w01, w11} as register pairs (cells from a queue record: bil_from_rec_pk), else made from b
};

template <int ST>                 // stage = plane * 2 + half; cells: the sample's bilinear cells from a 12-float queue record (else from its coordinates)
__device__ __forceinline__ void baked16_issue(const RenderArgs &A, const float rec[kRecFloats], int kq, BakedHalf &g, const RecCells *cells = nullptr)
{
    constexpr int P = ST >> 1, H = ST & 1;
    const Tex t = karg_tex(offsetof(RenderArgs, app) + P * sizeof(Tex));
    if (cells) {
        const BilPk bp = bil_from_rec_pk(cells->idx[P], cells->wx1[P], cells->wy1[P], (cells->bits >> (8 + P)) & 1);
        g.b.idx = bp.idx; g.wa = bp.wa; g.wb = bp.wb;
    } else {
        g.b = bil_setup(rec[2 + 2 * P], rec[3 + 2 * P], t);
        g.wa = f32x2{g.b.w00, g.b.w10}; g.wb = f32x2{g.b.w01, g.b.w11};
    }
    // channel 16*mt + 4*kq + r = hidden unit of accumulator (mt, r): natural order; 4 lanes read 64 contiguous bytes
    const f32x4 *t00 = tex_at<f32x4>(t.p, (uint32_t)g.b.idx * 64u + 4u * (kq + 8 * H));
    const f32x4 *t01 = tex_at<f32x4>(t.p, (uint32_t)(g.b.idx + t.stride) * 64u + 4u * (kq + 8 * H));
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        g.raw[0][q] = t00[4 * q];
#ifdef NGF_EXP_L3_ONE_TAP      // TIMING EXPERIMENT (wrong pixels): a quarter of the colour-plane load instructions -- what the L1 / texture-addresser traffic of the gather costs
        g.raw[1][q] = g.raw[0][q]; g.raw[2][q] = g.raw[0][q]; g.raw[3][q] = g.raw[0][q];
        (void)t01;
#else
        g.raw[1][q] = t00[16 + 4 * q];
        g.raw[2][q] = t01[4 * q];
        g.raw[3][q] = t01[16 + 4 * q];
#endif
    }
}

// The interpolation: four FMAs onto the running sum per channel (bil_mix + add would be five; the sums are compared with the oracle at 1e-5, not bit
// for bit).  -DNGF_PK_MIX (round 5, measured, NOT the default): the 16 channels of a lane as eight register pairs and one v_pk_fma_f32 per (tap, pair) --
// the same fused multiply-add per element (bit-identical sums), 96 instead of 192 vector instructions per pass, 102 fewer in the pass's gather block
// (274 -> 172) -- and the frame takes exactly as long: 4.345 ms either way (profiles/r05_level3_ablations.txt).  A wave64 v_fma_f32 costs the SIMD ~2.9
// cycles, a v_pk_fma_f32 ~4.8 (profiles/r05_micro_valu_cost.txt), and beside other waves' MFMAs a packed instruction costs more than its two halves
// (MI355X_MICROARCH.md); the pass is bound by its matrix instructions and its L1 traffic, not by its vector instruction count.
template <int ST>
__device__ __forceinline__ void baked16_consume(const BakedHalf &g, f32x2 sum[8])
{
    constexpr int H = ST & 1;
#ifndef NGF_PK_MIX
#pragma unroll
    for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = (2 * H + q) * 4 + e;
            const float s0 = ST < 2 ? 0.0f : sum[k >> 1][k & 1];
            sum[k >> 1][k & 1] = fmaf(g.wb[1], g.raw[3][q][e], fmaf(g.wb[0], g.raw[2][q][e], fmaf(g.wa[1], g.raw[1][q][e], fmaf(g.wa[0], g.raw[0][q][e], s0))));
        }
#else
#pragma unroll
    for (int tap = 0; tap < 4; ++tap)
#pragma unroll
        for (int q = 0; q < 2; ++q)
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
                const int k = ((2 * H + q) * 4 + e) >> 1;
                const f32x2 t = {g.raw[tap][q][e], g.raw[tap][q][e + 1]};
                if (tap == 0) sum[k] = ST < 2 ? pk_mul_wlo(g.wa, t) : pk_fma_wlo(g.wa, t, sum[k]);
                else if (tap == 1) sum[k] = pk_fma_whi(g.wa, t, sum[k]);
                else if (tap == 2) sum[k] = pk_fma_wlo(g.wb, t, sum[k]);
                else sum[k] = pk_fma_whi(g.wb, t, sum[k]);
            }
#endif
}

// Two half-plane buffers (32 VGPRs each): stage s+1 is in flight while stage s is accumulated.  The interpolated
// pre-activations are summed in plain VGPRs first and only then become the MFMA accumulator of the 16 view-input
// MFMAs (updating MFMA accumulators with VALU adds in between made hipcc spill heavily).
//
// Round 4: the GATHER runs in another lane order than the matrix instructions.  The L1 (TCP) looks up one cache line per cycle and works
// through a wave's load four consecutive lanes at a time; lane (s, kq) = kq * 16 + s of the matrix layout puts four DIFFERENT samples into every
// such quad -- four lines per quad, 44.5 tag look-ups per load instruction measured (TCP_TOTAL_CACHE_ACCESSES, profiles/r04_triplane_R1_bdc_pmc.txt:
// 2 138 per pass; the texture addresser 91 % busy was what bound level 3).  Here gather lane g = 4 s + kq fetches and interpolates what matrix
// lane kq * 16 + s needs: the four lanes of a quad read ONE 64-byte line of a sample's 256-byte texel (16 look-ups per load), and the sixteen
// sums reach their matrix lane through sixteen ds_bpermute_b32 (a lane permutation through the LDS crossbar: no LDS memory, no VALU slot).
// gcells: the bilinear cells of the GATHER lane's sample (lane >> 2) from its 12-float queue record, or null: then its coordinates are
// fetched from the record of matrix lane (lane >> 2, 0) with six lane permutations
// The front part of the baked pass for the image layout L (MlpLayout16Baked / MlpLayout16BakedBf16): gathers, interpolation, the lane permutation
// and the per-ray view fold (or bias + view-input MFMAs) -> acc = layer-1 pre-activations of the lane's sample, accumulator order.  blob: per_pass16'd.
template <typename L>
__device__ __forceinline__ void baked16_layer1(const RenderArgs &A, const float *blob, const float rec[kRecFloats], const f32x4 v,
                                               int lane, f32x4 acc[4], const float *pre, const RecCells *gcells)
{
    const int kq = lane >> 4;
    // gather role: sample lane >> 2, channel quarter lane & 3
    const int sg = lane >> 2, kqg = lane & 3;
    float rg[kRecFloats];
    rg[0] = rec[0]; rg[1] = rec[1];
    if (!gcells) {
#pragma unroll
        for (int k = 2; k < kRecFloats; ++k) rg[k] = __shfl(rec[k], sg);
    } else {
#pragma unroll
        for (int k = 2; k < kRecFloats; ++k) rg[k] = 0.0f;
    }
    BakedHalf ga, gb;
    baked16_issue<0>(A, rg, kqg, ga, gcells);
    __builtin_amdgcn_sched_barrier(0);
    baked16_issue<1>(A, rg, kqg, gb, gcells);
    __builtin_amdgcn_sched_barrier(0);
    f32x2 sum[8];
    baked16_consume<0>(ga, sum);
    __builtin_amdgcn_sched_barrier(0);
    baked16_issue<2>(A, rg, kqg, ga, gcells);
    __builtin_amdgcn_sched_barrier(0);
    baked16_consume<1>(gb, sum);
    __builtin_amdgcn_sched_barrier(0);
    baked16_issue<3>(A, rg, kqg, gb, gcells);
    __builtin_amdgcn_sched_barrier(0);
    baked16_consume<2>(ga, sum);
    __builtin_amdgcn_sched_barrier(0);
    baked16_issue<4>(A, rg, kqg, ga, gcells);
    __builtin_amdgcn_sched_barrier(0);
    baked16_consume<3>(gb, sum);
    __builtin_amdgcn_sched_barrier(0);
    baked16_issue<5>(A, rg, kqg, gb, gcells);
    __builtin_amdgcn_sched_barrier(0);
    baked16_consume<4>(ga, sum);
    __builtin_amdgcn_sched_barrier(0);
    baked16_consume<5>(gb, sum);
    __builtin_amdgcn_sched_barrier(0);
    // gather lane 4 s + kq -> matrix lane kq * 16 + s, plus the per-ray view fold (or the bias) of the matrix lane's own sample
    const int src = 4 * (lane & 15) + kq;
    const float *b0 = pre ? pre + kq * 16 : blob + L::B1 + kq * 16;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[mt][e] = b0[4 * mt + e] + __shfl(sum[(4 * mt + e) >> 1][e & 1], src);
    if (!pre) {
        const float *w1 = blob + L::W1V + lane;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = NGF_MFMA16(w1[(mt * 4 + j) * 64], v[j], acc[mt]);
    }
}

__device__ __forceinline__ void mlp_pass16_baked(const RenderArgs &A, const float *blob, const float rec[kRecFloats], const f32x4 v,
                                                 int lane, float rgb[3], const float *pre = nullptr, const RecCells *gcells = nullptr)
{
    using L = MlpLayout16Baked;
    blob = per_pass16(blob);
    f32x4 acc[4];
    baked16_layer1<L>(A, blob, rec, v, lane, acc, pre, gcells);
    mlp_tail16(blob, L::W2, L::B2, L::W3, L::B3, lane, acc, rgb);
}

}  // namespace ngf
