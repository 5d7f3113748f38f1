// ngf_infoinv.hpp -- InfoInv field policy for the fused kernel (InfoInv/models/Field.py:43-89,
// InfoInv/models/networks.py:34-54).
//
// No gauge; each plane has 24 density + 72 colour channels; the plane features are multiplied
// channel-wise by a sinusoidal encoding of the normalised position before the MLPs; density is a
// 72-32-32-1 MLP evaluated on EVERY in-box sample, so it runs on the matrix cores inside the march:
// per step the wave's 64 samples form two 32-column B tiles.  Lane l owns sample l and holds its 72
// inputs; v_permlane32_swap of (input 2t, input 2t+1) produces, in one instruction, the B operand of
// both column tiles (tile 0: lane (s,hi) carries input 2t+hi of sample s; tile 1: of sample 32+s).
#pragma once
#include <vector>

#include "ngf_render.hpp"

namespace ngf {

constexpr int kInfoInvWaves = 8;

struct InfoInvDensLayout {                  // floats, relative to MlpLayout<72>::TOTAL inside the blob
    static constexpr int D1 = 0;                    // [36 k-steps][64 lanes] : W1[l&31][2t + (l>>5)]
    static constexpr int D2 = D1 + 36 * 64;         // [16 k-steps][64 lanes] : W2[l&31][row(t, l>>5)]
    static constexpr int B1 = D2 + 16 * 64;         // [2 hi][16]
    static constexpr int B2 = B1 + 32;              // [2 hi][16]
    static constexpr int W3 = B2 + 32;              // [2 hi][16]
    static constexpr int B3 = W3 + 32;              // [4]
    static constexpr int TOTAL = B3 + 4;
};

inline void build_infoinv_density_image(const std::vector<float> &w1, const std::vector<float> &b1, const std::vector<float> &w2,
                                        const std::vector<float> &b2, const std::vector<float> &w3, const std::vector<float> &b3,
                                        float *img)
{
    using D = InfoInvDensLayout;
    for (int t = 0; t < 36; ++t)
        for (int l = 0; l < 64; ++l) img[D::D1 + t * 64 + l] = w1[(size_t)(l & 31) * 72 + 2 * t + (l >> 5)];
    for (int t = 0; t < 16; ++t)
        for (int l = 0; l < 64; ++l) img[D::D2 + t * 64 + l] = w2[(size_t)(l & 31) * 32 + ((t & 3) + 8 * (t >> 2) + 4 * (l >> 5))];
    for (int hi = 0; hi < 2; ++hi)
        for (int r = 0; r < 16; ++r) {
            const int n = (r & 3) + 8 * (r >> 2) + 4 * hi;
            img[D::B1 + hi * 16 + r] = b1[n];
            img[D::B2 + hi * 16 + r] = b2[n];
            img[D::W3 + hi * 16 + r] = w3[n];
        }
    img[D::B3] = b3[0];
    img[D::B3 + 1] = img[D::B3 + 2] = img[D::B3 + 3] = 0.0f;
}

__device__ __forceinline__ void swap32(float &a, float &b)
{
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}

// ---- NGF_F_SPLIT_BF16 for InfoInv: rgb_decoder (216 features + view -> 64 -> 64 -> 3) on v_mfma_f32_32x32x16_bf16 -----------------
// Same technique as ngf_shade_bf16.hpp (3-term bf16 splits, six products per fp32 product, fp32 accumulate) on the 32-sample /
// two-lanes-per-sample pass of mlp_pass<72>: lane (s, hi) supplies 8 of its inputs per k-block (k = 8 hi + e of the instruction's
// 16), 15 k-blocks cover its 108 features + 8 view inputs (+4 zero pads).  Layer 1's A fragments (92 KB) do not fit LDS next to the
// density MLP: they stream from L2 as [k-block][2 unit tiles][3 parts][lane][8 bf16], requested one stage (a gather wait + 48
// interpolations) before their MFMAs.  Layer 2 (4 k-blocks) keeps its fragments in LDS.  180 + 48 bf16 MFMAs of 8 passes replace
// 232 + 64 fp32 MFMAs of 16 passes per 32 samples.
// MEASURED (profiles/r02_infoinv_split.txt): correct (max |rgb - fp32 path| < 2e-6) but NOT faster -- 13.77 ms vs 13.56 ms per frame.
// The two-lanes-per-sample pass already carries 32 accumulators + 36 PE factors + 48 gather registers per lane; the fragments and
// splits push it past the 256 registers two waves per SIMD have (88 spills), and with one wave per SIMD (no spills) it is
// 14.8 ms.  Kept as a tested variant behind the flag; the way forward is a four-lanes-per-sample pass (16 accumulators, 18 channels
// per lane and plane), see DESIGN.md section 9.
struct MlpLayoutBf16II {                      // LDS image (floats)
    static constexpr int KB1 = 15, KB2 = 4;
    static constexpr int W2 = 0;                              // [2 mt][4 kb][3 parts][64 lanes][4]
    static constexpr int B1 = W2 + 2 * KB2 * 3 * 64 * 4;      // [2 hi][32] fp32, accumulator order
    static constexpr int B2 = B1 + 64;
    static constexpr int W3 = B2 + 64;                        // [3][2 hi][32]
    static constexpr int B3 = W3 + 192;
    static constexpr int TOTAL = B3 + 4;
};
constexpr int kW1PackII = MlpLayoutBf16II::KB1 * 2 * 3 * 64 * 4;      // floats of the streamed layer-1 image

// k-block b of the streamed layer-1 image, both unit tiles, one part (0 = hi, 1 = mid, 2 = lo): 8 registers
struct APartII { bf16x8 t0, t1; };
__device__ __forceinline__ APartII apart_ii_load(const float *pack, int b, int part, int lane)
{
    const bf16x8 *p = reinterpret_cast<const bf16x8 *>(pack) + ((size_t)b * 6) * 64 + lane;
    return APartII{p[part * 64], p[(3 + part) * 64]};
}
#define NGF_MFMA_BF16_32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
// the six products of order <= 2 for both unit tiles; lo, mid, hi = the k-block's A parts (smallest terms first)
__device__ __forceinline__ void six_products_ii(const APartII &lo, const APartII &mid, const APartII &hi, const Split8 &x, f32x16 &c0, f32x16 &c1)
{
    c0 = NGF_MFMA_BF16_32(lo.t0, x.h, c0);   c1 = NGF_MFMA_BF16_32(lo.t1, x.h, c1);
    c0 = NGF_MFMA_BF16_32(hi.t0, x.l, c0);   c1 = NGF_MFMA_BF16_32(hi.t1, x.l, c1);
    c0 = NGF_MFMA_BF16_32(mid.t0, x.m, c0);  c1 = NGF_MFMA_BF16_32(mid.t1, x.m, c1);
    c0 = NGF_MFMA_BF16_32(mid.t0, x.h, c0);  c1 = NGF_MFMA_BF16_32(mid.t1, x.h, c1);
    c0 = NGF_MFMA_BF16_32(hi.t0, x.m, c0);   c1 = NGF_MFMA_BF16_32(hi.t1, x.m, c1);
    c0 = NGF_MFMA_BF16_32(hi.t0, x.h, c0);   c1 = NGF_MFMA_BF16_32(hi.t1, x.h, c1);
}
struct AFragII { APartII hi, mid, lo; };
__device__ __forceinline__ void afrag_ii_load(const float *pack, int b, int lane, AFragII &f)
{
    f.hi = apart_ii_load(pack, b, 0, lane);
    f.mid = apart_ii_load(pack, b, 1, lane);
    f.lo = apart_ii_load(pack, b, 2, lane);
}
__device__ __forceinline__ void six_products_ii(const AFragII &f, const Split8 &x, f32x16 &c0, f32x16 &c1) { six_products_ii(f.lo, f.mid, f.hi, x, c0, c1); }

__device__ __forceinline__ void mlp_pass_bf16_ii(const RenderArgs &A, const float *blob, const float rec[kRecFloats], const float *vf, int lane,
                                                 int mode, float rgb[3])
{
    using L = MlpLayoutBf16II;
    constexpr int APP = 72, HALF = 36, CH = 3, CPP = 3, NST = 9;
    blob = per_pass(blob);
    const int hi = lane >> 5;
    f32x4 raw[4][CH];
    auto issue = [&](int st) {
        const int p = st / CPP, q0 = (st % CPP) * CH;
        const Tex &t = A.app[p];
        const Bil b = bil_setup(rec[2 + 2 * p], rec[3 + 2 * p], t);
        const f32x4 *t00 = reinterpret_cast<const f32x4 *>(t.p + (size_t)b.idx * APP + hi * HALF) + q0;
        const f32x4 *t01 = t00 + (size_t)t.stride * (APP / 4);
#pragma unroll
        for (int q = 0; q < CH; ++q) {
            raw[0][q] = t00[q];
            raw[1][q] = t00[APP / 4 + q];
            raw[2][q] = t01[q];
            raw[3][q] = t01[APP / 4 + q];
        }
    };
    issue(0);
    __builtin_amdgcn_sched_barrier(0);
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        acc0[r] = blob[L::B1 + hi * 32 + r];
        acc1[r] = blob[L::B1 + hi * 32 + 16 + r];
    }
    // plane_feature * PE_12(xyz): this lane needs the 36 factors of its half (hi = 0: sines, hi = 1: cosines), once per pass
    float pe[36];
    if (mode) {
        const float pe_xyz[3] = {rec[2], rec[3], rec[5]};                          // xyz = cat(xy, yz[:,1:])
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float sn[12], cs[12];
            pe_ladder<12>(pe_xyz[k], sn, cs);
#pragma unroll
            for (int f = 0; f < 12; ++f) pe[k * 12 + f] = hi ? cs[f] : sn[f];
        }
    }
    float carry[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    AFragII F0;
#pragma unroll
    for (int st = 0; st < NST; ++st) {
        const int p = st / CPP, q0 = (st % CPP) * CH;
        const int b0 = 3 * (st >> 1) + ((st & 1) ? 1 : 0);            // first k-block this stage completes
        __builtin_amdgcn_sched_barrier(0);
        afrag_ii_load(A.basis_pack, b0, lane, F0);                    // requested before the stage's gather is waited for
        __builtin_amdgcn_sched_barrier(0);
        float feat[4 * CH];
        {
            const Bil b = bil_setup(rec[2 + 2 * p], rec[3 + 2 * p], A.app[p]);
#pragma unroll
            for (int q = 0; q < CH; ++q)
#pragma unroll
                for (int e = 0; e < 4; ++e) feat[4 * q + e] = bil_mix(b, raw[0][q][e], raw[1][q][e], raw[2][q][e], raw[3][q][e]);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (st + 1 < NST) issue(st + 1);
        __builtin_amdgcn_sched_barrier(0);
        if (mode) {
#pragma unroll
            for (int jj = 0; jj < 4 * CH; ++jj) feat[jj] = feat[jj] * pe[4 * q0 + jj];
        }
        if (!(st & 1)) {
            six_products_ii(F0, split8(feat), acc0, acc1);
#pragma unroll
            for (int e = 0; e < 4; ++e) carry[e] = feat[8 + e];
        } else {
            const float xa[8] = {carry[0], carry[1], carry[2], carry[3], feat[0], feat[1], feat[2], feat[3]};
            six_products_ii(F0, split8(xa), acc0, acc1);
            __builtin_amdgcn_sched_barrier(0);
            afrag_ii_load(A.basis_pack, b0 + 1, lane, F0);            // the stage's second k-block: behind the twelve MFMAs just issued
            __builtin_amdgcn_sched_barrier(0);
            six_products_ii(F0, split8(feat + 4), acc0, acc1);
        }
    }
    {   // k-blocks 13, 14: the last four features + the lane's eight view inputs (+ four zero pads)
        __builtin_amdgcn_sched_barrier(0);
        afrag_ii_load(A.basis_pack, 13, lane, F0);
        const f32x4 va = *reinterpret_cast<const f32x4 *>(vf + hi * 8), vb = *reinterpret_cast<const f32x4 *>(vf + hi * 8 + 4);
        __builtin_amdgcn_sched_barrier(0);
        const float xa[8] = {carry[0], carry[1], carry[2], carry[3], va[0], va[1], va[2], va[3]};
        const float xb[8] = {vb[0], vb[1], vb[2], vb[3], 0.0f, 0.0f, 0.0f, 0.0f};
        six_products_ii(F0, split8(xa), acc0, acc1);
        __builtin_amdgcn_sched_barrier(0);
        afrag_ii_load(A.basis_pack, 14, lane, F0);
        __builtin_amdgcn_sched_barrier(0);
        six_products_ii(F0, split8(xb), acc0, acc1);
    }
    __builtin_amdgcn_sched_barrier(0);
    // layer 2: the lane's 32 hidden activations as four B fragments; A fragments from LDS
    f32x16 c0, c1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        c0[r] = blob[L::B2 + hi * 32 + r];
        c1[r] = blob[L::B2 + hi * 32 + 16 + r];
    }
    const float *w2 = blob + L::W2 + lane * 4;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float h[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { const int k = 8 * q + e; h[e] = relu1(k < 16 ? acc0[k & 15] : acc1[k & 15]); }
        auto frag = [&](int mt, int part) { return *reinterpret_cast<const bf16x8 *>(w2 + ((mt * L::KB2 + q) * 3 + part) * 64 * 4); };
        const APartII ahi{frag(0, 0), frag(1, 0)}, amid{frag(0, 1), frag(1, 1)}, alo{frag(0, 2), frag(1, 2)};
        six_products_ii(alo, amid, ahi, split8(h), c0, c1);
    }
    // layer 3 on the VALU as in mlp_tail
    const float *w3 = blob + L::W3 + hi * 32;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 32; ++k) s = fmaf(w3[c * 64 + k], relu1(k < 16 ? c0[k & 15] : c1[k & 15]), s);
        s = s + __shfl_xor(s, 32);
        s = s + blob[L::B3 + c];
        rgb[c] = 1.0f / (1.0f + expf(-s));
    }
}

template <bool SPLIT>
struct InfoInvPolicyT {
    static constexpr int RGB_FLOATS = SPLIT ? MlpLayoutBf16II::TOTAL : MlpLayout<72>::TOTAL;      // the density image follows the colour image in LDS
    static constexpr int APP = 72;
    static constexpr bool INFOINV = true;
    static constexpr int WAVES = kInfoInvWaves;
    static constexpr bool PROFILE = false;
    static constexpr bool VLDS = true;
    static constexpr bool VIEW_FOLD = false;
    static constexpr bool STAGED = false;
    static constexpr int STAGE_FLOATS = 0;
    static constexpr int VFEAT_FLOATS = kWave * kViewFeat;
    static constexpr int NSTEP = 1;
    static constexpr int BATCH = kBatch;
    static constexpr int RING = 128;

    // called by all 64 lanes; returns sigma of the lane's own sample (0 when !valid)
    __device__ static __forceinline__ float sigma(const RenderArgs &A, const float *smem, bool valid, const float x[3], int lane,
                                                  float t[6])
    {
        using D = InfoInvDensLayout;
        const float *img = per_pass(smem) + RGB_FLOATS;
        const int hi = lane >> 5;
        // transform (Field.py:43-50): identity split
        t[0] = x[0]; t[1] = x[1]; t[2] = x[1]; t[3] = x[2]; t[4] = x[0]; t[5] = x[2];
        if (!__any(valid)) return 0.0f;

        // PE_4(xyz): [x*1,x*2,x*4,x*8, y.., z..] -> sin(12), cos(12)   (networks.py:227-237)
        float pe[24];
        if (A.mode) {
#pragma unroll
            for (int k = 0; k < 3; ++k) pe_ladder<4>(x[k], pe + k * 4, pe + 12 + k * 4);
        }
        f32x16 h0, h1;   // column tile 0 (samples of lanes 0..31) and 1 (lanes 32..63)
#pragma unroll
        for (int r = 0; r < 16; ++r) h0[r] = h1[r] = img[D::B1 + hi * 16 + r];
        const float *d1 = img + D::D1 + lane;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const Tex &tx = A.dens[p];
            float feat[24];
            if (valid) {
                Bil b = bil_setup(t[2 * p], t[2 * p + 1], tx);
                const f32x4 *q00 = reinterpret_cast<const f32x4 *>(tx.p + (size_t)b.idx * 24);
                const f32x4 *q01 = q00 + (size_t)tx.stride * 6;
#pragma unroll
                for (int q = 0; q < 6; ++q) {
                    f32x4 v00 = q00[q], v10 = q00[6 + q], v01 = q01[q], v11 = q01[6 + q];
#pragma unroll
                    for (int e = 0; e < 4; ++e) feat[4 * q + e] = bil_mix(b, v00[e], v10[e], v01[e], v11[e]);
                }
                if (A.mode) {
#pragma unroll
                    for (int c = 0; c < 24; ++c) feat[c] = feat[c] * pe[c];
                }
            } else {
#pragma unroll
                for (int c = 0; c < 24; ++c) feat[c] = 0.0f;
            }
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                float a = feat[2 * j], b = feat[2 * j + 1];
                swap32(a, b);
                const float w = d1[(p * 12 + j) * 64];
                h0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, a, h0, 0, 0, 0);
                h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b, h1, 0, 0, 0);
            }
        }
        f32x16 g0, g1;
#pragma unroll
        for (int r = 0; r < 16; ++r) g0[r] = g1[r] = img[D::B2 + hi * 16 + r];
        const float *d2 = img + D::D2 + lane;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float w = d2[k * 64];
            g0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, fmaxf(h0[k], 0.0f), g0, 0, 0, 0);
            g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, fmaxf(h1[k], 0.0f), g1, 0, 0, 0);
        }
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float w = img[D::W3 + hi * 16 + r];
            s0 = fmaf(w, fmaxf(g0[r], 0.0f), s0);
            s1 = fmaf(w, fmaxf(g1[r], 0.0f), s1);
        }
        s0 = s0 + __shfl_xor(s0, 32);
        s1 = s1 + __shfl_xor(s1, 32);
        const float f = (hi ? s1 : s0) + img[D::B3];
        return valid ? softplus_shift(f) : 0.0f;
    }
    __device__ static __forceinline__ void shade(const RenderArgs &A, const float *smem, const float rec[kRecFloats], const float *vf,
                                                 const float *, int lane, float c[3], unsigned long long * = nullptr)
    {
        if constexpr (SPLIT) mlp_pass_bf16_ii(A, smem, rec, vf, lane, A.mode, c);
        else mlp_pass<72, true, 3>(A, smem, rec, vf, lane, A.mode, c);
    }
};
using InfoInvPolicy = InfoInvPolicyT<false>;
using InfoInvSplitPolicy = InfoInvPolicyT<true>;

}  // namespace ngf
