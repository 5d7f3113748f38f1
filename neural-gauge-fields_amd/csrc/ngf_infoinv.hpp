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

struct InfoInvPolicy {
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
        const float *img = per_pass(smem) + MlpLayout<72>::TOTAL;
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
        mlp_pass<72, true, 3>(A, smem, rec, vf, lane, A.mode, c);
    }
};

}  // namespace ngf
