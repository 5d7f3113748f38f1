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
#include <cstdint>
#include <cstring>
#include <vector>

#include "ngf_render.hpp"

namespace ngf {

constexpr int kInfoInvWaves = 12;
#ifndef NGF_II_TAP_PARTS
#define NGF_II_TAP_PARTS 3        // density taps of the fp32 march: the 24 channels are fetched in this many parts (2: 48 tap registers in flight and 44 B of
                                  // spills at the 168 registers of twelve waves per CU; 3: 32 and none -- same speed in an alternating A/B, profiles/r04_infoinv_tap_parts.txt)
#endif
constexpr int kInfoInvSplitWaves = 8;        // NGF_F_SPLIT_BF16: 102 KB of MLP images + 5.4 KB per wave

struct InfoInvDensLayout {                  // floats, relative to MlpLayout16<72>::TOTAL inside the blob
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

// NGF_F_SPLIT_BF16: the density MLP of the march on v_mfma_f32_32x32x16_bf16 with 3-term split operands (sigma_bf16 below).
// A fragments: lane (i = l & 31, hi = l >> 5), element e of k-block kb holds W[i][16 kb + 8 hi + e] -- layer 1 over the 72 inputs in
// their natural order (plane p, channel c -> 24 p + c; 8 zero pads), layer 2 over the hidden units in the accumulator order of the lane
// half (k = 8 q + e  ->  unit (k & 3) + 8 (k >> 2) + 4 hi), so that a lane's own ReLU'd accumulators are its B fragments.
struct InfoInvDensLayoutBf16 {              // floats (a bf16x8 fragment = 4 floats), relative to MlpLayoutBf16II::TOTAL inside the blob
    static constexpr int KB1 = 5, KB2 = 2;
    static constexpr int D1 = 0;                    // [5 kb][3 parts][64 lanes][4]
    static constexpr int D2 = D1 + KB1 * 3 * 64 * 4;   // [2 kb][3 parts][64 lanes][4]
    static constexpr int B1 = D2 + KB2 * 3 * 64 * 4;   // [2 hi][16]
    static constexpr int B2 = B1 + 32;
    static constexpr int W3 = B2 + 32;
    static constexpr int B3 = W3 + 32;
    static constexpr int TOTAL = B3 + 4;
};

inline void infoinv_split3(float x, uint16_t out[3])
{
    auto f2bf = [](float v) { uint32_t u; memcpy(&u, &v, 4); const uint32_t r = 0x7fffu + ((u >> 16) & 1u); return (uint16_t)((u + r) >> 16); };
    auto bf2f = [](uint16_t h) { const uint32_t u = (uint32_t)h << 16; float v; memcpy(&v, &u, 4); return v; };
    out[0] = f2bf(x);
    const float r1 = x - bf2f(out[0]);
    out[1] = f2bf(r1);
    const float r2 = r1 - bf2f(out[1]);
    out[2] = f2bf(r2);
}

inline void build_infoinv_density_image_bf16(const std::vector<float> &w1, const std::vector<float> &b1, const std::vector<float> &w2,
                                             const std::vector<float> &b2, const std::vector<float> &w3, const std::vector<float> &b3,
                                             float *img)
{
    using D = InfoInvDensLayoutBf16;
    uint16_t *h16 = reinterpret_cast<uint16_t *>(img);
    auto unit = [](int k, int hi) { return (k & 3) + 8 * (k >> 2) + 4 * hi; };
    for (int kb = 0; kb < D::KB1; ++kb)
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 8; ++e) {
                const int k = 16 * kb + 8 * (l >> 5) + e;
                uint16_t p3[3];
                infoinv_split3(k < 72 ? w1[(size_t)(l & 31) * 72 + k] : 0.0f, p3);
                for (int part = 0; part < 3; ++part) h16[((size_t)D::D1 + (((size_t)kb * 3 + part) * 64 + l) * 4) * 2 + e] = p3[part];
            }
    for (int q = 0; q < D::KB2; ++q)
        for (int l = 0; l < 64; ++l)
            for (int e = 0; e < 8; ++e) {
                uint16_t p3[3];
                infoinv_split3(w2[(size_t)(l & 31) * 32 + unit(8 * q + e, l >> 5)], p3);
                for (int part = 0; part < 3; ++part) h16[((size_t)D::D2 + (((size_t)q * 3 + part) * 64 + l) * 4) * 2 + e] = p3[part];
            }
    for (int hi = 0; hi < 2; ++hi)
        for (int r = 0; r < 16; ++r) {
            const int n = unit(r, hi);
            img[D::B1 + hi * 16 + r] = b1[n];
            img[D::B2 + hi * 16 + r] = b2[n];
            img[D::W3 + hi * 16 + r] = w3[n];
        }
    img[D::B3] = b3[0];
    img[D::B3 + 1] = img[D::B3 + 2] = img[D::B3 + 3] = 0.0f;
}

// One octave step of the positional factors: f_s *= sin, f_c *= cos, then the angle doubling (sin 2a = 2 sin cos, cos 2a = (cos - sin)(cos + sin)).
// Written as SINGLE VALU instructions on purpose.  hipcc 7.2 packs the plain C++ form of this chain into v_pk_mul_f32 / v_pk_add_f32 with
// op_sel / neg modifiers and v_pk_mov_b32, and with that code the NGF_F_SPLIT_BF16 colour pass was NOT deterministic: identical records,
// factors and features (dumped per lane, profiles/r03_determinism.txt) gave layer-1 accumulators that differed at the 1e-3 level in about one
// launch in 50 000 (one 4-ray tile each time; 1 in 5 with idle slots between the bf16 MFMAs), which is the once-in-forty-suite-runs failure
// of round 2.  The same IEEE operations as un-packed instructions: 0 differences in 600 000 launches.  -DNGF_EXP_PACKED_PE restores the
// compiler's code (profiles/exp_determinism_builds.sh); a stand-alone micro-benchmark of the instruction pattern
// (profiles/micro/pk_hi_forward.hip) does not reproduce the effect, so the mechanism below the ISA is not known.
__device__ __forceinline__ void pe_octave(float &fs, float &fc, float &sn, float &cs)
{
#ifdef NGF_EXP_PACKED_PE
    fs *= sn; fc *= cs;
    const float s2 = 2.0f * sn * cs, c2 = (cs - sn) * (cs + sn);
    sn = s2; cs = c2;
#else
    float t, d, e;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(fs) : "v"(fs), "v"(sn));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(fc) : "v"(fc), "v"(cs));
    asm volatile("v_add_f32 %0, %1, %1" : "=v"(t) : "v"(sn));               // 2 sin   (2.0f * sn, exact)
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(cs), "v"(sn));
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(e) : "v"(cs), "v"(sn));
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(sn) : "v"(t), "v"(cs));      // (2 sin) cos -- the association of `2.0f * sn * cs`
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(cs) : "v"(d), "v"(e));
#endif
}

__device__ __forceinline__ void swap32(float &a, float &b)
{
    auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    a = __uint_as_float(r[0]);
    b = __uint_as_float(r[1]);
}

// ---- NGF_F_SPLIT_BF16 for InfoInv: rgb_decoder layer 1 (216 features + 15 view inputs -> 64) on v_mfma_f32_16x16x32_bf16 ---------------
// The technique of ngf_shade_bf16.hpp (3-term bf16 splits, six products per fp32 product, fp32 accumulate, fp32-level error) on the
// 16-sample / FOUR-lanes-per-sample pass: lane (s, kq) owns 16 accumulators and supplies 64 of the 256 padded inputs -- 18 channels of
// each plane, its 4 view entries, 6 zero pads -- as eight 8-element B fragments.  Layer 2 runs the same way (two k-blocks), layer 3 on
// the VALU.  The hi and mid parts of layer 1 (64 KB), layer 2 (24 KB) and the density MLP make a 102 KB LDS image with room for eight
// waves per CU; layer 1's lo parts (32 KB, the term that only meets x.hi) stream from L2, one k-block pair ahead of their MFMAs.
//   * The colour channels of a plane are PERMUTED at pack time (pack_plane_kernel, perm = 1) so that a lane's 18 channels are 72
//     contiguous bytes: position kq*18 + (hi*3 + axis)*3 + j holds channel hi*36 + axis*12 + 3 kq + j, i.e. lane-quarter kq owns the
//     octaves 3kq .. 3kq+2 of the positional encoding of every axis, sine and cosine (Field.py:72-84: feature c is multiplied by
//     sin / cos (x_axis 2^f)).  Its 18 factors are three sincosf at octave 3kq and two angle doublings each -- the two-lanes-per-sample
//     pass computes all twelve octaves of the three axes in every lane.
//   * A plane's four taps arrive in two rows; the bilinear sum is accumulated in the order of bil_mix (w00 v00, + w10 v10, + w01 v01,
//     + w11 v11), so the features are bit-identical to the fp32 path's and 36 registers of gather are in flight instead of 72.
// The first version of this flag (two lanes per sample on v_mfma_f32_32x32x16_bf16, layer 1 streamed from L2) was register-bound and
// not faster (profiles/r02_infoinv_split.txt).
struct MlpLayoutBf16II {                      // LDS image (floats); a bf16x8 fragment = 4 floats
    static constexpr int KB1 = 8, KB2 = 2;
    static constexpr int W1 = 0;                              // [4 mt][8 kb][2 parts: hi, mid][64 lanes][4]
    static constexpr int W2 = W1 + 4 * KB1 * 2 * 64 * 4;      // [4 mt][2 kb][3 parts][64 lanes][4]
    static constexpr int B1 = W2 + 4 * KB2 * 3 * 64 * 4;      // [4 kq][16] fp32, accumulator order
    static constexpr int B2 = B1 + 64;
    static constexpr int W3 = B2 + 64;                        // [3][4 kq][16]
    static constexpr int B3 = W3 + 192;
    static constexpr int TOTAL = B3 + 4;
};
constexpr int kW1LoPackII = MlpLayoutBf16II::KB1 * 4 * 64 * 4;      // floats of the streamed image: layer 1's lo parts [kb][mt][lane][8 bf16]
// packed position (0..71) of a plane's colour channels -> channel of the reference layout
__host__ __device__ __forceinline__ int infoinv_split_channel(int pos)
{
    const int kq = pos / 18, r = pos % 18, g = r / 3, j = r % 3;        // g = hi*3 + axis
    return (g / 3) * 36 + (g % 3) * 12 + 3 * kq + j;
}

struct GatherRowII { f32x4 a[2][4]; f32x2 b[2]; };               // one row of a cell: two taps x 18 channels (36 registers)
__device__ __forceinline__ void gather_row_ii(const float *base, GatherRowII &g)
{
    // base: the lane's 18 channels of the row's first tap; the second tap is the next texel (72 channels on)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const float *p = base + t * 72;
#pragma unroll
        for (int q = 0; q < 4; ++q) g.a[t][q] = *reinterpret_cast<const f32x4 *>(p + 4 * q);
        g.b[t] = *reinterpret_cast<const f32x2 *>(p + 16);
    }
}

// layer 1's lo parts of one k-block (the term that only meets x.hi), streamed from L2: 16 registers
struct LoFragII { bf16x8 t[4]; };
__device__ __forceinline__ void lo_load_ii(const float *pack, int kb, int lane, LoFragII &f)
{
    const bf16x8 *p = reinterpret_cast<const bf16x8 *>(pack) + (size_t)kb * 4 * 64 + lane;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) f.t[mt] = p[mt * 64];
}
// acc[mt] += W1[mt][kb] . x: hi / mid fragments from LDS (w = image + kb offset + lane), lo from `lo`; the unit tiles two at a time
__device__ __forceinline__ void kblock_ii(const float *w, const LoFragII &lo, const Split8 &x, f32x4 acc[4])
{
    constexpr int MT = MlpLayoutBf16II::KB1 * 2 * 64 * 4;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        AFrag a0, a1;
        a0.h = *reinterpret_cast<const bf16x8 *>(w + (2 * h) * MT);      a0.m = *reinterpret_cast<const bf16x8 *>(w + (2 * h) * MT + 64 * 4);      a0.l = lo.t[2 * h];
        a1.h = *reinterpret_cast<const bf16x8 *>(w + (2 * h + 1) * MT);  a1.m = *reinterpret_cast<const bf16x8 *>(w + (2 * h + 1) * MT + 64 * 4);  a1.l = lo.t[2 * h + 1];
        six_products2(a0, a1, x, acc[2 * h], acc[2 * h + 1]);
        __builtin_amdgcn_sched_barrier(0);
    }
}

__device__ __forceinline__ void mlp_pass16_bf16_ii(const RenderArgs &A, const float *blob, const float rec[kRecFloats], const f32x4 v, int lane,
                                                   int mode, float rgb[3])
{
    using L = MlpLayoutBf16II;
    blob = per_pass16(blob);
    const int kq = lane >> 4;
    constexpr int KB_STRIDE = 2 * 64 * 4;
    const float *w1 = blob + L::W1 + lane * 4;
    GatherRowII g;
    Bil b = bil_setup(rec[2], rec[3], karg_tex(offsetof(RenderArgs, app) + (0) * sizeof(Tex)));
    gather_row_ii(tex_at<float>(karg_tex(offsetof(RenderArgs, app) + (0) * sizeof(Tex)).p, (uint32_t)b.idx * 72u + (uint32_t)kq * 18u), g);
    LoFragII lo0, lo1;                         // k-blocks 2i / 2i+1: each is re-requested for block +2 as soon as it has been used
    lo_load_ii(A.basis_pack, 0, lane, lo0);
    lo_load_ii(A.basis_pack, 1, lane, lo1);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = *reinterpret_cast<const f32x4 *>(blob + L::B1 + kq * 16 + mt * 4);
    // the positional factors of the lane's channels are sin / cos (x_axis 2^(3kq + j)), xyz = cat(xy, yz[:, 1:]): one sincosf per axis
    // at octave 3kq, the two doublings are redone per plane (6 registers instead of 18)
    float bs[3] = {0.0f, 0.0f, 0.0f}, bc[3] = {1.0f, 1.0f, 1.0f};
    if (mode) {
        const float pe_xyz[3] = {rec[2], rec[3], rec[5]};
        const float scale = (float)(1 << (3 * kq));
#pragma unroll
        for (int k = 0; k < 3; ++k) sincos_small(pe_xyz[k] * scale, bs[k], bc[k]);
    }
    float left[4];                             // inputs of a plane that did not fill a k-block yet
#ifdef NGF_EXP_DUMP
    float fh[3];                               // hash of the plane's 18 features (after the positional factors)
    float xfrag[3];                            // hashes of the hi / mid / lo bf16 fragments of k-block 0, read back after its MFMAs
#endif
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        float f[18];
        // row 0: w00 v00, then + w10 v10
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) f[4 * q + e] = fmaf(b.w10, g.a[1][q][e], b.w00 * g.a[0][q][e]);
#pragma unroll
        for (int e = 0; e < 2; ++e) f[16 + e] = fmaf(b.w10, g.b[1][e], b.w00 * g.b[0][e]);
        __builtin_amdgcn_sched_barrier(0);
        gather_row_ii(tex_at<float>(karg_tex(offsetof(RenderArgs, app) + (p) * sizeof(Tex)).p, (uint32_t)(b.idx + karg_tex(offsetof(RenderArgs, app) + (p) * sizeof(Tex)).stride) * 72u + (uint32_t)kq * 18u), g);
        __builtin_amdgcn_sched_barrier(0);
        // row 1: + w01 v01, + w11 v11 (bil_mix's order), then the positional factor
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) f[4 * q + e] = fmaf(b.w11, g.a[1][q][e], fmaf(b.w01, g.a[0][q][e], f[4 * q + e]));
#pragma unroll
        for (int e = 0; e < 2; ++e) f[16 + e] = fmaf(b.w11, g.b[1][e], fmaf(b.w01, g.b[0][e], f[16 + e]));
        __builtin_amdgcn_sched_barrier(0);
        if (p < 2) {                           // the next plane's first row travels behind this plane's MFMAs
            b = bil_setup(rec[4 + 2 * p], rec[5 + 2 * p], karg_tex(offsetof(RenderArgs, app) + (p + 1) * sizeof(Tex)));
            gather_row_ii(tex_at<float>(karg_tex(offsetof(RenderArgs, app) + (p + 1) * sizeof(Tex)).p, (uint32_t)b.idx * 72u + (uint32_t)kq * 18u), g);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (mode) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float sn = bs[k], cs = bc[k];
#pragma unroll
                for (int j = 0; j < 3; ++j) pe_octave(f[k * 3 + j], f[9 + k * 3 + j], sn, cs);
            }
        }
#ifdef NGF_EXP_DUMP
        {
            unsigned hsh = 0;
#pragma unroll
            for (int i = 0; i < 18; ++i) hsh = (hsh * 31u) ^ __float_as_uint(f[i]);
            fh[p] = __uint_as_float(hsh & 0x3fffffffu);
        }
#endif
        // inputs so far: 18 p + the 2 p leftovers of the planes before; whole k-blocks go to the matrix pipe now
        if (p == 0) {
#ifdef NGF_EXP_DUMP
            const Split8 x0 = split8(f);
            kblock_ii(w1 + 0 * KB_STRIDE, lo0, x0, acc);
            {   // the B fragments as the MFMAs saw them (read back AFTER the k-block)
                typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
                const u32x4 xh = __builtin_bit_cast(u32x4, x0.h), xm = __builtin_bit_cast(u32x4, x0.m), xl = __builtin_bit_cast(u32x4, x0.l);
                unsigned hsh = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) hsh = (hsh * 31u) ^ xh[i];
                xfrag[0] = __uint_as_float(hsh & 0x3fffffffu);
                hsh = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) hsh = (hsh * 31u) ^ xm[i];
                xfrag[1] = __uint_as_float(hsh & 0x3fffffffu);
                hsh = 0;
#pragma unroll
                for (int i = 0; i < 4; ++i) hsh = (hsh * 31u) ^ xl[i];
                xfrag[2] = __uint_as_float(hsh & 0x3fffffffu);
            }
            lo_load_ii(A.basis_pack, 2, lane, lo0);  __builtin_amdgcn_sched_barrier(0);
#else
            kblock_ii(w1 + 0 * KB_STRIDE, lo0, split8(f), acc);      lo_load_ii(A.basis_pack, 2, lane, lo0);  __builtin_amdgcn_sched_barrier(0);
#endif
            kblock_ii(w1 + 1 * KB_STRIDE, lo1, split8(f + 8), acc);  lo_load_ii(A.basis_pack, 3, lane, lo1);  __builtin_amdgcn_sched_barrier(0);
            left[0] = f[16]; left[1] = f[17];
        } else if (p == 1) {
            const float x2[8] = {left[0], left[1], f[0], f[1], f[2], f[3], f[4], f[5]};
            kblock_ii(w1 + 2 * KB_STRIDE, lo0, split8(x2), acc);     lo_load_ii(A.basis_pack, 4, lane, lo0);  __builtin_amdgcn_sched_barrier(0);
            kblock_ii(w1 + 3 * KB_STRIDE, lo1, split8(f + 6), acc);  lo_load_ii(A.basis_pack, 5, lane, lo1);  __builtin_amdgcn_sched_barrier(0);
            left[0] = f[14]; left[1] = f[15]; left[2] = f[16]; left[3] = f[17];
        } else {
            const float x4[8] = {left[0], left[1], left[2], left[3], f[0], f[1], f[2], f[3]};
            kblock_ii(w1 + 4 * KB_STRIDE, lo0, split8(x4), acc);     lo_load_ii(A.basis_pack, 6, lane, lo0);  __builtin_amdgcn_sched_barrier(0);
            kblock_ii(w1 + 5 * KB_STRIDE, lo1, split8(f + 4), acc);  lo_load_ii(A.basis_pack, 7, lane, lo1);  __builtin_amdgcn_sched_barrier(0);
            const float x6[8] = {f[12], f[13], f[14], f[15], f[16], f[17], v[0], v[1]};
            kblock_ii(w1 + 6 * KB_STRIDE, lo0, split8(x6), acc);
            const float x7[8] = {v[2], v[3], 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
            kblock_ii(w1 + 7 * KB_STRIDE, lo1, split8(x7), acc);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // layer 2 on the bf16 pipe as in mlp_pass16_bf16: the lane's 16 hidden activations are its two B fragments; layer 3 on the VALU
    f32x4 c[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) c[mt] = *reinterpret_cast<const f32x4 *>(blob + L::B2 + kq * 16 + mt * 4);
    {
        constexpr int KB2_STRIDE = 3 * 64 * 4, MT2 = L::KB2 * KB2_STRIDE;
        const float *w2 = blob + L::W2 + lane * 4;
        float h[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) h[k] = relu1(acc[k >> 2][k & 3]);
        kblock_bf16<MT2>(w2, split8(h), c);
        kblock_bf16<MT2>(w2 + KB2_STRIDE, split8(h + 8), c);
    }
    __builtin_amdgcn_sched_barrier(0);
    mlp_layer3_16(blob, L::W3, L::B3, lane, c, rgb);
#ifdef NGF_EXP_DUMP     // experiment build (profiles/exp_determinism_dump.py): every lane's view of the pass -> A.stats[8 + (pass * 64 + lane) * 64 ..]
    if (A.stats) {
        unsigned long long id = 0;
        if (lane == 0) id = atomicAdd(A.stats, 1ull);
        id = __shfl(id, 0);
        float *row = reinterpret_cast<float *>(A.stats + 8) + (id * 64 + lane) * 64;
        row[0] = (float)lane; row[1] = rec[0]; row[2] = rec[1]; row[3] = rec[2]; row[4] = rec[3]; row[5] = rec[5];
        row[6] = fh[0]; row[7] = fh[1]; row[8] = fh[2];
#pragma unroll
        for (int k = 0; k < 16; ++k) { row[9 + k] = acc[k >> 2][k & 3]; row[25 + k] = c[k >> 2][k & 3]; }
        row[41] = rgb[0]; row[42] = rgb[1]; row[43] = rgb[2];
        row[44] = bs[0]; row[45] = bs[1]; row[46] = bs[2]; row[47] = bc[0]; row[48] = bc[1]; row[49] = bc[2];
        row[50] = xfrag[0]; row[51] = xfrag[1]; row[52] = xfrag[2];
    }
#endif
}

// ---- the default (fp32) colour pass: 16 samples per pass, FOUR lanes per sample, v_mfma_f32_16x16x4_f32 -----------------------------------
// The data flow of mlp_pass16_bf16_ii with fp32 MFMAs (MlpLayout16<72>: W1 [4 mt][58 k-steps][64 lanes], one input of every lane per
// k-step): permuted channels (72 contiguous bytes per lane and tap), a lane's 18 positional factors from three sincos + two doublings,
// taps in two rows.  A lane carries 16 accumulators instead of the 32 (+36 factors, +48 gather registers) of round 1's
// two-lanes-per-sample pass on v_mfma_f32_32x32x2_f32, so the kernel fits the 168 registers of TWELVE waves per CU: the same matrix work,
// more waves to cover the gather and LDS latency that left the SIMDs idle 15 % of the time at eight.
__device__ __forceinline__ void mlp_pass16_ii(const RenderArgs &A, const float *blob, const float rec[kRecFloats], const f32x4 v, int lane, int mode,
                                              float rgb[3])
{
    using L = MlpLayout16<72>;
    blob = per_pass16(blob);
    asm volatile("" : "+v"(lane));       // the lane-derived offsets (kq * 18, lane * 4, ...) are recomputed per pass: hoisted out of the persistent
                                         // loops they lived in registers for the whole kernel and were what hipcc spilled (144 B of scratch per lane)
    const int kq = lane >> 4;
    const float *w1 = blob + L::W1 + lane;
    GatherRowII g;
    Bil b = bil_setup(rec[2], rec[3], karg_tex(offsetof(RenderArgs, app) + (0) * sizeof(Tex)));
    gather_row_ii(tex_at<float>(karg_tex(offsetof(RenderArgs, app) + (0) * sizeof(Tex)).p, (uint32_t)b.idx * 72u + (uint32_t)kq * 18u), g);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = *reinterpret_cast<const f32x4 *>(blob + L::B1 + kq * 16 + mt * 4);
    float bs[3] = {0.0f, 0.0f, 0.0f}, bc[3] = {1.0f, 1.0f, 1.0f};
    if (mode) {
        const float pe_xyz[3] = {rec[2], rec[3], rec[5]};
        const float scale = (float)(1 << (3 * kq));
#pragma unroll
        for (int k = 0; k < 3; ++k) sincos_small(pe_xyz[k] * scale, bs[k], bc[k]);
    }
    // the view inputs first: their MFMAs run while plane 0's first row is on its way
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = NGF_MFMA16(w1[(mt * L::KT + 54 + e) * 64], v[e], acc[mt]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        float f[18];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) f[4 * q + e] = fmaf(b.w10, g.a[1][q][e], b.w00 * g.a[0][q][e]);
#pragma unroll
        for (int e = 0; e < 2; ++e) f[16 + e] = fmaf(b.w10, g.b[1][e], b.w00 * g.b[0][e]);
        __builtin_amdgcn_sched_barrier(0);
        gather_row_ii(tex_at<float>(karg_tex(offsetof(RenderArgs, app) + (p) * sizeof(Tex)).p, (uint32_t)(b.idx + karg_tex(offsetof(RenderArgs, app) + (p) * sizeof(Tex)).stride) * 72u + (uint32_t)kq * 18u), g);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int e = 0; e < 4; ++e) f[4 * q + e] = fmaf(b.w11, g.a[1][q][e], fmaf(b.w01, g.a[0][q][e], f[4 * q + e]));
#pragma unroll
        for (int e = 0; e < 2; ++e) f[16 + e] = fmaf(b.w11, g.b[1][e], fmaf(b.w01, g.b[0][e], f[16 + e]));
        __builtin_amdgcn_sched_barrier(0);
        if (p < 2) {                           // the next plane's first row travels behind this plane's 72 MFMAs
            b = bil_setup(rec[4 + 2 * p], rec[5 + 2 * p], karg_tex(offsetof(RenderArgs, app) + (p + 1) * sizeof(Tex)));
            gather_row_ii(tex_at<float>(karg_tex(offsetof(RenderArgs, app) + (p + 1) * sizeof(Tex)).p, (uint32_t)b.idx * 72u + (uint32_t)kq * 18u), g);
#ifdef NGF_EXP_II_PIN_WEIGHTS
            // (the four weights through an opaque asm: they are pure values, and the instruction selector sank their last ten instructions among this plane's MFMAs)
            asm volatile("" : "+v"(b.w00), "+v"(b.w10), "+v"(b.w01), "+v"(b.w11));
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
        if (mode) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float sn = bs[k], cs = bc[k];
#pragma unroll
                for (int j = 0; j < 3; ++j) pe_octave(f[k * 3 + j], f[9 + k * 3 + j], sn, cs);
            }
        }
#ifdef NGF_EXP_II_PIN_WEIGHTS
        asm volatile("" : "+v"(f[0]), "+v"(f[17]));
        __builtin_amdgcn_sched_barrier(0);
#endif
#pragma unroll
        for (int i = 0; i < 18; ++i)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) acc[mt] = NGF_MFMA16(w1[(mt * L::KT + p * 18 + i) * 64], f[i], acc[mt]);
        __builtin_amdgcn_sched_barrier(0);
    }
    mlp_tail16(blob, L::W2, L::B2, L::W3, L::B3, lane, acc, rgb);
}

// ---- NGF_F_SPLIT_BF16: the density MLP (72 -> 32 -> 32 -> 1, every in-box sample) on the bf16 matrix pipe -------------------------------
// Lane l owns sample l and its 72 inputs.  A k-block is 16 inputs: the lane splits them into (hi, mid, lo) bf16 parts, packs inputs 0..7
// and 8..15 into one fragment each, and v_permlane32_swap of the two turns them into the B fragments of BOTH column tiles (tile 0: lane
// (s, half) carries inputs 8 half .. 8 half + 7 of sample s; tile 1: of sample 32 + s).  Six products per tile and k-block, fp32
// accumulate, smallest terms first -- 84 bf16 MFMAs of 32 cycles per 64 samples instead of 104 fp32 MFMAs of 64 cycles.
#define NGF_MFMA_BF16_32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)
struct SplitPair { Split8 t0, t1; };            // the B fragments of the two column tiles
__device__ __forceinline__ void swap_frag(bf16x8 &a, bf16x8 &b)
{
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 ua = __builtin_bit_cast(u32x4, a), ub = __builtin_bit_cast(u32x4, b);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        auto q = __builtin_amdgcn_permlane32_swap(ua[r], ub[r], false, false);
        ua[r] = q[0]; ub[r] = q[1];
    }
    a = __builtin_bit_cast(bf16x8, ua); b = __builtin_bit_cast(bf16x8, ub);
}
// x[0..15]: the lane's own 16 inputs of the k-block
__device__ __forceinline__ SplitPair split_pair(const float x[16])
{
    SplitPair s;
    s.t0 = split8(x);
    s.t1 = split8(x + 8);
    swap_frag(s.t0.h, s.t1.h); swap_frag(s.t0.m, s.t1.m); swap_frag(s.t0.l, s.t1.l);
    return s;
}
// w: the k-block's A fragments [3 parts][64 lanes][4] (+ lane * 4)
__device__ __forceinline__ void dens_kblock(const float *w, const Split8 &x0, const Split8 &x1, f32x16 &c0, f32x16 &c1)
{
    const bf16x8 ah = *reinterpret_cast<const bf16x8 *>(w), am = *reinterpret_cast<const bf16x8 *>(w + 64 * 4), al = *reinterpret_cast<const bf16x8 *>(w + 2 * 64 * 4);
    c0 = NGF_MFMA_BF16_32(al, x0.h, c0);  c1 = NGF_MFMA_BF16_32(al, x1.h, c1);
    c0 = NGF_MFMA_BF16_32(ah, x0.l, c0);  c1 = NGF_MFMA_BF16_32(ah, x1.l, c1);
    c0 = NGF_MFMA_BF16_32(am, x0.m, c0);  c1 = NGF_MFMA_BF16_32(am, x1.m, c1);
    c0 = NGF_MFMA_BF16_32(am, x0.h, c0);  c1 = NGF_MFMA_BF16_32(am, x1.h, c1);
    c0 = NGF_MFMA_BF16_32(ah, x0.m, c0);  c1 = NGF_MFMA_BF16_32(ah, x1.m, c1);
    c0 = NGF_MFMA_BF16_32(ah, x0.h, c0);  c1 = NGF_MFMA_BF16_32(ah, x1.h, c1);
}

// called by all 64 lanes; img: the density image (InfoInvDensLayoutBf16) in LDS; returns sigma of the lane's own sample (0 when !valid)
__device__ __forceinline__ float infoinv_sigma_bf16(const RenderArgs &A, const float *img, bool valid, const float x[3], int lane, float t[6])
{
    using D = InfoInvDensLayoutBf16;
    const int hi = lane >> 5;
    t[0] = x[0]; t[1] = x[1]; t[2] = x[1]; t[3] = x[2]; t[4] = x[0]; t[5] = x[2];
    if (!__any(valid)) return 0.0f;
    float pe[24];
    if (A.mode) {
#pragma unroll
        for (int k = 0; k < 3; ++k) pe_ladder<4>(x[k], pe + k * 4, pe + 12 + k * 4);
    }
    f32x16 h0, h1;   // column tile 0 (samples of lanes 0..31) and 1 (lanes 32..63)
#pragma unroll
    for (int r = 0; r < 16; ++r) h0[r] = h1[r] = img[D::B1 + hi * 16 + r];
    const float *d1 = img + D::D1 + lane * 4;
    constexpr int KBS = 3 * 64 * 4;
    float carry[8];                                // the 8 inputs of a plane that wait for the next plane's first 8
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const Tex tx = karg_tex(offsetof(RenderArgs, dens) + p * sizeof(Tex));
        float feat[24];
        if (valid) {
            Bil b = bil_setup(t[2 * p], t[2 * p + 1], tx);
            const f32x4 *q00 = tex_at<f32x4>(tx.p, (uint32_t)b.idx * 24u);       // 32-bit byte offsets from a scalar base (ngf_device.hpp tex_at): +1.2 %; as 24-bit multiplies of byte offsets: 0.5 % slower than this
            const f32x4 *q01 = tex_at<f32x4>(tx.p, (uint32_t)(b.idx + tx.stride) * 24u);
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
        // inputs 24 p .. 24 p + 23 -> k-blocks of 16: [p0 0..15] [p0 16..23 | p1 0..7] [p1 8..23] [p2 0..15] [p2 16..23 | 0 x 8]
        // (a branch-free, software-pipelined form -- MFMAs of block b under the interpolation of block b + 1 -- was slower: 10.7 vs 10.4 ms)
        if (p == 0) {
            const SplitPair s = split_pair(feat);
            dens_kblock(d1, s.t0, s.t1, h0, h1);
#pragma unroll
            for (int e = 0; e < 8; ++e) carry[e] = feat[16 + e];
        } else if (p == 1) {
            float xk[16];
#pragma unroll
            for (int e = 0; e < 8; ++e) { xk[e] = carry[e]; xk[8 + e] = feat[e]; }
            const SplitPair s = split_pair(xk);
            dens_kblock(d1 + KBS, s.t0, s.t1, h0, h1);
            const SplitPair s2 = split_pair(feat + 8);
            dens_kblock(d1 + 2 * KBS, s2.t0, s2.t1, h0, h1);
        } else {
            const SplitPair s = split_pair(feat);
            dens_kblock(d1 + 3 * KBS, s.t0, s.t1, h0, h1);
            float xk[16];
#pragma unroll
            for (int e = 0; e < 8; ++e) { xk[e] = feat[16 + e]; xk[8 + e] = 0.0f; }
            const SplitPair s2 = split_pair(xk);
            dens_kblock(d1 + 4 * KBS, s2.t0, s2.t1, h0, h1);
        }
    }
    // layer 2: the lane's ReLU'd accumulators (hidden units in the accumulator order of its half) are its B fragments, tile by tile
    f32x16 g0, g1;
#pragma unroll
    for (int r = 0; r < 16; ++r) g0[r] = g1[r] = img[D::B2 + hi * 16 + r];
    const float *d2 = img + D::D2 + lane * 4;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        float a0[8], a1[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { a0[e] = relu1(h0[8 * q + e]); a1[e] = relu1(h1[8 * q + e]); }
        dens_kblock(d2 + q * KBS, split8(a0), split8(a1), g0, g1);
    }
    float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const float w = img[D::W3 + hi * 16 + r];
        s0 = fmaf(w, relu1(g0[r]), s0);
        s1 = fmaf(w, relu1(g1[r]), s1);
    }
    s0 = s0 + __shfl_xor(s0, 32);
    s1 = s1 + __shfl_xor(s1, 32);
    const float f = (hi ? s1 : s0) + img[D::B3];
    return valid ? softplus_shift(f) : 0.0f;
}

// WIDE: tiles of up to 64 rays (the unsplit, one-ray-per-lane march: test / debug knobs only) -- the per-ray view inputs of 64 rays take
// 4 KB per wave, so eight waves per CU; the default keeps the view inputs of 16 rays and runs twelve.
template <bool SPLIT, bool WIDE = false>
struct InfoInvPolicyT {
    static constexpr int RGB_FLOATS = SPLIT ? MlpLayoutBf16II::TOTAL : MlpLayout16<72>::TOTAL;      // the density image follows the colour image in LDS
    static constexpr int APP = 72;
    static constexpr bool INFOINV = true;
    static constexpr bool MASK_SKIP = false;                      // ngf_render.hpp: MaskSkip<P> is the instantiation with the march's empty-space skipping
    static constexpr int WAVES = (SPLIT || WIDE) ? kInfoInvSplitWaves : kInfoInvWaves;        // 8 (split: 247 registers; wide tiles: LDS) / 12
    static constexpr bool PROFILE = false;
    static constexpr bool PROD = !WIDE;                           // has a production (DBG = false) instantiation of the split kernel
    static constexpr bool REC12 = false;
    static constexpr bool VLDS = true;
    static constexpr bool VIEW_FOLD = false;
    static constexpr bool STAGED = false;
    static constexpr int STAGE_FLOATS = 0;
    static constexpr int VFEAT_FLOATS = (WIDE ? kWave : 16) * kViewFeat;     // view inputs of the tile's rays (render_common picks the policy)
    static constexpr int NSTEP = 1;
    static constexpr int BATCH = kBatch16;
    static constexpr int RING = 128;

    // called by all 64 lanes; returns sigma of the lane's own sample (0 when !valid)
    __device__ static __forceinline__ float sigma(const RenderArgs &A, const float *smem, bool valid, const float x[3], int lane,
                                                  float t[6])
    {
        if constexpr (SPLIT) return infoinv_sigma_bf16(A, per_pass(smem) + RGB_FLOATS, valid, x, lane, t);
        using D = InfoInvDensLayout;
        const float *img = per_pass(smem) + RGB_FLOATS;
        asm volatile("" : "+v"(lane));       // as in mlp_pass16_ii: lane-derived offsets are cheaper to recompute than to keep for the whole kernel
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
            const Tex tx = karg_tex(offsetof(RenderArgs, dens) + p * sizeof(Tex));
            float feat[24];
            if (valid) {
                Bil b = bil_setup(t[2 * p], t[2 * p + 1], tx);
                const f32x4 *q00 = tex_at<f32x4>(tx.p, (uint32_t)b.idx * 24u);       // 32-bit byte offsets from a scalar base (ngf_device.hpp tex_at)
                const f32x4 *q01 = tex_at<f32x4>(tx.p, (uint32_t)(b.idx + tx.stride) * 24u);
#pragma unroll
                for (int part = 0; part < NGF_II_TAP_PARTS; ++part) {      // 24 / PARTS channels at a time: 96 / PARTS registers of taps in flight (the fp32 kernel
                                                            // runs at the 168 registers of twelve waves per CU; all 24 at once was what it spilled for)
                    constexpr int NQ = 6 / NGF_II_TAP_PARTS;
                    f32x4 v00[NQ], v10[NQ], v01[NQ], v11[NQ];
#pragma unroll
                    for (int j = 0; j < NQ; ++j) { const int q = NQ * part + j; v00[j] = q00[q]; v10[j] = q00[6 + q]; v01[j] = q01[q]; v11[j] = q01[6 + q]; }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < NQ; ++j)
#pragma unroll
                        for (int e = 0; e < 4; ++e) feat[4 * (NQ * part + j) + e] = bil_mix(b, v00[j][e], v10[j][e], v01[j][e], v11[j][e]);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (A.mode) {
#pragma unroll
                    for (int c = 0; c < 24; ++c) feat[c] = feat[c] * pe[c];
                }
            } else {
#pragma unroll
                for (int c = 0; c < 24; ++c) feat[c] = 0.0f;
            }
#ifndef NGF_EXP_II_DENS_INTERLEAVED
            // all twelve operand exchanges first, then the plane's 24 matrix instructions in a row (LDS reads of the weights between them are no vector
            // instructions): written `swap, 2 MFMAs, swap, 2 MFMAs ...` the SIMD switched between its vector and its matrix pipe 52 times per 64-sample
            // pass, and a switch costs ~38 cycles.  Same operations on the same values in the same accumulation order.
#pragma unroll
            for (int j = 0; j < 12; ++j) swap32(feat[2 * j], feat[2 * j + 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                const float w = d1[(p * 12 + j) * 64];
                h0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, feat[2 * j], h0, 0, 0, 0);
                h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, feat[2 * j + 1], h1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#else
#pragma unroll
            for (int j = 0; j < 12; ++j) {
                float a = feat[2 * j], b = feat[2 * j + 1];
                swap32(a, b);
                const float w = d1[(p * 12 + j) * 64];
                h0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, a, h0, 0, 0, 0);
                h1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, b, h1, 0, 0, 0);
            }
#endif
        }
#ifndef NGF_EXP_II_DENS_INTERLEAVED
        asm volatile("" : "+v"(h0), "+v"(h1) :: "memory");         // (see below; "memory": the bias reads of layer 2 and their address arithmetic stay behind layer 1's MFMAs too)
#endif
        f32x16 g0, g1;
#pragma unroll
        for (int r = 0; r < 16; ++r) g0[r] = g1[r] = img[D::B2 + hi * 16 + r];
        const float *d2 = img + D::D2 + lane;
#ifndef NGF_EXP_II_DENS_INTERLEAVED
        // (the accumulators behind an opaque asm: sched_barrier orders instructions with side effects; a ReLU is a pure value the instruction selector
        // is free to emit in front of the barrier -- it put tile 1's sixteen among tile 0's last MFMAs)
#pragma unroll
        for (int k = 0; k < 16; ++k) { h0[k] = relu1(h0[k]); h1[k] = relu1(h1[k]); }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float w = d2[k * 64];
            g0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, h0[k], g0, 0, 0, 0);
            g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, h1[k], g1, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#else
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float w = d2[k * 64];
            g0 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, relu1(h0[k]), g0, 0, 0, 0);
            g1 = __builtin_amdgcn_mfma_f32_32x32x2f32(w, relu1(h1[k]), g1, 0, 0, 0);
        }
#endif
        float s0 = 0.0f, s1 = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float w = img[D::W3 + hi * 16 + r];
            s0 = fmaf(w, relu1(g0[r]), s0);
            s1 = fmaf(w, relu1(g1[r]), s1);
        }
        s0 = s0 + __shfl_xor(s0, 32);
        s1 = s1 + __shfl_xor(s1, 32);
        const float f = (hi ? s1 : s0) + img[D::B3];
        return valid ? softplus_shift(f) : 0.0f;
    }
    __device__ static __forceinline__ void shade(const RenderArgs &A, const float *smem, const float rec[kRecFloats], const float *vf,
                                                 const float *, int lane, float c[3], unsigned long long * = nullptr)
    {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(vf + (lane >> 4) * 4);
        if constexpr (SPLIT) mlp_pass16_bf16_ii(A, smem, rec, v, lane, A.mode, c);
        else mlp_pass16_ii(A, smem, rec, v, lane, A.mode, c);
    }
};
using InfoInvPolicy = InfoInvPolicyT<false>;
using InfoInvWidePolicy = InfoInvPolicyT<false, true>;
using InfoInvSplitPolicy = InfoInvPolicyT<true>;

}  // namespace ngf
