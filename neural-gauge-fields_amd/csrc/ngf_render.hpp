// ngf_render.hpp -- the fused TriPlane ray-march kernel for gfx950 (MI355X).
//
// One persistent workgroup per CU; its waves are independent workers (no workgroup barrier after
// the MLP image has been staged into LDS).  A wave repeatedly takes a tile of 64 consecutive rays
// (one ray per lane) from a global counter and, per tile, alternates between two wave-uniform
// phases:
//
//   march  one sample step for all 64 rays: position, in-box / alpha-mask test, three gauge
//          fetches, three density fetches, sigma, alpha, transmittance, weight
//          (FieldBase.py:118-137, 251-288; Field.py:53-91).  Lanes whose weight exceeds the
//          threshold append a 32-byte record to the wave's LDS queue (ballot + prefix popcount),
//          so the queue is ordered by (step, lane).
//   shade  when >= 16 records are queued (or the march has ended): the colour path for 16
//          samples (Field.py:93-105, networks.py:25-32).  Four lanes share a sample: lane (s, kq)
//          gathers a quarter of the channels of the four bilinear taps of each colour plane and is
//          column s / k-slice kq of the B operand of v_mfma_f32_16x16x4_f32 (ngf_shade16.hpp; InfoInv:
//          ngf_infoinv.hpp; the bf16 forms: ngf_shade_bf16.hpp); the A operand is the pre-composed
//          layer-1 matrix read from LDS.  Layer 2 consumes the accumulators in place, layer 3 (64->3)
//          is a 16-term VALU dot per lane + two cross-quarter adds.  The weighted colours go to a
//          16-entry LDS result list and every lane, as ray owner, adds the entries that carry its
//          lane id in queue order -- the per-ray sum over samples is sequential in the sample index,
//          deterministic, and uses no atomics.
//
// Nothing of size [n,S,*] is ever materialised in HBM.
#pragma once
#include <cstddef>
#include <type_traits>

#include "ngf_device.hpp"
#include "ngf_shade16.hpp"
#include "ngf_shade_bf16.hpp"

namespace ngf {

// (karg_tex / karg / NGF_KARG_CONTRACT: ngf_device.hpp)

// LDS carve (floats): [blob | per wave: ring of RING records, result list, view inputs of the 64 rays]
template <typename P> constexpr int wave_lds_floats() { return P::RING * kRecFloats + P::BATCH * 4 + P::VFEAT_FLOATS + P::STAGE_FLOATS; }

// The 16 view-direction inputs of rgb_decoder layer 1 (networks.py:27-29, 205-216):
//   u[F..F+14] = [d(3), sin(d_x), sin(2 d_x), sin(d_y), sin(2 d_y), sin(d_z), sin(2 d_z), cos(same 6)], u[F+15] = 0 (pad)
// lane-half hi supplies entries hi*8 .. hi*8+7.  Computed once per ray per tile into LDS.
__device__ __forceinline__ void view_inputs(const float d[3], float v[16])
{
    v[0] = d[0]; v[1] = d[1]; v[2] = d[2];
#pragma unroll
    for (int k = 0; k < 3; ++k) {        // sincos_small (|argument| <= 2: <= 1.6 ulp), the same function view_inputs_tile uses: every tile shape sees the same bits
        sincos_small(d[k], v[3 + 2 * k], v[9 + 2 * k]);
        sincos_small(d[k] * 2.0f, v[4 + 2 * k], v[10 + 2 * k]);
    }
    v[15] = 0.0f;
}

// The same 16 inputs for the rays of a split tile, written to the tile's LDS table by the K = 64 / tile_w lanes that share a ray
// (lane (seg, ray)): the six sine / cosine arguments [d_x, 2 d_x, d_y, 2 d_y, d_z, 2 d_z] are dealt out over the segments -- one
// sincos_small per lane for K >= 6 (|argument| <= 2: <= 1.6 ulp) instead of twelve libm calls in every lane (12 x ~90 VALU instructions
// per tile, 5 % of the vector work of the headline frame) -- and segment 0 also stores d and the zero pad.
__device__ __forceinline__ void view_inputs_tile(const float d[3], float *dst, int seg, int K)
{
    if (seg == 0) { dst[0] = d[0]; dst[1] = d[1]; dst[2] = d[2]; dst[15] = 0.0f; }
    for (int a = seg; a < 6; a += K) {
        // the argument as a sum of three products with 0 / 1 / 2 (exact): a select over d[] by a runtime index turns d[] into a stack array
        const int dim = a >> 1;
        const float sc = (a & 1) ? 2.0f : 1.0f;
        const float arg = (dim == 0 ? sc : 0.0f) * d[0] + (dim == 1 ? sc : 0.0f) * d[1] + (dim == 2 ? sc : 0.0f) * d[2];
        float sn, cs;
        sincos_small(arg, sn, cs);
        dst[3 + a] = sn;
        dst[9 + a] = cs;
    }
}

// Copy the MLP image into LDS (every workgroup, once): 16-byte loads, four per thread requested before the first is stored.  The plain
// `smem[i] = blob[i]` loop compiled to load -> s_waitcnt vmcnt(0) -> ds_write per dword: 20 serial round trips to the L2 for the 58.6 KB TriPlane
// image (3.3-6.3 us of every launch, profiles/r04_timeline.txt) -- nothing for a frame, 2 % of a 4096-ray chunk.  The caller's barrier follows.
__device__ __forceinline__ void stage_blob(float *smem, const float *blob, int blob_floats)
{
    const f32x4 *src = reinterpret_cast<const f32x4 *>(blob);
    f32x4 *dst = reinterpret_cast<f32x4 *>(smem);
    const int n4 = blob_floats >> 2, nt = (int)blockDim.x;
    for (int i = threadIdx.x; i < n4; i += 4 * nt) {
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int j = i + k * nt; v[k] = src[j < n4 ? j : n4 - 1]; }
#pragma unroll
        for (int k = 0; k < 4; ++k) { const int j = i + k * nt; if (j < n4) dst[j] = v[k]; }
    }
    for (int i = (n4 << 2) + threadIdx.x; i < blob_floats; i += nt) smem[i] = blob[i];
}

// The MLP image in LDS is read-only after the initial barrier, so LICM would hoist every per-lane weight /
// bias read (160+ values) out of the persistent loops and pin them in VGPRs for the whole kernel (they
// then spill).  Adding an opaque zero to the pointer once per pass keeps those reads inside the pass.
__device__ __forceinline__ const float *per_pass(const float *blob)
{
    int z = 0;
    asm volatile("" : "+v"(z));
    return blob + z;
}

// sin / cos of x * 2^f for f = 0..F-1 (the InfoInv positional encodings, networks.py:227-237): an accurate sincos (sincos_small, <= 1.6 ulp) at every
// fourth octave and three angle doublings from each (sin 2a = 2 sc, cos 2a = (c - s)(c + s)); a doubling at most doubles the
// absolute error, so every value stays within ~8 ulp of 1 (5e-7) while 2F sinf/cosf calls with large-argument range reduction
// become F/4 sincos evaluations.
template <int F>
__device__ __forceinline__ void pe_ladder(float x, float sn[F], float cs[F])
{
#pragma unroll
    for (int base = 0; base < F; base += 4) {
        float s, c;
        sincos_small(x * (float)(1 << base), s, c);
        sn[base] = s; cs[base] = c;
#pragma unroll
        for (int j = 1; j < 4 && base + j < F; ++j) {
            const float s2 = 2.0f * s * c, c2 = (c - s) * (c + s);
            s = s2; c = c2;
            sn[base + j] = s; cs[base + j] = c;
        }
    }
}

// ---- TriPlane density at the gauge-shifted coordinates ------------------------------------------
// cells: if not null, the three bilinear cells of the fetch (the colour fetch of an active sample uses the same ones: same coordinates, plane of the
// same size -- they travel in the 12-float queue record, REC12)
template <bool BAKED>
__device__ __forceinline__ float triplane_density_feature(const RenderArgs &A, const float t[6], Bil *cells = nullptr)
{
    float f = 0.0f;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const Tex tx = karg_tex(offsetof(RenderArgs, dens) + p * sizeof(Tex));
        Bil b = bil_setup<BAKED>(t[2 * p], t[2 * p + 1], tx);
        if (cells) cells[p] = b;
        if (BAKED) {
            const float *q = tex_at<float>(tx.p, (uint32_t)b.idx), *q1 = tex_at<float>(tx.p, (uint32_t)(b.idx + tx.stride));
            f += bil_mix(b, q[0], q[1], q1[0], q1[1]);
        } else {
            const f32x4 *q00 = tex_at<f32x4>(tx.p, (uint32_t)b.idx * 16u);
            const f32x4 *q01 = tex_at<f32x4>(tx.p, (uint32_t)(b.idx + tx.stride) * 16u);
            float d00 = 0.0f, d10 = 0.0f, d01 = 0.0f, d11 = 0.0f;
            // the plane's 16 decoder weights are re-read from the kernel-argument segment here (one s_load_dwordx16 on the scalar
            // memory pipe): kept live across the march loop the 48 of them crowd the SGPR file and hipcc parks texture pointers in
            // VGPR lanes, ~36 v_readlane per march step on the (binding) vector pipe
            typedef const __attribute__((address_space(4))) float *kptr_t;
            kptr_t wdp = (kptr_t)((const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(RenderArgs, wd)) + p * 16;
            asm volatile("" : "+s"(wdp));
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v00 = q00[q], v10 = q00[4 + q], v01 = q01[q], v11 = q01[4 + q];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float w = wdp[4 * q + e];
                    d00 = fmaf(w, v00[e], d00);
                    d10 = fmaf(w, v10[e], d10);
                    d01 = fmaf(w, v01[e], d01);
                    d11 = fmaf(w, v11[e], d11);
                }
            }
            f += bil_mix(b, d00, d10, d01, d11);
        }
    }
    return f + A.bd;
}

// compute_gauge (Field.py:53-75): three 2-channel bilinear fetches + the reference's add order
template <bool MED3 = true>
__device__ __forceinline__ void triplane_gauge(const RenderArgs &A, const float x[3], int gauge_on, float t[6])
{
    const float u[3] = {x[0], x[1], x[0]}, v[3] = {x[1], x[2], x[2]};   // xy, yz, xz
    if (gauge_on) {
        float d[3][2];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const Tex tx = karg_tex(offsetof(RenderArgs, gau) + p * sizeof(Tex));
            Bil b = bil_setup<MED3>(u[p], v[p], tx);
            const f32x2 *g = tex_at<f32x2>(tx.p, (uint32_t)b.idx * 2u), *g1 = tex_at<f32x2>(tx.p, (uint32_t)(b.idx + tx.stride) * 2u);
            f32x2 g00 = g[0], g10 = g[1], g01 = g1[0], g11 = g1[1];
            d[p][0] = bil_mix(b, g00[0], g10[0], g01[0], g11[0]);
            d[p][1] = bil_mix(b, g00[1], g10[1], g01[1], g11[1]);
        }
        // d[0] = dxy, d[1] = dyz, d[2] = dxz
        t[0] = (u[0] + d[0][0]) + d[2][0];  t[1] = (v[0] + d[0][1]) + d[1][0];
        t[2] = (u[1] + d[1][0]) + d[0][1];  t[3] = (v[1] + d[1][1]) + d[2][1];
        t[4] = (u[2] + d[2][0]) + d[0][0];  t[5] = (v[2] + d[2][1]) + d[1][1];
    } else {
#pragma unroll
        for (int p = 0; p < 3; ++p) { t[2 * p] = u[p]; t[2 * p + 1] = v[p]; }
    }
}

// Field policy of the fused kernel: TriPlane (learned gauge + Linear(48,1) density on the VALU).
//   BAKE_D  density_decoder pre-composed into 1-channel planes (NGF_F_BAKE_DENSITY)
//   BAKE_C  rgb_decoder layer 1 pre-composed into 64-channel planes (NGF_F_BAKE_COLOR)
//   WAVES   waves per workgroup (= per CU); NSTEP march steps evaluated per loop iteration
//   The shade runs on v_mfma_f32_16x16x4_f32, 16 samples per pass (ngf_shade16.hpp).
// sigma() is called by ALL lanes of the wave (wave-uniform call site); invalid lanes return 0.
template <bool BAKE_D, bool BAKE_C, int WAVES_, int NSTEP_, bool PROFILE_ = false>
struct TriPlanePolicy {
    static constexpr bool PROFILE = PROFILE_;
    static constexpr bool PROD = WAVES_ == 12 && NSTEP_ == 1 && !PROFILE_;      // has a production (DBG = false) instantiation of the split kernel
    static constexpr bool INFOINV = false;
    static constexpr bool MASK_SKIP = false;                    // empty-space skipping through the mask's block image (march loop): MaskSkip<P> below
    static constexpr bool STAGED = false;                       // ngf_stage.hpp: LDS-staged texture strips
    static constexpr int STAGE_FLOATS = 0;
    static constexpr int WAVES = WAVES_;
    static constexpr int NSTEP = NSTEP_;
    static constexpr int BATCH = kBatch16;
    static constexpr bool VLDS = WAVES_ <= 12;                  // per-ray view inputs cached in LDS (4 KB / wave) or recomputed per pass
    static constexpr int VFEAT_FLOATS = VLDS ? kWave * kViewFeat : 0;       // view inputs of up to 64 rays (+ the per-ray fold table of small tiles)
    static constexpr int RING = NSTEP_ == 1 ? 128 : 256;        // >= BATCH-1 + 64*NSTEP records
    // REC12 (split kernel of the default twelve-wave policies): a queue record carries the sample's three bilinear cells (texel index, fractional
    // parts, in-range bit) instead of its six coordinates, so a shade pass starts its gathers at once and spends 8 instead of ~28 instructions per
    // plane on the cell -- all 64 lanes of a pass used to redo the three setups of their 16 samples that the march had already done.
    static constexpr bool REC12 = WAVES_ == 12 && NSTEP_ == 1 && !PROFILE_;
    static constexpr bool GATHER_QUAD = BAKE_C;                 // the shade's gather lane 4 s + kq works for sample lane >> 2 (ngf_shade16.hpp mlp_pass16_baked): shade12 takes ITS cells
    __device__ static __forceinline__ float sigma(const RenderArgs &A, const float *, bool valid, const float x[3], int, float t[6], Bil *cells = nullptr)
    {
        // branch-free: out-of-box samples have out-of-range coordinates, for which bil_setup clamps the
        // texel index and zeroes the weights, so their gathers are safe and their result is discarded.
        // Without the branch the NSTEP independent steps share one basic block and their gathers overlap.
        float tt[6];
        triplane_gauge<BAKE_D>(A, x, A.mode, tt);
        const float sg = softplus_shift(triplane_density_feature<BAKE_D>(A, tt, cells));
#pragma unroll
        for (int k = 0; k < 6; ++k) t[k] = valid ? tt[k] : 0.0f;
        return valid ? sg : 0.0f;
    }
    // vf: the owner ray's 16 cached view inputs (VLDS) or nullptr; od: the owner ray's direction
    static constexpr bool VIEW_FOLD = VLDS;                      // small split tiles: b1 + W1[:, view].view once per ray per tile
    __device__ static __forceinline__ void fold_view(const float *smem, const float *vfeat, float *pre, int n_rays, int lane)
    {
        if constexpr (BAKE_C) view_fold16<MlpLayout16Baked::W1V / 64, MlpLayout16Baked::B1, 4>(per_pass16(smem), vfeat, pre, n_rays, lane);
        else view_fold16<MlpLayout16<48>::W1 / 64 + 3 * MlpLayout16<48>::QCH, MlpLayout16<48>::B1, MlpLayout16<48>::KT>(per_pass16(smem), vfeat, pre, n_rays, lane);
    }
    // the same from view inputs held in registers (lane (s, kq): entries kq*4 .. kq*4+3 of ray s) -- ngf_render_pc.hpp
    __device__ static __forceinline__ void fold_view_regs(const float *smem, const f32x4 v, float *pre, int n_rays, int lane)
    {
        if constexpr (BAKE_C) view_fold16_regs<MlpLayout16Baked::W1V / 64, MlpLayout16Baked::B1, 4>(per_pass16(smem), v, pre, n_rays, lane);
        else view_fold16_regs<MlpLayout16<48>::W1 / 64 + 3 * MlpLayout16<48>::QCH, MlpLayout16<48>::B1, MlpLayout16<48>::KT>(per_pass16(smem), v, pre, n_rays, lane);
    }
    __device__ static __forceinline__ void shade(const RenderArgs &A, const float *smem, const float rec[kRecFloats], const float *vf,
                                                 const float od[3], int lane, float c[3], unsigned long long *tk = nullptr, const float *pre = nullptr)
    {
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (!pre) v = VLDS ? *reinterpret_cast<const f32x4 *>(vf + (lane >> 4) * 4) : view_entries16(od, lane >> 4);
        if constexpr (BAKE_C) mlp_pass16_baked(A, smem, rec, v, lane, c, pre);
        else mlp_pass16<48>(A, smem, rec, v, lane, c, tk, pre);
    }
    // REC12 form: the cells come from the record
    __device__ static __forceinline__ void shade12(const RenderArgs &A, const float *smem, const RecCells &cells, const float *vf, int lane, float c[3],
                                                   const float *pre)
    {
        const float rec[kRecFloats] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (!pre) v = *reinterpret_cast<const f32x4 *>(vf + (lane >> 4) * 4);
        if constexpr (BAKE_C) mlp_pass16_baked(A, smem, rec, v, lane, c, pre, &cells);
        else mlp_pass16<48>(A, smem, rec, v, lane, c, nullptr, pre, &cells);
    }
};

// The same kernel with the march's empty-space skipping compiled in (round 6): launched for fields that carry an alpha mask.  A second instantiation
// rather than a run-time branch in one: the level-3 kernel sits at its 168-register budget, and with the skip's branch in its march loop the frame
// WITHOUT a mask measured 0.2-0.4 % slower (profiles/r06_mask_skip.txt) -- the headline launch keeps the code it had.
template <typename P>
struct MaskSkip : P {
#ifdef NGF_EXP_NO_MASK_SKIP
    static constexpr bool MASK_SKIP = false;                    // A/B builds (profiles/r06_mask_skip.txt)
#else
    static constexpr bool MASK_SKIP = true;
#endif
};

// NGF_F_SPLIT_BF16: the colour MLP on the bf16 matrix pipe with 3-term split operands (ngf_shade_bf16.hpp).  Split tiles of <= 8 rays
// only (the wave keeps the view inputs of 8 rays); 8 waves per CU (the pass needs ~200 registers).
template <bool BAKE_D, int WAVES_ = 8>
struct TriPlaneBf16Policy : TriPlanePolicy<BAKE_D, false, WAVES_, 1> {
    static constexpr bool PROD = WAVES_ == 8;
    static constexpr bool REC12 = false;
    static constexpr bool VIEW_FOLD = false;
    static constexpr int VFEAT_FLOATS = 8 * kViewFeat;
    __device__ static __forceinline__ void shade(const RenderArgs &A, const float *smem, const float rec[kRecFloats], const float *vf,
                                                 const float od[3], int lane, float c[3], unsigned long long * = nullptr, const float * = nullptr)
    {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(vf + (lane >> 4) * 4);
        if constexpr (WAVES_ > 8) mlp_pass16_bf16_rows(A, smem, rec, v, lane, c);       // the 168-register form (12 waves per CU)
        else mlp_pass16_bf16(A, smem, rec, v, lane, c);
    }
};

// NGF_F_BAKE_DENSITY | NGF_F_BAKE_COLOR | NGF_F_SPLIT_BF16 (round 5, opt-in): level 3 with layer 2 on the bf16 matrix pipe (ngf_shade_bf16.hpp
// mlp_pass16_baked_bf16); everything else -- march, queue records, tile plan, twelve waves per CU -- is the level-3 policy's.
struct TriPlaneBakedBf16Policy : TriPlanePolicy<true, true, 12, 1> {
    using LB = MlpLayout16BakedBf16;
    __device__ static __forceinline__ void fold_view(const float *smem, const float *vfeat, float *pre, int n_rays, int lane)
    {
        view_fold16<LB::W1V / 64, LB::B1, 4>(per_pass16(smem), vfeat, pre, n_rays, lane);
    }
    __device__ static __forceinline__ void fold_view_regs(const float *smem, const f32x4 v, float *pre, int n_rays, int lane)
    {
        view_fold16_regs<LB::W1V / 64, LB::B1, 4>(per_pass16(smem), v, pre, n_rays, lane);
    }
    __device__ static __forceinline__ void shade(const RenderArgs &A, const float *smem, const float rec[kRecFloats], const float *vf,
                                                 const float od[3], int lane, float c[3], unsigned long long * = nullptr, const float *pre = nullptr)
    {
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (!pre) v = *reinterpret_cast<const f32x4 *>(vf + (lane >> 4) * 4);
        mlp_pass16_baked_bf16(A, smem, rec, v, lane, c, pre);
    }
    __device__ static __forceinline__ void shade12(const RenderArgs &A, const float *smem, const RecCells &cells, const float *vf, int lane, float c[3],
                                                   const float *pre)
    {
        const float rec[kRecFloats] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
        if (!pre) v = *reinterpret_cast<const f32x4 *>(vf + (lane >> 4) * 4);
        mlp_pass16_baked_bf16(A, smem, rec, v, lane, c, pre, &cells);
    }
};

// NGF_F_NO_FOLD (level 0): un-composed rgb_decoder, view inputs per sample; 8 waves per CU (the basis stage keeps 36 more accumulators)
struct TriPlaneNoFoldPolicy : TriPlanePolicy<false, false, 8, 1> {
    static constexpr bool PROD = false;
    static constexpr bool REC12 = false;
    static constexpr bool VIEW_FOLD = false;
    __device__ static __forceinline__ void shade(const RenderArgs &A, const float *smem, const float rec[kRecFloats], const float *vf,
                                                 const float od[3], int lane, float c[3], unsigned long long * = nullptr, const float * = nullptr)
    {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(vf + (lane >> 4) * 4);
        mlp_pass16_nofold(A, smem, rec, v, lane, c);
    }
};

// ---- raw2alpha of a split tile (FieldBase.py:12-19): the sequential cumprod / running sums of a ray, chained lane to lane with DPP ------------
// Lane layout of a split tile (tile_w = 4 M rays, M = 1, 2, 4, 8): every 16-lane ROW holds M rays, lane-in-row = seg * M + r, so the K = 16 / M
// lanes of a ray (its K consecutive steps) are M lanes apart inside one row and "the previous step's lane" is DPP row_shr:M -- a lane whose source
// would be outside the row (seg 0) is disabled by bound_ctrl:0 and KEEPS its value.  The chain  P[seg] = P[seg-1] * f[seg]  (P = transmittance after
// the lane's step) is therefore ONE v_mul_f32_dpp per round, in place: seg 0 starts from the ray's incoming T and is final at once, round r
// finalises seg r.  The same for acc and depth with v_add_f32_dpp.  Products and sums are taken in step order with the operations of the unsplit
// march (w = alpha T, T *= (1 - alpha) + 1e-10, acc += w, dep += w z): the results are its bits for every tile shape.  Rounds 1-3 used K-1 rounds of
// eight VALU instructions + three ds_bpermute each (lane - tile_w through the LDS crossbar): 165 issue slots and 15 LDS round trips per march
// iteration of a 4-ray tile, now ~55 VALU slots and none.  (Inline assembly: the in-place form -- destination = DPP source, disabled lanes keep the
// old value -- is not what hipcc makes of __builtin_amdgcn_update_dpp + a multiply; the s_nop are the VALU-write -> DPP-read wait states, which the
// hazard recogniser does not insert inside inline assembly.)
// In: T, acc, dep valid in the seg-0 lane of each ray.  Out: w of the lane's own step; T, acc, dep after the tile's K steps, again in the seg-0 lane
// (row_ror:M brings seg K-1 -- lane-in-row 16 - M + r -- to lane r); the other lanes hold values of no meaning.
#define NGF_STR2(x) #x
#define NGF_STR(x) NGF_STR2(x)
#define NGF_SPLIT_CHAIN(M)                                                                                                                        \
    template <> __device__ __forceinline__ void split_chain<M>(float alpha, float z, float &T, float &acc, float &dep, float &w)                    \
    {                                                                                                                                               \
        constexpr int K = 16 / M;                                                                                                                   \
        const float f = (1.0f - alpha) + 1e-10f;                                                                                                    \
        float Pt = T * f;                                                                                                                           \
        asm volatile("s_nop 4" ::: "memory");      /* an EXEC write right before the first DPP instruction needs 5 wait states */                   \
        _Pragma("unroll") for (int r = 1; r < K; ++r)                                                                                               \
            asm volatile("s_nop 1\n\tv_mul_f32_dpp %0, %0, %1 row_shr:" NGF_STR(M) " row_mask:0xf bank_mask:0xf" : "+v"(Pt) : "v"(f));              \
        float Tin = T;                             /* seg 0 keeps the incoming T, seg s takes P of seg s - 1 */                                      \
        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shr:" NGF_STR(M) " row_mask:0xf bank_mask:0xf" : "+v"(Tin) : "v"(Pt));                    \
        w = alpha * Tin;                                                                                                                            \
        const float wz = w * z;                                                                                                                     \
        float Pa = acc + w, Pd = dep + wz;                                                                                                          \
        asm volatile("s_nop 0" : "+v"(Pa), "+v"(Pd));   /* with the s_nop 0 below: 2 wait states between the VALU writes above and the first DPP read, whichever hipcc emits last */ \
        _Pragma("unroll") for (int r = 1; r < K; ++r)                                                                                               \
            asm volatile("s_nop 0\n\tv_add_f32_dpp %0, %0, %2 row_shr:" NGF_STR(M) " row_mask:0xf bank_mask:0xf\n\t"                                \
                         "v_add_f32_dpp %1, %1, %3 row_shr:" NGF_STR(M) " row_mask:0xf bank_mask:0xf" : "+v"(Pa), "+v"(Pd) : "v"(w), "v"(wz));      \
        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %3 row_ror:" NGF_STR(M) " row_mask:0xf bank_mask:0xf\n\t"                                        \
                     "v_mov_b32_dpp %1, %4 row_ror:" NGF_STR(M) " row_mask:0xf bank_mask:0xf\n\t"                                                   \
                     "v_mov_b32_dpp %2, %5 row_ror:" NGF_STR(M) " row_mask:0xf bank_mask:0xf" : "=&v"(T), "=&v"(acc), "=&v"(dep) : "v"(Pt), "v"(Pa), "v"(Pd)); \
    }
template <int M> __device__ __forceinline__ void split_chain(float alpha, float z, float &T, float &acc, float &dep, float &w);
NGF_SPLIT_CHAIN(1)
NGF_SPLIT_CHAIN(2)
NGF_SPLIT_CHAIN(4)
NGF_SPLIT_CHAIN(8)
#undef NGF_SPLIT_CHAIN

// Tiles of 2 or 1 rays (the tail of a launch's tile plan): a ray takes R = 2 or 4 whole rows (K = 32 / 64 consecutive steps).  The in-row chain is
// the one above with M = 1; the carry from a row's lane 15 into the next row's lane 0 is DPP row_bcast:15 (lane 15 of every row -> all lanes of
// the next row, written only into the rows whose turn it is: row_mask).  Row j of a ray is final after j + 1 runs of 15 rounds -- still one
// multiply / add per step in step order.  The rounds run in all rows at once; a row that is already final recomputes its own values, a row whose
// turn has not come holds values of no meaning until its carry arrives.
#define NGF_ROW_ROUNDS15(OP, ACC, X) \
    _Pragma("unroll") for (int r = 1; r < 16; ++r) asm volatile("s_nop 1\n\t" OP " %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(ACC) : "v"(X))
#define NGF_ROW_CARRY(OP, ACC, X, MASK) \
    asm volatile("s_nop 1\n\t" OP " %0, %0, %1 row_bcast:15 row_mask:" MASK " bank_mask:0xf" : "+v"(ACC) : "v"(X))
template <int R> __device__ __forceinline__ void split_chain_rows(float alpha, float z, float &T, float &acc, float &dep, float &w, int last)
{
    static_assert(R == 2 || R == 4, "a ray takes 2 or 4 rows");
    const float f = (1.0f - alpha) + 1e-10f;
    float Pt = T * f;                              // right in the ray's first lane; every other lane is rewritten by the rounds
    asm volatile("s_nop 4" ::: "memory");
    NGF_ROW_ROUNDS15("v_mul_f32_dpp", Pt, f);
    if constexpr (R == 2) { NGF_ROW_CARRY("v_mul_f32_dpp", Pt, f, "0xa"); NGF_ROW_ROUNDS15("v_mul_f32_dpp", Pt, f); }
    else {
        NGF_ROW_CARRY("v_mul_f32_dpp", Pt, f, "0x2"); NGF_ROW_ROUNDS15("v_mul_f32_dpp", Pt, f);
        NGF_ROW_CARRY("v_mul_f32_dpp", Pt, f, "0x4"); NGF_ROW_ROUNDS15("v_mul_f32_dpp", Pt, f);
        NGF_ROW_CARRY("v_mul_f32_dpp", Pt, f, "0x8"); NGF_ROW_ROUNDS15("v_mul_f32_dpp", Pt, f);
    }
    // T before the lane's own step: the ray's first lane keeps the incoming T, lane 0 of a later row takes lane 15 of the row before, the others their left neighbour
    float Tin = T;
    if constexpr (R == 2) asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf" : "+v"(Tin) : "v"(Pt));
    else asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_bcast:15 row_mask:0xe bank_mask:0xf" : "+v"(Tin) : "v"(Pt));
    asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(Tin) : "v"(Pt));
    w = alpha * Tin;
    const float wz = w * z;
    float Pa = acc + w, Pd = dep + wz;
    asm volatile("s_nop 4" ::: "memory");
    NGF_ROW_ROUNDS15("v_add_f32_dpp", Pa, w); NGF_ROW_ROUNDS15("v_add_f32_dpp", Pd, wz);
    if constexpr (R == 2) {
        NGF_ROW_CARRY("v_add_f32_dpp", Pa, w, "0xa"); NGF_ROW_CARRY("v_add_f32_dpp", Pd, wz, "0xa");
        NGF_ROW_ROUNDS15("v_add_f32_dpp", Pa, w); NGF_ROW_ROUNDS15("v_add_f32_dpp", Pd, wz);
    } else {
        NGF_ROW_CARRY("v_add_f32_dpp", Pa, w, "0x2"); NGF_ROW_CARRY("v_add_f32_dpp", Pd, wz, "0x2");
        NGF_ROW_ROUNDS15("v_add_f32_dpp", Pa, w); NGF_ROW_ROUNDS15("v_add_f32_dpp", Pd, wz);
        NGF_ROW_CARRY("v_add_f32_dpp", Pa, w, "0x4"); NGF_ROW_CARRY("v_add_f32_dpp", Pd, wz, "0x4");
        NGF_ROW_ROUNDS15("v_add_f32_dpp", Pa, w); NGF_ROW_ROUNDS15("v_add_f32_dpp", Pd, wz);
        NGF_ROW_CARRY("v_add_f32_dpp", Pa, w, "0x8"); NGF_ROW_CARRY("v_add_f32_dpp", Pd, wz, "0x8");
        NGF_ROW_ROUNDS15("v_add_f32_dpp", Pa, w); NGF_ROW_ROUNDS15("v_add_f32_dpp", Pd, wz);
    }
    asm volatile("s_nop 1" ::: "memory");
    // the ray's last lane holds the sums after its K steps: back to (all lanes of) the ray, the first lane is the one that counts
    T = __shfl(Pt, last); acc = __shfl(Pa, last); dep = __shfl(Pd, last);
}
#undef NGF_ROW_ROUNDS15
#undef NGF_ROW_CARRY

// ---- the fused kernel ---------------------------------------------------------------------------
// SPLIT = false: one ray per lane, tile_w (<= 64) rays per tile.  SPLIT = true (the default of every render): a tile holds
// tile_w = 64 >> k rays and every ray is marched by K = 64 / tile_w lanes that take K CONSECUTIVE steps per iteration; transmittance,
// acc and depth are then chained lane to lane in step order (split_chain: DPP row shifts), so each ray sees exactly the sequential
// arithmetic of the unsplit march (results are bit-identical) while a tile's critical path is K times shorter and all 64 lanes gather.
// DBG = false is the production instantiation: the per-sample debug outputs (dbg_weight / dbg_sigma), the ablation bits, skip_rgb and
// the statistics counters are compiled OUT (they cost scalar registers -- the kernel parks SGPRs in VGPR lanes -- and issue slots of a
// kernel that is bound by its vector pipe); launch_render picks DBG = true whenever one of them is requested.
template <typename P, bool SPLIT = false, bool DBG = true>
__global__ void __launch_bounds__(P::WAVES * 64) render_kernel(const RenderArgs A)
{
    static_assert(DBG || !P::PROFILE, "the section profile is a debug instantiation");
    NGF_KARG_CONTRACT((&render_kernel<P, SPLIT, DBG>));
    if constexpr (DBG) {       // karg_tex reads offset 0 of the kernel-argument segment: trap if a future kernel passes its RenderArgs elsewhere
        if (karg_tex(offsetof(RenderArgs, dens)).p != A.dens[0].p) __builtin_trap();
    }
    static_assert(!SPLIT || P::NSTEP == 1, "the split march is written for one step per lane per iteration");
    extern __shared__ __attribute__((aligned(16))) float smem[];
#ifdef NGF_EXP_TIMELINE     // experiment build (profiles/exp_timeline.py): per-wave wall-clock marks of the production kernel -> A.stats[16 + 8 * wave id ..]
    const unsigned long long tl_t0 = wall_clock64();
    unsigned long long tl_tiles = 0, tl_pass = 0, tl_iter = 0, tl_first = 0;
    unsigned long long tl_seg_t0 = 0, tl_seg_t1 = 0, tl_seg_t2 = 0, tl_seg_t3 = 0, tl_seg_n = 0, tl_tile_t = 0;      // round 6: wall clock per plan segment, tiles per segment (16 bits each)
    int tl_seg = 0;
#endif
    __shared__ unsigned wg_done;          // working waves of this workgroup that found the tile queue empty (queue_done)
    if (threadIdx.x == 0) wg_done = 0;
    stage_blob(smem, A.blob, A.blob_floats);
    __syncthreads();
#ifdef NGF_EXP_TIMELINE
    const unsigned long long tl_t1 = wall_clock64();
#endif
    // small launches: launch_render spreads the tiles over all CUs and lets only the first waves_active waves of a workgroup work (they are dealt
    // round robin to the four SIMDs): 1024 tiles are 4 waves on each of 256 CUs -- one per SIMD -- instead of 12 waves on 86 CUs
    if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) >= A.waves_active) return;

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    constexpr int NSTEP = P::NSTEP, BATCH = P::BATCH;
    // R12: 12-float records (three cells instead of six coordinates) in a ring of 80 -- a march iteration leaves at most 15 + 64 = 79 queued, the
    // ring index wraps by one compare-subtract instead of a mask.  The carve is SMALLER than the generic one of the policy (80 * 12 + 64 + 1024 =
    // 2048 floats against 2112), so the launch code's LDS size covers both kernels and every tile width keeps its view table.
    constexpr bool R12 = SPLIT && P::REC12;
    constexpr int RING = R12 ? 80 : P::RING, RECF = R12 ? 12 : kRecFloats;
    constexpr int WAVE_FLOATS = R12 ? RING * RECF + BATCH * 4 + P::VFEAT_FLOATS + P::STAGE_FLOATS : wave_lds_floats<P>();
    static_assert(WAVE_FLOATS <= wave_lds_floats<P>(), "the R12 carve must fit the policy's LDS allocation");
    auto wrap = [](int x) { return R12 ? (x >= RING ? x - RING : x) : (x & (RING - 1)); };      // ring index of head + (< RING)
    float *wl = smem + ((A.blob_floats + 3) & ~3) + wave * WAVE_FLOATS;
    float *ring = wl;
    float *res = wl + RING * RECF;
    float *vfeat = res + BATCH * 4;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const int S = A.S;
    unsigned long long st_valid = 0, st_active = 0, st_pass = 0, st_rays = 0;
    [[maybe_unused]] unsigned long long st_staged = 0;          // STAGED: march iterations served from LDS strips (stats[13])
    unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // PROFILE only: cycles per section, summed over the wave's life

    const int xcd = xcd_id();
    int q_first = 0;
    for (;;) {
        unsigned int tile = 0;
        if (!next_tile(A, xcd, lane, q_first, tile)) break;
        if constexpr (SPLIT && !P::INFOINV) {          // screen-space tile order (ngf_field_render_image): the queue position becomes a tile of the image-blocked walk.
            // (TriPlane only: the InfoInv frames do not move with it -- +-0.7 %, profiles/r06_r2_locality.txt -- and its fp32 kernel sits at its register
            // budget: the four scalar divisions of the map cost it 12 B of scratch)
            const uint32_t ord_n = NGF_KARG(ord_n);
            if (tile < ord_n) tile = tile_order(tile, ord_n, NGF_KARG(ord_tpr), NGF_KARG(ord_bw), NGF_KARG(ord_bh));
        }
#ifdef NGF_EXP_TIMELINE
        if (!tl_tiles) tl_first = wall_clock64();
        ++tl_tiles;
#endif
        // the tile's place in the launch's plan (RenderArgs::seg_*): which segment, hence how wide, and its first ray (scalar selects -- indexing
        // the kernel-argument arrays with a run-time index would move them to scratch)
        int ts = NGF_KARG_AT(seg_shift, 0);
        int64_t base = NGF_KARG_AT(seg_ray0, 0) + ((int64_t)tile << ts);
        if constexpr (SPLIT) {
            const uint32_t e0 = NGF_KARG_AT(seg_end, 0), e1 = NGF_KARG_AT(seg_end, 1), e2 = NGF_KARG_AT(seg_end, 2);
            if (tile >= e0) { ts = NGF_KARG_AT(seg_shift, 1); base = NGF_KARG_AT(seg_ray0, 1) + ((int64_t)(tile - e0) << ts); }
            if (tile >= e1) { ts = NGF_KARG_AT(seg_shift, 2); base = NGF_KARG_AT(seg_ray0, 2) + ((int64_t)(tile - e1) << ts); }
            if (tile >= e2) { ts = NGF_KARG_AT(seg_shift, 3); base = NGF_KARG_AT(seg_ray0, 3) + ((int64_t)(tile - e2) << ts); }
        }
#ifdef NGF_EXP_TIMELINE
        tl_tile_t = wall_clock64();
        tl_seg = 0;
        if constexpr (SPLIT) tl_seg = (tile >= NGF_KARG_AT(seg_end, 0)) + (tile >= NGF_KARG_AT(seg_end, 1)) + (tile >= NGF_KARG_AT(seg_end, 2));
#endif
        const int tile_w = 1 << ts;
        // split tile of >= 4 rays: a 16-lane row holds M = tile_w / 4 rays, lane-in-row = seg * M + r (split_chain above), ray slot = row * M + r;
        // of 2 / 1 rays: a ray takes R = 2 / 4 whole rows, seg = its lane index inside them (split_chain_rows)
        // (per-tile copies of the lane id and of two uniform scalars behind an opaque asm: hipcc hoists the few cheap VALU results derived from
        // them -- the lane's row, the clamp's canonicalised `far`, step * (S + 1) -- out of the persistent tile loop and then SPILLS them in the
        // kernels that run at their register budget (InfoInv fp32: 3 of its 4 scratch dwords); recomputed per tile they cost three instructions)
        int lane_t = lane, S_t = S;
        float far_t = NGF_KARG(far_), jit_t_ = 0.0f;
        asm volatile("" : "+v"(lane_t), "+s"(S_t), "+s"(far_t));
        const int mshift = SPLIT ? (ts >= 2 ? ts - 2 : 0) : 0;           // log2(M)
        const int rshift = SPLIT ? (ts >= 2 ? 0 : 2 - ts) : 0;           // log2(R)
        const int K = SPLIT ? (64 >> ts) : 1;
        const int seg = SPLIT ? (ts >= 2 ? ((lane_t & 15) >> mshift) : (lane_t & (K - 1))) : 0;              // which of the K consecutive steps this lane takes
        const int rl = SPLIT ? (ts >= 2 ? (((lane_t >> 4) << mshift) | (lane_t & ((1 << mshift) - 1))) : (lane_t >> (4 + rshift))) : lane_t;      // ray slot inside the tile (= owner id in the queues)
        const int64_t ray = base + rl;
        const int64_t n_rays = NGF_KARG(n);
        const bool live = (rl < tile_w) && (ray < n_rays);
        const int64_t rr = live ? ray : n_rays - 1;
        float o[3], d[3];
        {
            const float *rays_p = NGF_KARG(rays), *jit_p = NGF_KARG(jitter);
#pragma unroll
            for (int k = 0; k < 3; ++k) { o[k] = rays_p[rr * 6 + k]; d[k] = rays_p[rr * 6 + 3 + k]; }
            jit_t_ = jit_p ? jit_p[rr] : 0.0f;
        }
        const float jit = jit_t_;

        // sample_ray (FieldBase.py:122-125)
        float tmin = -INFINITY;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float vec = (d[k] == 0.0f) ? 1e-6f : d[k];
            float ra = (A.a1[k] - o[k]) / vec, rb = (A.a0[k] - o[k]) / vec;
            tmin = fmaxf(tmin, fminf(ra, rb));
        }
        tmin = fminf(fmaxf(tmin, NGF_KARG(near_)), far_t);

        if constexpr (P::VLDS) {   // view inputs of this lane's ray (networks.py:27-29), read back by the shade lanes
            if constexpr (SPLIT) {
                view_inputs_tile(d, vfeat + rl * kViewFeat, seg, K);            // the K lanes of a ray share the work
            } else {
                float v[16];
                view_inputs(d, v);
                f32x4 *dst = reinterpret_cast<f32x4 *>(vfeat + rl * kViewFeat);
#pragma unroll
                for (int q = 0; q < 4; ++q) dst[q] = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            }
        }
        // tiles of <= 8 rays: fold the view part of layer 1 into one 64-float vector per ray (behind the 8 x 16 view inputs)
        bool view_fold = false;
        if constexpr (SPLIT && P::VIEW_FOLD) {
            view_fold = tile_w <= 8;
            if (view_fold) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                P::fold_view(smem, vfeat, vfeat + 8 * kViewFeat, tile_w, lane);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
        float T = 1.0f, acc = 0.0f, dep = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f;
        int i = 0, head = 0, count = 0;
        const float zmax = tmin + A.step * (float)(S_t + 1);      // bound of every z of the ray (exact early termination below)
        const bool dbg_samples = DBG && A.dbg_weight;           // per-sample outputs requested (never in the production instantiation)
        for (;;) {
            [[maybe_unused]] unsigned long long t_sec = 0;
            if constexpr (P::PROFILE) t_sec = __builtin_readcyclecounter();
            if (count < BATCH && i < S) {
#ifdef NGF_EXP_TIMELINE
                ++tl_iter;
#endif
                // ---------------- march NSTEP steps (independent gathers, sequential transmittance) ----
                float z[NSTEP], dist[NSTEP], sigma[NSTEP], t[NSTEP][6];
                [[maybe_unused]] Bil cells[NSTEP][3];
                bool empty_step = false;
#pragma unroll
                for (int u = 0; u < NSTEP; ++u) {
                    const int si = i + u + seg;
                    z[u] = tmin + A.step * ((float)si + jit);
                    const float zn = tmin + A.step * ((float)(si + 1) + jit);
                    dist[u] = (si < S - 1) ? (zn - z[u]) : 0.0f;
                    float p[3], x[3];
                    bool valid = live && (si < S);
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        p[k] = o[k] + d[k] * z[u];
                        valid = valid && !(A.a0[k] > p[k] || p[k] > A.a1[k]);
                    }
#ifndef NGF_EXP_NO_ALPHA_MASK          // timing experiment: the march without the alpha-mask test compiled in (what a no-mask instantiation would be)
                    if (A.mask.bits && valid) valid = mask_occupied(A.mask, p);
#endif
#pragma unroll
                    for (int k = 0; k < 6; ++k) t[u][k] = 0.0f;
#pragma unroll
                    for (int k = 0; k < 3; ++k) x[k] = (p[k] - A.a0[k]) * A.inv[k] - 1.0f;   // normalize_coord (FieldBase.py:88-89)
                    if constexpr (NSTEP == 1) {
                        // no lane of the wave has a valid sample in this iteration (outside the box / in empty space of the alpha
                        // mask): sigma = alpha = w = 0 for all of them, T / acc / depth keep their values -> skip the whole step
                        if (!dbg_samples && !(DBG && (A.ablate & 64)) && !__any(valid)) { empty_step = true; break; }
                    }
                    if constexpr (SPLIT && P::STAGED) {
                        float *dscr = P::STAGE_FLOATS > 0 ? vfeat + P::VFEAT_FLOATS : nullptr;
                        sigma[u] = P::sigma_staged(A, vfeat, dscr, valid, x, lane, t[u], (DBG && A.stats) ? &st_staged : nullptr);
                    } else if constexpr (R12) {
                        sigma[u] = P::sigma(A, smem, valid, x, lane, t[u], cells[u]);
                    } else {
                        sigma[u] = P::sigma(A, smem, valid, x, lane, t[u]);
                    }
                    if constexpr (DBG) st_valid += __popcll(__ballot(valid));
                }
                if (empty_step) {
                    // Empty-space skipping (round 6).  The iteration held no valid sample; before the next one is evaluated, lane (ray, seg) looks at the
                    // step i + K + m + seg (2m + 1) of its ray: when the mask's block image says that nothing is occupied within 8 cells of that sample's
                    // cell, the 2m + 1 steps around it sample empty cells too -- m steps move a cell index by at most m r + 1 <= 8 (<= 4 for the finer image), r = the cells per step
                    // of the wave's fastest ray -- so a leading run of n passing lanes on EVERY ray certifies n (2m + 1) steps and the tile jumps over the
                    // whole iterations in them.  Certified samples have sigma = alpha = w = 0: T, acc, depth and the queue are what they would have been.
                    // The same test runs the march out of the box (blocks beyond the volume are empty).  DBG: ablate bit 128 switches it off.
                    int skip = 0;
                    if constexpr (NSTEP == 1 && P::MASK_SKIP) {
                        if (A.mask.coarse && !(DBG && (A.ablate & 128))) {
                            // (the mask's descriptor re-read from the kernel-argument segment HERE: what hipcc derives from it -- the clamps' float bounds -- it otherwise
                            // hoists to the kernel's entry and spills in the kernels that sit at their register budget: 52 B per lane at level 3)
                            const MaskVol mk = karg_mask(offsetof(RenderArgs, mask));
                            const float sz[3] = {(float)(mk.W - 1), (float)(mk.H - 1), (float)(mk.D - 1)};
                            float r = 0.0f;
#pragma unroll
                            for (int k = 0; k < 3; ++k) r = fmaxf(r, fabsf(d[k]) * (A.step * 0.5f * mk.inv[k] * sz[k]));
                            const bool sane = !live || (r == r && r < 1e30f);          // a NaN / infinite direction certifies nothing
                            r = live && sane ? r : 0.0f;
#pragma unroll
                            for (int off = 32; off >= 1; off >>= 1) r = fmaxf(r, __shfl_xor(r, off));
                            // whole iterations behind this one that the block image of 2^LOG-cell blocks certifies as empty
                            auto certified = [&](auto logc, const uint8_t *image) -> int {
                                constexpr int LOG = decltype(logc)::value;
                                const int m = (int)fminf(((float)(1 << LOG) - 1.1f) / fmaxf(r, 1e-6f), 2048.0f), gsteps = 2 * m + 1;
                                const int stest = i + K + m + seg * gsteps;
                                const float zt = tmin + A.step * ((float)stest + jit);
                                float pt[3];
#pragma unroll
                                for (int k = 0; k < 3; ++k) pt[k] = o[k] + d[k] * zt;
                                const bool pass = !live || (sane && (stest - m >= S || mask_clear_around<LOG>(mk, image, pt)));
                                const unsigned long long fm = __ballot(!pass);
                                int nrun = K;
                                if (fm) {
                                    if (!SPLIT) nrun = 0;
                                    else if (ts >= 2) nrun = __builtin_ctz((unsigned)((fm | (fm >> 16) | (fm >> 32) | (fm >> 48)) & 0xffffu)) >> mshift;
                                    else if (ts == 1) nrun = __builtin_ctz((unsigned)(fm | (fm >> 32)));
                                    else nrun = __builtin_ctzll(fm);
                                }
                                return (nrun * gsteps) / K;
                            };
                            skip = certified(std::integral_constant<int, 3>{}, mk.coarse);
                            if (skip == 0 && mk.fine) skip = certified(std::integral_constant<int, 2>{}, mk.fine);      // nothing in reach of the 8-cell blocks: the 4-cell ones (clutter, the approach to a surface)
                        }
                    }
                    i += NSTEP * K * (1 + skip);
                    if constexpr (P::PROFILE) prof[0] += __builtin_readcyclecounter() - t_sec;
                    continue;
                }
#pragma unroll
                for (int u = 0; u < NSTEP; ++u) {
                    // raw2alpha (FieldBase.py:12-19); steps past S have sigma = 0 -> alpha = w = 0, T unchanged
                    const float alpha = 1.0f - expf(-sigma[u] * (dist[u] * A.dscale));
                    float w;
                    if constexpr (SPLIT) {
                        // the K lanes of a ray chain T, acc and depth in step order (split_chain): the sequential arithmetic of the unsplit march
                        switch (ts) {
                        case 0: split_chain_rows<4>(alpha, z[u], T, acc, dep, w, 63); break;
                        case 1: split_chain_rows<2>(alpha, z[u], T, acc, dep, w, lane | 31); break;
                        case 2: split_chain<1>(alpha, z[u], T, acc, dep, w); break;
                        case 3: split_chain<2>(alpha, z[u], T, acc, dep, w); break;
                        case 4: split_chain<4>(alpha, z[u], T, acc, dep, w); break;
                        case 5: split_chain<8>(alpha, z[u], T, acc, dep, w); break;
                        default: w = alpha * T; T = T * ((1.0f - alpha) + 1e-10f); acc += w; dep += w * z[u]; break;      // tile_w = 64 through the split kernel (knob): one lane per ray
                        }
                    } else {
                        w = alpha * T;
                        T = T * ((1.0f - alpha) + 1e-10f);
                        acc += w;
                        dep += w * z[u];
                    }
                    if (dbg_samples && live && i + u + seg < S) {
                        A.dbg_weight[ray * S + i + u + seg] = w;
                        A.dbg_sigma[ray * S + i + u + seg] = sigma[u];
                    }
                    const bool active = (w > A.thr) && !(DBG && A.skip_rgb);
                    const unsigned long long m = __ballot(active);
                    if (active) {
                        const int slot = wrap(head + wrap(count + __popcll(m & lt_mask)));
                        f32x4 *r = reinterpret_cast<f32x4 *>(ring + slot * RECF);
                        if constexpr (R12) {
                            const Bil *c = cells[u];
                            r[0] = f32x4{__int_as_float(rl | (c[0].in << 8) | (c[1].in << 9) | (c[2].in << 10)), w, __int_as_float(c[0].idx), __int_as_float(c[1].idx)};
                            r[1] = f32x4{__int_as_float(c[2].idx), c[0].wx1, c[0].wy1, c[1].wx1};
                            r[2] = f32x4{c[1].wy1, c[2].wx1, c[2].wy1, 0.0f};
                        } else {
                            r[0] = f32x4{__int_as_float(rl), w, t[u][0], t[u][1]};
                            r[1] = f32x4{t[u][2], t[u][3], t[u][4], t[u][5]};
                        }
                    }
                    count += __popcll(m);
                    if constexpr (DBG) st_active += __popcll(m);
                }
                i += NSTEP * K;
                // Exact early termination.  Every later sample has w <= T.  Once T < thr no later sample can be active (no colour
                // work), and once T < 2^-26 min(acc, dep / z_max) every later `acc += w` and `dep += w z` is less than half an ulp
                // of its accumulator, i.e. a no-op in float32: the outputs cannot change by a single bit, so the rest of the ray is
                // skipped -- wave-uniformly, when all rays of the tile are there (8-ray tiles of adjacent pixels).  Rays behind an
                // opaque surface stop right after it.  Off for the per-sample debug outputs and with NGF_ABLATE=32.
                if (!dbg_samples && !(DBG && (A.ablate & 32))) {
                    const bool done = !live || seg != 0 || ((T < A.thr) & (zmax > 0.0f) & (T < 0x1p-26f * fminf(acc, dep / zmax)));      // T / acc / dep live in a ray's seg-0 lane
                    if (!__any(!done)) i = S;
                }
                if constexpr (P::PROFILE) prof[0] += __builtin_readcyclecounter() - t_sec;
            } else if (count > 0) {
                // ---------------- shade up to BATCH queued samples ----------------------------------
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                const int nb = count < BATCH ? count : BATCH;
                const int s = lane & (BATCH - 1);
                const int slot = wrap(head + (s < nb ? s : 0));
                const f32x4 *r = reinterpret_cast<const f32x4 *>(ring + slot * RECF);
                const f32x4 r0 = r[0], r1 = r[1];
                const float rec[kRecFloats] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
                const int owner = R12 ? (__float_as_int(r0[0]) & 0xff) : __float_as_int(r0[0]);
                // the lane that owns the ray's sums (its seg-0 lane): row owner / M, lane-in-row owner % M
                const int owner_lane = SPLIT ? (ts >= 2 ? (((owner >> mshift) << 4) | (owner & ((1 << mshift) - 1))) : (owner << (4 + rshift))) : owner;
                float od[3] = {0.0f, 0.0f, 0.0f};
                if constexpr (!P::VLDS) { od[0] = __shfl(d[0], owner_lane); od[1] = __shfl(d[1], owner_lane); od[2] = __shfl(d[2], owner_lane); }
                float c[3];
                [[maybe_unused]] const float *pre = view_fold ? vfeat + 8 * kViewFeat + owner * kFoldStride : nullptr;
                if constexpr (P::PROFILE) {
                    unsigned long long tk[5] = {0, 0, 0, 0, 0};
                    P::shade(A, smem, rec, vfeat + owner * kViewFeat, od, lane, c, tk, pre);
                    prof[1] += tk[0] - t_sec;      // ring read, address setup, gather 0 + view MFMAs issued
                    prof[2] += tk[1] - tk[0];      // wait for plane 0 + interpolate
                    prof[3] += tk[2] - tk[1];      // layer-1 MFMAs (planes 1, 2 gathered behind them)
                    prof[4] += tk[4] - tk[2];      // layer 2: 64 MFMAs issued
                    prof[6] += tk[3] - tk[4];      // layer 3 (VALU dot, 2 cross-lane adds, sigmoid)
                    t_sec = tk[3];
                } else if constexpr (R12) {
                    // the cells of the sample the lane GATHERS for: its own (lane & 15), or -- baked colour planes, quad-coalesced gather -- sample lane >> 2
                    f32x4 q0 = r0, q1 = r1, q2 = r[2];
                    if constexpr (P::GATHER_QUAD) {
                        const int sgq = lane >> 2;
                        const f32x4 *rq = reinterpret_cast<const f32x4 *>(ring + wrap(head + (sgq < nb ? sgq : 0)) * RECF);
                        q0 = rq[0]; q1 = rq[1]; q2 = rq[2];
                    }
                    const RecCells cl = {{__float_as_int(q0[2]), __float_as_int(q0[3]), __float_as_int(q1[0])}, {q1[1], q1[3], q2[1]}, {q1[2], q2[0], q2[2]}, __float_as_int(q0[0])};
                    P::shade12(A, smem, cl, vfeat + owner * kViewFeat, lane, c, pre);
                } else {
                    if constexpr (P::VIEW_FOLD) P::shade(A, smem, rec, vfeat + owner * kViewFeat, od, lane, c, nullptr, pre);
                    else P::shade(A, smem, rec, vfeat + owner * kViewFeat, od, lane, c);
                }
                // result list, structure-of-arrays: res[0..B) owner lane ids, then weighted r, g, b
                // c[] holds the three logits in all four lanes of a sample: lane quarter kq applies the sigmoid to channel kq and writes that entry of
                // the list (res[(1 + kq) * BATCH + s] is res[BATCH + lane]) -- one sigmoid per lane instead of three
                if (lane < BATCH) res[lane] = lane < nb ? __int_as_float(owner_lane) : __int_as_float(-1);
                if (lane < 3 * BATCH) {
                    const int kq_ = lane >> 4;
                    const float logit = kq_ == 0 ? c[0] : (kq_ == 1 ? c[1] : c[2]);
                    res[BATCH + lane] = r0[1] * sigmoid_rcp(logit);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                // every lane, as ray owner, adds its entries in queue (= sample) order; all reads are issued up front
                // (broadcast ds_read_b128), no per-entry LDS round trip
                // (an entry of another ray adds m * v = +0: the weighted colours are finite and the sums non-negative, so `x + 0` is
                // `x` to the bit -- one select + three FMAs per entry instead of three selects + three adds)
                {
#pragma unroll
                    for (int q = 0; q < BATCH / 4; ++q) {
                        const f32x4 id = *reinterpret_cast<const f32x4 *>(res + 4 * q);
                        const f32x4 vr = *reinterpret_cast<const f32x4 *>(res + BATCH + 4 * q);
                        const f32x4 vg = *reinterpret_cast<const f32x4 *>(res + 2 * BATCH + 4 * q);
                        const f32x4 vb = *reinterpret_cast<const f32x4 *>(res + 3 * BATCH + 4 * q);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
#ifndef NGF_COLLECT_SELECT
                            // the entry's owner lane alone adds it: v_cmpx narrows EXEC to that lane, three v_add_f32 run under it, EXEC comes back
                            // (4 VALU instructions per entry; written in C++ -- `if (id == lane) cr += v` -- hipcc if-converts it into three adds +
                            // three selects, and the select + FMA form below is a compare, a select, a packed FMA, an FMA and two moves).  Same sums:
                            // `cr + v` in the owner lane, untouched elsewhere (the FMA form added 0 * v = +0 there).
                            unsigned long long save;
                            asm volatile("s_mov_b64 %3, exec\n\t"
                                         "v_cmpx_eq_u32_e32 vcc, %4, %5\n\t"
                                         "v_add_f32_e32 %0, %0, %6\n\t"
                                         "v_add_f32_e32 %1, %1, %7\n\t"
                                         "v_add_f32_e32 %2, %2, %8\n\t"
                                         "s_mov_b64 exec, %3"
                                         : "+v"(cr), "+v"(cg), "+v"(cb), "=&s"(save)
                                         : "v"(id[e]), "v"(lane), "v"(vr[e]), "v"(vg[e]), "v"(vb[e])
                                         : "vcc");
#else
                            const float m = __float_as_int(id[e]) == lane ? 1.0f : 0.0f;
                            cr = fmaf(m, vr[e], cr);
                            cg = fmaf(m, vg[e], cg);
                            cb = fmaf(m, vb[e], cb);
#endif
                        }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                head = wrap(head + nb);
                count -= nb;
#ifdef NGF_EXP_TIMELINE
                ++tl_pass;
#endif
                if constexpr (DBG) ++st_pass;
                if constexpr (P::PROFILE) prof[5] += __builtin_readcyclecounter() - t_sec;     // result list + owner collect
            } else {
                break;
            }
        }
        if (live && seg == 0) {
            // compositing tail (FieldBase.py:296-306)
            float out[3] = {cr, cg, cb};
            float *rgb_p = NGF_KARG(rgb), *depth_p = NGF_KARG(depth);
            const int32_t white = NGF_KARG(white_bg);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                float v = out[c];
                if (white) v = v + (1.0f - acc);
                rgb_p[ray * 3 + c] = fminf(fmaxf(v, 0.0f), 1.0f);
            }
            depth_p[ray] = dep + (1.0f - acc) * d[2];
        }
        if constexpr (DBG) st_rays += __popcll(__ballot(live && seg == 0));
#ifdef NGF_EXP_TIMELINE
        {
            const unsigned long long dt_tile = wall_clock64() - tl_tile_t;
            tl_seg_t0 += tl_seg == 0 ? dt_tile : 0; tl_seg_t1 += tl_seg == 1 ? dt_tile : 0; tl_seg_t2 += tl_seg == 2 ? dt_tile : 0; tl_seg_t3 += tl_seg == 3 ? dt_tile : 0;
            tl_seg_n += 1ull << (16 * tl_seg);
        }
#endif
    }
    queue_done(A.tile_counter, &wg_done, (unsigned)A.waves_active, A.queue_waves, lane);
#ifdef NGF_EXP_TIMELINE
    if (!DBG && A.stats && lane == 0) {
        unsigned long long *row = A.stats + 16 + 16 * ((size_t)blockIdx.x * P::WAVES + wave);
        row[0] = tl_t0; row[1] = tl_t1; row[2] = tl_first; row[3] = wall_clock64(); row[4] = tl_tiles; row[5] = tl_pass; row[6] = tl_iter; row[7] = (unsigned long long)xcd;
        row[8] = tl_seg_t0; row[9] = tl_seg_t1; row[10] = tl_seg_t2; row[11] = tl_seg_t3; row[12] = tl_seg_n;
    }
#endif
    if (DBG && A.stats && lane == 0) {
        atomicAdd(A.stats + 0, st_valid);
        atomicAdd(A.stats + 1, st_active);
        atomicAdd(A.stats + 2, st_pass);
        atomicAdd(A.stats + 3, st_rays);
        if constexpr (P::PROFILE) {
            for (int k = 0; k < 7; ++k) atomicAdd(A.stats + 4 + k, prof[k]);
        }
        if constexpr (P::STAGED) atomicAdd(A.stats + 13, st_staged);
    }
}

// ---- compute_alpha for explicit world-space points (FieldBase.py:140-159) or on the lattice of getDenseAlpha (:161-178) ------------
// Lattice form (xyz == NULL): the three 1-D linspace(0,1,g) vectors come from the host (torch's own values); point (ix,iy,iz) is
// aabb0*(1-s) + aabb1*s per axis in float32 like the reference, and the result is written in the TRANSPOSED [gz,gy,gx] order that
// updateAlphaMask works in (FieldBase.py:184-185) -- no [g^3,3] coordinate tensor exists.
struct Lattice {
    const float *sx, *sy, *sz;
    int32_t gx, gy, gz;
};

template <typename P>
__global__ void __launch_bounds__(256) alpha_kernel(const RenderArgs A, const float *xyz, const Lattice L, int64_t n, float length, float *alpha)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    NGF_KARG_CONTRACT((&alpha_kernel<P>));
    if (karg_tex(offsetof(RenderArgs, dens)).p != A.dens[0].p) __builtin_trap();      // the RenderArgs must be the first kernel argument (karg_tex)
    if constexpr (P::INFOINV) {
        stage_blob(smem, A.blob, A.blob_floats);
        __syncthreads();
    }
    const int lane = threadIdx.x & 63;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); base < n; base += stride) {
        const int64_t i = base + lane;
        const int64_t ii = i < n ? i : n - 1;
        float p[3];
        if (xyz) {
            p[0] = xyz[ii * 3]; p[1] = xyz[ii * 3 + 1]; p[2] = xyz[ii * 3 + 2];
        } else {
            const int ix = (int)(ii % L.gx), iy = (int)((ii / L.gx) % L.gy), iz = (int)(ii / ((int64_t)L.gx * L.gy));
            const float s[3] = {L.sx[ix], L.sy[iy], L.sz[iz]};
#pragma unroll
            for (int k = 0; k < 3; ++k) p[k] = A.a0[k] * (1.0f - s[k]) + A.a1[k] * s[k];
        }
        bool valid = i < n;
        if (A.mask.bits && valid) valid = mask_occupied(A.mask, p);
        float x[3], t[6];
#pragma unroll
        for (int k = 0; k < 3; ++k) x[k] = (p[k] - A.a0[k]) * A.inv[k] - 1.0f;
        const float sigma = P::sigma(A, smem, valid, x, lane, t);
        if (i < n) alpha[i] = 1.0f - expf(-sigma * length);
    }
}

// updateAlphaMask (FieldBase.py:180-216) after getDenseAlpha: clamp(0,1), 3x3x3 max-pool (stride 1, padding 1), threshold to a
// {0,1} float volume [gz,gy,gx], and the index bounding box + count of the occupied voxels (integer atomics: deterministic)
__global__ void __launch_bounds__(256) mask_pool_kernel(const float *__restrict__ alpha, int gx, int gy, int gz, float thres, float *__restrict__ vol,
                                                        int *bounds, unsigned long long *count)
{
    const int64_t n = (int64_t)gx * gy * gz;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
    unsigned long long cnt = 0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int x = (int)(i % gx), y = (int)((i / gx) % gy), z = (int)(i / ((int64_t)gx * gy));
        float m = -INFINITY;
        for (int dz = -1; dz <= 1; ++dz) {
            const int zz = z + dz;
            if (zz < 0 || zz >= gz) continue;
            for (int dy = -1; dy <= 1; ++dy) {
                const int yy = y + dy;
                if (yy < 0 || yy >= gy) continue;
                const float *row = alpha + ((int64_t)zz * gy + yy) * gx;
#pragma unroll
                for (int dx = -1; dx <= 1; ++dx) {
                    const int xx = x + dx;
                    if (xx < 0 || xx >= gx) continue;
                    m = fmaxf(m, fminf(fmaxf(row[xx], 0.0f), 1.0f));
                }
            }
        }
        const bool occ = m >= thres;
        vol[i] = occ ? 1.0f : 0.0f;
        if (occ) {
            ++cnt;
            lo[0] = min(lo[0], x); lo[1] = min(lo[1], y); lo[2] = min(lo[2], z);
            hi[0] = max(hi[0], x); hi[1] = max(hi[1], y); hi[2] = max(hi[2], z);
        }
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        if (hi[k] >= 0) { atomicMin(bounds + k, lo[k]); atomicMax(bounds + 3 + k, hi[k]); }
    }
    if (cnt) atomicAdd(count, cnt);
}

__global__ void mask_bounds_init_kernel(int *bounds)
{
    if (threadIdx.x < 3) bounds[threadIdx.x] = 0x7fffffff;
    else if (threadIdx.x < 6) bounds[threadIdx.x] = -1;
}

// valid_xyz.amin(0) / amax(0) (FieldBase.py:204-208): coordinates are monotone in their lattice index, so the box of the occupied
// voxels is the lattice point of the index bounds (taken per axis with min/max to stay correct for a flipped aabb)
__global__ void mask_aabb_kernel(const RenderArgs A, const Lattice L, const int *bounds, float *new_aabb)
{
    const int k = threadIdx.x;
    if (k >= 3) return;
    const float *s = k == 0 ? L.sx : (k == 1 ? L.sy : L.sz);
    const float a = A.a0[k] * (1.0f - s[bounds[k]]) + A.a1[k] * s[bounds[k]];
    const float b = A.a0[k] * (1.0f - s[bounds[3 + k]]) + A.a1[k] * s[bounds[3 + k]];
    new_aabb[k] = fminf(a, b);
    new_aabb[3 + k] = fmaxf(a, b);
}

// ---- filtering_rays (FieldBase.py:218-246): the alpha-mask branch (S > 0) or the bbox_only slab test (S <= 0); one ray per thread --------------------------------------
__global__ void __launch_bounds__(256) ray_filter_kernel(const RenderArgs A, const float *rays, int64_t n, int S, uint8_t *keep)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
        float o[3], d[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) { o[k] = rays[r * 6 + k]; d[k] = rays[r * 6 + 3 + k]; }
        float tmin = -INFINITY;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            float vec = (d[k] == 0.0f) ? 1e-6f : d[k];
            float ra = (A.a1[k] - o[k]) / vec, rb = (A.a0[k] - o[k]) / vec;
            tmin = fmaxf(tmin, fminf(ra, rb));
        }
        if (S <= 0) {
            // bbox_only (FieldBase.py:226-233): keep the ray iff t_max > t_min of the slab test
            float tmax = INFINITY;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float vec = (d[k] == 0.0f) ? 1e-6f : d[k];
                float ra = (A.a1[k] - o[k]) / vec, rb = (A.a0[k] - o[k]) / vec;
                tmax = fminf(tmax, fmaxf(ra, rb));
            }
            keep[r] = tmax > tmin ? 1 : 0;
            continue;
        }
        tmin = fminf(fmaxf(tmin, A.near_), A.far_);
        bool hit = false;
        for (int i = 0; i < S && !hit; ++i) {
            const float z = tmin + A.step * (float)i;
            float p[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) p[k] = o[k] + d[k] * z;
            hit = mask_occupied(A.mask, p);
        }
        keep[r] = hit ? 1 : 0;
    }
}

// ---- compute_rgb alone on caller-supplied samples (parity-test entry point) ----------------------
template <typename P>
__global__ void __launch_bounds__(256) decode_rgb_kernel(const RenderArgs A, const float *coords, const float *dirs,
                                                         int64_t n, float *out)
{
    constexpr int BATCH = P::BATCH;
    NGF_KARG_CONTRACT((&decode_rgb_kernel<P>));
    extern __shared__ __attribute__((aligned(16))) float smem[];
    stage_blob(smem, A.blob, A.blob_floats);
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float *vfeat = smem + ((A.blob_floats + 3) & ~3) + wave * (32 * kViewFeat);
    const int64_t nbatch = (n + BATCH - 1) / BATCH;
    for (int64_t bt = (int64_t)blockIdx.x * (blockDim.x >> 6) + wave; bt < nbatch; bt += (int64_t)gridDim.x * (blockDim.x >> 6)) {
        const int s = lane & (BATCH - 1);
        int64_t q = bt * BATCH + s;
        const bool ok = q < n;
        if (!ok) q = n - 1;
        float rec[kRecFloats] = {0.0f, 1.0f, coords[q * 6 + 0], coords[q * 6 + 1], coords[q * 6 + 2],
                                 coords[q * 6 + 3], coords[q * 6 + 4], coords[q * 6 + 5]};
        if (lane < BATCH) {
            const float dq[3] = {dirs[q * 3], dirs[q * 3 + 1], dirs[q * 3 + 2]};
            float v[16];
            view_inputs(dq, v);
            f32x4 *dst = reinterpret_cast<f32x4 *>(vfeat + s * kViewFeat);
#pragma unroll
            for (int k = 0; k < 4; ++k) dst[k] = f32x4{v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]};
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        float c[3];
        const float dq2[3] = {dirs[q * 3], dirs[q * 3 + 1], dirs[q * 3 + 2]};
        P::shade(A, smem, rec, vfeat + s * kViewFeat, dq2, lane, c);         // logits
        if (lane < BATCH && ok) { out[q * 3] = sigmoid_rcp(c[0]); out[q * 3 + 1] = sigmoid_rcp(c[1]); out[q * 3 + 2] = sigmoid_rcp(c[2]); }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

}  // namespace ngf
