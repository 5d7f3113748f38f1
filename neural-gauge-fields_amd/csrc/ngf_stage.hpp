// ngf_stage.hpp -- the LDS-staged texture variant of the TriPlane march (BASELINE.json north_star: "coalesced HBM reads of the
// three plane textures into LDS tiles"), built to be MEASURED next to the gather form (ngf_debug_set("stage", 1); results in
// profiles/r02_lds_staging.txt).
//
// A split tile is 8 adjacent rays x 8 consecutive steps: before the gauge shift its 64 samples cover ~2.4 x 4 texels per plane, so
// the texels they tap form a small axis-aligned strip.  Per march iteration the wave reduces the bounds of the samples' normalised
// coordinates (6 DPP reductions), derives each plane's cell rectangle from them (the texel coordinate is monotone in the
// normalised coordinate, so the rectangle contains every lane's cell), and, when the rectangle has at most CAP texels, loads every
// texel ONCE -- one texel per lane, consecutive lanes on consecutive texels of a row (coalesced) -- into a wave-private LDS strip;
// the four bilinear taps of every lane then come from LDS.  Rectangles that do not fit fall back to the per-lane gathers.
//   gauge on  (iteration >= gauge_start): the three 2-channel GAUGE strips are staged (the gauge-shifted density coordinates
//             scatter by the learned offsets, +-3.6 texels on the bench's noise gauge: their rectangle is ~150 texels, no strip);
//   gauge off (compute_alpha, iteration < gauge_start): the three 16-channel DENSITY strips are staged, one plane at a time.
// The arithmetic on the fetched values is that of triplane_gauge / triplane_density_feature in the same order: bit-identical.
#pragma once
#include "ngf_render.hpp"

namespace ngf {

constexpr int kStageGaugeCap = 64;          // texels per staged gauge strip (one per lane)
constexpr int kStageDensCap = 48;           // texels per staged density strip: 48 x 20 floats = 3840 B per wave
constexpr int kStageDensStride = 20;        // floats per staged density texel (16 + 4 pad: consecutive texels start 20 banks apart)
constexpr int kStageGaugeOff = 8 * kViewFeat + 8 * kFoldStride;      // gauge strips (3 x 64 x 2 floats) behind the view inputs and the fold table of an 8-ray tile

template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float stage_dpp(float old, float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// wave-wide minimum of v (all 64 lanes take part), returned as a wave-uniform value
__device__ __forceinline__ float wave_min(float v)
{
    v = fminf(v, stage_dpp<0xB1>(v, v));             // quad_perm [1,0,3,2]
    v = fminf(v, stage_dpp<0x4E>(v, v));             // quad_perm [2,3,0,1]
    v = fminf(v, stage_dpp<0x141>(v, v));            // row_half_mirror
    v = fminf(v, stage_dpp<0x140>(v, v));            // row_mirror: every lane of a row holds the row's minimum
    v = fminf(v, stage_dpp<0x142, 0xa>(v, v));       // row_bcast15 into rows 1 and 3
    v = fminf(v, stage_dpp<0x143, 0xc>(v, v));       // row_bcast31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

struct CellRect {                 // padded texel rectangle [x0, x0 + w) x [y0, y0 + h) covering the taps of every valid lane
    int x0, y0, w, h;
    bool fits;
};
// u in [ulo, uhi], v in [vlo, vhi] (normalised coordinates of the valid lanes) -> the rectangle of their taps; cap = strip capacity
__device__ __forceinline__ CellRect cell_rect(const Tex &t, float ulo, float uhi, float vlo, float vhi, int cap, bool any)
{
    // same expressions as bil_setup: px = ((u + 1) / 2) * fw, cell = clamp(floor(px), -1, fw) + 1 (padded index)
    const float fx0 = fminf(fmaxf(floorf(((ulo + 1.0f) / 2.0f) * t.fw), -1.0f), t.fw);
    const float fx1 = fminf(fmaxf(floorf(((uhi + 1.0f) / 2.0f) * t.fw), -1.0f), t.fw);
    const float fy0 = fminf(fmaxf(floorf(((vlo + 1.0f) / 2.0f) * t.fh), -1.0f), t.fh);
    const float fy1 = fminf(fmaxf(floorf(((vhi + 1.0f) / 2.0f) * t.fh), -1.0f), t.fh);
    CellRect r;
    r.x0 = (int)fx0 + 1; r.y0 = (int)fy0 + 1;
    r.w = (int)fx1 - (int)fx0 + 2; r.h = (int)fy1 - (int)fy0 + 2;
    r.fits = any && r.w > 0 && r.h > 0 && r.w * r.h <= cap;
    return r;
}

template <int WAVES_>
struct TriPlaneStagedPolicy : TriPlanePolicy<false, false, WAVES_, 1> {
    static constexpr bool PROD = false;
    static constexpr bool REC12 = false;
    static constexpr bool STAGED = true;
    static constexpr int STAGE_FLOATS = WAVES_ <= 8 ? kStageDensCap * kStageDensStride : 0;      // density strip (8 waves per CU only: LDS)
    static constexpr int VFEAT_FLOATS = kStageGaugeOff + 3 * kStageGaugeCap * 2;                  // view inputs + fold table + gauge strips (1056 floats)
    static_assert(VFEAT_FLOATS >= kWave * kViewFeat, "the unsplit march keeps the view inputs of 64 rays here");

    // vscr: the wave's view-input block (gauge strips at +kStageGaugeOff); dscr: the wave's density strip or nullptr
    __device__ static __forceinline__ float sigma_staged(const RenderArgs &A, float *vscr, float *dscr, bool valid, const float x[3], int lane,
                                                         float t[6], unsigned long long *n_staged)
    {
        // bounds of the valid lanes' normalised coordinates, per axis (wave-uniform)
        const bool any = __any(valid);
        float lo[3], hi[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            lo[k] = wave_min(valid ? x[k] : INFINITY);
            hi[k] = -wave_min(valid ? -x[k] : INFINITY);
        }
        const float u[3] = {x[0], x[1], x[0]}, v[3] = {x[1], x[2], x[2]};          // xy, yz, xz
        const int ua[3] = {0, 1, 0}, va[3] = {1, 2, 2};
        float tt[6];
        if (A.mode) {
            // ---- compute_gauge (Field.py:53-75) from staged 2-channel strips ----
            float d[3][2];
            f32x2 *gs = reinterpret_cast<f32x2 *>(vscr + kStageGaugeOff);
            CellRect rc[3];
            bool all_fit = true;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                rc[p] = cell_rect(A.gau[p], lo[ua[p]], hi[ua[p]], lo[va[p]], hi[va[p]], kStageGaugeCap, any);
                all_fit = all_fit && rc[p].fits;
            }
            if (all_fit) {               // wave-uniform
#pragma unroll
                for (int p = 0; p < 3; ++p) {       // one texel per lane, row-major over the rectangle: consecutive lanes read consecutive texels
                    const Tex &tx = A.gau[p];
                    const int n = rc[p].w * rc[p].h;
                    const int l = lane < n ? lane : 0;
                    const int ly = (l * ((65536 + rc[p].w - 1) / rc[p].w)) >> 16, lx = l - ly * rc[p].w;
                    gs[p * 64 + lane] = reinterpret_cast<const f32x2 *>(tx.p)[(rc[p].y0 + ly) * tx.stride + rc[p].x0 + lx];
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const Tex &tx = A.gau[p];
                    const Bil b = bil_setup(u[p], v[p], tx);
                    int j = (b.cy - rc[p].y0) * rc[p].w + (b.cx - rc[p].x0);
                    j = valid ? j : 0;                                               // lanes outside the box: any in-range slot (discarded)
                    const f32x2 *g = gs + p * 64 + j;
                    const f32x2 g00 = g[0], g10 = g[1], g01 = g[rc[p].w], g11 = g[rc[p].w + 1];
                    d[p][0] = bil_mix(b, g00[0], g10[0], g01[0], g11[0]);
                    d[p][1] = bil_mix(b, g00[1], g10[1], g01[1], g11[1]);
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                if (n_staged) *n_staged += 1;
                tt[0] = (u[0] + d[0][0]) + d[2][0];  tt[1] = (v[0] + d[0][1]) + d[1][0];
                tt[2] = (u[1] + d[1][0]) + d[0][1];  tt[3] = (v[1] + d[1][1]) + d[2][1];
                tt[4] = (u[2] + d[2][0]) + d[0][0];  tt[5] = (v[2] + d[2][1]) + d[1][1];
            } else {
                triplane_gauge(A, x, 1, tt);
            }
        } else {
#pragma unroll
            for (int p = 0; p < 3; ++p) { tt[2 * p] = u[p]; tt[2 * p + 1] = v[p]; }
        }

        // ---- compute_density (Field.py:77-91): 16-channel strips when the coordinates are the un-shifted ones ----
        float f = 0.0f;
        bool staged_density = false;
        if constexpr (STAGE_FLOATS > 0) {
            if (!A.mode && dscr) {
                CellRect rc[3];
                bool all_fit = true;
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    rc[p] = cell_rect(A.dens[p], lo[ua[p]], hi[ua[p]], lo[va[p]], hi[va[p]], kStageDensCap, any);
                    all_fit = all_fit && rc[p].fits;
                }
                if (all_fit) {
                    staged_density = true;
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        const Tex &tx = A.dens[p];
                        const int n = rc[p].w * rc[p].h;
                        const int l = lane < n ? lane : 0;
                        const int ly = (l * ((65536 + rc[p].w - 1) / rc[p].w)) >> 16, lx = l - ly * rc[p].w;
                        const f32x4 *src = reinterpret_cast<const f32x4 *>(tx.p + (size_t)((rc[p].y0 + ly) * tx.stride + rc[p].x0 + lx) * 16);
                        const f32x4 q0 = src[0], q1 = src[1], q2 = src[2], q3 = src[3];
                        if (lane < kStageDensCap) {
                            f32x4 *dst = reinterpret_cast<f32x4 *>(dscr + lane * kStageDensStride);
                            dst[0] = q0; dst[1] = q1; dst[2] = q2; dst[3] = q3;
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                        const Bil b = bil_setup(tt[2 * p], tt[2 * p + 1], tx);
                        int j = (b.cy - rc[p].y0) * rc[p].w + (b.cx - rc[p].x0);
                        j = valid ? j : 0;
                        const f32x4 *q00 = reinterpret_cast<const f32x4 *>(dscr + j * kStageDensStride);
                        const f32x4 *q10 = reinterpret_cast<const f32x4 *>(dscr + (j + 1) * kStageDensStride);
                        const f32x4 *q01 = reinterpret_cast<const f32x4 *>(dscr + (j + rc[p].w) * kStageDensStride);
                        const f32x4 *q11 = reinterpret_cast<const f32x4 *>(dscr + (j + rc[p].w + 1) * kStageDensStride);
                        float d00 = 0.0f, d10 = 0.0f, d01 = 0.0f, d11 = 0.0f;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 v00 = q00[q], v10 = q10[q], v01 = q01[q], v11 = q11[q];
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float w = A.wd[p * 16 + 4 * q + e];
                                d00 = fmaf(w, v00[e], d00);
                                d10 = fmaf(w, v10[e], d10);
                                d01 = fmaf(w, v01[e], d01);
                                d11 = fmaf(w, v11[e], d11);
                            }
                        }
                        f += bil_mix(b, d00, d10, d01, d11);
                        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                    }
                    f = f + A.bd;
                    if (n_staged) *n_staged += 1;
                }
            }
        }
        if (!staged_density) f = triplane_density_feature<false>(A, tt);
        const float sg = softplus_shift(f);
#pragma unroll
        for (int k = 0; k < 6; ++k) t[k] = valid ? tt[k] : 0.0f;
        return valid ? sg : 0.0f;
    }
};

}  // namespace ngf
