// ngf_train.hpp -- one TriPlane training step on the device.
//
// What the reference does per iteration (TriPlane/main.py:264-299): field(rays, is_train=True) (FieldBase.py:251-312),
// rgb MSE + 8e-5 * density_L1 (Field.py:149-152), autograd, torch.optim.Adam over the 8 groups of get_optparam_groups
// (Field.py:34-46), lr decay.  Here the step is a short sequence of kernels over HBM-resident buffers (288 GB: the
// per-sample activations of a 4096-ray batch are simply kept, 1.7 KB per active sample):
//
//   train_density_kernel       every (step, ray) pair in parallel: sample_ray + normalize + compute_gauge + the 48-feature
//                              density fetch -> xs = Linear(48,1) - 10 (pre-softplus); stores exp(-softplus(xs) dist) and softplus'(xs)
//                              (every per-sample transcendental of the step is taken here, 3.6 M samples in parallel)
//   train_scan_kernel          sixteen lanes per ray, sequential raw2alpha (FieldBase.py:12-19) over the stored factors: weights,
//                              active counts (pass 0) and the (ray, step)-ordered active list (pass 1): deterministic
//   train_fold_kernel          W1' = W1[:, :144] . basis and the padded / transposed LDS images of the colour MLP
//   train_color_fwd_kernel     16 active samples per wave pass on v_mfma_f32_16x16x4_f32: [144 colour features, view] -> 64 ->
//                              64 -> 3 sigmoid; colours to the dense buffer, activations to HBM rows
//   train_composite_bwd_kernel sixteen lanes per ray: rgb_map, clamp, residual, loss; then d loss / d xs for every sample from
//                              the closed form of the cumprod backward (two sequential sweeps, no gathers)
//   train_color_bwd_kernel     the data gradients of the colour MLP with the TRANSPOSED weights on the matrix pipe, d loss / d t, the
//                              feature-gradient rows and every (plane, sample) pair's cell, weights and place in its bin
//   train_bin_*_kernel         d loss / d colour planes without atomics: pairs ordered by 8x8-cell bins (prefix, perm), summed per unit
//                              in an LDS tile (scatter), the units' tiles added up per texel (gather)            -- section 5b
//   xty_all_kernel             weight gradients as sample-reduction GEMMs  dW = Delta^T . In  (MFMA, operand slabs through LDS, one launch);
//   train_unfold_kernel        M -> d W1[:, :144], d basis
//   train_density_bwd_kernel   every valid sample: ONE scalar per tap into the density-gradient images (the decoder is linear:
//                              rank-one gradient, expanded by train_density_finish_kernel), d loss / d t -> gauge planes
//   adam_*_kernel              torch.optim.Adam's update; planes read their gradient from the packed layout and add the L1 term
// ngf_train_backward (ngf_field.hip) runs the weight-gradient GEMMs, the colour-plane scatter and the density backward side by side on
// three streams after the colour backward (event fork / join; the caller's stream order is kept).
//
// basis has neither bias nor activation (networks.py:17,26), so layer 1 acts on the features through W1' = W1[:, :144] . basis
// (64 x 144).  train_fold_kernel rebuilds W1' from the current weights every step (float64 accumulate); forward and data
// gradients then need ONE 160-wide layer instead of a 144x144 product plus a 160-wide layer, and by the chain rule
//     M = Delta1^T F (64 x 144, the only big sample reduction)   ->   dW1[:, :144] = M . basis^T ,  d basis = W1[:, :144]^T . M
// so g = basis . f is never materialised.  W1', W2 (and their transposes for the backward) are padded to bank-conflict-free
// strides and live in LDS for the whole kernel (57-61 KB per workgroup); tiles of 16 samples live in wave-private LDS as
// [k][16] so that a lane's MFMA B operand is in[k][sample].
#pragma once
#include "ngf_device.hpp"
#include "ngf_render.hpp"
#include "ngf_stage.hpp"          // wave_min

namespace ngf {

constexpr int kTrainWaves = 6;                 // waves per workgroup in the colour forward kernel (H2 overwrites the dead feature tile)
#ifndef NGF_TRAIN_WAVES_BWD
#define NGF_TRAIN_WAVES_BWD 7
#endif
constexpr int kTrainWavesBwd = NGF_TRAIN_WAVES_BWD;              // ... and in the colour backward kernel (its tiles alias, see kBwdTileFloats)
constexpr int kFeat = 144;                     // colour features (3 planes x 48)
constexpr int kIn1 = 159, kIn1Pad = 160;       // [g(144), view(15)] (+1 zero pad)
constexpr int kTs = 17;                        // row stride of the wave tiles [k][kTs]: odd, so that BOTH the MFMA operand reads (16 samples of a row) and the
                                               // row <-> global-row transposes (64 rows of one sample) spread over the banks (stride 16: 32 lanes per bank)
constexpr int kLd1 = 164, kLd2 = 68;           // LDS row strides == 4 (mod 32): lane (row n, k = 4j+q) -> bank 4n+q, two lanes per bank
// forward image : W1' [64][kLd1] (cols 144..158 = W1's view columns, 159 = 0) | W2 [64][kLd2] | W3 [3][64] | b1 [64] | b2 [64] | b3 [4]
// backward image: W1'^T [144][kLd2] | W2^T [64][kLd2] | W3 [3][64]
constexpr int kFwdW1 = 0, kFwdW2 = kFwdW1 + 64 * kLd1, kFwdW3 = kFwdW2 + 64 * kLd2, kFwdB1 = kFwdW3 + 192, kFwdB2 = kFwdB1 + 64, kFwdB3 = kFwdB2 + 64,
              kFwdImage = kFwdB3 + 4;
constexpr int kBwdW1T = 0, kBwdW2T = kBwdW1T + kFeat * kLd2, kBwdW3 = kBwdW2T + 64 * kLd2, kBwdImage = kBwdW3 + 192;

struct TrainArgs {
    RenderArgs R;            // rays, jitter, n, S, mode (= gauge on), geometry, packed textures, mask, white_bg
    float *g_dens[3], *g_app[3], *g_gau[3];     // gradient textures, same packed layouts as R.dens / R.app / R.gau
    const float *wd, *bd;                       // density_decoder.weight [48], bias [1]
    const float *basis, *w1, *b1, *w2, *b2, *w3, *b3;
    float *g_wd, *g_bd;
    float *g_b1, *g_b2, *g_b3;                  // bias gradients of the colour MLP (summed inside train_color_bwd_kernel)
    float *q_dens[3];        // wd-projected density planes (1 channel, packed): Q_p = sum_c wd[16p+c] plane_p[c]
    float *d_dens[3];        // scalar density-gradient images: D_p[texel] = sum_samples w_tap * dx, in 4x4-texel blocks (d_bw per row)
    int32_t d_bw[3], g_bw[3];                   // blocks per row of d_dens / g_gau (g_gau: 4x2-texel blocks of 2 channels)
    // ray-major dense per-sample buffers: index = ray * S + step (a wave's 64 consecutive steps of one ray are 64 consecutive floats)
    float *et, *sg, *w, *dx; // [n,S]  et = exp(-sigma * dist * distance_scale) (1 - alpha), sg = softplus'(xs) (0 where the sample is invalid)
    float *c;                // [n,S,3]
    float *dt;               // [n,S,6]   d loss / d t from the colour path (active samples only)
    const float *target;     // [n,3]
    float *G;                // [n,3]   d loss / d rgb_map (before the clamp), 0 where clamped
    int32_t *count;          // [n]     active samples per ray
    int32_t *offset;         // [n+1]   exclusive prefix of count
    int32_t *list;           // [cap,2] (ray, step) in (ray, step) order
    float *list_w;           // [cap]
    // activations of the current chunk, sample-major rows
    float *F, *V, *H1, *H2, *D3, *D2, *D1;          // [chunk, 144|16|64|64|16|64|64]  (V = the 15 view inputs + 0)
    const float *fwd_image, *bwd_image;             // LDS images built by train_fold_kernel
    const float *fwd16_image;                       // the forward's image in the eval pass's layout (MlpLayout16<48>, ngf_shade16.hpp), train_color_fwd16_kernel
    float *M;                                       // [64,144] = Delta1^T F, accumulated over the chunks
    // colour-plane scatter by bins (section 5b): a bin = the cells of one 8x8 block of one colour plane
    float *DF;                                      // [chunk, 144]  d loss / d colour features of the chunk's samples
    int32_t *pair_cell, *pair_rank;                 // [3][bin_cap]  cell (column | row << 16, padded) of pair (plane, sample); its rank inside its bin (-1: no weight)
    float *pair_w;                                  // [3][bin_cap][4]  the four tap weights
    int32_t *bin_count, *bin_off;                   // [nbins + 1]
    int32_t *perm;                                  // [3 * bin_cap]  sample rows ordered by bin
    int32_t *units;                                 // [unit_cap][3]  (bin, first entry of perm, entries): at most kBinChunk entries per unit
    int32_t *unit_total;                            // [1]
    int32_t *bin_unit;                              // [nbins + 1]  first unit of every bin
    float *slab;                                    // [unit_cap][81][48]  the units' texel sums (train_bin_gather_kernel adds them up)
    int32_t bin_base[3], bin_nbx[3], nbins, bin_cap;
    int32_t bin_accumulate;                         // train_bin_gather_kernel adds to the planes' gradients (chunks after the first) instead of writing them
    double *loss;            // [2]: [0] sum of squared residuals, [1] d loss / d density bias accumulated in double (train_density_bwd_kernel; rounded to g_bd once, by train_unblock_gauge_kernel)
    int32_t chunk_base, chunk_n;      // the slice of the active list this launch works on
    const int32_t *n_active_dev;      // non-NULL: the active count lives on the device (offset[n]); chunk_n is then only the capacity and
                                      // the kernels clip it themselves -- no host round trip between the scan and the colour kernels
    int32_t store;           // colour forward: also write F, V, H1, H2 rows of the chunk
    unsigned long long *prof; // [16] section clocks of the colour backward (ablate bit 1 << 20; profiles/exp_train_sections.py), else unused
    float inv_count;         // 1 / (3 n): the mean of the MSE
    // the two-call form of the step (ngf_train_forward / ngf_train_backward_grad: the torch.autograd boundary of Base.forward)
    float *rgb_out, *depth_out;       // [n,3], [n]: rgb_map after the clamp and depth_map (FieldBase.py:296-306), written by the forward call
    const float *d_rgb;               // [n,3]: d loss / d rgb_map handed in by the caller (instead of the MSE residual against `target`)
};

static_assert(offsetof(TrainArgs, R) == 0, "karg_tex (ngf_render.hpp) reads the RenderArgs at offset 0 of the kernel arguments");

// ---- geometry of sample (ray r, step i): Base.sample_ray + normalize_coord + compute_gauge --------------------------------
// returns valid; xn = normalised position (compute_gauge and the gauge gradient start from it)
__device__ __forceinline__ int chunk_rows(const TrainArgs &T)
{
    if (!T.n_active_dev) return T.chunk_n;
    const int left = *T.n_active_dev - T.chunk_base;
    return left < 0 ? 0 : (left < T.chunk_n ? left : T.chunk_n);
}

__device__ __forceinline__ bool sample_geometry(const RenderArgs &A, int64_t r, int i, float xn[3], float &z, float &dist)
{
    float o[3], d[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) { o[k] = A.rays[r * 6 + k]; d[k] = A.rays[r * 6 + 3 + k]; }
    const float jit = A.jitter ? A.jitter[r] : 0.0f;
    float tmin = -INFINITY;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        float vec = (d[k] == 0.0f) ? 1e-6f : d[k];
        float ra = (A.a1[k] - o[k]) / vec, rb = (A.a0[k] - o[k]) / vec;
        tmin = fmaxf(tmin, fminf(ra, rb));
    }
    tmin = fminf(fmaxf(tmin, A.near_), A.far_);
    z = tmin + A.step * ((float)i + jit);
    const float zn = tmin + A.step * ((float)(i + 1) + jit);
    dist = (i < A.S - 1) ? (zn - z) : 0.0f;
    float p[3];
    bool valid = true;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        p[k] = o[k] + d[k] * z;
        valid = valid & !((A.a0[k] > p[k]) | (p[k] > A.a1[k]));
    }
    if (A.mask.bits && valid) valid = mask_occupied(A.mask, p);
#pragma unroll
    for (int k = 0; k < 3; ++k) xn[k] = (p[k] - A.a0[k]) * A.inv[k] - 1.0f;
    return valid;
}

// bilinear cell with what the backward needs: tap weights, the fractions and the d(pixel)/d(coord) scales
struct BilG {
    int32_t idx;
    int32_t cx, cy;                // padded (column, row) of the first tap
    float w00, w10, w01, w11;
    float wx0, wx1, wy0, wy1;      // 0 when the cell is out of range
    float sx, sy;                  // d px / d u = (W-1)/2, d py / d v = (H-1)/2
};

__device__ __forceinline__ BilG bilg_setup(float u, float v, const Tex &t)
{
    float px = ((u + 1.0f) / 2.0f) * t.fw;
    float py = ((v + 1.0f) / 2.0f) * t.fh;
    float fx = floorf(px), fy = floorf(py);
    float wx1 = px - fx, wx0 = 1.0f - wx1;
    float wy1 = py - fy, wy0 = 1.0f - wy1;
    bool in = (fx >= -1.0f) & (fx <= t.fw) & (fy >= -1.0f) & (fy <= t.fh);
    float cx = fminf(fmaxf(fx, -1.0f), t.fw);
    float cy = fminf(fmaxf(fy, -1.0f), t.fh);
    BilG b;
    b.cx = (int)cx + 1; b.cy = (int)cy + 1;
    b.idx = b.cy * t.stride + b.cx;
    b.wx0 = in ? wx0 : 0.0f; b.wx1 = in ? wx1 : 0.0f; b.wy0 = in ? wy0 : 0.0f; b.wy1 = in ? wy1 : 0.0f;
    b.w00 = in ? wx0 * wy0 : 0.0f;
    b.w10 = in ? wx1 * wy0 : 0.0f;
    b.w01 = in ? wx0 * wy1 : 0.0f;
    b.w11 = in ? wx1 * wy1 : 0.0f;
    b.sx = in ? t.fw * 0.5f : 0.0f;
    b.sy = in ? t.fh * 0.5f : 0.0f;
    return b;
}

__device__ __forceinline__ float softplus_pre(float u)          // F.softplus, threshold 20; softplus(-inf) = 0
{
    return u > 20.0f ? u : log1pf(expf(u));
}

// ---- 1. density features of every (step, ray) pair ---------------------------------------------------------------------
__global__ void __launch_bounds__(256) train_density_kernel(const TrainArgs T)
{
    NGF_KARG_CONTRACT_T(&train_density_kernel, TrainArgs);      // TrainArgs starts with its RenderArgs (static_assert above)
    const RenderArgs &A = T.R;
    const int64_t total = (int64_t)A.S * A.n;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t wi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; wi < total; wi += stride) {
        // a wave takes 64 CONSECUTIVE steps of one ray: its gathers share texels (two steps per texel) instead of touching 64
        // unrelated cells of 64 random training rays, and its stores are one 256-byte run of the ray-major buffers
        const int64_t r = wi / A.S;
        const int i = (int)(wi - r * A.S);
        const int64_t idx = r * A.S + i;
        float xn[3], z, dist, tt[6];
        const bool valid = sample_geometry(A, r, i, xn, z, dist);
        if (!valid) {                               // outside the box / in free space of the alpha mask: sigma = 0, no fetches
            T.et[idx] = 1.0f;
            T.sg[idx] = 0.0f;
            continue;
        }
        triplane_gauge(A, xn, A.mode, tt);          // compute_gauge (Field.py:53-75), identity split when the gauge is off
        // density_decoder is linear: the four taps come from the wd-projected one-channel planes Q_p (train_project_density_kernel
        // sums the 16 channels of a texel in the order the per-tap dot product here used to) -- 12 floats per sample, not 192
        float f = 0.0f;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const Tex &tx = A.dens[p];
            Bil b = bil_setup(tt[2 * p], tt[2 * p + 1], tx);
            const float *q = T.q_dens[p] + b.idx;
            f += bil_mix(b, q[0], q[1], q[tx.stride], q[tx.stride + 1]);
        }
        f = (f + T.bd[0]) + (-10.0f);               // xs = Linear(48,1) - 10, the pre-softplus density
        // every per-sample transcendental is taken HERE, 3.6 M samples in parallel: the two sequential kernels (one wave per SIMD, nothing
        // to hide an instruction's latency behind) only chain products and sums.  raw2alpha (FieldBase.py:12-19): alpha = 1 - et
        T.et[idx] = expf(-softplus_pre(f) * (dist * A.dscale));
        T.sg[idx] = f > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-f));       // softplus' for the compositing backward
    }
}


// the per-ray scalars every sequential sweep needs
__device__ __forceinline__ float ray_tmin(const RenderArgs &A, int64_t r)
{
    float tmin = -INFINITY;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float o = A.rays[r * 6 + k], d = A.rays[r * 6 + 3 + k];
        float vec = (d == 0.0f) ? 1e-6f : d;
        float ra = (A.a1[k] - o) / vec, rb = (A.a0[k] - o) / vec;
        tmin = fmaxf(tmin, fminf(ra, rb));
    }
    return fminf(fmaxf(tmin, A.near_), A.far_);
}

// ---- 2. raw2alpha per ray; pass 0 writes weights / transmittances and counts the active samples, pass 1 the ordered active list ----
// Sixteen lanes per ray (one DPP row), each on one of 16 consecutive steps: the loads and the exp/log of a block of steps run
// in parallel and only the transmittance product is chained lane to lane IN STEP ORDER (row_shr:1), so weights are exactly
// those of the sequential cumprod (FieldBase.py:16) while a 4096-ray batch fills 1024 waves instead of 64.  That is ONE wave per
// SIMD: nothing hides a global load, so the loads of kScanAhead blocks are requested together (round 2 paid one exposed memory
// latency per 16 steps: 56 per ray and pass, 0.8 us each -- the whole of the kernel's time).
constexpr int kScanAhead = 8;

__device__ __forceinline__ float row_shr1(float v, float fill)      // lane (row, s) <- lane (row, s-1); lane s = 0 gets `fill`
{
    const int r = __builtin_amdgcn_update_dpp(__float_as_int(fill), __float_as_int(v), 0x111, 0xf, 0xf, false);
    return __int_as_float(r);
}

__global__ void __launch_bounds__(64) train_scan_kernel(const TrainArgs T, int pass)
{
    const RenderArgs &A = T.R;
    const int lane = threadIdx.x & 63, seg = lane & 15, rl = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * 4 + rl;
    const bool live = r0 < A.n;
    const int64_t r = live ? r0 : A.n - 1;
    int cnt = 0;
    if (pass) {
        // the weights are there: threshold, rank inside the ray, write
        const int base = T.offset[r];
        for (int i0 = 0; i0 < A.S; i0 += 16 * kScanAhead) {
            float wv[kScanAhead];
#pragma unroll
            for (int u = 0; u < kScanAhead; ++u) {
                const int i = i0 + 16 * u + seg;
                wv[u] = i < A.S ? T.w[r * A.S + i] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < kScanAhead; ++u) {
                const int i = i0 + 16 * u + seg;
                const bool active = live && (i < A.S) && (wv[u] > A.thr);
                const unsigned m16 = (unsigned)((__ballot(active) >> (16 * rl)) & 0xffffull);
                if (active) {
                    const int pos = base + cnt + __popc(m16 & ((1u << seg) - 1u));
                    T.list[2 * (int64_t)pos] = (int)r;
                    T.list[2 * (int64_t)pos + 1] = i;
                    T.list_w[pos] = wv[u];
                }
                cnt += __popc(m16);
            }
        }
        return;
    }
    float Tr = 1.0f;
    for (int i0 = 0; i0 < A.S; i0 += 16 * kScanAhead) {
        float xv[kScanAhead];
#pragma unroll
        for (int u = 0; u < kScanAhead; ++u) {
            const int i = i0 + 16 * u + seg;
            xv[u] = i < A.S ? T.et[r * A.S + i] : 1.0f;
        }
#pragma unroll
        for (int u = 0; u < kScanAhead; ++u) {
            if (i0 + 16 * u >= A.S) break;                          // wave-uniform
            const int i = i0 + 16 * u + seg;
            const bool inb = i < A.S;
            const int64_t idx = r * A.S + (inb ? i : 0);
            const float alpha = 1.0f - xv[u];
            const float keep = (1.0f - alpha) + 1e-10f;             // steps past S: alpha = 0, keep rounds to 1
            // T entering step i0+seg = Tr * keep_0 * ... * keep_{seg-1}, multiplied in that order: after round k lanes <= k are final
            float Tin = Tr;
#pragma unroll
            for (int k = 1; k < 16; ++k) {
                const float prev = row_shr1(Tin * keep, Tr);
                Tin = seg > 0 ? prev : Tr;
            }
            const float w = alpha * Tin;
            Tr = __shfl(Tin * keep, rl * 16 + 15);
            const bool active = live && inb && (w > A.thr);
            const unsigned m16 = (unsigned)((__ballot(active) >> (16 * rl)) & 0xffffull);
            if (live && inb) {
                T.w[idx] = w;
                T.dx[idx] = Tin;                                    // parked for the compositing backward
            }
            cnt += __popc(m16);
        }
    }
    if (live && seg == 0) T.count[r] = cnt;
}

// exclusive prefix of count[0..n) -> offset[0..n]; one block: every thread sums a run of consecutive counts, ONE scan over the 1024 run
// sums, then the run is written (round 2 scanned chunk after chunk of 1024: 80 barriers for a 4096-ray batch, 10 us)
// rows_cap / overflow (speculative rows, ngf_train_desc::chunk_samples < 0): when the batch has more active samples than the trainer keeps
// activation rows for, overflow[0] = 1 for THIS step (the Adam kernels then leave parameters and moments alone: a truncated gradient is never
// applied) and the sticky counter overflow[1] goes up (ngf_train_overflow_count); otherwise overflow[0] = 0.
__global__ void __launch_bounds__(1024) train_prefix_kernel(const int32_t *count, int64_t n, int32_t *offset, int64_t rows_cap = 0, int32_t *overflow = nullptr)
{
    __shared__ int sh[1024];
    const int t = threadIdx.x;
    const int64_t per = (n + 1023) / 1024;
    const int64_t b0 = t * per, b1 = b0 + per < n ? b0 + per : n;
    int c = 0;
    for (int64_t b = b0; b < b1; ++b) c += count[b];
    sh[t] = c;
    __syncthreads();
    for (int s = 1; s < 1024; s <<= 1) {
        const int add = t >= s ? sh[t - s] : 0;
        __syncthreads();
        sh[t] += add;
        __syncthreads();
    }
    int run = sh[t] - c;
    for (int64_t b = b0; b < b1; ++b) {
        offset[b] = run;
        run += count[b];
    }
    if (t == 1023) {
        offset[n] = sh[1023];
        if (overflow) {
            const int over = rows_cap > 0 && (int64_t)sh[1023] > rows_cap;
            overflow[0] = over;
            if (over) overflow[1] += 1;
        }
    }
}

// ---- MFMA building block: out[M][16] = act( W . in[K][16] + bias ) with 16 samples as the N dimension -------------------
// v_mfma_f32_16x16x4_f32: lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15]; it receives D[i = 4(l>>4)+r][j = l&15].
//   TRANS = false: Wm[m][k] = W[m*ldw + k] (forward, nn.Linear weight [out,in]);  TRANS = true: Wm[m][k] = W[k*ldw + m]
//   ACT: 0 none, 1 relu, 2 = multiply by (mask[m][n] > 0)   (the relu backward)
template <bool TRANS, int ACT, int M, int K, int KVALID, int OUT_T = 0>      // OUT_T > 0: out[sample][row] with row stride OUT_T
__device__ __forceinline__ void dense16(const float *__restrict__ W, int ldw, int Mvalid, const float *__restrict__ bias, const float *in,
                                        float *out, const float *mask, int lane)
{
    constexpr int KS = K / 4;                   // MFMA k-steps per 16-row block
    const int n = lane & 15, q = lane >> 4;
    // the B operands (this lane's sample, k = 4j + q) are the same for every row block: read the tile once
    float b[KS];
#pragma unroll
    for (int j = 0; j < KS; ++j) b[j] = in[(4 * j + q) * kTs + n];
    auto load_a = [&](int mb, float (&a)[KS]) {
        const int m = mb + n;                   // the A row this lane supplies
        const bool mok = m < Mvalid;
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const int k = 4 * j + q;
            const bool ok = mok && (KVALID == K || k < KVALID);
            const int off = TRANS ? k * ldw + m : m * ldw + k;
            a[j] = ok ? W[off] : 0.0f;
        }
    };
    auto block = [&](int mb, const float (&a)[KS]) {
        f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < KS; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], acc, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = mb + 4 * q + r;
            float v = acc[r];
            if (bias && row < Mvalid) v += bias[row];
            if (ACT == 1) v = fmaxf(v, 0.0f);
            if (ACT == 2) v = mask[row * kTs + n] > 0.0f ? v : 0.0f;
            if (OUT_T) out[n * OUT_T + row] = v;
            else out[row * kTs + n] = v;
        }
    };
    // software pipeline over the row blocks, two per trip (a real loop: a full unroll hoists every block's loads and spills):
    // the next block's weights are requested before this block's MFMAs are issued
    float a0[KS], a1[KS];
    load_a(0, a0);
#pragma unroll 1
    for (int mb = 0; mb < M; mb += 32) {
        if (mb + 16 < M) load_a(mb + 16, a1);
        __builtin_amdgcn_sched_barrier(0);
        block(mb, a0);
        __builtin_amdgcn_sched_barrier(0);
        if (mb + 16 < M) {
            if (mb + 32 < M) load_a(mb + 32, a0);
            __builtin_amdgcn_sched_barrier(0);
            block(mb + 16, a1);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// LDS tile [K][16] -> global rows [sample][ld] (and back): lane (q, n) moves rows k = q, q+4, ... of sample n
__device__ __forceinline__ void tile_to_rows(const float *tile, int K, float *rows, int ld, int64_t slot, bool live, int lane)
{
    // 64 lanes write 64 consecutive k of one sample at a time: coalesced 256-byte stores
    for (int s = 0; s < 16; ++s) {
        const bool ok = __shfl((int)live, s) != 0;       // lane s holds sample s's liveness (lanes 0..15 = q 0)
        if (!ok) continue;
        const int64_t sl = __shfl((int)(slot & 0x7fffffff), s);
        for (int k = lane; k < K; k += 64) rows[sl * ld + k] = tile[k * kTs + s];
    }
}

// global rows -> tile.  All 16 row reads of a lane are issued before the first LDS store: written as one loop the wave waited for every
// load in turn (16 global latencies per tile -- a third of the colour backward's time, profiles/exp_train_sections.py)
template <int K>
__device__ __forceinline__ void rows_to_tiles2(const float *rows_a, const float *rows_b, int ld, float *tile_a, float *tile_b, int64_t slot, bool live, int lane)
{
    static_assert(K == 64, "one element per lane and sample");
    float va[16], vb[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
        const bool ok = __shfl((int)live, s) != 0;
        const int64_t sl = __shfl((int)(slot & 0x7fffffff), s);
        va[s] = ok ? rows_a[sl * ld + lane] : 0.0f;
        vb[s] = ok ? rows_b[sl * ld + lane] : 0.0f;
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) tile_a[lane * kTs + s] = va[s];
#pragma unroll
    for (int s = 0; s < 16; ++s) tile_b[lane * kTs + s] = vb[s];
}

// gauge-shifted coordinates of a list sample (recomputed: three 2-channel fetches, cheaper than 24 bytes of HBM per valid sample)
__device__ __forceinline__ void list_sample_coords(const RenderArgs &A, int64_t r, int i, float t[6], float xn[3])
{
    float z, dist;
    (void)sample_geometry(A, r, i, xn, z, dist);
    triplane_gauge(A, xn, A.mode, t);
}

constexpr int kFwdTileFloats = (kIn1Pad + 64) * kTs;                       // [F; view] (H2 goes there once layer 1 is done and F is stored), H1
constexpr int kDfStride = kFeat + 1;                                       // DF is kept sample-major (bank-conflict-free rows)
// backward tiles per wave: [H1 | D2 | pad] is overwritten by DF^T once d1 exists and d2 / d1 have been written out; [H2, then D1]
constexpr int kBwdTileFloats = 16 * kDfStride + 64 * kTs;
static_assert(16 * kDfStride >= 2 * 64 * kTs, "DF^T must cover the H1 and D2 tiles it aliases");

// ---- per-step weight images -------------------------------------------------------------------------------------------------
// fwd16 (round 5): the same W1' / W2 / W3 / biases in the eval pass's image layout MlpLayout16<48> (ngf_shade16.hpp: what build_rgb_image16 makes on
// the host for a render handle) -- A operands [unit tile][k-step][lane], lane (i, kq) of k-step t = input kmap(t, kq) of unit mt * 16 + i;
// hidden units in accumulator order n = mt * 16 + 4 kq + r.
__global__ void __launch_bounds__(256) train_fold_kernel(const TrainArgs T, float *fwd, float *bwd, float *fwd16)
{
    using L16 = MlpLayout16<48>;
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    for (int i = tid; i < 64 * kIn1Pad; i += nth) {                  // W1' and its transpose
        const int m = i / kIn1Pad, k = i % kIn1Pad;
        float v = 0.0f;
        if (k < kFeat) {
            double s = 0.0;
            for (int j = 0; j < kFeat; ++j) s += (double)T.w1[m * kIn1 + j] * (double)T.basis[j * kFeat + k];
            v = (float)s;
            bwd[kBwdW1T + k * kLd2 + m] = v;
        } else if (k < kIn1) {
            v = T.w1[m * kIn1 + k];
        }
        fwd[kFwdW1 + m * kLd1 + k] = v;
        {   // the one (k-step, lane) of the 16-wide image that multiplies input k into unit m
            const int mt = m >> 4, iu = m & 15;
            int t16, kq;
            if (k < kFeat) { const int P = k / 48, c = k % 48; kq = (c & 15) >> 2; t16 = P * L16::QCH + 4 * (c >> 4) + (c & 3); }
            else { const int e = k - kFeat; kq = e >> 2; t16 = 3 * L16::QCH + (e & 3); }
            fwd16[L16::W1 + (mt * L16::KT + t16) * 64 + kq * 16 + iu] = v;
        }
    }
    for (int i = tid; i < 64 * 64; i += nth) {
        const int m = i / 64, k = i % 64;
        const float v = T.w2[m * 64 + k];
        fwd[kFwdW2 + m * kLd2 + k] = v;
        bwd[kBwdW2T + k * kLd2 + m] = v;
        {   // input hidden unit k = mt' * 16 + 4 kq + r is k-step t = 4 mt' + r of lane quarter kq
            const int mt = m >> 4, iu = m & 15, kq = (k & 15) >> 2, t16 = 4 * (k >> 4) + (k & 3);
            fwd16[L16::W2 + (mt * 16 + t16) * 64 + kq * 16 + iu] = v;
        }
    }
    for (int i = tid; i < 192; i += nth) {
        fwd[kFwdW3 + i] = T.w3[i]; bwd[kBwdW3 + i] = T.w3[i];
        const int c = i / 64, n = i % 64, kq = (n & 15) >> 2, kk = 4 * (n >> 4) + (n & 3);         // unit n = (kk >> 2) * 16 + 4 kq + (kk & 3)
        fwd16[L16::W3 + c * 64 + kq * 16 + kk] = T.w3[i];
    }
    for (int i = tid; i < 64; i += nth) {
        fwd[kFwdB1 + i] = T.b1[i]; fwd[kFwdB2 + i] = T.b2[i];
        const int kq = (i & 15) >> 2, kk = 4 * (i >> 4) + (i & 3);
        fwd16[L16::B1 + kq * 16 + kk] = T.b1[i];
        fwd16[L16::B2 + kq * 16 + kk] = T.b2[i];
    }
    for (int i = tid; i < 4; i += nth) { fwd[kFwdB3 + i] = i < 3 ? T.b3[i] : 0.0f; fwd16[L16::B3 + i] = i < 3 ? T.b3[i] : 0.0f; }
    // pad columns (never multiplied by a non-zero operand, but keep them defined)
    for (int i = tid; i < 64 * (kLd1 - kIn1Pad); i += nth) fwd[kFwdW1 + (i / (kLd1 - kIn1Pad)) * kLd1 + kIn1Pad + i % (kLd1 - kIn1Pad)] = 0.0f;
    for (int i = tid; i < 64 * (kLd2 - 64); i += nth) {
        fwd[kFwdW2 + (i / (kLd2 - 64)) * kLd2 + 64 + i % (kLd2 - 64)] = 0.0f;
        bwd[kBwdW2T + (i / (kLd2 - 64)) * kLd2 + 64 + i % (kLd2 - 64)] = 0.0f;
    }
    for (int i = tid; i < kFeat * (kLd2 - 64); i += nth) bwd[kBwdW1T + (i / (kLd2 - 64)) * kLd2 + 64 + i % (kLd2 - 64)] = 0.0f;
}

// M = Delta1^T F  ->  dW1[:, :144] += M . basis^T ,  d basis += W1[:, :144]^T . M        (chain rule through g = basis . f)
__global__ void __launch_bounds__(256) train_unfold_kernel(const TrainArgs T, float *g_w1, float *g_basis)
{
    const int tid = blockIdx.x * blockDim.x + threadIdx.x, nth = gridDim.x * blockDim.x;
    for (int i = tid; i < 64 * kFeat; i += nth) {
        const int m = i / kFeat, j = i % kFeat;
        float s = 0.0f;
        for (int k = 0; k < kFeat; ++k) s = fmaf(T.M[m * kFeat + k], T.basis[j * kFeat + k], s);
        g_w1[m * kIn1 + j] += s;
    }
    for (int i = tid; i < kFeat * kFeat; i += nth) {
        const int j = i / kFeat, k = i % kFeat;
        float s = 0.0f;
        for (int m = 0; m < 64; ++m) s = fmaf(T.w1[m * kIn1 + j], T.M[m * kFeat + k], s);
        g_basis[i] += s;
    }
}

// ---- 3. colour forward over the active list -------------------------------------------------------------------------------
__global__ void __launch_bounds__(kTrainWaves * 64) __attribute__((amdgpu_waves_per_eu(1, 2))) train_color_fwd_kernel(const TrainArgs T)
{
    NGF_KARG_CONTRACT_T(&train_color_fwd_kernel, TrainArgs);      // TrainArgs starts with its RenderArgs (static_assert above)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const RenderArgs &A = T.R;
    for (int i = threadIdx.x; i < kFwdImage; i += blockDim.x) smem[i] = T.fwd_image[i];
    __syncthreads();
    const float *img = smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    float *Ft = smem + ((kFwdImage + 3) & ~3) + wave * kFwdTileFloats, *H1t = Ft + kIn1Pad * kTs, *H2t = Ft;
    const int chunk_n = chunk_rows(T);
    const int passes = (chunk_n + 15) / 16;
    // The list entries of a pass and what hangs on them -- ray, gauge-shifted coordinates: three dependent global reads before the first
    // feature tap can be requested -- are fetched during the wave's PREVIOUS pass (requests unconditional: past the last pass, that pass again).
    const int stride = gridDim.x * kTrainWaves;
    auto want_list = [&](int ps, int &r_, int &i_) {
        const int lc = ps * 16 + n;
        const int64_t sl = T.chunk_base + (lc < chunk_n ? lc : 0);
        r_ = T.list[2 * sl]; i_ = T.list[2 * sl + 1];
    };
    auto want_coords = [&](int r_, int i_, float (&t_)[6], float (&d_)[3]) {
        float xn_[3];
        list_sample_coords(A, r_, i_, t_, xn_);
        d_[0] = A.rays[(int64_t)r_ * 6 + 3]; d_[1] = A.rays[(int64_t)r_ * 6 + 4]; d_[2] = A.rays[(int64_t)r_ * 6 + 5];
    };
    int pass = blockIdx.x * kTrainWaves + wave;
    int r_cur = 0, i_cur = 0;
    float t[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, dcur[3] = {0.0f, 0.0f, 1.0f};
    if (pass < passes) {
        want_list(pass, r_cur, i_cur);
        want_coords(r_cur, i_cur, t, dcur);
    }
    for (; pass < passes; pass += stride) {
        const int local = pass * 16 + n;
        const bool live = local < chunk_n;
        const int64_t r = r_cur;
        const int i = i_cur;
        const int next = pass + stride < passes ? pass + stride : pass;
        int r_nx, i_nx;
        want_list(next, r_nx, i_nx);
        // compute_rgb's fetch (Field.py:93-103): lane (q, n) interpolates channels 12q .. 12q+11 of every plane for sample n
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const Tex &tx = A.app[p];
            Bil b = bil_setup(t[2 * p], t[2 * p + 1], tx);
            const f32x4 *q00 = reinterpret_cast<const f32x4 *>(tx.p + (size_t)b.idx * 48) + 3 * q;
            const f32x4 *q01 = q00 + (size_t)tx.stride * 12;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                f32x4 v00 = q00[j], v10 = q00[12 + j], v01 = q01[j], v11 = q01[12 + j];
#pragma unroll
                for (int e = 0; e < 4; ++e) Ft[(p * 48 + 12 * q + 4 * j + e) * kTs + n] = live ? bil_mix(b, v00[e], v10[e], v01[e], v11[e]) : 0.0f;
            }
        }
        // the view inputs of layer 1 (networks.py:27-29): rows 144..159 of the input tile
        {
            float d[3] = {dcur[0], dcur[1], dcur[2]}, v[16];
            view_inputs(d, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) {          // v[4q + j] by selects: a q-indexed read would put v into scratch memory
                const float vq = q == 0 ? v[j] : (q == 1 ? v[4 + j] : (q == 2 ? v[8 + j] : v[12 + j]));
                Ft[(kFeat + 4 * q + j) * kTs + n] = live ? vq : 0.0f;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        dense16<false, 1, 64, kIn1Pad, kIn1Pad>(img + kFwdW1, kLd1, 64, img + kFwdB1, Ft, H1t, nullptr, lane);      // relu(W1' f + W1v view + b1)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (T.store) {          // the feature tile leaves LDS now: layer 2 writes its output over it
            const int64_t row = live ? local : 0;
            tile_to_rows(Ft, kFeat, T.F, kFeat, row, live, lane);
            tile_to_rows(Ft + kFeat * kTs, 16, T.V, 16, row, live, lane);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        float t_nx[6], d_nx[3];
        want_coords(r_nx, i_nx, t_nx, d_nx);
        dense16<false, 1, 64, 64, 64>(img + kFwdW2, kLd2, 64, img + kFwdB2, H1t, H2t, nullptr, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // layer 3 + sigmoid on the VALU: lane (q, n) sums its 16 hidden units, then the four quarters meet
        float c[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s = fmaf(img[kFwdW3 + j * 64 + 16 * q + k], H2t[(16 * q + k) * kTs + n], s);
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            s += img[kFwdB3 + j];
            c[j] = 1.0f / (1.0f + expf(-s));
        }
        if (live && q == 0) {
            float *dst = T.c + ((int64_t)r * A.S + i) * 3;
            dst[0] = c[0]; dst[1] = c[1]; dst[2] = c[2];
        }
        if (T.store) {
            const int64_t row = live ? local : 0;
            tile_to_rows(H1t, 64, T.H1, 64, row, live, lane);
            tile_to_rows(H2t, 64, T.H2, 64, row, live, lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        r_cur = r_nx; i_cur = i_nx;
#pragma unroll
        for (int k = 0; k < 6; ++k) t[k] = t_nx[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) dcur[k] = d_nx[k];
    }
}

// ---- 3b. colour forward in the eval pass's shape (round 5) -----------------------------------------------------------------------------------
// train_color_fwd_kernel above keeps its tiles in LDS ([k][17] transposes between MFMA operands and sample-major rows): 152 KB per workgroup, six
// waves per CU, 200 us for ~170 k samples -- the render kernel's level-1 shade pass does the same arithmetic for 16 samples in ~1.6 us of a SIMD
// shared by three waves, ten times the rate.  This kernel IS that pass (gather16_issue / mix16 / layer1_plane16 of ngf_shade16.hpp, twelve waves per
// CU, the 58.6 KB MlpLayout16<48> image train_fold_kernel now also builds) with the activation rows stored straight from the registers that hold
// them: lane (s, kq) has channels 16 q + 4 kq .. + 3 of every plane, its four view inputs and hidden units mt * 16 + 4 kq .. + 3 of its own sample --
// 16-byte pieces of the rows F [144], V [16], H1 [64], H2 [64] (post-ReLU, as before).  Same features to the bit (same bil_setup / bil_mix);
// layers 1-2 sum in the MFMA's k order of the eval pass instead of dense16's: last-bit differences in H1 / H2 / colours.
constexpr int kTrainWaves16 = 12;
__global__ void __launch_bounds__(kTrainWaves16 * 64) train_color_fwd16_kernel(const TrainArgs T)
{
    NGF_KARG_CONTRACT_T(&train_color_fwd16_kernel, TrainArgs);      // gather16_issue reads the colour-plane descriptors from the kernel-argument segment
    using L = MlpLayout16<48>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const RenderArgs &A = T.R;
    stage_blob(smem, T.fwd16_image, L::TOTAL);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, kq = lane >> 4;
    const int chunk_n = chunk_rows(T);
    const int passes = (chunk_n + 15) / 16;
    const int stride = gridDim.x * kTrainWaves16;
    auto want_list = [&](int ps, int &r_, int &i_) {
        const int lc = ps * 16 + n;
        const int64_t sl = T.chunk_base + (lc < chunk_n ? lc : 0);
        r_ = T.list[2 * sl]; i_ = T.list[2 * sl + 1];
    };
    auto want_coords = [&](int r_, int i_, float (&t_)[6], float (&d_)[3]) {
        float xn_[3];
        list_sample_coords(A, r_, i_, t_, xn_);
        d_[0] = A.rays[(int64_t)r_ * 6 + 3]; d_[1] = A.rays[(int64_t)r_ * 6 + 4]; d_[2] = A.rays[(int64_t)r_ * 6 + 5];
    };
    int pass = blockIdx.x * kTrainWaves16 + wave;
    int r_cur = 0, i_cur = 0;
    float t[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f}, dcur[3] = {0.0f, 0.0f, 1.0f};
    if (pass < passes) {
        want_list(pass, r_cur, i_cur);
        want_coords(r_cur, i_cur, t, dcur);
    }
    for (; pass < passes; pass += stride) {
        const float *blob = per_pass16(smem);
        const int local = pass * 16 + n;
        const bool live = local < chunk_n;
        const bool keep = live && T.store;
        const int64_t row = live ? local : 0;
        const int64_t r = r_cur;
        const int i = i_cur;
        const int next = pass + stride < passes ? pass + stride : pass;
        int r_nx, i_nx;
        want_list(next, r_nx, i_nx);
        const float rec[kRecFloats] = {0.0f, 0.0f, t[0], t[1], t[2], t[3], t[4], t[5]};
        f32x4 v;
        {
            float d[3] = {dcur[0], dcur[1], dcur[2]}, v16[16];
            view_inputs(d, v16);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = kq == 0 ? v16[j] : (kq == 1 ? v16[4 + j] : (kq == 2 ? v16[8 + j] : v16[12 + j]));      // selects: a kq-indexed read would put v16 into scratch
        }
        if (keep) *reinterpret_cast<f32x4 *>(T.V + row * 16 + 4 * kq) = v;
        Gather16<48> g;
        float feat[L::QCH];
        gather16_issue<48, 0>(A, rec, kq, g);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = *reinterpret_cast<const f32x4 *>(blob + L::B1 + kq * 16 + mt * 4);
        {
            const float *w1 = blob + L::W1 + lane;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) acc[mt] = NGF_MFMA16(w1[(mt * L::KT + 3 * L::QCH + j) * 64], v[j], acc[mt]);
        }
        __builtin_amdgcn_sched_barrier(0);
        auto keep_feat = [&](int P) {
            if (keep) {
#pragma unroll
                for (int q = 0; q < 3; ++q)
                    *reinterpret_cast<f32x4 *>(T.F + row * kFeat + P * 48 + 16 * q + 4 * kq) = f32x4{feat[4 * q], feat[4 * q + 1], feat[4 * q + 2], feat[4 * q + 3]};
            }
        };
        mix16<48>(g, feat);
        __builtin_amdgcn_sched_barrier(0);
        gather16_issue<48, 1>(A, rec, kq, g);
        __builtin_amdgcn_sched_barrier(0);
        keep_feat(0);
        layer1_plane16<48, 0>(blob, lane, feat, acc);
        __builtin_amdgcn_sched_barrier(0);
        mix16<48>(g, feat);
        __builtin_amdgcn_sched_barrier(0);
        gather16_issue<48, 2>(A, rec, kq, g);
        __builtin_amdgcn_sched_barrier(0);
        keep_feat(1);
        layer1_plane16<48, 1>(blob, lane, feat, acc);
        __builtin_amdgcn_sched_barrier(0);
        mix16<48>(g, feat);
        __builtin_amdgcn_sched_barrier(0);
        keep_feat(2);
        // the next pass's coordinates (three dependent reads) are requested here, behind plane 2's 48 MFMAs and layer 2
        float t_nx[6], d_nx[3];
        want_coords(r_nx, i_nx, t_nx, d_nx);
        layer1_plane16<48, 2>(blob, lane, feat, acc);
        __builtin_amdgcn_sched_barrier(0);
        // layer 2 on the lane's own 16 hidden units (accumulator order = B operand order), H1 / H2 rows from the registers
        float h[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) h[k] = relu1(acc[k >> 2][k & 3]);
        if (keep) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<f32x4 *>(T.H1 + row * 64 + mt * 16 + 4 * kq) = f32x4{h[4 * mt], h[4 * mt + 1], h[4 * mt + 2], h[4 * mt + 3]};
        }
        f32x4 c2[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) c2[mt] = *reinterpret_cast<const f32x4 *>(blob + L::B2 + kq * 16 + mt * 4);
        {
            const float *w2 = blob + L::W2 + lane;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 16; ++k)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) c2[mt] = NGF_MFMA16(w2[(mt * 16 + k) * 64], h[k], c2[mt]);
            __builtin_amdgcn_sched_barrier(0);
        }
        float hr[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) hr[k] = relu1(c2[k >> 2][k & 3]);
        if (keep) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<f32x4 *>(T.H2 + row * 64 + mt * 16 + 4 * kq) = f32x4{hr[4 * mt], hr[4 * mt + 1], hr[4 * mt + 2], hr[4 * mt + 3]};
        }
        // layer 3 + sigmoid as train_color_fwd_kernel: the lane's 16 units, then the four quarters meet
        const f32x4 *w3 = reinterpret_cast<const f32x4 *>(blob + L::W3 + kq * 16);
        float c[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float sacc = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 w = w3[j * 16 + q];
#pragma unroll
                for (int e = 0; e < 4; ++e) sacc = fmaf(w[e], hr[4 * q + e], sacc);
            }
            sacc += __shfl_xor(sacc, 16);
            sacc += __shfl_xor(sacc, 32);
            sacc += blob[L::B3 + j];
            c[j] = 1.0f / (1.0f + expf(-sacc));
        }
        if (live && kq == 0) {
            float *dst = T.c + ((int64_t)r * A.S + i) * 3;
            dst[0] = c[0]; dst[1] = c[1]; dst[2] = c[2];
        }
        r_cur = r_nx; i_cur = i_nx;
#pragma unroll
        for (int k = 0; k < 6; ++k) t[k] = t_nx[k];
#pragma unroll
        for (int k = 0; k < 3; ++k) dcur[k] = d_nx[k];
    }
}

// ---- 4. compositing forward tail + backward to the pre-softplus density of every sample ------------------------------------
// kCompLanes lanes per ray.  d loss / d alpha_i = dL/dw_i T_i - (sum_{j>i} w_j dL/dw_j) / (1 - alpha_i + 1e-10) with
// dL/dw_j = G . (c_j [active] - bg) is the cumprod backward; the suffix sum is accumulated from the END of the ray in float64
// (what ATen's reverse cumsum does on the CPU), T_i was parked in dx by the scan.
#ifndef NGF_COMP_AHEAD
#define NGF_COMP_AHEAD 4
#endif
constexpr int kCompAhead = NGF_COMP_AHEAD;
#ifndef NGF_COMP_LANES
#define NGF_COMP_LANES 32
#endif
constexpr int kCompLanes = NGF_COMP_LANES;                   // lanes per ray.  Round 5: 32 with the colours of active samples only (52 us; 16 lanes: 61, 64 lanes: 68; with every colour read 65 / 64 / 79; 8 blocks ahead instead of 4: 50.5).  Round 3: 16 (eight measured slower: 71 vs 64 us)
// PHASE 0: the fused step (residual against T.target, loss, backward).  PHASE 1: the forward call of the two-call form -- the same sums, then
// rgb_map (clamped) and depth_map are written and the kernel ends.  PHASE 2: its backward call -- the same sums again (they decide where the
// clamp passes a gradient), G = T.d_rgb where it does.
template <int PHASE>
__global__ void __launch_bounds__(64) train_composite_bwd_kernel(const TrainArgs T)
{
    const RenderArgs &A = T.R;
    const int lane = threadIdx.x & 63, seg = lane & (kCompLanes - 1), rl = lane / kCompLanes;
    const int64_t r0 = (int64_t)blockIdx.x * (64 / kCompLanes) + rl;
    const bool live = r0 < A.n;
    const int64_t r = live ? r0 : A.n - 1;
    const float bg = A.white_bg ? 1.0f : 0.0f;
    float acc = 0.0f, rgb[3] = {0.0f, 0.0f, 0.0f}, dep = 0.0f;
    const float tmin = ray_tmin(A, r);
    const float jit = A.jitter ? A.jitter[r] : 0.0f;
    // one wave per SIMD: the loads of kCompAhead blocks are requested together (as in the scan); the colours are read whether the
    // sample is active or not (inactive entries are never written: whatever they hold is dropped by the select below)
    for (int i0 = seg; i0 < A.S; i0 += kCompLanes * kCompAhead) {
        float wv[kCompAhead], cv[kCompAhead][3];
#pragma unroll
        for (int u = 0; u < kCompAhead; ++u) {
            const int i = i0 + kCompLanes * u;
            const int64_t idx = r * A.S + (i < A.S ? i : 0);
            wv[u] = i < A.S ? T.w[idx] : 0.0f;
        }
        // the colours of the ACTIVE samples only (4-5 % of the pairs, in runs along the ray: most 64-byte lines of T.c are never touched): the kernel
        // moves ~190 MB per step at ~3 TB/s, a quarter of it were colours nothing reads
#pragma unroll
        for (int u = 0; u < kCompAhead; ++u) {
            const int i = i0 + kCompLanes * u;
            const int64_t idx = r * A.S + (i < A.S ? i : 0);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) cv[u][ch] = wv[u] > A.thr ? T.c[idx * 3 + ch] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < kCompAhead; ++u) {
            if (i0 + kCompLanes * u >= A.S) continue;
            acc += wv[u];
            if (PHASE == 1) dep += wv[u] * (tmin + A.step * ((float)(i0 + kCompLanes * u) + jit));      // depth_map: sum of weight * z_vals (FieldBase.py:305)
            if (wv[u] > A.thr) {
#pragma unroll
                for (int ch = 0; ch < 3; ++ch) rgb[ch] += wv[u] * cv[u][ch];
            }
        }
    }
#pragma unroll
    for (int d = 1; d < kCompLanes; d <<= 1) {
        acc += __shfl_xor(acc, d);
        if (PHASE == 1) dep += __shfl_xor(dep, d);
#pragma unroll
        for (int ch = 0; ch < 3; ++ch) rgb[ch] += __shfl_xor(rgb[ch], d);
    }
    double sq = 0.0;
    float G[3];
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float pre = rgb[ch];
        if (A.white_bg) pre = pre + (1.0f - acc);
        const float out = fminf(fmaxf(pre, 0.0f), 1.0f);
        if (PHASE == 1) {
            if (live && seg == 0) T.rgb_out[r * 3 + ch] = out;
            G[ch] = 0.0f;
            continue;
        }
        const bool pass = (pre >= 0.0f) & (pre <= 1.0f);                            // clamp passes the gradient on [0,1]
        if (PHASE == 2) {
            G[ch] = pass ? T.d_rgb[r * 3 + ch] : 0.0f;
        } else {
            const float res = out - T.target[r * 3 + ch];
            if (live && seg == 0) sq += (double)(res * res);
            G[ch] = pass ? 2.0f * res * T.inv_count : 0.0f;
        }
        if (live && seg == 0) T.G[r * 3 + ch] = G[ch];
    }
    if (PHASE == 1) {
        if (live && seg == 0) T.depth_out[r] = dep + (1.0f - acc) * A.rays[r * 6 + 5];      // + (1 - acc_map) * rays_chunk[..., -1] (FieldBase.py:306)
        return;
    }
    const float gbg = (G[0] + G[1] + G[2]) * bg;
    double carry = 0.0;                                       // sum of the terms of all steps above the current block
    for (int j0 = ((A.S - 1) / kCompLanes) * kCompLanes; j0 >= 0; j0 -= kCompLanes * kCompAhead) {
        float xv[kCompAhead], sv[kCompAhead], tv[kCompAhead], wv[kCompAhead], cv[kCompAhead][3];
#pragma unroll
        for (int u = 0; u < kCompAhead; ++u) {
            const int i = j0 - kCompLanes * u + seg;
            const bool inb = (j0 - kCompLanes * u >= 0) && (i < A.S);
            const int64_t idx = r * A.S + (inb ? i : 0);
            xv[u] = inb ? T.et[idx] : 1.0f;
            sv[u] = inb ? T.sg[idx] : 0.0f;
            tv[u] = inb ? T.dx[idx] : 0.0f;
            wv[u] = inb ? T.w[idx] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < kCompAhead; ++u) {
            const int i = j0 - kCompLanes * u + seg;
            const bool inb = (j0 - kCompLanes * u >= 0) && (i < A.S);
            const int64_t idx = r * A.S + (inb ? i : 0);
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) cv[u][ch] = wv[u] > A.thr ? T.c[idx * 3 + ch] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < kCompAhead; ++u) {
            const int i0 = j0 - kCompLanes * u;
            if (i0 < 0) break;                                           // wave-uniform
            const int i = i0 + seg;
            const bool inb = i < A.S;
            const int64_t idx = r * A.S + (inb ? i : 0);
            const float z = tmin + A.step * ((float)i + jit);
            const float zn = tmin + A.step * ((float)(i + 1) + jit);
            const float delta = ((i < A.S - 1) ? (zn - z) : 0.0f) * A.dscale;
            const float e = xv[u];
            const float alpha = 1.0f - e;
            const float Ti = tv[u];
            const float w = wv[u];
            float dw = -gbg;
            if (w > A.thr) dw += G[0] * cv[u][0] + G[1] * cv[u][1] + G[2] * cv[u][2];
            const double term = inb ? (double)((dw * alpha) * Ti) : 0.0;     // autograd's order: dL/dT_j = dL/dw_j alpha_j, then times T_j
            double incl = term;                                              // inclusive suffix sum over the row: steps >= seg
#pragma unroll
            for (int d = 1; d < kCompLanes; d <<= 1) {
                const double o = __shfl_down(incl, d, kCompLanes);
                incl += (seg + d < kCompLanes) ? o : 0.0;
            }
            const double suffix = carry + (incl - term);
            const float keep = (1.0f - alpha) + 1e-10f;
            const float dalpha = dw * Ti - (float)suffix / keep;
            const float dsigma = dalpha * delta * e;                        // d alpha / d sigma = delta exp(-sigma delta)
            const float sig = sv[u];                                         // softplus'; 0 where the sample is invalid
            if (live && inb) T.dx[idx] = (sig == 0.0f) ? 0.0f : dsigma * sig;
            carry += __shfl(incl, rl * kCompLanes);
        }
    }
    if (PHASE == 0) {      // one double atomic per wave for the loss
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) sq += __shfl_xor(sq, s);
        if (lane == 0) atomicAdd(T.loss, sq);
    }
}

// ---- 5. colour backward over the active list --------------------------------------------------------------------------------
__global__ void __launch_bounds__(kTrainWavesBwd * 64) __attribute__((amdgpu_waves_per_eu(1, 2))) train_color_bwd_kernel(const TrainArgs T)
{
    NGF_KARG_CONTRACT_T(&train_color_bwd_kernel, TrainArgs);      // TrainArgs starts with its RenderArgs (static_assert above)
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const RenderArgs &A = T.R;
    for (int i = threadIdx.x; i < kBwdImage; i += blockDim.x) smem[i] = T.bwd_image[i];
    __syncthreads();
    const float *img = smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    float *H1t = smem + ((kBwdImage + 3) & ~3) + wave * kBwdTileFloats, *D2t = H1t + 64 * kTs, *DFt = H1t, *H2t = H1t + 16 * kDfStride;
    float *D1t = H2t;                        // h2 is dead once d2 exists
    const int chunk_n = chunk_rows(T);
    const int passes = (chunk_n + 15) / 16;
    float gb1 = 0.0f, gb2 = 0.0f, gb3[3] = {0.0f, 0.0f, 0.0f};      // bias gradients: lane l sums unit l of d1 / d2 over this wave's samples
    const bool prof = (A.ablate & (1 << 20)) != 0;
    unsigned long long pc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, t0 = 0;
#define NGF_SEC(k) if (prof) { const unsigned long long t1 = __builtin_readcyclecounter(); pc[k] += t1 - t0; t0 = t1; }
    // A pass's inputs -- its list entries, its 32 activation rows, the colours and ray gradients behind the list entries -- are requested
    // during the PREVIOUS pass of the wave (1.75 waves per SIMD hide little: as plain loads at the head of the pass they were a quarter of
    // the kernel, profiles/exp_train_sections.py).  Every request is unconditional (past the wave's last pass: that pass again).
    const int stride = gridDim.x * kTrainWavesBwd;
    auto want_list = [&](int ps, int &r_, int &i_, float &w_) {
        const int lc = ps * 16 + n;
        const int64_t sl = T.chunk_base + (lc < chunk_n ? lc : 0);
        r_ = T.list[2 * sl]; i_ = T.list[2 * sl + 1]; w_ = T.list_w[sl];
    };
    auto want_rows = [&](int ps, float (&va)[16], float (&vb)[16]) {          // rows ps * 16 + s, 64 lanes on the 64 units of a row
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
            const int rw = ps * 16 + s_;
            const size_t o = (size_t)(rw < chunk_n ? rw : 0) * 64 + lane;
            va[s_] = T.H1[o]; vb[s_] = T.H2[o];
        }
    };
    auto want_cg = [&](int r_, int i_, float (&c_)[3], float (&g_)[3]) {
        const float *cc = T.c + ((int64_t)r_ * A.S + i_) * 3;
#pragma unroll
        for (int jj = 0; jj < 3; ++jj) { c_[jj] = cc[jj]; g_[jj] = T.G[(int64_t)r_ * 3 + jj]; }
    };
    int pass = blockIdx.x * kTrainWavesBwd + wave;
    int r_cur = 0, i_cur = 0;
    float w_cur = 0.0f, va[16], vb[16], c_cur[3] = {0.0f, 0.0f, 0.0f}, g_cur[3] = {0.0f, 0.0f, 0.0f};
    if (pass < passes) {
        want_list(pass, r_cur, i_cur, w_cur);
        want_rows(pass, va, vb);
        want_cg(r_cur, i_cur, c_cur, g_cur);
    }
    for (; pass < passes; pass += stride) {
        const int local = pass * 16 + n;
        const bool live = local < chunk_n;
        const int64_t row = live ? local : 0;
        const int64_t r = r_cur;
        const int i = i_cur;
        const float w = live ? w_cur : 0.0f;
        if (prof) t0 = __builtin_readcyclecounter();
        // the sample's coordinates (rays -> gauge planes: two dependent global reads) are requested first: they are not read before the
        // d loss / d t section
        float t[6], xn[3];
        list_sample_coords(A, r, i, t, xn);
        // the activation rows of this pass -> the wave's tiles (zeros for the entries past the chunk's end)
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
            const bool ok = pass * 16 + s_ < chunk_n;
            H1t[lane * kTs + s_] = ok ? va[s_] : 0.0f;
        }
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
            const bool ok = pass * 16 + s_ < chunk_n;
            H2t[lane * kTs + s_] = ok ? vb[s_] : 0.0f;
        }
        // d3 = dL/dc * sigmoid' ; dL/dc = G_ray * w
        float d3[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) d3[j] = live ? g_cur[j] * w * (c_cur[j] * (1.0f - c_cur[j])) : 0.0f;
        const int next = pass + stride < passes ? pass + stride : pass;
        int r_nx, i_nx;
        float w_nx;
        want_list(next, r_nx, i_nx, w_nx);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        NGF_SEC(0)
        // d2 = (W3^T d3) * [h2 > 0] on the VALU (3 terms per hidden unit)
        for (int k = 16 * q; k < 16 * q + 16; ++k) {
            float s = img[kBwdW3 + k] * d3[0] + img[kBwdW3 + 64 + k] * d3[1] + img[kBwdW3 + 128 + k] * d3[2];
            D2t[k * kTs + n] = H2t[k * kTs + n] > 0.0f ? s : 0.0f;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        dense16<false, 2, 64, 64, 64>(img + kBwdW2T, kLd2, 64, nullptr, D2t, D1t, H1t, lane);              // d1 = (W2^T d2) * [h1 > 0]
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        NGF_SEC(1)
        // rows for the weight-gradient GEMMs (d2 leaves LDS here: its tile and h1's are about to become DF^T)
        if (live && q == 0) {
            float *d = T.D3 + row * 16;
            d[0] = d3[0]; d[1] = d3[1]; d[2] = d3[2];
#pragma unroll
            for (int j = 3; j < 16; ++j) d[j] = 0.0f;
        }
        tile_to_rows(D2t, 64, T.D2, 64, row, live, lane);
        tile_to_rows(D1t, 64, T.D1, 64, row, live, lane);
        // column sums for the biases while the tiles are here (dead samples hold zeros)
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            gb2 += D2t[lane * kTs + c];
            gb1 += D1t[lane * kTs + c];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) gb3[j] += d3[j];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        NGF_SEC(2)
        dense16<false, 0, kFeat, 64, 64, kDfStride>(img + kBwdW1T, kLd2, kFeat, nullptr, D1t, DFt, nullptr, lane);   // df = W1'^T d1
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        NGF_SEC(3)
        // d loss / d t through the bilinear cell (lane (q, n): channels 12q..12q+11 of every plane of sample n) and the tap table
        float dt[6];
        int bin[3] = {-1, -1, -1};
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const Tex &tx = A.app[p];
            BilG b = bilg_setup(t[2 * p], t[2 * p + 1], tx);
            const size_t base = (size_t)b.idx * 48 + 12 * q;
            const f32x4 *q00 = reinterpret_cast<const f32x4 *>(tx.p + base);
            const f32x4 *q01 = q00 + (size_t)tx.stride * 12;
            if (q == 0 && live) {          // the pair (plane, sample): its cell, its tap weights, its place in its bin (section 5b)
                const bool any = (b.w00 != 0.0f) | (b.w10 != 0.0f) | (b.w01 != 0.0f) | (b.w11 != 0.0f);
                const size_t pi = (size_t)p * T.bin_cap + row;
                T.pair_cell[pi] = b.cx | (b.cy << 16);
                *reinterpret_cast<f32x4 *>(T.pair_w + pi * 4) = f32x4{b.w00, b.w10, b.w01, b.w11};
                bin[p] = any ? T.bin_base[p] + (b.cy >> 3) * T.bin_nbx[p] + (b.cx >> 3) : -1;
            }
            float du = 0.0f, dv = 0.0f;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                f32x4 v00 = q00[j], v10 = q00[12 + j], v01 = q01[j], v11 = q01[12 + j];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float g = DFt[n * kDfStride + p * 48 + 12 * q + 4 * j + e];
                    du += g * (b.wy0 * (v10[e] - v00[e]) + b.wy1 * (v11[e] - v01[e]));
                    dv += g * (b.wx0 * (v01[e] - v00[e]) + b.wx1 * (v11[e] - v10[e]));
                }
            }
            du *= b.sx; dv *= b.sy;
            du += __shfl_xor(du, 16); du += __shfl_xor(du, 32);
            dv += __shfl_xor(dv, 16); dv += __shfl_xor(dv, 32);
            dt[2 * p] = du; dt[2 * p + 1] = dv;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        NGF_SEC(4)
        // the next pass of this wave: its rows now (the 32 row registers are free from here: requested earlier in the pass they spill), its
        // colours / ray gradients at the end of the pass (the list entries they hang on were requested at its head)
        want_rows(next, va, vb);
        // feature gradients: the rows leave for the bin-ordered scatter (train_bin_scatter_kernel) -- DF^T is sample-major, 64 lanes
        // write 64 consecutive floats of one row
        {
            const unsigned livem = (unsigned)(__ballot(live) & 0xffffull);        // lanes 0..15 carry q == 0: bit s = sample s of the pass
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                if (!((livem >> s) & 1u)) continue;      // wave-uniform
                float *dst = T.DF + (size_t)(pass * 16 + s) * kFeat;
                dst[lane] = DFt[s * kDfStride + lane];
                dst[64 + lane] = DFt[s * kDfStride + 64 + lane];
                if (lane < kFeat - 128) dst[128 + lane] = DFt[s * kDfStride + 128 + lane];
            }
        }
        // the pairs' places in their bins: the 16 samples of a pass mostly share a bin, so the first lane of every distinct bin asks for the
        // places of all its lanes with ONE returning atomic (one per pair made the hot bins' counters the bottleneck of the kernel)
        // (reading the answers one pass later -- a returning atomic is a round trip past the L2 -- was tried: no gain, 16 bytes of spills)
        {
            int leader[3], before[3], base[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const int mine = (q == 0 && live) ? bin[p] : -1;
                int total = 0;
                leader[p] = 0; before[p] = 0;
#pragma unroll
                for (int j = 15; j >= 0; --j) {
                    const int bj = __builtin_amdgcn_readlane(mine, j);
                    const bool same = (bj == mine);
                    leader[p] = same ? j : leader[p];
                    before[p] += (same && j < lane) ? 1 : 0;
                    total += same ? 1 : 0;
                }
                base[p] = 0;
                if (mine >= 0 && leader[p] == lane) base[p] = atomicAdd(T.bin_count + mine, total);       // the three requests travel together
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const int b0 = __shfl(base[p], leader[p]);
                if (q == 0 && live) T.pair_rank[(size_t)p * T.bin_cap + row] = bin[p] >= 0 ? b0 + before[p] : -1;
            }
        }
        if (live && q == 0) {
            float *d = T.dt + ((int64_t)r * A.S + i) * 6;
#pragma unroll
            for (int k = 0; k < 6; ++k) d[k] = dt[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        NGF_SEC(5)
        if (prof) pc[7] += 1;
        r_cur = r_nx; i_cur = i_nx; w_cur = w_nx;
        want_cg(r_cur, i_cur, c_cur, g_cur);
    }
#undef NGF_SEC
    if (prof && lane == 0)
        for (int k = 0; k < 8; ++k) atomicAdd(T.prof + k, pc[k]);
    atomicAdd(T.g_b1 + lane, gb1);
    atomicAdd(T.g_b2 + lane, gb2);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        float v = q == 0 ? gb3[j] : 0.0f;                 // the four quarters of a sample carry the same d3
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) v += __shfl_xor(v, d);
        if (lane == 0) atomicAdd(T.g_b3 + j, v);
    }
}

// ---- 5b. colour-plane scatter, ordered by bins ------------------------------------------------------------------------------------
// A global float atomic costs one transaction per (instruction, 64-byte line) at 21 G transactions/s for the whole device, whatever its
// scope and wherever the line lives (profiles/micro/atomic_cost.hip, atomic_scope.hip) -- and a tap of a colour plane is 48 channels =
// three lines.  Round 2 sent the taps from the colour backward itself through a 16-texel cache per wave: 4.4 M transactions per
// iteration, 0.21 ms of floor under a 0.29 ms section.  The samples of a PASS share few texels (16 consecutive active steps of one
// ray); the samples of the BATCH share nearly all of them (165 k samples x 4 taps on 65 k texels per plane).  So the pairs
// (plane, sample) are ordered by bin -- a bin = the cells of one 8x8 block of a plane -- with a counting sort whose counting step
// rides in the colour backward (one returning atomic per pair on the bin's counter: consecutive samples mostly share the bin, i.e.
// the line), and one wave per unit (<= kBinChunk pairs of one bin) sums the block's 9x9 texels in LDS (plain read-add-writes, one
// lane per channel) and sends every touched texel once: ~9x fewer transactions.
constexpr int kBinChunk = 256;                     // pairs per unit
constexpr int kBinTile = 81 * 48;                  // floats of a unit's LDS tile
constexpr int kBinDummy = 10 * 48 + 16;            // ... and of the stretch behind it that lanes 48..63 add into

// exclusive prefix of bin_count -> bin_off, and the unit list; one workgroup
__global__ void __launch_bounds__(1024) train_bin_prefix_kernel(const TrainArgs T)
{
    __shared__ int sc[1024], su[1024];
    const int t = threadIdx.x;
    const int per = (T.nbins + 1023) / 1024;
    const int b0 = t * per, b1 = min(T.nbins, b0 + per);
    int c = 0, u = 0;
    for (int b = b0; b < b1; ++b) {
        const int k = T.bin_count[b];
        c += k;
        u += (k + kBinChunk - 1) / kBinChunk;
    }
    sc[t] = c; su[t] = u;
    __syncthreads();
    for (int s = 1; s < 1024; s <<= 1) {
        const int ac = t >= s ? sc[t - s] : 0, au = t >= s ? su[t - s] : 0;
        __syncthreads();
        sc[t] += ac; su[t] += au;
        __syncthreads();
    }
    int rc = sc[t] - c, ru = su[t] - u;
    for (int b = b0; b < b1; ++b) {
        const int k = T.bin_count[b];
        T.bin_off[b] = rc;
        T.bin_unit[b] = ru;
        for (int j = 0; j < k; j += kBinChunk) {
            int32_t *un = T.units + (size_t)ru * 3;
            un[0] = b; un[1] = rc + j; un[2] = min(kBinChunk, k - j);
            ++ru;
        }
        rc += k;
    }
    if (t == 1023) { T.bin_off[T.nbins] = sc[1023]; T.bin_unit[T.nbins] = su[1023]; *T.unit_total = su[1023]; }
}

// perm[bin_off[bin] + rank] = sample row, for every pair that has a weight
__global__ void __launch_bounds__(256) train_bin_perm_kernel(const TrainArgs T)
{
    const int rows = chunk_rows(T);
    const int total = 3 * rows;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int p = i / rows, n = i - p * rows;
        const size_t pi = (size_t)p * T.bin_cap + n;
        const int rank = T.pair_rank[pi];
        if (rank < 0) continue;
        const int cell = T.pair_cell[pi];
        const int bin = T.bin_base[p] + ((cell >> 16) >> 3) * T.bin_nbx[p] + ((cell & 0xffff) >> 3);
        T.perm[T.bin_off[bin] + rank] = n;
    }
}

// one wave per unit
__global__ void __launch_bounds__(256) train_bin_scatter_kernel(const TrainArgs T)
{
    __shared__ __attribute__((aligned(16))) float s_tile[4][kBinTile + kBinDummy];
    const RenderArgs &A = T.R;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *tile = s_tile[wave];
    const int nunits = *T.unit_total;
    const bool prof = (A.ablate & (1 << 20)) != 0;             // section clocks -> T.prof[11..15] (profiles/exp_train_sections.py)
    unsigned long long pc[4] = {0, 0, 0, 0}, t0 = 0, n_units = 0;
#define NGF_SEC(k) if (prof) { const unsigned long long t1 = __builtin_readcyclecounter(); pc[k] += t1 - t0; t0 = t1; }
    for (int u = blockIdx.x * 4 + wave; u < nunits; u += gridDim.x * 4) {
        if (prof) { t0 = __builtin_readcyclecounter(); ++n_units; }
        const int bin = __builtin_amdgcn_readfirstlane(T.units[3 * u]), first = __builtin_amdgcn_readfirstlane(T.units[3 * u + 1]),
                  len = __builtin_amdgcn_readfirstlane(T.units[3 * u + 2]);
        const int p = bin >= T.bin_base[2] ? 2 : (bin >= T.bin_base[1] ? 1 : 0);
        const int bl = bin - T.bin_base[p];
        const int by = bl / T.bin_nbx[p], bx = bl - by * T.bin_nbx[p];
        const int x0 = bx * 8, y0 = by * 8;
        for (int e = lane * 4; e < kBinTile; e += 256) *reinterpret_cast<f32x4 *>(tile + e) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        const int32_t *cellp = T.pair_cell + (size_t)p * T.bin_cap;
        const float *wp = T.pair_w + (size_t)p * T.bin_cap * 4;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // lane l holds the records of pairs l, 64 + l, 128 + l, 192 + l (row, tile offset of the cell, weights): two rounds of loads for the
        // whole unit; the rows' gradients then arrive 32 at a time, the next 32 requested before the current 32 are added.  The kernel is
        // bound by the instructions it issues per pair (profiles/exp_train_sections.py), so the inner loops carry no branch: entries past
        // the unit's end have zero weights and the tile's first cell, lanes 48..63 (no channel) add into a dummy stretch
        int rowr[4], offr[4];
        f32x4 wr[4];
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) rowr[jb] = (jb * 64 + lane < len) ? T.perm[first + jb * 64 + lane] : 0;
#ifdef NGF_EXP_SEQ_ROWS
        for (int jb = 0; jb < 4; ++jb) rowr[jb] = (jb * 64 + lane < len) ? (first + jb * 64 + lane) / 3 : 0;      // timing experiment: rows in storage order
#endif
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
            const bool have = jb * 64 + lane < len;
            const int cell = have ? cellp[rowr[jb]] : 0;
            wr[jb] = have ? *reinterpret_cast<const f32x4 *>(wp + (size_t)rowr[jb] * 4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            offr[jb] = have ? (((cell >> 16) - y0) * 9 + ((cell & 0xffff) - x0)) * (48 * 4) : 0;       // bytes
        }
        NGF_SEC(0)
        constexpr int SB = 32;                              // pairs per request group: two groups = 64 rows in flight per wave
        const int nsb = (len + SB - 1) / SB;
        {
            // lanes 48..63: every lane stays active (the record picks below are per-lane selects that lanes 48..63 are read from), they add
            // channel 47's values into a dummy stretch behind the tile (offset multiplier 0)
            const char *dfl = reinterpret_cast<const char *>(T.DF + p * 48 + (lane < 48 ? lane : 47));
            char *tl = reinterpret_cast<char *>(lane < 48 ? tile + lane : tile + kBinTile + (lane - 48));
            const unsigned om = lane < 48 ? 1u : 0u;
            auto pick = [&](const int (&v)[4], int jb) { return jb == 0 ? v[0] : (jb == 1 ? v[1] : (jb == 2 ? v[2] : v[3])); };
            auto load_sb = [&](int sb, float (&g)[SB]) {
                const int rsel = pick(rowr, sb >> 1), l0 = (sb & 1) * SB;
#pragma unroll
                for (int k = 0; k < SB; ++k)
                    g[k] = *reinterpret_cast<const float *>(dfl + (size_t)(unsigned)__builtin_amdgcn_readlane(rsel, l0 + k) * (kFeat * 4));
            };
            auto add_sb = [&](int sb, const float (&g)[SB]) {
                const int jb = sb >> 1, l0 = (sb & 1) * SB;
                const int osel = pick(offr, jb);
                const f32x4 wsel = jb == 0 ? wr[0] : (jb == 1 ? wr[1] : (jb == 2 ? wr[2] : wr[3]));
#pragma unroll
                for (int k = 0; k < SB; ++k) {
                    float *t00 = reinterpret_cast<float *>(tl + __umul24((unsigned)__builtin_amdgcn_readlane(osel, l0 + k), om));
                    const float w00 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wsel[0]), l0 + k));
                    const float w10 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wsel[1]), l0 + k));
                    const float w01 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wsel[2]), l0 + k));
                    const float w11 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(wsel[3]), l0 + k));
                    const float a = t00[0], b = t00[48], c = t00[9 * 48], d = t00[10 * 48];
                    t00[0] = a + w00 * g[k];
                    t00[48] = b + w10 * g[k];
                    t00[9 * 48] = c + w01 * g[k];
                    t00[10 * 48] = d + w11 * g[k];
                }
            };
            // the requests are UNCONDITIONAL (past the unit's end they re-read its last group): a request behind a branch makes the compiler's
            // s_waitcnt for the group being added cover the path on which nothing else is in flight -- i.e. wait for the group just requested
            float ga[SB], gb[SB];
            load_sb(0, ga);
#pragma unroll 1
            for (int sb = 0; sb < nsb; sb += 2) {
                load_sb(min(sb + 1, nsb - 1), gb);
                __builtin_amdgcn_sched_barrier(0);
                add_sb(sb, ga);
                __builtin_amdgcn_sched_barrier(0);
                load_sb(min(sb + 2, nsb - 1), ga);
                __builtin_amdgcn_sched_barrier(0);
                if (sb + 1 < nsb) add_sb(sb + 1, gb);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        NGF_SEC(1)
        // the tile leaves as it is (15.5 KB of plain, contiguous stores): train_bin_gather_kernel adds up the units that hold a texel
        {
            float *slab = T.slab + (size_t)u * kBinTile;
            for (int e = lane * 4; e < kBinTile; e += 256) *reinterpret_cast<f32x4 *>(slab + e) = *reinterpret_cast<const f32x4 *>(tile + e);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        NGF_SEC(2)
    }
#undef NGF_SEC
    if (prof && lane == 0) {
        for (int k = 0; k < 3; ++k) atomicAdd(T.prof + 11 + k, pc[k]);
        atomicAdd(T.prof + 15, n_units);
    }
}

#ifdef NGF_EXPERIMENTS
// The same sums on the matrix pipe, the tile in registers: tile[81 texels][48 channels] += Wm[81][pairs] . G[pairs][48], where column k of
// Wm holds pair k's four tap weights at its four texels and zeros elsewhere.  v_mfma_f32_16x16x4_f32 takes four pairs per instruction
// (6 texel tiles x 3 channel tiles = 18 instructions per four pairs, 72 accumulator registers); lane (kq, j) builds its A operands --
// the weight of pair 4g + kq at texel 16 mt + j -- with four compares per texel tile and loads its B operands straight from the
// gradient rows.  No LDS -- built to get away from the LDS version's ~500-cycle read-add-write chain per pair and from sharing the LDS with
// the density backward's tiles.  MEASURED SLOWER (experiment library only, ablate bit 1 << 18): 111 k cycles per unit against 62 k, the step
// 1.28 ms against 1.24 ms -- four groups of rows in flight at two waves per SIMD (204 registers) do not cover the memory latency, and the
// matrix work itself (18 MFMAs per four pairs) is what the LDS version's chain costs.
__global__ void __launch_bounds__(256) train_bin_scatter_mfma_kernel(const TrainArgs T)
{
    const RenderArgs &A = T.R;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    const int nunits = *T.unit_total;
    const bool prof = (A.ablate & (1 << 20)) != 0;             // section clocks -> T.prof[11..15] (profiles/exp_train_sections.py)
    unsigned long long pc[4] = {0, 0, 0, 0}, t0 = 0, n_units = 0;
#define NGF_SEC(k) if (prof) { const unsigned long long t1 = __builtin_readcyclecounter(); pc[k] += t1 - t0; t0 = t1; }
    for (int u = blockIdx.x * 4 + wave; u < nunits; u += gridDim.x * 4) {
        if (prof) { t0 = __builtin_readcyclecounter(); ++n_units; }
        const int bin = __builtin_amdgcn_readfirstlane(T.units[3 * u]), first = __builtin_amdgcn_readfirstlane(T.units[3 * u + 1]),
                  len = __builtin_amdgcn_readfirstlane(T.units[3 * u + 2]);
        const int p = bin >= T.bin_base[2] ? 2 : (bin >= T.bin_base[1] ? 1 : 0);
        const int bl = bin - T.bin_base[p];
        const int by = bl / T.bin_nbx[p], bx = bl - by * T.bin_nbx[p];
        const int x0 = bx * 8, y0 = by * 8;
        const int32_t *cellp = T.pair_cell + (size_t)p * T.bin_cap;
        const float *wp = T.pair_w + (size_t)p * T.bin_cap * 4;
        const float *dfl = T.DF + p * 48 + j;
        f32x4 acc[6][3];
#pragma unroll
        for (int mt = 0; mt < 6; ++mt)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) acc[mt][nt] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        for (int j0 = 0; j0 < len; j0 += 64) {
            // lane l: the record of pair j0 + l.  Entries past the unit's end: weights 0 at the tile's first texel, gradient row 0 (a row
            // that exists whenever a unit does) -- they add 0 . g, and no loop below needs a branch
            const bool have = j0 + lane < len;
            const int row = have ? T.perm[first + j0 + lane] : 0;
            const int cell = have ? cellp[row] : 0;
            const f32x4 w = have ? *reinterpret_cast<const f32x4 *>(wp + (size_t)row * 4) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            const int c = have ? ((cell >> 16) - y0) * 9 + ((cell & 0xffff) - x0) : 0;
            const int ng = ((min(64, len - j0) + 15) >> 4) << 2;         // groups of four pairs, in fours
            auto fetch = [&](int g, float (&b)[3]) {
                const int r = __shfl(row, 4 * g + kq);
                const float *src = dfl + (size_t)(unsigned)r * kFeat;
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) b[nt] = src[16 * nt];
            };
            auto mma = [&](int g, const float (&b)[3]) {
                const int sl = 4 * g + kq;
                const int cc = __shfl(c, sl);
                const float w00 = __shfl(w[0], sl), w10 = __shfl(w[1], sl), w01 = __shfl(w[2], sl), w11 = __shfl(w[3], sl);
#pragma unroll
                for (int mt = 0; mt < 6; ++mt) {
                    const int d = 16 * mt + j - cc;
                    float a = 0.0f;                  // four independent selects (a nested conditional compiles to divergent branches here)
                    a = d == 0 ? w00 : a;
                    a = d == 1 ? w10 : a;
                    a = d == 9 ? w01 : a;
                    a = d == 10 ? w11 : a;
#pragma unroll
                    for (int nt = 0; nt < 3; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[nt], acc[mt][nt], 0, 0, 0);
                }
            };
            // four groups of rows in flight; every request is unconditional (past the batch's end: its last group again)
            float b0[3], b1[3], b2[3], b3[3];
            fetch(0, b0); fetch(1, b1); fetch(2, b2); fetch(3, b3);
#pragma unroll 1
            for (int g = 0; g < ng; g += 4) {
                mma(g, b0);
                fetch(min(g + 4, 15), b0);
                mma(g + 1, b1);
                fetch(min(g + 5, 15), b1);
                mma(g + 2, b2);
                fetch(min(g + 6, 15), b2);
                mma(g + 3, b3);
                fetch(min(g + 7, 15), b3);
            }
        }
        NGF_SEC(1)
        // accumulator (mt, nt)[r] = texel 16 mt + 4 kq + r, channel 16 nt + j
        {
            float *slab = T.slab + (size_t)u * kBinTile + j;
#pragma unroll
            for (int mt = 0; mt < 6; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int i = 16 * mt + 4 * kq + r;
                    if (i < 81) {
#pragma unroll
                        for (int nt = 0; nt < 3; ++nt) slab[i * 48 + 16 * nt] = acc[mt][nt][r];
                    }
                }
        }
        NGF_SEC(2)
    }
#undef NGF_SEC
    if (prof && lane == 0) {
        for (int k = 0; k < 3; ++k) atomicAdd(T.prof + 11 + k, pc[k]);
        atomicAdd(T.prof + 15, n_units);
    }
}
#endif

// d loss / d colour plane: every texel of the padded planes adds up the tiles that hold it -- the tile of the bin of its own block and,
// on a block's first column / row, of the neighbours before it (a cell's second taps) -- over the units of those bins.  No atomics, and
// every texel is written: the planes' gradients need no zero fill.  One thread per (texel, channel).
__global__ void __launch_bounds__(256) train_bin_gather_kernel(const TrainArgs T)
{
    const RenderArgs &A = T.R;
    const int p = blockIdx.y;
    const Tex &tx = A.app[p];
    const int W2 = tx.stride, H2 = tx.H + 2, nbx = T.bin_nbx[p];
    const int nby = (H2 + 7) >> 3;
    const int64_t total = (int64_t)W2 * H2 * 48;
    float *gp = T.g_app[p];
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int t = (int)(i / 48), c = (int)(i - (int64_t)t * 48);
        const int y = t / W2, x = t - y * W2;
        float s = 0.0f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // k & 1: the block to the left (tile column 8), k & 2: the block above (tile row 8)
            if ((k & 1) && (x & 7)) continue;
            if ((k & 2) && (y & 7)) continue;
            const int bx = (x >> 3) - (k & 1), by = (y >> 3) - (k >> 1);
            if (bx < 0 || by < 0 || bx >= nbx || by >= nby) continue;
            const int bin = T.bin_base[p] + by * nbx + bx;
            const int tcol = (k & 1) ? 8 : (x & 7), trow = (k & 2) ? 8 : (y & 7);
            const float *src = T.slab + (size_t)(trow * 9 + tcol) * 48 + c;
            const int u1 = T.bin_unit[bin + 1];
            for (int u = T.bin_unit[bin]; u < u1; ++u) s += src[(size_t)u * kBinTile];
        }
        gp[i] = T.bin_accumulate ? gp[i] + s : s;
    }
}

// ---- 6. weight gradients: out[M][N] += X^T . Y over `rows` samples (sample-major X [rows, ldx], Y [rows, ldy]) ------------------
// A workgroup walks over 32-sample chunks: both operand slabs go through LDS once (every row of X and Y leaves HBM exactly once;
// a tile-per-wave version re-read X and Y once per output tile, 0.94 GB for the 64 x 144 reduction), its four waves own the
// MT x NT output tiles in registers for the whole walk and add them to `out` (the reference-layout gradient tensor
// [Mvalid][ldo]) at the end.
// The next chunk's rows are requested (into registers) before the current chunk's MFMAs are issued: as load -> store -> sync -> MFMA -> sync
// the kernel read its 137 MB at 1.5 TB/s.  The four GEMMs of a step are ONE launch (blockIdx.y): each alone is a few hundred workgroups.
template <int MT, int NT>
__device__ __forceinline__ void xty_block(const float *__restrict__ X, int ldx, const float *__restrict__ Y, int ldy, int rows, int Mvalid, int Nvalid,
                                          float *out, int ldo, float *sx, float *sy)
{
    constexpr int CH = 32;                                              // samples per chunk = 8 MFMA k-steps
    constexpr int LX = 16 * MT + ((MT & 1) ? 0 : 16), LY = 16 * NT + ((NT & 1) ? 0 : 16);   // row strides == 16 (mod 32): 2 lanes per bank
    constexpr int TILES = MT * NT, PER_WAVE = (TILES + 3) / 4;
    constexpr int EX = CH * 4 * MT, EY = CH * 4 * NT, NX = (EX + 255) / 256, NY = (EY + 255) / 256;      // float4 elements of a slab / per thread
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    f32x4 acc[PER_WAVE];
#pragma unroll
    for (int t = 0; t < PER_WAVE; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int nchunks = (rows + CH - 1) / CH;
    f32x4 px[NX], py[NY];
    auto request = [&](int c) {                 // coalesced float4 loads of chunk c's rows, zeros past the end
        const int r0 = c * CH;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = threadIdx.x + 256 * i, r = e / (4 * MT), q = e - r * (4 * MT);
            px[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (e < EX && r0 + r < rows) px[i] = *reinterpret_cast<const f32x4 *>(X + (size_t)(r0 + r) * ldx + 4 * q);
        }
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int e = threadIdx.x + 256 * i, r = e / (4 * NT), q = e - r * (4 * NT);
            py[i] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (e < EY && r0 + r < rows) py[i] = *reinterpret_cast<const f32x4 *>(Y + (size_t)(r0 + r) * ldy + 4 * q);
        }
    };
    int c = blockIdx.x;
    if (c < nchunks) request(c);
    for (; c < nchunks; c += gridDim.x) {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int e = threadIdx.x + 256 * i, r = e / (4 * MT), q = e - r * (4 * MT);
            if (e < EX) *reinterpret_cast<f32x4 *>(sx + r * LX + 4 * q) = px[i];
        }
#pragma unroll
        for (int i = 0; i < NY; ++i) {
            const int e = threadIdx.x + 256 * i, r = e / (4 * NT), q = e - r * (4 * NT);
            if (e < EY) *reinterpret_cast<f32x4 *>(sy + r * LY + 4 * q) = py[i];
        }
        __syncthreads();
        request(min(c + (int)gridDim.x, nchunks));            // unconditional (past the end: zeros, no loads): see train_bin_scatter_kernel
#pragma unroll
        for (int t = 0; t < PER_WAVE; ++t) {
            const int tile = wave + 4 * t;
            if (tile < TILES) {
                const int m0 = (tile / NT) * 16, n0 = (tile % NT) * 16;
#pragma unroll
                for (int u = 0; u < CH / 4; ++u)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(sx[(4 * u + kq) * LX + m0 + j], sy[(4 * u + kq) * LY + n0 + j], acc[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int t = 0; t < PER_WAVE; ++t) {
        const int tile = wave + 4 * t;
        if (tile < TILES) {
            const int m0 = (tile / NT) * 16, n0 = (tile % NT) * 16;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = m0 + 4 * kq + r, nn = n0 + j;
                if (m < Mvalid && nn < Nvalid) atomicAdd(out + (size_t)m * ldo + nn, acc[t][r]);
            }
        }
    }
}

struct XtyAll {                                  // the four weight-gradient GEMMs of a chunk
    const float *X[4], *Y[4];
    float *out[4];
    int32_t ldx[4], ldy[4], ldo[4], mvalid[4], nvalid[4];
    int32_t rows;
    const int32_t *rows_dev;                    // non-NULL: the row count lives on the device (no host sync on the active count)
};

__global__ void __launch_bounds__(256) xty_all_kernel(const XtyAll G)
{
    __shared__ __attribute__((aligned(16))) float sx[32 * 80], sy[32 * 144];
    const int rows = G.rows_dev ? min(G.rows, *G.rows_dev) : G.rows;
    const int g = blockIdx.y;
    // heaviest first: D1^T F (64 x 144), D2^T H1 (64 x 64), D1^T V (64 x 16), D3^T H2 (16 x 64)
    if (g == 0) xty_block<4, 9>(G.X[0], G.ldx[0], G.Y[0], G.ldy[0], rows, G.mvalid[0], G.nvalid[0], G.out[0], G.ldo[0], sx, sy);
    else if (g == 1) xty_block<4, 4>(G.X[1], G.ldx[1], G.Y[1], G.ldy[1], rows, G.mvalid[1], G.nvalid[1], G.out[1], G.ldo[1], sx, sy);
    else if (g == 2) xty_block<4, 1>(G.X[2], G.ldx[2], G.Y[2], G.ldy[2], rows, G.mvalid[2], G.nvalid[2], G.out[2], G.ldo[2], sx, sy);
    else xty_block<1, 4>(G.X[3], G.ldx[3], G.Y[3], G.ldy[3], rows, G.mvalid[3], G.nvalid[3], G.out[3], G.ldo[3], sx, sy);
}

// ---- 7. density / gauge backward for every valid sample -----------------------------------------------------------------------
// density_decoder is LINEAR, so the gradient of the 16 density channels of a texel is rank one:
//     d loss / d plane_p[c][texel] = wd[16p+c] * D_p[texel],   D_p[texel] = sum over samples of (tap weight * dx)
// and d loss / d wd[16p+c] = sum over texels of D_p[texel] * plane_p[c][texel].  The per-sample kernel therefore scatters ONE
// float per tap (12 atomics instead of 192) and takes the spatial derivative from the wd-projected 1-channel planes Q_p;
// train_density_finish_kernel expands D_p afterwards.
struct ProjectArgs { const float *tex16[3]; const float *wd; int64_t texels[3]; float *q[3]; };      // blockIdx.y = plane
__global__ void __launch_bounds__(256) train_project_density_kernel(const ProjectArgs P)
{
    const int p = blockIdx.y;
    const float *__restrict__ tex16 = P.tex16[p];
    const float *__restrict__ wd = P.wd + 16 * p;
    float *__restrict__ q = P.q[p];
    const int64_t texels = P.texels[p];
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < texels; i += stride) {
        const f32x4 *v = reinterpret_cast<const f32x4 *>(tex16 + i * 16);
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 x = v[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) s = fmaf(wd[4 * j + e], x[e], s);
        }
        q[i] = s;
    }
}

// What the scatter costs (profiles/micro/atomic_cost.hip, lds_atomic_cost.hip):
//   * a global float atomic is priced per (instruction, 64-byte line): 21 G line-transactions/s for the whole device whether 1 or all
//     16 floats of the line take part -- the per-sample scatter of round 1 (one 8- or 16-byte piece per sample and instruction) spent
//     0.77 of its 0.90 ms there;
//   * ds_add_f32 runs at ~3 cycles per ACTIVE LANE (193 cycles for a full wave; ds_add_u32: 5.6), a plain LDS read-add-write at 8.
// So the scalar images D_p and the gauge-gradient planes are BLOCKED -- one 64-byte line = 4x4 texels of D_p / 4x2 texels x 2 channels
// of a gauge plane: the 2x2 footprints along 64 half-texel steps of one ray touch ~10-20 such blocks whatever its direction -- and a
// wave (64 consecutive steps of ONE ray) sums its taps in a wave-private LDS tile over the blocks of its bounding box before it sends
// every touched block as ONE whole-line atomic.  The LDS sums are plain read-add-writes made collision-free: runs of consecutive lanes
// in the SAME cell are added up in registers (segmented scan over the 16-lane DPP rows), and of the run leaders only those that win a
// last-writer-wins election on their cell write in a round (distinct cells => distinct texels for every tap index; normally one round).
#ifndef NGF_SCAT_CAP
#define NGF_SCAT_CAP 64
#endif
constexpr int kScatCap = NGF_SCAT_CAP;          // blocks of a wave tile (4 KB: seven workgroups per CU); a bounding box beyond it is taken in halves
                                                // (128 with the per-tap path as the only fallback was round 2's choice; 32 / 48 / 192 / 256 measured slower)
constexpr int kDensBwdGroupsPerCu = (160 * 1024) / ((NGF_SCAT_CAP * 16 + 256) * 4 * 4 + 64);      // workgroups (4 waves) of train_density_bwd_kernel an LDS holds
constexpr int kScatOwn = 256;                   // election slots per wave
constexpr int kScatWaveFloats = kScatCap * 16 + kScatOwn;

__device__ __forceinline__ int wave_min_i(int v) { return (int)wave_min((float)v); }       // |v| < 2^24
__device__ __forceinline__ int wave_max_i(int v) { return -(int)wave_min((float)-v); }

template <int D>
__device__ __forceinline__ float row_shr_f(float v)      // lane (row, s) <- lane (row, s - D); 0 for s < D
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + D, 0xf, 0xf, true));
}
template <int D>
__device__ __forceinline__ int row_shr_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x110 + D, 0xf, 0xf, true); }

// BYS, CH: a block is 4 x (1 << BYS) texels x CH channels = 16 floats -- (2, 1) the D_p images, (1, 2) the gauge-gradient planes.
template <int BYS, int CH>
__device__ __forceinline__ int blocked_offset(int X, int Y, int bw)
{
    return ((Y >> BYS) * bw + (X >> 2)) * 16 + ((Y & ((1 << BYS) - 1)) * 4 + (X & 3)) * CH;
}

// (X, Y) = padded (column, row) of the cell's first tap, val[tap][channel] the contributions (tap = 2 * dy + dx), bw = blocks per row;
// tile: kScatWaveFloats floats of wave-private LDS.  Every lane of the wave must call it (wave-uniform control flow inside).
template <int BYS, int CH, int N = 4 * CH, int DEPTH = 0>
__device__ __forceinline__ void scatter_blocked(float *tile, int lane, bool work, int X, int Y, float (&val)[N], float *gbuf, int bw, bool lines, unsigned &n_line, unsigned &n_tap)        // lines: profiling -- n_line / n_tap count {whole-line atomics, per-tap fallbacks}.  Two scalars by reference, not an array behind a pointer that may be null: that form kept the counters in scratch (28 B per lane in every launch, round 5)
{
    const int big = 1 << 20;
    const int bx0 = wave_min_i(work ? X >> 2 : big), bx1 = wave_max_i(work ? (X + 1) >> 2 : -big);
    const int by0 = wave_min_i(work ? Y >> BYS : big), by1 = wave_max_i(work ? (Y + 1) >> BYS : -big);
    const int nbx = bx1 - bx0 + 1, nb = nbx * (by1 - by0 + 1);
    if (nb > kScatCap) {                        // wave-uniform
        // a box beyond the tile: the two halves of the lanes in turn (half the steps of a ray: a quarter of the box), twice at most;
        // then the per-tap path (a gauge field that tears the ray apart)
        if constexpr (DEPTH < 2) {
            constexpr int H = 32 >> DEPTH;
            float v0[N], v1[N];
#pragma unroll
            for (int j = 0; j < N; ++j) v0[j] = v1[j] = val[j];
            scatter_blocked<BYS, CH, N, DEPTH + 1>(tile, lane, work && !(lane & H), X, Y, v0, gbuf, bw, lines, n_line, n_tap);
            scatter_blocked<BYS, CH, N, DEPTH + 1>(tile, lane, work && (lane & H), X, Y, v1, gbuf, bw, lines, n_line, n_tap);
            return;
        }
        if (lines) n_tap += 1;
        if (work) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int c = 0; c < CH; ++c)
                    if (val[k * CH + c] != 0.0f) atomicAdd(gbuf + blocked_offset<BYS, CH>(X + (k & 1), Y + (k >> 1), bw) + c, val[k * CH + c]);
        }
        return;
    }
    for (int e = lane * 4; e < nb * 16; e += 256) *reinterpret_cast<f32x4 *>(tile + e) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // runs of lanes in the same cell -> their last lane (inclusive segmented scan inside each 16-lane row)
    const int cell = work ? (Y << 12) | X : -1 - lane;
    const int prev = row_shr_i<1>(cell);
    const bool head = ((lane & 15) == 0) | (cell != prev);
    int f = head ? 1 : 0;
#define NGF_SEG_STEP(D)                                                                                     \
    {                                                                                                       \
        const int fo = row_shr_i<D>(f);                                                                     \
        _Pragma("unroll") for (int j = 0; j < N; ++j) { const float o = row_shr_f<D>(val[j]); val[j] += f ? 0.0f : o; }   \
        f |= fo;                                                                                            \
    }
    NGF_SEG_STEP(1) NGF_SEG_STEP(2) NGF_SEG_STEP(4) NGF_SEG_STEP(8)
#undef NGF_SEG_STEP
    const int next_head = __builtin_amdgcn_update_dpp(1, head ? 1 : 0, 0x101, 0xf, 0xf, false);      // row_shl:1; lane 15 of a row keeps 1
    bool pending = work & (next_head != 0);
    int *own = reinterpret_cast<int *>(tile + kScatCap * 16);
    const int slot = (X + 17 * Y) & (kScatOwn - 1);
    int off[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int tx = X + (k & 1), ty = Y + (k >> 1);
        off[k] = (((ty >> BYS) - by0) * nbx + ((tx >> 2) - bx0)) * 16 + ((ty & ((1 << BYS) - 1)) * 4 + (tx & 3)) * CH;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    while (__any(pending)) {
        if (pending) own[slot] = lane;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const bool win = pending && own[slot] == lane;
        if (win) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                if (CH == 1) {
                    tile[off[k]] += val[k];
                } else {
                    f32x2 *q = reinterpret_cast<f32x2 *>(tile + off[k]);
                    f32x2 t = *q;
                    t[0] += val[2 * k]; t[1] += val[2 * k + 1];
                    *q = t;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            }
        }
        pending = pending & !win;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    const unsigned magic = (65536u + nbx - 1) / nbx;            // b / nbx == (b * magic) >> 16 for b < 128, nbx <= 128
    const int sub = lane >> 4, e = lane & 15;
    for (int b0 = 0; b0 < nb; b0 += 8) {                        // four blocks = four whole lines per instruction, two instructions in flight
        const int ba = b0 + sub, bb = b0 + 4 + sub;
        const float va = ba < nb ? tile[ba * 16 + e] : 0.0f;
        const float vb = bb < nb ? tile[bb * 16 + e] : 0.0f;
        if (lines) {                            // profiling: one transaction per 16-lane group with a non-zero float
            const unsigned long long ma = __ballot(va != 0.0f), mb = __ballot(vb != 0.0f);
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) n_line += (((ma >> (16 * gq)) & 0xffffull) ? 1u : 0u) + (((mb >> (16 * gq)) & 0xffffull) ? 1u : 0u);
        }
        if (va != 0.0f) {
            const int q = (int)(((unsigned)ba * magic) >> 16), r = ba - q * nbx;
            atomicAdd(gbuf + ((size_t)(by0 + q) * bw + (bx0 + r)) * 16 + e, va);
        }
        if (vb != 0.0f) {
            const int q = (int)(((unsigned)bb * magic) >> 16), r = bb - q * nbx;
            atomicAdd(gbuf + ((size_t)(by0 + q) * bw + (bx0 + r)) * 16 + e, vb);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// DENS: the D_p images and d loss / d density bias -- needs d loss / d sigma only, so it runs beside the colour backward.  GAUGE: d loss / d t
// of both paths (the density path's recomputed from the projected planes, the colour path's from T.dt) -> gauge planes, after the colour
// backward.  <true, true> is the whole backward in one pass (single-stream mode).
template <bool DENS, bool GAUGE>
__global__ void __launch_bounds__(256) train_density_bwd_kernel(const TrainArgs T)
{
    NGF_KARG_CONTRACT_T((&train_density_bwd_kernel<DENS, GAUGE>), TrainArgs);
    const RenderArgs &A = T.R;
    __shared__ __attribute__((aligned(16))) float s_tile[4][kScatWaveFloats];
    __shared__ double s_bd;
    if (threadIdx.x == 0) s_bd = 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *tile = s_tile[wave];
    // d loss / d density bias = the sum of d loss / d sigma' over every sample: hundreds of thousands of terms of both signs that cancel to ~1e-3 of
    // their absolute sum.  As float atomics in arrival order two runs of one batch differed by 1e-4 of the result (round 6: a gradient test at 1e-4 failed
    // once in ~10 suite runs); accumulated in double -- per lane, per wave, per workgroup, one double atomic per workgroup into T.loss[1] -- the order no
    // longer reaches the float result, which train_unblock_gauge_kernel rounds once at the end of the step.
    double bsum = 0.0;
    const bool count_lines = (A.ablate & (1 << 21)) != 0;      // profiling: atomic line transactions -> T.prof[8]
    unsigned n_line = 0, n_tap = 0;
    // a wave item = 64 consecutive steps of ONE ray (the last chunk of a ray is short): its taps stay inside a small bounding box
    const int chunks = (A.S + 63) / 64;
    const int64_t items = A.n * chunks;
    for (int64_t item = (int64_t)blockIdx.x * 4 + wave; item < items; item += (int64_t)gridDim.x * 4) {
        const int64_t r = item / chunks;
        const int i0 = (int)(item - r * chunks) * 64 + lane;
        const bool inr = i0 < A.S;
        const int i = inr ? i0 : A.S - 1;
        const int64_t idx = r * A.S + i;
        const float dx = inr ? T.dx[idx] : 0.0f;
        const bool active = GAUGE && inr && (T.w[idx] > A.thr);
        const bool work = (dx != 0.0f) | active;
        if (!__any(work)) continue;
        float t[6], xn[3];
        list_sample_coords(A, r, i, t, xn);
        float dt[6];
        if (DENS) bsum += (double)dx;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const Tex &tx = A.dens[p];
            BilG b = bilg_setup(t[2 * p], t[2 * p + 1], tx);
            const float *q = T.q_dens[p] + b.idx;
            const float v00 = q[0], v10 = q[1], v01 = q[tx.stride], v11 = q[tx.stride + 1];
            dt[2 * p] = dx * (b.wy0 * (v10 - v00) + b.wy1 * (v11 - v01)) * b.sx;
            dt[2 * p + 1] = dx * (b.wx0 * (v01 - v00) + b.wx1 * (v11 - v10)) * b.sy;
            if (DENS && !(A.ablate & 256)) {
                float dw[4] = {b.w00 * dx, b.w10 * dx, b.w01 * dx, b.w11 * dx};
                scatter_blocked<2, 1>(tile, lane, work, b.cx, b.cy, dw, T.d_dens[p], T.d_bw[p], count_lines, n_line, n_tap);
            }
        }
        if (GAUGE && A.mode) {
            if (active) {
                const float *dc = T.dt + idx * 6;
#pragma unroll
                for (int k = 0; k < 6; ++k) dt[k] += dc[k];
            }
            if (!work) {
#pragma unroll
                for (int k = 0; k < 6; ++k) dt[k] = 0.0f;
            }
            // t_xy = ((x+dxy0)+dxz0, (y+dxy1)+dyz0), t_yz = ((y+dyz0)+dxy1, (z+dyz1)+dxz1), t_xz = ((x+dxz0)+dxy0, (z+dxz1)+dyz1)
            const float dg[3][2] = {{dt[0] + dt[4], dt[1] + dt[2]}, {dt[2] + dt[1], dt[3] + dt[5]}, {dt[4] + dt[0], dt[5] + dt[3]}};
            const float u[3] = {xn[0], xn[1], xn[0]}, v[3] = {xn[1], xn[2], xn[2]};
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                Bil b = bil_setup(u[p], v[p], A.gau[p]);
                float gv[8] = {b.w00 * dg[p][0], b.w00 * dg[p][1], b.w10 * dg[p][0], b.w10 * dg[p][1],
                               b.w01 * dg[p][0], b.w01 * dg[p][1], b.w11 * dg[p][0], b.w11 * dg[p][1]};
                if (!(A.ablate & 512)) scatter_blocked<1, 2>(tile, lane, work, b.cx, b.cy, gv, T.g_gau[p], T.g_bw[p], count_lines, n_line, n_tap);
            }
        }
    }
    if (count_lines && lane == 0) { atomicAdd(T.prof + 8, (unsigned long long)n_line); atomicAdd(T.prof + 10, (unsigned long long)n_tap); }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) bsum += __shfl_xor(bsum, s);
    if (lane == 0) atomicAdd(&s_bd, bsum);
    __syncthreads();
    if (DENS && threadIdx.x == 0 && s_bd != 0.0) atomicAdd(T.loss + 1, s_bd);
}

// blocked gauge-gradient plane -> [texel][2] (what the Adam kernel and ngf_train_get_grad read); blockIdx.y = plane
struct UnblockArgs { const float *src[3]; float *dst[3]; int32_t w2[3], h2[3], bw[3]; const double *loss_src; double *loss_dst; int32_t loss_len; double inv_count; const int32_t *overflow; float *g_bd; };   // g_bd: d loss / d density bias, rounded once from the double sum at loss_src[1]   // + the step's loss to the caller's buffer: [0] sum of squared residuals, [1] their mean -- NaN when the step overflowed its speculative rows (train_prefix_kernel): the colour forward of such a step was truncated, its loss means nothing
__global__ void __launch_bounds__(256) train_unblock_gauge_kernel(const UnblockArgs U)
{
    const int p = blockIdx.y;
    const int total = U.w2[p] * U.h2[p];
    if (U.g_bd && blockIdx.x == 0 && p == 0 && threadIdx.x == 0) U.g_bd[0] = (float)U.loss_src[1];
    if (U.loss_dst && blockIdx.x == 0 && p == 0 && threadIdx.x == 0) {
        const double sum = (U.overflow && U.overflow[0]) ? __longlong_as_double(0x7FF8000000000000ll) : *U.loss_src;
        U.loss_dst[0] = sum;
        if (U.loss_len >= 2) U.loss_dst[1] = sum * U.inv_count;
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int y = i / U.w2[p], x = i - y * U.w2[p];
        const f32x2 v = *reinterpret_cast<const f32x2 *>(U.src[p] + blocked_offset<1, 2>(x, y, U.bw[p]));
        *reinterpret_cast<f32x2 *>(U.dst[p] + (size_t)i * 2) = v;
    }
}

// expand the scalar gradient image of plane p: g_dens[texel][c] = wd[c] * D[texel]; g_wd[c] += sum_texels D[texel] * tex16[texel][c]
struct FinishArgs { const float *D[3], *tex16[3]; const float *wd; int32_t w2[3], bw[3]; int64_t texels[3]; float *g_dens[3]; float *g_wd; };   // blockIdx.y = plane
__global__ void __launch_bounds__(256) train_density_finish_kernel(const FinishArgs F)
{
    const int p = blockIdx.y;
    const float *__restrict__ D = F.D[p], *__restrict__ tex16 = F.tex16[p], *__restrict__ wd = F.wd + 16 * p;
    float *__restrict__ g_dens = F.g_dens[p];
    float *g_wd = F.g_wd + 16 * p;
    const int w2 = F.w2[p], bw = F.bw[p];
    const int64_t texels = F.texels[p];
    __shared__ float sh[16];
    if (threadIdx.x < 16) sh[threadIdx.x] = 0.0f;
    __syncthreads();
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.0f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < texels; i += stride) {
        const int ty = (int)(i / w2), tx = (int)(i - (int64_t)ty * w2);
        const float d = D[blocked_offset<2, 1>(tx, ty, bw)];          // D is blocked 4x4 (train_density_bwd_kernel)
        const f32x4 *v = reinterpret_cast<const f32x4 *>(tex16 + i * 16);
        f32x4 *g = reinterpret_cast<f32x4 *>(g_dens + i * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const f32x4 x = v[j];
            g[j] = f32x4{wd[4 * j] * d, wd[4 * j + 1] * d, wd[4 * j + 2] * d, wd[4 * j + 3] * d};
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[4 * j + e] = fmaf(d, x[e], acc[4 * j + e]);
        }
    }
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        float v = acc[c];
#pragma unroll
        for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s);
        if ((threadIdx.x & 63) == 0) atomicAdd(&sh[c], v);
    }
    __syncthreads();
    if (threadIdx.x < 16) atomicAdd(g_wd + threadIdx.x, sh[threadIdx.x]);
}

// ---- 8. torch.optim.Adam (betas, eps, no weight decay, no amsgrad), float32 like the reference ------------------------------------
struct AdamArgs {
    float lr, beta1, beta2, eps;
    float bc1, bc2_sqrt;       // 1 - beta1^t, sqrt(1 - beta2^t)
    float l1;                  // planes: L1_reg_weight / numel, added as l1 * sign(p); 0 otherwise
};

__device__ __forceinline__ float adam_one(float p, float g, float &m, float &v, const AdamArgs &a)
{
    m = m + (1.0f - a.beta1) * (g - m);                   // exp_avg.lerp_(grad, 1 - beta1)
    v = v * a.beta2 + ((1.0f - a.beta2) * g) * g;         // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1-beta2)
    const float step = a.lr / a.bc1;
    const float denom = sqrtf(v) / a.bc2_sqrt + a.eps;
    return p - step * (m / denom);
}

// skip: device flag of the trainer (overflow[0], see train_prefix_kernel) -- non-zero: the step's gradient is incomplete, nothing is updated
__global__ void __launch_bounds__(256) adam_dense_kernel(float *p, const float *g, float *m, float *v, int64_t n, const AdamArgs a, const int32_t *skip)
{
    if (skip && *skip) return;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float mi = m[i], vi = v[i];
        p[i] = adam_one(p[i], g[i], mi, vi, a);
        m[i] = mi; v[i] = vi;
    }
}

// every MLP parameter in one launch: segment k = elements [begin[k], begin[k+1]) of the concatenation, each with its own step count / lr
constexpr int kDenseParams = 9;
struct AdamDenseAll {
    float *p[kDenseParams], *m[kDenseParams], *v[kDenseParams];
    const float *g[kDenseParams];
    int32_t begin[kDenseParams + 1];           // begin[k+1] == begin[k] for a skipped parameter
    AdamArgs a[kDenseParams];
    const int32_t *skip;
};

__global__ void __launch_bounds__(256) adam_dense_all_kernel(const AdamDenseAll D)
{
    if (D.skip && *D.skip) return;
    const int total = D.begin[kDenseParams];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int k = 0;
#pragma unroll
        for (int j = 1; j < kDenseParams; ++j) k += (i >= D.begin[j]) ? 1 : 0;
        const int e = i - D.begin[k];
        float mi = D.m[k][e], vi = D.v[k][e];
        D.p[k][e] = adam_one(D.p[k][e], D.g[k][e], mi, vi, D.a[k]);
        D.m[k][e] = mi; D.v[k][e] = vi;
    }
}

// NCHW parameter [C,H,W]; gradient AND the trainer's packed copy of the parameter in the packed layouts: channels [0,CS) in ga / ta
// (CS per texel), [CS,C) in gb / tb (C-CS per texel).  A workgroup takes 64 texels of one row: the packed gradient rows go through LDS
// so that every global access is a contiguous run (the parameter / moments along x, the packed arrays along the channels), and the
// updated values leave the same way into the packed copy -- the next step's kernels read it without a re-pack.
// EXT (round 6, ngf_train_adam_ext): the gradient is the caller's tensor `gext` in the parameter's own NCHW layout (torch's p.grad after
// total_loss.backward(): render gradient + whatever else the caller's loss added, e.g. density_L1) instead of the trainer's packed buffers.
template <int C, int CS, bool EXT = false>
__global__ void __launch_bounds__(256) adam_plane_kernel(float *p, float *m, float *v, int H, int W, const float *ga, const float *gb, float *ta, float *tb,
                                                         const AdamArgs a, const int32_t *skip, const float *gext = nullptr)
{
    if (skip && *skip) return;
    constexpr int CB = C - CS;
    __shared__ float sg[64 * (C + 1)];
    const int tiles_x = (W + 63) / 64;
    for (int tile = blockIdx.x; tile < H * tiles_x; tile += gridDim.x) {
        const int y = tile / tiles_x, x0 = (tile - y * tiles_x) * 64, nx = min(64, W - x0);
        const size_t texel0 = (size_t)(y + 1) * (W + 2) + (x0 + 1);
        if (!EXT) {
            for (int e = threadIdx.x; e < nx * CS; e += 256) sg[(e / CS) * (C + 1) + e % CS] = ga[texel0 * CS + e];
            if (CB > 0)
                for (int e = threadIdx.x; e < nx * CB; e += 256) sg[(e / (CB > 0 ? CB : 1)) * (C + 1) + CS + e % (CB > 0 ? CB : 1)] = gb[texel0 * CB + e];
            __syncthreads();
        }
        for (int e = threadIdx.x; e < C * 64; e += 256) {
            const int c = e >> 6, x = e & 63;
            if (x < nx) {
                const size_t i = ((size_t)c * H + y) * W + x0 + x;
                const float pv = p[i];
                const float g = (EXT ? gext[i] : sg[x * (C + 1) + c]) + a.l1 * (pv > 0.0f ? 1.0f : (pv < 0.0f ? -1.0f : 0.0f));
                float mi = m[i], vi = v[i];
                const float pn = adam_one(pv, g, mi, vi, a);
                p[i] = pn; m[i] = mi; v[i] = vi;
                sg[x * (C + 1) + c] = pn;
            }
        }
        __syncthreads();
        for (int e = threadIdx.x; e < nx * CS; e += 256) ta[texel0 * CS + e] = sg[(e / CS) * (C + 1) + e % CS];
        if (CB > 0)
            for (int e = threadIdx.x; e < nx * CB; e += 256) tb[texel0 * CB + e] = sg[(e / (CB > 0 ? CB : 1)) * (C + 1) + CS + e % (CB > 0 ? CB : 1)];
        __syncthreads();
    }
}

// packed gradient -> NCHW through LDS, 64 texels of one row per workgroup pass: contiguous runs on both sides (ngf_train_get_grads: the autograd
// path copies every plane gradient out once per step; unpack_plane_kernel below reads its source 4 bytes per 64 / 192-byte stride)
template <int C, int CS>
__global__ void __launch_bounds__(256) unpack_plane_tiled_kernel(const float *ga, const float *gb, int H, int W, float *dst)
{
    constexpr int CB = C - CS;
    __shared__ float sg[64 * (C + 1)];
    const int tiles_x = (W + 63) / 64;
    for (int tile = blockIdx.x; tile < H * tiles_x; tile += gridDim.x) {
        const int y = tile / tiles_x, x0 = (tile - y * tiles_x) * 64, nx = min(64, W - x0);
        const size_t texel0 = (size_t)(y + 1) * (W + 2) + (x0 + 1);
        for (int e = threadIdx.x; e < nx * CS; e += 256) sg[(e / CS) * (C + 1) + e % CS] = ga[texel0 * CS + e];
        if (CB > 0)
            for (int e = threadIdx.x; e < nx * CB; e += 256) sg[(e / (CB > 0 ? CB : 1)) * (C + 1) + CS + e % (CB > 0 ? CB : 1)] = gb[texel0 * CB + e];
        __syncthreads();
        for (int e = threadIdx.x; e < C * 64; e += 256) {
            const int c = e >> 6, x = e & 63;
            if (x < nx) dst[((size_t)c * H + y) * W + x0 + x] = sg[x * (C + 1) + c];
        }
        __syncthreads();
    }
}

// the MLP parameters' gradients (reference layout already) to the caller's tensors in one launch: segment k = [begin[k], begin[k+1])
struct CopyDenseAll { const float *src[kDenseParams]; float *dst[kDenseParams]; int32_t begin[kDenseParams + 1]; };
__global__ void __launch_bounds__(256) copy_dense_all_kernel(const CopyDenseAll D)
{
    const int total = D.begin[kDenseParams];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        int k = 0;
#pragma unroll
        for (int j = 1; j < kDenseParams; ++j) k += (i >= D.begin[j]) ? 1 : 0;
        D.dst[k][i - D.begin[k]] = D.src[k][i - D.begin[k]];
    }
}

// packed gradient -> NCHW (inspection / tests)
__global__ void __launch_bounds__(256) unpack_plane_kernel(const float *ga, int cs, const float *gb, int C, int H, int W, float *dst)
{
    const int64_t total = (int64_t)C * H * W;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int x = (int)(i % W), y = (int)((i / W) % H), c = (int)(i / ((int64_t)W * H));
        const size_t texel = (size_t)(y + 1) * (W + 2) + (x + 1);
        dst[i] = c < cs ? ga[texel * cs + c] : gb[texel * (C - cs) + (c - cs)];
    }
}

}  // namespace ngf
