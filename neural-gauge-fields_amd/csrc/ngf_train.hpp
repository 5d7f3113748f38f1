// ngf_train.hpp -- one TriPlane training step on the device.
//
// What the reference does per iteration (TriPlane/main.py:264-299): field(rays, is_train=True) (FieldBase.py:251-312),
// rgb MSE + 8e-5 * density_L1 (Field.py:149-152), autograd, torch.optim.Adam over the 8 groups of get_optparam_groups
// (Field.py:34-46), lr decay.  Here the step is a short sequence of kernels over HBM-resident buffers (288 GB: the
// per-sample activations of a 4096-ray batch are simply kept, 1.7 KB per active sample):
//
//   train_density_kernel       every (step, ray) pair in parallel: sample_ray + normalize + compute_gauge + the 48-feature
//                              density fetch -> xs = Linear(48,1) - 10 (pre-softplus), -inf where the sample is invalid
//   train_scan_kernel          one lane per ray, sequential raw2alpha (FieldBase.py:12-19) over the stored xs: weights,
//                              active counts (pass 0) and the (ray, step)-ordered active list (pass 1): deterministic
//   train_fold_kernel          W1' = W1[:, :144] . basis and the padded / transposed LDS images of the colour MLP
//   train_color_fwd_kernel     16 active samples per wave pass on v_mfma_f32_16x16x4_f32: [144 colour features, view] -> 64 ->
//                              64 -> 3 sigmoid; colours to the dense buffer, activations to HBM rows
//   train_composite_bwd_kernel one lane per ray: rgb_map, clamp, residual, loss; then d loss / d xs for every sample from
//                              the closed form of the cumprod backward (two sequential sweeps, no gathers)
//   train_color_bwd_kernel     the data gradients of the colour MLP with the TRANSPOSED weights on the matrix pipe, the
//                              feature gradients scattered into the packed colour planes (float atomics) and d loss / d t
//   xty_block_kernel           weight gradients as sample-reduction GEMMs  dW = Delta^T . In  (MFMA, operand slabs through LDS);
//   train_unfold_kernel        M -> d W1[:, :144], d basis
//   train_density_bwd_kernel   every valid sample: ONE scalar per tap into the density-gradient images (the decoder is linear:
//                              rank-one gradient, expanded by train_density_finish_kernel), d loss / d t -> gauge planes
//   adam_*_kernel              torch.optim.Adam's update; planes read their gradient from the packed layout and add the L1 term
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

namespace ngf {

constexpr int kTrainWaves = 6;                 // waves per workgroup in the colour forward kernel (H2 overwrites the dead feature tile)
constexpr int kTrainWavesBwd = 6;              // ... and in the colour backward kernel (its tiles alias, see kBwdTileFloats)
constexpr int kFeat = 144;                     // colour features (3 planes x 48)
constexpr int kIn1 = 159, kIn1Pad = 160;       // [g(144), view(15)] (+1 zero pad)
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
    float *q_dens[3];        // wd-projected density planes (1 channel, packed): Q_p = sum_c wd[16p+c] plane_p[c]
    float *d_dens[3];        // scalar density-gradient images: D_p[texel] = sum_samples w_tap * dx
    // step-major dense per-sample buffers: index = step * n + ray
    float *xs, *w, *dx;      // [S,n]
    float *c;                // [S,n,3]
    float *dt;               // [S,n,6]   d loss / d t from the colour path (active samples only)
    const float *target;     // [n,3]
    float *G;                // [n,3]   d loss / d rgb_map (before the clamp), 0 where clamped
    int32_t *count;          // [n]     active samples per ray
    int32_t *offset;         // [n+1]   exclusive prefix of count
    int32_t *list;           // [cap,2] (ray, step) in (ray, step) order
    float *list_w;           // [cap]
    // activations of the current chunk, sample-major rows
    float *F, *V, *H1, *H2, *D3, *D2, *D1;          // [chunk, 144|16|64|64|16|64|64]  (V = the 15 view inputs + 0)
    const float *fwd_image, *bwd_image;             // LDS images built by train_fold_kernel
    float *M;                                       // [64,144] = Delta1^T F, accumulated over the chunks
    double *loss;            // [2]: sum of squared residuals, (unused)
    int32_t chunk_base, chunk_n;      // the slice of the active list this launch works on
    const int32_t *n_active_dev;      // non-NULL: the active count lives on the device (offset[n]); chunk_n is then only the capacity and
                                      // the kernels clip it themselves -- no host round trip between the scan and the colour kernels
    int32_t store;           // colour forward: also write F, V, H1, H2 rows of the chunk
    float inv_count;         // 1 / (3 n): the mean of the MSE
};

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
    b.idx = ((int)cy + 1) * t.stride + ((int)cx + 1);
    b.wx0 = in ? wx0 : 0.0f; b.wx1 = in ? wx1 : 0.0f; b.wy0 = in ? wy0 : 0.0f; b.wy1 = in ? wy1 : 0.0f;
    b.w00 = in ? wx0 * wy0 : 0.0f;
    b.w10 = in ? wx1 * wy0 : 0.0f;
    b.w01 = in ? wx0 * wy1 : 0.0f;
    b.w11 = in ? wx1 * wy1 : 0.0f;
    b.sx = in ? t.fw * 0.5f : 0.0f;
    b.sy = in ? t.fh * 0.5f : 0.0f;
    return b;
}

// ---- 1. density features of every (step, ray) pair ---------------------------------------------------------------------
__global__ void __launch_bounds__(256) train_density_kernel(const TrainArgs T)
{
    const RenderArgs &A = T.R;
    const int64_t total = (int64_t)A.S * A.n;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t wi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; wi < total; wi += stride) {
        // a wave takes 64 CONSECUTIVE steps of one ray: its gathers share texels (two steps per texel) instead of touching 64
        // unrelated cells of 64 random training rays; the dense buffers stay step-major for the per-ray sweeps
        const int64_t r = wi / A.S;
        const int i = (int)(wi - r * A.S);
        const int64_t idx = (int64_t)i * A.n + r;
        float xn[3], z, dist, tt[6];
        const bool valid = sample_geometry(A, r, i, xn, z, dist);
        if (!valid) {                               // outside the box / in free space of the alpha mask: sigma = 0, no fetches
            T.xs[idx] = -INFINITY;
            continue;
        }
        triplane_gauge(A, xn, A.mode, tt);          // compute_gauge (Field.py:53-75), identity split when the gauge is off
        float f = 0.0f;
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const Tex &tx = A.dens[p];
            Bil b = bil_setup(tt[2 * p], tt[2 * p + 1], tx);
            const f32x4 *q00 = reinterpret_cast<const f32x4 *>(tx.p + (size_t)b.idx * 16);
            const f32x4 *q01 = q00 + (size_t)tx.stride * 4;
            float d00 = 0.0f, d10 = 0.0f, d01 = 0.0f, d11 = 0.0f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v00 = q00[q], v10 = q00[4 + q], v01 = q01[q], v11 = q01[4 + q];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float w = T.wd[p * 16 + 4 * q + e];
                    d00 = fmaf(w, v00[e], d00);
                    d10 = fmaf(w, v10[e], d10);
                    d01 = fmaf(w, v01[e], d01);
                    d11 = fmaf(w, v11[e], d11);
                }
            }
            f += bil_mix(b, d00, d10, d01, d11);
        }
        f = (f + T.bd[0]) + (-10.0f);
        T.xs[idx] = f;
    }
}

__device__ __forceinline__ float softplus_pre(float u)          // F.softplus, threshold 20; softplus(-inf) = 0
{
    return u > 20.0f ? u : log1pf(expf(u));
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

// ---- 2. raw2alpha per ray; pass 0 counts the active samples, pass 1 writes weights and the ordered active list ----------
// Sixteen lanes per ray (one DPP row), each on one of 16 consecutive steps: the loads and the exp/log of a block of steps run
// in parallel and only the transmittance product is chained lane to lane IN STEP ORDER (row_shr:1), so weights are exactly
// those of the sequential cumprod (FieldBase.py:16) while a 4096-ray batch fills 1024 waves instead of 64.
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
    const float tmin = ray_tmin(A, r);
    const float jit = A.jitter ? A.jitter[r] : 0.0f;
    float Tr = 1.0f;
    int cnt = 0;
    const int base = pass ? T.offset[r] : 0;
    for (int i0 = 0; i0 < A.S; i0 += 16) {
        const int i = i0 + seg;
        const bool inb = i < A.S;
        const int64_t idx = (int64_t)(inb ? i : 0) * A.n + r;
        const float z = tmin + A.step * ((float)i + jit);
        const float zn = tmin + A.step * ((float)(i + 1) + jit);
        const float dist = (i < A.S - 1) ? (zn - z) : 0.0f;
        const float sigma = inb ? softplus_pre(T.xs[idx]) : 0.0f;
        const float alpha = 1.0f - expf(-sigma * (dist * A.dscale));
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
        if (pass && live && inb) {
            T.w[idx] = w;
            T.dx[idx] = Tin;                                    // parked for the compositing backward
            if (active) {
                const int pos = base + cnt + __popc(m16 & ((1u << seg) - 1u));
                T.list[2 * (int64_t)pos] = (int)r;
                T.list[2 * (int64_t)pos + 1] = i;
                T.list_w[pos] = w;
            }
        }
        cnt += __popc(m16);
    }
    if (!pass && live && seg == 0) T.count[r] = cnt;
}

// exclusive prefix of count[0..n) -> offset[0..n]; one block, sequential over chunks of 1024
__global__ void __launch_bounds__(1024) train_prefix_kernel(const int32_t *count, int64_t n, int32_t *offset)
{
    __shared__ int sh[1024];
    __shared__ int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int64_t b0 = 0; b0 < n; b0 += 1024) {
        const int64_t i = b0 + threadIdx.x;
        const int v = i < n ? count[i] : 0;
        sh[threadIdx.x] = v;
        __syncthreads();
        for (int s = 1; s < 1024; s <<= 1) {
            int add = (int)threadIdx.x >= s ? sh[threadIdx.x - s] : 0;
            __syncthreads();
            sh[threadIdx.x] += add;
            __syncthreads();
        }
        if (i < n) offset[i] = carry + sh[threadIdx.x] - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry += sh[1023];
        __syncthreads();
    }
    if (threadIdx.x == 0) offset[n] = carry;
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
    for (int j = 0; j < KS; ++j) b[j] = in[(4 * j + q) * 16 + n];
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
            if (ACT == 2) v = mask[row * 16 + n] > 0.0f ? v : 0.0f;
            if (OUT_T) out[n * OUT_T + row] = v;
            else out[row * 16 + n] = v;
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
        for (int k = lane; k < K; k += 64) rows[sl * ld + k] = tile[k * 16 + s];
    }
}

__device__ __forceinline__ void rows_to_tile(const float *rows, int ld, int K, float *tile, int64_t slot, bool live, int lane)
{
    for (int s = 0; s < 16; ++s) {
        const bool ok = __shfl((int)live, s) != 0;
        const int64_t sl = __shfl((int)(slot & 0x7fffffff), s);
        for (int k = lane; k < K; k += 64) tile[k * 16 + s] = ok ? rows[sl * ld + k] : 0.0f;
    }
}

// gauge-shifted coordinates of a list sample (recomputed: three 2-channel fetches, cheaper than 24 bytes of HBM per valid sample)
__device__ __forceinline__ void list_sample_coords(const RenderArgs &A, int64_t r, int i, float t[6], float xn[3])
{
    float z, dist;
    (void)sample_geometry(A, r, i, xn, z, dist);
    triplane_gauge(A, xn, A.mode, t);
}

constexpr int kFwdTileFloats = (kIn1Pad + 64) * 16;                       // [F; view] (H2 goes there once layer 1 is done and F is stored), H1
constexpr int kDfStride = kFeat + 1;                                       // DF is kept sample-major (bank-conflict-free rows)
// backward tiles per wave: [H1 | D2 | pad] is overwritten by DF^T once d1 exists and d2 / d1 have been written out; [H2, then D1]; tap table
constexpr int kBwdTileFloats = 16 * kDfStride + 64 * 16 + 16 * 16;
static_assert(16 * kDfStride >= 2 * 64 * 16, "DF^T must cover the H1 and D2 tiles it aliases");

// ---- per-step weight images -------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) train_fold_kernel(const TrainArgs T, float *fwd, float *bwd)
{
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
    }
    for (int i = tid; i < 64 * 64; i += nth) {
        const int m = i / 64, k = i % 64;
        const float v = T.w2[m * 64 + k];
        fwd[kFwdW2 + m * kLd2 + k] = v;
        bwd[kBwdW2T + k * kLd2 + m] = v;
    }
    for (int i = tid; i < 192; i += nth) { fwd[kFwdW3 + i] = T.w3[i]; bwd[kBwdW3 + i] = T.w3[i]; }
    for (int i = tid; i < 64; i += nth) { fwd[kFwdB1 + i] = T.b1[i]; fwd[kFwdB2 + i] = T.b2[i]; }
    for (int i = tid; i < 4; i += nth) fwd[kFwdB3 + i] = i < 3 ? T.b3[i] : 0.0f;
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
__global__ void __launch_bounds__(kTrainWaves * 64) train_color_fwd_kernel(const TrainArgs T)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const RenderArgs &A = T.R;
    for (int i = threadIdx.x; i < kFwdImage; i += blockDim.x) smem[i] = T.fwd_image[i];
    __syncthreads();
    const float *img = smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    float *Ft = smem + ((kFwdImage + 3) & ~3) + wave * kFwdTileFloats, *H1t = Ft + kIn1Pad * 16, *H2t = Ft;
    const int chunk_n = chunk_rows(T);
    const int passes = (chunk_n + 15) / 16;
    for (int pass = blockIdx.x * kTrainWaves + wave; pass < passes; pass += gridDim.x * kTrainWaves) {
        const int local = pass * 16 + n;
        const bool live = local < chunk_n;
        const int64_t slot = T.chunk_base + (live ? local : 0);
        const int64_t r = T.list[2 * slot];
        const int i = T.list[2 * slot + 1];
        float t[6], xn[3];
        list_sample_coords(A, r, i, t, xn);
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
                for (int e = 0; e < 4; ++e) Ft[(p * 48 + 12 * q + 4 * j + e) * 16 + n] = live ? bil_mix(b, v00[e], v10[e], v01[e], v11[e]) : 0.0f;
            }
        }
        // the view inputs of layer 1 (networks.py:27-29): rows 144..159 of the input tile
        {
            float d[3] = {A.rays[r * 6 + 3], A.rays[r * 6 + 4], A.rays[r * 6 + 5]}, v[16];
            view_inputs(d, v);
#pragma unroll
            for (int j = 0; j < 4; ++j) Ft[(kFeat + 4 * q + j) * 16 + n] = live ? v[4 * q + j] : 0.0f;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        dense16<false, 1, 64, kIn1Pad, kIn1Pad>(img + kFwdW1, kLd1, 64, img + kFwdB1, Ft, H1t, nullptr, lane);      // relu(W1' f + W1v view + b1)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (T.store) {          // the feature tile leaves LDS now: layer 2 writes its output over it
            const int64_t row = live ? local : 0;
            tile_to_rows(Ft, kFeat, T.F, kFeat, row, live, lane);
            tile_to_rows(Ft + kFeat * 16, 16, T.V, 16, row, live, lane);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        dense16<false, 1, 64, 64, 64>(img + kFwdW2, kLd2, 64, img + kFwdB2, H1t, H2t, nullptr, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // layer 3 + sigmoid on the VALU: lane (q, n) sums its 16 hidden units, then the four quarters meet
        float c[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            float s = 0.0f;
#pragma unroll
            for (int k = 0; k < 16; ++k) s = fmaf(img[kFwdW3 + j * 64 + 16 * q + k], H2t[(16 * q + k) * 16 + n], s);
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            s += img[kFwdB3 + j];
            c[j] = 1.0f / (1.0f + expf(-s));
        }
        if (live && q == 0) {
            float *dst = T.c + ((int64_t)i * A.n + r) * 3;
            dst[0] = c[0]; dst[1] = c[1]; dst[2] = c[2];
        }
        if (T.store) {
            const int64_t row = live ? local : 0;
            tile_to_rows(H1t, 64, T.H1, 64, row, live, lane);
            tile_to_rows(H2t, 64, T.H2, 64, row, live, lane);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// ---- 4. compositing forward tail + backward to the pre-softplus density of every sample ------------------------------------
// Sixteen lanes per ray as in the scan.  d loss / d alpha_i = dL/dw_i T_i - (sum_{j>i} w_j dL/dw_j) / (1 - alpha_i + 1e-10) with
// dL/dw_j = G . (c_j [active] - bg) is the cumprod backward; the suffix sum is accumulated from the END of the ray in float64
// (what ATen's reverse cumsum does on the CPU), T_i was parked in dx by the scan.
__global__ void __launch_bounds__(64) train_composite_bwd_kernel(const TrainArgs T)
{
    const RenderArgs &A = T.R;
    const int lane = threadIdx.x & 63, seg = lane & 15, rl = lane >> 4;
    const int64_t r0 = (int64_t)blockIdx.x * 4 + rl;
    const bool live = r0 < A.n;
    const int64_t r = live ? r0 : A.n - 1;
    const float bg = A.white_bg ? 1.0f : 0.0f;
    float acc = 0.0f, rgb[3] = {0.0f, 0.0f, 0.0f};
    for (int i = seg; i < A.S; i += 16) {
        const int64_t idx = (int64_t)i * A.n + r;
        const float w = T.w[idx];
        acc += w;
        if (w > A.thr) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) rgb[ch] += w * T.c[idx * 3 + ch];
        }
    }
#pragma unroll
    for (int d = 1; d < 16; d <<= 1) {
        acc += __shfl_xor(acc, d);
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
        const float res = out - T.target[r * 3 + ch];
        if (live && seg == 0) sq += (double)(res * res);
        G[ch] = (pre >= 0.0f && pre <= 1.0f) ? 2.0f * res * T.inv_count : 0.0f;      // clamp passes the gradient on [0,1]
        if (live && seg == 0) T.G[r * 3 + ch] = G[ch];
    }
    const float gbg = (G[0] + G[1] + G[2]) * bg;
    const float tmin = ray_tmin(A, r);
    const float jit = A.jitter ? A.jitter[r] : 0.0f;
    double carry = 0.0;                                       // sum of the terms of all steps above the current block
    for (int i0 = ((A.S - 1) / 16) * 16; i0 >= 0; i0 -= 16) {
        const int i = i0 + seg;
        const bool inb = i < A.S;
        const int64_t idx = (int64_t)(inb ? i : 0) * A.n + r;
        const float x = inb ? T.xs[idx] : -INFINITY;
        const float z = tmin + A.step * ((float)i + jit);
        const float zn = tmin + A.step * ((float)(i + 1) + jit);
        const float delta = ((i < A.S - 1) ? (zn - z) : 0.0f) * A.dscale;
        const float e = expf(-softplus_pre(x) * delta);
        const float alpha = 1.0f - e;
        const float Ti = inb ? T.dx[idx] : 0.0f;
        const float w = inb ? T.w[idx] : 0.0f;
        float dw = -gbg;
        if (w > A.thr) dw += G[0] * T.c[idx * 3] + G[1] * T.c[idx * 3 + 1] + G[2] * T.c[idx * 3 + 2];
        const double term = inb ? (double)((dw * alpha) * Ti) : 0.0;     // autograd's order: dL/dT_j = dL/dw_j alpha_j, then times T_j
        double incl = term;                                              // inclusive suffix sum over the row: steps >= seg
#pragma unroll
        for (int d = 1; d < 16; d <<= 1) {
            const double o = __shfl_down(incl, d, 16);
            incl += (seg + d < 16) ? o : 0.0;
        }
        const double suffix = carry + (incl - term);
        const float keep = (1.0f - alpha) + 1e-10f;
        const float dalpha = dw * Ti - (float)suffix / keep;
        const float dsigma = dalpha * delta * e;                        // d alpha / d sigma = delta exp(-sigma delta)
        const float sig = x > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-x));   // softplus'
        if (live && inb) T.dx[idx] = (x == -INFINITY) ? 0.0f : dsigma * sig;
        carry += __shfl(incl, rl * 16);
    }
    // one double atomic per wave for the loss
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) sq += __shfl_xor(sq, s);
    if (lane == 0) atomicAdd(T.loss, sq);
}

// ---- 5. colour backward over the active list --------------------------------------------------------------------------------
__global__ void __launch_bounds__(kTrainWavesBwd * 64) train_color_bwd_kernel(const TrainArgs T)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const RenderArgs &A = T.R;
    for (int i = threadIdx.x; i < kBwdImage; i += blockDim.x) smem[i] = T.bwd_image[i];
    __syncthreads();
    const float *img = smem;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n = lane & 15, q = lane >> 4;
    float *H1t = smem + ((kBwdImage + 3) & ~3) + wave * kBwdTileFloats, *D2t = H1t + 64 * 16, *DFt = H1t, *H2t = H1t + 16 * kDfStride,
          *tap = H2t + 64 * 16;              // tap[s][p] = {texel index, w00, w10, w01, w11}
    float *D1t = H2t;                        // h2 is dead once d2 exists
    const int chunk_n = chunk_rows(T);
    const int passes = (chunk_n + 15) / 16;
    for (int pass = blockIdx.x * kTrainWavesBwd + wave; pass < passes; pass += gridDim.x * kTrainWavesBwd) {
        const int local = pass * 16 + n;
        const bool live = local < chunk_n;
        const int64_t row = live ? local : 0;
        const int64_t slot = T.chunk_base + row;
        const int64_t r = T.list[2 * slot];
        const int i = T.list[2 * slot + 1];
        const float w = live ? T.list_w[slot] : 0.0f;
        rows_to_tile(T.H1, 64, 64, H1t, row, live, lane);
        rows_to_tile(T.H2, 64, 64, H2t, row, live, lane);
        // d3 = dL/dc * sigmoid' ; dL/dc = G_ray * w
        float d3[3];
        {
            const float *cc = T.c + ((int64_t)i * A.n + r) * 3;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const float c = cc[j];
                d3[j] = live ? T.G[r * 3 + j] * w * (c * (1.0f - c)) : 0.0f;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // d2 = (W3^T d3) * [h2 > 0] on the VALU (3 terms per hidden unit)
        for (int k = 16 * q; k < 16 * q + 16; ++k) {
            float s = img[kBwdW3 + k] * d3[0] + img[kBwdW3 + 64 + k] * d3[1] + img[kBwdW3 + 128 + k] * d3[2];
            D2t[k * 16 + n] = H2t[k * 16 + n] > 0.0f ? s : 0.0f;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        dense16<false, 2, 64, 64, 64>(img + kBwdW2T, kLd2, 64, nullptr, D2t, D1t, H1t, lane);              // d1 = (W2^T d2) * [h1 > 0]
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // rows for the weight-gradient GEMMs (d2 leaves LDS here: its tile and h1's are about to become DF^T)
        if (live && q == 0) {
            float *d = T.D3 + row * 16;
            d[0] = d3[0]; d[1] = d3[1]; d[2] = d3[2];
#pragma unroll
            for (int j = 3; j < 16; ++j) d[j] = 0.0f;
        }
        tile_to_rows(D2t, 64, T.D2, 64, row, live, lane);
        tile_to_rows(D1t, 64, T.D1, 64, row, live, lane);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        dense16<false, 0, kFeat, 64, 64, kDfStride>(img + kBwdW1T, kLd2, kFeat, nullptr, D1t, DFt, nullptr, lane);   // df = W1'^T d1
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // d loss / d t through the bilinear cell (lane (q, n): channels 12q..12q+11 of every plane of sample n) and the tap table
        float t[6], xn[3];
        list_sample_coords(A, r, i, t, xn);
        float dt[6];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const Tex &tx = A.app[p];
            BilG b = bilg_setup(t[2 * p], t[2 * p + 1], tx);
            const size_t base = (size_t)b.idx * 48 + 12 * q;
            const f32x4 *q00 = reinterpret_cast<const f32x4 *>(tx.p + base);
            const f32x4 *q01 = q00 + (size_t)tx.stride * 12;
            if (q == 0) {
                float *tp = tap + (n * 3 + p) * 5;
                tp[0] = __int_as_float(b.idx);
                tp[1] = live ? b.w00 : 0.0f; tp[2] = live ? b.w10 : 0.0f; tp[3] = live ? b.w01 : 0.0f; tp[4] = live ? b.w11 : 0.0f;
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
        // feature gradients -> packed colour planes: the 48 channels of one tap are CONSECUTIVE floats, so the wave scatters
        // (sample, plane) by (sample, plane): 4 taps x 48 channels = 3 instructions of 64 lanes over 12 cache lines, instead of
        // every lane hitting its own line
        // (the list is in (ray, step) order: consecutive samples of a pass are half a texel apart, and runs of samples in the SAME
        // cell are summed in registers first and scattered once)
#pragma unroll 1
        for (int p = 0; p < 3; ++p) {
            const Tex &tx = A.app[p];
            int s0 = 0;
            while (s0 < 16) {
                const int idx = __float_as_int(tap[(s0 * 3 + p) * 5]);
                float acc3[3] = {0.0f, 0.0f, 0.0f};
                int s1 = s0;
                do {
                    const float *tp = tap + (s1 * 3 + p) * 5;
#pragma unroll
                    for (int u = 0; u < 3; ++u) {
                        const int e = u * 64 + lane;                 // 0..191 = tap * 48 + channel
                        const int tp_i = e / 48, c = e - 48 * tp_i;
                        acc3[u] = fmaf(tp[1 + tp_i], DFt[s1 * kDfStride + p * 48 + c], acc3[u]);
                    }
                    ++s1;
                } while (s1 < 16 && __float_as_int(tap[(s1 * 3 + p) * 5]) == idx);
#pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int e = u * 64 + lane;
                    const int tp_i = e / 48, c = e - 48 * tp_i;
                    const size_t off = ((size_t)idx + (tp_i & 1) + (tp_i >> 1) * (size_t)tx.stride) * 48 + c;
                    if (acc3[u] != 0.0f) atomicAdd(T.g_app[p] + off, acc3[u]);
                }
                s0 = s1;
            }
        }
        if (live && q == 0) {
            float *d = T.dt + ((int64_t)i * A.n + r) * 6;
#pragma unroll
            for (int k = 0; k < 6; ++k) d[k] = dt[k];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
}

// ---- 6. weight gradients: out[M][N] += X^T . Y over `rows` samples (sample-major X [rows, ldx], Y [rows, ldy]) ------------------
// A workgroup walks over 32-sample chunks: both operand slabs go through LDS once (every row of X and Y leaves HBM exactly once;
// a tile-per-wave version re-read X and Y once per output tile, 0.94 GB for the 64 x 144 reduction), its four waves own the
// MT x NT output tiles in registers for the whole walk and add them to `out` (the reference-layout gradient tensor
// [Mvalid][ldo]) at the end.
template <int MT, int NT>
__global__ void __launch_bounds__(256) xty_block_kernel(const float *__restrict__ X, int ldx, const float *__restrict__ Y, int ldy, int rows,
                                                        int Mvalid, int Nvalid, float *out, int ldo, const int32_t *rows_dev)
{
    if (rows_dev) rows = min(rows, *rows_dev);              // the row count lives on the device (no host sync on the active count)
    constexpr int CH = 32;                                              // samples per chunk = 8 MFMA k-steps
    constexpr int LX = 16 * MT + ((MT & 1) ? 0 : 16), LY = 16 * NT + ((NT & 1) ? 0 : 16);   // row strides == 16 (mod 32): 2 lanes per bank
    constexpr int TILES = MT * NT, PER_WAVE = (TILES + 3) / 4;
    __shared__ __attribute__((aligned(16))) float sx[CH * LX], sy[CH * LY];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 15, kq = lane >> 4;
    f32x4 acc[PER_WAVE];
#pragma unroll
    for (int t = 0; t < PER_WAVE; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    const int nchunks = (rows + CH - 1) / CH;
    for (int c = blockIdx.x; c < nchunks; c += gridDim.x) {
        const int r0 = c * CH;
        // cooperative, coalesced slab loads (float4 per thread), zero rows past the end
        for (int e = threadIdx.x; e < CH * (4 * MT); e += 256) {
            const int r = e / (4 * MT), q = e - r * (4 * MT);
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (r0 + r < rows) v = *reinterpret_cast<const f32x4 *>(X + (size_t)(r0 + r) * ldx + 4 * q);
            *reinterpret_cast<f32x4 *>(sx + r * LX + 4 * q) = v;
        }
        for (int e = threadIdx.x; e < CH * (4 * NT); e += 256) {
            const int r = e / (4 * NT), q = e - r * (4 * NT);
            f32x4 v = {0.0f, 0.0f, 0.0f, 0.0f};
            if (r0 + r < rows) v = *reinterpret_cast<const f32x4 *>(Y + (size_t)(r0 + r) * ldy + 4 * q);
            *reinterpret_cast<f32x4 *>(sy + r * LY + 4 * q) = v;
        }
        __syncthreads();
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

// bias gradients: column sums of a row-major [rows, ld] matrix (cols <= 64); blocks walk 256-row slabs
__global__ void __launch_bounds__(256) colsum_kernel(const float *__restrict__ X, int ld, int rows, int cols, float *out, const int32_t *rows_dev)
{
    if (rows_dev) rows = min(rows, *rows_dev);
    __shared__ float sh[4][64];
    const int c = threadIdx.x & 63, part = threadIdx.x >> 6;
    float s = 0.0f;
    if (c < cols)
        for (int r0 = blockIdx.x * 256; r0 < rows; r0 += gridDim.x * 256) {
            const int r1 = min(rows, r0 + 256);
            for (int r = r0 + part; r < r1; r += 4) s += X[(size_t)r * ld + c];
        }
    sh[part][c] = s;
    __syncthreads();
    if (part == 0 && c < cols) atomicAdd(out + c, (sh[0][c] + sh[1][c]) + (sh[2][c] + sh[3][c]));
}

// ---- 7. density / gauge backward for every valid sample -----------------------------------------------------------------------
// density_decoder is LINEAR, so the gradient of the 16 density channels of a texel is rank one:
//     d loss / d plane_p[c][texel] = wd[16p+c] * D_p[texel],   D_p[texel] = sum over samples of (tap weight * dx)
// and d loss / d wd[16p+c] = sum over texels of D_p[texel] * plane_p[c][texel].  The per-sample kernel therefore scatters ONE
// float per tap (12 atomics instead of 192) and takes the spatial derivative from the wd-projected 1-channel planes Q_p;
// train_density_finish_kernel expands D_p afterwards.
__global__ void __launch_bounds__(256) train_project_density_kernel(const float *__restrict__ tex16, const float *__restrict__ wd, int64_t texels,
                                                                    float *__restrict__ q)
{
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

__global__ void __launch_bounds__(256) train_density_bwd_kernel(const TrainArgs T)
{
    const RenderArgs &A = T.R;
    __shared__ float s_bd;
    if (threadIdx.x == 0) s_bd = 0.0f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float bsum = 0.0f;
    const int64_t total = (int64_t)A.S * A.n;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    // wave-uniform trip count: the scatter below exchanges values between lanes
    for (int64_t base = (int64_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63); base < total; base += stride) {
        // ray-major work order like the forward kernel: the scatter of a wave lands on the few cache lines along one ray
        const int64_t wi0 = base + lane;
        const bool inr = wi0 < total;
        const int64_t wi = inr ? wi0 : total - 1;
        const int64_t rr = wi / A.S;
        const int64_t idx = (int64_t)(wi - rr * A.S) * A.n + rr;
        float dx = inr ? T.dx[idx] : 0.0f;
        const bool active = inr && (T.w[idx] > A.thr);
        const bool work = (dx != 0.0f) | active;
        if (!__any(work)) continue;
        const int i = (int)(idx / A.n);
        const int64_t r = idx % A.n;
        float t[6], xn[3];
        list_sample_coords(A, r, i, t, xn);
        float dt[6];
        bsum += dx;
        int didx[3];
        float dw[3][4];
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const Tex &tx = A.dens[p];
            BilG b = bilg_setup(t[2 * p], t[2 * p + 1], tx);
            const float *q = T.q_dens[p] + b.idx;
            const float v00 = q[0], v10 = q[1], v01 = q[tx.stride], v11 = q[tx.stride + 1];
            didx[p] = b.idx;
            dw[p][0] = b.w00 * dx; dw[p][1] = b.w10 * dx; dw[p][2] = b.w01 * dx; dw[p][3] = b.w11 * dx;
            dt[2 * p] = dx * (b.wy0 * (v10 - v00) + b.wy1 * (v11 - v01)) * b.sx;
            dt[2 * p + 1] = dx * (b.wx0 * (v01 - v00) + b.wx1 * (v11 - v10)) * b.sy;
        }
        // consecutive lanes are consecutive steps of one ray, half a texel apart: merge the contributions of lanes that hit the
        // SAME cell inside aligned pairs, then quads, before anything goes to L2 (a merged lane's values become 0 and are skipped)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int d = 1; d <= 2; d <<= 1) {
                const int oi = __shfl_xor(didx[p], d);
                const bool take = ((lane & d) == 0) & (oi == didx[p]) & ((lane & (d - 1)) == 0);      // receiver: lower lane of the pair / quad
                const bool give = ((lane & d) != 0) & (oi == didx[p]) & ((lane & (d - 1)) == 0);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float ov = __shfl_xor(dw[p][k], d);
                    dw[p][k] = take ? dw[p][k] + ov : (give ? 0.0f : dw[p][k]);
                }
            }
        }
        // density-gradient images: the two taps of a row are consecutive floats -> lane pairs write them together
        // (32 samples x 2 floats per instruction: half as many cache lines per atomic instruction)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const int strd = A.dens[p].stride;
#pragma unroll
            for (int row = 0; row < 2; ++row) {
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    const int src = half * 32 + (lane >> 1), e = lane & 1;
                    const int ii = __shfl(didx[p], src);
                    const float va = __shfl(dw[p][2 * row], src), vb = __shfl(dw[p][2 * row + 1], src);
                    const float val = e ? vb : va;
                    if (val != 0.0f) atomicAdd(T.d_dens[p] + (size_t)ii + (size_t)row * strd + e, val);
                }
            }
        }
        if (A.mode) {
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
                const Tex &tx = A.gau[p];
                Bil b = bil_setup(u[p], v[p], tx);
                // the 8 contributions of this sample's cell: gv[tap][channel] = tap weight * gradient; same-cell lanes merged as above
                float gv[4][2] = {{b.w00 * dg[p][0], b.w00 * dg[p][1]}, {b.w10 * dg[p][0], b.w10 * dg[p][1]},
                                  {b.w01 * dg[p][0], b.w01 * dg[p][1]}, {b.w11 * dg[p][0], b.w11 * dg[p][1]}};
#pragma unroll
                for (int d = 1; d <= 2; d <<= 1) {
                    const int oi = __shfl_xor(b.idx, d);
                    const bool take = ((lane & d) == 0) & (oi == b.idx) & ((lane & (d - 1)) == 0);
                    const bool give = ((lane & d) != 0) & (oi == b.idx) & ((lane & (d - 1)) == 0);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
                            const float ov = __shfl_xor(gv[k][c], d);
                            gv[k][c] = take ? gv[k][c] + ov : (give ? 0.0f : gv[k][c]);
                        }
                }
                // a row of the cell is 4 consecutive floats (2 texels x 2 channels): 16 samples x 4 floats per instruction
#pragma unroll
                for (int row = 0; row < 2; ++row) {
#pragma unroll
                    for (int sq = 0; sq < 4; ++sq) {
                        const int src = sq * 16 + (lane >> 2), e = lane & 3;
                        const int ii = __shfl(b.idx, src);
                        const float v0 = __shfl(gv[2 * row][0], src), v1 = __shfl(gv[2 * row][1], src);
                        const float v2 = __shfl(gv[2 * row + 1][0], src), v3 = __shfl(gv[2 * row + 1][1], src);
                        const float val = e == 0 ? v0 : (e == 1 ? v1 : (e == 2 ? v2 : v3));
                        if (val != 0.0f) atomicAdd(T.g_gau[p] + ((size_t)ii + (size_t)row * tx.stride) * 2 + e, val);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) bsum += __shfl_xor(bsum, s);
    if (lane == 0) atomicAdd(&s_bd, bsum);
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(T.g_bd, s_bd);
}

// expand the scalar gradient image of plane p: g_dens[texel][c] = wd[c] * D[texel]; g_wd[c] += sum_texels D[texel] * tex16[texel][c]
__global__ void __launch_bounds__(256) train_density_finish_kernel(const float *__restrict__ D, const float *__restrict__ tex16,
                                                                   const float *__restrict__ wd, int64_t texels, float *__restrict__ g_dens, float *g_wd)
{
    __shared__ float sh[16];
    if (threadIdx.x < 16) sh[threadIdx.x] = 0.0f;
    __syncthreads();
    float acc[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc[c] = 0.0f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < texels; i += stride) {
        const float d = D[i];
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

__global__ void __launch_bounds__(256) adam_dense_kernel(float *p, const float *g, float *m, float *v, int64_t n, const AdamArgs a)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float mi = m[i], vi = v[i];
        p[i] = adam_one(p[i], g[i], mi, vi, a);
        m[i] = mi; v[i] = vi;
    }
}

// NCHW parameter [C,H,W]; gradient in the packed layouts: channels [0,cs) in ga (cs per texel), [cs,C) in gb (C-cs per texel)
__global__ void __launch_bounds__(256) adam_plane_kernel(float *p, float *m, float *v, int C, int H, int W, const float *ga, int cs, const float *gb,
                                                         const AdamArgs a)
{
    const int64_t total = (int64_t)C * H * W;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int x = (int)(i % W), y = (int)((i / W) % H), c = (int)(i / ((int64_t)W * H));
        const size_t texel = (size_t)(y + 1) * (W + 2) + (x + 1);
        float g = c < cs ? ga[texel * cs + c] : gb[texel * (C - cs) + (c - cs)];
        const float pv = p[i];
        g += a.l1 * (pv > 0.0f ? 1.0f : (pv < 0.0f ? -1.0f : 0.0f));
        float mi = m[i], vi = v[i];
        p[i] = adam_one(pv, g, mi, vi, a);
        m[i] = mi; v[i] = vi;
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
