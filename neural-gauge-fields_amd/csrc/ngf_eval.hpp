// ngf_eval.hpp -- the eval output stage on the device: what the reference's `evaluation`
// (TriPlane/main.py:73-138) does on the host with numpy / scipy / cv2 after every rendered frame.  Keeping it on the
// GPU means a frame leaves HBM as 8-bit images (1.9 MB + 1.9 MB) and three scalars instead of 10 MB of float32.
//
//   frame_u8_kernel        rgb_map.clamp(0,1) (main.py:98) ; (rgb_map.numpy()*255).astype('uint8') (main.py:117)
//   depth_range_kernels    mi = min(x[x>0]), ma = max(x)                       (utils.py:38-40)
//   depth_colormap_kernel  nan_to_num ; (x-mi)/(ma-mi+1e-8) ; (255*x).astype(uint8) ; cv2.applyColorMap  (utils.py:37-46)
//   mse_kernels            torch.mean((rgb_map - gt_rgb)**2)                   (main.py:105)
//   ssim_*_kernel          rgb_ssim (utils.py:109-155): separable 'valid' Gaussian blur in float64, SSIM map, mean
//
// Everything is HBM-streaming byte/float work: one coalesced pass per stage, reductions in two deterministic steps
// (per-block partials in a fixed order, then one block) -- no float atomics, results do not depend on scheduling.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ngf {

typedef float f32x4e __attribute__((ext_vector_type(4)));
constexpr int kEvalThreads = 256;
constexpr int kEvalMaxBlocks = 1024;
constexpr int kSsimMaxTaps = 33;

struct SsimArgs {
    double filt[kSsimMaxTaps];
    int taps, H, W, Ho, Wo;
    double c1, c2;
};

__global__ void __launch_bounds__(kEvalThreads) frame_u8_kernel(const float *rgb, int64_t n, uint8_t *out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float v = rgb[i];
        v = fminf(fmaxf(v, 0.0f), 1.0f);           // clamp propagates NaN in torch; fminf/fmaxf would drop it
        if (rgb[i] != rgb[i]) v = 0.0f;            // numpy casts NaN to 0 on x86 (cvttss2si -> INT_MIN -> low byte 0)
        out[i] = (uint8_t)(int)(v * 255.0f);       // astype('uint8') truncates
    }
}

// block-level reduction helper: every thread contributes v; thread 0 gets op over the block in a FIXED order
template <typename T, typename Op>
__device__ __forceinline__ T block_reduce(T v, Op op, T *sh)
{
    sh[threadIdx.x] = v;
    __syncthreads();
    for (int s = kEvalThreads / 2; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] = op(sh[threadIdx.x], sh[threadIdx.x + s]);
        __syncthreads();
    }
    return sh[0];
}

// pass 1: per-block (min positive, max) of nan_to_num(depth); pass 2 (one block) folds the partials -> range[2]
__global__ void __launch_bounds__(kEvalThreads) depth_range_partial_kernel(const float *depth, int64_t n, float *partial)
{
    __shared__ float sh[kEvalThreads];
    float mn = INFINITY, mx = -INFINITY;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float x = depth[i];
        if (x != x) x = 0.0f;
        else if (x == INFINITY) x = 3.4028234663852886e38f;      // np.nan_to_num: +-inf -> +-FLT_MAX
        else if (x == -INFINITY) x = -3.4028234663852886e38f;
        if (x > 0.0f) mn = fminf(mn, x);
        mx = fmaxf(mx, x);
    }
    mn = block_reduce(mn, [](float a, float b) { return fminf(a, b); }, sh);
    __syncthreads();
    mx = block_reduce(mx, [](float a, float b) { return fmaxf(a, b); }, sh);
    if (threadIdx.x == 0) { partial[2 * blockIdx.x] = mn; partial[2 * blockIdx.x + 1] = mx; }
}

__global__ void __launch_bounds__(kEvalThreads) depth_range_final_kernel(const float *partial, int blocks, float *range)
{
    __shared__ float sh[kEvalThreads];
    float mn = INFINITY, mx = -INFINITY;
    for (int i = threadIdx.x; i < blocks; i += blockDim.x) { mn = fminf(mn, partial[2 * i]); mx = fmaxf(mx, partial[2 * i + 1]); }
    mn = block_reduce(mn, [](float a, float b) { return fminf(a, b); }, sh);
    __syncthreads();
    mx = block_reduce(mx, [](float a, float b) { return fmaxf(a, b); }, sh);
    if (threadIdx.x == 0) { range[0] = mn; range[1] = mx; }
}

// x = (x - mi) / (ma - mi + 1e-8) in float32 (numpy keeps float32 for float32-array (op) python-float), u8 = (255*x) cast
// the way numpy does on x86-64 (float -> int32 truncation, low byte kept: out-of-range depths wrap, they do occur because
// depth_map carries the reference's (1-acc)*rays[...,-1] term), then the 256-entry colour table (BGR, like cv2).
__global__ void __launch_bounds__(kEvalThreads) depth_colormap_kernel(const float *depth, int64_t n, const float *range, const uint8_t *lut,
                                                                      uint8_t *out)
{
    const float mi = range[0], ma = range[1];
    const float den = (ma - mi) + 1e-8f;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float x = depth[i];
        if (x != x) x = 0.0f;
        else if (x == INFINITY) x = 3.4028234663852886e38f;
        else if (x == -INFINITY) x = -3.4028234663852886e38f;
        x = (x - mi) / den;
        const float y = 255.0f * x;
        int q;
        if (!(y > -2147483648.0f && y < 2147483648.0f)) q = (int)0x80000000;     // cvttss2si "integer indefinite"
        else q = (int)y;
        const int idx = q & 255;
        out[3 * i] = lut[3 * idx];
        out[3 * i + 1] = lut[3 * idx + 1];
        out[3 * i + 2] = lut[3 * idx + 2];
    }
}

__global__ void __launch_bounds__(kEvalThreads) mse_partial_kernel(const float *a, const float *b, int64_t n, double *partial)
{
    __shared__ double sh[kEvalThreads];
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float d = a[i] - b[i];
        s += (double)(d * d);                      // the square is float32 in the reference, the mean's accumulation is wider here
    }
    s = block_reduce(s, [](double x, double y) { return x + y; }, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ void __launch_bounds__(kEvalThreads) mean_final_kernel(const double *partial, int blocks, double count, double *out)
{
    __shared__ double sh[kEvalThreads];
    double s = 0.0;
    for (int i = threadIdx.x; i < blocks; i += blockDim.x) s += partial[i];
    s = block_reduce(s, [](double x, double y) { return x + y; }, sh);
    if (threadIdx.x == 0) out[0] = s / count;
}

// ---- density_L1 (Field.py:149-152): sum over the three planes of mean(|plane|), and its gradient ---------------------------------------------
// The reference's loop adds L1_reg_weight * field.density_L1() to its loss (main.py:279-281); as torch ops that is abs + mean per plane and, in the
// backward, sign x scale per plane with a 16.7 MB |plane| intermediate each -- ~14 launches over 150 MB.  Here: a partial-sum launch over the three
// planes (blockIdx.y = plane; float32 |x| summed in double per thread, fixed order: deterministic), a one-block finish, and ONE backward launch.
struct PlanesL1 { const float *p[3]; int64_t n[3]; float *g[3]; };
constexpr int kL1Blocks = 256;
__global__ void __launch_bounds__(kEvalThreads) planes_l1_partial_kernel(const PlanesL1 a, double *partial)
{
    __shared__ double sh[kEvalThreads];
    const int pl = blockIdx.y;
    const float *__restrict__ x = a.p[pl];
    const int64_t n = a.n[pl], n4 = n >> 2;
    double s = 0.0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4e v = reinterpret_cast<const f32x4e *>(x)[i];
        s += (double)((fabsf(v[0]) + fabsf(v[1])) + (fabsf(v[2]) + fabsf(v[3])));
    }
    if (blockIdx.x == 0)
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) s += (double)fabsf(x[i]);
    s = block_reduce(s, [](double u, double w) { return u + w; }, sh);
    if (threadIdx.x == 0) partial[pl * kL1Blocks + blockIdx.x] = s;
}
__global__ void __launch_bounds__(kEvalThreads) planes_l1_final_kernel(const double *partial, const PlanesL1 a, float *out)
{
    __shared__ double sh[kEvalThreads];
    double total = 0.0;
    for (int pl = 0; pl < 3; ++pl) {
        double s = threadIdx.x < kL1Blocks ? partial[pl * kL1Blocks + threadIdx.x] : 0.0;
        s = block_reduce(s, [](double u, double w) { return u + w; }, sh);
        __syncthreads();
        total += (double)(float)(s / (double)a.n[pl]);          // each plane's mean is a float32 in the reference; their sum is a float32 addition chain
        if (pl == 1) total = (double)(float)total;
    }
    if (threadIdx.x == 0) out[0] = (float)total;
}
// d/dp [mean(|p|)] * upstream = sign(p) * (upstream / n): torch's abs backward (grad * sgn(p), sgn(0) = 0) after mean's (grad / n)
__global__ void __launch_bounds__(kEvalThreads) planes_l1_backward_kernel(const PlanesL1 a, const float *upstream)
{
    const int pl = blockIdx.y;
    float *__restrict__ g = a.g[pl];
    if (!g) return;
    const float *__restrict__ x = a.p[pl];
    const int64_t n = a.n[pl], n4 = n >> 2;
    const float sc = upstream[0] * (1.0f / (float)n);          // torch's mean backward divides by a host scalar as a * (1 / b) in float32 (its div kernel's scalar path): the same bits
    auto sg = [sc](float v) { return v > 0.0f ? sc : (v < 0.0f ? -sc : (v == 0.0f ? 0.0f * sc : v)); };          // NaN stays NaN like torch.sgn
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4e v = reinterpret_cast<const f32x4e *>(x)[i];
        reinterpret_cast<f32x4e *>(g)[i] = f32x4e{sg(v[0]), sg(v[1]), sg(v[2]), sg(v[3])};
    }
    if (blockIdx.x == 0)
        for (int64_t i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) g[i] = sg(x[i]);
}

// vertical blur of the five moment images (img0, img1, img0^2, img1^2, img0*img1; the products are float32 as in the
// reference, which squares torch float32 tensors before scipy widens them): tmp[q][yo][x][c], float64
__global__ void __launch_bounds__(kEvalThreads) ssim_vertical_kernel(const SsimArgs a, const float *img0, const float *img1, double *tmp)
{
    const int64_t plane = (int64_t)a.Ho * a.W * 3;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < plane; i += stride) {
        const int64_t yo = i / ((int64_t)a.W * 3), rest = i % ((int64_t)a.W * 3);
        double s0 = 0, s1 = 0, s00 = 0, s11 = 0, s01 = 0;
        for (int k = 0; k < a.taps; ++k) {
            const int64_t j = (yo + k) * (int64_t)a.W * 3 + rest;
            const float u = img0[j], v = img1[j];
            const double f = a.filt[a.taps - 1 - k];          // convolution: the filter runs backwards (it is symmetric)
            s0 += f * (double)u;
            s1 += f * (double)v;
            s00 += f * (double)(u * u);
            s11 += f * (double)(v * v);
            s01 += f * (double)(u * v);
        }
        tmp[i] = s0; tmp[plane + i] = s1; tmp[2 * plane + i] = s00; tmp[3 * plane + i] = s11; tmp[4 * plane + i] = s01;
    }
}

// horizontal blur + the SSIM formula (utils.py:137-153) + per-block partial sums of the map
__global__ void __launch_bounds__(kEvalThreads) ssim_horizontal_kernel(const SsimArgs a, const double *tmp, double *map, double *partial)
{
    __shared__ double sh[kEvalThreads];
    const int64_t plane = (int64_t)a.Ho * a.W * 3;
    const int64_t n_out = (int64_t)a.Ho * a.Wo * 3;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    double acc = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_out; i += stride) {
        const int c = (int)(i % 3);
        const int64_t xo = (i / 3) % a.Wo, yo = i / (3 * (int64_t)a.Wo);
        double m[5] = {0, 0, 0, 0, 0};
        for (int k = 0; k < a.taps; ++k) {
            const int64_t j = (yo * a.W + xo + k) * 3 + c;
            const double f = a.filt[a.taps - 1 - k];
#pragma unroll
            for (int q = 0; q < 5; ++q) m[q] += f * tmp[q * plane + j];
        }
        const double mu00 = m[0] * m[0], mu11 = m[1] * m[1], mu01 = m[0] * m[1];
        double s00 = m[2] - mu00, s11 = m[3] - mu11, s01 = m[4] - mu01;
        s00 = fmax(0.0, s00);
        s11 = fmax(0.0, s11);
        const double lim = fmin(sqrt(s00 * s11), fabs(s01));
        s01 = (s01 > 0.0 ? 1.0 : (s01 < 0.0 ? -1.0 : 0.0)) * lim;
        const double numer = (2.0 * mu01 + a.c1) * (2.0 * s01 + a.c2);
        const double denom = (mu00 + mu11 + a.c1) * (s00 + s11 + a.c2);
        const double v = numer / denom;
        if (map) map[i] = v;
        acc += v;
    }
    acc = block_reduce(acc, [](double x, double y) { return x + y; }, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

// np.packbits(volume.bool().reshape(-1)) (the checkpoint's alpha-mask image, FieldBase.py:104-108): MSB first, zero tail
__global__ void __launch_bounds__(kEvalThreads) pack_mask_bits_kernel(const float *__restrict__ vol, int64_t n, uint8_t *__restrict__ bits)
{
    const int64_t nbytes = (n + 7) / 8;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < nbytes; b += stride) {
        unsigned v = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t i = b * 8 + k;
            v |= (unsigned)(i < n && vol[i] != 0.0f) << (7 - k);
        }
        bits[b] = (uint8_t)v;
    }
}

// F.interpolate(mode='bilinear', align_corners=True) of an NCHW tensor (TriPlane.up_sampling, Field.py:108-114):
// ATen upsample_bilinear2d: src = dst * (in-1)/(out-1); the +1 neighbour is dropped on the last row/column.
__global__ void __launch_bounds__(kEvalThreads) resize_bilinear_kernel(const float *__restrict__ src, int C, int Hi, int Wi, float *__restrict__ dst,
                                                                       int Ho, int Wo)
{
    const float sh = Ho > 1 ? (float)(Hi - 1) / (float)(Ho - 1) : 0.0f;
    const float sw = Wo > 1 ? (float)(Wi - 1) / (float)(Wo - 1) : 0.0f;
    const int64_t total = (int64_t)C * Ho * Wo;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho), c = (int)(i / ((int64_t)Wo * Ho));
        const float fy = sh * (float)y, fx = sw * (float)x;
        const int y0 = (int)fy, x0 = (int)fx;
        const int yp = y0 < Hi - 1 ? 1 : 0, xp = x0 < Wi - 1 ? 1 : 0;
        const float ly1 = fy - (float)y0, ly0 = 1.0f - ly1, lx1 = fx - (float)x0, lx0 = 1.0f - lx1;
        const float *p = src + ((int64_t)c * Hi + y0) * Wi + x0;
        dst[i] = ly0 * (lx0 * p[0] + lx1 * p[xp]) + ly1 * (lx0 * p[(int64_t)yp * Wi] + lx1 * p[(int64_t)yp * Wi + xp]);
    }
}

}  // namespace ngf
