// ngf_eval.hip -- C ABI (include/ngf.h), eval output stage + small tensor utilities (SURVEY 8 N4 / N2 / N3 helpers).
#include "ngf_host.hpp"
#include "ngf_eval.hpp"

using namespace ngf;

// ================================ eval output stage ========================================================
static int eval_grid(int64_t n)
{
    int64_t g = (n + kEvalThreads - 1) / kEvalThreads;
    if (g > kEvalMaxBlocks) g = kEvalMaxBlocks;
    return g < 1 ? 1 : (int)g;
}

extern "C" int ngf_eval_frame_u8(const float *rgb, int64_t n_values, uint8_t *out, void *hip_stream)
{
    if (n_values < 0 || (n_values > 0 && (!rgb || !out))) return fail(NGF_E_ARG, "ngf_eval_frame_u8: bad argument");
    if (n_values == 0) return NGF_OK;
    hipLaunchKernelGGL(frame_u8_kernel, dim3(eval_grid(n_values)), dim3(kEvalThreads), 0, (hipStream_t)hip_stream, rgb, n_values, out);
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

extern "C" int64_t ngf_eval_workspace_bytes(int32_t H, int32_t W, int32_t filter_size)
{
    // reductions: 2 floats or 1 double per block; SSIM: five float64 moment images after the vertical blur
    int64_t b = (int64_t)kEvalMaxBlocks * 2 * sizeof(double);
    if (H > 0 && W > 0 && filter_size > 0 && filter_size <= H) b += 5 * (int64_t)(H - filter_size + 1) * W * 3 * (int64_t)sizeof(double);
    return b;
}

extern "C" int ngf_eval_depth_range(const float *depth, int64_t n, float *range, void *workspace, void *hip_stream)
{
    if (!depth || !range || !workspace || n <= 0) return fail(NGF_E_ARG, "ngf_eval_depth_range: bad argument");
    const int g = eval_grid(n);
    hipStream_t st = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(depth_range_partial_kernel, dim3(g), dim3(kEvalThreads), 0, st, depth, n, (float *)workspace);
    hipLaunchKernelGGL(depth_range_final_kernel, dim3(1), dim3(kEvalThreads), 0, st, (const float *)workspace, g, range);
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

extern "C" int ngf_eval_depth_colormap(const float *depth, int64_t n, const float *range, const uint8_t *lut, uint8_t *out, void *hip_stream)
{
    if (n < 0 || (n > 0 && (!depth || !range || !lut || !out))) return fail(NGF_E_ARG, "ngf_eval_depth_colormap: bad argument");
    if (n == 0) return NGF_OK;
    hipLaunchKernelGGL(depth_colormap_kernel, dim3(eval_grid(n)), dim3(kEvalThreads), 0, (hipStream_t)hip_stream, depth, n, range, lut, out);
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

extern "C" int ngf_eval_mse(const float *a, const float *b, int64_t n, double *out, void *workspace, void *hip_stream)
{
    if (!a || !b || !out || !workspace || n <= 0) return fail(NGF_E_ARG, "ngf_eval_mse: bad argument");
    const int g = eval_grid(n);
    hipStream_t st = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(mse_partial_kernel, dim3(g), dim3(kEvalThreads), 0, st, a, b, n, (double *)workspace);
    hipLaunchKernelGGL(mean_final_kernel, dim3(1), dim3(kEvalThreads), 0, st, (const double *)workspace, g, (double)n, out);
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

extern "C" int ngf_eval_ssim(const float *img0, const float *img1, int32_t H, int32_t W, double max_val, int32_t filter_size,
                             double filter_sigma, double k1, double k2, double *mean_out, double *map_out, void *workspace, void *hip_stream)
{
    if (!img0 || !img1 || !mean_out || !workspace) return fail(NGF_E_ARG, "ngf_eval_ssim: null argument");
    if (filter_size < 1 || filter_size > kSsimMaxTaps) return fail(NGF_E_UNSUPPORTED, "ngf_eval_ssim: filter_size must be in 1..%d", kSsimMaxTaps);
    if (H < filter_size || W < filter_size) return fail(NGF_E_ARG, "ngf_eval_ssim: image %dx%d smaller than the %d-tap filter", H, W, filter_size);
    SsimArgs a;
    // the reference's 1-D Gaussian (utils.py:121-125), float64
    const int hw = filter_size / 2;
    const double shift = (2 * hw - filter_size + 1) / 2.0;
    double sum = 0.0;
    for (int i = 0; i < filter_size; ++i) {
        const double t = ((double)(i - hw) + shift) / filter_sigma;
        a.filt[i] = exp(-0.5 * (t * t));
        sum += a.filt[i];
    }
    for (int i = 0; i < filter_size; ++i) a.filt[i] /= sum;
    a.taps = filter_size; a.H = H; a.W = W; a.Ho = H - filter_size + 1; a.Wo = W - filter_size + 1;
    a.c1 = (k1 * max_val) * (k1 * max_val);
    a.c2 = (k2 * max_val) * (k2 * max_val);
    hipStream_t st = (hipStream_t)hip_stream;
    double *partial = (double *)workspace;
    double *tmp = partial + 2 * kEvalMaxBlocks;
    const int64_t n_v = (int64_t)a.Ho * W * 3, n_o = (int64_t)a.Ho * a.Wo * 3;
    const int g = eval_grid(n_o);
    hipLaunchKernelGGL(ssim_vertical_kernel, dim3(eval_grid(n_v)), dim3(kEvalThreads), 0, st, a, img0, img1, tmp);
    hipLaunchKernelGGL(ssim_horizontal_kernel, dim3(g), dim3(kEvalThreads), 0, st, a, (const double *)tmp, map_out, partial);
    hipLaunchKernelGGL(mean_final_kernel, dim3(1), dim3(kEvalThreads), 0, st, (const double *)partial, g, (double)n_o, mean_out);
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

// ABI 5: TriPlane.density_L1 (Field.py:149-152) and its gradient, three planes per launch.  planes[k] / n[k]: the k-th plane's values (any layout:
// the mean runs over all of them) -- contiguous float32, 16-byte aligned; out: one float (device); workspace: 3 * 256 doubles (device).
extern "C" int ngf_planes_l1(const float *const *planes, const int64_t *n, float *out, void *workspace, void *hip_stream)
{
    if (!planes || !n || !out || !workspace) return fail(NGF_E_ARG, "ngf_planes_l1: null argument");
    PlanesL1 a;
    for (int k = 0; k < 3; ++k) {
        if (!planes[k] || n[k] <= 0 || ((uintptr_t)planes[k] & 15)) return fail(NGF_E_ARG, "ngf_planes_l1: plane %d missing, empty or not 16-byte aligned", k);
        a.p[k] = planes[k]; a.n[k] = n[k]; a.g[k] = nullptr;
    }
    hipStream_t st = (hipStream_t)hip_stream;
    hipLaunchKernelGGL(planes_l1_partial_kernel, dim3(kL1Blocks, 3), dim3(kEvalThreads), 0, st, a, (double *)workspace);
    hipLaunchKernelGGL(planes_l1_final_kernel, dim3(1), dim3(kEvalThreads), 0, st, (const double *)workspace, a, out);
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}
// grads[k] (NULL = not wanted) <- sign(planes[k]) * upstream[0] / n[k]; upstream: one float on the device (d loss / d density_L1)
extern "C" int ngf_planes_l1_backward(const float *const *planes, const int64_t *n, const float *upstream, float *const *grads, void *hip_stream)
{
    if (!planes || !n || !upstream || !grads) return fail(NGF_E_ARG, "ngf_planes_l1_backward: null argument");
    PlanesL1 a;
    bool any = false;
    for (int k = 0; k < 3; ++k) {
        if (!planes[k] || n[k] <= 0 || ((uintptr_t)planes[k] & 15) || ((uintptr_t)grads[k] & 15)) return fail(NGF_E_ARG, "ngf_planes_l1_backward: plane %d missing, empty or not 16-byte aligned", k);
        a.p[k] = planes[k]; a.n[k] = n[k]; a.g[k] = grads[k];
        any |= grads[k] != nullptr;
    }
    if (!any) return NGF_OK;
    hipLaunchKernelGGL(planes_l1_backward_kernel, dim3(1024, 3), dim3(kEvalThreads), 0, (hipStream_t)hip_stream, a, upstream);
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

extern "C" int ngf_pack_mask_bits(const float *volume, int64_t n, uint8_t *bits, void *hip_stream)
{
    if (!volume || !bits || n <= 0) return fail(NGF_E_ARG, "ngf_pack_mask_bits: bad argument");
    hipLaunchKernelGGL(pack_mask_bits_kernel, dim3(eval_grid((n + 7) / 8)), dim3(kEvalThreads), 0, (hipStream_t)hip_stream, volume, n, bits);
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

extern "C" int ngf_resize_bilinear(const float *src, int32_t C, int32_t Hi, int32_t Wi, float *dst, int32_t Ho, int32_t Wo, void *hip_stream)
{
    if (!src || !dst || C <= 0 || Hi <= 0 || Wi <= 0 || Ho <= 0 || Wo <= 0) return fail(NGF_E_ARG, "ngf_resize_bilinear: bad argument");
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3(eval_grid((int64_t)C * Ho * Wo)), dim3(kEvalThreads), 0, (hipStream_t)hip_stream, src, C, Hi, Wi, dst,
                       Ho, Wo);
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

