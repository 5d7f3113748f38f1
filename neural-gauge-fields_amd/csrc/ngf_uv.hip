// ngf_uv.hip -- C ABI (include/ngf.h), UV-Mapping (NeuTex) part: weight packing, render, texture editing.
#include "ngf_host.hpp"
#include "ngf_uv.hpp"

using namespace ngf;

// ================================ UV-Mapping (NeuTex) ===================================================================
struct ngf_uv {
    float *w = nullptr;
    float *tex = nullptr;          // owned copy of the edit texture (ngf_uv_set_texture)
    unsigned int *counters = nullptr;
    mutable std::atomic<unsigned> next_counter{0};
    UvArgs proto;
    int num_cus = 256;
    int dev = 0;                   // the device the handle's buffers live on (the caller's current device at create)
};

extern "C" int ngf_uv_destroy(ngf_uv *m)
{
    if (!m) return NGF_OK;
    DeviceScope ds(m->dev);        // hipFree waits for the device's work: on the handle's device, whatever is current in the calling thread
    if (m->w) (void)hipFree(m->w);
    if (m->tex) (void)hipFree(m->tex);
    if (m->counters) (void)hipFree(m->counters);
    delete m;
    return NGF_OK;
}

namespace {
struct UvPacker {
    std::vector<float> buf;
    static int hidden(int t, int kq) { return (t >> 2) * 16 + 4 * kq + (t & 3); }
    int align() { while (buf.size() & 3) buf.push_back(0.0f); return (int)buf.size(); }
    // imap(t, kq) -> input index (or -1); KT k-steps; NT unit tiles (multiple of 4)
    template <typename F>
    int dense(const std::vector<float> &W, int out_f, int in_f, int KT, int NT, F imap)
    {
        const int off = align();
        buf.resize(off + (size_t)KT * NT * 64, 0.0f);
        for (int t = 0; t < KT; ++t)
            for (int g = 0; g < NT / 4; ++g)
                for (int l = 0; l < 64; ++l)
                    for (int e = 0; e < 4; ++e) {
                        const int o = (4 * g + e) * 16 + (l & 15), i = imap(t, l >> 4);
                        buf[off + (((size_t)t * (NT / 4) + g) * 64 + l) * 4 + e] = (o < out_f && i >= 0 && i < in_f) ? W[(size_t)o * in_f + i] : 0.0f;
                    }
        return off;
    }
    // output layer with <= 3 units: [KT / 4][64 lanes][4] -- a lane's A operands of four k-steps in one 16-byte load (dense_out)
    template <typename F>
    int out_layer(const std::vector<float> &W, int out_f, int in_f, int KT, F imap)
    {
        const int off = align();
        buf.resize(off + (size_t)KT * 64, 0.0f);
        for (int t = 0; t < KT; ++t)
            for (int l = 0; l < 64; ++l) {
                const int o = l & 15, i = imap(t, l >> 4);
                buf[off + ((size_t)(t >> 2) * 64 + l) * 4 + (t & 3)] = (o < out_f && i >= 0 && i < in_f) ? W[(size_t)o * in_f + i] : 0.0f;
            }
        return off;
    }
    // split-bf16 image of a 256-unit layer: [KB][4 groups][3 parts][4 tiles][64 lanes][8 bf16]; imap(j, kq) = input index of the lane
    // quarter's j-th input (j = 8 kb + e), as in dense()
    static uint16_t f2bf(float x)
    {
        uint32_t u;
        memcpy(&u, &x, 4);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);
        return (uint16_t)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
    }
    static float bf2f(uint16_t h)
    {
        const uint32_t u = (uint32_t)h << 16;
        float f;
        memcpy(&f, &u, 4);
        return f;
    }
    template <typename F>
    int dense_bf16(const std::vector<float> &W, int out_f, int in_f, int KB, F imap)
    {
        const int off = align();
        buf.resize(off + (size_t)KB * 4 * 12 * 64 * 4, 0.0f);
        uint16_t *h16 = reinterpret_cast<uint16_t *>(buf.data() + off);
        for (int kb = 0; kb < KB; ++kb)
            for (int g = 0; g < 4; ++g)
                for (int e = 0; e < 4; ++e)
                    for (int l = 0; l < 64; ++l)
                        for (int ee = 0; ee < 8; ++ee) {
                            const int o = (4 * g + e) * 16 + (l & 15), i = imap(kb * 8 + ee, l >> 4);
                            const float wv = (o < out_f && i >= 0 && i < in_f) ? W[(size_t)o * in_f + i] : 0.0f;
                            uint16_t p3[3];
                            p3[0] = f2bf(wv);
                            const float r1 = wv - bf2f(p3[0]);
                            p3[1] = f2bf(r1);
                            p3[2] = f2bf(r1 - bf2f(p3[1]));
                            for (int part = 0; part < 3; ++part)
                                h16[(((((size_t)kb * 4 + g) * 3 + part) * 4 + e) * 64 + l) * 8 + ee] = p3[part];
                        }
        return off;
    }
    int bias(const std::vector<float> &b, int out_f, int NT)
    {
        const int off = align();
        buf.resize(off + (size_t)NT * 16, 0.0f);
        for (int kq = 0; kq < 4; ++kq)
            for (int mt = 0; mt < NT; ++mt)
                for (int r = 0; r < 4; ++r) {
                    const int o = mt * 16 + 4 * kq + r;
                    buf[off + kq * (NT * 4) + mt * 4 + r] = o < out_f ? b[o] : 0.0f;
                }
        return off;
    }
    int bias4(const std::vector<float> &b, int out_f)
    {
        const int off = align();
        for (int e = 0; e < 4; ++e) buf.push_back(e < out_f ? b[e] : 0.0f);
        return off;
    }
};
}  // namespace

extern "C" int ngf_uv_create(const ngf_uv_desc *d, ngf_uv **out, void *hip_stream)
{
    if (!d || !out) return fail(NGF_E_ARG, "ngf_uv_create: null argument");
    *out = nullptr;
    hipStream_t st = (hipStream_t)hip_stream;
    const int ud = d->sphere ? 3 : 2;
    const int in_uv = ud + 20 * ud;
    static const int kOut[NGF_UV_LAYERS] = {256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 1, 64, 128, 128, 128, 0,
                                            256, 256, 256, 256, 256, 256, 3, 256, 256, 256, 256, 3};
    static const int kIn[NGF_UV_LAYERS] = {63, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 256, 63, 64, 128, 128, 128,
                                           0, 256, 256, 256, 256, 256, 256, 295, 256, 256, 256, 256};
    std::vector<std::vector<float>> W(NGF_UV_LAYERS), B(NGF_UV_LAYERS);
    for (int l = 0; l < NGF_UV_LAYERS; ++l) {
        const int o = l == 16 ? ud : kOut[l], i = l == 17 ? in_uv : kIn[l];
        if (!d->w[l] || !d->b[l]) return fail(NGF_E_ARG, "ngf_uv_create: layer %d missing", l);
        int rc;
        if ((rc = d2h(W[l], d->w[l], (size_t)o * i, st)) || (rc = d2h(B[l], d->b[l], o, st))) return rc;
    }
    HIP_TRY(hipStreamSynchronize(st));
    ngf_uv *m = new (std::nothrow) ngf_uv();
    if (!m) return fail(NGF_E_HIP, "out of host memory");
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) m->num_cus = prop.multiProcessorCount;
    m->dev = dev;
    UvArgs &A = m->proto;
    memset(&A, 0, sizeof(A));
    A.sphere = d->sphere ? 1 : 0;
    if (d->flags & ~NGF_UV_F_SPLIT_BF16) {
        delete m;
        return fail(NGF_E_ARG, "ngf_uv_create: unknown bits in flags (0x%x) -- a caller built against ABI 1 passes padding here", d->flags);
    }
    A.split_bf16 = (d->flags & NGF_UV_F_SPLIT_BF16) ? 1 : 0;
    UvPacker P;
    auto nat = [](int t, int kq) { return 4 * t + kq; };                      // positional-encoding inputs: natural order
    auto hid = [](int t, int kq) { return UvPacker::hidden(t, kq); };         // previous layer's accumulator order
    // geometry
    A.geo_w0 = P.dense(W[0], 256, 63, 16, 16, nat);  A.geo_b0 = P.bias(B[0], 256, 16);
    for (int l = 0; l < 10; ++l) {
        const int o = P.dense(W[1 + l], 256, 256, 64, 16, hid);
        if (l == 0) A.geo_wh = o;
    }
    for (int l = 0; l < 10; ++l) {
        const int o = P.bias(B[1 + l], 256, 16);
        if (l == 0) A.geo_bh = o;
    }
    A.geo_wo = P.out_layer(W[11], 1, 256, 64, hid);  A.geo_bo = P.bias4(B[11], 1);
    // gauge
    A.ga_w0 = P.dense(W[12], 64, 63, 16, 4, nat);    A.ga_b0 = P.bias(B[12], 64, 4);
    A.ga_w1 = P.dense(W[13], 128, 64, 16, 8, hid);   A.ga_b1 = P.bias(B[13], 128, 8);
    A.ga_w2 = P.dense(W[14], 128, 128, 32, 8, hid);  A.ga_b2 = P.bias(B[14], 128, 8);
    A.ga_w3 = P.dense(W[15], 128, 128, 32, 8, hid);  A.ga_b3 = P.bias(B[15], 128, 8);
    A.ga_wo = P.out_layer(W[16], ud, 128, 32, hid);  A.ga_bo = P.bias4(B[16], ud);
    // texture
    A.t1_w0 = P.dense(W[17], 256, in_uv, d->sphere ? 16 : 12, 16, nat);  A.t1_b0 = P.bias(B[17], 256, 16);
    for (int l = 0; l < 5; ++l) {
        const int o = P.dense(W[18 + l], 256, 256, 64, 16, hid);
        if (l == 0) A.t1_wh = o;
    }
    for (int l = 0; l < 5; ++l) {
        const int o = P.bias(B[18 + l], 256, 16);
        if (l == 0) A.t1_bh = o;
    }
    A.c1_w = P.out_layer(W[23], 3, 256, 64, hid);    A.c1_b = P.bias4(B[23], 3);
    A.t2_w0 = P.dense(W[24], 256, 295, 76, 16, [](int t, int kq) { return t < 64 ? UvPacker::hidden(t, kq) : 256 + 4 * (t - 64) + kq; });
    A.t2_b0 = P.bias(B[24], 256, 16);
    for (int l = 0; l < 3; ++l) {
        const int o = P.dense(W[25 + l], 256, 256, 64, 16, hid);
        if (l == 0) A.t2_wh = o;
    }
    for (int l = 0; l < 3; ++l) {
        const int o = P.bias(B[25 + l], 256, 16);
        if (l == 0) A.t2_bh = o;
    }
    A.t2_wo = P.out_layer(W[28], 3, 256, 64, hid);   A.t2_bo = P.bias4(B[28], 3);
    if (A.split_bf16) {
        for (int l = 0; l < 10; ++l) {
            const int o = P.dense_bf16(W[1 + l], 256, 256, 8, hid);
            if (l == 0) A.geo_qh = o;
        }
        for (int l = 0; l < 5; ++l) {
            const int o = P.dense_bf16(W[18 + l], 256, 256, 8, hid);
            if (l == 0) A.t1_qh = o;
        }
        A.t2_q0 = P.dense_bf16(W[24], 256, 295, 10, [](int t, int kq) { return t < 64 ? UvPacker::hidden(t, kq) : (t < 76 ? 256 + 4 * (t - 64) + kq : -1); });
        for (int l = 0; l < 3; ++l) {
            const int o = P.dense_bf16(W[25 + l], 256, 256, 8, hid);
            if (l == 0) A.t2_qh = o;
        }
    }
    P.align();
    auto bail = [&](int code) { ngf_uv_destroy(m); return code; };
    if (hipMalloc((void **)&m->w, P.buf.size() * sizeof(float)) != hipSuccess) return bail(fail(NGF_E_HIP, "hipMalloc(uv weights) failed"));
    if (hipMemcpyAsync(m->w, P.buf.data(), P.buf.size() * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
        return bail(fail(NGF_E_HIP, "uploading the packed UV weights failed"));
    if (hipMalloc((void **)&m->counters, kCounters * sizeof(unsigned)) != hipSuccess) return bail(fail(NGF_E_HIP, "hipMalloc(counters) failed"));
    A.w = m->w;
    *out = m;
    return NGF_OK;
}

extern "C" int ngf_uv_set_texture(ngf_uv *m, const float *tex, int32_t faces, int32_t H, int32_t W, int32_t C, int32_t mode, void *hip_stream)
{
    if (!m) return fail(NGF_E_ARG, "ngf_uv_set_texture: null model");
    if (m->tex) { (void)hipFree(m->tex); m->tex = nullptr; }
    m->proto.tex = nullptr;
    if (!tex) return NGF_OK;                       // cubemap_ = None: back to the plain texture branch
    const bool sphere = m->proto.sphere != 0;
    if (faces != (sphere ? 6 : 1) || H < 1 || W < 1 || C < 3 || C > 4 || (sphere && H != W) || mode < 0 || mode > 4)
        return fail(NGF_E_ARG, "ngf_uv_set_texture: expected %s, 3-4 channels, mode 0..4 (got faces=%d %dx%dx%d mode %d)",
                    sphere ? "a [6,R,R,C] cube map" : "a [H,W,C] square", faces, H, W, C, mode);
    const size_t n = (size_t)faces * H * W * C;
    HIP_TRY(hipMalloc((void **)&m->tex, n * sizeof(float)));
    HIP_TRY(hipMemcpyAsync(m->tex, tex, n * sizeof(float), hipMemcpyDeviceToDevice, (hipStream_t)hip_stream));
    m->proto.tex = m->tex; m->proto.tex_h = H; m->proto.tex_w = W; m->proto.tex_c = C; m->proto.tex_mode = mode;
    return NGF_OK;
}

extern "C" int ngf_uv_texture_edit(const ngf_uv *m, const float *uv, const float *orig, int64_t n, float *out, void *hip_stream)
{
    if (!m || !uv || !orig || !out || n < 0) return fail(NGF_E_ARG, "ngf_uv_texture_edit: bad argument");
    if (!m->proto.tex) return fail(NGF_E_ARG, "ngf_uv_texture_edit: no texture set (ngf_uv_set_texture)");
    if (n == 0) return NGF_OK;
    int64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    const UvArgs &A = m->proto;
    hipLaunchKernelGGL(uv_texture_edit_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)hip_stream, A.tex, A.tex_h, A.tex_w, A.tex_c, A.tex_mode, A.sphere,
                       uv, orig, n, out);
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

static int uv_launch(const ngf_uv *m, UvArgs &A, hipStream_t st)
{
    const unsigned slot = m->next_counter.fetch_add(1) % kCounters;
    A.ray_counter = m->counters + slot;
    HIP_TRY(hipMemsetAsync(A.ray_counter, 0, sizeof(unsigned), st));
    if (int rc = poison_lds(st)) return rc;
    int64_t grid = (A.R + 7) / 8;
    if (grid > (int64_t)m->num_cus) grid = m->num_cus;
    const size_t lds = (size_t)8 * kUvWaveLds * sizeof(float);
    // two rays per wave (every weight load feeds two MFMAs, 4 waves per CU) unless ngf_debug_set("uv_tiles", 1) (one ray per wave, 8 waves)
    int tiles = 2;
    if (knob(KNOB_UV_TILES) >= 0) tiles = knob(KNOB_UV_TILES);
    if (tiles == 2 && A.split_bf16) {
        HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(uv_render_kernel<2, true>), lds));
        hipLaunchKernelGGL((uv_render_kernel<2, true>), dim3((unsigned)grid), dim3(256), lds, st, A);
    } else if (tiles == 2) {
        HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(uv_render_kernel<2, false>), lds));
        hipLaunchKernelGGL((uv_render_kernel<2, false>), dim3((unsigned)grid), dim3(256), lds, st, A);
    } else if (tiles == 1 && A.split_bf16) {
        return fail(NGF_E_ARG, "NGF_UV_F_SPLIT_BF16 runs the two-rays-per-wave kernel only (knob uv_tiles must stay 2)");
    } else if (tiles == 1) {
        HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(uv_render_kernel<1, false>), lds));
        hipLaunchKernelGGL((uv_render_kernel<1, false>), dim3((unsigned)grid), dim3(512), lds, st, A);
    } else {
        return fail(NGF_E_ARG, "knob uv_tiles must be 1 or 2");
    }
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

extern "C" int ngf_uv_render(const ngf_uv *m, const float *campos_host, const float *raydir, const float *bg_host, const float *jitter_u,
                             int64_t n_rays, int32_t n_samples, float *color, float *transmittance, float *dbg_sigma, float *dbg_col,
                             uint64_t *stats, void *hip_stream)
{
    if (!m || !campos_host || !raydir || !jitter_u || !color || !transmittance) return fail(NGF_E_ARG, "ngf_uv_render: null argument");
    if (n_rays < 0 || n_samples <= 0) return fail(NGF_E_ARG, "ngf_uv_render: n_rays=%lld n_samples=%d", (long long)n_rays, n_samples);
    if (n_rays >= (int64_t)1 << 31) return fail(NGF_E_ARG, "ngf_uv_render: at most 2^31 - 1 rays per call");
    if ((dbg_sigma == nullptr) != (dbg_col == nullptr)) return fail(NGF_E_ARG, "ngf_uv_render: dbg_sigma and dbg_col go together");
    if (n_rays == 0) return NGF_OK;
    UvArgs A = m->proto;
    A.raydir = raydir; A.U = jitter_u; A.color = color; A.trans = transmittance; A.dbg_sigma = dbg_sigma; A.dbg_col = dbg_col;
    A.R = n_rays; A.S = n_samples; A.stats = (unsigned long long *)stats;
    for (int k = 0; k < 3; ++k) { A.campos[k] = campos_host[k]; A.bg[k] = bg_host ? bg_host[k] : 0.0f; }
    A.has_bg = bg_host ? 1 : 0;
    A.cam_dev = nullptr; A.bg_dev = nullptr; A.rays_per_cam = 1;
    return uv_launch(m, A, (hipStream_t)hip_stream);
}

extern "C" int ngf_uv_render_batch(const ngf_uv *m, const float *campos_dev, const float *raydir, const float *bg_dev, const float *jitter_u,
                                   int32_t n_cams, int64_t rays_per_cam, int32_t n_samples, float *color, float *transmittance,
                                   float *dbg_sigma, float *dbg_col, uint64_t *stats, void *hip_stream)
{
    if (!m || !campos_dev || !raydir || !jitter_u || !color || !transmittance) return fail(NGF_E_ARG, "ngf_uv_render_batch: null argument");
    if (n_cams < 0 || rays_per_cam < 0 || n_samples <= 0)
        return fail(NGF_E_ARG, "ngf_uv_render_batch: n_cams=%d rays_per_cam=%lld n_samples=%d", n_cams, (long long)rays_per_cam, n_samples);
    if ((dbg_sigma == nullptr) != (dbg_col == nullptr)) return fail(NGF_E_ARG, "ngf_uv_render_batch: dbg_sigma and dbg_col go together");
    const int64_t n_rays = (int64_t)n_cams * rays_per_cam;
    if (n_rays >= (int64_t)1 << 31) return fail(NGF_E_ARG, "ngf_uv_render_batch: at most 2^31 - 1 rays per call");
    if (n_rays == 0) return NGF_OK;
    UvArgs A = m->proto;
    A.raydir = raydir; A.U = jitter_u; A.color = color; A.trans = transmittance; A.dbg_sigma = dbg_sigma; A.dbg_col = dbg_col;
    A.R = n_rays; A.S = n_samples; A.stats = (unsigned long long *)stats;
    for (int k = 0; k < 3; ++k) A.campos[k] = A.bg[k] = 0.0f;
    A.has_bg = 0;
    A.cam_dev = campos_dev; A.bg_dev = bg_dev; A.rays_per_cam = (uint32_t)rays_per_cam;
    return uv_launch(m, A, (hipStream_t)hip_stream);
}
