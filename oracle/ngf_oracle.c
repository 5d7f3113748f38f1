/*
 * ngf_oracle.c -- CPU restatement of the reference's TriPlane / InfoInv ray-march path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library, and only
 * as the checker.  The shipped path is the HIP library (neural-gauge-fields_amd/csrc).
 *
 * Parity pinning: the reference (fnzhan/Neural-Gauge-Fields) has no tests, golden vectors or
 * fixtures of its own, and its arithmetic lives in PyTorch ATen, which is
 * not vendored.  This restatement is therefore pinned against OUTPUTS OF THE REFERENCE ITSELF,
 * captured in the build container by importing /root/reference/{TriPlane,InfoInv}/models with
 * torch 2.10.0 CPU (tests/golden/make_golden.py -> tests/golden/ npz files;
 * tests/test_oracle_golden.py checks this file against them).
 *
 * Scalar fp32, one rounding per operation (build with -ffp-contract=off), same operation
 * order as the reference wherever the order is observable (position / mask arithmetic):
 *
 *   sample_ray        TriPlane/models/FieldBase.py:118-137
 *   alpha-mask test   TriPlane/models/FieldBase.py:33-40, 261-267   (ATen grid_sampler_3d, zeros pad)
 *   normalize_coord   TriPlane/models/FieldBase.py:88-89
 *   compute_gauge     TriPlane/models/Field.py:53-75               (ATen grid_sampler_2d, zeros pad)
 *   compute_density   TriPlane/models/Field.py:77-91, 48-50        InfoInv/models/Field.py:52-70
 *   raw2alpha         TriPlane/models/FieldBase.py:12-19
 *   compute_rgb       TriPlane/models/Field.py:93-105               InfoInv/models/Field.py:72-89
 *   rgb_decoder       TriPlane/models/networks.py:12-32, 205-216    InfoInv/models/networks.py:34-54
 *   compositing       TriPlane/models/FieldBase.py:288-306
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ngf_oracle.h"

/* ---- ATen grid_sampler_2d, bilinear, align_corners=True, padding_mode='zeros' ------------- */
/* plane: [C,H,W] (NCHW with N=1); (u,v): u indexes W, v indexes H; channels [c0,c1) -> out     */
static void bilerp2d(const float *plane, int H, int W, int c0, int c1, float u, float v, float *out)
{
    float px = ((u + 1.0f) / 2.0f) * (float)(W - 1);
    float py = ((v + 1.0f) / 2.0f) * (float)(H - 1);
    float fx0 = floorf(px), fy0 = floorf(py);
    float wx1 = px - fx0, wx0 = 1.0f - wx1;
    float wy1 = py - fy0, wy0 = 1.0f - wy1;
    float w00 = wx0 * wy0, w10 = wx1 * wy0, w01 = wx0 * wy1, w11 = wx1 * wy1;
    /* guard the float->int conversion for far-out-of-range coordinates */
    int in_x0 = (fx0 >= 0.0f) && (fx0 <= (float)(W - 1));
    int in_x1 = (fx0 + 1.0f >= 0.0f) && (fx0 + 1.0f <= (float)(W - 1));
    int in_y0 = (fy0 >= 0.0f) && (fy0 <= (float)(H - 1));
    int in_y1 = (fy0 + 1.0f >= 0.0f) && (fy0 + 1.0f <= (float)(H - 1));
    int x0 = in_x0 ? (int)fx0 : 0, x1 = in_x1 ? (int)fx0 + 1 : 0;
    int y0 = in_y0 ? (int)fy0 : 0, y1 = in_y1 ? (int)fy0 + 1 : 0;
    for (int c = c0; c < c1; ++c) {
        const float *p = plane + (size_t)c * H * W;
        float acc = 0.0f;
        if (in_x0 && in_y0) acc += p[(size_t)y0 * W + x0] * w00;
        if (in_x1 && in_y0) acc += p[(size_t)y0 * W + x1] * w10;
        if (in_x0 && in_y1) acc += p[(size_t)y1 * W + x0] * w01;
        if (in_x1 && in_y1) acc += p[(size_t)y1 * W + x1] * w11;
        out[c - c0] = acc;
    }
}

/* ---- ATen grid_sampler_3d on a {0,1} volume stored as np.packbits bits -------------------- */
static int mask_bit(const uint8_t *bits, int D, int H, int W, int z, int y, int x)
{
    if (x < 0 || y < 0 || z < 0 || x >= W || y >= H || z >= D) return 0;
    size_t idx = ((size_t)z * H + y) * W + x;
    return (bits[idx >> 3] >> (7 - (idx & 7))) & 1;
}

static float trilerp_mask(const uint8_t *bits, int D, int H, int W, float qx, float qy, float qz)
{
    float ix = ((qx + 1.0f) / 2.0f) * (float)(W - 1);
    float iy = ((qy + 1.0f) / 2.0f) * (float)(H - 1);
    float iz = ((qz + 1.0f) / 2.0f) * (float)(D - 1);
    float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    if (!(fx >= -2.0f && fx <= (float)W + 1.0f && fy >= -2.0f && fy <= (float)H + 1.0f &&
          fz >= -2.0f && fz <= (float)D + 1.0f))
        return 0.0f;
    int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    float ex = fx + 1.0f, ey = fy + 1.0f, ez = fz + 1.0f; /* "bse" corner coordinates */
    float wx[2] = {ex - ix, ix - fx}, wy[2] = {ey - iy, iy - fy}, wz[2] = {ez - iz, iz - fz};
    float acc = 0.0f;
    for (int dz = 0; dz < 2; ++dz)
        for (int dy = 0; dy < 2; ++dy)
            for (int dx = 0; dx < 2; ++dx)
                if (mask_bit(bits, D, H, W, z0 + dz, y0 + dy, x0 + dx))
                    acc += wx[dx] * wy[dy] * wz[dz];
    return acc;
}

static float softplus_shift(float f)
{
    float u = f + (-10.0f); /* feature2density: F.softplus(x + density_shift), shift hard-coded -10 */
    return u > 20.0f ? u : log1pf(expf(u));
}

/* positional_encoding(positions[3], F): [x*2^0..x*2^(F-1), y*.., z*..] -> sin(all), cos(all)   */
static void posenc3(const float p[3], int F, float *out)
{
    int n = 3 * F;
    for (int k = 0; k < 3; ++k)
        for (int f = 0; f < F; ++f) {
            float a = p[k] * (float)(1 << f);
            out[k * F + f] = sinf(a);
            out[n + k * F + f] = cosf(a);
        }
}

static void linear(const float *W, const float *b, int out_f, int in_f, const float *x, float *y, int relu)
{
    for (int o = 0; o < out_f; ++o) {
        float acc = 0.0f;
        const float *w = W + (size_t)o * in_f;
        for (int i = 0; i < in_f; ++i) acc += w[i] * x[i];
        if (b) acc += b[o];
        y[o] = (relu && acc < 0.0f) ? 0.0f : acc;
    }
}

#define NGF_MAX_FEAT 216

/* rgb_decoder.forward (networks.py:25-32) */
static void rgb_decode(const ngf_oracle_model *m, const float *feat, const float d[3], float rgb[3])
{
    int F = m->app_feat;
    float u[NGF_MAX_FEAT + 15], h1[64], h2[64], o[3];
    linear(m->basis, NULL, F, F, feat, u, 0);
    u[F + 0] = d[0]; u[F + 1] = d[1]; u[F + 2] = d[2];
    posenc3(d, 2, u + F + 3);
    linear(m->w1, m->b1, 64, F + 15, u, h1, 1);
    linear(m->w2, m->b2, 64, 64, h1, h2, 1);
    linear(m->w3, m->b3, 3, 64, h2, o, 0);
    for (int c = 0; c < 3; ++c) rgb[c] = 1.0f / (1.0f + expf(-o[c]));
}

/* density from the three (gauge-shifted) plane coordinates; xyz = un-shifted normalised position */
static float density_at(const ngf_oracle_model *m, const float t[3][2])
{
    int dd = m->dens_dim;
    float f[72];
    for (int p = 0; p < 3; ++p)
        bilerp2d(m->plane[p], m->plane_h[p], m->plane_w[p], 0, dd, t[p][0], t[p][1], f + p * dd);
    if (m->model == NGF_MODEL_TRIPLANE) {
        float acc = 0.0f;
        for (int i = 0; i < 3 * dd; ++i) acc += m->dens_w1[i] * f[i];
        acc += m->dens_b1[0];
        return softplus_shift(acc);
    }
    /* InfoInv (InfoInv/models/Field.py:52-70): xyz = cat(xy, yz[:,1:]) */
    if (m->infoinv) {
        float xyz[3] = {t[0][0], t[0][1], t[1][1]}, pe[24];
        posenc3(xyz, 4, pe);
        for (int p = 0; p < 3; ++p)
            for (int c = 0; c < dd; ++c) f[p * dd + c] = f[p * dd + c] * pe[c];
    }
    float h1[32], h2[32], o[1];
    linear(m->dens_w1, m->dens_b1, 32, 3 * dd, f, h1, 1);
    linear(m->dens_w2, m->dens_b2, 32, 32, h1, h2, 1);
    linear(m->dens_w3, m->dens_b3, 1, 32, h2, o, 0);
    return softplus_shift(o[0]);
}

static void color_at(const ngf_oracle_model *m, const float t[3][2], const float d[3], float rgb[3])
{
    int dd = m->dens_dim, ad = m->app_feat / 3;
    float f[NGF_MAX_FEAT];
    for (int p = 0; p < 3; ++p)
        bilerp2d(m->plane[p], m->plane_h[p], m->plane_w[p], dd, dd + ad, t[p][0], t[p][1], f + p * ad);
    if (m->model == NGF_MODEL_INFOINV && m->infoinv) {
        float xyz[3] = {t[0][0], t[0][1], t[1][1]}, pe[72];
        posenc3(xyz, 12, pe);
        for (int p = 0; p < 3; ++p)
            for (int c = 0; c < ad; ++c) f[p * ad + c] = f[p * ad + c] * pe[c];
    }
    rgb_decode(m, f, d, rgb);
}

/* compute_gauge (Field.py:53-75) / transform (InfoInv Field.py:43-50) */
static void gauge_coords(const ngf_oracle_model *m, const float x[3], float t[3][2])
{
    float xy[2] = {x[0], x[1]}, yz[2] = {x[1], x[2]}, xz[2] = {x[0], x[2]};
    if (m->model == NGF_MODEL_TRIPLANE && m->gauge_on) {
        float dxy[2], dyz[2], dxz[2];
        bilerp2d(m->gauge[0], m->gauge_h[0], m->gauge_w[0], 0, 2, xy[0], xy[1], dxy);
        bilerp2d(m->gauge[1], m->gauge_h[1], m->gauge_w[1], 0, 2, yz[0], yz[1], dyz);
        bilerp2d(m->gauge[2], m->gauge_h[2], m->gauge_w[2], 0, 2, xz[0], xz[1], dxz);
        t[0][0] = (xy[0] + dxy[0]) + dxz[0]; t[0][1] = (xy[1] + dxy[1]) + dyz[0];
        t[1][0] = (yz[0] + dyz[0]) + dxy[1]; t[1][1] = (yz[1] + dyz[1]) + dxz[1];
        t[2][0] = (xz[0] + dxz[0]) + dxy[0]; t[2][1] = (xz[1] + dxz[1]) + dyz[1];
    } else {
        t[0][0] = xy[0]; t[0][1] = xy[1];
        t[1][0] = yz[0]; t[1][1] = yz[1];
        t[2][0] = xz[0]; t[2][1] = xz[1];
    }
}

/* one ray: Base.forward (FieldBase.py:251-312) restricted to a single row */
static void render_ray(const ngf_oracle_model *m, const float *ray, int S, int white_bg, const float *jitter,
                       float *rgb_out, float *depth_out, ngf_oracle_debug *dbg, int64_t r)
{
    const float *o = ray, *d = ray + 3;
    const float *a0 = m->aabb, *a1 = m->aabb + 3;
    /* sample_ray (FieldBase.py:118-137) */
    float tmin = -INFINITY;
    for (int k = 0; k < 3; ++k) {
        float vec = (d[k] == 0.0f) ? 1e-6f : d[k];
        float ra = (a1[k] - o[k]) / vec, rb = (a0[k] - o[k]) / vec;
        float mn = ra < rb ? ra : rb;
        if (mn > tmin) tmin = mn;
    }
    if (tmin < m->near_) tmin = m->near_;
    if (tmin > m->far_) tmin = m->far_;
    float inv[3], minv[3];
    for (int k = 0; k < 3; ++k) {
        inv[k] = 2.0f / (a1[k] - a0[k]);                          /* invaabbSize, FieldBase.py:67 */
        minv[k] = 1.0f / (m->mask_aabb[3 + k] - m->mask_aabb[k]) * 2.0f; /* invgridSize, :29 */
    }
    float jit = jitter ? jitter[r] : 0.0f;                          /* is_train: rng += U[0,1) per ray */

    float T = 1.0f, acc = 0.0f, cr = 0.0f, cg = 0.0f, cb = 0.0f, dep = 0.0f;
    for (int i = 0; i < S; ++i) {
        float z = tmin + m->step * ((float)i + jit);
        float zn = tmin + m->step * ((float)(i + 1) + jit);
        float dist = (i < S - 1) ? (zn - z) : 0.0f;
        float p[3];
        int valid = 1;
        for (int k = 0; k < 3; ++k) {
            p[k] = o[k] + d[k] * z;
            if (a0[k] > p[k] || p[k] > a1[k]) valid = 0;
        }
        if (valid && m->mask_bits) {
            float q[3];
            for (int k = 0; k < 3; ++k) q[k] = (p[k] - m->mask_aabb[k]) * minv[k] - 1.0f;
            float a = trilerp_mask(m->mask_bits, m->mask_d, m->mask_h, m->mask_w, q[0], q[1], q[2]);
            if (!(a > 0.0f)) valid = 0;
        }
        float sigma = 0.0f, t[3][2] = {{0, 0}, {0, 0}, {0, 0}};
        if (valid) {
            float x[3];
            for (int k = 0; k < 3; ++k) x[k] = (p[k] - a0[k]) * inv[k] - 1.0f;
            gauge_coords(m, x, t);
            sigma = density_at(m, t);
        }
        /* raw2alpha (FieldBase.py:12-19) with dist * distance_scale */
        float alpha = 1.0f - expf(-sigma * (dist * m->dscale));
        float w = alpha * T;
        T = T * ((1.0f - alpha) + 1e-10f);
        int act = w > m->thr;
        float c[3] = {0.0f, 0.0f, 0.0f};
        if (act) color_at(m, t, d, c);
        acc += w;
        cr += w * c[0]; cg += w * c[1]; cb += w * c[2];
        dep += w * z;
        if (dbg && r < dbg->n_rays) {
            size_t q = (size_t)r * S + i;
            if (dbg->z) dbg->z[q] = z;
            if (dbg->valid) dbg->valid[q] = (uint8_t)valid;
            if (dbg->sigma) dbg->sigma[q] = sigma;
            if (dbg->alpha) dbg->alpha[q] = alpha;
            if (dbg->weight) dbg->weight[q] = w;
            if (dbg->active) dbg->active[q] = (uint8_t)act;
            if (dbg->rgb) { dbg->rgb[3 * q] = c[0]; dbg->rgb[3 * q + 1] = c[1]; dbg->rgb[3 * q + 2] = c[2]; }
            if (dbg->coords) for (int k = 0; k < 6; ++k) dbg->coords[6 * q + k] = t[k / 2][k % 2];
        }
    }
    if (dbg && r < dbg->n_rays && dbg->tmin) dbg->tmin[r] = tmin;
    float out[3] = {cr, cg, cb};
    for (int c = 0; c < 3; ++c) {
        float v = out[c];
        if (white_bg) v = v + (1.0f - acc);
        v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
        rgb_out[c] = v;
    }
    *depth_out = dep + (1.0f - acc) * ray[5]; /* the reference's rays_chunk[..., -1] quirk, FieldBase.py:306 */
}

int ngf_oracle_render(const ngf_oracle_model *m, const float *rays, int64_t n, int32_t S, int32_t white_bg,
                      const float *jitter, float *rgb, float *depth, ngf_oracle_debug *dbg, int32_t threads)
{
    if (!m || !rays || !rgb || !depth || S <= 0) return 1;
    if (m->app_feat > NGF_MAX_FEAT || 3 * m->dens_dim > 72) return 2;
    (void)threads;
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t r = 0; r < n; ++r)
        render_ray(m, rays + 6 * r, S, white_bg, jitter, rgb + 3 * r, depth + r, dbg, r);
    return 0;
}

/* stand-alone pieces, exported so tests can pin them one by one */
void ngf_oracle_bilerp2d(const float *plane, int H, int W, int C, const float *uv, int64_t n, float *out)
{
    for (int64_t i = 0; i < n; ++i) bilerp2d(plane, H, W, 0, C, uv[2 * i], uv[2 * i + 1], out + (size_t)i * C);
}

void ngf_oracle_mask_sample(const uint8_t *bits, int D, int H, int W, const float *q, int64_t n, float *out)
{
    for (int64_t i = 0; i < n; ++i) out[i] = trilerp_mask(bits, D, H, W, q[3 * i], q[3 * i + 1], q[3 * i + 2]);
}

void ngf_oracle_rgb_decode(const ngf_oracle_model *m, const float *feat, const float *dirs, int64_t n, float *rgb)
{
    for (int64_t i = 0; i < n; ++i) rgb_decode(m, feat + (size_t)i * m->app_feat, dirs + 3 * i, rgb + 3 * i);
}

/* compute_rgb / compute_density for explicit (already gauge-shifted) plane coordinates [n,6] */
void ngf_oracle_color_at(const ngf_oracle_model *m, const float *coords, const float *dirs, int64_t n, float *rgb)
{
    for (int64_t i = 0; i < n; ++i) {
        float t[3][2];
        for (int k = 0; k < 6; ++k) t[k / 2][k % 2] = coords[6 * i + k];
        color_at(m, t, dirs + 3 * i, rgb + 3 * i);
    }
}

void ngf_oracle_density_at(const ngf_oracle_model *m, const float *coords, int64_t n, float *sigma)
{
    for (int64_t i = 0; i < n; ++i) {
        float t[3][2];
        for (int k = 0; k < 6; ++k) t[k / 2][k % 2] = coords[6 * i + k];
        sigma[i] = density_at(m, t);
    }
}

/* ======================= UV-Mapping (NeuTex) colour path ======================================================
 * cube_ray_generation   UV-Mapping/model/renderer.py:79-141  (jitter = 0.05 always, model.py:30; the uniforms are an
 *                       input here because the reference draws them with torch.rand inside the call)
 * GeometryMlpDecoder    UV-Mapping/model/decoder.py:219-237      GaugeTransform  UV-Mapping/model/gauge_fields.py:60-74
 * TextureMlpDecoder     UV-Mapping/model/decoder.py:63-78 (clamp=False, no target texture: model.py:22-23)
 * ray_march             UV-Mapping/model/renderer.py:222-247     simple_tone_map :7-8
 * positional_encoding   UV-Mapping/util.py:427-438
 */
static void uv_linear(const ngf_oracle_linear *L, const float *x, float *y)
{
    for (int o = 0; o < L->out_f; ++o) {
        float acc = 0.0f;
        const float *w = L->w + (size_t)o * L->in_f;
        for (int i = 0; i < L->in_f; ++i) acc += w[i] * x[i];
        if (L->b) acc += L->b[o];
        if (L->act == 1) acc = acc < 0.0f ? 0.0f : acc;
        else if (L->act == 2) acc = acc > 0.0f ? acc : 0.2f * acc;
        y[o] = acc;
    }
}

static void posenc_n(const float *p, int D, int F, float *out) /* [sin(D*F), cos(D*F)], per-dimension-major */
{
    int n = D * F;
    for (int k = 0; k < D; ++k)
        for (int f = 0; f < F; ++f) {
            float a = p[k] * (float)(1 << f);
            out[k * F + f] = sinf(a);
            out[n + k * F + f] = cosf(a);
        }
}

static float softplus20(float u) { return u > 20.0f ? u : log1pf(expf(u)); }

static void uv_sample(const ngf_oracle_uv_model *m, const float p[3], const float v[3], float *sigma, float uv[3], float col[3])
{
    float a[320], b[320];
    /* geometry: softplus(MLP([p, PE10(p)])) */
    a[0] = p[0]; a[1] = p[1]; a[2] = p[2];
    posenc_n(p, 3, 10, a + 3);
    float in63[63];
    memcpy(in63, a, sizeof(in63));
    float *x = a, *y = b;
    for (int l = 0; l < 12; ++l) { uv_linear(&m->geo[l], x, y); float *t = x; x = y; y = t; }
    *sigma = softplus20(x[0]);
    /* gauge */
    x = in63; y = b;
    float g1[128], g2[128];
    uv_linear(&m->gauge[0], in63, g1);
    uv_linear(&m->gauge[1], g1, g2);
    uv_linear(&m->gauge[2], g2, g1);
    uv_linear(&m->gauge[3], g1, g2);
    float q[3] = {0, 0, 0};
    uv_linear(&m->gauge[4], g2, q);
    int ud = m->sphere ? 3 : 2;
    if (m->sphere) {
        float nrm = sqrtf((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]);
        if (nrm < 1e-12f) nrm = 1e-12f;
        for (int k = 0; k < 3; ++k) uv[k] = q[k] / nrm;
    } else {
        uv[0] = tanhf(q[0]); uv[1] = tanhf(q[1]); uv[2] = 0.0f;
    }
    /* texture */
    for (int k = 0; k < ud; ++k) a[k] = uv[k];
    posenc_n(uv, ud, 10, a + ud);
    x = a; y = b;
    for (int l = 0; l < 6; ++l) { uv_linear(&m->tex1[l], x, y); float *t = x; x = y; y = t; }
    float c1[3];
    uv_linear(&m->color1, x, c1);
    for (int k = 0; k < 3; ++k) c1[k] = softplus20(c1[k]);
    float in2[295];
    memcpy(in2, x, 256 * sizeof(float));
    in2[256] = v[0]; in2[257] = v[1]; in2[258] = v[2];
    posenc_n(v, 3, 6, in2 + 259);
    x = in2; y = (x == a) ? b : a;
    float h1[256], h2[256];
    uv_linear(&m->tex2[0], in2, h1);
    uv_linear(&m->tex2[1], h1, h2);
    uv_linear(&m->tex2[2], h2, h1);
    uv_linear(&m->tex2[3], h1, h2);
    float c2[3];
    uv_linear(&m->tex2[4], h2, c2);
    for (int k = 0; k < 3; ++k) { float c = c1[k] + c2[k]; col[k] = c < 0.0f ? 0.0f : c; }
}

int ngf_oracle_uv_render(const ngf_oracle_uv_model *m, const float *campos, const float *raydir, const float *bg,
                         const float *U, int64_t R, int32_t S, float *color, float *trans, float *dbg_sigma,
                         float *dbg_uv, float *dbg_col, uint8_t *dbg_valid, int32_t threads)
{
    if (!m || !campos || !raydir || !U || !color || !trans || S <= 0 || S > 1024) return 1;
    (void)threads;
    const float dt = (float)(1.0 * 2 / S);                 /* domain_size * 2 / point_count */
    const float dtj = (float)((1.0 * 2 / S) * 0.05);       /* dt * jitter (python floats) */
#ifdef _OPENMP
#pragma omp parallel for schedule(dynamic, 4) num_threads(threads > 0 ? threads : 1)
#endif
    for (int64_t r = 0; r < R; ++r) {
        const float *d = raydir + 3 * r;
        float t1[3], t2[3];
        for (int k = 0; k < 3; ++k) { t1[k] = (-1.0f - campos[k]) / d[k]; t2[k] = (1.0f - campos[k]) / d[k]; }
        float tmin = fmaxf(fminf(t1[0], t2[0]), fmaxf(fminf(t1[1], t2[1]), fminf(t1[2], t2[2])));
        float tmax = fminf(fmaxf(t1[0], t2[0]), fminf(fmaxf(t1[1], t2[1]), fmaxf(t1[2], t2[2])));
        float t = (tmin < tmax) ? tmin : 0.0f;
        if (t < 0.0f) t = 0.0f;
        /* torch.cumsum on CPU accumulates float32 inputs in double (at::acc_type<float,false>) and rounds each
           prefix to float; the positions feed sin(512 x), so this rounding is visible at the 1e-4 level */
        double cum = 0.0;
        float Tacc = 1.0f, rc[3] = {0, 0, 0};
        float e_prev = t + 0.0f;
        for (int i = 0; i < S; ++i) {
            float seg = dt + dtj * (U[r * S + i] - 0.5f);
            cum = cum + (double)seg;
            float e_next = t + (float)cum;
            float mid = (e_prev + e_next) / 2.0f;
            e_prev = e_next;
            float p[3];
            int valid = 1;
            for (int k = 0; k < 3; ++k) {
                p[k] = campos[k] + d[k] * mid;
                if (!(p[k] > -1.0f && p[k] < 1.0f)) valid = 0;
            }
            float sigma, uv[3], col[3];
            uv_sample(m, p, d, &sigma, uv, col);
            float sg = sigma * (float)valid;
            float op = 1.0f - expf(-sg * seg);
            float w = op * Tacc;
            Tacc = Tacc * ((1.0f - op) + 1e-10f);
            for (int k = 0; k < 3; ++k) rc[k] += col[k] * w;
            size_t q = (size_t)r * S + i;
            if (dbg_sigma) dbg_sigma[q] = sigma;
            if (dbg_valid) dbg_valid[q] = (uint8_t)valid;
            if (dbg_uv) for (int k = 0; k < 3; ++k) dbg_uv[3 * q + k] = uv[k];
            if (dbg_col) for (int k = 0; k < 3; ++k) dbg_col[3 * q + k] = col[k];
        }
        for (int k = 0; k < 3; ++k) {
            float c = rc[k];
            if (bg) c += bg[k] * Tacc;
            c = powf(c * 1.0f + 1e-5f, (float)(1.0 / 2.2));
            color[3 * r + k] = c < 0.0f ? 0.0f : (c > 1.0f ? 1.0f : c);
        }
        trans[r] = Tacc;
    }
    return 0;
}
