/* ngf_oracle.h -- interface of the CPU restatement (test infrastructure; see ngf_oracle.c). */
#ifndef NGF_ORACLE_H
#define NGF_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

enum { NGF_MODEL_TRIPLANE = 0, NGF_MODEL_INFOINV = 1 };

/* All pointers are host pointers into the reference's own tensor layouts (NCHW, row-major Linear
 * weights [out,in]); nothing is re-packed. */
typedef struct ngf_oracle_model {
    int32_t model;          /* NGF_MODEL_* */
    int32_t gauge_on;       /* TriPlane: iteration >= gauge_start (Field.py:58) */
    int32_t infoinv;        /* InfoInv: forward(..., infoinv=...) (InfoInv/models/FieldBase.py:228) */
    int32_t dens_dim;       /* 16 (TriPlane) | 24 (InfoInv) channels per plane feeding density */
    int32_t app_feat;       /* 144 | 216 = 3 * colour channels per plane */
    float aabb[6];          /* aabb[0] (xyz), aabb[1] (xyz) */
    float near_, far_;
    float step;             /* stepSize (FieldBase.py:70) */
    float dscale;           /* distance_scale */
    float thr;              /* rayMarch_weight_thres */
    const float *plane[3];  /* plane_xy, plane_yz, plane_xz : [C, H, W] */
    int32_t plane_h[3], plane_w[3];
    const float *gauge[3];  /* gauge_xy, gauge_yz, gauge_xz : [2, H, W] (TriPlane only) */
    int32_t gauge_h[3], gauge_w[3];
    /* density decoder: TriPlane Linear(48,1) in (dens_w1, dens_b1); InfoInv MLP 72-32-32-1 */
    const float *dens_w1, *dens_b1, *dens_w2, *dens_b2, *dens_w3, *dens_b3;
    /* rgb_decoder: basis [F,F], mlp.0 [64,F+15], mlp.2 [64,64], mlp.4 [3,64] */
    const float *basis, *w1, *b1, *w2, *b2, *w3, *b3;
    /* optional alpha mask: np.packbits image of a [D,H,W] {0,1} volume + its own aabb */
    const uint8_t *mask_bits;
    int32_t mask_d, mask_h, mask_w;
    float mask_aabb[6];
} ngf_oracle_model;

/* optional per-sample intermediates for the first n_rays rays ([n_rays, S] row-major) */
typedef struct ngf_oracle_debug {
    int64_t n_rays;
    float *tmin;      /* [n_rays] */
    float *z;         /* z_vals */
    uint8_t *valid;   /* in-box (and alpha-mask) flag */
    float *sigma, *alpha, *weight;
    uint8_t *active;  /* weight > thr */
    float *rgb;       /* [n_rays, S, 3] per-sample colour (0 where inactive) */
    float *coords;    /* [n_rays, S, 6] gauge-shifted (xy, yz, xz) coordinates (0 where invalid) */
} ngf_oracle_debug;

int ngf_oracle_render(const ngf_oracle_model *m, const float *rays, int64_t n, int32_t S, int32_t white_bg,
                      const float *jitter, float *rgb, float *depth, ngf_oracle_debug *dbg, int32_t threads);
void ngf_oracle_bilerp2d(const float *plane, int H, int W, int C, const float *uv, int64_t n, float *out);
void ngf_oracle_mask_sample(const uint8_t *bits, int D, int H, int W, const float *q, int64_t n, float *out);
void ngf_oracle_color_at(const ngf_oracle_model *m, const float *coords, const float *dirs, int64_t n, float *rgb);
void ngf_oracle_density_at(const ngf_oracle_model *m, const float *coords, int64_t n, float *sigma);
void ngf_oracle_rgb_decode(const ngf_oracle_model *m, const float *feat, const float *dirs, int64_t n, float *rgb);

/* ---- UV-Mapping (NeuTex) colour path: UV-Mapping/model/model.py:30-50 and the sub-modules it calls ---------- */
typedef struct ngf_oracle_linear {
    const float *w, *b; /* nn.Linear weight [out,in], bias [out] */
    int32_t in_f, out_f;
    int32_t act;        /* 0 none, 1 ReLU, 2 LeakyReLU(0.2) */
} ngf_oracle_linear;

typedef struct ngf_oracle_uv_model {
    int32_t sphere;                 /* 1: uv = normalize(q) in R^3, 0: uv = tanh(q) in R^2 (gauge_fields.py:65-74) */
    ngf_oracle_linear geo[12];      /* GeometryMlpDecoder.block: 63-256, 10x 256-256, 256-1   (decoder.py:201-237) */
    ngf_oracle_linear gauge[5];     /* GaugeNetwork: 63-64, 64-128, 128-128 x2, 128-(3|2)     (gauge_fields.py:8-46) */
    ngf_oracle_linear tex1[6];      /* TextureMlpDecoder.block1: (63|42)-256, 5x 256-256     (decoder.py:19-25) */
    ngf_oracle_linear color1;       /* 256-3 */
    ngf_oracle_linear tex2[5];      /* block2: 295-256, 3x 256-256, 256-3                    (decoder.py:27-34) */
} ngf_oracle_uv_model;

/* campos [3], raydir [R,3], bg [3] or NULL, U [R,S] uniforms of the segment jitter (renderer.py:112-117),
 * out: color [R,3], transmittance [R]; optional per-sample debug [R,S]: sigma, uv [R,S,3], col [R,S,3], valid */
int ngf_oracle_uv_render(const ngf_oracle_uv_model *m, const float *campos, const float *raydir, const float *bg,
                         const float *U, int64_t R, int32_t S, float *color, float *trans, float *dbg_sigma,
                         float *dbg_uv, float *dbg_col, uint8_t *dbg_valid, int32_t threads);

#ifdef __cplusplus
}
#endif
#endif
