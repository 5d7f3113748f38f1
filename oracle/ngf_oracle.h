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

#ifdef __cplusplus
}
#endif
#endif
