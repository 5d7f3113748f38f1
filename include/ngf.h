/*
 * ngf.h -- C ABI of libngf_hip.so: the MI355X (gfx950) implementation of the reference's
 * TriPlane / InfoInv ray-march hot path (fnzhan/Neural-Gauge-Fields).
 *
 * The reference has no native code and no FFI: its boundary for this path is the Python call
 *     field(rays_chunk, white_bg, is_train, N_samples, iteration)  -> {'rgb_map','depth_map'}
 *         TriPlane/models/FieldBase.py:251-312   (InfoInv/models/FieldBase.py:228-282)
 *     renderer(rays, field, chunk, N_samples, white_bg, is_train, device) -> (rgb, depth)
 *         TriPlane/main.py:60-71                 (InfoInv/main.py:61-72)
 * over parameters created by TriPlane.init_model (TriPlane/models/Field.py:17-32;
 * InfoInv/models/Field.py:14-24) and stored in the checkpoint dict of Base.save/load
 * (TriPlane/models/FieldBase.py:94-116).  Each entry point below names the piece of that
 * boundary it replaces.  INTEGRATION.md shows the ctypes binding a maintainer adds on the
 * reference side.
 *
 * Conventions: plain pointers and sizes only; every data pointer is a DEVICE pointer unless
 * stated otherwise; all work is enqueued on the caller's HIP stream (hipStream_t passed as
 * void*).  What waits for the device today: ngf_field_create / ngf_uv_create wait for THEIR
 * stream once (they copy ~150 KB of MLP weights to the host to permute them into the LDS image);
 * ngf_field_destroy waits for the streams the handle was used on -- one event per stream, on the
 * handle's own device, no device-wide synchronisation -- before it parks the buffers in the pool;
 * ngf_uv_destroy / ngf_trainer_destroy / ngf_pool_trim call hipFree (which waits for the device);
 * the *_host out-parameters of the training calls wait for their stream.  Every render / march /
 * decode / alpha / filter / eval / ray-generation call is asynchronous.
 * Devices: a handle belongs to the device that was current in the calling thread when it was
 * created; its calls must be made with that device current (the launches go to the caller's
 * stream), its destroy may be called under any current device.
 * Return value 0 = success, otherwise an NGF_E_* code and ngf_last_error() holds a message for
 * the calling thread.  No C++ exception crosses the ABI.
 */
#ifndef NGF_H
#define NGF_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NGF_ABI_VERSION 5      /* 5: handles remember their device and streams (ngf_field_destroy needs no current-device contract), ngf_pool_set_limit / ngf_pool_bytes, NGF_E_STALE from ngf_train_backward_grad, ngf_train_get_grads / ngf_train_adam_ext, ngf_field_render_image; 4: ngf_train_forward / ngf_train_backward_grad (the step in two calls, d loss / d rgb_map handed in), trainers without Adam moments; 3: ngf_train_overflow_count + speculative rows (ngf_train_desc.chunk_samples < 0), ngf_train_backward2 (loss_len travels with the call; ngf_train_backward is back to its ABI-1 contract of ONE double); 2: ngf_uv_desc.flags (was padding), ngf_uv_render_batch, ngf_debug_dirty_lds, trainer chunk_samples = 0 means the whole batch */

enum { NGF_OK = 0, NGF_E_ARG = 1, NGF_E_HIP = 2, NGF_E_UNSUPPORTED = 3,
       NGF_E_STALE = 4 /* ngf_train_backward_grad: the ticket is not the trainer's last forward -- run the forward again; nothing else returns it */ };
enum { NGF_MODEL_TRIPLANE = 0, NGF_MODEL_INFOINV = 1 };

/* ngf_field_desc.flags */
enum {
    NGF_F_BAKE_DENSITY = 1, /* TriPlane: pre-compose density_decoder Linear(48,1) with the 16 density
                              channels of each plane (exact algebra: a Linear commutes with bilinear
                              interpolation); the march then gathers 1 instead of 16 channels per tap.
                              Optimisation level 2; the Python boundary (ngf_amd.fieldbase.Base) sets it by default since
                              round 3, the C ABI leaves the choice to the caller. */
    NGF_F_NO_FOLD = 4,      /* level 0 of the optimisation ladder: rgb_decoder exactly as written (networks.py:25-30) -- `basis` is its own
                              144 x 144 matrix stage and the view inputs enter layer 1 per sample.  Without this flag layer 1 is
                              pre-composed with `basis` (W1' = W1[:, :F] . basis, fp64 accumulate) and, for small tiles, its
                              view-direction part is evaluated once per ray.  For measuring what the pre-compositions buy. */
    NGF_F_SPLIT_BF16 = 8,   /* colour-MLP products on v_mfma_f32_16x16x32_bf16 with every fp32 operand split into three bf16 terms and
                              fp32 accumulation: fp32-level error (dropped cross terms < 2^-24 of a product) at ~0.4 x the matrix
                              cycles of the fp32 MFMA path, which on gfx950 runs on the vector datapath.  Not bit-identical to the
                              default (different summation tree); opt-in, see DESIGN.md.  Combinations: alone or with NGF_F_BAKE_DENSITY it
                              covers layers 1 and 2 of the pre-composed formulation (levels 1 / 2); with NGF_F_BAKE_DENSITY | NGF_F_BAKE_COLOR
                              (round 5) it is level 3 with LAYER 2 -- all the matrix work level 3 has left -- on the bf16 pipe; not with
                              NGF_F_NO_FOLD, not with NGF_F_BAKE_COLOR alone. */
    NGF_F_BAKE_COLOR = 2    /* pre-compose rgb_decoder layer 1 (W1[:, :F] . basis, no activation in between:
                              networks.py:17,26-30) with the colour channels of each plane: colour planes
                              become 64-channel layer-1 pre-activation planes, the shade pass keeps only the
                              view-direction inputs and layers 2-3 on the matrix pipe.  Same algebra. */
};

/* Parameter set of one field, in the reference's own tensor layouts (NCHW planes, nn.Linear
 * weights [out,in], all float32, contiguous).  Mirrors TriPlane.init_model (Field.py:17-32) /
 * InfoInv init_model (InfoInv/models/Field.py:14-24) + the scalars of Base.__init__/init_para
 * (FieldBase.py:45-74) + the optional AlphaGridMask (FieldBase.py:22-40, ckpt keys
 * 'alphaMask.shape' / 'alphaMask.mask' (np.packbits) / 'alphaMask.aabb', FieldBase.py:104-108). */
typedef struct ngf_field_desc {
    int32_t model;                 /* NGF_MODEL_* */
    int32_t flags;                 /* NGF_F_* */
    int32_t plane_c;               /* 64 (TriPlane) | 96 (InfoInv) */
    int32_t dens_dim;              /* 16 | 24: leading channels of each plane that feed density */
    const float *plane[3];         /* plane_xy [1,C,Ny,Nx], plane_yz [1,C,Nz,Ny], plane_xz [1,C,Nz,Nx] */
    int32_t plane_h[3], plane_w[3];
    const float *gauge[3];         /* gauge_xy, gauge_yz, gauge_xz [1,2,H,W]; NULL for InfoInv */
    int32_t gauge_h[3], gauge_w[3];
    /* density decoder: TriPlane Linear(48,1) -> (dens_w1 [1,48], dens_b1 [1]), others NULL;
       InfoInv density_decoder.mlp.{0,2,4}: [32,72],[32]; [32,32],[32]; [1,32],[1] */
    const float *dens_w1, *dens_b1, *dens_w2, *dens_b2, *dens_w3, *dens_b3;
    /* rgb_decoder: basis.weight [F,F] (F = 3*(plane_c-dens_dim)); mlp.0 [64,F+15],[64];
       mlp.2 [64,64],[64]; mlp.4 [3,64],[3]   (networks.py:12-32) */
    const float *basis, *w1, *b1, *w2, *b2, *w3, *b3;
    float aabb[6];                 /* aabb[0] xyz, aabb[1] xyz */
    float near_, far_;             /* near_far */
    float step;                    /* stepSize (FieldBase.py:70) -- computed by the caller in float32 */
    float distance_scale;          /* 25 */
    float weight_thres;            /* rayMarch_weight_thres, 1e-4 */
    /* optional alpha mask: the checkpoint's np.packbits image of the [D,H,W] volume (bit = occupied).  ngf_field_create derives two device images
     * from it (the 8 corner bits of every trilinear cell as one byte, (D+1)(H+1)(W+1) bytes -- one gather per sample decides sample_alpha > 0,
     * FieldBase.py:33-40,263-267 -- and a block image for the march's empty-space skipping); (D+1)(H+1)(W+1) must stay below 2^32 */
    const uint8_t *mask_bits;      /* NULL = no mask */
    int32_t mask_d, mask_h, mask_w;
    float mask_aabb[6];
} ngf_field_desc;

typedef struct ngf_field ngf_field; /* opaque: packed (channel-last, zero-bordered) textures + MLP image */

/* Replaces: model construction + load_state_dict (TriPlane/main.py:34-38, FieldBase.py:111-116).
 * Re-packs the parameters into the kernel's HBM layout (DESIGN.md "Data layout"); the source
 * tensors are only read during this call and may be freed afterwards.  Call again (and destroy
 * the old handle) whenever the parameters change. */
int ngf_field_create(const ngf_field_desc *desc, ngf_field **out, void *hip_stream);
int ngf_field_destroy(ngf_field *f);

/* Replaces: Base.forward (FieldBase.py:251-312 / InfoInv FieldBase.py:228-282) for n rays, i.e. also
 * the whole chunk loop of renderer (main.py:60-71) when n is the full frame.
 *   rays      [n,6] float32 (origin, direction) row-major
 *   n_samples S > 0 (the caller resolves N_samples<=0 to nSamples)
 *   white_bg  rgb += 1 - acc                                       (FieldBase.py:299-300)
 *   mode      TriPlane: 1 = gauge on (iteration >= gauge_start), 0 = identity split (Field.py:58,73)
 *             InfoInv : 1 = infoinv sinusoidal modulation on, 0 = off  (InfoInv Field.py:63,83)
 *   jitter    NULL (eval) or [n] per-ray U[0,1) offsets of the sample index (is_train, FieldBase.py:128-130)
 *   rgb [n,3], depth [n]  outputs
 *   stats     NULL or 4 x uint64: += {in-box samples, active (weight>thr) samples, MLP passes, rays}
 */
int ngf_field_render(const ngf_field *f, const float *rays, int64_t n, int32_t n_samples, int32_t white_bg,
                     int32_t mode, const float *jitter, float *rgb, float *depth, uint64_t *stats,
                     void *hip_stream);
/* ABI 5: the same call for a ray list that is an IMAGE -- rays [n,6] row-major with row_width rays per image row, which is what the reference's
 * evaluation hands renderer (samples.view(-1, 6) of an H x W frame, TriPlane/main.py:88-94).  Pixels are bit-identical to ngf_field_render; the
 * launch walks the image in screen-space blocks (80 rows x 80 pixels) instead of row by row, which keeps the colour-plane taps of the tiles in flight
 * inside an XCD's L2 (dense scenes: -4 ... -6 % frame time, profiles/r06_r2_locality.txt).  row_width <= 0 or not a multiple of 8: ngf_field_render. */
int ngf_field_render_image(const ngf_field *f, const float *rays, int64_t n, int32_t row_width, int32_t n_samples, int32_t white_bg,
                           int32_t mode, const float *jitter, float *rgb, float *depth, uint64_t *stats, void *hip_stream);

/* Pieces of the path exposed for parity tests (same device code as ngf_field_render):
 *   ngf_field_decode_rgb : compute_rgb + rgb_decoder (Field.py:93-105, networks.py:25-32) for n samples
 *                          given their (gauge-shifted) plane coordinates [n,6] and view directions [n,3]
 *   ngf_field_march      : per-sample sigma and weight [n,S] (sample_ray .. raw2alpha) */
int ngf_field_decode_rgb(const ngf_field *f, const float *coords, const float *dirs, int64_t n, int32_t mode,
                         float *rgb, void *hip_stream);
int ngf_field_march(const ngf_field *f, const float *rays, int64_t n, int32_t n_samples, int32_t mode,
                    const float *jitter, float *sigma, float *weight, void *hip_stream);

/* Model-management helpers that reuse the march's device code:
 *   ngf_field_alpha      : Base.compute_alpha (FieldBase.py:140-159) for n world-space points [n,3]:
 *                          alpha = 1 - exp(-sigma * length), sigma = 0 where the alpha mask (if any) is empty;
 *                          mode as in ngf_field_render (the reference calls it with the gauge OFF, iteration = -1).
 *   ngf_field_ray_filter : the alpha-mask branch of Base.filtering_rays (FieldBase.py:237-239): keep[i] = 1 iff any of
 *                          the n_samples points of ray i samples the alpha mask > 0 (requires a mask). */
int ngf_field_alpha(const ngf_field *f, const float *xyz, int64_t n, int32_t mode, float length, float *alpha,
                    void *hip_stream);
/* Replaces: getDenseAlpha + updateAlphaMask (FieldBase.py:161-216) in one pass over the [gx,gy,gz] lattice of the aabb:
 * sx/sy/sz = the three torch.linspace(0,1,g) vectors (device), mode as ngf_field_alpha, length = stepSize, thres =
 * alphaMask_thres.  Outputs (device): alpha_zyx [gz,gy,gx] (dense alpha, transposed like FieldBase.py:185), volume_zyx
 * [gz,gy,gx] in {0,1} after clamp / 3x3x3 max-pool / threshold, new_aabb [2,3] = box of the occupied lattice points,
 * *count = number of occupied voxels (0 => new_aabb is meaningless; the reference raises there). */
int ngf_field_alpha_mask_build(const ngf_field *f, int32_t mode, const float *sx, const float *sy, const float *sz, int32_t gx,
                               int32_t gy, int32_t gz, float length, float thres, float *alpha_zyx, float *volume_zyx,
                               float *new_aabb, uint64_t *count, void *hip_stream);
/* filtering_rays (FieldBase.py:218-246): keep[i] = 1 iff ray i touches an occupied voxel of the alpha mask within n_samples
 * steps (n_samples > 0; the field must carry a mask), or -- n_samples <= 0 -- passes the bbox_only slab test t_max > t_min. */
int ngf_field_ray_filter(const ngf_field *f, const float *rays, int64_t n, int32_t n_samples, uint8_t *keep,
                         void *hip_stream);

/* Replaces: get_ray_directions + get_rays for a pin-hole camera (TriPlane/dataLoader/ray_utils.py:24-42,
 * 66-87; blender.py:46-53,84-85): rays [rows*W,6] for image rows [row0,row0+rows), c2w = HOST float[12]
 * (3x4 row-major, OpenCV axes), directions normalised as blender.py:52. */
int ngf_generate_rays(int32_t H, int32_t W, float focal, const float *c2w_host, int32_t row0, int32_t rows,
                      float *rays, void *hip_stream);

/* Replaces: get_rays_dir on the full-image ('no_crop') pixel grid of DtuDataset.__getitem__ (UV-Mapping/data/dtu.py:27-37,
 * 160-168): raydir [rows*W,3] for image rows [row0,row0+rows).  HOST pointers: focal[2], princpt[2] (in_camFocal /
 * in_camPrincpt of the view) and rot[9] = extrinsics[view][0:3,0:3] row-major (dtu.py:133; applied transposed and
 * normalised with the reference's + 1e-5, all float32 like the reference's numpy arithmetic). */
int ngf_generate_rays_dtu(int32_t H, int32_t W, const float *focal_host, const float *princpt_host, const float *rot_host,
                          int32_t row0, int32_t rows, float *raydir, void *hip_stream);

/* ---- eval output stage: what `evaluation` (TriPlane/main.py:73-138) does to every
 * rendered frame on the host, here on the device so that a frame leaves HBM as 8-bit images and scalars.
 * `workspace` is caller-owned device scratch of at least ngf_eval_workspace_bytes(H, W, filter_size) bytes
 * (pass H = W = 0 for the calls that only reduce).  All results stay on the device; nothing synchronises.
 *   ngf_eval_frame_u8       rgb_map.clamp(0,1) (main.py:98) then (rgb*255).astype('uint8') (main.py:117): truncation.
 *   ngf_eval_depth_range    range[0] = min(x[x>0]), range[1] = max(x) of nan_to_num(depth)   (utils.py:37-40)
 *   ngf_eval_depth_colormap visualize_depth_numpy (utils.py:32-47): x=(x-mi)/(ma-mi+1e-8) in float32, (255*x) cast to
 *                           uint8 as numpy does on x86-64 (int32 truncation, low byte), then the 256x3 colour table
 *                           `lut` (device, B,G,R per entry like cv2.applyColorMap) -> out [n,3].
 *   ngf_eval_mse            mean((a-b)**2) over n float32 values -> *out (float64; PSNR = -10 log10, main.py:105-106)
 *   ngf_eval_ssim           rgb_ssim (utils.py:109-155) of two [H,W,3] float32 images: 'valid' separable Gaussian
 *                           blur and the SSIM map in float64; *mean_out = mean of the map, map_out (nullable)
 *                           [H-fs+1, W-fs+1, 3]. */
int64_t ngf_eval_workspace_bytes(int32_t H, int32_t W, int32_t filter_size);
int ngf_eval_frame_u8(const float *rgb, int64_t n_values, uint8_t *out, void *hip_stream);
int ngf_eval_depth_range(const float *depth, int64_t n, float *range, void *workspace, void *hip_stream);
int ngf_eval_depth_colormap(const float *depth, int64_t n, const float *range, const uint8_t *lut, uint8_t *out,
                            void *hip_stream);
int ngf_eval_mse(const float *a, const float *b, int64_t n, double *out, void *workspace, void *hip_stream);
int ngf_eval_ssim(const float *img0, const float *img1, int32_t H, int32_t W, double max_val, int32_t filter_size,
                  double filter_sigma, double k1, double k2, double *mean_out, double *map_out, void *workspace,
                  void *hip_stream);

/* Replaces: np.packbits(alpha_volume.bool().reshape(-1)) (FieldBase.py:104-108) on the device: volume [n] float32 (non-zero =
 * occupied) -> bits [(n+7)/8], most significant bit first -- the mask_bits image ngf_field_create takes. */
int ngf_pack_mask_bits(const float *volume, int64_t n, uint8_t *bits, void *hip_stream);

/* Replaces: F.interpolate(plane, size=(Ho,Wo), mode='bilinear', align_corners=True) in TriPlane.up_sampling
 * (TriPlane/models/Field.py:108-114): src [C,Hi,Wi] -> dst [C,Ho,Wo], both device float32. */
int ngf_resize_bilinear(const float *src, int32_t C, int32_t Hi, int32_t Wi, float *dst, int32_t Ho, int32_t Wo,
                        void *hip_stream);

/* ---- one TriPlane training step: TriPlane/main.py:264-299 ----------------------------
 * Replaces, per iteration: field(rays_train, is_train=True, ...) (FieldBase.py:251-312), rgb MSE (main.py:281),
 * + L1_reg_weight * density_L1() (main.py:288-291, Field.py:149-152), total_loss.backward(), optimizer.step() of
 * torch.optim.Adam(get_optparam_groups(...), betas=(0.9,0.99)) (main.py:234-242, Field.py:34-46).
 * The parameters and the Adam moments stay caller-owned device tensors in the REFERENCE layouts and are updated IN
 * PLACE; the trainer owns packed copies, gradient buffers and the per-sample scratch (ngf_trainer_bytes()).
 * Parameter indices (`which`): 0-2 plane_xy/yz/xz [1,64,H,W], 3-5 gauge_xy/yz/xz [1,2,H,W], 6 density_decoder.weight
 * [1,48], 7 .bias [1], 8 rgb_decoder.basis.weight [144,144], 9/10 mlp.0 weight [64,159] / bias, 11/12 mlp.2, 13/14 mlp.4. */
#define NGF_TRAIN_PARAMS 15
typedef struct ngf_train_desc {
    float aabb[6];
    float near_, far_, step, distance_scale, weight_thres;
    float *plane[3];
    int32_t plane_h[3], plane_w[3];
    float *gauge[3];
    int32_t gauge_h[3], gauge_w[3];
    float *dens_w, *dens_b, *basis, *w1, *b1, *w2, *b2, *w3, *b3;
    float *exp_avg[NGF_TRAIN_PARAMS], *exp_avg_sq[NGF_TRAIN_PARAMS]; /* Adam state, parameter layouts, zero-initialised by the caller; ALL NULL = a trainer without optimiser (ngf_train_forward / ngf_train_backward_grad) */
    const uint8_t *mask_bits;      /* optional alpha mask as in ngf_field_desc; NULL = none */
    int32_t mask_d, mask_h, mask_w;
    float mask_aabb[6];
    int64_t max_rays;              /* largest batch (args.batch_size) */
    int32_t max_samples;           /* largest N_samples */
    int64_t chunk_samples;         /* active samples whose activation rows (2.4 KB each) are kept at once; 0 = the whole batch (up to 9 Mi samples: 9 GB for
                                      4096 rays x 884 samples); > 0 = that many, the step then reads the active count on the host (one sync) to cut the list into
                                      chunks; < 0 = SPECULATIVE: |chunk_samples| rows and never a host round trip -- a batch with more active samples than rows is
                                      flagged on the device, its ngf_train_adam* updates leave parameters and moments untouched, ngf_train_overflow_count reports it */
} ngf_train_desc;
typedef struct ngf_trainer ngf_trainer;
int ngf_trainer_create(const ngf_train_desc *desc, ngf_trainer **out, void *hip_stream);
int ngf_trainer_destroy(ngf_trainer *t);
int64_t ngf_trainer_bytes(const ngf_trainer *t);
int32_t ngf_sizeof_train_desc(void);
/* forward (training mode) + backward of  mean((rgb_map - rgb_train)^2): the gradients of all 15 parameters land in the
 * trainer's buffers (zeroed first).  jitter [n] = the per-ray U[0,1) of sample_ray (FieldBase.py:129-130; NULL = 0),
 * white_bg = `white_bg or coin` of FieldBase.py:299, gauge_on = (iteration >= gauge_start).  rgb_loss (DEVICE, loss_len doubles, loss_len >= 1)
 * receives the SUM of squared residuals in [0] and, when loss_len >= 2, the mean (sum / 3n = the reference's rgb loss, main.py:277) in [1]; *n_active_host
 * (HOST, nullable) the active-sample count.  ngf_train_backward (no loss_len) is the ABI-1 entry point and writes ONE double, the sum.
 * On an error return after the call has forked, the caller's stream has still been made to wait for the trainer's streams.
 * Asynchronous on the stream when n_active_host is NULL and one chunk holds every sample of the batch (chunk_samples = 0: the
 * default): the colour kernels then read the active count on the device.  Passing n_active_host, or a chunk smaller than
 * n * n_samples, costs one stream synchronisation per call.  ngf_train_get_active copies the count of the last backward of an
 * n-ray batch into a DEVICE int32 without synchronising.
 * Streams: the trainer owns two non-blocking HIP streams.  After the colour backward the weight-gradient GEMMs and the colour-plane
 * scatter run on them beside the density backward on hip_stream, forked and joined with events -- everything the call enqueues is
 * ordered after what hip_stream held before the call and before what is enqueued on it afterwards, exactly as if it had run on
 * hip_stream alone (ngf_debug_set("ablate", 1 << 19) makes it so, for A/B timing).  ngf_train_adam_all does the same for the planes.
 * A trainer must not be used from two streams at once.  (1.7 KB per sample in the line above: 2.3 KB + 0.1 KB of scatter pairs since
 * round 3 -- the feature-gradient rows are kept as well.) */
int ngf_train_backward2(ngf_trainer *t, const float *rays, const float *rgb_train, const float *jitter, int64_t n,
                        int32_t n_samples, int32_t white_bg, int32_t gauge_on, double *rgb_loss, int32_t loss_len,
                        int64_t *n_active_host, void *hip_stream);
int ngf_train_backward(ngf_trainer *t, const float *rays, const float *rgb_train, const float *jitter, int64_t n,
                       int32_t n_samples, int32_t white_bg, int32_t gauge_on, double *rgb_loss, int64_t *n_active_host,
                       void *hip_stream);
int ngf_train_get_active(ngf_trainer *t, int64_t n, int32_t *out_device, void *hip_stream);
/* The step in TWO calls, for a caller that owns the loss and the optimiser -- the reference's loop as written (TriPlane/main.py:272-296):
 *     output = field(rays_train, is_train=True, ...)                     -> ngf_train_forward      (FieldBase.py:251-312 with autograd recording)
 *     total_loss = f(output['rgb_map'], ...); total_loss.backward()       -> ngf_train_backward_grad (autograd of that forward)
 *     optimizer.step()                                                    -> the caller's optimiser on the gradients of ngf_train_get_grad
 * ngf_train_forward renders the batch in training mode exactly as ngf_train_backward2 does (same kernels, same per-sample buffers) and writes
 * rgb_map [n,3] (after the clamp of FieldBase.py:302) and depth_map [n] (the no_grad block of FieldBase.py:304-306); *ticket names this forward.
 * ngf_train_backward_grad takes d loss / d rgb_map [n,3] (DEVICE) and leaves d loss / d parameter for all 15 parameters in the trainer's
 * buffers (ngf_train_get_grad; planes WITHOUT any L1 term -- density_L1 is the caller's own graph here).  depth_map carries no gradient
 * (the reference computes it under torch.no_grad()).  Between the two calls `rays`, `jitter` and the parameter tensors must stay alive and
 * unchanged, and the trainer must not run another forward or fused step: its per-sample buffers hold THIS batch.  A ticket that is no longer
 * the trainer's last forward is refused (NGF_E_ARG): run ngf_train_forward again on the same inputs, then the backward (ngf_amd does).
 * A trainer used only this way may be created with every exp_avg / exp_avg_sq pointer NULL (ngf_train_adam* then refuse).
 * n_active_host / synchronisation: as ngf_train_backward2 (the forward call is the one that may wait for the host). */
int ngf_train_forward(ngf_trainer *t, const float *rays, const float *jitter, int64_t n, int32_t n_samples, int32_t white_bg, int32_t gauge_on,
                      float *rgb_map, float *depth_map, int64_t *n_active_host, int64_t *ticket, void *hip_stream);
int ngf_train_backward_grad(ngf_trainer *t, int64_t ticket, const float *d_rgb_map, void *hip_stream);
/* speculative rows (chunk_samples < 0): *count_host = steps since the trainer was made whose batch had more active samples than the trainer keeps
 * rows for (their updates were skipped), *rows_host (nullable) = the row count.  Synchronises hip_stream. */
int ngf_train_overflow_count(ngf_trainer *t, int64_t *count_host, int64_t *rows_host, void *hip_stream);
/* the gradient of parameter `which` in its reference layout -> out (device) */
int ngf_train_get_grad(ngf_trainer *t, int32_t which, float *out, void *hip_stream);
/* torch.optim.Adam's update of parameter `which` from the gradient held by the trainer; step_count >= 1 is that
 * parameter's own step number; l1_weight (planes only) adds d/dp [l1_weight * mean(|p|)] to the gradient. */
int ngf_train_adam(ngf_trainer *t, int32_t which, int32_t step_count, float lr, float beta1, float beta2, float eps,
                   float l1_weight, void *hip_stream);
/* optimizer.step() for all 15 parameters: step_count[k] >= 1 updates parameter k with lr[k], step_count[k] <= 0 skips it (a
 * parameter whose .grad is None).  The six plane updates are one launch each, the nine MLP parameters share one. */
int ngf_train_adam_all(ngf_trainer *t, const int32_t step_count[NGF_TRAIN_PARAMS], const float lr[NGF_TRAIN_PARAMS], float beta1,
                       float beta2, float eps, float l1_weight, void *hip_stream);
/* ABI 5 -- the two calls behind the reference's own loop on the drop-in field (TriPlane/main.py:234-242,294-302: torch's loss, total_loss.backward(),
 * optimizer.step() with ngf_amd.optim.Adam in torch.optim.Adam's place):
 * ngf_train_get_grads: every wanted gradient of the last ngf_train_backward_grad / ngf_train_backward in its reference layout, out[k] NULL = skip
 * (one call and at most seven launches instead of fifteen ngf_train_get_grad calls).
 * ngf_train_adam_ext: torch.optim.Adam's update of the trainer's parameters from the CALLER's gradient tensors and moments (reference layouts:
 * p.grad, state['exp_avg'], state['exp_avg_sq']); step_count[k] <= 0 or grad[k] NULL leaves parameter k alone; the plane kernels also write the
 * trainer's channel-last copies, so the next forward needs no re-pack.  No L1 term is added (the caller's loss carries it). */
int ngf_train_get_grads(ngf_trainer *t, float *const out[NGF_TRAIN_PARAMS], void *hip_stream);
int ngf_train_adam_ext(ngf_trainer *t, const float *const grad[NGF_TRAIN_PARAMS], float *const exp_avg[NGF_TRAIN_PARAMS],
                       float *const exp_avg_sq[NGF_TRAIN_PARAMS], const int32_t step_count[NGF_TRAIN_PARAMS], const float lr[NGF_TRAIN_PARAMS],
                       float beta1, float beta2, float eps, void *hip_stream);
/* ABI 5: TriPlane.density_L1 (TriPlane/models/Field.py:149-152: mean|plane_xy| + mean|plane_yz| + mean|plane_xz|) as two launches, and its gradient
 * sign(p) * upstream / n as one, for the reference's `total_loss += L1_reg_weight * field.density_L1()` (main.py:279-281).  planes[k]: contiguous float32,
 * 16-byte aligned, n[k] values; out / upstream: one float on the device; workspace: 3 * 256 doubles on the device; grads[k] NULL = not wanted. */
int ngf_planes_l1(const float *const planes[3], const int64_t n[3], float *out, void *workspace, void *hip_stream);
int ngf_planes_l1_backward(const float *const planes[3], const int64_t n[3], const float *upstream, float *const grads[3], void *hip_stream);
/* The trainer keeps channel-last copies of the planes and gauge planes; ngf_train_adam keeps them current.  After writing to a
 * plane's memory by any other means (checkpoint load, in-place edit) call this: the next backward re-packs all of them. */
int ngf_train_params_changed(ngf_trainer *t);
/* profiling aid: with ngf_debug_set("ablate", 1 << 20) the colour backward adds its per-section clock counts (summed over
 * waves) to 8 counters; with bit 1 << 21 the backward kernels count their atomic line transactions (counters 8, 9) and the
 * scatter calls that took the per-tap path (10).  This reads and clears them (out16: 16 x uint64, host).  Synchronises. */
int ngf_train_debug_sections(ngf_trainer *t, uint64_t *out16);

/* ---- UV-Mapping (NeuTex) colour path: UV-Mapping/model/model.py:27-59 ----------------------------------------
 * 29 nn.Linear layers in evaluation order, reference layouts (weight [out,in], bias [out], float32, device):
 *   [0..11]  net_geometry_decoder.block.{0,2,..,22}   63-256, 10x 256-256, 256-1          (decoder.py:201-237)
 *   [12..16] gauge_transform.encoder.{linear1,linear2,linear_list.0,linear_list.1,last_linear}
 *                                                      63-64, 64-128, 128-128, 128-128, 128-(3|2) (gauge_fields.py:8-46)
 *   [17..22] net_texture.block1.{0,2,..,10}           (63|42)-256, 5x 256-256             (decoder.py:19-25)
 *   [23]     net_texture.color1                        256-3
 *   [24..28] net_texture.block2.{0,2,4,6,8}            295-256, 3x 256-256, 256-3          (decoder.py:27-34)
 * sphere: primitive_type == 'sphere' (uv = normalize(q) in R^3) else 'square' (uv = tanh(q) in R^2). */
#define NGF_UV_LAYERS 29
enum { NGF_UV_F_SPLIT_BF16 = 1 };   /* ngf_uv_desc.flags: the eighteen 256 -> 256 layers and block2.0 (94 % of the MACs) as six bf16 MFMA
                                       products per fp32 product (3-term split operands, fp32 accumulate): fp32-level error; opt-in */
typedef struct ngf_uv_desc {
    int32_t sphere;
    int32_t flags;                  /* NGF_UV_F_* (occupies what was alignment padding: the layout of the other members is unchanged) */
    const float *w[NGF_UV_LAYERS];
    const float *b[NGF_UV_LAYERS];
} ngf_uv_desc;
typedef struct ngf_uv ngf_uv;
int ngf_uv_create(const ngf_uv_desc *desc, ngf_uv **out, void *hip_stream);
int ngf_uv_destroy(ngf_uv *m);
/* Texture editing: TextureMlpDecoder with `cubemap_` set (UV-Mapping/model/decoder.py:52-58,79-121; util.py:172-238,277-282).
 * tex = device float [6,R,R,C] cube map (sphere models) or [H,W,C] square (faces = 1), C = 3 or 4, mode = cubemap_mode_ 0..4;
 * copied into the model.  tex = NULL switches editing off.  ngf_uv_texture_edit applies the same stage to explicit
 * (uv [n,3], orig = color1 + color2 [n,3]) pairs -> out [n,3]. */
int ngf_uv_set_texture(ngf_uv *m, const float *tex, int32_t faces, int32_t H, int32_t W, int32_t C, int32_t mode,
                       void *hip_stream);
int ngf_uv_texture_edit(const ngf_uv *m, const float *uv, const float *orig, int64_t n, float *out, void *hip_stream);
/* Replaces NeuTex.forward's colour outputs for one camera (model.py:30-52):
 *   campos_host float[3], bg_host float[3] or NULL (HOST pointers), raydir [R,3], jitter_u [R,S] = the uniforms
 *   cube_ray_generation draws with torch.rand (renderer.py:112-117; jitter = 0.05 always, model.py:30);
 *   color [R,3] (tone-mapped), transmittance [R]; dbg_sigma [R,S] / dbg_col [R,S,3] optional (NULL);
 *   stats NULL or 2 x uint64: += {in-cube samples evaluated, MLP passes of 16 samples}. */
int ngf_uv_render(const ngf_uv *m, const float *campos_host, const float *raydir, const float *bg_host,
                  const float *jitter_u, int64_t n_rays, int32_t n_samples, float *color, float *transmittance,
                  float *dbg_sigma, float *dbg_col, uint64_t *stats, void *hip_stream);

/* The same for n_cams cameras in ONE launch, camera positions / backgrounds in HBM -- no host pointer, nothing to wait for
 * (NeuTex.forward's call pattern, UV-Mapping/test.py:108-114, is one call per 1024-ray chunk: a per-call D2H read of the camera would
 * synchronise the stream every chunk): campos_dev [n_cams,3], bg_dev [n_cams,3] or NULL, raydir [n_cams*rays_per_cam,3], jitter_u
 * [n_cams*rays_per_cam,S]; ray r is seen from camera r / rays_per_cam.  Outputs as ngf_uv_render, for all rays. */
int ngf_uv_render_batch(const ngf_uv *m, const float *campos_dev, const float *raydir, const float *bg_dev, const float *jitter_u,
                        int32_t n_cams, int64_t rays_per_cam, int32_t n_samples, float *color, float *transmittance,
                        float *dbg_sigma, float *dbg_col, uint64_t *stats, void *hip_stream);

/* Experiment / test knobs of the launch code (tile shapes, kernel variants, A/B ablations used by the bit-identity tests and the
 * scripts under profiles/).  Process-wide; value -1 restores the library default.  The library never reads the environment:
 * this explicit call is the only way to change what a launch does.  Names: "tile_w" (64/32/16/8/4/2/1: ONE tile width for the whole
 * launch instead of the tile plan), "tail" (16 x the narrow tiles per resident wave and width at the end of a launch's tile plan; 0 = wide
 * tiles only), "split" (0/1), "waves",
 * "nstep", "profile", "ablate" (bit mask, see ngf_device.hpp), "uv_tiles" (1/2), "kernel" (0 = fused march+shade waves, 1 =
 * specialised march / shade waves), "stage" (1 = LDS-staged density strips), "poison" (bit 0: before every kernel of the library
 * a launch fills the LDS of every CU with the quiet-NaN pattern 0x7FC0DEAD, so that a read of LDS the kernel did not write shows up
 * as NaN instead of depending on the previous kernel; bit 1: the allocations of a new handle are filled with the pattern before
 * they are packed), "grid" (upper bound of the workgroups of a render launch: with 1 every wave of the only workgroup takes many tiles
 * one after the other), "xcd" (1: one tile queue per XCD with stealing; default 0: a single queue -- measured neutral on the
 * SIMD-bound frames, profiles/r03_xcd_queues.txt).  The knobs are independent atomics: setting one while another thread launches is safe, but a launch sees
 * whatever values are current when it reads them -- they are test / experiment switches, not a per-call API. */
int ngf_debug_set(const char *name, int32_t value);
int32_t ngf_debug_get(const char *name);
/* the "poison" bit-0 launch on its own: fill the LDS of every CU with 0x7FC0DEAD on `hip_stream` (tests) */
int ngf_debug_dirty_lds(void *hip_stream);
/* out8[x] = number of workgroups of a `workgroups`-wide launch that ran on XCD x (HW_REG_XCC_ID): the render launches keep one tile
 * queue per XCD (knob "xcd": 1 / 0 forces it on / off) and rely on this id */
int ngf_debug_xcd_histogram(unsigned *out8, int32_t workgroups, void *hip_stream);
/* The tile plan a render launch of n rays would use (host arithmetic only, no GPU): widest tile `wide` (8 TriPlane, 16 InfoInv), `resident`
 * waves in the persistent grid (CUs x waves per workgroup), tail16 as knob "tail" (-1 = the library default).  The ray list is cut into up
 * to four contiguous segments of seg_rays[k] rays in tiles of 1 << seg_shift[k] rays, widest first; every segment but the last holds whole
 * tiles.  Returns the number of segments (1..4); unused entries are zeroed.  (Why: a persistent grid ends when its last wave does -- narrow
 * tiles for the last rays let the waves run dry together.) */
int ngf_debug_tile_plan(int64_t n, int32_t wide, int64_t resident, int32_t tail16, int64_t *seg_rays, int32_t *seg_shift);
/* test hook: the queue-position -> tile map of ngf_field_render_image evaluated on the host for `count` positions (RenderArgs::ord_*: ord_n positions
 * re-ordered, tpr tiles per image row, blocks of bw tiles x bh rows) */
int ngf_debug_tile_order(const uint32_t *q, int64_t count, uint32_t ord_n, uint32_t tpr, uint32_t bw, uint32_t bh, uint32_t *out);

const char *ngf_last_error(void);
int ngf_abi_version(void);
/* sizeof(ngf_field_desc) as compiled into the library (binding self-check) */
int ngf_sizeof_field_desc(void);
/* bytes of HBM the handle owns (packed textures + MLP image + the mask's three images + tile-queue heads) */
int64_t ngf_field_bytes(const ngf_field *f);
/* ngf_field_destroy parks a handle's device buffers in a per-process pool (exact-size reuse by the next ngf_field_create on the same device -- the
 * device the HANDLE was created on, not the calling thread's current one): a handle is rebuilt after every parameter change of an eval field, with
 * the same shapes, and hipFree / hipMalloc cost more than the rebuild's own work.  At most 64 buffers / ngf_pool_set_limit bytes (default 1 GiB) are
 * parked; when a new buffer does not fit, the OLDEST parked ones go back to the driver (sizes that never match again after up_sampling / shrink age out),
 * and a hipMalloc that fails inside ngf_field_create empties the pool and is tried once more.
 * ngf_pool_trim returns every parked buffer to the driver (e.g. before another framework needs the memory); ngf_pool_set_limit changes the byte cap
 * (0 = park nothing) and evicts down to it; ngf_pool_bytes reports what is parked on one device (device < 0: on all). */
int ngf_pool_trim(void);
int ngf_pool_set_limit(int64_t bytes);
int64_t ngf_pool_bytes(int32_t device);

#ifdef __cplusplus
}
#endif
#endif
