// ngf_uv.hpp -- UV-Mapping (NeuTex) colour path on gfx950: UV-Mapping/model/model.py:30-50.
//
//   cube_ray_generation  renderer.py:79-141     GeometryMlpDecoder  decoder.py:201-237
//   GaugeTransform       gauge_fields.py:8-74   TextureMlpDecoder   decoder.py:11-78 (un-edited texture branch)
//   ray_march            renderer.py:176-247    simple_tone_map     renderer.py:7-8
//
// 1.33 M MAC per sample, no data to speak of (5.4 MB of weights): the path is bound by the fp32 matrix pipe.
// One wave renders rays two at a time (NS = 2): lane i owns sample i of a 64-sample chunk of each (segment jitter, prefix sum,
// position, in-cube test).  The in-cube samples of both rays are compacted into one list and pushed through the three MLPs 32 at
// a time on
// v_mfma_f32_16x16x4_f32 with FOUR lanes per sample (lane (s, kq)); the product is evaluated transposed (rows =
// units, columns = samples) so a lane's accumulators are 64 of the 256 activations of its own sample and are the
// B operand of the next layer without any data movement.  Weights are pre-permuted at create time into
// [k-step][unit-tile group][lane][4] so the A operand of four MFMAs is one coalesced 16-byte load per lane,
// streamed from L2 (they do not fit LDS).  Out-of-cube samples are skipped: their opacity is exactly 0 in the
// reference (sigma * valid, renderer.py:222), so skipping them changes nothing.
#pragma once
#include "ngf_device.hpp"

namespace ngf {

#define NGF_UV_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
// Wait states between a layer's last matrix instructions and the LDS stores that take their data from the accumulator registers (inline assembly:
// hipcc's hazard recogniser does not see the pair).  16 + 4 = 20 states cover every matrix shape (gfx950: 8-pass 12, 16-pass 20).
// profiles/isa_hazards.py --hand counts them on the shipped assembly (tests/test_isa_lint.py); -DNGF_EXP_UV_SHORT_NOPS is that lint's negative control.
#ifdef NGF_EXP_UV_SHORT_NOPS
#define NGF_UV_MFMA_LDS_NOPS "s_nop 1"
#else
#define NGF_UV_MFMA_LDS_NOPS "s_nop 15\n\ts_nop 3"
#endif

// EXPERIMENT builds (make expuv; profiles/exp_uv_sections.py): -DNGF_EXP_UV_SECTIONS accumulates a wave's cycles per code section (s_memtime at the
// section boundaries, never inside the k loop) into UvArgs::stats[2 + i]; -DNGF_EXP_UV_SAMEW lets every 256 -> 256 layer read the weights of geometry
// layer 0 (wrong pixels, same instruction stream: what the launch costs when the weight set fits one XCD's L2).  Neither is in the product library.
#ifdef NGF_EXP_UV_SECTIONS
struct UvSec { unsigned long long t[8]; };
#define NGF_UVSEC_PARAM , UvSec &uvsec
#define NGF_UVSEC_ARG , uvsec
#define NGF_UVSEC_T(var) __builtin_amdgcn_sched_barrier(0); const unsigned long long var = __builtin_readcyclecounter(); __builtin_amdgcn_sched_barrier(0)
#define NGF_UVSEC_ADD(i, a, b) uvsec.t[i] += (b) - (a)
#else
#define NGF_UVSEC_PARAM
#define NGF_UVSEC_ARG
#define NGF_UVSEC_T(var)
#define NGF_UVSEC_ADD(i, a, b)
#endif

struct UvArgs {
    const float *raydir;   // [R,3]
    const float *U;        // [R,S] jitter uniforms
    float *color;          // [R,3]
    float *trans;          // [R]
    float *dbg_sigma;      // [R,S] or NULL
    float *dbg_col;        // [R,S,3] or NULL
    unsigned int *ray_counter;
    unsigned long long *stats;  // {in-cube samples, passes} or NULL
    int64_t R;
    int32_t S, sphere, has_bg, pad_;
    float campos[3], bg[3];     // one camera by value (ngf_uv_render: host pointers) ...
    const float *cam_dev;       // ... or [n_cams,3] camera positions in HBM (ngf_uv_render_batch: no host round trip); NULL = by value
    const float *bg_dev;        // [n_cams,3] background colours in HBM or NULL (then `bg` / has_bg decide)
    uint32_t rays_per_cam;      // ray r belongs to camera r / rays_per_cam
    uint32_t pad3_;
    // texture editing (decoder.py:79-121): cubemap_ [6,R,R,C] (sphere) or square [H,W,C], NULL = plain texture branch
    const float *tex;
    int32_t tex_h, tex_w, tex_c, tex_mode;
    int32_t split_bf16;    // NGF_UV_F_SPLIT_BF16: the 256 -> 256 layers and block2.0 run as 3-term split bf16 products (dense_bf16)
    int32_t pad2_;
    const float *w;        // packed weights / biases (offsets below, floats)
    // geometry
    int32_t geo_w0, geo_b0, geo_wh, geo_bh, geo_wo, geo_bo;       // wh/bh: 10 hidden layers, strides 65536 / 256
    // gauge
    int32_t ga_w0, ga_b0, ga_w1, ga_b1, ga_w2, ga_b2, ga_w3, ga_b3, ga_wo, ga_bo;
    // texture
    int32_t t1_w0, t1_b0, t1_wh, t1_bh, c1_w, c1_b, t2_w0, t2_b0, t2_wh, t2_bh, t2_wo, t2_bo;
    // split mode: bf16 images of the 256 -> 256 layers (stride kUvQLayer floats) and of block2.0 (10 k-blocks)
    int32_t geo_qh, t1_qh, t2_q0, t2_qh;
};

// Activations of the three MLPs.  ReLU (geometry, gauge: decoder.py:205-215, gauge_fields.py:20-33) as ONE integer maximum of the bit
// pattern with 0 (ngf_shade16.hpp relu1); LeakyReLU(0.2) (texture, decoder.py:20-45) as max(x, 0.2 x): x for x >= 0, the single product
// 0.2 x below -- the same bits as `x >= 0 ? x : 0.2 x`.  Round 2's generic `fmaxf(x,0) + slope * fminf(x,0)` cost five instructions per
// value (640 per 256-unit layer and pass, ~4 % of the launch) because IEEE rules keep the compiler from folding the zero slope.
template <int LEAKY, bool ONE_OP = false>
__device__ __forceinline__ float act_fn(float x)
{
    if constexpr (LEAKY && ONE_OP) {
        // max(x, 0.2 x) as ONE v_max_f32 (the one-wave-per-SIMD k loops, where every vector instruction is the matrix pipe's time): fmaxf() puts a
        // canonicalising `v_max_f32 x, x` in front (IEEE mode; x may be a signalling NaN for all hipcc knows), and so does fmed3(x, 0.2 x, inf), which
        // it folds back to fmaxf.  Not everywhere: the inline assembly's VGPR operands cost the two-waves-per-SIMD kernel (uv_tiles = 1) 100 B of scratch
        const float y = 0.2f * x;
        float r;
        asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y));
        return r;
    } else if constexpr (LEAKY) return fmaxf(x, 0.2f * x);
    else return __int_as_float(max(__float_as_int(x), 0));
}

// fp32 kernels: a layer's outputs go to LDS RAW, straight from the accumulator registers (store_act), and the activation is applied where the next
// layer READS its inputs.  ACT of a consumer = the activation of the layer that produced its inputs: kUvActNone (positional encodings, or a producer
// that applied it at the store: the split-bf16 kernel, kUvRd / kUvSt below), 0 ReLU, 1 LeakyReLU.
constexpr int kUvActNone = -1;
template <int ACT, bool ONE_OP = false>
__device__ __forceinline__ float act_in(float x)
{
    if constexpr (ACT < 0) return x;
    else return act_fn<ACT, ONE_OP>(x);
}

// ---- texture editing: TextureMlpDecoder.forward with cubemap_ set (decoder.py:79-121) -------------------------------------
// F.grid_sample(texture [H,W,C] as [1,C,H,W], (u,v), bilinear, padding_mode='border', align_corners=False) for one point
__device__ __forceinline__ void tex_sample_border(const float *tex, int H, int W, int C, float u, float v, float out[4])
{
    float ix = ((u + 1.0f) * (float)W - 1.0f) / 2.0f, iy = ((v + 1.0f) * (float)H - 1.0f) / 2.0f;
    ix = fminf((float)(W - 1), fmaxf(ix, 0.0f));              // clip_coordinates
    iy = fminf((float)(H - 1), fmaxf(iy, 0.0f));
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float tx = ix - fx, ty = iy - fy;
    const float w00 = (1.0f - tx) * (1.0f - ty), w10 = tx * (1.0f - ty), w01 = (1.0f - tx) * ty, w11 = tx * ty;
    const bool x1ok = x0 + 1 <= W - 1, y1ok = y0 + 1 <= H - 1;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        float s = 0.0f;
        if (c < C && ix == ix && iy == iy) {
            s = tex[((size_t)y0 * W + x0) * C + c] * w00;
            if (x1ok) s += tex[((size_t)y0 * W + x0 + 1) * C + c] * w10;
            if (y1ok) s += tex[((size_t)(y0 + 1) * W + x0) * C + c] * w01;
            if (x1ok && y1ok) s += tex[((size_t)(y0 + 1) * W + x0 + 1) * C + c] * w11;
        }
        out[c] = s;
    }
}

// sample_cubemap (util.py:172-238): the six face masks are applied in order, a later face overwrites an earlier one on ties;
// a point that belongs to no face (NaN) keeps zeros
__device__ __forceinline__ void cubemap_sample(const float *cube, int R, int C, const float xyz[3], float out[4])
{
    const float x = xyz[0], y = xyz[1], z = xyz[2];
    const float ax = fabsf(x), ay = fabsf(y), az = fabsf(z);
    const bool px = x > 0.0f, py = y > 0.0f, pz = z > 0.0f;
    const bool mx = (ax >= ay) && (ax >= az), my = (ay >= ax) && (ay >= az), mz = (az >= ax) && (az >= ay);
    int face = -1;
    if (px && mx) face = 0;
    if (!px && mx) face = 1;
    if (py && my) face = 2;
    if (!py && my) face = 3;
    if (pz && mz) face = 4;
    if (!pz && mz) face = 5;
    out[0] = out[1] = out[2] = out[3] = 0.0f;
    if (face < 0) return;
    float u, v;
    switch (face) {
    case 0: u = -z / ax; v = y / ax; break;
    case 1: u = z / ax; v = y / ax; break;
    case 2: u = x / ay; v = -z / ay; break;
    case 3: u = x / ay; v = z / ay; break;
    case 4: u = x / az; v = y / az; break;
    default: u = -x / az; v = y / az; break;
    }
    tex_sample_border(cube + (size_t)face * R * R * C, R, R, C, u, v, out);
}

// orig = color1 + color2 (before any clamp), uv = the gauge output; the five cubemap_mode_ branches of decoder.py:101-121
__device__ __forceinline__ void uv_texture_edit(const float *tex, int H, int W, int C, int mode, int sphere, const float uv[3], const float orig[3],
                                                float col[3])
{
    float cc[4];
    if (sphere) cubemap_sample(tex, H, C, uv, cc);
    else tex_sample_border(tex, H, W, C, uv[0], uv[1], cc);
    float o[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) o[k] = fminf(fmaxf(mode == 0 ? orig[k] * 8.0f : orig[k], 0.0f), 1.0f);
    if (mode == 0) {
        const float m = ((o[0] + o[1]) + o[2]) / 3.0f;
#pragma unroll
        for (int k = 0; k < 3; ++k) col[k] = cc[k] * m;
    } else if (mode == 1) {
#pragma unroll
        for (int k = 0; k < 3; ++k) col[k] = cc[0] < 0.99f ? o[k] * cc[k] : o[k];
    } else if (mode == 2) {
#pragma unroll
        for (int k = 0; k < 3; ++k) col[k] = cc[0] < 0.99f ? o[k] * (1.0f / cc[k]) : o[k];
    } else if (mode == 3) {
        const bool msk = ((cc[0] + cc[1]) + cc[2]) > 0.01f;
        const float m2 = 2.0f * (((o[0] + o[1]) + o[2]) / 3.0f);
#pragma unroll
        for (int k = 0; k < 3; ++k) col[k] = (msk ? m2 * cc[k] : o[k]) + cc[k];
    } else {
#pragma unroll
        for (int k = 0; k < 3; ++k) col[k] = fminf(fmaxf(cc[k], 0.0f), 1.0f);
    }
}

// elementwise form of the edit stage (ngf_uv_texture_edit: utility + parity hook)
__global__ void __launch_bounds__(256) uv_texture_edit_kernel(const float *tex, int H, int W, int C, int mode, int sphere, const float *uv, const float *orig,
                                                              int64_t n, float *out)
{
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float q[3] = {uv[i * 3], uv[i * 3 + 1], uv[i * 3 + 2]}, o[3] = {orig[i * 3], orig[i * 3 + 1], orig[i * 3 + 2]};
        float c[3];
        uv_texture_edit(tex, H, W, C, mode, sphere, q, o, c);
        out[i * 3] = c[0]; out[i * 3 + 1] = c[1]; out[i * 3 + 2] = c[2];
    }
}

constexpr int kUvQLayer = 8 * 4 * 12 * 64 * 4;      // floats of one 256 -> 256 layer's split-bf16 image: [8 k-blocks][4 groups][12 fragments][64 lanes][8 bf16]
constexpr int kUvActSteps = 80;                    // k-steps of per-wave activation storage (74 used by block2.0)
constexpr int kUvWaveLds = kUvActSteps * 64;       // floats per wave

// Activations live in a wave-private LDS array act[k-step][lane]: lane (s, kq) keeps there, for k-step t, the input
// it supplies as B operand (unit (t>>2)*16 + 4*kq + (t&3) of its own sample).  Every lane only ever reads back what
// it wrote itself, so there is no cross-lane hazard and no barrier; LDS is used because the k loop must be a real
// loop (a fully unrolled 256x256 layer makes hipcc hoist ~256 16-byte weight loads and spill thousands of VGPRs)
// and registers cannot be indexed by a loop variable.

// NS = sample tiles per wave pass (16 samples each).  The weights (5.4 MB) do not fit one XCD's 4 MB L2, so every 16-sample pass
// streams them from the Infinity Cache (PMC: 248 GB of fabric reads per 76 800-ray launch, L2 hit 81 %, MFMA 52 % busy).  With
// NS = 2 a wave renders TWO rays at once and every weight load feeds two MFMAs: half the traffic for the same matrix work.

// acc[mt] <- bias (accumulator order: unit mt*16 + 4*kq + r)
template <int NT, int NS>
__device__ __forceinline__ void load_bias(const float *b, int kq, f32x4 acc[NS][NT])
{
#pragma unroll
    for (int mt = 0; mt < NT; ++mt) {
        const f32x4 v = *reinterpret_cast<const f32x4 *>(b + kq * (NT * 4) + mt * 4);
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s][mt] = v;
    }
}

// A operand of one k-step: NT unit tiles from packed weights [t][NT/4][64 lanes][4] (one coalesced 16-byte load per 4 NS MFMAs)
template <int NT, int NS>
struct KStepA {
    f32x4 a[NT / 4];
    float b[NS];
};

template <int NT, int NS>
__device__ __forceinline__ void kload_w(const float *w, int t, int lane, KStepA<NT, NS> &k)
{
    const f32x4 *wp = reinterpret_cast<const f32x4 *>(w) + ((size_t)t * (NT / 4)) * 64 + lane;
#pragma unroll
    for (int g = 0; g < NT / 4; ++g) k.a[g] = wp[g * 64];
}
// ONE of a k-step's NT / 4 weight loads (tile group g), as a buffer load: the k-step's offset is an SGPR (soffset), the lane's a loop-invariant VGPR,
// the tile group's an immediate -- NO vector instruction per load, so it can sit between two MFMAs without a matrix -> vector -> matrix switch of
// the SIMD's datapath (a global load there needs a 64-bit vector add for its address: that form measured 3 % SLOWER than the burst, this one 2 % faster)
template <int NT, int NS>
__device__ __forceinline__ void kload_wg(const float *w, int t, int g, int lane, KStepA<NT, NS> &k)
{
#ifdef NGF_EXP_UV_GLOBAL_LOADS
    k.a[g] = (reinterpret_cast<const f32x4 *>(w) + ((size_t)t * (NT / 4)) * 64 + lane)[g * 64];
#else
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(w), 0, 0x7fffffff, 0x00020000);
    k.a[g] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, lane * 16 + g * 1024, t * ((NT / 4) * 1024), 0));
#endif
}
// B operands of k-step t0 + j; j: compile-time position in a group of k-steps that share t0 (one LDS address per group, j in the offset field)
template <int NT, int NS>
__device__ __forceinline__ void kload_b(const float *act, int t0, int lane, KStepA<NT, NS> &k, int j = 0)
{
    const float *ap = act + t0 * 64 + lane;
#pragma unroll
    for (int s = 0; s < NS; ++s) k.b[s] = ap[s * kUvWaveLds + j * 64];
}
template <int NT, int NS>
__device__ __forceinline__ void kload(const float *w, const float *act, int t0, int lane, KStepA<NT, NS> &k, int j = 0)
{
    kload_w<NT, NS>(w, t0 + j, lane, k);
    kload_b<NT, NS>(act, t0, lane, k, j);
}

// the weights of a layer's k-steps 0..3, requested ahead of the layer (one wave per SIMD: in front of the positional encodings or the activation store
// that produce its inputs, instead of in front of an idle matrix pipe once they are stored)
template <int NT, int NS>
__device__ __forceinline__ void uv_wprefetch(const float *w, int lane, KStepA<NT, NS> pre[4])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) kload_w<NT, NS>(w, j, lane, pre[j]);
}

// the activation of the producing layer on a k-step's B operands (in place, once the LDS read has landed); k-steps from t_none on are inputs that
// no layer produced (block2.0's view encodings behind the 64 k-steps of block 1's output)
template <int ACT, bool ONE_OP = false, int NT, int NS>
__device__ __forceinline__ void kact(KStepA<NT, NS> &k, int t, int t_none)
{
    if constexpr (ACT >= 0) {
#pragma unroll
        for (int s = 0; s < NS; ++s) k.b[s] = t < t_none ? act_in<ACT, ONE_OP>(k.b[s]) : k.b[s];
    }
}

template <int NT, int NS>
__device__ __forceinline__ void kmma(const KStepA<NT, NS> &k, f32x4 acc[NS][NT])
{
#pragma unroll
    for (int g = 0; g < NT / 4; ++g)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[s][4 * g + e] = NGF_UV_MFMA(k.a[g][e], k.b[s], acc[s][4 * g + e]);
}

// The one-wave-per-SIMD k loop: consume k-step k (B operands activated) and request k-step t's weights into kn, ONE load behind every 4 x NS MFMAs
// (a tile group).  Sixteen loads in a burst in front of 128 MFMAs left the matrix pipe idle while they issued (~280 cycles per burst: 34.5 cycles per
// MFMA against 32.8 for a pure MFMA loop).  NGF_EXP_UV_BURST: the burst form, for the A/B.
template <int NT, int NS>
__device__ __forceinline__ void kstep(const KStepA<NT, NS> &k, f32x4 acc[NS][NT], const float *w, int t, int lane, KStepA<NT, NS> &kn)
{
#ifdef NGF_EXP_UV_BURST
    kmma<NT, NS>(k, acc);
#else
#pragma unroll
    for (int g = 0; g < NT / 4; ++g) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[s][4 * g + e] = NGF_UV_MFMA(k.a[g][e], k.b[s], acc[s][4 * g + e]);
        __builtin_amdgcn_sched_barrier(0);
        kload_wg<NT, NS>(w, t, g, lane, kn);
        __builtin_amdgcn_sched_barrier(0);
    }
#endif
}
// a group of four k-steps: B operands of the group requested (kb*: k-steps tn ..), the group consumed (ka*: k-steps t ..) activated, then the MFMAs with
// the requested group's weight loads between them
template <int ACT, int NT, int NS>
__device__ __forceinline__ void kgroup(KStepA<NT, NS> &ka0, KStepA<NT, NS> &ka1, KStepA<NT, NS> &ka2, KStepA<NT, NS> &ka3, f32x4 acc[NS][NT], const float *w,
                                       const float *act, int t, int t_none, int tn, int lane, KStepA<NT, NS> &kb0, KStepA<NT, NS> &kb1, KStepA<NT, NS> &kb2,
                                       KStepA<NT, NS> &kb3)
{
#ifdef NGF_EXP_UV_BURST
    kload_w<NT, NS>(w, tn, lane, kb0); kload_w<NT, NS>(w, tn + 1, lane, kb1); kload_w<NT, NS>(w, tn + 2, lane, kb2); kload_w<NT, NS>(w, tn + 3, lane, kb3);
#endif
    kload_b<NT, NS>(act, tn, lane, kb0); kload_b<NT, NS>(act, tn, lane, kb1, 1); kload_b<NT, NS>(act, tn, lane, kb2, 2); kload_b<NT, NS>(act, tn, lane, kb3, 3);
    // (the activation of the four k-steps about to be consumed: one cluster of vector instructions per 128 matrix instructions -- at one wave per
    // SIMD every switch between the pipes costs the wave ~38 cycles)
    kact<ACT, true>(ka0, t, t_none); kact<ACT, true>(ka1, t + 1, t_none); kact<ACT, true>(ka2, t + 2, t_none); kact<ACT, true>(ka3, t + 3, t_none);
    __builtin_amdgcn_sched_barrier(0);
    kstep<NT, NS>(ka0, acc, w, tn, lane, kb0); kstep<NT, NS>(ka1, acc, w, tn + 1, lane, kb1);
    kstep<NT, NS>(ka2, acc, w, tn + 2, lane, kb2); kstep<NT, NS>(ka3, acc, w, tn + 3, lane, kb3);
    __builtin_amdgcn_sched_barrier(0);
}

// Dense layer, KT4 k-steps (multiple of 4; padded steps have zero weights and zero inputs).  Software pipeline with
// two k-steps of weight loads in flight behind the MFMAs (the un-pipelined loop left the waves 67 % of their cycles
// in s_waitcnt with the matrix pipe 31 % busy: latency-, not bandwidth-bound).
// DEEP: the four-k-steps-ahead pipeline of the one-wave-per-SIMD kernel also for a single tile (its last pass of a ray pair, below); without it
// NS = 1 is the two-waves-per-SIMD kernel's shallower pipeline.
template <int NT_OUT, int NS, bool DEEP = (NS > 1), int ACT = kUvActNone>
__device__ __forceinline__ void dense(const float *w, const float *bias, int KT4, int lane, const float *act, f32x4 out[NS][NT_OUT] NGF_UVSEC_PARAM, int t_none = 1 << 30,
                                      const KStepA<NT_OUT, NS> *pre = nullptr, bool use_pre = false)        // pre: the weights of k-steps 0..3, requested by the caller (uv_wprefetch) before the layer's inputs were stored
{
    NGF_UVSEC_T(ts0);
    load_bias<NT_OUT, NS>(bias, lane >> 4, out);
    if constexpr (!DEEP) {
        KStepA<NT_OUT, NS> k0, k1, k2, k3;
        kload<NT_OUT, NS>(w, act, 0, lane, k0);
        kload<NT_OUT, NS>(w, act, 1, lane, k1);
#pragma unroll 1
        for (int t = 0; t < KT4; t += 4) {
            // sched_barrier(0) pins the order: without it hipcc sinks every load next to its first use and waits
            // vmcnt(0) before each group of four MFMAs
            kload<NT_OUT, NS>(w, act, t + 2, lane, k2);
            kload<NT_OUT, NS>(w, act, t + 3, lane, k3);
            kact<ACT>(k0, t, t_none); kact<ACT>(k1, t + 1, t_none);
            __builtin_amdgcn_sched_barrier(0);
            kmma<NT_OUT, NS>(k0, out);
            kmma<NT_OUT, NS>(k1, out);
            __builtin_amdgcn_sched_barrier(0);
            const int tn = t + 4 < KT4 ? t + 4 : t;          // last iteration: harmless reload instead of a branch
            kload<NT_OUT, NS>(w, act, tn, lane, k0);
            kload<NT_OUT, NS>(w, act, tn + 1, lane, k1);
            kact<ACT>(k2, t + 2, t_none); kact<ACT>(k3, t + 3, t_none);
            __builtin_amdgcn_sched_barrier(0);
            kmma<NT_OUT, NS>(k2, out);
            kmma<NT_OUT, NS>(k3, out);
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        // one wave per SIMD: nobody else covers the L2 / Infinity-Cache latency, so FOUR k-steps of weights are in flight
        // behind four k-steps of MFMAs (4 x 32 x NS MFMAs = 4096 matrix-pipe cycles of cover at NS = 2)
        KStepA<NT_OUT, NS> a0, a1, a2, a3, b0, b1, b2, b3;
        if (use_pre) {
            a0 = pre[0]; a1 = pre[1]; a2 = pre[2]; a3 = pre[3];
            kload_b<NT_OUT, NS>(act, 0, lane, a0); kload_b<NT_OUT, NS>(act, 0, lane, a1, 1); kload_b<NT_OUT, NS>(act, 0, lane, a2, 2); kload_b<NT_OUT, NS>(act, 0, lane, a3, 3);
        } else {
            kload<NT_OUT, NS>(w, act, 0, lane, a0); kload<NT_OUT, NS>(w, act, 0, lane, a1, 1);
            kload<NT_OUT, NS>(w, act, 0, lane, a2, 2); kload<NT_OUT, NS>(w, act, 0, lane, a3, 3);
        }
#ifdef NGF_EXP_UV_SECTIONS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // section 0 = bias + the first four k-steps of weights ARRIVED (the layer's uncovered latency)
#endif
        NGF_UVSEC_T(ts1);
        NGF_UVSEC_ADD(0, ts0, ts1);
#pragma unroll 1
        for (int t = 0; t < KT4; t += 8) {
            const bool has_b = t + 4 < KT4;
            const int tb = has_b ? t + 4 : t;                    // harmless reload when the layer ends on the first half
            kgroup<ACT, NT_OUT, NS>(a0, a1, a2, a3, out, w, act, t, t_none, tb, lane, b0, b1, b2, b3);
            if (has_b) {
                const int ta = t + 8 < KT4 ? t + 8 : t;
                kgroup<ACT, NT_OUT, NS>(b0, b1, b2, b3, out, w, act, t + 4, t_none, ta, lane, a0, a1, a2, a3);
            }
        }
        NGF_UVSEC_T(ts2);
        NGF_UVSEC_ADD(1, ts1, ts2);          // section 1 = the k loop (KT4 / 4 x 128 MFMAs at NT_OUT = 16)
#ifdef NGF_EXP_UV_SECTIONS
        uvsec.t[6] += (unsigned long long)(KT4 * (NT_OUT * NS));     // MFMAs issued by the loops of section 1
#endif
    }
}


// ---- NGF_UV_F_SPLIT_BF16: a 256-unit dense layer on v_mfma_f32_16x16x32_bf16 with 3-term split operands -----------------------------
// The technique of ngf_shade_bf16.hpp: every fp32 product becomes the six bf16 products of order <= 2 of (hi, mid, lo) splits, fp32
// accumulate, fp32-level error.  A k-block is 8 of the lane's inputs (32 of the layer's): the B fragments are split from the fp32
// activations in LDS once per k-block and serve all 16 unit tiles; the A fragments (weights, split on the host) stream from L2 as
// [k-block][4 tile groups][3 parts][4 tiles][lane][8 bf16], in tile pairs, four pairs (96 registers) in flight.  A layer costs 768 x NS bf16 MFMAs (~17 cycles each) instead of 1024 x NS fp32 MFMAs (~33).
typedef short uv_bf16x8 __attribute__((ext_vector_type(8)));
struct UvSplit8 { uv_bf16x8 h, m, l; };
__device__ __forceinline__ UvSplit8 uv_split8(const float x[8])
{
    UvSplit8 s;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 hb = (__bf16)x[e];
        const float r1 = x[e] - (float)hb;
        const __bf16 mb = (__bf16)r1;
        const float r2 = r1 - (float)mb;
        const __bf16 lb = (__bf16)r2;
        s.h[e] = __builtin_bit_cast(short, hb);
        s.m[e] = __builtin_bit_cast(short, mb);
        s.l[e] = __builtin_bit_cast(short, lb);
    }
    return s;
}
struct UvAPair { uv_bf16x8 a[3][2]; };            // [part hi / mid / lo][tile of the pair]: 24 registers
// pair g2 (0..7) of k-block kb: tiles 2 g2, 2 g2 + 1
__device__ __forceinline__ void uv_gload(const float *wq, int kb, int g2, int lane, UvAPair &G)
{
    // (global loads: as buffer loads -- kload_wg -- these measured 2.8 % SLOWER; a buffer load issues more slowly, and the bf16 loop has no
    // matrix -> vector -> matrix switch to save: its address arithmetic hides among the split's vector instructions)
    const uv_bf16x8 *p = reinterpret_cast<const uv_bf16x8 *>(wq) + ((size_t)(kb * 4 + (g2 >> 1)) * 12 + (g2 & 1) * 2) * 64 + lane;
#pragma unroll
    for (int part = 0; part < 3; ++part)
#pragma unroll
        for (int e = 0; e < 2; ++e) {
#ifdef NGF_EXP_UV_SPLIT_NOLO            // timing experiment (wrong pixels): a third of the weight bytes not loaded
            if (part == 2) { G.a[2][e] = G.a[1][e]; continue; }
#endif
            G.a[part][e] = p[(part * 4 + e) * 64];
        }
}
#define NGF_UV_MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16((a), (b), (c), 0, 0, 0)
// the six products for the two tiles of a pair, product-major so that consecutive MFMAs hit different accumulators
template <int NS>
__device__ __forceinline__ void uv_gmma(const UvAPair &G, const UvSplit8 x[NS], f32x4 out[NS][16], int base)
{
#pragma unroll
    for (int pr = 0; pr < 6; ++pr) {
        // smallest terms first: (A lo, B hi), (A hi, B lo), (A mid, B mid), (A mid, B hi), (A hi, B mid), (A hi, B hi)
        const int pa = pr == 0 ? 2 : (pr == 2 || pr == 3 ? 1 : 0);
        const int pb = pr == 1 ? 2 : (pr == 2 || pr == 4 ? 1 : 0);
#pragma unroll
        for (int e = 0; e < 2; ++e)
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const uv_bf16x8 b = pb == 0 ? x[s].h : (pb == 1 ? x[s].m : x[s].l);
                out[s][base + e] = NGF_UV_MFMA_BF16(G.a[pa][e], b, out[s][base + e]);
            }
    }
}

// KB k-blocks (8 inputs of every lane each; padded inputs have zero weights); act as in dense(): act[s][t][lane], t = the lane's input
// index.  Four tile pairs (96 registers) are in flight: the pair consumed now was requested three pairs (72 x NS MFMAs) earlier.
template <int NS, int ACT = kUvActNone>
__device__ __forceinline__ void dense_bf16(const float *wq, const float *bias, int KB, int lane, const float *act, f32x4 out[NS][16], int t_none = 1 << 30)
{
    load_bias<16, NS>(bias, lane >> 4, out);
    UvAPair A0, A1, A2, A3;
    uv_gload(wq, 0, 0, lane, A0); uv_gload(wq, 0, 1, lane, A1); uv_gload(wq, 0, 2, lane, A2); uv_gload(wq, 0, 3, lane, A3);
    // the lane's 8 raw inputs of a k-block are read from LDS one k-block ahead (behind the 96 x NS MFMAs of the previous one): read at the top of the
    // k-block they are split in, each tile's LDS latency stood in front of an idle matrix pipe
    float raw[NS][8];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) raw[s][e] = act[s * kUvWaveLds + e * 64 + lane];
#pragma unroll 1
    for (int kb = 0; kb < KB; ++kb) {
        UvSplit8 x[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            float v[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (ACT >= 0 && kb * 8 < t_none) ? act_in<ACT>(raw[s][e]) : raw[s][e];          // a k-block lies wholly before or behind t_none (a multiple of 8)
            x[s] = uv_split8(v);
        }
        const int kn = kb + 1 < KB ? kb + 1 : kb;                // last k-block: harmless reload instead of a branch
        __builtin_amdgcn_sched_barrier(0);
        {
            const float *ap = act + kn * 8 * 64 + lane;
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int e = 0; e < 8; ++e) raw[s][e] = ap[s * kUvWaveLds + e * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
        uv_gmma<NS>(A0, x, out, 0);   __builtin_amdgcn_sched_barrier(0);  uv_gload(wq, kb, 4, lane, A0);  __builtin_amdgcn_sched_barrier(0);
        uv_gmma<NS>(A1, x, out, 2);   __builtin_amdgcn_sched_barrier(0);  uv_gload(wq, kb, 5, lane, A1);  __builtin_amdgcn_sched_barrier(0);
        uv_gmma<NS>(A2, x, out, 4);   __builtin_amdgcn_sched_barrier(0);  uv_gload(wq, kb, 6, lane, A2);  __builtin_amdgcn_sched_barrier(0);
        uv_gmma<NS>(A3, x, out, 6);   __builtin_amdgcn_sched_barrier(0);  uv_gload(wq, kb, 7, lane, A3);  __builtin_amdgcn_sched_barrier(0);
        uv_gmma<NS>(A0, x, out, 8);   __builtin_amdgcn_sched_barrier(0);  uv_gload(wq, kn, 0, lane, A0);  __builtin_amdgcn_sched_barrier(0);
        uv_gmma<NS>(A1, x, out, 10);  __builtin_amdgcn_sched_barrier(0);  uv_gload(wq, kn, 1, lane, A1);  __builtin_amdgcn_sched_barrier(0);
        uv_gmma<NS>(A2, x, out, 12);  __builtin_amdgcn_sched_barrier(0);  uv_gload(wq, kn, 2, lane, A2);  __builtin_amdgcn_sched_barrier(0);
        uv_gmma<NS>(A3, x, out, 14);  __builtin_amdgcn_sched_barrier(0);  uv_gload(wq, kn, 3, lane, A3);  __builtin_amdgcn_sched_barrier(0);
    }
}

// a 256-unit layer on KT4 fp32 k-steps or (split mode) KT4 / 8 bf16 k-blocks; wq: the layer's packed bf16 weights (split mode)
template <int NS, bool SPLIT, bool DEEP = (NS > 1), int ACT = kUvActNone>
__device__ __forceinline__ void dense256(const UvArgs &A, const float *w, const float *wq, const float *bias, int KT4, int lane, const float *act,
                                         f32x4 out[NS][16] NGF_UVSEC_PARAM, int t_none = 1 << 30, const KStepA<16, NS> *pre = nullptr, bool use_pre = false)
{
    if constexpr (SPLIT) dense_bf16<NS, ACT>(wq, bias, (KT4 + 7) / 8, lane, act, out, t_none);
    else dense<16, NS, DEEP, ACT>(w, bias, KT4, lane, act, out NGF_UVSEC_ARG, t_none, pre, use_pre);
}

// A layer's NT x 16 outputs of every tile -> the wave's activation rows, RAW (the consumer applies the activation: act_in), straight from the
// accumulator registers: `ds_write2st64_b32 addr, a[..], a[..]` with AGPR data operands.  In C++ (`act[...] = acc[s][mt][r]`) hipcc copies every
// accumulator to a VGPR first (v_accvgpr_read_b32), and with the activation on this side the store was 128 x (copy + activation + half a write) per
// lane: 3.2 % of a wave's life (profiles/r04_uv_sections.txt).  Row (mt * 4 + r) of tile s is 64-dword block s * kUvActSteps + mt * 4 + r behind the
// lane's own slot.  The s_nop are the matrix-write -> LDS-read wait states of the layer's last MFMAs (the hazard recogniser does not look into
// inline assembly); accumulators that live in VGPRs (the small gauge layers) are copied to AGPRs by the constraint -- no worse than before.
template <int O>       // two consecutive rows (64-dword blocks O, O + 1 behind addr) from two accumulator registers
__device__ __forceinline__ void ds_write2_rows_agpr(unsigned addr, float a0, float a1)
{
    asm volatile("ds_write2st64_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(addr), "a"(a0), "a"(a1), "n"(O), "n"(O + 1) : "memory");
}
template <int NT, int NS, int I = 0>       // I enumerates (tile s, unit tile mt, register pair): the offsets are immediates, hence the recursion
__device__ __forceinline__ void store_act_rows(unsigned addr, const f32x4 acc[NS][NT])
{
    if constexpr (I < NS * NT * 2) {
        constexpr int s = I / (NT * 2), mt = (I / 2) % NT, r = (I % 2) * 2;
        ds_write2_rows_agpr<s * kUvActSteps + mt * 4 + r>(addr, acc[s][mt][r], acc[s][mt][r + 1]);
        store_act_rows<NT, NS, I + 1>(addr, acc);
    }
}
// ACT = kUvActNone: raw rows for a consumer that applies the activation itself (the fp32 kernels); else the activation here (the split-bf16 kernel,
// whose k loop is not the matrix pipe's but the vector pipe's: measured 0.6 % slower with the activation on its read side)
template <int NT, int NS, int ACT = kUvActNone>
__device__ __forceinline__ void store_act(float *act, int lane, const f32x4 acc[NS][NT])
{
    if constexpr (ACT < 0) {
        static_assert(NS * kUvActSteps <= 256, "ds_write2st64 offsets are 8 bits");
        const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) float *)(act + lane);
        asm volatile(NGF_UV_MFMA_LDS_NOPS ::: "memory");
        store_act_rows<NT, NS>(addr, acc);
    } else {
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int mt = 0; mt < NT; ++mt)
#pragma unroll
                for (int r = 0; r < 4; ++r) act[s * kUvWaveLds + (mt * 4 + r) * 64 + lane] = act_fn<ACT>(acc[s][mt][r]);
    }
}
// where a layer's activation ACT is applied: by its consumer (fp32 kernels) or at its store (split-bf16 kernel)
#ifdef NGF_EXP_UV_SPLIT_ACT_ON_READ      // EXPERIMENT: the split-bf16 kernel with raw stores and the activation on its read side too
template <bool SPLIT, int ACT> constexpr int kUvRd = ACT;
template <bool SPLIT, int ACT> constexpr int kUvSt = kUvActNone;
#else
template <bool SPLIT, int ACT> constexpr int kUvRd = SPLIT ? kUvActNone : ACT;
template <bool SPLIT, int ACT> constexpr int kUvSt = SPLIT ? ACT : kUvActNone;
#endif

// store_act (raw rows) with the NEXT layer's bias loaded into each accumulator tile as soon as its rows are on their way: by the time the last tile is
// stored the first biases have arrived, instead of a full L2 round trip between the store and the next layer's first MFMA
template <int NT, int NS, int MT = 0>
__device__ __forceinline__ void store_act_rows_next_bias(unsigned addr, f32x4 acc[NS][NT], const float *bias_lane)
{
    if constexpr (MT < NT) {
        if constexpr (NS > 0) { ds_write2_rows_agpr<MT * 4>(addr, acc[0][MT][0], acc[0][MT][1]); ds_write2_rows_agpr<MT * 4 + 2>(addr, acc[0][MT][2], acc[0][MT][3]); }
        if constexpr (NS > 1) { ds_write2_rows_agpr<kUvActSteps + MT * 4>(addr, acc[1][MT][0], acc[1][MT][1]); ds_write2_rows_agpr<kUvActSteps + MT * 4 + 2>(addr, acc[1][MT][2], acc[1][MT][3]); }
        static_assert(NS <= 2, "tiles per pass");
        const f32x4 v = *reinterpret_cast<const f32x4 *>(bias_lane + MT * 4);
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s][MT] = v;
        store_act_rows_next_bias<NT, NS, MT + 1>(addr, acc, bias_lane);
    }
}

// Output layer with <= 3 units on the matrix pipe (rows >= n_out of the 16-row tile are zero; rows 0..3 land in the lanes of quarter kq = 0).
// Packed [KT / 4][64 lanes][4]: one 16-byte load per lane holds its A operands of four k-steps.  Round 4: all of a layer's weights are requested
// in one go (out_prefetch, right behind the previous layer's activation store) and the KT MFMAs of a tile run as two independent chains over
// fully unrolled k-steps.  Rounds 1-3 ran `for t: load w[t]; MFMA` with one dependent chain per tile: a latency-bound loop of ~150 cycles
// per MFMA, 4.6 % of a wave's life for 1 % of its matrix work (profiles/r04_uv_sections.txt).
template <int KT>
struct UvOutW { f32x4 w[KT / 4]; };

template <int KT>
__device__ __forceinline__ void out_prefetch(const float *w, int lane, UvOutW<KT> &o)
{
    const f32x4 *wp = reinterpret_cast<const f32x4 *>(w) + lane;
#pragma unroll
    for (int q = 0; q < KT / 4; ++q) o.w[q] = wp[q * 64];
}

template <int NS, int KT, int ACT, bool ONE_OP = false>
__device__ __forceinline__ void dense_out(const UvOutW<KT> &o, const float *bias4, int lane, const float *act, f32x4 r[NS])
{
    f32x4 acc[NS][2];
#pragma unroll
    for (int s = 0; s < NS; ++s) acc[s][0] = acc[s][1] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    // the B operands of a group of G k-steps are read one group ahead of their MFMAs (sched_barrier pins it: left alone hipcc hoists all 2 KT LDS reads
    // and spills) and activated in one cluster of vector instructions per group.  (G = 16 -- four times fewer matrix <-> vector switches -- measured
    // 0.2 % slower: the first group's 32 LDS reads stand in front of the layer)
    constexpr int G = 4;
    static_assert(KT % G == 0, "k-steps per group");
    float bq[2][NS][G];
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int e = 0; e < G; ++e) bq[0][s][e] = act[s * kUvWaveLds + e * 64 + lane];
#pragma unroll
    for (int q = 0; q < KT / G; ++q) {
        if (q + 1 < KT / G) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int e = 0; e < G; ++e) bq[(q + 1) & 1][s][e] = act[s * kUvWaveLds + (G * (q + 1) + e) * 64 + lane];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < G; ++e)
#pragma unroll
            for (int s = 0; s < NS; ++s) bq[q & 1][s][e] = act_in<ACT, ONE_OP>(bq[q & 1][s][e]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int e = 0; e < G; ++e)
#pragma unroll
            for (int s = 0; s < NS; ++s) acc[s][e & 1] = NGF_UV_MFMA(o.w[q * (G / 4) + e / 4][e & 3], bq[q & 1][s][e], acc[s][e & 1]);
        __builtin_amdgcn_sched_barrier(0);
    }
    const int src = lane & 15;
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int e = 0; e < 4; ++e) r[s][e] = __shfl(acc[s][0][e] + acc[s][1][e], src) + bias4[e];
}

// Program-order point for the wave's LDS traffic: an IR-level fence (no instruction at wavefront scope), a compiler memory barrier and a
// scheduling barrier for the machine scheduler.  The hardware needs nothing: a wave's LDS instructions execute in order.
__device__ __forceinline__ void uv_lds_order()
{
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    asm volatile("" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// Positional-encoding inputs [x(D), sin(x_d 2^f) (d-major: D*F values), cos(same)] (util.py:427-438) as B operands: entry f lives in row
// t0 + (f >> 2) of the activation array, lane quarter f & 3 of its sample; entries past D + 2 D F up to the layer's 4 KT inputs are zero.
// ONE sincos per (dimension, frequency) pair: lane (s, kq) takes the pairs g = kq, kq + 4, ... of its sample and writes the sine to entry D + g and
// the cosine to entry D + D F + g -- slots of OTHER lane quarters of the same sample, read back after the wave-level fence below (the LDS queue of
// a wave is in order).  Rounds 1-3: every lane evaluated a sincos for each of its own 16 entries and kept one of the two results (2.75 % of a wave's
// life, profiles/r04_uv_sections.txt).  t0b >= 0: a second copy at rows t0b.. (the geometry and the gauge network both start from PE10(p): the
// gauge's first layer reads the copy in rows 64..79, which the geometry network never touches).
// (the three coordinates by value: as an array in memory the compiler turns the selects below into indexed loads and the array into
// scratch -- 128 bytes per lane in rounds 1 and 2)
template <int D, int F, int NS>
__device__ __forceinline__ void store_pe(float *act, int t0, int KT, int lane, const float x[NS][3], int t0b = -1)
{
    constexpr int N = D * F;
    static_assert((2 * N) % 4 == 0, "the raw / padding entries are dealt out four at a time");
    const int kq = lane >> 4;
    // The writes below land in slots that OTHER lanes read (before: the previous consumer of these rows; after: the next layer).  In one thread's
    // view they alias none of its own reads, so these barriers are what keeps a wave-wide ds_write on its side of a wave-wide ds_read.
    uv_lds_order();
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const float x0 = x[s][0], x1 = x[s][1], x2 = x[s][2];
        float *a = act + s * kUvWaveLds + (lane & 15);
        auto put = [&](int f, float v) {
            a[(t0 + (f >> 2)) * 64 + (f & 3) * 16] = v;
            if (t0b >= 0) a[(t0b + (f >> 2)) * 64 + (f & 3) * 16] = v;
        };
        // the entries that are neither sine nor cosine -- the D coordinates and the zero padding past D + 2 N -- are 4 KT - 2 N in number, a multiple
        // of four: lane quarter kq takes the (kq + 4 i)-th of them
        const int npad4 = KT - N / 2;
#pragma unroll 1
        for (int i = 0; i < npad4; ++i) {
            const int j = kq + 4 * i;
            const float raw = j == 0 ? x0 : (j == 1 ? x1 : x2);
            put(j < D ? j : 2 * N + j, j < D ? raw : 0.0f);
        }
        // NO lane-dependent control flow in this function: uniform trip counts, and a lane quarter that runs out of pairs repeats its last one (same
        // values to the same slots).  Round 4's first version looped `for (g = kq; g < N; g += 4)` -- EXEC-masked loops in a kernel that runs at 512
        // registers with AGPR spill code around them -- and, TOGETHER with 64 more live registers elsewhere in the pass, gave wrong densities in 17 % of
        // the samples of the split kernel (or in 8 samples of the fp32 kernel, depending on the build), deterministically, while each change alone
        // passed every test.  The cause below the source was not found; the kernel keeps the property rounds 1-3 had by
        // accident: no divergent control flow between the ray set-up and the compositing.
#ifdef NGF_EXP_UV_LANE_LOOPS     // EXPERIMENT (tests/test_isa_lint.py, profiles/micro/uv_exec_agpr): round 4's first form, the lane-dependent trip count -- an EXEC-masked loop
#pragma unroll 1
        for (int g = kq; g < N; g += 4) {
#else
#pragma unroll 1
        for (int i = 0; i < (N + 3) / 4; ++i) {
            int g = kq + 4 * i;
            g = g < N ? g : g - 4;
#endif
            const int dim = g / F, fr = g - dim * F;
            const float xd = dim == 0 ? x0 : (dim == 1 ? x1 : x2);
            float sn, cs;
            sincos_small(xd * (float)(1 << fr), sn, cs);
            put(D + g, sn);
            put(D + N + g, cs);
        }
    }
    uv_lds_order();
}

// A run of n 256 -> 256 layers (weights w + l * 65536, split images wq + l * kUvQLayer, biases b + l * 256) followed by an output layer with <= 3
// units, whose weights (w_out) are requested here.  (Requested one step earlier -- between the last layer's MFMAs and its activation store, a peeled
// last iteration -- they would be covered by the store's ~1.8 k cycles; measured: 0.2 % SLOWER, 64 more live registers and 30 % more code.)
#ifdef NGF_EXP_UV_SAMEW
constexpr size_t kUvLayerStride = 0;          // every layer of a run reads the run's first weights (timing experiment, wrong pixels)
#else
constexpr size_t kUvLayerStride = 65536;
#endif
template <int NS, int MT, int END>
__device__ __forceinline__ void store_act_rows_next_bias_range(unsigned addr, f32x4 acc[NS][16], const float *bias_lane)
{
    if constexpr (MT < END) {
        if constexpr (NS > 0) { ds_write2_rows_agpr<MT * 4>(addr, acc[0][MT][0], acc[0][MT][1]); ds_write2_rows_agpr<MT * 4 + 2>(addr, acc[0][MT][2], acc[0][MT][3]); }
        if constexpr (NS > 1) { ds_write2_rows_agpr<kUvActSteps + MT * 4>(addr, acc[1][MT][0], acc[1][MT][1]); ds_write2_rows_agpr<kUvActSteps + MT * 4 + 2>(addr, acc[1][MT][2], acc[1][MT][3]); }
        const f32x4 v = *reinterpret_cast<const f32x4 *>(bias_lane + MT * 4);
#pragma unroll
        for (int s = 0; s < NS; ++s) acc[s][MT] = v;
        store_act_rows_next_bias_range<NS, MT + 1, END>(addr, acc, bias_lane);
    }
}
// A hidden layer's LAST k-step (NGF_EXP_UV_STORE_AFTER_LOOP: the store behind the k loop, round 4's earlier form) -- behind the MFMAs of tile group g the rows of group g - 1 (final by then, and 8 x NS / 2
// MFMAs = 256 cycles behind their last write: the matrix-write -> LDS-read wait states are covered) are stored and take the next layer's bias
template <int NS>
__device__ __forceinline__ void kstep_last(const KStepA<16, NS> &k, f32x4 acc[NS][16], const float *w, int t, int lane, KStepA<16, NS> &kn, unsigned addr,
                                           const float *bias_lane)
{
#define NGF_UV_LAST_GROUP(g)                                                                                                  \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                                             \
        _Pragma("unroll") for (int s = 0; s < NS; ++s) acc[s][4 * g + e] = NGF_UV_MFMA(k.a[g][e], k.b[s], acc[s][4 * g + e]);    \
    __builtin_amdgcn_sched_barrier(0);                                                                                        \
    kload_wg<16, NS>(w, t, g, lane, kn);
    // (the wait states in front of every group's stores as well: the MFMAs of group g normally stand between group g - 1's last write and its stores, but
    // they are pure values the instruction selector may emit elsewhere, and the hazard recogniser does not look into inline assembly)
#define NGF_UV_STORE_GROUP(g) asm volatile(NGF_UV_MFMA_LDS_NOPS ::: "memory"); store_act_rows_next_bias_range<NS, 4 * (g), 4 * (g) + 4>(addr, acc, bias_lane);
    NGF_UV_LAST_GROUP(0) __builtin_amdgcn_sched_barrier(0);
    NGF_UV_LAST_GROUP(1) NGF_UV_STORE_GROUP(0) __builtin_amdgcn_sched_barrier(0);
    NGF_UV_LAST_GROUP(2) NGF_UV_STORE_GROUP(1) __builtin_amdgcn_sched_barrier(0);
    NGF_UV_LAST_GROUP(3) NGF_UV_STORE_GROUP(2) __builtin_amdgcn_sched_barrier(0);
    NGF_UV_STORE_GROUP(3)
#undef NGF_UV_LAST_GROUP
#undef NGF_UV_STORE_GROUP
}

// fp32, one wave per SIMD: ONE software pipeline across the run's layers.  The layers' weights are contiguous, so the load group that follows a layer's
// last k-steps ("k-steps 64..67") IS the next layer's first four k-steps: they arrive behind the last MFMAs and the activation store instead of in front
// of an idle matrix pipe (the per-layer prologue was 1.35 % of a wave's life, profiles/r04_uv_sections.txt); the B operands read with them (LDS rows
// 64..67: they exist) are re-read once the layer's outputs are stored, and the next bias is loaded tile by tile inside the store.
template <int NS, int ACT>
__device__ __forceinline__ void hidden_run_deep(const float *w, const float *b, int n, int lane, float *act, f32x4 x[NS][16], const KStepA<16, NS> *pre, bool use_pre NGF_UVSEC_PARAM)
{
    KStepA<16, NS> a0, a1, a2, a3, b0, b1, b2, b3;
    const int kq = lane >> 4;
    if (use_pre) { a0 = pre[0]; a1 = pre[1]; a2 = pre[2]; a3 = pre[3]; }
    else { kload_w<16, NS>(w, 0, lane, a0); kload_w<16, NS>(w, 1, lane, a1); kload_w<16, NS>(w, 2, lane, a2); kload_w<16, NS>(w, 3, lane, a3); }
    load_bias<16, NS>(b, kq, x);
#pragma unroll 1
    for (int l = 0; l < n; ++l) {
        const float *wl = w + (size_t)l * kUvLayerStride;
        const bool more = kUvLayerStride != 0 && l + 1 < n;
        NGF_UVSEC_T(ts0);
        kload_b<16, NS>(act, 0, lane, a0); kload_b<16, NS>(act, 0, lane, a1, 1); kload_b<16, NS>(act, 0, lane, a2, 2); kload_b<16, NS>(act, 0, lane, a3, 3);
#ifdef NGF_EXP_UV_SECTIONS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
        NGF_UVSEC_T(ts1);
        NGF_UVSEC_ADD(0, ts0, ts1);
#ifndef NGF_EXP_UV_ACT_PER_GROUP
        // ONE cluster of vector instructions per EIGHT k-steps (256 MFMAs at NS = 2): the activations of both groups at the top of the first, all LDS
        // reads of the next eight k-steps at the top of the second (LDS reads are no vector instructions: no switch of the datapath).  The second
        // group's operands wait in eight registers of their own (bn) while the current ones are in use.
        float bn[4][NS];
        {
            const float *ap = act + 4 * 64 + lane;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int s = 0; s < NS; ++s) bn[j][s] = ap[s * kUvWaveLds + j * 64];
        }
        // LDS byte address of the lane's slot in row t + 8, advanced in the activation cluster and kept opaque: formed where the reads are (from t),
        // it is one more vector instruction in the middle of the 256 MFMAs -- one more switch
        typedef __attribute__((address_space(3))) const float lds_cf;
        unsigned la = (unsigned)(size_t)(__attribute__((address_space(3))) float *)(act + lane);
#ifndef NGF_EXP_UV_STORE_AFTER_LOOP
        constexpr int T_LOOP = 56;          // the eighth iteration is peeled below: its last k-step stores the layer's outputs between its MFMAs
#else
        constexpr int T_LOOP = 64;
#endif
#pragma unroll 1
        for (int t = 0; t < T_LOOP; t += 8) {
            la += 8 * 64 * 4;
            asm volatile("" : "+v"(la));
            const unsigned cur = la;
            kact<ACT, true>(a0, 0, 1); kact<ACT, true>(a1, 0, 1); kact<ACT, true>(a2, 0, 1); kact<ACT, true>(a3, 0, 1);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                b0.b[s] = act_in<ACT, true>(bn[0][s]); b1.b[s] = act_in<ACT, true>(bn[1][s]);
                b2.b[s] = act_in<ACT, true>(bn[2][s]); b3.b[s] = act_in<ACT, true>(bn[3][s]);
            }
            __builtin_amdgcn_sched_barrier(0);
            kstep<16, NS>(a0, x, wl, t + 4, lane, b0); kstep<16, NS>(a1, x, wl, t + 5, lane, b1);
            kstep<16, NS>(a2, x, wl, t + 6, lane, b2); kstep<16, NS>(a3, x, wl, t + 7, lane, b3);
            __builtin_amdgcn_sched_barrier(0);
            const int ta = (t + 8 < 64 || more) ? t + 8 : t;       // the run's last layer: harmless reload instead of reading behind the run
            {
                lds_cf *ap = (lds_cf *)(size_t)cur;         // rows t + 8 .. t + 15 <= 71 (80 exist); behind a layer's last k-steps the values are re-read above
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    a0.b[s] = ap[s * kUvWaveLds]; a1.b[s] = ap[s * kUvWaveLds + 64]; a2.b[s] = ap[s * kUvWaveLds + 128]; a3.b[s] = ap[s * kUvWaveLds + 192];
#pragma unroll
                    for (int j = 0; j < 4; ++j) bn[j][s] = ap[s * kUvWaveLds + (4 + j) * 64];
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            kstep<16, NS>(b0, x, wl, ta, lane, a0); kstep<16, NS>(b1, x, wl, ta + 1, lane, a1);
            kstep<16, NS>(b2, x, wl, ta + 2, lane, a2); kstep<16, NS>(b3, x, wl, ta + 3, lane, a3);
            __builtin_amdgcn_sched_barrier(0);
        }
#ifndef NGF_EXP_UV_STORE_AFTER_LOOP
        {
            kact<ACT, true>(a0, 0, 1); kact<ACT, true>(a1, 0, 1); kact<ACT, true>(a2, 0, 1); kact<ACT, true>(a3, 0, 1);
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                b0.b[s] = act_in<ACT, true>(bn[0][s]); b1.b[s] = act_in<ACT, true>(bn[1][s]);
                b2.b[s] = act_in<ACT, true>(bn[2][s]); b3.b[s] = act_in<ACT, true>(bn[3][s]);
            }
            __builtin_amdgcn_sched_barrier(0);
            kstep<16, NS>(a0, x, wl, 60, lane, b0); kstep<16, NS>(a1, x, wl, 61, lane, b1);
            kstep<16, NS>(a2, x, wl, 62, lane, b2); kstep<16, NS>(a3, x, wl, 63, lane, b3);
            __builtin_amdgcn_sched_barrier(0);
            const int ta = more ? 64 : 56;
            const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) float *)(act + lane);
            const float *bias_lane = b + (more ? l + 1 : l) * 256 + kq * 64;
            kstep<16, NS>(b0, x, wl, ta, lane, a0); kstep<16, NS>(b1, x, wl, ta + 1, lane, a1); kstep<16, NS>(b2, x, wl, ta + 2, lane, a2);
            kstep_last<NS>(b3, x, wl, ta + 3, lane, a3, addr, bias_lane);
            __builtin_amdgcn_sched_barrier(0);
        }
#endif
#else
#pragma unroll 1
        for (int t = 0; t < 64; t += 8) {
            kgroup<ACT, 16, NS>(a0, a1, a2, a3, x, wl, act, 0, 1 << 30, t + 4, lane, b0, b1, b2, b3);
            const int ta = (t + 8 < 64 || more) ? t + 8 : t;       // the run's last layer: harmless reload instead of reading behind the run
            kgroup<ACT, 16, NS>(b0, b1, b2, b3, x, wl, act, 0, 1 << 30, ta, lane, a0, a1, a2, a3);
        }
#endif
        NGF_UVSEC_T(ts2);
        NGF_UVSEC_ADD(1, ts1, ts2);
#ifdef NGF_EXP_UV_SECTIONS
        uvsec.t[6] += (unsigned long long)(64 * 16 * NS);
#endif
#if defined(NGF_EXP_UV_STORE_AFTER_LOOP) || defined(NGF_EXP_UV_ACT_PER_GROUP)
        {
            const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) float *)(act + lane);
            asm volatile(NGF_UV_MFMA_LDS_NOPS ::: "memory");
            // (behind the run's last layer: the same bias again, unused)
            store_act_rows_next_bias<16, NS>(addr, x, b + (more ? l + 1 : l) * 256 + kq * 64);
        }
#endif
        NGF_UVSEC_T(ts3);
        NGF_UVSEC_ADD(2, ts2, ts3);
    }
}

template <int NS, bool SPLIT, int LEAKY, bool DEEP = (NS > 1)>
__device__ __forceinline__ void hidden_run(const UvArgs &A, const float *w, const float *wq, const float *b, int n, int lane, float *act, f32x4 x[NS][16],
                                           const float *w_out, UvOutW<64> &ow NGF_UVSEC_PARAM, const KStepA<16, NS> *pre = nullptr, bool use_pre = false)
{
    if constexpr (!SPLIT && DEEP) hidden_run_deep<NS, LEAKY>(w, b, n, lane, act, x, pre, use_pre NGF_UVSEC_ARG);
    else {
#pragma unroll 1
    for (int l = 0; l < n; ++l) {
        dense256<NS, SPLIT, DEEP, kUvRd<SPLIT, LEAKY>>(A, w + (size_t)l * kUvLayerStride, wq + (size_t)l * kUvQLayer, b + l * 256, 64, lane, act, x NGF_UVSEC_ARG);
        NGF_UVSEC_T(u0);
        store_act<16, NS, kUvSt<SPLIT, LEAKY>>(act, lane, x);
        NGF_UVSEC_T(u1);
        NGF_UVSEC_ADD(2, u0, u1);
    }
    }
    out_prefetch<64>(w_out, lane, ow);
    __builtin_amdgcn_sched_barrier(0);
}

// ---- the three networks for NS x 16 samples -----------------------------------------------------------------------------
// p: position of the lane's sample in each tile, v: its ray direction.  Returns sigma and colour (identical in the 4 lanes of a
// sample).
template <int NS, bool SPLIT, bool DEEP = (NS > 1)>
__device__ __forceinline__ void uv_networks(const UvArgs &A, float *act, int lane, const float p[NS][3], const float v[NS][3], float sigma[NS],
                                            float col[NS][3] NGF_UVSEC_PARAM)
{
    // (the weight base plus an opaque zero and the lane id behind an opaque asm, once per pass: otherwise hipcc forms the ~60 per-layer weight / bias addresses of the
    // lane ONCE, outside the persistent ray loop, and parks them in scratch -- 143 spilled dwords, reloaded in front of every layer)
    // (an opaque ZERO OFFSET, not an opaque pointer: behind the asm a pointer loses its address space and every weight load becomes a flat_load,
    // which counts on both memory counters -- the k loop then waits vmcnt(0) lgkmcnt(0) once per trip: 34.3 -> 37.2 cycles per MFMA, measured)
    int wz = 0;
    asm volatile("" : "+s"(wz), "+v"(lane));
    const float *W = A.w + wz;
#ifdef NGF_EXP_UV_SAMEW
#define NGF_UV_SAME(x) A.geo_wh
#else
#define NGF_UV_SAME(x) (x)
#endif
    f32x4 x[NS][16];
    // geometry: 63 -> 256 -> (10x) 256 -> 1, ReLU.  PE10(p) goes to rows 0..15 AND to rows 64..79, where the gauge network finds it later
    // (one wave per SIMD: the first weights of a 256-wide layer are requested BEFORE the encodings / the store that produce its inputs)
    KStepA<16, NS> pre[4];
    constexpr bool PRE = DEEP && !SPLIT, PRE0 = PRE;         // the fp32 one-wave-per-SIMD kernel (the split-bf16 kernel measured 0.4 % slower with its fp32 layers 0 prefetched)
    if constexpr (PRE0) uv_wprefetch<16, NS>(W + A.geo_w0, lane, pre);
    { NGF_UVSEC_T(u11a); store_pe<3, 10, NS>(act, 0, 16, lane, p, 64); NGF_UVSEC_T(u11b); NGF_UVSEC_ADD(3, u11a, u11b); }
    dense<16, NS, DEEP>(W + A.geo_w0, W + A.geo_b0, 16, lane, act, x NGF_UVSEC_ARG, 1 << 30, pre, PRE0);
    if constexpr (PRE) uv_wprefetch<16, NS>(W + A.geo_wh, lane, pre);
    { NGF_UVSEC_T(u1a); store_act<16, NS, kUvSt<SPLIT, 0>>(act, lane, x); NGF_UVSEC_T(u1b); NGF_UVSEC_ADD(2, u1a, u1b); }
    {
        UvOutW<64> ow;
        hidden_run<NS, SPLIT, 0, DEEP>(A, W + A.geo_wh, W + A.geo_qh, W + A.geo_bh, 10, lane, act, x, W + A.geo_wo, ow NGF_UVSEC_ARG, pre, PRE);
        f32x4 o[NS];
        { NGF_UVSEC_T(u16a); dense_out<NS, 64, kUvRd<SPLIT, 0>, DEEP>(ow, W + A.geo_bo, lane, act, o); NGF_UVSEC_T(u16b); NGF_UVSEC_ADD(4, u16a, u16b); }
#pragma unroll
        for (int s = 0; s < NS; ++s) sigma[s] = o[s][0] > 20.0f ? o[s][0] : log1pf(expf(o[s][0]));
    }
    // gauge: 63 -> 64 -> 128 -> 128 -> 128 -> 3|2, ReLU (first layer on the copy of PE10(p) in rows 64..79)
    float uv[NS][3];
    {
        f32x4 g4[NS][4], g[NS][8];         // (the gauge layers' first weights requested ahead of their inputs, as for the 256-wide layers: measured neutral, not kept)
        dense<4, NS, DEEP>(W + A.ga_w0, W + A.ga_b0, 16, lane, act + 64 * 64, g4 NGF_UVSEC_ARG);
        { NGF_UVSEC_T(u3a); store_act<4, NS, kUvSt<SPLIT, 0>>(act, lane, g4); NGF_UVSEC_T(u3b); NGF_UVSEC_ADD(2, u3a, u3b); }
        dense<8, NS, DEEP, kUvRd<SPLIT, 0>>(W + A.ga_w1, W + A.ga_b1, 16, lane, act, g NGF_UVSEC_ARG);
        { NGF_UVSEC_T(u4a); store_act<8, NS, kUvSt<SPLIT, 0>>(act, lane, g); NGF_UVSEC_T(u4b); NGF_UVSEC_ADD(2, u4a, u4b); }
        dense<8, NS, DEEP, kUvRd<SPLIT, 0>>(W + A.ga_w2, W + A.ga_b2, 32, lane, act, g NGF_UVSEC_ARG);
        { NGF_UVSEC_T(u5a); store_act<8, NS, kUvSt<SPLIT, 0>>(act, lane, g); NGF_UVSEC_T(u5b); NGF_UVSEC_ADD(2, u5a, u5b); }
        dense<8, NS, DEEP, kUvRd<SPLIT, 0>>(W + A.ga_w3, W + A.ga_b3, 32, lane, act, g NGF_UVSEC_ARG);
        { NGF_UVSEC_T(u6a); store_act<8, NS, kUvSt<SPLIT, 0>>(act, lane, g); NGF_UVSEC_T(u6b); NGF_UVSEC_ADD(2, u6a, u6b); }
        UvOutW<32> ow;
        out_prefetch<32>(W + A.ga_wo, lane, ow);
        __builtin_amdgcn_sched_barrier(0);
        f32x4 q[NS];
        { NGF_UVSEC_T(u17a); dense_out<NS, 32, kUvRd<SPLIT, 0>, DEEP>(ow, W + A.ga_bo, lane, act, q); NGF_UVSEC_T(u17b); NGF_UVSEC_ADD(4, u17a, u17b); }
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if (A.sphere) {
                float nrm = sqrtf((q[s][0] * q[s][0] + q[s][1] * q[s][1]) + q[s][2] * q[s][2]);
                nrm = fmaxf(nrm, 1e-12f);
                uv[s][0] = q[s][0] / nrm; uv[s][1] = q[s][1] / nrm; uv[s][2] = q[s][2] / nrm;
            } else {
                uv[s][0] = tanhf(q[s][0]); uv[s][1] = tanhf(q[s][1]); uv[s][2] = 0.0f;
            }
        }
    }
    // texture block1: (63|42) -> 256 -> (5x) 256, LeakyReLU(0.2)
    if constexpr (PRE0) uv_wprefetch<16, NS>(W + A.t1_w0, lane, pre);
    if (A.sphere) {
        { NGF_UVSEC_T(u13a); store_pe<3, 10, NS>(act, 0, 16, lane, uv); NGF_UVSEC_T(u13b); NGF_UVSEC_ADD(3, u13a, u13b); }
        dense<16, NS, DEEP>(W + A.t1_w0, W + A.t1_b0, 16, lane, act, x NGF_UVSEC_ARG, 1 << 30, pre, PRE0);
    } else {
        { NGF_UVSEC_T(u14a); store_pe<2, 10, NS>(act, 0, 12, lane, uv); NGF_UVSEC_T(u14b); NGF_UVSEC_ADD(3, u14a, u14b); }
        dense<16, NS, DEEP>(W + A.t1_w0, W + A.t1_b0, 12, lane, act, x NGF_UVSEC_ARG, 1 << 30, pre, PRE0);
    }
    if constexpr (PRE) uv_wprefetch<16, NS>(W + NGF_UV_SAME(A.t1_wh), lane, pre);
    { NGF_UVSEC_T(u7a); store_act<16, NS, kUvSt<SPLIT, 1>>(act, lane, x); NGF_UVSEC_T(u7b); NGF_UVSEC_ADD(2, u7a, u7b); }
    // act[0..63] = block1 output h; color1 and block2 both read it
    f32x4 c1[NS], c2[NS];
    {
        UvOutW<64> ow;
        hidden_run<NS, SPLIT, 1, DEEP>(A, W + NGF_UV_SAME(A.t1_wh), W + A.t1_qh, W + A.t1_bh, 5, lane, act, x, W + A.c1_w, ow NGF_UVSEC_ARG, pre, PRE);
        { NGF_UVSEC_T(u18a); dense_out<NS, 64, kUvRd<SPLIT, 1>, DEEP>(ow, W + A.c1_b, lane, act, c1); NGF_UVSEC_T(u18b); NGF_UVSEC_ADD(4, u18a, u18b); }
    }
    // block2: [h(256), v(3), PE6(v)(36)] -> 256 -> (3x) 256 -> 3
    if constexpr (PRE) uv_wprefetch<16, NS>(W + A.t2_w0, lane, pre);
    { NGF_UVSEC_T(u15a); store_pe<3, 6, NS>(act, 64, 12, lane, v); NGF_UVSEC_T(u15b); NGF_UVSEC_ADD(3, u15a, u15b); }                 // 39 inputs + zero padding up to k-step 76
    dense256<NS, SPLIT, DEEP, kUvRd<SPLIT, 1>>(A, W + A.t2_w0, W + A.t2_q0, W + A.t2_b0, 76, lane, act, x NGF_UVSEC_ARG, 64, pre, PRE);            // 76 k-steps -> 10 k-blocks in split mode (rows 76..79: zero weights on finite leftovers)
    if constexpr (PRE) uv_wprefetch<16, NS>(W + NGF_UV_SAME(A.t2_wh), lane, pre);
    { NGF_UVSEC_T(u9a); store_act<16, NS, kUvSt<SPLIT, 1>>(act, lane, x); NGF_UVSEC_T(u9b); NGF_UVSEC_ADD(2, u9a, u9b); }
    {
        UvOutW<64> ow;
        hidden_run<NS, SPLIT, 1, DEEP>(A, W + NGF_UV_SAME(A.t2_wh), W + A.t2_qh, W + A.t2_bh, 3, lane, act, x, W + A.t2_wo, ow NGF_UVSEC_ARG, pre, PRE);
        { NGF_UVSEC_T(u19a); dense_out<NS, 64, kUvRd<SPLIT, 1>, DEEP>(ow, W + A.t2_bo, lane, act, c2); NGF_UVSEC_T(u19b); NGF_UVSEC_ADD(4, u19a, u19b); }
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        float orig[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const float s1 = c1[s][k] > 20.0f ? c1[s][k] : log1pf(expf(c1[s][k]));      // softplus(color1) (clamp = False, decoder.py:66-67)
            orig[k] = s1 + c2[s][k];
            col[s][k] = fmaxf(orig[k], 0.0f);                                         // (color1 + color2).clamp(min=0)   decoder.py:78
        }
        if (A.tex) uv_texture_edit(A.tex, A.tex_h, A.tex_w, A.tex_c, A.tex_mode, A.sphere, uv[s], orig, col[s]);      // decoder.py:79-121
    }
}

// One wave renders NS rays at a time: lane i owns sample i of a 64-sample chunk of each (segment jitter, prefix sum, position,
// in-cube test); the in-cube samples of the NS rays form ONE list (ray 0's first) that is pushed through the networks NS x 16 at
// a time.
template <int NS, bool SPLIT = false>
__global__ void __launch_bounds__(512 / NS) uv_render_kernel(const UvArgs A)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int lane = threadIdx.x & 63;
    float *act = smem + (threadIdx.x >> 6) * (NS * kUvWaveLds);
    const int S = A.S;
    if constexpr (SPLIT) {
        // block2.0 has 76 k-steps = 9.5 k-blocks: its tenth k-block also reads rows 76..79, which nothing else writes.  Their weights
        // are zero, but LDS keeps whatever the previous kernel left there (a NaN or Inf pattern would poison the sums): zero them once.
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int t = 76; t < kUvActSteps; ++t) act[s * kUvWaveLds + t * 64 + lane] = 0.0f;
    }
    unsigned long long st_samples = 0, st_pass = 0;
#ifdef NGF_EXP_UV_SECTIONS
    UvSec uvsec = {{0, 0, 0, 0, 0, 0, 0, 0}};
    const unsigned long long uv_t_begin = __builtin_readcyclecounter();
#endif
    const float dt = (float)(2.0 / S), dtj = (float)((2.0 / S) * 0.05);     // renderer.py:107-117 (python floats)
    for (;;) {
        unsigned int ray0 = 0;
        if (lane == 0) ray0 = atomicAdd(A.ray_counter, (unsigned)NS);
        ray0 = __builtin_amdgcn_readfirstlane(ray0);
        if ((int64_t)ray0 >= A.R) break;
        float d[NS][3], t0[NS], T[NS], rc[NS][3], cp[NS][3];
        double cum[NS];
        bool rlive[NS];
        size_t rayi[NS];
#pragma unroll
        for (int j = 0; j < NS; ++j) {
            rlive[j] = (int64_t)ray0 + j < A.R;
            rayi[j] = rlive[j] ? (size_t)ray0 + j : (size_t)ray0;
#pragma unroll
            for (int k = 0; k < 3; ++k) d[j][k] = A.raydir[rayi[j] * 3 + k];
            // the ray's camera: by value, or row (ray / rays_per_cam) of the device table (wave-uniform index: scalar loads)
            const uint32_t ci = A.cam_dev ? (uint32_t)rayi[j] / A.rays_per_cam : 0u;
#pragma unroll
            for (int k = 0; k < 3; ++k) cp[j][k] = A.cam_dev ? A.cam_dev[(size_t)ci * 3 + k] : A.campos[k];
            // slab test (renderer.py:90-105)
            float t1[3], t2[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { t1[k] = (-1.0f - cp[j][k]) / d[j][k]; t2[k] = (1.0f - cp[j][k]) / d[j][k]; }
            const float tmin = fmaxf(fminf(t1[0], t2[0]), fmaxf(fminf(t1[1], t2[1]), fminf(t1[2], t2[2])));
            const float tmax = fminf(fmaxf(t1[0], t2[0]), fminf(fmaxf(t1[1], t2[1]), fmaxf(t1[2], t2[2])));
            t0[j] = fmaxf((tmin < tmax) ? tmin : 0.0f, 0.0f);
            cum[j] = 0.0;          // torch.cumsum accumulates float32 in double on the CPU path we are pinned to
            T[j] = 1.0f; rc[j][0] = rc[j][1] = rc[j][2] = 0.0f;
        }
        for (int base = 0; base < S; base += 64) {
            const int i = base + lane;
            const bool in = i < S;
            float seg[NS], p[NS][3], sigma[NS], col[NS][3];
            bool valid[NS];
            unsigned long long vm[NS];
            int nv[NS], rank[NS], total = 0;
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                seg[j] = in ? dt + dtj * (A.U[rayi[j] * S + i] - 0.5f) : 0.0f;
                // inclusive prefix sum over the chunk, sequential in the sample index
                double mycum = 0.0, prev = 0.0;
                for (int q = 0; q < 64; ++q) {
                    const double before = cum[j];
                    cum[j] += (double)__shfl(seg[j], q);
                    if (lane == q) { mycum = cum[j]; prev = before; }
                }
                const float e0 = t0[j] + (float)prev, e1 = t0[j] + (float)mycum;
                const float mid = (e0 + e1) / 2.0f;
                valid[j] = in && rlive[j];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    p[j][k] = cp[j][k] + d[j][k] * mid;
                    valid[j] = valid[j] && (p[j][k] > -1.0f) && (p[j][k] < 1.0f);
                }
                vm[j] = __ballot(valid[j]);
                nv[j] = __popcll(vm[j]);
                rank[j] = __popcll(vm[j] & ((1ull << lane) - 1ull)) + total;     // index in the combined list
                total += nv[j];
                sigma[j] = 0.0f; col[j][0] = col[j][1] = col[j][2] = 0.0f;
            }
            st_samples += total;
            st_pass += (total + 15) / 16;
            for (int g0 = 0; g0 < total; g0 += 16 * NS) {
                // lane (s, kq) of tile ts evaluates entry g0 + 16 ts + s of the combined list: find its ray and owner lane
                float q[NS][3], vq[NS][3];
#pragma unroll
                for (int ts = 0; ts < NS; ++ts) {
                    int want = g0 + 16 * ts + (lane & 15);
                    if (want >= total) want = g0;                        // padding: reuse the first entry of the pass
                    int jr = 0, off = 0;
#pragma unroll
                    for (int j = 0; j + 1 < NS; ++j)
                        if (want >= off + nv[j]) { off += nv[j]; jr = j + 1; }
                    unsigned long long m = vm[0];
#pragma unroll
                    for (int j = 1; j < NS; ++j) m = jr == j ? vm[j] : m;
                    for (int b = 0; b < want - off; ++b) m &= m - 1;     // select the (want - off)-th set bit
                    const int owner = __ffsll((long long)m) - 1;
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        // (the directions through an opaque copy: a select between elements of d[][] becomes an indexed load of the array,
                        // and the array then lives in scratch)
                        float pv = __shfl(p[0][k], owner), dv = d[0][k];
                        asm volatile("" : "+v"(dv));
#pragma unroll
                        for (int j = 1; j < NS; ++j) {
                            const float pj = __shfl(p[j][k], owner);
                            float dj = d[j][k];
                            asm volatile("" : "+v"(dj));
                            pv = jr == j ? pj : pv;
                            dv = jr == j ? dj : dv;
                        }
                        q[ts][k] = pv; vq[ts][k] = dv;
                    }
                }
                float sg[NS], cc[NS][3];
                NGF_UVSEC_T(un0);
                // the pair's last pass may hold one tile only (6 % of the passes of the DTU frame): the single-tile networks cost about 0.6 of a
                // two-tile pass (same weight stream, half the matrix instructions).  Same arithmetic per sample: a sample's result does not depend on
                // which tile, or which instantiation, evaluates it (the k loops accumulate in the same order).
                // (fp32 kernel only: the split kernel is bound by its weight stream, which a single tile does not shorten -- 41.1 -> 43.7 ms with it)
                if (NS == 2 && !SPLIT && total - g0 <= 16) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) cc[NS - 1][k] = 0.0f;
                    sg[NS - 1] = 0.0f;
                    uv_networks<1, SPLIT, true>(A, act, lane, q, vq, sg, cc NGF_UVSEC_ARG);
                } else
                uv_networks<NS, SPLIT>(A, act, lane, q, vq, sg, cc NGF_UVSEC_ARG);
                NGF_UVSEC_T(un1);
                NGF_UVSEC_ADD(5, un0, un1);          // section 5 = the three networks of a pass, everything included
                // owners pull their result from lane slot (kq = 0 copy) of their tile
#pragma unroll
                for (int j = 0; j < NS; ++j) {
                    const int rel = rank[j] - g0;
                    const bool mine = valid[j] && rel >= 0 && rel < 16 * NS;
                    const int slot = rel & 15, tsel = (rel >> 4) & (NS - 1);
                    float psg = __shfl(sg[0], slot), p0 = __shfl(cc[0][0], slot), p1 = __shfl(cc[0][1], slot), p2 = __shfl(cc[0][2], slot);
#pragma unroll
                    for (int ts = 1; ts < NS; ++ts) {
                        const float a = __shfl(sg[ts], slot), b0 = __shfl(cc[ts][0], slot), b1 = __shfl(cc[ts][1], slot), b2 = __shfl(cc[ts][2], slot);
                        if (tsel == ts) { psg = a; p0 = b0; p1 = b1; p2 = b2; }
                    }
                    if (mine) { sigma[j] = psg; col[j][0] = p0; col[j][1] = p1; col[j][2] = p2; }
                }
            }
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                if (A.dbg_sigma && in && rlive[j]) {
                    A.dbg_sigma[rayi[j] * S + i] = sigma[j];
#pragma unroll
                    for (int k = 0; k < 3; ++k) A.dbg_col[(rayi[j] * S + i) * 3 + k] = col[j][k];
                }
                // ray_march (renderer.py:222-233): sequential in the sample index
                const float op = 1.0f - expf(-(sigma[j] * (valid[j] ? 1.0f : 0.0f)) * seg[j]);
                for (int q = 0; q < 64; ++q) {
                    if (base + q >= S) break;
                    const float o = __shfl(op, q);
                    const float w = o * T[j];
                    T[j] = T[j] * ((1.0f - o) + 1e-10f);
#pragma unroll
                    for (int k = 0; k < 3; ++k) rc[j][k] += __shfl(col[j][k], q) * w;
                }
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int j = 0; j < NS; ++j) {
                if (!rlive[j]) continue;
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    float c = rc[j][k];
                    if (A.bg_dev) c += A.bg_dev[(size_t)((uint32_t)rayi[j] / A.rays_per_cam) * 3 + k] * T[j];
                    else if (A.has_bg) c += A.bg[k] * T[j];
                    c = powf(c * 1.0f + 1e-5f, (float)(1.0 / 2.2));          // simple_tone_map (renderer.py:7-8)
                    A.color[rayi[j] * 3 + k] = fminf(fmaxf(c, 0.0f), 1.0f);
                }
                A.trans[rayi[j]] = T[j];
            }
        }
    }
    if (A.stats && lane == 0) {
        atomicAdd(A.stats + 0, st_samples);
        atomicAdd(A.stats + 1, st_pass);
#ifdef NGF_EXP_UV_SECTIONS
        uvsec.t[7] = __builtin_readcyclecounter() - uv_t_begin;       // the wave's life
        for (int k = 0; k < 8; ++k) atomicAdd(A.stats + 2 + k, uvsec.t[k]);
#endif
    }
}

}  // namespace ngf
