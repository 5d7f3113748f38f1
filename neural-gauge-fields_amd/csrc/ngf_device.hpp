// ngf_device.hpp -- shared device-side definitions of the gfx950 ray-march kernels.
//
// Build note: the translation unit is compiled with -ffp-contract=off.  Every fused multiply-add
// in this code is an explicit fmaf(); everything written as a*b+c is two roundings.  That keeps
// the sample positions, the in-box test and the bilinear cell selection bit-identical to the
// reference's eager fp32 arithmetic (SURVEY.md section 7 hazard 1), which is what makes the
// threshold masks (valid, weight > thr) flip-free.
#pragma once
#include <cstddef>
#include <type_traits>
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace ngf {

constexpr int kWave = 64;
constexpr int kRing = 128;          // wave-private queue of active samples (records)
constexpr int kRecFloats = 8;       // {ray lane, weight, t_xy(2), t_yz(2), t_xz(2)}
constexpr int kViewFeat = 16;       // [d(3), sin(6), cos(6), 0]

// A packed texture: channel-last texels with a one-texel zero border (zeros padding of
// grid_sample becomes an in-bounds read of a zero texel).  p points at padded texel (0,0).
struct Tex {
    const float *p;
    int32_t W, H;       // un-padded size; u indexes W, v indexes H
    int32_t stride;     // W + 2 (texels per padded row)
    float fw, fh;       // (float)(W-1), (float)(H-1): host-computed so they live in SGPRs (a v_cvt of a
                        // uniform int is a VALU op whose loop-invariant result the compiler pins in a VGPR)
};

struct MaskVol {
    const uint8_t *bits;  // np.packbits image of [D,H,W]; NULL = none
    const uint8_t *cells; // [(D+1)][(H+1)][(W+1)]: per trilinear cell (base corner -1 .. D-1 per axis) the bits of its 8 corners, bit dz*4 + dy*2 + dx,
                          // corners outside the volume 0 (grid_sample's zeros padding) -- built from `bits` by mask_cells_kernel at create (round 6)
    int32_t D, H, W;
    float a0[3], inv[3];  // mask aabb[0], invgridSize = 1/(aabb1-aabb0)*2
    // Block images for the march's empty-space skipping (mask_clear_around): blocks of B^3 cells over the cell indices -16 .. X + 16 (block = (cell index + 16) >> log2 B; the
    // margin lands a position far outside in a block whose neighbours are outside too), dimensions ((X + 32) >> log2 B) + 1 per axis; 1 = the block and its 26 neighbours
    // hold no occupied corner, i.e. every cell within B cells per axis of a cell of this block is empty.
    const uint8_t *coarse; // B = 8: open space (up to 27 iterations of 8 steps per test at the headline geometry)
    const uint8_t *fine;   // B = 4: asked when the coarse image certifies nothing -- the gaps of cluttered occupancy, the approach to a surface
    int32_t cD, cH, cW;    // the coarse image's dimensions (host side; the kernels derive both from D, H, W)
};

struct RenderArgs {
    const float *rays;     // [n,6]
    const float *jitter;   // [n] or NULL
    float *rgb;            // [n,3]
    float *depth;          // [n]
    float *dbg_sigma;      // [n,S] or NULL   (ngf_field_march)
    float *dbg_weight;     // [n,S] or NULL
    unsigned long long *stats;  // 4 counters or NULL
    unsigned int *tile_counter; // the launch's slot of queue heads, all zero when the launch starts: [0] alone (one queue) or [0..7] (one queue per XCD,
                                // xcd_queues = 8); [8] counts the workgroups whose waves all found their queue empty -- the last of `queue_waves` zeroes the slot again
                                // (queue_done), so no fill kernel runs in front of a render launch
    int64_t n;
    int32_t S, white_bg, mode, skip_rgb;
    int32_t tile_w;        // rays per wave tile of the launch's FIRST (widest) plan segment: 64 (unsplit kernel) or 32 .. 1
    int32_t tile_shift;    // log2(tile_w); the split march gives every ray 64 >> tile_shift lanes (consecutive steps)
    // Tile plan (launch_render): the ray list is cut into up to four contiguous segments of tiles of decreasing width -- wide tiles first (the
    // efficient ones: fewer partial shade passes, finer early termination), narrow tiles for the last rays, so that the waves of the persistent
    // grid run dry together instead of one tile apart (a launch used to pay ~one tile duration, 0.17 ms, for its tail whatever its size).
    // Tile t belongs to segment g = #{k : t >= seg_end[k]}; its rays are seg_ray0[g] + ((t - seg_end[g-1]) << seg_shift[g]) ...
    uint32_t seg_end[3];   // exclusive tile-index ends of segments 0..2 (segment 3 ends at `tiles`)
    int32_t seg_shift[4];  // log2(tile width) of each segment
    int64_t seg_ray0[4];   // first ray of each segment
    int32_t xcd_queues;    // 8: the tile range is cut into 8 contiguous chunks, XCD x starts on chunk x and steals from the others when it is done
                           // (each XCD has its own 4 MiB L2: its concurrent tiles then cover ONE compact ray range instead of an eighth of everybody's);
                           // 0 / 1: one queue
    int32_t waves_active;  // waves of a workgroup that take tiles (the others leave after the LDS image barrier): < WAVES for launches with fewer tiles than
                           // resident waves, so that the working waves are spread over all CUs and SIMDs
    uint32_t tiles;        // number of tiles of the launch
    // Screen-space tile order (round 6; ngf_field_render_image: the caller says that the ray list is an image of `row_width` rays per row).  The queue
    // hands out positions 0, 1, 2 ...; position q < ord_n is the tile tile_order(q): the launch walks the image in blocks of ord_bh rows x ord_bw
    // tiles instead of row by row, so that the tiles in flight together cover a compact window -- narrow frusta whose footprints on the planes
    // are strips an XCD's L2 can hold -- instead of a fan over the whole plane.  Every tile still is 8 consecutive rays of one row and computes
    // what it always did: pixels are bit-identical, only the order changes.  ord_n = 0: the list's own order.
    uint32_t ord_n;        // queue positions that are re-ordered: the full bands of segment 0 (whole rows, ord_bh at a time)
    uint32_t ord_tpr;      // tiles per image row (row_width >> seg_shift[0])
    uint32_t ord_bw;       // block width in tiles
    uint32_t ord_bh;       // block height in rows
    uint32_t queue_waves;  // workgroups of the launch (each reports once, when its last working wave found the queue empty): what tile_counter[8] counts up to
    int32_t ablate;        // debug instantiations only (render_kernel<.., DBG = true>): 32 no early termination, 64 no empty-iteration skip, 128 no empty-space skipping through the mask's block image -- both
                           // EXACT (A/B timing and the bit-identity tests).  Round 1-2's bits 1 / 2 / 4 / 16 (skip collect, skip layers 2-3, cached
                           // gathers, wave priority) produced wrong images and are gone; the trainer keeps its own bits (ngf_train.hpp)
    float a0[3], a1[3], inv[3];
    float near_, far_, step, dscale, thr;
    Tex dens[3];           // TriPlane: 16-ch (faithful) or 1-ch (baked) density texels
    Tex app[3];            // colour texels (48 | 72 channels)
    Tex gau[3];            // 2-ch gauge offsets (TriPlane)
    MaskVol mask;
    const float *basis_pack;  // NGF_F_NO_FOLD: rgb_decoder.basis packed for the basis MFMA stage (ngf_shade16.hpp), else NULL
    const float *blob;     // packed MLP image (copied into LDS by every workgroup)
    int32_t blob_floats;
    float wd[48];          // TriPlane faithful density_decoder.weight
    float bd;              // density_decoder.bias
};

// The XCD (accelerator complex die, 0..7 on MI355X) this wave runs on: HW_REG_XCC_ID, bits [3:0].  s_getreg_b32 simm16 = (size-1) << 11 |
// offset << 6 | id with id 20, offset 0, size 4.
__device__ __forceinline__ int xcd_id() { return (int)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u); }

// Next tile of a render launch for this wave (wave-uniform result).  `first` is the wave's memory of how many chunks, starting at its own,
// are already exhausted: a chunk that ran dry stays dry, so it is not polled again.
__device__ __forceinline__ bool next_tile(const RenderArgs &A, int xcd, int lane, int &first, unsigned &tile)
{
    if (A.xcd_queues <= 1) {
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(A.tile_counter, 1u);
        tile = __builtin_amdgcn_readfirstlane(t);
        return tile < A.tiles;
    }
    for (; first < 8; ++first) {
        const unsigned c = (unsigned)(xcd + first) & 7u;
        const unsigned lo = (unsigned)(((unsigned long long)c * A.tiles) >> 3), hi = (unsigned)(((unsigned long long)(c + 1) * A.tiles) >> 3);
        if (hi == lo) continue;
        unsigned t = 0;
        if (lane == 0) t = atomicAdd(A.tile_counter + c, 1u);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t < hi - lo) { tile = lo + t; return true; }
    }
    return false;
}

// queue position -> tile (RenderArgs::ord_*): bands of ord_bh image rows, each walked block by block (the last block of a band may be narrower).
// Wave-uniform scalar arithmetic, once per tile.
__host__ __device__ __forceinline__ unsigned tile_order(unsigned q, unsigned n, unsigned tpr, unsigned bw, unsigned bh)
{
    if (q >= n) return q;
    const unsigned band_tiles = tpr * bh, full = bw * bh;
    const unsigned band = q / band_tiles, p = q - band * band_tiles;
    const unsigned last = (tpr - 1) / bw;                   // index of the band's last block
    unsigned b = p / full;
    b = b < last ? b : last;
    const unsigned inb = p - b * full;
    const unsigned wb = b == last ? tpr - last * bw : bw;
    const unsigned r = inb / wb, c = b * bw + (inb - r * wb);
    return (band * bh + r) * tpr + c;
}

// A wave that found the queue empty reports to its workgroup's LDS counter; the workgroup's last wave reports to the launch's slot, and the last
// WORKGROUP zeroes the slot for the slot's next launch.  Every wave's queue atomics have returned before its own report (it used their values), so
// when the count reaches queue_waves / waves_active nobody will touch the heads again.  Replaces a hipMemsetAsync per launch (a 4 us fill kernel +
// its launch gap in front of every render).  One global atomic per WORKGROUP: a first version had every wave report to the slot -- 3072 returning
// atomics on one address at the moment the launch ends, +30 us on every launch (profiles/r05_shard_latency.txt, first block).
__device__ __forceinline__ void queue_done(unsigned int *slot, unsigned *wg_done, unsigned waves_in_wg, unsigned workgroups, int lane)
{
    if (lane != 0) return;
    const unsigned d = __hip_atomic_fetch_add(wg_done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (d + 1u != waves_in_wg) return;
    // acq_rel: the report is ordered after this workgroup's queue atomics, and the last workgroup's zero stores after everybody's
    // report -- on today's hardware the relaxed form worked because device-scope atomics resolve at the L2, but nothing promised it.  One atomic
    // per workgroup and launch: the cache maintenance it implies is not measurable (profiles/r06_shard_latency.txt).
    const unsigned done = __hip_atomic_fetch_add(slot + 8, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (done + 1u == workgroups) {
#pragma unroll
        for (int k = 0; k < 9; ++k) __hip_atomic_store(slot + k, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// A Tex of the kernel's RenderArgs (byte offset `off` inside it), re-read from the kernel-argument segment on the scalar memory pipe at
// the point of use.  RenderArgs holds nine Tex (63 dwords) next to 48 decoder weights: kept live across the march loop they overflow the
// SGPR file and hipcc parks texture pointers in VGPR lanes -- ~38 v_readlane per march step on the vector pipe, which is the binding one
// (measured: R1 frame 9.93 -> 9.82 ms, same bits).  REQUIRES the RenderArgs to be the kernel's first argument (offset 0 of the kernarg
// segment): true for every kernel of this library (TrainArgs starts with its RenderArgs, static_assert in ngf_train.hpp), and CHECKED at
// run time: alpha_kernel and the DBG instantiations of render_kernel (every test goes through them) trap when the descriptor read from the
// kernarg segment is not the one in their `A`.
// The compile-time half of that contract (round 4; the production instantiation of render_kernel carries no run-time check): every kernel whose
// body reaches karg_tex asserts, through its OWN function type, that its first parameter is the RenderArgs (or a struct that starts with one).
template <typename K> struct kernel_first_param;
template <typename A0, typename... Rest> struct kernel_first_param<void (*)(A0, Rest...)> { typedef A0 type; };
template <typename K, typename Args = RenderArgs>
constexpr bool kernel_args_at_offset_0 = __is_same(typename kernel_first_param<K>::type, Args);
#define NGF_KARG_CONTRACT(kernel_ptr_expr) NGF_KARG_CONTRACT_T(kernel_ptr_expr, RenderArgs)
#define NGF_KARG_CONTRACT_T(kernel_ptr_expr, ArgsType) \
    static_assert(kernel_args_at_offset_0<decltype(kernel_ptr_expr), ArgsType>, "karg_tex reads the RenderArgs at offset 0 of the kernel-argument segment: it must be this kernel's first parameter")

__device__ __forceinline__ Tex karg_tex(size_t off)
{
    typedef const __attribute__((address_space(4))) Tex *tptr_t;
    tptr_t t = (tptr_t)((const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr() + off);
    asm volatile("" : "+s"(t));
    Tex r;
    r.p = t->p; r.W = t->W; r.H = t->H; r.stride = t->stride; r.fw = t->fw; r.fh = t->fh;
    return r;
}

// A plain field of the RenderArgs re-read from the kernel-argument segment (same contract as karg_tex).  Round 4: the per-TILE fields -- the tile
// plan's segments, the ray / output pointers, n, near / far -- are read this way at the tile's start and end: kept live across the march and shade
// loops they were among the ~45 scalars hipcc parks in VGPR lanes (v_writelane at kernel entry, v_readlane at every use: 22 vector instructions per
// march iteration of the level-3 kernel); a scalar load is latency the wave's neighbours cover, a v_readlane is an issue slot of the binding pipe.
template <typename T>
__device__ __forceinline__ T karg(size_t off)
{
    typedef const __attribute__((address_space(4))) T *ptr_t;
    ptr_t q = (ptr_t)((const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr() + off);
    asm volatile("" : "+s"(q));
    return *q;
}
// The alpha mask's descriptor re-read the same way (member by member, like karg_tex).
__device__ __forceinline__ MaskVol karg_mask(size_t off)
{
    typedef const __attribute__((address_space(4))) MaskVol *mptr_t;
    mptr_t t = (mptr_t)((const __attribute__((address_space(4))) char *)__builtin_amdgcn_kernarg_segment_ptr() + off);
    asm volatile("" : "+s"(t));
    MaskVol r;
    r.bits = t->bits; r.cells = t->cells; r.D = t->D; r.H = t->H; r.W = t->W;
#pragma unroll
    for (int k = 0; k < 3; ++k) { r.a0[k] = t->a0[k]; r.inv[k] = t->inv[k]; }
    r.coarse = t->coarse; r.fine = t->fine; r.cD = t->cD; r.cH = t->cH; r.cW = t->cW;
    return r;
}
#define NGF_KARG(field) karg<decltype(RenderArgs::field)>(offsetof(RenderArgs, field))
#define NGF_KARG_AT(field, k) karg<std::remove_extent_t<decltype(RenderArgs::field)>>(offsetof(RenderArgs, field) + (k) * sizeof(std::remove_extent_t<decltype(RenderArgs::field)>))

// ---- bilinear cell: ATen grid_sampler_2d, align_corners=True, padding_mode='zeros' -------------
struct Bil {
    int32_t idx;                 // padded texel index of the (x0,y0) tap; +1, +stride, +stride+1 are the others
    int32_t cx, cy;              // the same tap as padded (column, row); only the LDS-staged variant reads them
    float w00, w10, w01, w11;    // (1-fx)(1-fy), fx(1-fy), (1-fx)fy, fx*fy ; all 0 when the cell is out of range
    float wx1, wy1;              // the fractional parts the weights are made of, and whether the cell is in range: what a 12-float queue record
    int32_t in;                  // carries per plane (bil_from_rec rebuilds the four weights with the same operations)
};

// MED3 = false: the two-instruction clamp of rounds 1-3.  Only the level-1 render kernel asks for it: with the one-instruction form its 16-channel
// density taps spill 60 instead of 24 B per lane and the frame loses 0.9 % (the kernel sits at its 168-register budget; same values either way).
template <bool MED3 = true>
__device__ __forceinline__ Bil bil_setup(float u, float v, const Tex &t)
{
    float px = ((u + 1.0f) / 2.0f) * t.fw;
    float py = ((v + 1.0f) / 2.0f) * t.fh;
    float fx = floorf(px), fy = floorf(py);
    float wx1 = px - fx, wx0 = 1.0f - wx1;
    float wy1 = py - fy, wy0 = 1.0f - wy1;
    // x0 in [-1, W-1] <=> at least one of the taps x0, x0+1 can be inside; border texels are zero
    // bitwise &, not &&: the short-circuit form compiles to branches, which splits every gather stage into basic
    // blocks and makes hipcc spill hundreds of VGPRs in the shade pipeline (measured: 139 spills -> 0)
    // (x0 in range <=> the clamp leaves it alone: two compares instead of four; a NaN coordinate clamps to -1 and compares unequal.  The
    // 1-D weights of an out-of-range cell are zeroed BEFORE the four products.  The y weights are zeroed too: with only the x weights at 0
    // a NON-FINITE y coordinate (inf - floor(inf) = NaN, e.g. from a diverged gauge) would give 0 * NaN = NaN weights and poison sigma and
    // the trainer's density fetch instead of the exact-zero out-of-range result of grid_sample.)
    // (the clamp as ONE v_med3_f32: fminf(fmaxf(fx, -1), t.fw) compiles to v_max + v_min + a v_max_f32 s, s that canonicalises the scalar bound at
    // every use -- 36 instead of 12 instructions per march iteration.  Same value for every input: the median of (fx, -1, fw) is the clamp for
    // -1 <= fw, and a NaN fx gives min3 = -1 like fmaxf(NaN, -1) did.)
    float cx = MED3 ? __builtin_amdgcn_fmed3f(fx, -1.0f, t.fw) : fminf(fmaxf(fx, -1.0f), t.fw);
    float cy = MED3 ? __builtin_amdgcn_fmed3f(fy, -1.0f, t.fh) : fminf(fmaxf(fy, -1.0f), t.fh);
    bool in = (cx == fx) & (cy == fy);
#ifdef NGF_EXP_NOMASK       // TIMING EXPERIMENT (wrong pixels at the plane borders): what the in-range compares and selects cost
    in = true;
#endif
    Bil b;
    b.wx1 = wx1; b.wy1 = wy1; b.in = in ? 1 : 0;
    wx0 = in ? wx0 : 0.0f;
    wx1 = in ? wx1 : 0.0f;
    wy0 = in ? wy0 : 0.0f;
    wy1 = in ? wy1 : 0.0f;
    b.cx = (int)cx + 1;
    b.cy = (int)cy + 1;
    b.idx = (int)__umul24((unsigned)b.cy, (unsigned)t.stride) + b.cx;      // both factors < 2^24: the full-rate 24-bit multiply
    b.w00 = wx0 * wy0;
    b.w10 = wx1 * wy0;
    b.w01 = wx0 * wy1;
    b.w11 = wx1 * wy1;
    return b;
}

// A texel address as (uniform base pointer) + (32-bit unsigned offset in floats): hipcc then issues the gather as
// `global_load ... v_offset, s[base:base+1]` instead of building a 64-bit address per lane (v_ashrrev + two v_lshl_add_u64 per tap row --
// quarter-rate instructions on the pipe that binds these kernels).  A packed texture is far below 2^32 bytes (ngf_field_create checks it).
template <typename T>
__device__ __forceinline__ const T *tex_at(const float *base, uint32_t float_index)
{
    return reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + (size_t)(uint32_t)(float_index * 4u));
}

// The cell of a queued sample from what its 12-float record carries (texel index, fractional parts, in-range bit): the four weights by the very
// operations of bil_setup (1 - wx1, the two selects, four products), so they are its bits -- 8 instructions instead of ~28 per plane.
__device__ __forceinline__ Bil bil_from_rec(int32_t idx, float wx1, float wy1, bool in)
{
    Bil b;
    b.idx = idx; b.cx = 0; b.cy = 0; b.wx1 = wx1; b.wy1 = wy1; b.in = in ? 1 : 0;
    float wx0 = 1.0f - wx1;
    float wy0 = 1.0f - wy1;
    wx0 = in ? wx0 : 0.0f;
    wx1 = in ? wx1 : 0.0f;
    wy0 = in ? wy0 : 0.0f;
    wy1 = in ? wy1 : 0.0f;
    b.w00 = wx0 * wy0;
    b.w10 = wx1 * wy0;
    b.w01 = wx0 * wy1;
    b.w11 = wx1 * wy1;
    return b;
}

// ---- packed fp32 (v_pk_*_f32: two IEEE operations per lane and instruction, the same roundings as the scalar forms) ------------------------------
// Written as inline assembly: from `__builtin_elementwise_fma` on splatted weights hipcc builds the weight pairs it wants THROUGH SCRATCH (a 20-byte
// stack array per stage, 48 B per lane measured), and whether two scalars share an aligned register pair is the allocator's business otherwise.
// op_sel picks the source half for the LOW result, op_sel_hi for the HIGH result (default: low / high): a weight is broadcast to both halves by
// naming one half of its pair twice.  No packed result of these helpers feeds a bf16 matrix instruction (tests/test_isa_lint.py's fence).
__device__ __forceinline__ f32x2 pk_fma_wlo(f32x2 w, f32x2 t, f32x2 acc)          // acc + w.lo * t
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(w), "v"(t));
    return acc;
}
__device__ __forceinline__ f32x2 pk_fma_whi(f32x2 w, f32x2 t, f32x2 acc)          // acc + w.hi * t
{
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc) : "v"(w), "v"(t));
    return acc;
}
__device__ __forceinline__ f32x2 pk_mul_wlo(f32x2 w, f32x2 t)                      // w.lo * t
{
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(r) : "v"(w), "v"(t));
    return r;
}
__device__ __forceinline__ f32x2 pk_mul_whi(f32x2 w, f32x2 t)                      // w.hi * t
{
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0]" : "=v"(r) : "v"(w), "v"(t));
    return r;
}

// The four tap weights of a queued sample's cell as two register pairs, wa = {w00, w10}, wb = {w01, w11}: the operations of bil_from_rec (its
// products, hence its bits), the products two at a time.
struct BilPk {
    int32_t idx;
    f32x2 wa, wb;
};
__device__ __forceinline__ BilPk bil_from_rec_pk(int32_t idx, float wx1, float wy1, bool in)
{
    BilPk b;
    b.idx = idx;
#ifdef NGF_EXP_NOMASK
    in = true;
#endif
    float wx0 = 1.0f - wx1;
    float wy0 = 1.0f - wy1;
    const f32x2 X = {in ? wx0 : 0.0f, in ? wx1 : 0.0f}, Y = {in ? wy0 : 0.0f, in ? wy1 : 0.0f};
    b.wa = pk_mul_wlo(Y, X);          // {wx0 wy0, wx1 wy0}
    b.wb = pk_mul_whi(Y, X);          // {wx0 wy1, wx1 wy1}
    return b;
}

// the raw cell data of a 12-float queue record (three planes): the weights are rebuilt plane by plane, right before that plane's gather
struct RecCells {
    int32_t idx[3];
    float wx1[3], wy1[3];
    int32_t bits;                // in-range flag of plane p in bit 8 + p (bits 0..7: the owner ray's slot)
};

__device__ __forceinline__ float bil_mix(const Bil &b, float v00, float v10, float v01, float v11)
{
    return fmaf(b.w11, v11, fmaf(b.w01, v01, fmaf(b.w10, v10, b.w00 * v00)));
}

// sin / cos by a two-constant Cody-Waite reduction by pi/2 (both steps are FMAs) + the cephes single-precision kernels: <= 1.6 ulp of 1
// for |x| <= 2048 (checked against double precision on 4 M random arguments per decade), about half the instructions of the full-range
// sincosf.  Every positional-encoding argument of this library is below 2^10 (coordinates in [-1, 1] times at most 2^9).
__device__ __forceinline__ void sincos_small(float x, float &s, float &c)
{
    const float k = rintf(x * 0.636619772f);
    float r = fmaf(-k, 1.5707962513e+00f, x);
    r = fmaf(-k, 7.5497894159e-08f, r);
    const float z = r * r;
    const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f) * z, r, r);
    const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f), z * z, fmaf(-0.5f, z, 1.0f));
    const int q = (int)k & 3;
    const float s0 = (q & 1) ? cp : sp, c0 = (q & 1) ? sp : cp;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
}

// ---- alpha mask: sign of ATen grid_sampler_3d on a {0,1} volume (FieldBase.py:33-40, 263-267) ----
__device__ __forceinline__ int mask_bit(const MaskVol &m, int z, int y, int x)
{
    if (x < 0 || y < 0 || z < 0 || x >= m.W || y >= m.H || z >= m.D) return 0;
    size_t idx = ((size_t)z * m.H + y) * m.W + x;
    return (m.bits[idx >> 3] >> (7 - (int)(idx & 7))) & 1;
}

// One byte per sample (round 6): the 8 corner bits of the sample's cell come from `cells` in ONE gather (the packbits image cost eight dependent
// byte gathers + bounds tests per sample: 0.94 ms of the 5.25 ms S = 884 frame).  Cells with no corner set (empty space) and cells with all eight
// set (inside the object) are decided by the byte alone -- exactly: with every corner set the sum holds the product of the three weights >= 1/2
// (w0 + w1 = 1 per axis), so it is > 0 whatever the other terms round to; only boundary cells evaluate the weighted sum, with the same expression
// as before.
__device__ __forceinline__ bool mask_occupied(const MaskVol &m, const float p[3])
{
    float q[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) q[k] = (p[k] - m.a0[k]) * m.inv[k] - 1.0f;
    float ix = ((q[0] + 1.0f) / 2.0f) * (float)(m.W - 1);
    float iy = ((q[1] + 1.0f) / 2.0f) * (float)(m.H - 1);
    float iz = ((q[2] + 1.0f) / 2.0f) * (float)(m.D - 1);
    float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    // a cell with at least one corner inside the volume has its base corner in -1 .. size - 1 (NaN fails every compare)
    if (!(fx >= -1.0f && fx <= (float)(m.W - 1) && fy >= -1.0f && fy <= (float)(m.H - 1) && fz >= -1.0f && fz <= (float)(m.D - 1)))
        return false;
    const unsigned x1 = (unsigned)((int)fx + 1), y1 = (unsigned)((int)fy + 1), z1 = (unsigned)((int)fz + 1);
    const unsigned c = m.cells[(z1 * (unsigned)(m.H + 1) + y1) * (unsigned)(m.W + 1) + x1];
    if (c == 0u) return false;
    if (c == 255u) return true;
    float wx[2] = {(fx + 1.0f) - ix, ix - fx}, wy[2] = {(fy + 1.0f) - iy, iy - fy}, wz[2] = {(fz + 1.0f) - iz, iz - fz};
    float acc = 0.0f;
#pragma unroll
    for (int dz = 0; dz < 2; ++dz)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx)
                if ((c >> (dz * 4 + dy * 2 + dx)) & 1u) acc += wx[dx] * wy[dy] * wz[dz];
    return acc > 0.0f;
}

// Empty-space skipping (round 6).  True = every sample whose cell lies within B = 2^LOG cells (per axis) of the cell of p samples the mask as 0: cells outside
// the volume are empty (zeros padding), so the cell index is clamped into the blocks around the volume.  A NaN position certifies nothing.
template <int LOG>
__device__ __forceinline__ bool mask_clear_around(const MaskVol &m, const uint8_t *image, const float p[3])
{
    float q[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) q[k] = (p[k] - m.a0[k]) * m.inv[k] - 1.0f;
    const float fx = floorf(((q[0] + 1.0f) / 2.0f) * (float)(m.W - 1)), fy = floorf(((q[1] + 1.0f) / 2.0f) * (float)(m.H - 1)),
                fz = floorf(((q[2] + 1.0f) / 2.0f) * (float)(m.D - 1));
    if (!(fx == fx && fy == fy && fz == fz)) return false;
    // cell index + 16 = floor + 17, clamped to 0 .. size + 32 (beyond the clamp every neighbour within B cells is outside the volume, like those of the clamped cell)
    const int X = (int)fminf(fmaxf(fx + 17.0f, 0.0f), (float)(m.W + 32)) >> LOG, Y = (int)fminf(fmaxf(fy + 17.0f, 0.0f), (float)(m.H + 32)) >> LOG,
              Z = (int)fminf(fmaxf(fz + 17.0f, 0.0f), (float)(m.D + 32)) >> LOG;
    const int bH = ((m.H + 32) >> LOG) + 1, bW = ((m.W + 32) >> LOG) + 1;
    return image[(Z * bH + Y) * bW + X] != 0;
}

// feature2density (Field.py:48-50): F.softplus(x - 10), threshold 20.
// log1p(e) as log(t) + (e - (t - 1)) / t with t = fl(1 + e): the second term puts back what the rounding of the sum lost (exactly, for
// e <= 1).  ocml's log1pf is a double-float evaluation of ~130 VALU instructions -- a sixth of a march step of render_kernel -- for 1.6 ulp;
// this form measures 2.7 ulp worst case / 0.42 ulp mean over u in [-40, 22] against double precision (profiles/micro/softplus_accuracy.hip,
// profiles/r03_micro_softplus_accuracy.txt) at ~35 instructions, branch-free.  The reference's own softplus (ATen / SLEEF) is <= 1 ulp:
// sigma moves by < 4e-7 relative, pixels by < 1e-6.
__device__ __forceinline__ float softplus_shift(float f)
{
    const float u = f + (-10.0f);
    const float e = expf(u);
    const float t = 1.0f + e;
    const float c = e - (t - 1.0f);
    const float r = logf(t) + c * __builtin_amdgcn_rcpf(t);
    return u > 20.0f ? u : r;
}

// MFMA bookkeeping.  v_mfma_f32_32x32x2_f32 computes D[32x32] += A[32x2] * B[2x32]:
//   lane l supplies A[i = l&31][k = l>>5] and B[k = l>>5][j = l&31];
//   lane l, register r holds D[i = (r&3) + 8*(r>>2) + 4*(l>>5)][j = l&31].
// The MLP is evaluated transposed (rows = hidden units from LDS-resident weights, columns =
// samples), so a lane's 16 accumulator registers are 16 hidden activations OF ITS OWN SAMPLE and
// feed the next layer's B operand directly -- no transposition or cross-lane traffic between layers.
__device__ __forceinline__ int mfma_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

}  // namespace ngf
