// ngf_field.hip -- C ABI (include/ngf.h), TriPlane / InfoInv part: field handle, render / march / decode / alpha-mask / ray
// entry points and the training step (they share the plane packing kernels).  No torch, no CPU fallback: every entry point
// runs HIP kernels or fails.  Build: see Makefile (hipcc --offload-arch=gfx950 -O3 -ffp-contract=off).
#include <map>
#include <mutex>
#include <utility>

#include "ngf_host.hpp"
#include "ngf_infoinv.hpp"
#include "ngf_render.hpp"
#ifdef NGF_EXPERIMENTS      // libngf_hip_exp.so only (make: second target): kernels that were built, measured and lost -- specialised march / shade
                            // waves, LDS-staged texture strips -- and the tuning variants (8 / 16 waves, two steps per lane, section profile).  The
                            // product library carries only kernels that can be the default; the experiment tests load the other one.
#include "ngf_render_pc.hpp"
#include "ngf_stage.hpp"
#endif
#include "ngf_train.hpp"

using namespace ngf;

// ------------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int ngf::fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

static std::atomic<int> g_knob[ngf::KNOB_COUNT];
static const char *const g_knob_name[ngf::KNOB_COUNT] = {"tile_w", "split", "waves", "nstep", "profile", "ablate", "uv_tiles", "kernel", "stage", "poison", "grid", "xcd", "tail", "ord_rows", "ord_px", "train_dwg"};
static bool g_knob_init = [] { for (auto &k : g_knob) k.store(-1); return true; }();

int ngf::knob(int id) { return g_knob[id].load(std::memory_order_relaxed); }

hipError_t ngf::ensure_dynamic_lds(const void *kernel, size_t bytes)
{
    static std::mutex mu;
    static std::map<std::pair<int, const void *>, size_t> done;         // (device, kernel) -> largest size set so far
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    std::lock_guard<std::mutex> lock(mu);
    size_t &have = done[{dev, kernel}];
    if (have >= bytes && have != 0) return hipSuccess;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}

// ---- knob "poison": LDS / allocation poisoning (ngf_host.hpp) ---------------------------------------------------------------------
__global__ void __launch_bounds__(1024) dirty_lds_kernel(unsigned pattern, int words, int spin)
{
    extern __shared__ unsigned dirty_smem[];
    for (int i = threadIdx.x; i < words; i += blockDim.x) dirty_smem[i] = pattern;
    __syncthreads();
    // hold the CU's LDS for a moment so that the other blocks of the grid land on the other CUs (one 160 KB block per CU at a time)
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < (unsigned long long)spin) __builtin_amdgcn_s_sleep(8);
    if (dirty_smem[(threadIdx.x * 97) % words] != pattern) __builtin_trap();
}

int ngf::poison_lds(hipStream_t st)
{
    if (knob(KNOB_POISON) < 0 || !(knob(KNOB_POISON) & 1)) return NGF_OK;
    constexpr int kBytes = 160 * 1024;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDevice(&dev));
        HIP_TRY(hipGetDeviceProperties(&prop, dev));
        cus = prop.multiProcessorCount;
    }
    HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(dirty_lds_kernel), kBytes));
    hipLaunchKernelGGL(dirty_lds_kernel, dim3(2 * cus), dim3(1024), kBytes, st, kPoisonPattern, kBytes / 4, 40000);
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

int ngf::poison_alloc(void *p, size_t bytes, hipStream_t st)
{
    if (knob(KNOB_POISON) < 0 || !(knob(KNOB_POISON) & 2) || !p) return NGF_OK;
    HIP_TRY(hipMemsetD32Async((hipDeviceptr_t)p, (int)kPoisonPattern, bytes / 4, st));
    return NGF_OK;
}

extern "C" int ngf_debug_dirty_lds(void *hip_stream)
{
    const int old = g_knob[ngf::KNOB_POISON].load();
    g_knob[ngf::KNOB_POISON].store(1);
    const int rc = ngf::poison_lds((hipStream_t)hip_stream);
    g_knob[ngf::KNOB_POISON].store(old);
    return rc;
}

// where do the workgroups of a launch land?  out[x] += 1 per workgroup on XCD x (tests: every XCD id 0..7 is seen, evenly)
__global__ void xcd_histogram_kernel(unsigned *out)
{
    if (threadIdx.x == 0) atomicAdd(out + xcd_id(), 1u);
    const unsigned long long t0 = __builtin_readcyclecounter();
    while (__builtin_readcyclecounter() - t0 < 20000ull) __builtin_amdgcn_s_sleep(8);      // keep the CU busy: one workgroup per CU like the render launches
}
extern "C" int ngf_debug_xcd_histogram(unsigned *out8, int32_t workgroups, void *hip_stream)
{
    if (!out8 || workgroups <= 0) return fail(NGF_E_ARG, "ngf_debug_xcd_histogram: bad argument");
    HIP_TRY(hipMemsetAsync(out8, 0, 8 * sizeof(unsigned), (hipStream_t)hip_stream));
    hipLaunchKernelGGL(xcd_histogram_kernel, dim3(workgroups), dim3(768), 0, (hipStream_t)hip_stream, out8);
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

extern "C" int ngf_debug_set(const char *name, int32_t value)
{
    if (!name) return fail(NGF_E_ARG, "ngf_debug_set: null name");
    for (int k = 0; k < ngf::KNOB_COUNT; ++k)
        if (!strcmp(name, g_knob_name[k])) { g_knob[k].store(value); return NGF_OK; }
    return fail(NGF_E_ARG, "ngf_debug_set: unknown knob '%s'", name);
}

extern "C" int32_t ngf_debug_get(const char *name)
{
    if (name)
        for (int k = 0; k < ngf::KNOB_COUNT; ++k)
            if (!strcmp(name, g_knob_name[k])) return g_knob[k].load();
    return -1;
}

struct ngf_field {
    int32_t model = 0, flags = 0, plane_c = 0, dens_dim = 0, app = 0;
    float *tex[9] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};  // dens[3], app[3], gau[3]
    float *blob = nullptr;
    float *basis_pack = nullptr;
    double *w1pd_tmp = nullptr;               // create only: the same values as doubles (bake_color_kernel's weights)
    float *w1p_tmp = nullptr;                 // create only: W1' = W1[:, :F] . basis folded on the device (freed before create returns; here so that an error exit frees it)
    uint8_t *mask = nullptr;
    uint8_t *mask_cells = nullptr;            // the mask's corner bytes per trilinear cell (mask_cells_kernel)
    uint8_t *mask_any = nullptr, *mask_clear = nullptr, *mask_clear4 = nullptr;   // blocks of 8^3 (4^3) cells: occupied at all (scratch of the build) / nothing within 8 (4) cells (mask_block_*_kernel)
    unsigned int *counters = nullptr;
    mutable std::atomic<unsigned> next_counter{0};
    RenderArgs proto;
    int64_t bytes = 0;
    int num_cus = 256;
    std::vector<std::pair<void *, size_t>> allocs;      // every device buffer of the handle (pointer, bytes): ngf_field_destroy hands them to the pool
    // round 6: the device the handle lives on and every stream a launch that reads its buffers was put on -- ngf_field_destroy waits for
    // THOSE streams (one event each) on THAT device instead of a hipDeviceSynchronize on whatever device is current in the calling thread
    int dev = 0;
    mutable std::mutex use_mu;
    mutable std::vector<hipStream_t> streams;
    mutable std::atomic<void *> last_stream{(void *)(intptr_t)-1};
};

// every launch that reads the handle's buffers reports its stream (a pointer compare in the steady state)
static inline void field_use(const ngf_field *f, hipStream_t st)
{
    if (f->last_stream.load(std::memory_order_relaxed) == (void *)st) return;
    std::lock_guard<std::mutex> lk(f->use_mu);
    bool known = false;
    for (hipStream_t s : f->streams) known |= s == st;
    if (!known) f->streams.push_back(st);
    f->last_stream.store((void *)st, std::memory_order_relaxed);
}

// ---- packing kernels -----------------------------------------------------------------------------
// NCHW [C,H,W] channels [c0,c0+nc) -> zero-bordered channel-last [(H+2)][(W+2)][nc]
// perm = 1: the colour channels in the order of infoinv_split_channel (InfoInv NGF_F_SPLIT_BF16, nc = 72)
__global__ void pack_plane_kernel(const float *__restrict__ src, int H, int W, int c0, int nc, float *__restrict__ dst, int perm = 0)
{
    const size_t total = (size_t)(H + 2) * (W + 2) * nc;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % nc);
        const size_t tx = i / nc;
        const int x = (int)(tx % (W + 2)), y = (int)(tx / (W + 2));
        float v = 0.0f;
        if (x >= 1 && x <= W && y >= 1 && y <= H) v = src[((size_t)(c0 + (perm ? infoinv_split_channel(c) : c)) * H + (y - 1)) * W + (x - 1)];
        dst[i] = v;
    }
}

// Alpha mask, second image (round 6): per trilinear cell -- base corner (z, y, x) in -1 .. D-1 / H-1 / W-1 -- one byte with the bits of its 8 corners
// (bit dz*4 + dy*2 + dx; corners outside the volume are 0 = grid_sample's zeros padding), so that mask_occupied needs ONE gather per sample.
__global__ void __launch_bounds__(256) mask_cells_kernel(const uint8_t *__restrict__ bits, int D, int H, int W, uint8_t *__restrict__ cells)
{
    ngf::MaskVol m{};
    m.bits = bits; m.D = D; m.H = H; m.W = W;
    const size_t total = (size_t)(D + 1) * (H + 1) * (W + 1);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % (size_t)(W + 1)) - 1, y = (int)((i / (size_t)(W + 1)) % (size_t)(H + 1)) - 1, z = (int)(i / ((size_t)(W + 1) * (H + 1))) - 1;
        unsigned c = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) c |= (unsigned)ngf::mask_bit(m, z + (k >> 2), y + ((k >> 1) & 1), x + (k & 1)) << k;
        cells[i] = (uint8_t)c;
    }
}

// Alpha mask, block images (round 6): empty-space skipping.  Blocks of B^3 cells, B = 2^LOG, over the cell indices -16 .. size + 16 per axis (block = (cell index + 16) >> LOG).
// Pass 1: any[block] = some cell of the block has an occupied corner (one wave per block, a lane per (z, y) row of B cells).
template <int LOG>
__global__ void __launch_bounds__(64) mask_block_any_kernel(const uint8_t *__restrict__ cells, int D, int H, int W, int bH, int bW, uint8_t *__restrict__ any)
{
    constexpr int B = 1 << LOG;
    const int b = blockIdx.x, X = b % bW, Y = (b / bW) % bH, Z = b / (bW * bH);
    const int z = Z * B - 16 + ((int)threadIdx.x >> LOG), y = Y * B - 16 + ((int)threadIdx.x & (B - 1));
    unsigned acc = 0;
    if ((int)threadIdx.x < B * B && z >= 0 && z <= D && y >= 0 && y <= H) {
        const uint8_t *row = cells + ((size_t)z * (H + 1) + y) * (W + 1);
#pragma unroll
        for (int t = 0; t < B; ++t) {
            const int x = X * B - 16 + t;
            if (x >= 0 && x <= W) acc |= row[x];
        }
    }
    const unsigned long long m = __ballot(acc != 0);
    if (threadIdx.x == 0) any[b] = m ? 1 : 0;
}

// Pass 2: clear[block] = the block and its 26 neighbours hold nothing (blocks outside the grid are empty space).
__global__ void __launch_bounds__(256) mask_block_clear_kernel(const uint8_t *__restrict__ any, int cD, int cH, int cW, uint8_t *__restrict__ clear)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= cD * cH * cW) return;
    const int X = b % cW, Y = (b / cW) % cH, Z = b / (cW * cH);
    unsigned acc = 0;
    for (int dz = -1; dz <= 1; ++dz)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
                const int x = X + dx, y = Y + dy, z = Z + dz;
                if (x >= 0 && x < cW && y >= 0 && y < cH && z >= 0 && z < cD) acc |= any[(z * cH + y) * cW + x];
            }
    clear[b] = acc ? 0 : 1;
}

// density_decoder Linear(48,1) pre-composed with the density channels of one plane (fp64 accumulate)
__global__ void bake_density_kernel(const float *__restrict__ src, int H, int W, int nc, const float *__restrict__ wd,
                                    float *__restrict__ dst)
{
    const size_t total = (size_t)(H + 2) * (W + 2);
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % (W + 2)), y = (int)(i / (W + 2));
        double v = 0.0;
        if (x >= 1 && x <= W && y >= 1 && y <= H)
            for (int c = 0; c < nc; ++c) v += (double)wd[c] * (double)src[((size_t)c * H + (y - 1)) * W + (x - 1)];
        dst[i] = (float)v;
    }
}

// rgb_decoder layer 1 (pre-composed with basis) applied per texel: dst[(y,x)][j] = sum_c wp[j][c] * src[c0+c][y][x]
// (fp64 accumulate); wp = this plane's nc (<= 48) columns of W1' [64][ldw], as doubles (fold_w1_basis_kernel's second output: the fp32-rounded values);
// channel n of a baked texel = unit n.
// Round 5: a workgroup takes 64 consecutive padded texels (one per lane: every channel's read is one 256-byte run of the plane), wave w the outputs
// 16 w .. 16 w + 15 -- sixteen independent fp64 chains per lane whose weights are wave-uniform (scalar loads, SGPR operands of v_fma_f64).  Round 4 had one
// thread per output (48 serial loads of one address per wave): 375 us per 256^2 plane.  Same sums: a product of two floats is exact in fp64, so every
// partial sum is rounded where the one-thread-per-output loop rounded it -- bit-identical planes.
__global__ void __launch_bounds__(256) bake_color_kernel(const float *__restrict__ src, int H, int W, int c0, int nc, const double *__restrict__ wp, int ldw,
                                                         float *__restrict__ dst)
{
    const size_t texels = (size_t)(H + 2) * (W + 2), plane = (size_t)H * W;
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const double *wr = wp + (size_t)(16 * w) * ldw;            // wave-uniform
    for (size_t t0 = (size_t)blockIdx.x * 64; t0 < texels; t0 += (size_t)gridDim.x * 64) {
        const size_t tx = t0 + lane;
        if (tx >= texels) continue;
        const int x = (int)(tx % (W + 2)), y = (int)(tx / (W + 2));
        const bool inside = x >= 1 && x <= W && y >= 1 && y <= H;
        const float *sp = src + ((size_t)c0 * H + (inside ? y - 1 : 0)) * W + (inside ? x - 1 : 0);
        double acc[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[k] = 0.0;
#pragma unroll 4
        for (int c = 0; c < nc; ++c) {
            const double v = (double)sp[(size_t)c * plane];
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[k] += wr[(size_t)k * ldw + c] * v;
        }
        f32x4 *o = reinterpret_cast<f32x4 *>(dst + tx * 64 + 16 * w);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            o[q] = inside ? f32x4{(float)acc[4 * q], (float)acc[4 * q + 1], (float)acc[4 * q + 2], (float)acc[4 * q + 3]} : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }
}

// W1' = W1[:, :F] . basis, [64][F], fp64 accumulate with j ascending, rounded to fp32 once -- what every image builder below places (they used to fold
// on the host: 1.3 M (TriPlane) / 3 M (InfoInv) fp64 multiply-adds of one thread, 0.5 / 1.2 ms of every create; the same sums to the bit: a product of
// two floats is exact in fp64, one rounding per addition, -ffp-contract=off on both sides anyway).  w1p_d (may be NULL): the same fp32 values as doubles,
// what bake_color_kernel multiplies with.  Round 5.
__global__ void __launch_bounds__(256) fold_w1_basis_kernel(const float *__restrict__ w1, const float *__restrict__ basis, int F, float *__restrict__ w1p,
                                                            double *__restrict__ w1p_d)
{
    const int IN = F + 15, total = 64 * F;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int k = i % F, n = i / F;                       // lanes on consecutive k: basis rows are read in runs
        double s = 0.0;
#pragma unroll 8
        for (int j = 0; j < F; ++j) s += (double)w1[(size_t)n * IN + j] * (double)basis[(size_t)j * F + k];
        w1p[i] = (float)s;
        if (w1p_d) w1p_d[i] = (double)(float)s;
    }
}

// get_ray_directions + get_rays (ray_utils.py:24-42, 66-87; blender.py:52)
__global__ void generate_rays_kernel(int H, int W, float focal, float r00, float r01, float r02, float r10, float r11,
                                     float r12, float r20, float r21, float r22, float ox, float oy, float oz, int row0,
                                     int rows, float *__restrict__ rays)
{
    const int64_t total = (int64_t)rows * W;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(q % W), row = row0 + (int)(q / W);
        const float i = (float)col + 0.5f, j = (float)row + 0.5f;
        float dx = (i - (float)W / 2) / focal, dy = (j - (float)H / 2) / focal, dz = 1.0f;
        const float nrm = sqrtf((dx * dx + dy * dy) + dz * dz);
        dx = dx / nrm; dy = dy / nrm; dz = dz / nrm;
        float *r = rays + q * 6;
        r[0] = ox; r[1] = oy; r[2] = oz;
        r[3] = (dx * r00 + dy * r01) + dz * r02;
        r[4] = (dx * r10 + dy * r11) + dz * r12;
        r[5] = (dx * r20 + dy * r21) + dz * r22;
    }
}

// get_rays_dir on the 'no_crop' pixel grid (UV-Mapping/data/dtu.py:27-37, 160-168): integer pixel coordinates, float32 focal /
// principal point / rotation as the shipped in_cam*.npy hold them; dirs = rot^T (x, y, 1) summed in row order, / (norm + 1e-5)
__global__ void generate_rays_dtu_kernel(int W, float fx, float fy, float cx, float cy, float r00, float r01, float r02, float r10,
                                         float r11, float r12, float r20, float r21, float r22, int row0, int rows, float *__restrict__ raydir)
{
    const int64_t total = (int64_t)rows * W;
    for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < total; q += (int64_t)gridDim.x * blockDim.x) {
        const int col = (int)(q % W), row = row0 + (int)(q / W);
        const float x = ((float)col - cx) / fx, y = ((float)row - cy) / fy;
        float dx = (r00 * x + r10 * y) + r20, dy = (r01 * x + r11 * y) + r21, dz = (r02 * x + r12 * y) + r22;
        const float nrm = sqrtf((dx * dx + dy * dy) + dz * dz) + 1e-5f;
        float *r = raydir + q * 3;
        r[0] = dx / nrm; r[1] = dy / nrm; r[2] = dz / nrm;
    }
}

// ---- MLP images ------------------------------------------------------------------------------------
// LDS images of rgb_decoder: basis (no bias, no activation; networks.py:17,26) is pre-composed with layer 1 in fp64,
// W1' = W1[:, :F] @ basis; row / column permutations put every MFMA operand at [k-step][lane].
// 16-wide (v_mfma_f32_16x16x4_f32) image of rgb_decoder for TriPlane (ngf_shade16.hpp).  Lane (s, kq): hidden
// unit of accumulator (mt, r) is n = mt*16 + 4*kq + r.  With bake = true the plane part of layer 1 goes to the
// texture baker instead: wp[p][n][c] = W1'[n][p*APPc + c] (natural unit order: channel n of a baked texel = unit n).
// NGF_F_SPLIT_BF16: bf16 round-to-nearest-even and the 3-term split of a weight
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
static void split3(float x, uint16_t out[3])
{
    out[0] = f2bf(x);
    const float r1 = x - bf2f(out[0]);
    out[1] = f2bf(r1);
    const float r2 = r1 - bf2f(out[1]);
    out[2] = f2bf(r2);
}

// LDS image of ngf_shade_bf16.hpp: A fragments [mt][k-block][part][lane][8 bf16]; lane (i, kq), element e of k-block kb holds the
// weight of output unit mt*16 + i for the (kb*8 + e)-th input that lane quarter kq supplies
static void build_rgb_image_bf16(int F, const std::vector<float> &w1p, const std::vector<float> &w1, const std::vector<float> &b1,
                                 const std::vector<float> &w2, const std::vector<float> &b2, const std::vector<float> &w3, const std::vector<float> &b3,
                                 float *img)
{
    using L = MlpLayoutBf16;
    const int IN = F + 15, APPc = F / 3;
    std::vector<double> w1f((size_t)64 * (F + 16), 0.0);       // W1' = [W1[:, :F] . basis | W1[:, F:F+15] | 0]
    for (int n = 0; n < 64; ++n) {
        for (int k = 0; k < F; ++k) w1f[(size_t)n * (F + 16) + k] = (double)w1p[(size_t)n * F + k];      // W1[:, :F] . basis, folded on the device (fold_w1_basis_kernel)
        for (int k = 0; k < 15; ++k) w1f[(size_t)n * (F + 16) + F + k] = w1[(size_t)n * IN + F + k];
    }
    auto hidden = [](int mt, int r, int kq) { return mt * 16 + 4 * kq + r; };
    uint16_t *h16 = reinterpret_cast<uint16_t *>(img);
    auto put = [&](int base_floats, int nkb, int mt, int kb, int l, int e, float wv) {
        uint16_t p3[3];
        split3(wv, p3);
        for (int part = 0; part < 3; ++part)
            h16[((size_t)base_floats + ((((size_t)mt * nkb + kb) * 3 + part) * 64 + l) * 4) * 2 + e] = p3[part];
    };
    for (int mt = 0; mt < 4; ++mt)
        for (int kb = 0; kb < L::KB1; ++kb)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int kq = l >> 4, n = mt * 16 + (l & 15), j = kb * 8 + e;
                    int col;
                    if (j < 36) { const int P = j / 12, jj = j % 12; col = P * APPc + 16 * (jj / 4) + 4 * kq + (jj & 3); }
                    else col = F + kq * 4 + (j - 36);                                   // view entry (entry 15 = zero pad column)
                    put(L::W1, L::KB1, mt, kb, l, e, (float)w1f[(size_t)n * (F + 16) + col]);
                }
    for (int mt = 0; mt < 4; ++mt)
        for (int kb = 0; kb < L::KB2; ++kb)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int kq = l >> 4, n = mt * 16 + (l & 15), j = kb * 8 + e;
                    put(L::W2, L::KB2, mt, kb, l, e, w2[(size_t)n * 64 + hidden(j >> 2, j & 3, kq)]);
                }
    for (int kq = 0; kq < 4; ++kq)
        for (int k = 0; k < 16; ++k) {
            const int n = hidden(k >> 2, k & 3, kq);
            img[L::B1 + kq * 16 + k] = b1[n];
            img[L::B2 + kq * 16 + k] = b2[n];
            for (int c = 0; c < 3; ++c) img[L::W3 + c * 64 + kq * 16 + k] = w3[(size_t)c * 64 + n];
        }
    for (int c = 0; c < 3; ++c) img[L::B3 + c] = b3[c];
    img[L::B3 + 3] = 0.0f;
}

// NGF_F_BAKE_COLOR | NGF_F_SPLIT_BF16: the level-3 image (built by build_rgb_image16 with bake = true) with its fp32 layer-2 matrix replaced by the
// bf16 A fragments of build_rgb_image_bf16's layer 2 (MlpLayout16BakedBf16: the view k-steps stay where they are, the fp32 tables move behind)
static void rebuild_baked_image_bf16(const std::vector<float> &baked, const std::vector<float> &w2, float *img)
{
    using LS = MlpLayout16Baked;
    using LD = MlpLayout16BakedBf16;
    memcpy(img + LD::W1V, baked.data() + LS::W1V, sizeof(float) * 4 * 4 * 64);
    memcpy(img + LD::B1, baked.data() + LS::B1, sizeof(float) * 64);
    memcpy(img + LD::B2, baked.data() + LS::B2, sizeof(float) * 64);
    memcpy(img + LD::W3, baked.data() + LS::W3, sizeof(float) * 192);
    memcpy(img + LD::B3, baked.data() + LS::B3, sizeof(float) * 4);
    auto hidden = [](int mt, int r, int kq) { return mt * 16 + 4 * kq + r; };
    uint16_t *h16 = reinterpret_cast<uint16_t *>(img);
    for (int mt = 0; mt < 4; ++mt)
        for (int kb = 0; kb < 2; ++kb)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int kq = l >> 4, n = mt * 16 + (l & 15), j = kb * 8 + e;
                    uint16_t p3[3];
                    split3(w2[(size_t)n * 64 + hidden(j >> 2, j & 3, kq)], p3);
                    for (int part = 0; part < 3; ++part)
                        h16[((size_t)LD::W2 + ((((size_t)mt * 2 + kb) * 3 + part) * 64 + l) * 4) * 2 + e] = p3[part];
                }
}

// NGF_F_SPLIT_BF16 for InfoInv (ngf_infoinv.hpp mlp_pass16_bf16_ii): A fragments as in build_rgb_image_bf16 -- lane (i, kq), element e of
// k-block kb holds the weight of unit mt*16 + i for the lane quarter's (8 kb + e)-th input: 18 channels of each plane in the PACKED
// channel order (infoinv_split_channel), its 4 view entries, 6 zero pads.  Layer 1: hi / mid parts in the LDS image
// [mt][kb][2][lane][8 bf16], lo parts in the streamed image [kb][mt][lane][8 bf16]; layer 2: [mt][kb][3][lane][8 bf16].
static void build_rgb_image_bf16_ii(int F, const std::vector<float> &w1p, const std::vector<float> &w1, const std::vector<float> &b1,
                                    const std::vector<float> &w2, const std::vector<float> &b2, const std::vector<float> &w3, const std::vector<float> &b3,
                                    float *img, std::vector<float> &lopack)
{
    using L = MlpLayoutBf16II;
    const int IN = F + 15, APPc = F / 3;
    std::vector<double> w1f((size_t)64 * (F + 16), 0.0);       // W1' = [W1[:, :F] . basis | W1[:, F:F+15] | 0]
    for (int n = 0; n < 64; ++n) {
        for (int k = 0; k < F; ++k) w1f[(size_t)n * (F + 16) + k] = (double)w1p[(size_t)n * F + k];      // W1[:, :F] . basis, folded on the device (fold_w1_basis_kernel)
        for (int k = 0; k < 15; ++k) w1f[(size_t)n * (F + 16) + F + k] = w1[(size_t)n * IN + F + k];
    }
    auto hidden = [](int mt, int r, int kq) { return mt * 16 + 4 * kq + r; };
    uint16_t *h16 = reinterpret_cast<uint16_t *>(img);
    lopack.assign(kW1LoPackII, 0.0f);
    uint16_t *l16 = reinterpret_cast<uint16_t *>(lopack.data());
    for (int mt = 0; mt < 4; ++mt)
        for (int kb = 0; kb < L::KB1; ++kb)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int kq = l >> 4, n = mt * 16 + (l & 15), j = kb * 8 + e;
                    float wv = 0.0f;
                    if (j < 54) wv = (float)w1f[(size_t)n * (F + 16) + (j / 18) * APPc + infoinv_split_channel(kq * 18 + j % 18)];
                    else if (j < 58) wv = (float)w1f[(size_t)n * (F + 16) + F + kq * 4 + (j - 54)];      // view entry (entry 15 = zero pad column)
                    uint16_t p3[3];
                    split3(wv, p3);
                    for (int part = 0; part < 2; ++part)
                        h16[((size_t)L::W1 + ((((size_t)mt * L::KB1 + kb) * 2 + part) * 64 + l) * 4) * 2 + e] = p3[part];
                    l16[((((size_t)kb * 4 + mt) * 64 + l) * 4) * 2 + e] = p3[2];
                }
    for (int mt = 0; mt < 4; ++mt)
        for (int kb = 0; kb < L::KB2; ++kb)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 8; ++e) {
                    const int kq = l >> 4, n = mt * 16 + (l & 15), j = kb * 8 + e;
                    uint16_t p3[3];
                    split3(w2[(size_t)n * 64 + hidden(j >> 2, j & 3, kq)], p3);
                    for (int part = 0; part < 3; ++part)
                        h16[((size_t)L::W2 + ((((size_t)mt * L::KB2 + kb) * 3 + part) * 64 + l) * 4) * 2 + e] = p3[part];
                }
    for (int kq = 0; kq < 4; ++kq)
        for (int k = 0; k < 16; ++k) {
            const int n = hidden(k >> 2, k & 3, kq);
            img[L::B1 + kq * 16 + k] = b1[n];
            img[L::B2 + kq * 16 + k] = b2[n];
            for (int c = 0; c < 3; ++c) img[L::W3 + c * 64 + kq * 16 + k] = w3[(size_t)c * 64 + n];
        }
    for (int c = 0; c < 3; ++c) img[L::B3 + c] = b3[c];
    img[L::B3 + 3] = 0.0f;
}

// InfoInv default (ngf_infoinv.hpp mlp_pass16_ii): MlpLayout16<72>; k-step t of lane quarter kq is its t-th input -- 18 channels of each
// plane in the PACKED channel order (infoinv_split_channel), then its 4 view entries
static void build_rgb_image16_ii(int F, const std::vector<float> &w1p, const std::vector<float> &w1, const std::vector<float> &b1,
                                 const std::vector<float> &w2, const std::vector<float> &b2, const std::vector<float> &w3, const std::vector<float> &b3,
                                 float *img)
{
    using L = MlpLayout16<72>;
    const int IN = F + 15, APPc = F / 3;
    std::vector<double> w1f((size_t)64 * (F + 16), 0.0);       // W1' = [W1[:, :F] . basis | W1[:, F:F+15] | 0]
    for (int n = 0; n < 64; ++n) {
        for (int k = 0; k < F; ++k) w1f[(size_t)n * (F + 16) + k] = (double)w1p[(size_t)n * F + k];      // W1[:, :F] . basis, folded on the device (fold_w1_basis_kernel)
        for (int k = 0; k < 15; ++k) w1f[(size_t)n * (F + 16) + F + k] = w1[(size_t)n * IN + F + k];
    }
    auto hidden = [](int mt, int r, int kq) { return mt * 16 + 4 * kq + r; };
    for (int mt = 0; mt < 4; ++mt)
        for (int t = 0; t < L::KT; ++t)
            for (int l = 0; l < 64; ++l) {
                const int kq = l >> 4, n = mt * 16 + (l & 15);
                const int col = t < 54 ? (t / 18) * APPc + infoinv_split_channel(kq * 18 + t % 18) : F + kq * 4 + (t - 54);
                img[L::W1 + ((size_t)mt * L::KT + t) * 64 + l] = (float)w1f[(size_t)n * (F + 16) + col];
            }
    for (int mt = 0; mt < 4; ++mt)
        for (int t = 0; t < 16; ++t)
            for (int l = 0; l < 64; ++l) img[L::W2 + ((size_t)mt * 16 + t) * 64 + l] = w2[(size_t)(mt * 16 + (l & 15)) * 64 + hidden(t >> 2, t & 3, l >> 4)];
    for (int kq = 0; kq < 4; ++kq)
        for (int k = 0; k < 16; ++k) {
            const int n = hidden(k >> 2, k & 3, kq);
            img[L::B1 + kq * 16 + k] = b1[n];
            img[L::B2 + kq * 16 + k] = b2[n];
            for (int c = 0; c < 3; ++c) img[L::W3 + c * 64 + kq * 16 + k] = w3[(size_t)c * 64 + n];
        }
    for (int c = 0; c < 3; ++c) img[L::B3 + c] = b3[c];
    img[L::B3 + 3] = 0.0f;
}

// NGF_F_NO_FOLD: layer 1 un-composed, inputs in the accumulator order of the basis stage; basis packed for streaming
static void build_rgb_image16_nofold(int F, const std::vector<float> &basis, const std::vector<float> &w1, const std::vector<float> &b1,
                                     const std::vector<float> &w2, const std::vector<float> &b2, const std::vector<float> &w3, const std::vector<float> &b3,
                                     float *img, std::vector<float> &bpack)
{
    using L = MlpLayout16NoFold;
    const int IN = F + 15, APPc = F / 3, QCH = APPc / 4;
    auto hidden = [](int mt, int r, int kq) { return mt * 16 + 4 * kq + r; };
    // layer 1: k-step t < 36 takes g unit (t/4)*16 + 4kq + (t&3); t >= 36 view entry kq*4 + (t-36) (entry 15 = zero pad)
    for (int mt = 0; mt < 4; ++mt)
        for (int t = 0; t < L::KT; ++t)
            for (int l = 0; l < 64; ++l) {
                const int kq = l >> 4, n = mt * 16 + (l & 15);
                float wv = 0.0f;
                if (t < 36) wv = w1[(size_t)n * IN + hidden(t >> 2, t & 3, kq)];
                else if (kq * 4 + (t - 36) < 15) wv = w1[(size_t)n * IN + F + kq * 4 + (t - 36)];
                img[L::W1 + ((size_t)mt * L::KT + t) * 64 + l] = wv;
            }
    for (int mt = 0; mt < 4; ++mt)
        for (int t = 0; t < 16; ++t)
            for (int l = 0; l < 64; ++l)
                img[L::W2 + ((size_t)mt * 16 + t) * 64 + l] = w2[(size_t)(mt * 16 + (l & 15)) * 64 + hidden(t >> 2, t & 3, l >> 4)];
    for (int kq = 0; kq < 4; ++kq)
        for (int k = 0; k < 16; ++k) {
            const int n = hidden(k >> 2, k & 3, kq);
            img[L::B1 + kq * 16 + k] = b1[n];
            img[L::B2 + kq * 16 + k] = b2[n];
            for (int c = 0; c < 3; ++c) img[L::W3 + c * 64 + kq * 16 + k] = w3[(size_t)c * 64 + n];
        }
    for (int c = 0; c < 3; ++c) img[L::B3 + c] = b3[c];
    img[L::B3 + 3] = 0.0f;
    // basis stage: k-step t = P*12 + j takes colour channel P*48 + 16*(j/4) + 4kq + (j&3) (the gather order of mlp_pass16);
    // output unit tile mt (9 tiles), group g = mt/4, element e = mt%4
    bpack.assign(kBasisPackFloats, 0.0f);
    for (int t = 0; t < 36; ++t)
        for (int mt = 0; mt < 9; ++mt)
            for (int l = 0; l < 64; ++l) {
                const int kq = l >> 4, j = t % QCH;
                const int ch = (t / QCH) * APPc + 16 * (j / 4) + 4 * kq + (j & 3);
                bpack[(((size_t)t * 3 + mt / 4) * 64 + l) * 4 + (mt & 3)] = basis[(size_t)(mt * 16 + (l & 15)) * F + ch];
            }
}

static void build_rgb_image16(int F, bool bake, const std::vector<float> &w1p, const std::vector<float> &w1, const std::vector<float> &b1,
                              const std::vector<float> &w2, const std::vector<float> &b2, const std::vector<float> &w3,
                              const std::vector<float> &b3, float *img)
{
    const int IN = F + 15, APPc = F / 3, QCH = APPc / 4, KT = 3 * QCH + 4;
    std::vector<double> w1f((size_t)64 * (F + 16), 0.0);       // W1' = [W1[:, :F] . basis | W1[:, F:F+15] | 0]
    for (int n = 0; n < 64; ++n) {
        for (int k = 0; k < F; ++k) w1f[(size_t)n * (F + 16) + k] = (double)w1p[(size_t)n * F + k];      // W1[:, :F] . basis, folded on the device (fold_w1_basis_kernel)
        for (int k = 0; k < 15; ++k) w1f[(size_t)n * (F + 16) + F + k] = w1[(size_t)n * IN + F + k];
    }
    auto hidden = [](int mt, int r, int kq) { return mt * 16 + 4 * kq + r; };
    int oW1, oW2, oB1, oB2, oW3, oB3;
    if (bake) {
        using L = MlpLayout16Baked;
        oW1 = L::W1V; oW2 = L::W2; oB1 = L::B1; oB2 = L::B2; oW3 = L::W3; oB3 = L::B3;
        for (int mt = 0; mt < 4; ++mt)
            for (int j = 0; j < 4; ++j)
                for (int l = 0; l < 64; ++l)
                    img[oW1 + ((size_t)mt * 4 + j) * 64 + l] = (float)w1f[(size_t)(mt * 16 + (l & 15)) * (F + 16) + F + (l >> 4) * 4 + j];
    } else {
        using L = MlpLayout16<48>;
        oW1 = L::W1; oW2 = L::W2; oB1 = L::B1; oB2 = L::B2; oW3 = L::W3; oB3 = L::B3;
        auto kmap = [&](int t, int kq) {
            if (t < 3 * QCH) return (t / QCH) * APPc + 16 * ((t % QCH) / 4) + 4 * kq + ((t % QCH) & 3);   // channel 16q + 4kq + e
            return F + kq * 4 + (t - 3 * QCH);
        };
        for (int mt = 0; mt < 4; ++mt)
            for (int t = 0; t < KT; ++t)
                for (int l = 0; l < 64; ++l)
                    img[oW1 + ((size_t)mt * KT + t) * 64 + l] = (float)w1f[(size_t)(mt * 16 + (l & 15)) * (F + 16) + kmap(t, l >> 4)];
    }
    for (int mt = 0; mt < 4; ++mt)
        for (int t = 0; t < 16; ++t)
            for (int l = 0; l < 64; ++l)
                img[oW2 + ((size_t)mt * 16 + t) * 64 + l] = w2[(size_t)(mt * 16 + (l & 15)) * 64 + hidden(t >> 2, t & 3, l >> 4)];
    for (int kq = 0; kq < 4; ++kq)
        for (int k = 0; k < 16; ++k) {
            const int n = hidden(k >> 2, k & 3, kq);
            img[oB1 + kq * 16 + k] = b1[n];
            img[oB2 + kq * 16 + k] = b2[n];
            for (int c = 0; c < 3; ++c) img[oW3 + c * 64 + kq * 16 + k] = w3[(size_t)c * 64 + n];
        }
    for (int c = 0; c < 3; ++c) img[oB3 + c] = b3[c];
    img[oB3 + 3] = 0.0f;
}


// ---- buffer pool of the field handles (round 5) ---------------------------------------------------------------------------------
// A handle is rebuilt after every parameter change of an eval field (Base.handle()): same shapes, so the same ten buffer sizes.  hipFree
// synchronises the device and cost 0.7 ms per destroy -- more than the create's own work (0.5 ms) -- and hipMalloc is not free either.
// Destroyed handles therefore park their buffers here (exact-size reuse, per device; at most kPoolEntries buffers / kPoolBytes; the rest
// goes back to the driver), after ONE hipDeviceSynchronize in ngf_field_destroy (what the first hipFree used to do implicitly: kernels
// of any stream may still read the buffers).  ngf_pool_trim() returns everything to the driver.
struct PoolEntry { void *p; size_t bytes; int dev; uint64_t age; };
static std::mutex g_pool_mu;
static std::vector<PoolEntry> g_pool;
static size_t g_pool_bytes = 0;
static uint64_t g_pool_clock = 0;
static size_t g_pool_cap_bytes = (size_t)1 << 30;        // ngf_pool_set_limit; a level-3 TriPlane handle at 256^2 is 52 MB + 67 MB, a 300^2 one 160 MB
constexpr size_t kPoolEntries = 64;

static void pool_release(const std::vector<PoolEntry> &out)          // hipFree outside the lock, each on its own device
{
    for (const PoolEntry &e : out) {
        DeviceScope ds(e.dev);
        (void)hipFree(e.p);
    }
}

// entries of `dev` (or of every device: dev < 0) leave the pool, oldest first, until at most keep_bytes / keep_entries are parked
static void pool_evict_locked(std::vector<PoolEntry> &out, int dev, size_t keep_bytes, size_t keep_entries)
{
    while (!g_pool.empty() && (g_pool_bytes > keep_bytes || g_pool.size() > keep_entries)) {
        size_t oldest = g_pool.size();
        for (size_t i = 0; i < g_pool.size(); ++i)
            if ((dev < 0 || g_pool[i].dev == dev) && (oldest == g_pool.size() || g_pool[i].age < g_pool[oldest].age)) oldest = i;
        if (oldest == g_pool.size()) break;
        out.push_back(g_pool[oldest]);
        g_pool_bytes -= g_pool[oldest].bytes;
        g_pool[oldest] = g_pool.back();
        g_pool.pop_back();
    }
}

// `dev` = the device the buffer is for (the handle's); the caller has made it the current one
static hipError_t pool_malloc(void **p, size_t bytes, int dev)
{
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        for (size_t i = 0; i < g_pool.size(); ++i)
            if (g_pool[i].bytes == bytes && g_pool[i].dev == dev) {
                *p = g_pool[i].p;
                g_pool_bytes -= bytes;
                g_pool[i] = g_pool.back();
                g_pool.pop_back();
                return hipSuccess;
            }
    }
    hipError_t e = hipMalloc(p, bytes);
    if (e != hipSuccess) {
        // out of memory while this pool sits on parked buffers nobody else can see (torch's caching allocator cannot): give them all back and try once more
        (void)hipGetLastError();
        std::vector<PoolEntry> out;
        {
            std::lock_guard<std::mutex> lk(g_pool_mu);
            pool_evict_locked(out, -1, 0, 0);
        }
        if (out.empty()) return e;
        pool_release(out);
        e = hipMalloc(p, bytes);
    }
    return e;
}

static void pool_free(void *p, size_t bytes, int dev)          // the caller has made sure that nothing on device `dev` still uses p
{
    if (!p) return;
    std::vector<PoolEntry> out;
    bool parked = false;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        if (bytes <= g_pool_cap_bytes) {
            // full: the OLDEST parked buffers make room (after up_sampling / shrink the old sizes never match again -- round 5 refused the
            // new ones instead and kept the stale ones for the life of the process)
            pool_evict_locked(out, -1, g_pool_cap_bytes - bytes, kPoolEntries - 1);
            g_pool.push_back(PoolEntry{p, bytes, dev, ++g_pool_clock});
            g_pool_bytes += bytes;
            parked = true;
        }
    }
    pool_release(out);
    if (!parked) {
        DeviceScope ds(dev);
        (void)hipFree(p);
    }
}

extern "C" int ngf_pool_trim(void)
{
    std::vector<PoolEntry> out;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        pool_evict_locked(out, -1, 0, 0);
    }
    pool_release(out);
    return NGF_OK;
}

extern "C" int ngf_pool_set_limit(int64_t bytes)
{
    if (bytes < 0) return fail(NGF_E_ARG, "ngf_pool_set_limit: %lld", (long long)bytes);
    std::vector<PoolEntry> out;
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        g_pool_cap_bytes = (size_t)bytes;
        pool_evict_locked(out, -1, g_pool_cap_bytes, kPoolEntries);
    }
    pool_release(out);
    return NGF_OK;
}

extern "C" int64_t ngf_pool_bytes(int32_t device)          // parked bytes of one device (device < 0: of all) -- tests, memory reports
{
    std::lock_guard<std::mutex> lk(g_pool_mu);
    int64_t b = 0;
    for (const PoolEntry &e : g_pool)
        if (device < 0 || e.dev == device) b += (int64_t)e.bytes;
    return b;
}

static int field_alloc(ngf_field *f, void **p, size_t bytes, const char *what)
{
    if (pool_malloc(p, bytes, f->dev) != hipSuccess) { *p = nullptr; return fail(NGF_E_HIP, "hipMalloc(%s, %zu bytes) failed on device %d", what, bytes, f->dev); }
    f->allocs.emplace_back(*p, bytes);
    f->bytes += (int64_t)bytes;
    return NGF_OK;
}

static int alloc_f(float **p, size_t n, ngf_field *f, hipStream_t st)
{
    int rc = field_alloc(f, (void **)p, n * sizeof(float), "texture / image");
    if (rc) return rc;
    return poison_alloc(*p, n * sizeof(float), st);
}

// ------------------------------------------------------------------------------------------------
extern "C" int ngf_abi_version(void) { return NGF_ABI_VERSION; }
extern "C" int ngf_sizeof_field_desc(void) { return (int)sizeof(ngf_field_desc); }
extern "C" const char *ngf_last_error(void) { return g_err; }
extern "C" int64_t ngf_field_bytes(const ngf_field *f) { return f ? f->bytes : 0; }

extern "C" int ngf_field_destroy(ngf_field *f)
{
    if (!f) return NGF_OK;
    if (!f->allocs.empty()) {
        // Launches may still read the buffers: wait for the streams the handle was used on -- on the handle's device, whatever device is
        // current in the calling thread (round 5 synchronised the CURRENT device and parked the buffers under ITS id) -- and for nothing
        // else: no device-wide synchronisation.  A stream the caller has destroyed in the meantime has finished its work (hipStreamDestroy
        // drains it); if the runtime refuses it, or the event cannot be made, the device-wide wait is the fallback.
        DeviceScope ds(f->dev);
        std::vector<hipStream_t> used;
        {
            std::lock_guard<std::mutex> lk(f->use_mu);
            used = f->streams;
        }
        bool waited = true;
        hipEvent_t ev = nullptr;
        if (!used.empty()) {
            if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess) { ev = nullptr; waited = false; }
            for (size_t i = 0; waited && i < used.size(); ++i)
                waited = hipEventRecord(ev, used[i]) == hipSuccess && hipEventSynchronize(ev) == hipSuccess;
            if (ev) (void)hipEventDestroy(ev);
        }
        if (!waited) { (void)hipGetLastError(); (void)hipDeviceSynchronize(); }
        for (const auto &a : f->allocs) pool_free(a.first, a.second, f->dev);
    }
    delete f;
    return NGF_OK;
}

extern "C" int ngf_field_create(const ngf_field_desc *d, ngf_field **out, void *hip_stream)
{
    if (!d || !out) return fail(NGF_E_ARG, "ngf_field_create: null argument");
    *out = nullptr;
    const bool tri = d->model == NGF_MODEL_TRIPLANE;
    if (!tri && d->model != NGF_MODEL_INFOINV) return fail(NGF_E_ARG, "unknown model %d", d->model);
    if (tri && (d->plane_c != 64 || d->dens_dim != 16)) return fail(NGF_E_UNSUPPORTED, "TriPlane expects 64-channel planes with 16 density channels");
    if (!tri && (d->plane_c != 96 || d->dens_dim != 24)) return fail(NGF_E_UNSUPPORTED, "InfoInv expects 96-channel planes with 24 density channels");
    for (int p = 0; p < 3; ++p) {
        if (!d->plane[p] || d->plane_h[p] < 2 || d->plane_w[p] < 2) return fail(NGF_E_ARG, "plane %d missing or smaller than 2x2", p);
        if (tri && (!d->gauge[p] || d->gauge_h[p] < 2 || d->gauge_w[p] < 2)) return fail(NGF_E_ARG, "gauge plane %d missing or smaller than 2x2", p);
    }
    if (!(d->step > 0.0f)) return fail(NGF_E_ARG, "step must be > 0");
    hipStream_t st = (hipStream_t)hip_stream;
#ifdef NGF_EXP_CREATE_TIMES                    // experiment build (profiles/exp_create_phases.py): the phases of a create on stderr
    auto tc0 = std::chrono::steady_clock::now();
#define NGF_CT(label) do { auto t1 = std::chrono::steady_clock::now(); fprintf(stderr, "create %-28s %8.1f us\n", label, std::chrono::duration<double, std::micro>(t1 - tc0).count()); tc0 = t1; } while (0)
#else
#define NGF_CT(label) do { } while (0)
#endif
    ngf_field *f = new (std::nothrow) ngf_field();
    if (!f) return fail(NGF_E_HIP, "out of host memory");
    f->model = d->model; f->flags = d->flags; f->plane_c = d->plane_c; f->dens_dim = d->dens_dim;
    f->app = d->plane_c - d->dens_dim;
    const int F = 3 * f->app;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) f->num_cus = prop.multiProcessorCount;
    f->dev = dev;                   // the caller's current device: where the parameter tensors and the stream live
    field_use(f, st);               // the packing kernels below write the handle's buffers on this stream

    int rc = NGF_OK;
    RenderArgs &A = f->proto;
    memset(&A, 0, sizeof(A));
    auto bail = [&](int code) { ngf_field_destroy(f); return code; };

    const bool bake = tri && (d->flags & NGF_F_BAKE_DENSITY);
    const bool bake_c = tri && (d->flags & NGF_F_BAKE_COLOR);
    const bool no_fold = tri && (d->flags & NGF_F_NO_FOLD);
    const bool split_bf16 = tri && (d->flags & NGF_F_SPLIT_BF16);
    const bool split_ii = !tri && (d->flags & NGF_F_SPLIT_BF16);
    if (no_fold && (bake || bake_c)) return bail(fail(NGF_E_ARG, "NGF_F_NO_FOLD is the un-composed formulation: it excludes the NGF_F_BAKE_* flags"));
    if (split_bf16 && no_fold) return bail(fail(NGF_E_ARG, "NGF_F_SPLIT_BF16 applies to the pre-composed formulations (not with NGF_F_NO_FOLD)"));
    if (split_bf16 && bake_c && !bake) return bail(fail(NGF_E_ARG, "NGF_F_BAKE_COLOR | NGF_F_SPLIT_BF16 (level 3 with layer 2 on the bf16 matrix pipe) is built on top of NGF_F_BAKE_DENSITY"));
    const bool split_l3 = split_bf16 && bake_c;        // round 5: level 3, layer 2 as split bf16 products (layer 1 is folded into the planes there)

    // MLP weights: to the host once, pre-compose, permute, back to HBM as one LDS image
    // W1' = W1[:, :F] . basis comes folded from the device (fold_w1_basis_kernel) for every formulation that pre-composes it; level 0 streams basis itself
    std::vector<float> basis, w1p, w1, b1, w2, b2, w3, b3;
    if (!d->basis || !d->w1) return bail(fail(NGF_E_ARG, "missing weight tensor"));
    if (no_fold) {
        if ((rc = d2h(basis, d->basis, (size_t)F * F, st))) return bail(rc);
    } else {
        if ((rc = field_alloc(f, (void **)&f->w1p_tmp, (size_t)64 * F * sizeof(float), "W1'"))) return bail(rc);
        if (bake_c && (rc = field_alloc(f, (void **)&f->w1pd_tmp, (size_t)64 * F * sizeof(double), "W1' (fp64)"))) return bail(rc);
        fold_w1_basis_kernel<<<(64 * F + 255) / 256, 256, 0, st>>>(d->w1, d->basis, F, f->w1p_tmp, f->w1pd_tmp);
        if ((rc = d2h(w1p, f->w1p_tmp, (size_t)64 * F, st))) return bail(rc);
    }
    if ((rc = d2h(w1, d->w1, (size_t)64 * (F + 15), st)) ||
        (rc = d2h(b1, d->b1, 64, st)) || (rc = d2h(w2, d->w2, 64 * 64, st)) || (rc = d2h(b2, d->b2, 64, st)) ||
        (rc = d2h(w3, d->w3, 3 * 64, st)) || (rc = d2h(b3, d->b3, 3, st)))
        return bail(rc);
    std::vector<float> dw1, db1, dw2, db2, dw3, db3;
    if (tri) {
        if ((rc = d2h(dw1, d->dens_w1, 48, st)) || (rc = d2h(db1, d->dens_b1, 1, st))) return bail(rc);
    } else {
        if ((rc = d2h(dw1, d->dens_w1, 32 * 72, st)) || (rc = d2h(db1, d->dens_b1, 32, st)) ||
            (rc = d2h(dw2, d->dens_w2, 32 * 32, st)) || (rc = d2h(db2, d->dens_b2, 32, st)) ||
            (rc = d2h(dw3, d->dens_w3, 32, st)) || (rc = d2h(db3, d->dens_b3, 1, st)))
            return bail(rc);
    }
    if (hipStreamSynchronize(st) != hipSuccess) return bail(fail(NGF_E_HIP, "hipStreamSynchronize failed in ngf_field_create"));
    NGF_CT("fold kernel + D2H + sync");

    const int rgb_floats = tri ? (split_l3 ? MlpLayout16BakedBf16::TOTAL : split_bf16 ? MlpLayoutBf16::TOTAL : no_fold ? MlpLayout16NoFold::TOTAL : (bake_c ? MlpLayout16Baked::TOTAL : MlpLayout16<48>::TOTAL)) : (split_ii ? MlpLayoutBf16II::TOTAL : MlpLayout16<72>::TOTAL);
    const int dens_floats = tri ? 0 : (split_ii ? InfoInvDensLayoutBf16::TOTAL : InfoInvDensLayout::TOTAL);
    std::vector<float> img((size_t)rgb_floats + dens_floats, 0.0f);
    std::vector<float> bpack;
    if (split_l3) {
        std::vector<float> baked((size_t)MlpLayout16Baked::TOTAL, 0.0f);
        build_rgb_image16(F, true, w1p, w1, b1, w2, b2, w3, b3, baked.data());
        rebuild_baked_image_bf16(baked, w2, img.data());
    } else if (split_bf16) build_rgb_image_bf16(F, w1p, w1, b1, w2, b2, w3, b3, img.data());
    else if (no_fold) build_rgb_image16_nofold(F, basis, w1, b1, w2, b2, w3, b3, img.data(), bpack);
    else if (tri) build_rgb_image16(F, bake_c, w1p, w1, b1, w2, b2, w3, b3, img.data());
    else if (split_ii) build_rgb_image_bf16_ii(F, w1p, w1, b1, w2, b2, w3, b3, img.data(), bpack);
    else build_rgb_image16_ii(F, w1p, w1, b1, w2, b2, w3, b3, img.data());
    if (!tri && split_ii) build_infoinv_density_image_bf16(dw1, db1, dw2, db2, dw3, db3, img.data() + rgb_floats);
    else if (!tri) build_infoinv_density_image(dw1, db1, dw2, db2, dw3, db3, img.data() + rgb_floats);
    NGF_CT("host images");
    if ((rc = alloc_f(&f->blob, img.size(), f, st))) return bail(rc);
    if (hipMemcpyAsync(f->blob, img.data(), img.size() * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess)
        return bail(fail(NGF_E_HIP, "uploading the MLP image failed"));
    A.blob = f->blob;
    A.blob_floats = (int)img.size();
    if (no_fold || split_ii) {        // the matrix the shade streams from L2 (level-0 basis / lo parts of InfoInv's split layer 1)
        if ((rc = alloc_f(&f->basis_pack, bpack.size(), f, st))) return bail(rc);
        if (hipMemcpyAsync(f->basis_pack, bpack.data(), bpack.size() * sizeof(float), hipMemcpyHostToDevice, st) != hipSuccess)
            return bail(fail(NGF_E_HIP, "uploading the packed basis matrix failed"));
        A.basis_pack = f->basis_pack;
    }
    if (tri) {
        for (int i = 0; i < 48; ++i) A.wd[i] = dw1[i];
        A.bd = db1[0];
    }

    NGF_CT("image upload");
    // textures: channel-last, zero-bordered
    const int app_c = bake_c ? 64 : f->app;
    for (int p = 0; p < 3; ++p) {
        const int H = d->plane_h[p], W = d->plane_w[p];
        const size_t texels = (size_t)(H + 2) * (W + 2);
        const int dc = bake ? 1 : d->dens_dim;
        if (texels * (size_t)(app_c > 96 ? app_c : 96) * sizeof(float) >= ((size_t)1 << 32)) {     // the kernels address a texture with 32-bit byte offsets (tex_at)
            return bail(fail(NGF_E_UNSUPPORTED, "plane %d: %d x %d texels do not fit a 4 GiB packed texture", p, H, W));
        }
        if ((rc = alloc_f(&f->tex[p], texels * dc, f, st)) || (rc = alloc_f(&f->tex[3 + p], texels * app_c, f, st))) return bail(rc);
        if (bake) bake_density_kernel<<<1024, 256, 0, st>>>(d->plane[p], H, W, d->dens_dim, d->dens_w1 + p * d->dens_dim, f->tex[p]);
        else pack_plane_kernel<<<2048, 256, 0, st>>>(d->plane[p], H, W, 0, d->dens_dim, f->tex[p]);
        if (bake_c) bake_color_kernel<<<2048, 256, 0, st>>>(d->plane[p], H, W, d->dens_dim, f->app, f->w1pd_tmp + (size_t)p * f->app, F, f->tex[3 + p]);
        else pack_plane_kernel<<<2048, 256, 0, st>>>(d->plane[p], H, W, d->dens_dim, f->app, f->tex[3 + p], tri ? 0 : 1);
        A.dens[p] = Tex{f->tex[p], W, H, W + 2, (float)(W - 1), (float)(H - 1)};
        A.app[p] = Tex{f->tex[3 + p], W, H, W + 2, (float)(W - 1), (float)(H - 1)};
        if (tri) {
            const int gh = d->gauge_h[p], gw = d->gauge_w[p];
            if ((rc = alloc_f(&f->tex[6 + p], (size_t)(gh + 2) * (gw + 2) * 2, f, st))) return bail(rc);
            pack_plane_kernel<<<256, 256, 0, st>>>(d->gauge[p], gh, gw, 0, 2, f->tex[6 + p]);
            A.gau[p] = Tex{f->tex[6 + p], gw, gh, gw + 2, (float)(gw - 1), (float)(gh - 1)};
        }
    }
    NGF_CT("texture allocs + launches");
    const bool launch_ok = hipGetLastError() == hipSuccess && hipStreamSynchronize(st) == hipSuccess;
    NGF_CT("sync (pack / bake kernels)");
    f->w1p_tmp = nullptr; f->w1pd_tmp = nullptr;          // (the buffers stay in f->allocs: it goes to the pool with the handle -- the next create of these shapes takes it from there)
    if (!launch_ok) return bail(fail(NGF_E_HIP, "packing kernels failed"));

    for (int k = 0; k < 3; ++k) {
        A.a0[k] = d->aabb[k];
        A.a1[k] = d->aabb[3 + k];
        A.inv[k] = 2.0f / (d->aabb[3 + k] - d->aabb[k]);      // invaabbSize (FieldBase.py:67)
    }
    A.near_ = d->near_; A.far_ = d->far_; A.step = d->step; A.dscale = d->distance_scale; A.thr = d->weight_thres;
    if (d->mask_bits) {
        const size_t nbytes = ((size_t)d->mask_d * d->mask_h * d->mask_w + 7) / 8;
        if ((rc = field_alloc(f, (void **)&f->mask, nbytes, "mask"))) return bail(rc);
        if (hipMemcpyAsync(f->mask, d->mask_bits, nbytes, hipMemcpyDeviceToDevice, st) != hipSuccess)
            return bail(fail(NGF_E_HIP, "copying the alpha mask failed"));
        A.mask.bits = f->mask;
        A.mask.D = d->mask_d; A.mask.H = d->mask_h; A.mask.W = d->mask_w;
        const size_t ncells = (size_t)(d->mask_d + 1) * (d->mask_h + 1) * (d->mask_w + 1);
        if (ncells >= ((size_t)1 << 32)) return bail(fail(NGF_E_ARG, "alpha mask of %d x %d x %d cells: the cell image is indexed with 32 bits", d->mask_d, d->mask_h, d->mask_w));
        if ((rc = field_alloc(f, (void **)&f->mask_cells, ncells, "mask cells"))) return bail(rc);
        hipLaunchKernelGGL(mask_cells_kernel, dim3(2048), dim3(256), 0, st, (const uint8_t *)f->mask, d->mask_d, d->mask_h, d->mask_w, f->mask_cells);
        if (hipGetLastError() != hipSuccess) return bail(fail(NGF_E_HIP, "mask_cells_kernel failed to launch"));
        A.mask.cells = f->mask_cells;
        const int cD = ((d->mask_d + 32) >> 3) + 1, cH = ((d->mask_h + 32) >> 3) + 1, cW = ((d->mask_w + 32) >> 3) + 1;
        const int fD = ((d->mask_d + 32) >> 2) + 1, fH = ((d->mask_h + 32) >> 2) + 1, fW = ((d->mask_w + 32) >> 2) + 1;
        if ((rc = field_alloc(f, (void **)&f->mask_any, (size_t)fD * fH * fW, "mask blocks"))) return bail(rc);          // scratch of both builds (the finer image is the larger one)
        if ((rc = field_alloc(f, (void **)&f->mask_clear, (size_t)cD * cH * cW, "mask blocks"))) return bail(rc);
        if ((rc = field_alloc(f, (void **)&f->mask_clear4, (size_t)fD * fH * fW, "mask blocks"))) return bail(rc);
        hipLaunchKernelGGL(mask_block_any_kernel<3>, dim3((unsigned)(cD * cH * cW)), dim3(64), 0, st, (const uint8_t *)f->mask_cells, d->mask_d, d->mask_h, d->mask_w, cH, cW, f->mask_any);
        hipLaunchKernelGGL(mask_block_clear_kernel, dim3((unsigned)((cD * cH * cW + 255) / 256)), dim3(256), 0, st, (const uint8_t *)f->mask_any, cD, cH, cW, f->mask_clear);
        hipLaunchKernelGGL(mask_block_any_kernel<2>, dim3((unsigned)(fD * fH * fW)), dim3(64), 0, st, (const uint8_t *)f->mask_cells, d->mask_d, d->mask_h, d->mask_w, fH, fW, f->mask_any);
        hipLaunchKernelGGL(mask_block_clear_kernel, dim3((unsigned)((fD * fH * fW + 255) / 256)), dim3(256), 0, st, (const uint8_t *)f->mask_any, fD, fH, fW, f->mask_clear4);
        if (hipGetLastError() != hipSuccess) return bail(fail(NGF_E_HIP, "mask_block kernels failed to launch"));
        A.mask.fine = f->mask_clear4;
        A.mask.coarse = f->mask_clear; A.mask.cD = cD; A.mask.cH = cH; A.mask.cW = cW;
        for (int k = 0; k < 3; ++k) {
            A.mask.a0[k] = d->mask_aabb[k];
            A.mask.inv[k] = 1.0f / (d->mask_aabb[3 + k] - d->mask_aabb[k]) * 2;   // invgridSize (FieldBase.py:29)
        }
    }
    if ((rc = field_alloc(f, (void **)&f->counters, (size_t)kCounters * kQueueHeads * sizeof(unsigned), "counters"))) return bail(rc);
    if (hipMemsetAsync(f->counters, 0, (size_t)kCounters * kQueueHeads * sizeof(unsigned), st) != hipSuccess) return bail(fail(NGF_E_HIP, "zeroing the queue heads failed"));
    if (hipStreamSynchronize(st) != hipSuccess) return bail(fail(NGF_E_HIP, "packing failed: %s", hipGetErrorString(hipGetLastError())));
    NGF_CT("mask + counters + sync");
    *out = f;
    return NGF_OK;
}

// The tile plan of a render launch (see launch_render): n rays, widest tile `wide`, `resident` waves in the persistent grid, tail16 / 16 narrow
// tiles per resident wave of each narrower width.  Fills up to four (rays, log2 width) segments in ray order, returns their number.  Every
// segment but the last holds whole tiles.  Pure host arithmetic (ngf_debug_tile_plan exposes it to the CPU tests).
static constexpr int kTailDefault16 = 16;
static int make_tile_plan(int64_t n, int wide, int64_t resident, int tail16, int64_t seg_rays[4], int seg_shift[4])
{
    auto shift_of = [](int w) { int sft = 0; while ((1 << sft) < w) ++sft; return sft; };
    // A launch that fits the persistent grid with ONE tile per wave (n <= resident x wide): the narrowest pair of widths (w, w / 2) that does it, as
    // many w / 2-ray tiles as the wave count allows -- the launch then lasts one w-ray tile.  (Round 5.  The tail rule below gave a 4096-ray chunk on
    // 3072 waves 512 two-ray and 3072 one-ray tiles: 512 waves took a second tile behind their first, 0.114 ms; 1024 two-ray + 2048 one-ray tiles are
    // one tile per wave, profiles/r05_shard_latency.txt.)  Knob tail = 0 keeps its meaning: wide tiles only.
    if (tail16 > 0 && resident > 0 && n > 0 && n <= resident * wide) {
        int w = 1;
        while ((int64_t)w * resident < n) w <<= 1;                  // smallest width whose tiles cover n rays with <= resident tiles
        if (w == 1) { seg_rays[0] = n; seg_shift[0] = 0; return 1; }
        const int a = w >> 1;
        auto allowed = [&](int v) { return v == wide || v == 4 || v == 2 || v == 1; };      // the widths the kernels' tile plans are tested with
        if (allowed(w) && allowed(a)) {
        // x tiles of w rays first, then tiles of a rays: x + y <= resident, w x + a y >= n  ->  x = ceil((n - a resident) / a)
        int64_t x = (n - (int64_t)a * resident + a - 1) / a;
        if (x < 0) x = 0;
        if (x * w > n) x = n / w;
        const int64_t rw = x * w, ra = n - rw;
        int nseg = 0;
        if (rw > 0) { seg_rays[nseg] = rw; seg_shift[nseg] = shift_of(w); ++nseg; }
        if (ra > 0) { seg_rays[nseg] = ra; seg_shift[nseg] = shift_of(a); ++nseg; }
        return nseg;
        }
    }
    const int widths[4] = {wide, 4, 2, 1};
    int64_t left = n, want[4] = {0, 0, 0, 0};
    for (int k = 3; k >= 1; --k) {      // the narrow segments are sized from the END of the ray list
        if (widths[k] >= wide) continue;
        // tail16 < 256: the same count for every narrow width; >= 256 (sweeps): one byte per width -- bits 0-7 one-ray, 8-15 two-ray, 16-23 four-ray tiles
        const int tk = tail16 < 256 ? tail16 : (tail16 >> (8 * (3 - k))) & 0xff;
        want[k] = std::min<int64_t>(left, (resident * tk / 16) * widths[k]);
        left -= want[k];
    }
    want[0] = left;
    int nseg = 0;
    int64_t carry = 0;
    for (int k = 0; k < 4; ++k) {       // what does not fill a tile moves to the next narrower segment
        int64_t r = want[k] + carry;
        carry = 0;
        if (k < 3) { carry = r % widths[k]; r -= carry; }
        if (r <= 0) continue;
        if (nseg > 0 && seg_shift[nseg - 1] == shift_of(widths[k])) { seg_rays[nseg - 1] += r; continue; }
        seg_rays[nseg] = r; seg_shift[nseg] = shift_of(widths[k]); ++nseg;
    }
    if (nseg == 0) { seg_rays[0] = n; seg_shift[0] = 0; nseg = 1; }
    return nseg;
}

// the device's queue-position -> tile map of ngf_field_render_image, on the host (CPU test: a bijection of [0, ord_n) for every plan)
extern "C" int ngf_debug_tile_order(const uint32_t *q, int64_t count, uint32_t ord_n, uint32_t tpr, uint32_t bw, uint32_t bh, uint32_t *out)
{
    if (!q || !out || count < 0 || !tpr || !bw || !bh) return fail(NGF_E_ARG, "ngf_debug_tile_order: bad argument");
    for (int64_t i = 0; i < count; ++i) out[i] = tile_order(q[i], ord_n, tpr, bw, bh);
    return NGF_OK;
}

extern "C" int ngf_debug_tile_plan(int64_t n, int32_t wide, int64_t resident, int32_t tail16, int64_t *seg_rays, int32_t *seg_shift)
{
    int64_t r[4] = {0, 0, 0, 0};
    int sh[4] = {0, 0, 0, 0};
    const int nseg = make_tile_plan(n, wide, resident, tail16 >= 0 ? tail16 : kTailDefault16, r, sh);
    for (int k = 0; k < 4; ++k) { seg_rays[k] = k < nseg ? r[k] : 0; seg_shift[k] = k < nseg ? sh[k] : 0; }
    return nseg;
}

// ---- launches ------------------------------------------------------------------------------------
// kernel / kernel_split: the DBG = true instantiations (every feature); kernel_prod: the split kernel's production instantiation (no debug
// outputs, ablation bits, statistics: render_kernel<P, true, false>) or null when the policy has none
template <typename K>
static int launch_render(K kernel, K kernel_split, K kernel_prod, const ngf_field *f, RenderArgs &A, int threads, size_t lds_bytes, hipStream_t st, int wide_tile)
{
    // the slot's queue heads are zero: ngf_field_create zeroed all slots, and the last wave of every launch zeroes its slot again (queue_done)
    const unsigned slot = f->next_counter.fetch_add(1) % kCounters;
    A.tile_counter = f->counters + (size_t)slot * kQueueHeads;
    if (int rc = poison_lds(st)) return rc;
    // Split march (render_kernel<P, true>): a tile holds tile_w rays and every ray is marched by 64 / tile_w lanes on
    // consecutive steps (bit-identical results).  Small tiles shorten the critical path of a tile and even out the
    // tiles-per-wave quantisation, which bounds small launches (one rank's shard of a frame, the reference's 4096-ray
    // chunks) and still buys 5 % on a full frame.  Measured in profiles/r01_split_march.txt: tile_w = 8 is best from
    // 160 000 rays to the full frame, tile_w = 4 below (80 000 rays: 1.43-1.49 vs 1.51-1.53 ms; 4000 rays: 0.29 vs 0.51 ms).  ngf_debug_set("tile_w" / "split") override for experiments.
    const int waves = threads / kWave;
    const int cus = knob(KNOB_GRID) > 0 && knob(KNOB_GRID) < f->num_cus ? knob(KNOB_GRID) : f->num_cus;      // knob grid (tests): fewer workgroups
    const int64_t resident = (int64_t)cus * waves;      // waves of the persistent grid
    // ---- tile plan -------------------------------------------------------------------------------------------------------------------
    // Wide tiles are the efficient ones (fewer partial shade passes, finer early termination); but the grid is persistent and a launch ends
    // when its LAST wave is done: with one tile width the waves ran dry one tile duration apart -- ~0.17 ms of every launch, 15 % of one
    // rank's 80 000-ray shard (profiles/r04_timeline.txt).  So the ray list is cut into segments of decreasing width: the widest tiles for the
    // bulk, then about `tail` tiles per resident wave of each narrower width down to one ray per tile (a 1-ray tile is ~1/8 of an 8-ray one).
    // A launch with fewer rays than that starts further down the list: a 4096-ray chunk is all 2-ray tiles on 2048 waves.
    // Knobs: tile_w forces ONE width (bit-identity tests, A/B timing); tail = 16 x the narrow tiles per wave and width (0: none).
    int64_t seg_rays[4] = {0, 0, 0, 0};
    int seg_shift[4] = {0, 0, 0, 0};
    int nseg = 1;
    bool split = kernel_split != nullptr;
    if (knob(KNOB_SPLIT) >= 0) split = knob(KNOB_SPLIT) != 0 && kernel_split;
    const int forced = knob(KNOB_TILE_W);
    auto shift_of = [](int w) { int sft = 0; while ((1 << sft) < w) ++sft; return sft; };
    if (forced >= 0 || !split) {
        int tw = forced >= 0 ? forced : 64;
        if (tw != 64 && tw != 32 && tw != 16 && tw != 8 && tw != 4 && tw != 2 && tw != 1) return fail(NGF_E_ARG, "knob tile_w must be 64, 32, 16, 8, 4, 2 or 1");
        if (knob(KNOB_SPLIT) < 0) split = tw < 64 && kernel_split;
        if (tw < 4 && !split) return fail(NGF_E_ARG, "tiles of 2 or 1 rays exist in the split kernel only");
        seg_rays[0] = A.n; seg_shift[0] = shift_of(tw);
    } else {
        nseg = make_tile_plan(A.n, wide_tile, resident, knob(KNOB_TAIL) >= 0 ? knob(KNOB_TAIL) : kTailDefault16, seg_rays, seg_shift);
    }
    A.tile_shift = seg_shift[0];
    A.tile_w = 1 << seg_shift[0];
    int64_t tiles = 0, ray0 = 0;
    for (int k = 0; k < 4; ++k) {
        const int sft = k < nseg ? seg_shift[k] : seg_shift[nseg - 1];
        const int64_t r = k < nseg ? seg_rays[k] : 0;
        A.seg_shift[k] = sft;
        A.seg_ray0[k] = ray0;
        tiles += (r + (1 << sft) - 1) >> sft;
        ray0 += r;
        if (k < 3) {
            if (tiles >= ((int64_t)1 << 32)) return fail(NGF_E_ARG, "render launch of %lld tiles: split the ray list", (long long)tiles);
            A.seg_end[k] = (uint32_t)tiles;
        }
    }
    // Screen-space tile order (ngf_field_render_image): the caller declared the list an image of A.ord_tpr (= row_width here) rays per row.  Blocks of
    // 80 rows x 80 pixels measured best on the MLP-stress frames (profiles/r06_r2_locality.txt: R2 -3.8 %, R2 at S = 884 -6.1 %; R1 and InfoInv
    // within 0.6 %); only whole rows of the widest segment take part, the plan's narrow tail keeps the list's order.
    {
        const int64_t row_w = A.ord_tpr;
        A.ord_n = A.ord_tpr = A.ord_bw = A.ord_bh = 0;
        const int sft0 = seg_shift[0];
        if (split && row_w > 0 && (row_w & ((1 << sft0) - 1)) == 0 && (seg_rays[0] >> sft0) < ((int64_t)1 << 31)) {
            const int64_t tpr = row_w >> sft0, rows = seg_rays[0] / row_w;
            int64_t bh = knob(KNOB_ORD_ROWS) > 0 ? knob(KNOB_ORD_ROWS) : 80;
            int64_t bw = (knob(KNOB_ORD_PX) > 0 ? knob(KNOB_ORD_PX) : 80) >> sft0;
            if (bw < 1) bw = 1;
            if (bw > tpr) bw = tpr;
            if (knob(KNOB_ORD_ROWS) != 0 && tpr > bw && rows >= 2 && bh >= 2) {          // knob ord_rows = 0: the list's order
                if (bh > rows) bh = rows;
                A.ord_tpr = (uint32_t)tpr; A.ord_bw = (uint32_t)bw; A.ord_bh = (uint32_t)bh;
                A.ord_n = (uint32_t)((rows / bh) * bh * tpr);
            }
        }
    }
    K k = split ? kernel_split : kernel;
#if defined(NGF_EXP_DUMP) || defined(NGF_EXP_TIMELINE)      // experiment builds: `stats` is the dump / timeline buffer of the production kernel
    if (split && kernel_prod) k = kernel_prod;
#else
    if (split && kernel_prod && !A.dbg_weight && !A.dbg_sigma && !A.stats && !A.skip_rgb && !A.ablate) k = kernel_prod;
#endif
    HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k), lds_bytes));
    if (tiles >= ((int64_t)1 << 32)) return fail(NGF_E_ARG, "render launch of %lld tiles: split the ray list", (long long)tiles);
    A.tiles = (uint32_t)tiles;
    // One tile queue per XCD (knob "xcd" = 1) is built, bit-identical and OFF by default: it cuts the fabric traffic of the MLP-stress frame
    // (R2: profiles/r03_triplane_R2_bd_xcd{0,1}_pmc.txt) but not its time -- these launches are bound by the SIMDs' own MFMA + VALU
    // cycles, not by L2 misses (R1 / R2 / InfoInv within +-0.2 %) -- and the march-only frame loses 8 % to the stealing tail
    // (profiles/r03_xcd_queues.txt).
    A.xcd_queues = knob(KNOB_XCD) > 0 ? 8 : 1;
    // One workgroup per CU as long as there are tiles, and only as many working waves per workgroup as the tiles need: a 4096-ray chunk (1024
    // tiles) used to fill 86 CUs with 12 waves each -- three per SIMD, sharing its matrix pipe -- while 170 CUs idled.
    int64_t grid = tiles < f->num_cus ? tiles : f->num_cus;
    if (knob(KNOB_GRID) > 0 && grid > knob(KNOB_GRID)) grid = knob(KNOB_GRID);      // tests: fewer workgroups -> every wave takes many tiles
    if (grid < 1) grid = 1;
    const int64_t per_wg = (tiles + grid - 1) / grid;
    A.waves_active = per_wg < waves ? (int)per_wg : waves;
    A.queue_waves = (uint32_t)grid;          // one report per workgroup (queue_done)
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(threads), lds_bytes, st, A);
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

template <typename P>
static int launch_policy(const ngf_field *f, RenderArgs &A, hipStream_t st)
{
    const size_t lds = ((size_t)((A.blob_floats + 3) & ~3) + P::WAVES * wave_lds_floats<P>()) * sizeof(float);
    if (lds > 160 * 1024) return fail(NGF_E_ARG, "this waves-per-CU setting needs %zu bytes of LDS (> 160 KiB)", lds);
    constexpr int wide = P::INFOINV ? 16 : 8;          // measured best full-frame tile width (profiles/r01_split_march.txt; InfoInv: 30.3 vs 29.3 Mray/s)
    using KP = decltype(&render_kernel<P, false>);
    if constexpr (P::NSTEP == 1 && P::PROD) return launch_render<KP>(render_kernel<P, false>, render_kernel<P, true>, render_kernel<P, true, false>, f, A, P::WAVES * kWave, lds, st, wide);
    else if constexpr (P::NSTEP == 1) return launch_render<KP>(render_kernel<P, false>, render_kernel<P, true>, nullptr, f, A, P::WAVES * kWave, lds, st, wide);
    else return launch_render<KP>(render_kernel<P, false>, nullptr, nullptr, f, A, P::WAVES * kWave, lds, st, wide);
}

#ifdef NGF_EXPERIMENTS
// Specialised march / shade waves (ngf_render_pc.hpp): NM march waves + NS shade waves per CU.
template <typename P, int NM, int NS, int TW>
static int launch_pc(const ngf_field *f, RenderArgs &A, hipStream_t st)
{
    const size_t lds = ((size_t)((A.blob_floats + 3) & ~3) + PcLds<NM>::TOTAL) * sizeof(float);
    if (lds > 160 * 1024) return fail(NGF_E_ARG, "the specialised kernel needs %zu bytes of LDS (> 160 KiB)", lds);
    const unsigned slot = f->next_counter.fetch_add(1) % kCounters;
    A.tile_counter = f->counters + (size_t)slot * kQueueHeads;
    HIP_TRY(hipMemsetAsync(A.tile_counter, 0, kQueueHeads * sizeof(unsigned), st));
    if (int rc = poison_lds(st)) return rc;
    A.xcd_queues = 1;
    A.tile_w = TW;
    A.tile_shift = TW == 8 ? 3 : 2;
    auto k = render_pc_kernel<P, NM, NS, TW>;
    HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(k), lds));
    const int64_t tiles = (A.n + TW - 1) / TW;
    int64_t grid = (tiles + NM - 1) / NM;
    if (grid > f->num_cus) grid = f->num_cus;
    if (grid < 1) grid = 1;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3((NM + NS) * kWave), lds, st, A);
    HIP_TRY(hipMemsetAsync(A.tile_counter, 0, kQueueHeads * sizeof(unsigned), st));      // this kernel does not zero its slot itself (render_kernel does: queue_done)
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}
#endif

template <bool BD, bool BC>
static int launch_triplane(const ngf_field *f, RenderArgs &A, hipStream_t st)
{
#ifndef NGF_EXPERIMENTS
    // product library: the fused kernel, twelve waves per CU, one march step per lane (measured best, profiles/r01_sweep.txt)
    if (knob(KNOB_KERNEL) > 0 || knob(KNOB_STAGE) > 0 || knob(KNOB_PROFILE) > 0 || knob(KNOB_NSTEP) > 1 || (knob(KNOB_WAVES) >= 0 && knob(KNOB_WAVES) != 12))
        return fail(NGF_E_UNSUPPORTED, "knobs kernel / stage / profile / nstep / waves select experiment kernels: load libngf_hip_exp.so (built with -DNGF_EXPERIMENTS)");
    if constexpr (BD) {
        if (A.mask.coarse) return launch_policy<MaskSkip<TriPlanePolicy<BD, BC, 12, 1>>>(f, A, st);      // a field with an alpha mask: the march skips empty space (levels 2 and 3)
    }
    return launch_policy<TriPlanePolicy<BD, BC, 12, 1>>(f, A, st);
#else
    // default: the fused kernel (every wave marches and shades).  ngf_debug_set("kernel", 1) selects the specialised march / shade
    // waves of ngf_render_pc.hpp: bit-identical, but SLOWER on gfx950 (R1 frame 10.7-12.7 ms vs 10.0 ms, profiles/r02_pc_kernel.txt),
    // because fp32 MFMA and fp32 VALU execute on the same SIMD datapath and never overlap -- not across waves, not within a wave
    // (profiles/micro/mfma_valu_overlap.hip: 8.7 ms + 3.1 ms run together in 11.6 ms) -- so there is nothing for the split to overlap.
    int kernel = knob(KNOB_KERNEL) >= 0 ? knob(KNOB_KERNEL) : 0;
    if (A.dbg_weight || A.skip_rgb || knob(KNOB_NSTEP) > 1 || (knob(KNOB_PROFILE) > 0 && knob(KNOB_KERNEL) != 1)) kernel = 0;
    if (knob(KNOB_TILE_W) > 8 || knob(KNOB_SPLIT) == 0) kernel = 0;
    if (kernel == 1 && knob(KNOB_PROFILE) > 0) {      // section cycles: stats[4..12] (profiles/exp_sections_pc.py); stats must hold 13 counters
        if constexpr (!BC) {
            using PP = TriPlanePolicy<BD, false, 12, 1, true>;
            return knob(KNOB_WAVES) == 88 ? launch_pc<PP, 8, 8, 8>(f, A, st) : launch_pc<PP, 12, 4, 8>(f, A, st);
        }
    }
    if (kernel == 1) {
        using P = TriPlanePolicy<BD, BC, 12, 1>;
        int tw = A.n < 40 * (int64_t)f->num_cus * 12 ? 4 : 8;
        if (knob(KNOB_TILE_W) == 4 || knob(KNOB_TILE_W) == 8) tw = knob(KNOB_TILE_W);
        const int w = knob(KNOB_WAVES) >= 0 ? knob(KNOB_WAVES) : 124;         // experiment: march waves * 10 + shade waves
        switch (w) {
        case 124: return tw == 8 ? launch_pc<P, 12, 4, 8>(f, A, st) : launch_pc<P, 12, 4, 4>(f, A, st);
        case 88: return tw == 8 ? launch_pc<P, 8, 8, 8>(f, A, st) : launch_pc<P, 8, 8, 4>(f, A, st);
        case 84: return tw == 8 ? launch_pc<P, 8, 4, 8>(f, A, st) : launch_pc<P, 8, 4, 4>(f, A, st);
        case 128: return launch_pc<P, 12, 4, 8>(f, A, st);
        default: return fail(NGF_E_ARG, "specialised kernel: knob waves must be 124, 88 or 84");
        }
    }
    // tuning knobs (measurements in profiles/): waves per CU and march steps in flight per lane
    int w = 12, ns = 1;                   // measured best (profiles/r01_sweep.txt)
    if (knob(KNOB_WAVES) >= 0) w = knob(KNOB_WAVES);
    if (knob(KNOB_NSTEP) >= 0) ns = knob(KNOB_NSTEP);
    if (ns == 2) {
        if (w == 8) return launch_policy<TriPlanePolicy<BD, BC, 8, 2>>(f, A, st);
        return fail(NGF_E_ARG, "knob nstep = 2 is built for waves = 8 only");
    }
    if (knob(KNOB_STAGE) > 0 && !A.dbg_weight && !A.skip_rgb) {      // LDS-staged texture strips (ngf_stage.hpp), faithful layout only; stats[13] += staged iterations
        if constexpr (!BD && !BC) {
            if (w == 8) return launch_policy<TriPlaneStagedPolicy<8>>(f, A, st);
            if (w == 12) return launch_policy<TriPlaneStagedPolicy<12>>(f, A, st);
            return fail(NGF_E_ARG, "knob stage: waves must be 8 (gauge + density strips) or 12 (gauge strips)");
        }
    }
    if (knob(KNOB_PROFILE) > 0) {      // stats[4..9] += section cycles (profiles/exp_sections.py); stats must hold 10 counters
        if constexpr (!BC) return launch_policy<TriPlanePolicy<BD, false, 12, 1, true>>(f, A, st);
    }
    switch (w) {
    case 8: return launch_policy<TriPlanePolicy<BD, BC, 8, 1>>(f, A, st);
    case 12:
        if constexpr (BD) {
            if (A.mask.coarse) return launch_policy<MaskSkip<TriPlanePolicy<BD, BC, 12, 1>>>(f, A, st);
        }
        return launch_policy<TriPlanePolicy<BD, BC, 12, 1>>(f, A, st);
    case 16: return launch_policy<TriPlanePolicy<BD, BC, 16, 1>>(f, A, st);
    default: return fail(NGF_E_ARG, "knob waves must be 8, 12 or 16");
    }
#endif
}

static int render_common(const ngf_field *f, RenderArgs &A, hipStream_t st)
{
    field_use(f, st);
    if (f->model == NGF_MODEL_INFOINV) {
        const bool wide = knob(KNOB_TILE_W) > 16 || knob(KNOB_SPLIT) == 0;      // debug knobs only: launch_render never picks more than 16 rays per tile
        if (f->flags & NGF_F_SPLIT_BF16) {
            if (wide) return fail(NGF_E_ARG, "InfoInv NGF_F_SPLIT_BF16 renders with split tiles of at most 16 rays");
            return A.mask.coarse ? launch_policy<MaskSkip<InfoInvSplitPolicy>>(f, A, st) : launch_policy<InfoInvSplitPolicy>(f, A, st);
        }
        if (wide) return launch_policy<InfoInvWidePolicy>(f, A, st);
        return A.mask.coarse ? launch_policy<MaskSkip<InfoInvPolicy>>(f, A, st) : launch_policy<InfoInvPolicy>(f, A, st);
    }
    if (f->flags & NGF_F_NO_FOLD) return launch_policy<TriPlaneNoFoldPolicy>(f, A, st);
    if ((f->flags & NGF_F_SPLIT_BF16) && (f->flags & NGF_F_BAKE_COLOR)) return A.mask.coarse ? launch_policy<MaskSkip<TriPlaneBakedBf16Policy>>(f, A, st) : launch_policy<TriPlaneBakedBf16Policy>(f, A, st);
    if (f->flags & NGF_F_SPLIT_BF16) {
        if (knob(KNOB_TILE_W) > 8 || knob(KNOB_SPLIT) == 0) return fail(NGF_E_ARG, "NGF_F_SPLIT_BF16 renders with split tiles of 4 or 8 rays");
        // 8 waves per CU: the pass keeps 48 registers of A fragments next to the gather buffer -- at the 168 registers that 12 waves
        // leave it spills 84 and runs 12.8 ms instead of 8.4 ms (profiles/r02_split_bf16.txt)
#ifdef NGF_EXPERIMENTS
        if (knob(KNOB_WAVES) == 12)
            return (f->flags & NGF_F_BAKE_DENSITY) ? launch_policy<TriPlaneBf16Policy<true, 12>>(f, A, st) : launch_policy<TriPlaneBf16Policy<false, 12>>(f, A, st);
#endif
        return (f->flags & NGF_F_BAKE_DENSITY) ? launch_policy<TriPlaneBf16Policy<true, 8>>(f, A, st) : launch_policy<TriPlaneBf16Policy<false, 8>>(f, A, st);
    }
    const bool bd = f->flags & NGF_F_BAKE_DENSITY, bc = f->flags & NGF_F_BAKE_COLOR;
    if (bd) return bc ? launch_triplane<true, true>(f, A, st) : launch_triplane<true, false>(f, A, st);
    return bc ? launch_triplane<false, true>(f, A, st) : launch_triplane<false, false>(f, A, st);
}

extern "C" int ngf_field_render(const ngf_field *f, const float *rays, int64_t n, int32_t n_samples, int32_t white_bg,
                                int32_t mode, const float *jitter, float *rgb, float *depth, uint64_t *stats, void *hip_stream)
{
    if (!f || !rays || !rgb || !depth) return fail(NGF_E_ARG, "ngf_field_render: null argument");
    if (n < 0 || n_samples <= 0) return fail(NGF_E_ARG, "ngf_field_render: n=%lld n_samples=%d", (long long)n, n_samples);
    if (n == 0) return NGF_OK;
    RenderArgs A = f->proto;
    A.rays = rays; A.jitter = jitter; A.rgb = rgb; A.depth = depth; A.n = n; A.S = n_samples;
    A.white_bg = white_bg ? 1 : 0; A.mode = mode ? 1 : 0; A.stats = (unsigned long long *)stats;
    if (knob(KNOB_ABLATE) > 0) A.ablate = knob(KNOB_ABLATE);          // ngf_debug_set("ablate", bits): A/B timing and bit-identity tests only
    return render_common(f, A, (hipStream_t)hip_stream);
}

// ABI 5: ngf_field_render for a ray list that IS an image -- rays [n,6] row-major, row_width rays per image row (the reference's evaluation hands
// `renderer` exactly that: samples.view(-1, 6) of an H x W frame, TriPlane/main.py:88-94).  Same pixels, bit for bit; the launch walks the image in
// screen-space blocks (RenderArgs::ord_*).  row_width <= 0, or a width the tile plan cannot use (not a multiple of the 8-ray tile): ngf_field_render.
extern "C" int ngf_field_render_image(const ngf_field *f, const float *rays, int64_t n, int32_t row_width, int32_t n_samples, int32_t white_bg,
                                      int32_t mode, const float *jitter, float *rgb, float *depth, uint64_t *stats, void *hip_stream)
{
    if (!f || !rays || !rgb || !depth) return fail(NGF_E_ARG, "ngf_field_render_image: null argument");
    if (n < 0 || n_samples <= 0) return fail(NGF_E_ARG, "ngf_field_render_image: n=%lld n_samples=%d", (long long)n, n_samples);
    if (n == 0) return NGF_OK;
    RenderArgs A = f->proto;
    A.rays = rays; A.jitter = jitter; A.rgb = rgb; A.depth = depth; A.n = n; A.S = n_samples;
    A.white_bg = white_bg ? 1 : 0; A.mode = mode ? 1 : 0; A.stats = (unsigned long long *)stats;
    if (knob(KNOB_ABLATE) > 0) A.ablate = knob(KNOB_ABLATE);
    A.ord_tpr = row_width > 0 && (int64_t)row_width < n && f->model == NGF_MODEL_TRIPLANE ? (uint32_t)row_width : 0u;          // launch_render turns the width into the plan (or drops it); InfoInv: the list's order (its kernels do not re-order)
    return render_common(f, A, (hipStream_t)hip_stream);
}

extern "C" int ngf_field_march(const ngf_field *f, const float *rays, int64_t n, int32_t n_samples, int32_t mode,
                               const float *jitter, float *sigma, float *weight, void *hip_stream)
{
    if (!f || !rays || !sigma || !weight) return fail(NGF_E_ARG, "ngf_field_march: null argument");
    if (n <= 0 || n_samples <= 0) return fail(NGF_E_ARG, "ngf_field_march: n=%lld n_samples=%d", (long long)n, n_samples);
    hipStream_t st = (hipStream_t)hip_stream;
    float *scratch = nullptr;
    HIP_TRY(hipMallocAsync((void **)&scratch, (size_t)n * 4 * sizeof(float), st));
    RenderArgs A = f->proto;
    A.rays = rays; A.jitter = jitter; A.rgb = scratch; A.depth = scratch + 3 * n; A.n = n; A.S = n_samples;
    A.white_bg = 0; A.mode = mode ? 1 : 0; A.skip_rgb = 1; A.dbg_sigma = sigma; A.dbg_weight = weight;
    int rc = render_common(f, A, st);
    (void)hipFreeAsync(scratch, st);
    return rc;
}

extern "C" int ngf_field_decode_rgb(const ngf_field *f, const float *coords, const float *dirs, int64_t n, int32_t mode,
                                    float *rgb, void *hip_stream)
{
    if (!f || !coords || !dirs || !rgb) return fail(NGF_E_ARG, "ngf_field_decode_rgb: null argument");
    if (n <= 0) return fail(NGF_E_ARG, "ngf_field_decode_rgb: n=%lld", (long long)n);
    hipStream_t st = (hipStream_t)hip_stream;
    field_use(f, st);
    RenderArgs A = f->proto;
    A.mode = mode ? 1 : 0;
    const size_t lds = ((size_t)((A.blob_floats + 3) & ~3) + 4 * 32 * kViewFeat) * sizeof(float);
    const int64_t nb = (n + 15) / 16;
    int grid = (int)((nb + 3) / 4);
    if (grid > 4 * f->num_cus) grid = 4 * f->num_cus;
    if (int prc = poison_lds(st)) return prc;
    auto go = [&](auto kern) -> int {
        HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds));
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, st, A, coords, dirs, n, rgb);
        return NGF_OK;
    };
    int rc;
    if (f->model == NGF_MODEL_INFOINV) rc = (f->flags & NGF_F_SPLIT_BF16) ? go(decode_rgb_kernel<InfoInvSplitPolicy>) : go(decode_rgb_kernel<InfoInvPolicy>);
    else if (f->flags & NGF_F_NO_FOLD) rc = go(decode_rgb_kernel<TriPlaneNoFoldPolicy>);
    else if ((f->flags & NGF_F_SPLIT_BF16) && (f->flags & NGF_F_BAKE_COLOR)) rc = go(decode_rgb_kernel<TriPlaneBakedBf16Policy>);
    else if (f->flags & NGF_F_SPLIT_BF16) rc = go(decode_rgb_kernel<TriPlaneBf16Policy<false, 8>>);
    else if (f->flags & NGF_F_BAKE_COLOR) rc = go(decode_rgb_kernel<TriPlanePolicy<false, true, 8, 1>>);
    else rc = go(decode_rgb_kernel<TriPlanePolicy<false, false, 8, 1>>);
    if (rc) return rc;
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

static int launch_alpha(const ngf_field *f, const float *xyz, const Lattice &L, int64_t n, int32_t mode, float length, float *alpha, hipStream_t st)
{
    field_use(f, st);
    RenderArgs A = f->proto;
    A.mode = mode ? 1 : 0;
    if (int rc = poison_lds(st)) return rc;
    int64_t grid = (n + 255) / 256;
    if (grid > 8 * (int64_t)f->num_cus) grid = 8 * (int64_t)f->num_cus;
    if (f->model == NGF_MODEL_INFOINV) {
        const size_t lds = (size_t)((A.blob_floats + 3) & ~3) * sizeof(float);
        if (grid > (int64_t)f->num_cus) grid = f->num_cus;
        if (f->flags & NGF_F_SPLIT_BF16) {        // same density MLP, other offset of its image in the LDS blob
            HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(alpha_kernel<InfoInvSplitPolicy>), lds));
            hipLaunchKernelGGL(alpha_kernel<InfoInvSplitPolicy>, dim3((unsigned)grid), dim3(256), lds, st, A, xyz, L, n, length, alpha);
        } else {
            HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(alpha_kernel<InfoInvPolicy>), lds));
            hipLaunchKernelGGL(alpha_kernel<InfoInvPolicy>, dim3((unsigned)grid), dim3(256), lds, st, A, xyz, L, n, length, alpha);
        }
    } else if (f->flags & NGF_F_BAKE_DENSITY) {
        hipLaunchKernelGGL((alpha_kernel<TriPlanePolicy<true, false, 8, 1>>), dim3((unsigned)grid), dim3(256), 0, st, A, xyz, L, n, length, alpha);
    } else {
        hipLaunchKernelGGL((alpha_kernel<TriPlanePolicy<false, false, 8, 1>>), dim3((unsigned)grid), dim3(256), 0, st, A, xyz, L, n, length, alpha);
    }
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

extern "C" int ngf_field_alpha(const ngf_field *f, const float *xyz, int64_t n, int32_t mode, float length, float *alpha, void *hip_stream)
{
    if (!f || !xyz || !alpha) return fail(NGF_E_ARG, "ngf_field_alpha: null argument");
    if (n < 0) return fail(NGF_E_ARG, "ngf_field_alpha: n=%lld", (long long)n);
    if (n == 0) return NGF_OK;
    return launch_alpha(f, xyz, Lattice{nullptr, nullptr, nullptr, 0, 0, 0}, n, mode, length, alpha, (hipStream_t)hip_stream);
}

extern "C" int ngf_field_alpha_mask_build(const ngf_field *f, int32_t mode, const float *sx, const float *sy, const float *sz, int32_t gx, int32_t gy,
                                          int32_t gz, float length, float thres, float *alpha_zyx, float *volume_zyx, float *new_aabb,
                                          uint64_t *count, void *hip_stream)
{
    if (!f || !sx || !sy || !sz || !alpha_zyx || !volume_zyx || !new_aabb || !count) return fail(NGF_E_ARG, "ngf_field_alpha_mask_build: null argument");
    if (gx < 1 || gy < 1 || gz < 1) return fail(NGF_E_ARG, "ngf_field_alpha_mask_build: grid %dx%dx%d", gx, gy, gz);
    hipStream_t st = (hipStream_t)hip_stream;
    const Lattice L{sx, sy, sz, gx, gy, gz};
    const int64_t n = (int64_t)gx * gy * gz;
    int rc = launch_alpha(f, nullptr, L, n, mode, length, alpha_zyx, st);
    if (rc) return rc;
    // index bounds of the occupied voxels: per-call scratch on the call's stream (concurrent builds on one handle do not share it)
    int *bounds = nullptr;
    HIP_TRY(hipMallocAsync((void **)&bounds, 6 * sizeof(int), st));
    hipLaunchKernelGGL(mask_bounds_init_kernel, dim3(1), dim3(64), 0, st, bounds);
    HIP_TRY(hipMemsetAsync(count, 0, sizeof(uint64_t), st));
    int64_t grid = (n + 255) / 256;
    if (grid > 16 * (int64_t)f->num_cus) grid = 16 * (int64_t)f->num_cus;
    hipLaunchKernelGGL(mask_pool_kernel, dim3((unsigned)grid), dim3(256), 0, st, (const float *)alpha_zyx, gx, gy, gz, thres, volume_zyx, bounds,
                       (unsigned long long *)count);
    hipLaunchKernelGGL(mask_aabb_kernel, dim3(1), dim3(64), 0, st, f->proto, L, (const int *)bounds, new_aabb);
    const hipError_t launch_err = hipGetLastError();
    (void)hipFreeAsync(bounds, st);
    HIP_TRY(launch_err);
    return NGF_OK;
}

extern "C" int ngf_field_ray_filter(const ngf_field *f, const float *rays, int64_t n, int32_t n_samples, uint8_t *keep, void *hip_stream)
{
    if (!f || !rays || !keep) return fail(NGF_E_ARG, "ngf_field_ray_filter: null argument");
    if (n_samples > 0 && !f->proto.mask.bits) return fail(NGF_E_ARG, "ngf_field_ray_filter: the field has no alpha mask");
    if (n < 0) return fail(NGF_E_ARG, "ngf_field_ray_filter: n=%lld", (long long)n);
    if (n == 0) return NGF_OK;
    field_use(f, (hipStream_t)hip_stream);
    RenderArgs A = f->proto;
    int64_t grid = (n + 255) / 256;
    if (grid > 16 * (int64_t)f->num_cus) grid = 16 * (int64_t)f->num_cus;
    hipLaunchKernelGGL(ray_filter_kernel, dim3((unsigned)grid), dim3(256), 0, (hipStream_t)hip_stream, A, rays, n, n_samples, keep);
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

extern "C" int ngf_generate_rays(int32_t H, int32_t W, float focal, const float *c, int32_t row0, int32_t rows, float *rays,
                                 void *hip_stream)
{
    if (!c || !rays || H <= 0 || W <= 0 || rows < 0 || row0 < 0 || row0 + rows > H) return fail(NGF_E_ARG, "ngf_generate_rays: bad argument");
    if (rows == 0) return NGF_OK;
    const int64_t total = (int64_t)rows * W;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(generate_rays_kernel, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, H, W, focal, c[0], c[1], c[2], c[4],
                       c[5], c[6], c[8], c[9], c[10], c[3], c[7], c[11], row0, rows, rays);
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

extern "C" int ngf_generate_rays_dtu(int32_t H, int32_t W, const float *focal, const float *princpt, const float *rot, int32_t row0,
                                     int32_t rows, float *raydir, void *hip_stream)
{
    if (!focal || !princpt || !rot || !raydir || H <= 0 || W <= 0 || rows < 0 || row0 < 0 || row0 + rows > H)
        return fail(NGF_E_ARG, "ngf_generate_rays_dtu: bad argument");
    if (rows == 0) return NGF_OK;
    const int64_t total = (int64_t)rows * W;
    int grid = (int)((total + 255) / 256);
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(generate_rays_dtu_kernel, dim3(grid), dim3(256), 0, (hipStream_t)hip_stream, W, focal[0], focal[1], princpt[0],
                       princpt[1], rot[0], rot[1], rot[2], rot[3], rot[4], rot[5], rot[6], rot[7], rot[8], row0, rows, raydir);
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

// ================================ training step ============================================================
enum { TP_PLANE = 0, TP_GAUGE = 3, TP_DENS_W = 6, TP_DENS_B = 7, TP_BASIS = 8, TP_W1 = 9, TP_B1 = 10, TP_W2 = 11, TP_B2 = 12, TP_W3 = 13,
       TP_B3 = 14, TP_COUNT = 15 };

struct ngf_trainer {
    int dev = 0;                             // the device the trainer's buffers and streams live on
    ngf_train_desc d;
    TrainArgs proto;
    std::vector<void *> allocs;
    float *tex_d[3] = {}, *tex_a[3] = {}, *tex_g[3] = {};
    float *g_d[3] = {}, *g_a[3] = {}, *g_g[3] = {};
    float *g_gb[3] = {};                     // gauge-plane gradients in the blocked layout the scatter writes (g_g: [texel][2])
    float *q_d[3] = {}, *d_d[3] = {};        // wd-projected density planes, scalar density-gradient images
    float *fwd_image = nullptr, *bwd_image = nullptr;      // LDS images of the colour MLP (train_fold_kernel)
    float *fwd16_image = nullptr;                          // ... and the forward's image in the eval pass's layout (train_color_fwd16_kernel)
    float *g_dense[TP_COUNT] = {};          // reference-layout gradient buffers of the MLP parameters (index TP_*)
    int64_t dense_n[TP_COUNT] = {};
    uint8_t *mask = nullptr;
    uint8_t *mask_cells = nullptr;
    bool tex_fresh[6] = {};                 // the packed copy of plane / gauge plane k holds the parameter's current values
    char *zero_arena = nullptr;             // every buffer a step accumulates into (gradients, M, loss): one memset per step
    size_t zero_bytes = 0;
    int64_t chunk = 0;
    bool speculative = false;               // chunk_samples < 0: `chunk` rows, never a host round trip; a batch with more active samples is flagged on the device
    int32_t *overflow = nullptr;            // device: [0] this step's batch had more active samples than rows (Adam then skips), [1] how often that happened
    int64_t bytes = 0;
    int num_cus = 256;
    // after the colour backward the step forks: weight-gradient GEMMs | colour-plane scatter | density / gauge backward are independent
    // chains of kernels none of which fills the device on its own (row transposes, LDS latency, the atomic unit); ngf_train_adam_all
    // updates the three planes side by side
    static constexpr int kAux = 2;
    hipStream_t aux[kAux] = {nullptr, nullptr};
    hipEvent_t ev_fork = nullptr, ev_join[kAux] = {nullptr, nullptr};
    bool has_adam = true;                   // false: built without Adam moments (ngf_train_forward / ngf_train_backward_grad only; ngf_train_adam* refuse)
    // what ngf_train_forward leaves for ngf_train_backward_grad (the two-call form of the step)
    struct Pending {
        bool valid = false;
        TrainArgs T;
        bool fork = false, no_sync = false, single = true;
        int64_t n = 0, list_len = 0;
        int32_t n_samples = 0;
        int64_t ticket = 0;
    } pending;
    int64_t tickets = 0;
};

template <typename T>
static int tr_alloc(ngf_trainer *t, T **p, size_t count)
{
    void *q = nullptr;
    if (hipMalloc(&q, count * sizeof(T) + 64) != hipSuccess) return fail(NGF_E_HIP, "hipMalloc(%zu bytes) failed for the trainer", count * sizeof(T));
    t->allocs.push_back(q);
    t->bytes += (int64_t)(count * sizeof(T));
    *p = (T *)q;
    return NGF_OK;
}

extern "C" int ngf_trainer_destroy(ngf_trainer *t)
{
    if (!t) return NGF_OK;
    DeviceScope ds(t->dev);              // hipFree / stream teardown on the trainer's device, whatever is current in the calling thread
    for (void *q : t->allocs) (void)hipFree(q);
    for (int k = 0; k < ngf_trainer::kAux; ++k) {
        if (t->aux[k]) { (void)hipStreamSynchronize(t->aux[k]); (void)hipStreamDestroy(t->aux[k]); }
        if (t->ev_join[k]) (void)hipEventDestroy(t->ev_join[k]);
    }
    if (t->ev_fork) (void)hipEventDestroy(t->ev_fork);
    delete t;
    return NGF_OK;
}

extern "C" int64_t ngf_trainer_bytes(const ngf_trainer *t) { return t ? t->bytes : 0; }
extern "C" int32_t ngf_sizeof_train_desc(void) { return (int32_t)sizeof(ngf_train_desc); }

extern "C" int ngf_trainer_create(const ngf_train_desc *d, ngf_trainer **out, void *hip_stream)
{
    if (!d || !out) return fail(NGF_E_ARG, "ngf_trainer_create: null argument");
    if (d->max_rays <= 0 || d->max_samples <= 0) return fail(NGF_E_ARG, "ngf_trainer_create: max_rays / max_samples must be positive");
    for (int p = 0; p < 3; ++p) {
        if (!d->plane[p] || d->plane_h[p] < 2 || d->plane_w[p] < 2) return fail(NGF_E_ARG, "plane %d missing or smaller than 2x2", p);
        if (!d->gauge[p] || d->gauge_h[p] < 2 || d->gauge_w[p] < 2) return fail(NGF_E_ARG, "gauge plane %d missing or smaller than 2x2", p);
    }
    if (!d->dens_w || !d->dens_b || !d->basis || !d->w1 || !d->b1 || !d->w2 || !d->b2 || !d->w3 || !d->b3)
        return fail(NGF_E_ARG, "ngf_trainer_create: missing MLP parameter");
    // Adam moments: all fifteen pairs, or none at all (a trainer that only serves ngf_train_forward / ngf_train_backward_grad -- the caller's
    // own optimiser applies the gradients, e.g. torch.optim.Adam in the reference's loop, TriPlane/main.py:241,294-296)
    int moments = 0;
    for (int k = 0; k < TP_COUNT; ++k) moments += (d->exp_avg[k] ? 1 : 0) + (d->exp_avg_sq[k] ? 1 : 0);
    if (moments != 0 && moments != 2 * TP_COUNT) return fail(NGF_E_ARG, "ngf_trainer_create: Adam state must be given for all %d parameters or for none", (int)TP_COUNT);
    ngf_trainer *t = new (std::nothrow) ngf_trainer();
    if (!t) return fail(NGF_E_HIP, "out of host memory");
    t->d = *d;
    t->has_adam = moments != 0;
    auto bail = [&](int rc) { ngf_trainer_destroy(t); return rc; };
    hipStream_t st = (hipStream_t)hip_stream;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) t->num_cus = prop.multiProcessorCount;
    t->dev = dev;
    int rc;
    for (int k = 0; k < ngf_trainer::kAux; ++k)
        if (hipStreamCreateWithFlags(&t->aux[k], hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&t->ev_join[k], hipEventDisableTiming) != hipSuccess)
            return bail(fail(NGF_E_HIP, "trainer: stream / event creation failed"));
    if (hipEventCreateWithFlags(&t->ev_fork, hipEventDisableTiming) != hipSuccess) return bail(fail(NGF_E_HIP, "trainer: event creation failed"));
    TrainArgs &T = t->proto;
    memset(&T, 0, sizeof(T));
    RenderArgs &A = T.R;
    const int64_t dn[TP_COUNT] = {0, 0, 0, 0, 0, 0, 48, 1, 144 * 144, 64 * 159, 64, 64 * 64, 64, 3 * 64, 3};
    // blocked gradient images (ngf_train.hpp, scatter_blocked): 16-float blocks of 4x4 texels (D_p) / 4x2 texels x 2 channels (gauge)
    auto dblk = [&](int p) { return (size_t)((d->plane_w[p] + 2 + 3) / 4) * ((d->plane_h[p] + 2 + 3) / 4) * 16; };
    auto gblk = [&](int p) { return (size_t)((d->gauge_w[p] + 2 + 3) / 4) * ((d->gauge_h[p] + 2 + 1) / 2) * 16; };
    // bins of the colour-plane scatter (ngf_train.hpp section 5b): 8x8 blocks of cells of the padded planes
    int nbins = 0;
    for (int p = 0; p < 3; ++p) {
        T.bin_base[p] = nbins;
        T.bin_nbx[p] = (d->plane_w[p] + 2 + 7) / 8;
        nbins += T.bin_nbx[p] * ((d->plane_h[p] + 2 + 7) / 8);
    }
    T.nbins = nbins;
    {   // the zero arena: [per plane: D_p, gauge gradient] [MLP gradients] [M] [loss] [bin counters], 256-byte aligned pieces
        size_t total = 0;
        auto add = [&](size_t floats) { total += (floats * sizeof(float) + 255) & ~(size_t)255; };
        for (int p = 0; p < 3; ++p) { add(dblk(p)); add(gblk(p)); }
        for (int k = TP_DENS_W; k < TP_COUNT; ++k) add((size_t)dn[k]);
        add((size_t)64 * 144); add(4);
        add((size_t)nbins + 1);
        if ((rc = tr_alloc(t, &t->zero_arena, total))) return bail(rc);
        t->zero_bytes = total;
    }
    size_t carved = 0;
    auto carve = [&](size_t floats) {
        float *q = reinterpret_cast<float *>(t->zero_arena + carved);
        carved += (floats * sizeof(float) + 255) & ~(size_t)255;
        return q;
    };
    for (int p = 0; p < 3; ++p) {
        const int H = d->plane_h[p], W = d->plane_w[p], gh = d->gauge_h[p], gw = d->gauge_w[p];
        const size_t tex = (size_t)(H + 2) * (W + 2), gtex = (size_t)(gh + 2) * (gw + 2);
        if ((rc = tr_alloc(t, &t->tex_d[p], tex * 16)) || (rc = tr_alloc(t, &t->tex_a[p], tex * 48)) || (rc = tr_alloc(t, &t->tex_g[p], gtex * 2)) ||
            (rc = tr_alloc(t, &t->q_d[p], tex)) || (rc = tr_alloc(t, &t->g_d[p], tex * 16)) || (rc = tr_alloc(t, &t->g_g[p], gtex * 2)))
            return bail(rc);
        t->d_d[p] = carve(dblk(p)); t->g_gb[p] = carve(gblk(p));
        // the colour planes' gradients are WRITTEN by train_bin_gather_kernel (every texel): no fill per step
        if ((rc = tr_alloc(t, &t->g_a[p], tex * 48))) return bail(rc);
        if (hipMemsetAsync(t->g_a[p], 0, tex * 48 * sizeof(float), st) != hipSuccess) return bail(fail(NGF_E_HIP, "trainer setup failed"));
        T.d_bw[p] = (W + 2 + 3) / 4; T.g_bw[p] = (gw + 2 + 3) / 4;
        A.dens[p] = Tex{t->tex_d[p], W, H, W + 2, (float)(W - 1), (float)(H - 1)};
        A.app[p] = Tex{t->tex_a[p], W, H, W + 2, (float)(W - 1), (float)(H - 1)};
        A.gau[p] = Tex{t->tex_g[p], gw, gh, gw + 2, (float)(gw - 1), (float)(gh - 1)};
        T.g_dens[p] = t->g_d[p]; T.g_app[p] = t->g_a[p]; T.g_gau[p] = t->g_gb[p];
        T.q_dens[p] = t->q_d[p]; T.d_dens[p] = t->d_d[p];
    }
    for (int k = TP_DENS_W; k < TP_COUNT; ++k) {
        t->dense_n[k] = dn[k];
        t->g_dense[k] = carve((size_t)dn[k]);
    }
    T.M = carve((size_t)64 * 144);
    T.loss = reinterpret_cast<double *>(carve(4));
    T.bin_count = reinterpret_cast<int32_t *>(carve((size_t)nbins + 1));
    if (carved != t->zero_bytes) return bail(fail(NGF_E_ARG, "trainer: zero arena layout mismatch"));
    T.wd = d->dens_w; T.bd = d->dens_b; T.basis = d->basis; T.w1 = d->w1; T.b1 = d->b1; T.w2 = d->w2; T.b2 = d->b2; T.w3 = d->w3; T.b3 = d->b3;
    T.g_wd = t->g_dense[TP_DENS_W]; T.g_bd = t->g_dense[TP_DENS_B];
    if ((rc = tr_alloc(t, &T.prof, (size_t)16))) return bail(rc);
    if (hipMemsetAsync(T.prof, 0, 16 * sizeof(unsigned long long), st) != hipSuccess) return bail(fail(NGF_E_HIP, "trainer setup failed"));
    T.g_b1 = t->g_dense[TP_B1]; T.g_b2 = t->g_dense[TP_B2]; T.g_b3 = t->g_dense[TP_B3];
    for (int k = 0; k < 3; ++k) {
        A.a0[k] = d->aabb[k];
        A.a1[k] = d->aabb[3 + k];
        A.inv[k] = 2.0f / (d->aabb[3 + k] - d->aabb[k]);
    }
    A.near_ = d->near_; A.far_ = d->far_; A.step = d->step; A.dscale = d->distance_scale; A.thr = d->weight_thres;
    if (d->mask_bits) {
        const size_t nbytes = ((size_t)d->mask_d * d->mask_h * d->mask_w + 7) / 8;
        if ((rc = tr_alloc(t, &t->mask, nbytes))) return bail(rc);
        if (hipMemcpyAsync(t->mask, d->mask_bits, nbytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return bail(fail(NGF_E_HIP, "copying the alpha mask failed"));
        A.mask.bits = t->mask;
        A.mask.D = d->mask_d; A.mask.H = d->mask_h; A.mask.W = d->mask_w;
        const size_t ncells = (size_t)(d->mask_d + 1) * (d->mask_h + 1) * (d->mask_w + 1);
        if (ncells >= ((size_t)1 << 32)) return bail(fail(NGF_E_ARG, "alpha mask of %d x %d x %d cells: the cell image is indexed with 32 bits", d->mask_d, d->mask_h, d->mask_w));
        if ((rc = tr_alloc(t, &t->mask_cells, ncells))) return bail(rc);
        hipLaunchKernelGGL(mask_cells_kernel, dim3(2048), dim3(256), 0, st, (const uint8_t *)t->mask, d->mask_d, d->mask_h, d->mask_w, t->mask_cells);
        if (hipGetLastError() != hipSuccess) return bail(fail(NGF_E_HIP, "mask_cells_kernel failed to launch"));
        A.mask.cells = t->mask_cells;
        for (int k = 0; k < 3; ++k) {
            A.mask.a0[k] = d->mask_aabb[k];
            A.mask.inv[k] = 1.0f / (d->mask_aabb[3 + k] - d->mask_aabb[k]) * 2;
        }
    }
    const size_t cap = (size_t)d->max_rays * d->max_samples;
    // activation rows kept at once: by default the whole batch (1.7 KB per sample -- 6.2 GB for 4096 rays x 884 samples; the MI355X
    // has 288 GB), which also lets the step run without a host round trip; at most ~16 GB unless the caller asks otherwise
    // chunk_samples < 0 (speculative rows): |chunk_samples| rows, the step never waits for the host; a batch with more active samples than that
    // is flagged by the device (Adam skips the step, ngf_train_overflow_count reports it) -- 4096 x 884 pairs with rows for a third of them
    // are 3.4 GiB instead of 9.0
    t->speculative = d->chunk_samples < 0;
    const int64_t want = d->chunk_samples < 0 ? -d->chunk_samples : d->chunk_samples;
    t->chunk = want > 0 ? want : (int64_t)std::min<size_t>(cap, (size_t)9 << 20);
    if ((size_t)t->chunk > cap) t->chunk = (int64_t)((cap + 15) & ~(size_t)15);
    if (want <= 0 && (size_t)t->chunk >= cap) t->chunk = (int64_t)((cap + 15) & ~(size_t)15);
    if (t->speculative) t->chunk = (t->chunk + 15) & ~(int64_t)15;
    if ((rc = tr_alloc(t, &t->overflow, 2))) return bail(rc);
    if (hipMemsetAsync(t->overflow, 0, 2 * sizeof(int32_t), st) != hipSuccess) return bail(fail(NGF_E_HIP, "trainer setup failed"));
    if ((rc = tr_alloc(t, &T.et, cap)) || (rc = tr_alloc(t, &T.sg, cap)) || (rc = tr_alloc(t, &T.w, cap)) || (rc = tr_alloc(t, &T.dx, cap)) || (rc = tr_alloc(t, &T.c, cap * 3)) ||
        (rc = tr_alloc(t, &T.dt, cap * 6)) || (rc = tr_alloc(t, &T.G, (size_t)d->max_rays * 3)) || (rc = tr_alloc(t, &T.count, (size_t)d->max_rays)) ||
        (rc = tr_alloc(t, &T.offset, (size_t)d->max_rays + 1)) || (rc = tr_alloc(t, &T.list, cap * 2)) || (rc = tr_alloc(t, &T.list_w, cap)) ||
        (rc = tr_alloc(t, &t->fwd_image, (size_t)kFwdImage)) || (rc = tr_alloc(t, &t->bwd_image, (size_t)kBwdImage)) || (rc = tr_alloc(t, &t->fwd16_image, (size_t)MlpLayout16<48>::TOTAL)) ||
        (rc = tr_alloc(t, &T.bin_off, (size_t)nbins + 1)) || (rc = tr_alloc(t, &T.bin_unit, (size_t)nbins + 1)) || (rc = tr_alloc(t, &T.unit_total, (size_t)1)))
        return bail(rc);
    // The activation rows (576 floats per sample + the 18 of its scatter pairs).  If the whole-batch default does not fit the free HBM, fall back to round 1's chunked
    // mode (262 144 rows = 453 MB; the colour kernels then run chunk by chunk and the step reads the active count on the host) instead of
    // failing the create -- an explicit chunk_samples is taken as asked.
    for (int attempt = 0;; ++attempt) {
        const size_t ch = (size_t)t->chunk, mark = t->allocs.size();
        const int64_t bytes_mark = t->bytes;
        T.bin_cap = (int32_t)ch;
        if (!((rc = tr_alloc(t, &T.F, ch * 144)) || (rc = tr_alloc(t, &T.V, ch * 16)) || (rc = tr_alloc(t, &T.H1, ch * 64)) || (rc = tr_alloc(t, &T.H2, ch * 64)) ||
              (rc = tr_alloc(t, &T.D3, ch * 16)) || (rc = tr_alloc(t, &T.D2, ch * 64)) || (rc = tr_alloc(t, &T.D1, ch * 64)) ||
              (rc = tr_alloc(t, &T.DF, ch * 144)) || (rc = tr_alloc(t, &T.pair_cell, ch * 3)) || (rc = tr_alloc(t, &T.pair_rank, ch * 3)) ||
              (rc = tr_alloc(t, &T.pair_w, ch * 12)) || (rc = tr_alloc(t, &T.perm, ch * 3)) ||
              (rc = tr_alloc(t, &T.units, ((size_t)nbins + ch * 3 / kBinChunk + 8) * 3)) ||
              (rc = tr_alloc(t, &T.slab, ((size_t)nbins + ch * 3 / kBinChunk + 8) * kBinTile))))
            break;
        while (t->allocs.size() > mark) { (void)hipFree(t->allocs.back()); t->allocs.pop_back(); }
        t->bytes = bytes_mark;
        (void)hipGetLastError();
        if (attempt > 0 || d->chunk_samples != 0 || ch <= ((size_t)1 << 18)) return bail(rc);
        t->chunk = (int64_t)1 << 18;
    }
    // the packed copies (their zero borders are written here and never again) and defined gradients before the first backward
    for (int p = 0; p < 3; ++p) {
        pack_plane_kernel<<<2048, 256, 0, st>>>(d->plane[p], d->plane_h[p], d->plane_w[p], 0, 16, t->tex_d[p]);
        pack_plane_kernel<<<2048, 256, 0, st>>>(d->plane[p], d->plane_h[p], d->plane_w[p], 16, 48, t->tex_a[p]);
        pack_plane_kernel<<<256, 256, 0, st>>>(d->gauge[p], d->gauge_h[p], d->gauge_w[p], 0, 2, t->tex_g[p]);
        t->tex_fresh[p] = t->tex_fresh[3 + p] = true;
        if (hipMemsetAsync(t->g_d[p], 0, (size_t)(d->plane_h[p] + 2) * (d->plane_w[p] + 2) * 16 * sizeof(float), st) != hipSuccess ||
            hipMemsetAsync(t->g_g[p], 0, (size_t)(d->gauge_h[p] + 2) * (d->gauge_w[p] + 2) * 2 * sizeof(float), st) != hipSuccess)
            return bail(fail(NGF_E_HIP, "trainer setup failed"));
    }
    if (hipMemsetAsync(t->zero_arena, 0, t->zero_bytes, st) != hipSuccess) return bail(fail(NGF_E_HIP, "trainer setup failed"));
    if (hipStreamSynchronize(st) != hipSuccess) return bail(fail(NGF_E_HIP, "trainer setup failed: %s", hipGetErrorString(hipGetLastError())));
    *out = t;
    return NGF_OK;
}

static int tr_grid(const ngf_trainer *t, int64_t items, int per_block, int waves_per_cu = 8)
{
    int64_t g = (items + per_block - 1) / per_block;
    const int64_t cap = (int64_t)t->num_cus * waves_per_cu;
    if (g > cap) g = cap;
    return g < 1 ? 1 : (int)g;
}

// What of the step's forks is still open on the aux streams: an error exit of the body must not leave them un-joined to the caller's stream
// (the next call on that stream would otherwise run beside this one's aux-stream kernels).
struct ForkState { bool fold = false, chains = false; };

// The step in two parts.  Part A (train_forward_part): everything up to and including the colour forward -- after it the per-sample weights and
// colours of the batch are in the trainer's buffers.  Part B (train_backward_part): compositing backward, colour backward, the three forked
// chains.  The fused entry points run A, the fused compositing kernel <0> and B in one call; ngf_train_forward runs A and the compositing
// kernel <1> (rgb_map / depth_map out), ngf_train_backward_grad the compositing kernel <2> (d loss / d rgb_map in) and B.
static int train_forward_part(ngf_trainer *t, const float *rays, const float *jitter, int64_t n, int32_t n_samples, int32_t white_bg, int32_t gauge_on,
                              int64_t *n_active_host, void *hip_stream, ForkState &fs, ngf_trainer::Pending &P)
{
    if (n <= 0 || n > t->d.max_rays || n_samples <= 0 || n_samples > t->d.max_samples)
        return fail(NGF_E_ARG, "ngf_train_backward: n=%lld (max %lld), n_samples=%d (max %d)", (long long)n, (long long)t->d.max_rays, n_samples,
                    t->d.max_samples);
    hipStream_t st = (hipStream_t)hip_stream;
    const ngf_train_desc &d = t->d;
    P.valid = false;
    P.T = t->proto;
    TrainArgs &T = P.T;
    RenderArgs &A = T.R;
    A.rays = rays; A.jitter = jitter; A.n = n; A.S = n_samples; A.white_bg = white_bg ? 1 : 0; A.mode = gauge_on ? 1 : 0;
    T.inv_count = 1.0f / (3.0f * (float)n);
    A.ablate = knob(KNOB_ABLATE) > 0 ? knob(KNOB_ABLATE) : 0;      // timing experiments only (profiles/exp_train_ablate.sh)
    if (int prc = poison_lds(st)) return prc;
    // ngf_debug_set("ablate", 1 << 19) keeps the whole step on the caller's stream (see the forks below)
    const bool fork = !(A.ablate & (1 << 19));
    hipStream_t sx = fork ? t->aux[0] : st;
    ProjectArgs PJ;
    PJ.wd = d.dens_w;
    // parameters -> packed textures where the trainer's copy is not current (ngf_train_adam writes the copy along with the parameter;
    // ngf_train_params_changed marks every copy stale); gradient buffers -> 0
    for (int p = 0; p < 3; ++p) {
        const int H = d.plane_h[p], W = d.plane_w[p], gh = d.gauge_h[p], gw = d.gauge_w[p];
        const size_t tex = (size_t)(H + 2) * (W + 2);
        if (!t->tex_fresh[p]) {
            pack_plane_kernel<<<2048, 256, 0, st>>>(d.plane[p], H, W, 0, 16, t->tex_d[p]);
            pack_plane_kernel<<<2048, 256, 0, st>>>(d.plane[p], H, W, 16, 48, t->tex_a[p]);
            t->tex_fresh[p] = true;
        }
        if (!t->tex_fresh[3 + p]) {
            pack_plane_kernel<<<256, 256, 0, st>>>(d.gauge[p], gh, gw, 0, 2, t->tex_g[p]);
            t->tex_fresh[3 + p] = true;
        }
        PJ.tex16[p] = t->tex_d[p]; PJ.texels[p] = (int64_t)tex; PJ.q[p] = t->q_d[p];
    }
    hipLaunchKernelGGL(train_project_density_kernel, dim3(128, 3), dim3(256), 0, st, PJ);
    HIP_TRY(hipMemsetAsync(t->zero_arena, 0, t->zero_bytes, st));
    // the LDS images of the colour MLP are first read by the colour forward: built beside the density kernel and the scan
    T.fwd_image = t->fwd_image; T.bwd_image = t->bwd_image; T.fwd16_image = t->fwd16_image;
    if (fork) {
        HIP_TRY(hipEventRecord(t->ev_fork, st));
        HIP_TRY(hipStreamWaitEvent(sx, t->ev_fork, 0));
    }
    hipLaunchKernelGGL(train_fold_kernel, dim3(48), dim3(256), 0, sx, T, t->fwd_image, t->bwd_image, t->fwd16_image);
    if (fork) { HIP_TRY(hipEventRecord(t->ev_join[0], sx)); fs.fold = true; }

    const int64_t pairs = n * n_samples;
    hipLaunchKernelGGL(train_density_kernel, dim3(tr_grid(t, pairs, 256)), dim3(256), 0, st, T);
    const int ray_blocks = (int)((n + 3) / 4);        // sixteen lanes per ray
    hipLaunchKernelGGL(train_scan_kernel, dim3(ray_blocks), dim3(64), 0, st, T, 0);
    // The overflow flag belongs to the path that cannot cut the list into chunks: speculative rows WITHOUT a host count.  A caller that passes
    // n_active_host gets the chunked path and a complete gradient -- flagging that step would make Adam skip a valid update.
    const bool spec_rows = t->speculative && !n_active_host;
    hipLaunchKernelGGL(train_prefix_kernel, dim3(1), dim3(1024), 0, st, (const int32_t *)T.count, n, T.offset, spec_rows ? t->chunk : (int64_t)0, t->overflow);
    // The colour kernels walk the active list.  When one chunk of activation rows holds every sample of the batch (the default:
    // HBM is sized for it) they read the active count from the device and run with fixed grids -- the stream never waits for the
    // host.  With a smaller chunk (chunk_samples of the descriptor) the count comes to the host to cut the list into chunks.
    const bool no_sync = (t->chunk >= pairs || t->speculative) && !n_active_host;
    int32_t n_active = 0;
    if (!no_sync) {
        HIP_TRY(hipMemcpyAsync(&n_active, T.offset + n, sizeof(int32_t), hipMemcpyDeviceToHost, st));
        HIP_TRY(hipStreamSynchronize(st));
        if (n_active_host) *n_active_host = n_active;
    }
    hipLaunchKernelGGL(train_scan_kernel, dim3(ray_blocks), dim3(64), 0, st, T, 1);

    const size_t lds_f = (size_t)(((kFwdImage + 3) & ~3) + kTrainWaves * kFwdTileFloats) * sizeof(float),
                 lds_b = (size_t)(((kBwdImage + 3) & ~3) + kTrainWavesBwd * kBwdTileFloats) * sizeof(float);
    static_assert((((kFwdImage + 3) & ~3) + kTrainWaves * kFwdTileFloats) * 4 <= 160 * 1024, "colour forward LDS");
    static_assert((((kBwdImage + 3) & ~3) + kTrainWavesBwd * kBwdTileFloats) * 4 <= 160 * 1024, "colour backward LDS");
    HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(train_color_fwd_kernel), lds_f));
    const size_t lds_f16 = (size_t)((MlpLayout16<48>::TOTAL + 3) & ~3) * sizeof(float);
    HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(train_color_fwd16_kernel), lds_f16));
    HIP_TRY(ensure_dynamic_lds(reinterpret_cast<const void *>(train_color_bwd_kernel), lds_b));
    if (fork) { HIP_TRY(hipStreamWaitEvent(st, t->ev_join[0], 0)); fs.fold = false; }          // the images
    const int32_t *cnt = no_sync ? T.offset + n : nullptr;
    T.n_active_dev = cnt;
    // upper bound of the list length the loops below are sized for (speculative rows: what the trainer keeps rows for -- the kernels stop at
    // min(that, the device's count); a longer list is flagged, train_prefix_kernel)
    const int64_t list_len = no_sync ? std::min<int64_t>(pairs, t->chunk) : n_active;
    const bool single = list_len <= t->chunk;
    // colour forward over the whole list (activations kept when the list fits one chunk)
    for (int64_t base = 0; base < list_len; base += t->chunk) {
        T.chunk_base = (int32_t)base;
        T.chunk_n = (int32_t)std::min<int64_t>(t->chunk, list_len - base);
        T.store = single ? 1 : 0;
        // round 5: the colour forward in the eval pass's shape (twelve waves per CU, rows stored from registers); ngf_debug_set("ablate", 1 << 23) = the LDS-tile kernel (A/B)
        if (A.ablate & (1 << 23)) hipLaunchKernelGGL(train_color_fwd_kernel, dim3(tr_grid(t, (T.chunk_n + 15) / 16, kTrainWaves, 1)), dim3(kTrainWaves * 64), lds_f, st, T);
        else hipLaunchKernelGGL(train_color_fwd16_kernel, dim3(tr_grid(t, (T.chunk_n + 15) / 16, kTrainWaves16, 1)), dim3(kTrainWaves16 * 64), lds_f16, st, T);
    }
    P.fork = fork; P.no_sync = no_sync; P.single = single; P.n = n; P.n_samples = n_samples; P.list_len = list_len;
    P.valid = true;
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

static inline dim3 comp_grid(int64_t n) { return dim3((unsigned)((n + 64 / kCompLanes - 1) / (64 / kCompLanes))); }

// Part B; the compositing kernel of the calling form has run on hip_stream.  rgb_loss may be NULL (no loss is delivered).
static int train_backward_part(ngf_trainer *t, ngf_trainer::Pending &P, double *rgb_loss, int32_t loss_len, void *hip_stream, ForkState &fs)
{
    hipStream_t st = (hipStream_t)hip_stream;
    const ngf_train_desc &d = t->d;
    TrainArgs &T = P.T;
    RenderArgs &A = T.R;
    (void)A;
    const bool fork = P.fork, no_sync = P.no_sync, single = P.single;
    const int64_t n = P.n, list_len = P.list_len;
    const int32_t n_samples = P.n_samples;
    hipStream_t sx = fork ? t->aux[0] : st, sb = fork ? t->aux[1] : st;
    const size_t lds_f = (size_t)(((kFwdImage + 3) & ~3) + kTrainWaves * kFwdTileFloats) * sizeof(float),
                 lds_b = (size_t)(((kBwdImage + 3) & ~3) + kTrainWavesBwd * kBwdTileFloats) * sizeof(float);
    const int32_t *cnt = no_sync ? T.offset + n : nullptr;
    const dim3 dgrid(tr_grid(t, n * ((n_samples + 63) / 64), 4, knob(KNOB_TRAIN_DWG) > 0 ? knob(KNOB_TRAIN_DWG) : kDensBwdGroupsPerCu + 1));          // knob train_dwg (sweeps): workgroups per CU
    // After the colour backward of a chunk the step forks: the weight-gradient GEMMs (sx) and the colour-plane scatter (sb) leave the
    // caller's stream, which goes on with the density / gauge backward and waits for both before it returns to the caller's order.  None of
    // these chains fills the device alone (LDS transposes, LDS latency, the atomic unit).  (Tried: the density / gauge backward beside the
    // colour backward, with the colour path's d loss / d t scattered into the gauge planes by a kernel of its own -- the active entries of a
    // ray span the whole plane, their scatter cannot be merged in LDS, and the step got 0.17 ms slower.)
    auto join = [&]() -> int {
        if (fork && fs.chains) {
            HIP_TRY(hipStreamWaitEvent(st, t->ev_join[0], 0));
            HIP_TRY(hipStreamWaitEvent(st, t->ev_join[1], 0));
        }
        fs.chains = false;
        return NGF_OK;
    };
    for (int64_t base = 0; base < list_len; base += t->chunk) {
        T.chunk_base = (int32_t)base;
        T.chunk_n = (int32_t)std::min<int64_t>(t->chunk, list_len - base);
        const int passes = (T.chunk_n + 15) / 16;
        if (int jrc = join()) return jrc;              // the previous chunk's chains read the rows this chunk overwrites
        if (!single) {
            T.store = 1;
            if (A.ablate & (1 << 23)) hipLaunchKernelGGL(train_color_fwd_kernel, dim3(tr_grid(t, passes, kTrainWaves, 1)), dim3(kTrainWaves * 64), lds_f, st, T);
            else hipLaunchKernelGGL(train_color_fwd16_kernel, dim3(tr_grid(t, passes, kTrainWaves16, 1)), dim3(kTrainWaves16 * 64), (size_t)((MlpLayout16<48>::TOTAL + 3) & ~3) * sizeof(float), st, T);
        }
        T.bin_accumulate = base > 0 ? 1 : 0;
        if (base > 0) HIP_TRY(hipMemsetAsync(T.bin_count, 0, ((size_t)T.nbins + 1) * sizeof(int32_t), st));      // the first chunk's counters: the zero arena
        hipLaunchKernelGGL(train_color_bwd_kernel, dim3(tr_grid(t, passes, kTrainWavesBwd, 1)), dim3(kTrainWavesBwd * 64), lds_b, st, T);
        // colour-plane scatter: order the chunk's (plane, sample) pairs by bin, then one wave per unit (ngf_train.hpp section 5b); the
        // single-workgroup prefix stays on this stream, ahead of the fork (beside three full-device kernels it took 0.13 ms instead of 0.01)
        hipLaunchKernelGGL(train_bin_prefix_kernel, dim3(1), dim3(1024), 0, st, T);
        if (fork) {
            HIP_TRY(hipEventRecord(t->ev_fork, st));
            HIP_TRY(hipStreamWaitEvent(sb, t->ev_fork, 0));
            HIP_TRY(hipStreamWaitEvent(sx, t->ev_fork, 0));
        }
        hipLaunchKernelGGL(train_bin_perm_kernel, dim3(tr_grid(t, 3 * (int64_t)T.chunk_n, 256)), dim3(256), 0, sb, T);
#ifdef NGF_EXPERIMENTS
        if (A.ablate & (1 << 18)) hipLaunchKernelGGL(train_bin_scatter_mfma_kernel, dim3(3 * t->num_cus), dim3(256), 0, sb, T);      // the matrix-pipe version (slower)
        else
#endif
        hipLaunchKernelGGL(train_bin_scatter_kernel, dim3(2 * t->num_cus), dim3(256), 0, sb, T);
        {   // one (texel, channel) per thread and two dependent loads each: as many workgroups as the largest plane has items (a few per thread
            // left the kernel waiting on memory latency: 40 us -> see profiles/r03_train_R1_kernel_stats.txt)
            int64_t items = 0;
            for (int p = 0; p < 3; ++p) items = std::max<int64_t>(items, (int64_t)(d.plane_h[p] + 2) * (d.plane_w[p] + 2) * 48);
            const int gx = (int)std::min<int64_t>((items + 255) / 256, (int64_t)64 * t->num_cus);
            hipLaunchKernelGGL(train_bin_gather_kernel, dim3(gx, 3), dim3(256), 0, sb, T);
        }
        const int rows = T.chunk_n;
        int xg = (rows + 31) / 32;                      // 32-sample chunks; at most two workgroups per CU walk them
        if (xg > 2 * t->num_cus) xg = 2 * t->num_cus;
        // heaviest first: M = Delta1^T F (train_unfold_kernel turns it into d W1[:, :144] and d basis), d W2, the 15 view columns of d W1, d W3
        XtyAll G;
        G.X[0] = T.D1; G.ldx[0] = 64; G.Y[0] = T.F;  G.ldy[0] = 144; G.mvalid[0] = 64; G.nvalid[0] = 144; G.out[0] = T.M; G.ldo[0] = 144;
        G.X[1] = T.D2; G.ldx[1] = 64; G.Y[1] = T.H1; G.ldy[1] = 64;  G.mvalid[1] = 64; G.nvalid[1] = 64;  G.out[1] = t->g_dense[TP_W2]; G.ldo[1] = 64;
        G.X[2] = T.D1; G.ldx[2] = 64; G.Y[2] = T.V;  G.ldy[2] = 16;  G.mvalid[2] = 64; G.nvalid[2] = 15;  G.out[2] = t->g_dense[TP_W1] + 144; G.ldo[2] = 159;
        G.X[3] = T.D3; G.ldx[3] = 16; G.Y[3] = T.H2; G.ldy[3] = 64;  G.mvalid[3] = 3;  G.nvalid[3] = 64;  G.out[3] = t->g_dense[TP_W3]; G.ldo[3] = 64;
        G.rows = rows; G.rows_dev = cnt;
        hipLaunchKernelGGL(xty_all_kernel, dim3(xg, 4), dim3(256), 0, sx, G);
        if (base + t->chunk >= list_len) hipLaunchKernelGGL(train_unfold_kernel, dim3(96), dim3(256), 0, sx, T, t->g_dense[TP_W1], t->g_dense[TP_BASIS]);
        if (fork) {
            // both events are recorded before an error can return: the wrapper's rescue join waits on events of THIS step
            const hipError_t e0 = hipEventRecord(t->ev_join[0], sx), e1 = hipEventRecord(t->ev_join[1], sb);
            fs.chains = true;
            HIP_TRY(e0);
            HIP_TRY(e1);
        }
    }
    if (list_len <= 0) {        // no active sample and the host knows it: nothing wrote the colour planes' gradients
        for (int p = 0; p < 3; ++p) HIP_TRY(hipMemsetAsync(t->g_a[p], 0, (size_t)(d.plane_h[p] + 2) * (d.plane_w[p] + 2) * 48 * sizeof(float), st));
        hipLaunchKernelGGL(train_unfold_kernel, dim3(96), dim3(256), 0, st, T, t->g_dense[TP_W1], t->g_dense[TP_BASIS]);
    }
    // (Tried: the D_p share <true, false> beside the colour backward and the gauge share <false, true> after it -- each half takes as long
    // as the whole, 0.31 ms: the kernel is bound by the dependent chain of an item, not by what it scatters.)
    hipLaunchKernelGGL((train_density_bwd_kernel<true, true>), dgrid, dim3(256), 0, st, T);
    UnblockArgs U;
    FinishArgs FA;
    FA.wd = d.dens_w; FA.g_wd = t->g_dense[TP_DENS_W];
    for (int p = 0; p < 3; ++p) {
        FA.D[p] = t->d_d[p]; FA.tex16[p] = t->tex_d[p]; FA.w2[p] = d.plane_w[p] + 2; FA.bw[p] = T.d_bw[p];
        FA.texels[p] = (int64_t)(d.plane_h[p] + 2) * (d.plane_w[p] + 2); FA.g_dens[p] = t->g_d[p];
        U.src[p] = t->g_gb[p]; U.dst[p] = t->g_g[p]; U.w2[p] = d.gauge_w[p] + 2; U.h2[p] = d.gauge_h[p] + 2; U.bw[p] = T.g_bw[p];
    }
    U.g_bd = T.g_bd;
    U.loss_src = T.loss; U.loss_dst = rgb_loss; U.loss_len = loss_len; U.inv_count = 1.0 / (3.0 * (double)n); U.overflow = t->overflow;        // the loss travels with the last kernel of the caller's stream (a copy of 8 bytes is a launch of its own)
    hipLaunchKernelGGL(train_density_finish_kernel, dim3(128, 3), dim3(256), 0, st, FA);
    hipLaunchKernelGGL(train_unblock_gauge_kernel, dim3(128, 3), dim3(256), 0, st, U);
    if (int jrc = join()) return jrc;
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

// error exit after a fork: the caller's stream still waits for what the aux streams hold (best effort: the error being reported stands)
static int rescue_join(ngf_trainer *t, int rc, const ForkState &fs, void *hip_stream)
{
    if (rc != NGF_OK && t && (fs.fold || fs.chains)) {
        hipStream_t st = (hipStream_t)hip_stream;
        (void)hipStreamWaitEvent(st, t->ev_join[0], 0);
        if (fs.chains) (void)hipStreamWaitEvent(st, t->ev_join[1], 0);
    }
    return rc;
}

static int train_backward_joined(ngf_trainer *t, const float *rays, const float *rgb_train, const float *jitter, int64_t n, int32_t n_samples,
                                 int32_t white_bg, int32_t gauge_on, double *rgb_loss, int32_t loss_len, int64_t *n_active_host, void *hip_stream)
{
    if (!t || !rays || !rgb_train || !rgb_loss) return fail(NGF_E_ARG, "ngf_train_backward: null argument");
    if (loss_len < 1) return fail(NGF_E_ARG, "ngf_train_backward2: loss_len=%d (1 = the sum of squared residuals, 2 = sum and mean)", loss_len);
    ForkState fs;
    ngf_trainer::Pending P;                 // the fused call keeps its state to itself: a pending ngf_train_forward is not disturbed ...
    t->pending.valid = false;               // ... but the trainer's per-sample buffers are: its backward must re-run the forward
    int rc = train_forward_part(t, rays, jitter, n, n_samples, white_bg, gauge_on, n_active_host, hip_stream, fs, P);
    if (rc == NGF_OK) {
        P.T.target = rgb_train;
        hipLaunchKernelGGL(train_composite_bwd_kernel<0>, comp_grid(n), dim3(64), 0, (hipStream_t)hip_stream, P.T);
        rc = train_backward_part(t, P, rgb_loss, loss_len, hip_stream, fs);
    }
    return rescue_join(t, rc, fs, hip_stream);
}

// ---- the two-call form: the torch.autograd boundary of Base.forward(is_train=True) (TriPlane/main.py:272-296) -----------------------------
extern "C" int ngf_train_forward(ngf_trainer *t, const float *rays, const float *jitter, int64_t n, int32_t n_samples, int32_t white_bg,
                                 int32_t gauge_on, float *rgb_map, float *depth_map, int64_t *n_active_host, int64_t *ticket, void *hip_stream)
{
    if (!t || !rays || !rgb_map || !depth_map || !ticket) return fail(NGF_E_ARG, "ngf_train_forward: null argument");
    ForkState fs;
    int rc = train_forward_part(t, rays, jitter, n, n_samples, white_bg, gauge_on, n_active_host, hip_stream, fs, t->pending);
    if (rc == NGF_OK) {
        t->pending.T.rgb_out = rgb_map; t->pending.T.depth_out = depth_map;
        hipLaunchKernelGGL(train_composite_bwd_kernel<1>, comp_grid(n), dim3(64), 0, (hipStream_t)hip_stream, t->pending.T);
        rc = hipGetLastError() == hipSuccess ? NGF_OK : fail(NGF_E_HIP, "ngf_train_forward: launch failed");
        t->pending.ticket = *ticket = ++t->tickets;
    }
    if (rc != NGF_OK) t->pending.valid = false;
    return rescue_join(t, rc, fs, hip_stream);
}

extern "C" int ngf_train_backward_grad(ngf_trainer *t, int64_t ticket, const float *d_rgb_map, void *hip_stream)
{
    if (!t || !d_rgb_map) return fail(NGF_E_ARG, "ngf_train_backward_grad: null argument");
    if (!t->pending.valid || t->pending.ticket != ticket)
        return fail(NGF_E_STALE, "ngf_train_backward_grad: ticket %lld is not the trainer's last forward (%s) -- another forward or a fused step used the "
                    "trainer's buffers since; run ngf_train_forward again", (long long)ticket, t->pending.valid ? "a newer one is pending" : "none is pending");
    ForkState fs;
    ngf_trainer::Pending &P = t->pending;
    P.T.d_rgb = d_rgb_map;
    hipLaunchKernelGGL(train_composite_bwd_kernel<2>, comp_grid(P.n), dim3(64), 0, (hipStream_t)hip_stream, P.T);
    const int rc = train_backward_part(t, P, nullptr, 0, hip_stream, fs);
    P.valid = false;                        // the gradients are in the trainer's buffers (ngf_train_get_grad); the per-sample state is spent
    return rescue_join(t, rc, fs, hip_stream);
}

// ABI 3: the loss buffer's length travels with the call (loss_len = 2: [0] the sum of squared residuals, [1] their mean)
extern "C" int ngf_train_backward2(ngf_trainer *t, const float *rays, const float *rgb_train, const float *jitter, int64_t n, int32_t n_samples,
                                   int32_t white_bg, int32_t gauge_on, double *rgb_loss, int32_t loss_len, int64_t *n_active_host, void *hip_stream)
{
    return train_backward_joined(t, rays, rgb_train, jitter, n, n_samples, white_bg, gauge_on, rgb_loss, loss_len, n_active_host, hip_stream);
}

// The ABI-1 entry point keeps the ABI-1 contract: rgb_loss is ONE double (the sum of squared residuals).  (ABI 2 wrote two doubles through
// this symbol: a caller built against ABI 1 got an 8-byte out-of-bounds device write.)
extern "C" int ngf_train_backward(ngf_trainer *t, const float *rays, const float *rgb_train, const float *jitter, int64_t n, int32_t n_samples,
                                  int32_t white_bg, int32_t gauge_on, double *rgb_loss, int64_t *n_active_host, void *hip_stream)
{
    return train_backward_joined(t, rays, rgb_train, jitter, n, n_samples, white_bg, gauge_on, rgb_loss, 1, n_active_host, hip_stream);
}

extern "C" int ngf_train_get_active(ngf_trainer *t, int64_t n, int32_t *out, void *hip_stream)
{
    if (!t || !out || n <= 0 || n > t->d.max_rays) return fail(NGF_E_ARG, "ngf_train_get_active: bad argument");
    HIP_TRY(hipMemcpyAsync(out, t->proto.offset + n, sizeof(int32_t), hipMemcpyDeviceToDevice, (hipStream_t)hip_stream));
    return NGF_OK;
}

extern "C" int ngf_train_get_grad(ngf_trainer *t, int32_t which, float *out, void *hip_stream)
{
    if (!t || !out || which < 0 || which >= TP_COUNT) return fail(NGF_E_ARG, "ngf_train_get_grad: bad argument");
    hipStream_t st = (hipStream_t)hip_stream;
    if (which < 3) {
        const int p = which;
        hipLaunchKernelGGL(unpack_plane_kernel, dim3(1024), dim3(256), 0, st, (const float *)t->g_d[p], 16, (const float *)t->g_a[p], 64, t->d.plane_h[p],
                           t->d.plane_w[p], out);
    } else if (which < 6) {
        const int p = which - 3;
        hipLaunchKernelGGL(unpack_plane_kernel, dim3(256), dim3(256), 0, st, (const float *)t->g_g[p], 2, (const float *)nullptr, 2, t->d.gauge_h[p],
                           t->d.gauge_w[p], out);
    } else {
        HIP_TRY(hipMemcpyAsync(out, t->g_dense[which], (size_t)t->dense_n[which] * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

// ABI 5: every requested gradient in ONE call (out[k] NULL = not wanted): three tiled transposes for the planes, the gauge planes, one launch for
// the nine MLP parameters -- the autograd path (ngf_amd.train.RenderGrad.backward) made fifteen calls of ngf_train_get_grad per step.
extern "C" int ngf_train_get_grads(ngf_trainer *t, float *const *out, void *hip_stream)
{
    if (!t || !out) return fail(NGF_E_ARG, "ngf_train_get_grads: null argument");
    hipStream_t st = (hipStream_t)hip_stream;
    for (int p = 0; p < 3; ++p) {
        if (out[p]) {
            const int H = t->d.plane_h[p], W = t->d.plane_w[p];
            hipLaunchKernelGGL((unpack_plane_tiled_kernel<64, 16>), dim3(H * ((W + 63) / 64)), dim3(256), 0, st, (const float *)t->g_d[p], (const float *)t->g_a[p], H, W, out[p]);
        }
        if (out[3 + p])
            hipLaunchKernelGGL(unpack_plane_kernel, dim3(256), dim3(256), 0, st, (const float *)t->g_g[p], 2, (const float *)nullptr, 2, t->d.gauge_h[p], t->d.gauge_w[p], out[3 + p]);
    }
    CopyDenseAll D;
    int32_t at = 0;
    for (int j = 0; j < kDenseParams; ++j) {
        const int k = TP_DENS_W + j;
        D.src[j] = t->g_dense[k]; D.dst[j] = out[k]; D.begin[j] = at;
        if (out[k]) at += (int32_t)t->dense_n[k];
    }
    D.begin[kDenseParams] = at;
    if (at > 0) hipLaunchKernelGGL(copy_dense_all_kernel, dim3((at + 255) / 256), dim3(256), 0, st, D);
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

// ABI 5: torch.optim.Adam's update for the trainer's fifteen parameters from the CALLER'S gradient tensors and moments (reference layouts: what
// p.grad, state['exp_avg'], state['exp_avg_sq'] are after total_loss.backward()), planes in one pass that also writes the trainer's packed
// copy -- the next ngf_train_forward reads it without a re-pack.  step_count[k] <= 0 or grad[k] NULL = parameter k is left alone.  The arithmetic
// is ngf_train_adam's (adam_one); no L1 term is added here: the caller's loss put it into the gradient.  Works on trainers with or without
// their own moments.  ngf_amd.optim.Adam is the Python face (TriPlane/main.py:234-242,294-302 unchanged).
extern "C" int ngf_train_adam_ext(ngf_trainer *t, const float *const *grad, float *const *exp_avg, float *const *exp_avg_sq, const int32_t *step_count,
                                  const float *lr, float beta1, float beta2, float eps, void *hip_stream)
{
    if (!t || !grad || !exp_avg || !exp_avg_sq || !step_count || !lr) return fail(NGF_E_ARG, "ngf_train_adam_ext: null argument");
    const ngf_train_desc &d = t->d;
    hipStream_t st = (hipStream_t)hip_stream;
    auto args = [&](int k) {
        AdamArgs a;
        a.lr = lr[k]; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.l1 = 0.0f;
        a.bc1 = (float)(1.0 - pow((double)beta1, (double)step_count[k]));
        a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step_count[k]));
        return a;
    };
    for (int k = 0; k < TP_COUNT; ++k)
        if (step_count[k] > 0 && grad[k] && (!exp_avg[k] || !exp_avg_sq[k])) return fail(NGF_E_ARG, "ngf_train_adam_ext: parameter %d has a gradient but no moments", k);
    const bool fork = !(knob(KNOB_ABLATE) > 0 && (knob(KNOB_ABLATE) & (1 << 19)));
    const bool big[3] = {step_count[0] > 0 && grad[0], step_count[1] > 0 && grad[1], step_count[2] > 0 && grad[2]};
    const bool forked = fork && (big[1] || big[2]);
    if (forked) {
        HIP_TRY(hipEventRecord(t->ev_fork, st));
        HIP_TRY(hipStreamWaitEvent(t->aux[0], t->ev_fork, 0));
        HIP_TRY(hipStreamWaitEvent(t->aux[1], t->ev_fork, 0));
    }
    for (int p : {1, 2, 0}) {
        if (!big[p]) continue;
        const int H = d.plane_h[p], W = d.plane_w[p];
        hipStream_t sk = (forked && p > 0) ? t->aux[p - 1] : st;
        // a stale packed copy (ngf_train_params_changed since the last forward) is simply rewritten: the kernel stores every interior texel and the
        // zero border was written when the copy was first packed
        hipLaunchKernelGGL((adam_plane_kernel<64, 16, true>), dim3(H * ((W + 63) / 64)), dim3(256), 0, sk, d.plane[p], exp_avg[p], exp_avg_sq[p], H, W,
                           (const float *)nullptr, (const float *)nullptr, t->tex_d[p], t->tex_a[p], args(p), (const int32_t *)nullptr, grad[p]);
        t->tex_fresh[p] = true;
    }
    for (int p = 0; p < 3; ++p) {
        const int k = 3 + p;
        if (!(step_count[k] > 0 && grad[k])) continue;
        hipLaunchKernelGGL((adam_plane_kernel<2, 2, true>), dim3(d.gauge_h[p] * ((d.gauge_w[p] + 63) / 64)), dim3(256), 0, st, d.gauge[p], exp_avg[k], exp_avg_sq[k],
                           d.gauge_h[p], d.gauge_w[p], (const float *)nullptr, (const float *)nullptr, t->tex_g[p], (float *)nullptr, args(k), (const int32_t *)nullptr, grad[k]);
        t->tex_fresh[k] = true;
    }
    float *params[TP_COUNT] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, d.dens_w, d.dens_b, d.basis, d.w1, d.b1, d.w2, d.b2, d.w3, d.b3};
    AdamDenseAll D;
    int32_t at = 0;
    for (int j = 0; j < kDenseParams; ++j) {
        const int k = TP_DENS_W + j;
        D.p[j] = params[k]; D.m[j] = exp_avg[k]; D.v[j] = exp_avg_sq[k]; D.g[j] = grad[k];
        D.begin[j] = at;
        D.a[j] = args(k);
        if (step_count[k] > 0 && grad[k]) at += (int32_t)t->dense_n[k];
        else { D.a[j].bc1 = 1.0f; D.a[j].bc2_sqrt = 1.0f; }          // empty segment
    }
    D.begin[kDenseParams] = at;
    D.skip = nullptr;
    if (at > 0) hipLaunchKernelGGL(adam_dense_all_kernel, dim3((at + 255) / 256), dim3(256), 0, st, D);
    if (forked)
        for (int j = 0; j < 2; ++j) {
            HIP_TRY(hipEventRecord(t->ev_join[j], t->aux[j]));
            HIP_TRY(hipStreamWaitEvent(st, t->ev_join[j], 0));
        }
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

extern "C" int ngf_train_adam(ngf_trainer *t, int32_t which, int32_t step_count, float lr, float beta1, float beta2, float eps, float l1_weight,
                              void *hip_stream)
{
    if (!t || which < 0 || which >= TP_COUNT || step_count < 1) return fail(NGF_E_ARG, "ngf_train_adam: bad argument");
    if (!t->has_adam) return fail(NGF_E_ARG, "ngf_train_adam: the trainer was made without Adam moments (ngf_train_desc.exp_avg / exp_avg_sq all NULL)");
    hipStream_t st = (hipStream_t)hip_stream;
    const ngf_train_desc &d = t->d;
    AdamArgs a;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps;
    a.bc1 = (float)(1.0 - pow((double)beta1, (double)step_count));
    a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step_count));
    a.l1 = 0.0f;
    if (which < 3) {
        const int p = which, H = d.plane_h[p], W = d.plane_w[p];
        a.l1 = l1_weight / (float)((int64_t)64 * H * W);             // d/dp of l1_weight * mean(|p|)
        if (!t->tex_fresh[p]) return fail(NGF_E_ARG, "ngf_train_adam: the planes changed (ngf_train_params_changed) and no backward has re-packed them");
        hipLaunchKernelGGL((adam_plane_kernel<64, 16>), dim3(H * ((W + 63) / 64)), dim3(256), 0, st, d.plane[p], d.exp_avg[which], d.exp_avg_sq[which], H, W,
                           (const float *)t->g_d[p], (const float *)t->g_a[p], t->tex_d[p], t->tex_a[p], a, (const int32_t *)t->overflow);
    } else if (which < 6) {
        const int p = which - 3;
        if (!t->tex_fresh[which]) return fail(NGF_E_ARG, "ngf_train_adam: the planes changed (ngf_train_params_changed) and no backward has re-packed them");
        hipLaunchKernelGGL((adam_plane_kernel<2, 2>), dim3(d.gauge_h[p] * ((d.gauge_w[p] + 63) / 64)), dim3(256), 0, st, d.gauge[p], d.exp_avg[which],
                           d.exp_avg_sq[which], d.gauge_h[p], d.gauge_w[p], (const float *)t->g_g[p], (const float *)nullptr, t->tex_g[p], (float *)nullptr, a, (const int32_t *)t->overflow);
    } else {
        float *params[TP_COUNT] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, d.dens_w, d.dens_b, d.basis, d.w1, d.b1, d.w2, d.b2, d.w3, d.b3};
        hipLaunchKernelGGL(adam_dense_kernel, dim3(64), dim3(256), 0, st, params[which], (const float *)t->g_dense[which], d.exp_avg[which],
                           d.exp_avg_sq[which], t->dense_n[which], a, (const int32_t *)t->overflow);
    }
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

// Speculative rows (ngf_train_desc::chunk_samples < 0): how many steps since the trainer was made had more active samples than activation rows
// (their Adam updates were skipped on the device) and the trainer's row count.  Synchronises the stream.
extern "C" int ngf_train_overflow_count(ngf_trainer *t, int64_t *count_host, int64_t *rows_host, void *hip_stream)
{
    if (!t || !count_host) return fail(NGF_E_ARG, "ngf_train_overflow_count: null argument");
    int32_t v[2] = {0, 0};
    HIP_TRY(hipMemcpyAsync(v, t->overflow, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)hip_stream));
    HIP_TRY(hipStreamSynchronize((hipStream_t)hip_stream));
    *count_host = v[1];
    if (rows_host) *rows_host = t->chunk;
    return NGF_OK;
}

// debug: the section clocks of the colour backward (ngf_debug_set("ablate", 1 << 20)), summed over waves since the last call
extern "C" int ngf_train_debug_sections(ngf_trainer *t, uint64_t *out16)
{
    if (!t || !out16) return fail(NGF_E_ARG, "ngf_train_debug_sections: null argument");
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(hipMemcpy(out16, t->proto.prof, 16 * sizeof(uint64_t), hipMemcpyDeviceToHost));
    HIP_TRY(hipMemset(t->proto.prof, 0, 16 * sizeof(uint64_t)));
    return NGF_OK;
}

extern "C" int ngf_train_params_changed(ngf_trainer *t)
{
    if (!t) return fail(NGF_E_ARG, "ngf_train_params_changed: null trainer");
    for (bool &f : t->tex_fresh) f = false;
    return NGF_OK;
}

extern "C" int ngf_train_adam_all(ngf_trainer *t, const int32_t *step_count, const float *lr, float beta1, float beta2, float eps, float l1_weight,
                                  void *hip_stream)
{
    if (!t || !step_count || !lr) return fail(NGF_E_ARG, "ngf_train_adam_all: null argument");
    const ngf_train_desc &d = t->d;
    // the three (plane, gauge plane) pairs are independent streams of reads and writes, none of which reaches the HBM rate alone: planes 1
    // and 2 go to the trainer's own streams (ablate bit 1 << 19: all on the caller's)
    hipStream_t st = (hipStream_t)hip_stream;
    const bool fork = !(knob(KNOB_ABLATE) > 0 && (knob(KNOB_ABLATE) & (1 << 19)));
    if (fork) {
        HIP_TRY(hipEventRecord(t->ev_fork, st));
        HIP_TRY(hipStreamWaitEvent(t->aux[0], t->ev_fork, 0));
        HIP_TRY(hipStreamWaitEvent(t->aux[1], t->ev_fork, 0));
    }
    // planes 1 and 2 leave first; the caller's stream takes the small updates (gauge planes, the MLP parameters below) and plane 0
    for (int k : {1, 2, 3, 4, 5})
        if (step_count[k] > 0) {
            hipStream_t sk = (fork && (k == 1 || k == 2)) ? t->aux[k - 1] : st;
            const int rc = ngf_train_adam(t, k, step_count[k], lr[k], beta1, beta2, eps, l1_weight, (void *)sk);
            if (rc != NGF_OK) return rc;
        }
    float *params[TP_COUNT] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, d.dens_w, d.dens_b, d.basis, d.w1, d.b1, d.w2, d.b2, d.w3, d.b3};
    AdamDenseAll D;
    int32_t at = 0;
    for (int j = 0; j < kDenseParams; ++j) {
        const int k = TP_DENS_W + j;
        D.p[j] = params[k]; D.m[j] = d.exp_avg[k]; D.v[j] = d.exp_avg_sq[k]; D.g[j] = t->g_dense[k];
        D.begin[j] = at;
        AdamArgs &a = D.a[j];
        a.lr = lr[k]; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.l1 = 0.0f; a.bc1 = 1.0f; a.bc2_sqrt = 1.0f;
        if (step_count[k] > 0) {
            a.bc1 = (float)(1.0 - pow((double)beta1, (double)step_count[k]));
            a.bc2_sqrt = (float)sqrt(1.0 - pow((double)beta2, (double)step_count[k]));
            at += (int32_t)t->dense_n[k];
        }
    }
    D.begin[kDenseParams] = at;
    D.skip = t->overflow;
    if (at > 0) hipLaunchKernelGGL(adam_dense_all_kernel, dim3((at + 255) / 256), dim3(256), 0, (hipStream_t)hip_stream, D);
    if (step_count[0] > 0) {
        const int rc = ngf_train_adam(t, 0, step_count[0], lr[0], beta1, beta2, eps, l1_weight, hip_stream);
        if (rc != NGF_OK) return rc;
    }
    if (fork) {
        for (int j = 0; j < 2; ++j) {
            HIP_TRY(hipEventRecord(t->ev_join[j], t->aux[j]));
            HIP_TRY(hipStreamWaitEvent(st, t->ev_join[j], 0));
        }
    }
    HIP_TRY(hipGetLastError());
    return NGF_OK;
}

