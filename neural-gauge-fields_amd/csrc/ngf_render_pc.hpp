// ngf_render_pc.hpp -- the TriPlane ray-march kernel with SPECIALISED waves: march waves feed shade waves through LDS queues.
//
// Why: in the fused kernel (ngf_render.hpp) every wave alternates between marching (gather-latency / VALU bound) and
// shading (matrix-pipe bound); rocprofv3 shows the two phases barely overlap (R1 frame: matrix pipe 52 % busy, TA 62 %,
// 59 % of the wave cycles waiting to issue).  Here the roles are split inside one persistent workgroup per CU:
//
//   waves 0 .. NM-1   MARCH  a tile of TW rays x K = 64 / TW consecutive steps per iteration: sample_ray, alpha-mask test, gauge,
//                     density, raw2alpha (FieldBase.py:118-137, 251-288; Field.py:53-91); the K lanes of a ray are ADJACENT
//                     lanes (lane = ray * K + segment), so transmittance / acc / depth chain from step to step through DPP
//                     row_shr:1 moves (no LDS round trips) -- still the sequential cumprod of the reference, bit for bit.
//                     Active samples (weight > thr) are appended to the wave's LDS ring as 32-byte records.
//   waves NM .. NM+NS-1  SHADE  NS / 4 per SIMD: each serves NM / NS rings, takes 16 records at a time and runs the colour MLP on the
//                     matrix pipe (ngf_shade16.hpp: compute_rgb + rgb_decoder, Field.py:93-105, networks.py:25-32), adds the
//                     weighted colours to the tile's per-ray sums IN RECORD ORDER (= sample order: deterministic, no atomics,
//                     bit-identical to the fused kernel) and writes rgb_map when the tile's last record has been consumed.
//
// The queues are single-producer / single-consumer rings in LDS with monotonically increasing head / tail words; tile boundaries
// travel out of band (tile-info slots, double buffered), so a batch never spans two tiles.  All waiting is on LDS words written
// by waves of the SAME workgroup (resident together): no inter-workgroup protocol.
#pragma once
#include "ngf_render.hpp"

namespace ngf {

constexpr int kPcRing = 128;                // records per march wave's ring
constexpr int kPcInfoWords = 16;            // tile-info slot: {end index, base lo, base hi, -, acc[<=8], ...}
constexpr unsigned kPcWatchdog = 1u << 24;  // consecutive s_sleep polls without progress (~1 s) before a wave gives up: a protocol bug
                                            // must end in a trapped launch (hipErrorLaunchFailure at the next sync), never in a hung GPU

template <int NM>
struct PcLds {                              // floats after the MLP image
    static constexpr int RING = 0;                                  // [NM][kPcRing][kRecFloats]
    static constexpr int VTAB = RING + NM * kPcRing * kRecFloats;   // [NM][2][8 rays][kFoldStride]: b1 + W1[:, view] . view per ray, accumulator order
    static constexpr int CTRL = VTAB + NM * 2 * 8 * kFoldStride;             // ints: tail[NM], head[NM], produced[NM], done[NM], finished[NM], info[NM][2][16], csum[NM][32]
    static constexpr int CTRL_WORDS = 5 * NM + NM * 2 * kPcInfoWords + NM * 32;       // + per-ring colour sums [8 rays][3] (+ pad)
    static constexpr int TOTAL = CTRL + CTRL_WORDS;
};

// Control words are accessed through EXPLICIT LDS (address space 3) pointers: a `volatile` access through a generic pointer is not
// rewritten by the address-space inference and compiles to flat_load / flat_store ... sc0 sc1 + s_waitcnt vmcnt(0) -- hundreds of
// cycles per word (measured: 2.4k cycles per pass for six such accesses).
typedef __attribute__((address_space(3))) unsigned lds_u32;
typedef __attribute__((address_space(3))) float lds_f32;
__device__ __forceinline__ unsigned lds_load(lds_u32 *p) { return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_store(lds_u32 *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP); }
// plain (relaxed) accesses to words only this wave writes, or that an acquire / release above already orders
__device__ __forceinline__ unsigned lds_peek(lds_u32 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_poke(lds_u32 *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ float lds_peekf(lds_u32 *p) { return __uint_as_float(lds_peek(p)); }
__device__ __forceinline__ void lds_pokef(lds_u32 *p, float v) { lds_poke(p, __float_as_uint(v)); }

template <int CTRL, int BANK = 0xf>
__device__ __forceinline__ float dpp(float old, float src)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, 0xf, BANK, false));
}
// value of the group's last lane (segment K-1) in all K lanes of the group
template <int K>
__device__ __forceinline__ float group_last(float v)
{
    if constexpr (K == 16) return dpp<0x15F>(v, v);                              // row_newbcast:15
    else if constexpr (K == 8) return dpp<0x15F, 0xC>(dpp<0x157, 0x3>(v, v), v);     // lanes 0-7 <- lane 7, lanes 8-15 <- lane 15
    else return __shfl(v, (int)(threadIdx.x & 63) | (K - 1));
}

template <int E>
__device__ __forceinline__ void collect_entry(int own, float wr, float wg, float wb, int lane, float &cr, float &cg, float &cb)
{
    constexpr int BC = 0x150 + E;           // row_newbcast:E
    const int oe = __builtin_amdgcn_update_dpp(own, own, BC, 0xf, 0xf, false);
    const float m = oe == lane ? 1.0f : 0.0f;
    cr = fmaf(dpp<BC>(wr, wr), m, cr);
    cg = fmaf(dpp<BC>(wg, wg), m, cg);
    cb = fmaf(dpp<BC>(wb, wb), m, cb);
    if constexpr (E + 1 < kBatch16) collect_entry<E + 1>(own, wr, wg, wb, lane, cr, cg, cb);
}
__device__ __forceinline__ void collect16(int own, float wr, float wg, float wb, int lane, float &cr, float &cg, float &cb)
{
    collect_entry<0>(own, wr, wg, wb, lane, cr, cg, cb);
}

// P: a TriPlanePolicy (sigma / shade / fold_view).  NM march + NS shade waves per workgroup; TW = rays per tile (8 or 4).
template <typename P, int NM, int NS, int TW>
__global__ void __launch_bounds__((NM + NS) * 64) render_pc_kernel(const RenderArgs A)
{
    static_assert(NM % NS == 0, "every shade wave serves NM / NS rings");
    static_assert(TW == 8 || TW == 4, "tiles of 8 or 4 rays");
    constexpr int K = 64 / TW, LOGK = (K == 8 ? 3 : 4), QPS = NM / NS, BATCH = kBatch16;
    using L = PcLds<NM>;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float *base_f = smem + ((A.blob_floats + 3) & ~3);
    lds_u32 *ctrl = (lds_u32 *)reinterpret_cast<unsigned *>(base_f + L::CTRL);
    for (int i = threadIdx.x; i < A.blob_floats; i += blockDim.x) smem[i] = A.blob[i];
    for (int i = threadIdx.x; i < L::CTRL_WORDS; i += blockDim.x) ctrl[i] = 0u;
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    lds_u32 *q_tail = ctrl, *q_head = ctrl + NM, *q_produced = ctrl + 2 * NM, *q_done = ctrl + 3 * NM, *q_finished = ctrl + 4 * NM;
    lds_u32 *q_info = ctrl + 5 * NM, *q_csum = ctrl + 5 * NM + NM * 2 * kPcInfoWords;
    const int S = A.S;

    if (wave < NM) {
        // =========================================== MARCH =====================================================================
        const int m = wave;
        float *ring = base_f + L::RING + m * kPcRing * kRecFloats;
        float *vtab = base_f + L::VTAB + m * 2 * 8 * kFoldStride;
        lds_u32 *info = q_info + m * 2 * kPcInfoWords;
        const unsigned long long lt_mask = (1ull << lane) - 1ull;
        const int seg = lane & (K - 1), rl = lane >> LOGK;
        unsigned tail = 0, head_seen = 0, t_count = 0;
        unsigned long long st_valid = 0, st_active = 0, st_rays = 0;
        [[maybe_unused]] unsigned long long prof_march = 0, prof_wait = 0;          // PROFILE: cycles in march iterations / waiting for ring space
        for (;;) {
            unsigned int tile = 0;
            if (lane == 0) tile = atomicAdd(A.tile_counter, 1u);
            tile = __builtin_amdgcn_readfirstlane(tile);
            const int64_t base = (int64_t)tile * TW;
            if (base >= A.n) break;
            const int64_t ray = base + rl;
            const bool live = ray < A.n;
            const int64_t rr = live ? ray : A.n - 1;
            float o[3], d[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { o[k] = A.rays[rr * 6 + k]; d[k] = A.rays[rr * 6 + 3 + k]; }
            const float jit = A.jitter ? A.jitter[rr] : 0.0f;
            float tmin = -INFINITY;          // sample_ray (FieldBase.py:122-125)
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                float vec = (d[k] == 0.0f) ? 1e-6f : d[k];
                float ra = (A.a1[k] - o[k]) / vec, rb = (A.a0[k] - o[k]) / vec;
                tmin = fmaxf(tmin, fminf(ra, rb));
            }
            tmin = fminf(fmaxf(tmin, A.near_), A.far_);

            // the slot pair (tile info, view table) of tile t is free once tile t-2 has been written out by the shade wave
            const int par = (int)(t_count & 1u);
            for (unsigned spins = 0; (int)(t_count - lds_load(q_finished + m)) >= 2; ++spins) {
                if (spins > kPcWatchdog) __builtin_trap();
                __builtin_amdgcn_s_sleep(2);
            }
            {   // per-ray view fold: lane (s, kq) evaluates b1 + W1[:, view] . view of ray s with the 16 MFMAs a pass would spend on it
                const int s = lane & 15, kq = lane >> 4;
                const int src = (s < TW ? s : 0) << LOGK;
                const float od[3] = {__shfl(d[0], src), __shfl(d[1], src), __shfl(d[2], src)};
                float v[16];
                view_inputs(od, v);
                f32x4 v4;
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = kq == 0 ? v[e] : (kq == 1 ? v[4 + e] : (kq == 2 ? v[8 + e] : v[12 + e]));
                P::fold_view_regs(smem, v4, vtab + par * 8 * kFoldStride, TW, lane);
            }
            float T = 1.0f, acc = 0.0f, dep = 0.0f;
            int i = 0;
            while (i < S) {
                [[maybe_unused]] unsigned long long t_it = 0;
                if constexpr (P::PROFILE) t_it = __builtin_readcyclecounter();
                // ring space for the 64 records this iteration may append
                for (unsigned spins = 0; (int)(tail - head_seen) > kPcRing - 64; ++spins) {
                    head_seen = lds_load(q_head + m);
                    if ((int)(tail - head_seen) > kPcRing - 64) {
                        if (spins > kPcWatchdog) __builtin_trap();
                        __builtin_amdgcn_s_sleep(1);
                    }
                }
                if constexpr (P::PROFILE) { const unsigned long long t1 = __builtin_readcyclecounter(); prof_wait += t1 - t_it; t_it = t1; }
                const int si = i + seg;
                const float z = tmin + A.step * ((float)si + jit);
                const float zn = tmin + A.step * ((float)(si + 1) + jit);
                const float dist = (si < S - 1) ? (zn - z) : 0.0f;
                float p[3], x[3], t[6];
                bool valid = live && (si < S);
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    p[k] = o[k] + d[k] * z;
                    valid = valid && !(A.a0[k] > p[k] || p[k] > A.a1[k]);
                }
                if (A.mask.bits && valid) valid = mask_occupied(A.mask, p);
#pragma unroll
                for (int k = 0; k < 3; ++k) x[k] = (p[k] - A.a0[k]) * A.inv[k] - 1.0f;       // normalize_coord (FieldBase.py:88-89)
                // no lane has a valid sample (outside the box / empty space of the alpha mask): sigma = alpha = w = 0 for all of them
                if (!(A.ablate & 64) && !__any(valid)) { i += K; continue; }
                const float sigma = P::sigma(A, smem, valid, x, lane, t);
                st_valid += __popcll(__ballot(valid));
                // raw2alpha (FieldBase.py:12-19), chained over the K consecutive steps of the ray in step order: after round r the
                // lanes of segments <= r hold their final inputs (T, acc, depth after the previous step of their ray)
                const float alpha = 1.0f - expf(-sigma * (dist * A.dscale));
                const float f = (1.0f - alpha) + 1e-10f;
                float Tin = T, ain = acc, din = dep;
#pragma unroll
                for (int r = 1; r < K; ++r) {
                    const float wr = alpha * Tin;
                    const float To = Tin * f, ao = ain + wr, dz = wr * z;
                    const float dn = din + dz;
                    const float Tp = dpp<0x111>(To, To), ap = dpp<0x111>(ao, ao), dp = dpp<0x111>(dn, dn);      // row_shr:1
                    if (seg > 0) { Tin = Tp; ain = ap; din = dp; }
                }
                const float w = alpha * Tin;
                {
                    const float To = Tin * f, ao = ain + w, dz = w * z;
                    const float dn = din + dz;
                    T = group_last<K>(To); acc = group_last<K>(ao); dep = group_last<K>(dn);
                }
                const bool active = (w > A.thr);
                const unsigned long long am = __ballot(active);
                if (active) {
                    const int slot = (int)((tail + (unsigned)__popcll(am & lt_mask)) & (kPcRing - 1));
                    f32x4 *r = reinterpret_cast<f32x4 *>(ring + slot * kRecFloats);
                    r[0] = f32x4{__int_as_float(rl), w, t[0], t[1]};
                    r[1] = f32x4{t[2], t[3], t[4], t[5]};
                }
                if (am) {
                    tail += (unsigned)__popcll(am);
                    st_active += __popcll(am);
                    lds_store(q_tail + m, tail);                 // release: the records above are visible before the new tail
                }
                i += K;
                // exact early termination (see ngf_render.hpp): no later sample can change an output bit
                if (!(A.ablate & 32)) {
                    const float zmax = tmin + A.step * (float)(S + 1);
                    const bool done = !live || ((T < A.thr) & (zmax > 0.0f) & (T < 0x1p-26f * fminf(acc, dep / zmax)));
                    if (!__any(!done)) i = S;
                }
                if constexpr (P::PROFILE) prof_march += __builtin_readcyclecounter() - t_it;
            }
            // tile trailer: depth_map by the march wave, rgb_map by the shade wave from the tile info
            if (live && seg == 0) A.depth[ray] = dep + (1.0f - acc) * d[2];
            if (seg == 0) lds_pokef(info + par * kPcInfoWords + 4 + rl, acc);
            if (lane == 0) {
                lds_poke(info + par * kPcInfoWords + 0, tail);
                lds_poke(info + par * kPcInfoWords + 1, (unsigned)(base & 0xffffffffll));
                lds_poke(info + par * kPcInfoWords + 2, (unsigned)(base >> 32));
            }
            ++t_count;
            lds_store(q_produced + m, t_count);
            st_rays += __popcll(__ballot(live && seg == 0));
        }
        lds_store(q_done + m, 1u);
        if (A.stats && lane == 0) {
            atomicAdd(A.stats + 0, st_valid);
            atomicAdd(A.stats + 1, st_active);
            atomicAdd(A.stats + 3, st_rays);
            if constexpr (P::PROFILE) { atomicAdd(A.stats + 4, prof_march); atomicAdd(A.stats + 11, prof_wait); }
        }
    } else {
        // =========================================== SHADE =====================================================================
        // Per-ring consumer state lives in LDS (head / finished words, per-ray colour sums), so ONE copy of the pass code serves
        // all rings of this wave (an unrolled ring loop triples the code and the register pressure).
        const int sw = wave - NM;
        unsigned long long st_pass = 0;
        [[maybe_unused]] unsigned long long prof[8] = {0, 0, 0, 0, 0, 0, 0, 0};          // PROFILE: [1..6] pass sections as in render_kernel, [7] idle polling
        unsigned over_mask = 0;                 // bit qi: ring sw + NS qi is exhausted
        int n_over = 0;
        unsigned idle = 0;
        while (n_over < QPS) {
            bool progress = false;
#pragma clang loop unroll(disable)
            for (int qi = 0; qi < QPS; ++qi) {
                if (over_mask & (1u << qi)) continue;
                const int m = sw + NS * qi;
                // order matters: tail BEFORE produced (if the tile still counts as open afterwards, every record below that tail is
                // the open tile's), done BEFORE produced (done is set after the last tile was produced)
                const unsigned tail_seen = lds_load(q_tail + m);
                const unsigned was_done = lds_load(q_done + m);
                const unsigned produced = lds_load(q_produced + m);
                const unsigned head = lds_peek(q_head + m), fin = lds_peek(q_finished + m);          // written by this wave only
                lds_u32 *csum = q_csum + m * 32;
                int nb = 0;
                const int par = (int)(fin & 1u);
                lds_u32 *info = q_info + (m * 2 + par) * kPcInfoWords;
                if (produced != fin) {
                    // the tile being consumed is complete: its records end at info[0]
                    const int left = (int)(lds_peek(info) - head);
                    if (left == 0) {
                        // ---- tile finished: rgb_map (FieldBase.py:296-302) ----
                        const int64_t base = (int64_t)lds_peek(info + 1) | ((int64_t)lds_peek(info + 2) << 32);
                        if (lane < TW) {
                            const float acc = lds_peekf(info + 4 + lane);
#pragma unroll
                            for (int c = 0; c < 3; ++c) {
                                float v = lds_peekf(csum + lane * 3 + c);
                                lds_pokef(csum + lane * 3 + c, 0.0f);
                                if (A.white_bg) v = v + (1.0f - acc);
                                if (base + lane < A.n) A.rgb[(base + lane) * 3 + c] = fminf(fmaxf(v, 0.0f), 1.0f);
                            }
                        }
                        lds_store(q_finished + m, fin + 1u);
                        progress = true;
                        if (was_done && produced == fin + 1u) { over_mask |= 1u << qi; ++n_over; }
                        continue;
                    }
                    nb = left < BATCH ? left : BATCH;
                } else {
                    if (was_done) { over_mask |= 1u << qi; ++n_over; continue; }       // done was set after the last tile was produced
                    const int avail = (int)(tail_seen - head);
                    if (avail >= BATCH) nb = BATCH;
                }
                if (nb == 0) continue;
                [[maybe_unused]] unsigned long long t_sec = 0;
                if constexpr (P::PROFILE) t_sec = __builtin_readcyclecounter();
                // ---- shade nb records of ring m (all of one tile) ----
                const float *ring = base_f + L::RING + m * kPcRing * kRecFloats;
                const int s = lane & (BATCH - 1);
                const int slot = (int)((head + (unsigned)(s < nb ? s : 0)) & (kPcRing - 1));
                const f32x4 *r = reinterpret_cast<const f32x4 *>(ring + slot * kRecFloats);
                const f32x4 r0 = r[0], r1 = r[1];
                const float rec[kRecFloats] = {r0[0], r0[1], r0[2], r0[3], r1[0], r1[1], r1[2], r1[3]};
                const int owner = __float_as_int(r0[0]);
                const float *pre = base_f + L::VTAB + ((m * 2 + par) * 8 + owner) * kFoldStride;
                float c[3];
                const float od[3] = {0.0f, 0.0f, 0.0f};
                if constexpr (P::PROFILE) {
                    unsigned long long tk[5] = {0, 0, 0, 0, 0};
                    P::shade(A, smem, rec, nullptr, od, lane, c, tk, pre);
                    prof[1] += tk[0] - t_sec; prof[2] += tk[1] - tk[0]; prof[3] += tk[2] - tk[1]; prof[4] += tk[4] - tk[2]; prof[6] += tk[3] - tk[4];
                    t_sec = tk[3];
                } else {
                    P::shade(A, smem, rec, nullptr, od, lane, c, nullptr, pre);
                }
                lds_store(q_head + m, head + (unsigned)nb);   // release: the record reads above are complete
                // lane e < 16 holds sample e: every lane < TW, as ray owner, adds its entries in record (= sample) order.  Entries and
                // owners sit in row 0 of the wave, so entry e reaches all of them through a DPP row broadcast (no LDS, no SGPR
                // round trip); fmaf(v, 1, c) = c + v and fmaf(v, 0, c) = c exactly (v is finite), i.e. the fused kernel's sums
                const int own = s < nb ? owner : -1;
                const float wr_ = r0[1] * sigmoid_rcp(c[0]), wg_ = r0[1] * sigmoid_rcp(c[1]), wb_ = r0[1] * sigmoid_rcp(c[2]);   // shade returns logits
                float cr = 0.0f, cg = 0.0f, cb = 0.0f;
                if (lane < TW) { cr = lds_peekf(csum + lane * 3); cg = lds_peekf(csum + lane * 3 + 1); cb = lds_peekf(csum + lane * 3 + 2); }
                collect16(own, wr_, wg_, wb_, lane, cr, cg, cb);
                if (lane < TW) { lds_pokef(csum + lane * 3, cr); lds_pokef(csum + lane * 3 + 1, cg); lds_pokef(csum + lane * 3 + 2, cb); }
                if constexpr (P::PROFILE) prof[5] += __builtin_readcyclecounter() - t_sec;
                ++st_pass;
                progress = true;
            }
            if (progress) idle = 0;
            else {
                [[maybe_unused]] unsigned long long t_idle = 0;
                if constexpr (P::PROFILE) t_idle = __builtin_readcyclecounter();
                if (++idle > kPcWatchdog) __builtin_trap();
                __builtin_amdgcn_s_sleep(2);
                if constexpr (P::PROFILE) prof[7] += __builtin_readcyclecounter() - t_idle;
            }
        }
        if (A.stats && lane == 0) {
            atomicAdd(A.stats + 2, st_pass);
            if constexpr (P::PROFILE) {
                for (int k = 1; k < 7; ++k) atomicAdd(A.stats + 4 + k, prof[k]);
                atomicAdd(A.stats + 12, prof[7]);
            }
        }
    }
}

}  // namespace ngf
