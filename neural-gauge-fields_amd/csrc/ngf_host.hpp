// ngf_host.hpp -- what the host-side translation units of libngf_hip.so share: the thread-local error slot behind
// ngf_last_error(), the HIP_TRY early-return macro, launch-grid helpers.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <chrono>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <new>
#include <vector>

#include "../../include/ngf.h"

namespace ngf {

// records the message for ngf_last_error() (thread local, ngf_field.hip) and returns `code`
int fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

static constexpr int kCounters = 256;      // launches in flight per handle (tile / ray queue heads)
static constexpr int kQueueHeads = 16;     // unsigned ints per launch slot: 8 per-XCD tile queue heads, one 64-byte line

// Experiment / test knobs (ngf_debug_set, include/ngf.h).  Process-wide, set ONLY through the explicit ABI call -- never read
// from the environment, so a stray variable in a user's shell cannot change what a launch computes.  -1 = library default.
enum Knob { KNOB_TILE_W = 0, KNOB_SPLIT, KNOB_WAVES, KNOB_NSTEP, KNOB_PROFILE, KNOB_ABLATE, KNOB_UV_TILES, KNOB_KERNEL, KNOB_STAGE, KNOB_POISON, KNOB_GRID, KNOB_XCD, KNOB_TAIL, KNOB_ORD_ROWS, KNOB_ORD_PX, KNOB_TRAIN_DWG, KNOB_COUNT };
int knob(int id);                          // current value (ngf_field.hip)
// hipFuncSetAttribute(kernel, MaxDynamicSharedMemorySize, bytes) once per (device, kernel) and size, not on every launch (ngf_field.hip)
hipError_t ensure_dynamic_lds(const void *kernel, size_t bytes);
// Knob "poison" (tests / hunting state-dependent reads; default off).  Bit 0: before every kernel of the library a "dirty" launch fills
// the whole LDS of every CU with the quiet-NaN pattern 0x7FC0DEAD, so a kernel that reads LDS it did not write (k-padding rows, ragged
// last passes, a missing wave-level fence) turns NaN instead of depending on what the previous kernel left there.  Bit 1: every device
// allocation of a handle is filled with the same pattern before it is packed (unwritten borders / pads show up the same way).
constexpr unsigned kPoisonPattern = 0x7FC0DEADu;
int poison_lds(hipStream_t st);                              // no-op unless knob poison & 1 (ngf_field.hip)
int poison_alloc(void *p, size_t bytes, hipStream_t st);     // no-op unless knob poison & 2

}  // namespace ngf

#define HIP_TRY(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return ngf::fail(NGF_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

namespace ngf {

// Scoped switch to the device a handle / a parked buffer lives on (hipFree, hipMalloc and events act on the calling thread's CURRENT device).
struct DeviceScope {
    int prev = -1;
    bool switched = false;
    explicit DeviceScope(int dev)
    {
        if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceScope() { if (switched) (void)hipSetDevice(prev); }
};

// device -> host copy of n floats (weights are folded / packed on the host at create time)
inline int d2h(std::vector<float> &dst, const float *src, size_t n, hipStream_t st)
{
    dst.resize(n);
    if (!src) return fail(NGF_E_ARG, "missing weight tensor");
    HIP_TRY(hipMemcpyAsync(dst.data(), src, n * sizeof(float), hipMemcpyDeviceToHost, st));
    return NGF_OK;
}

}  // namespace ngf
