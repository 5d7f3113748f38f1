// ngf_host.hpp -- what the host-side translation units of libngf_hip.so share: the thread-local error slot behind
// ngf_last_error(), the HIP_TRY early-return macro, launch-grid helpers.
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
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

}  // namespace ngf

#define HIP_TRY(expr)                                                                                        \
    do {                                                                                                     \
        hipError_t e_ = (expr);                                                                              \
        if (e_ != hipSuccess) return ngf::fail(NGF_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_));    \
    } while (0)

namespace ngf {

// device -> host copy of n floats (weights are folded / packed on the host at create time)
inline int d2h(std::vector<float> &dst, const float *src, size_t n, hipStream_t st)
{
    dst.resize(n);
    if (!src) return fail(NGF_E_ARG, "missing weight tensor");
    HIP_TRY(hipMemcpyAsync(dst.data(), src, n * sizeof(float), hipMemcpyDeviceToHost, st));
    return NGF_OK;
}

}  // namespace ngf
