// Micro-benchmark: accuracy of cheaper softplus(u) = log1p(exp(u)) forms against double precision, and their instruction counts.
// render_kernel's march spends ~150 of its ~800 VALU instructions per step in ocml's log1pf (a double-float evaluation, ~130
// instructions) -- the reference needs float32 accuracy, not more (F.softplus on the CPU is SLEEF / glibc at <= 1 ulp).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o softplus_accuracy softplus_accuracy.hip && ./softplus_accuracy
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

// candidate 0: what the kernel does today
__device__ float sp_ocml(float u) { return u > 20.0f ? u : log1pf(expf(u)); }

// candidate 1: t = 1 + e, log(t) (ocml logf: v_log_f32 + extended-precision ln2 scaling), Kahan correction for the rounding of the sum
__device__ float sp_kahan(float u)
{
    if (u > 20.0f) return u;
    const float e = expf(u);
    const float t = 1.0f + e;
    const float c = e - (t - 1.0f);               // exact for e <= 1; for e > 1 the lost part is below half an ulp of t anyway
    return logf(t) + c * __builtin_amdgcn_rcpf(t);
}

// candidate 2: small e through the atanh series (no cancellation anywhere), large e through logf
__device__ float sp_series(float u)
{
    if (u > 20.0f) return u;
    const float e = expf(u);
    const float s = e * __builtin_amdgcn_rcpf(2.0f + e), z = s * s;
    const float p = fmaf(fmaf(fmaf(fmaf(fmaf(z, 1.0f / 11.0f, 1.0f / 9.0f), z, 1.0f / 7.0f), z, 0.2f), z, 1.0f / 3.0f), z, 1.0f);
    const float small = 2.0f * s * p;
    const float t = 1.0f + e;
    const float big = logf(t) + (e - (t - 1.0f)) * __builtin_amdgcn_rcpf(t);
    return e < 0.5f ? small : big;
}

// candidate 3: candidate 1 on the raw hardware log (v_log_f32 * ln2 in two pieces)
__device__ float sp_hwlog(float u)
{
    if (u > 20.0f) return u;
    const float e = expf(u);
    const float t = 1.0f + e;
    const float c = e - (t - 1.0f);
    const float l2 = __builtin_amdgcn_logf(t);
    const float hi = l2 * 0.693145751953125f;                  // ln2 split: 0x3f317200 (exact product for |l2| < 2^11 multiples) + remainder
    const float r = fmaf(l2, 1.428606765330187e-06f, hi);
    return r + c * __builtin_amdgcn_rcpf(t);
}

template <int WHICH>
__global__ void eval(const float *u, float *out, double *ref, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = u[i];
    float y;
    if (WHICH == 0) y = sp_ocml(x);
    else if (WHICH == 1) y = sp_kahan(x);
    else if (WHICH == 2) y = sp_series(x);
    else y = sp_hwlog(x);
    out[i] = y;
    if (ref) ref[i] = x > 20.0f ? (double)x : log1p(exp((double)x));     // the float32 function's exact value at the float32 argument
}

int main()
{
    const int n = 1 << 22;
    std::vector<float> u(n);
    for (int i = 0; i < n; ++i) u[i] = -40.0f + 62.0f * (float)i / (float)(n - 1);       // [-40, 22]
    float *du, *dy;
    double *dr;
    (void)hipMalloc(&du, n * 4); (void)hipMalloc(&dy, n * 4); (void)hipMalloc(&dr, n * 8);
    (void)hipMemcpy(du, u.data(), n * 4, hipMemcpyHostToDevice);
    std::vector<float> y(n);
    std::vector<double> r(n);
    const char *names[4] = {"ocml log1pf(expf(u))           ", "logf(1+e) + Kahan correction    ", "atanh series (e<0.5) | logf+Kahan", "v_log_f32*ln2 + Kahan correction"};
    for (int w = 0; w < 4; ++w) {
        if (w == 0) eval<0><<<n / 256, 256>>>(du, dy, dr, n);
        if (w == 1) eval<1><<<n / 256, 256>>>(du, dy, nullptr, n);
        if (w == 2) eval<2><<<n / 256, 256>>>(du, dy, nullptr, n);
        if (w == 3) eval<3><<<n / 256, 256>>>(du, dy, nullptr, n);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(y.data(), dy, n * 4, hipMemcpyDeviceToHost);
        if (w == 0) (void)hipMemcpy(r.data(), dr, n * 8, hipMemcpyDeviceToHost);
        double worst = 0, worst_u = 0, sum = 0;
        for (int i = 0; i < n; ++i) {
            const float rf = (float)r[i];
            if (rf == 0.0f || !std::isfinite(rf)) continue;
            uint32_t b; memcpy(&b, &rf, 4);
            uint32_t b1 = b + 1; float nf; memcpy(&nf, &b1, 4);
            const double ulp = (double)nf - (double)rf;
            const double err = std::fabs((double)y[i] - r[i]) / ulp;
            sum += err;
            if (err > worst) { worst = err; worst_u = u[i]; }
        }
        printf("%s: max error %.2f ulp (at u = %.4f), mean %.3f ulp\n", names[w], worst, worst_u, sum / n);
    }
    return 0;
}
