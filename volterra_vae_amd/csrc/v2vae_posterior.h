// Soft demapper shared by every kernel that needs the per-symbol posterior and its central moments
// (lib/real_equalization.py:816-823 and the moment sums of :851-866).
#ifndef V2VAE_POSTERIOR_H
#define V2VAE_POSTERIOR_H
#include <hip/hip_runtime.h>
#include <math.h>

#include "v2vae_layout.h"

namespace v2vae {

struct Posterior {
    float q[MAX_C];
    float mu, var, kap, gam;
};

// inv_den[i] = 1 / (noise variance * scaling[i]) in double, computed once per thread (a double division costs
// more than the rest of the posterior together)
template <int CB>
__device__ __forceinline__ void posterior_temperatures(const float* __restrict__ sc, int C, float nv, double (&inv_den)[MAX_C]) {
#pragma unroll
    for (int i = 0; i < CB; ++i) inv_den[i] = i < C ? 1.0 / ((double)nv * (double)sc[i]) : 0.0;
}

// EX: the level count equals CB (known at compile time): no per-level bounds checks
template <int CB, bool EX = false>
__device__ __forceinline__ void posterior(double x, const float* __restrict__ lv, const double (&inv_den)[MAX_C], int C_rt,
                                          Posterior& p) {
    const int C = EX ? CB : C_rt;
    // Logits reach |z| ~ 1e2..1e3 for small noise variance; rounding them to fp32 before the max is
    // subtracted would put ~|z| * 6e-8 of absolute error into the exponent, i.e. that much RELATIVE
    // error into q.  The difference z - max is therefore formed in double and only then rounded.
    double zd[MAX_C];
    double mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < CB; ++i)
        if (i < C) {
            const double d = x - (double)lv[i];
            zd[i] = -(d * d) * inv_den[i];
            mx = fmax(mx, zd[i]);
        }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < CB; ++i)
        if (i < C) {
            // exp(t), t <= 0 a difference formed in double.  Rounding t to float first would put |t| * 6e-8 of RELATIVE
            // error into q (1e-5 at t = -150), and the gradients of a peaked posterior are proportional to exactly those
            // small q.  So the range reduction t log2(e) = n + f is done in double and only f in [-1/2, 1/2] goes through the
            // hardware exponential (1 ulp); 2^n is applied exactly (flushes to 0 below 2^-149 like the library function).
            const double t2 = (zd[i] - mx) * 1.4426950408889634;
            const double nr = rint(t2);
            p.q[i] = ldexpf(__builtin_amdgcn_exp2f((float)(t2 - nr)), (int)fmax(nr, -400.0));
            sum += p.q[i];
        }
    const float inv = __builtin_amdgcn_rcpf(sum);  // sum is in [1, C]
    float mu = 0.f;
#pragma unroll
    for (int i = 0; i < CB; ++i)
        if (i < C) {
            p.q[i] *= inv;
            mu = fmaf(p.q[i], lv[i], mu);
        }
    float v = 0.f, k3 = 0.f, c4 = 0.f;
#pragma unroll
    for (int i = 0; i < CB; ++i)
        if (i < C) {
            const float d = lv[i] - mu, d2 = d * d;
            v = fmaf(p.q[i], d2, v);
            k3 = fmaf(p.q[i], d2 * d, k3);
            c4 = fmaf(p.q[i], d2 * d2, c4);
        }
    p.mu = mu; p.var = v; p.kap = k3; p.gam = c4 - 3.f * v * v;
}

}  // namespace v2vae
#endif
