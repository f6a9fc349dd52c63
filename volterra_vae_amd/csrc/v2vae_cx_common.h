// Pieces shared by the matrix-core ELBO kernels of the complex model (v2vae_elbo_cx.hip, v2vae_elbo_cxb.hip).
#ifndef V2VAE_CX_COMMON_H
#define V2VAE_CX_COMMON_H
#include <hip/hip_runtime.h>

#include "v2vae_layout.h"

namespace v2vae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// tap (within a block of 16) held in accumulator register q of lane half h of a 32x32 MFMA result
typedef float f2 __attribute__((ext_vector_type(2)));
// two neighbouring floats as a register pair (one ds_read2_b32 / global_load_dwordx2; no alignment requirement)
__device__ __forceinline__ f2 ld2(const float* p) { return f2{p[0], p[1]}; }

__device__ __forceinline__ int cx_kk(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Lagged products of s = v_a + v_b over the owned symbols: C[d] = sum_m s[m] s[m + d], d = 0 .. number of lags - 1.
__device__ __forceinline__ float cx_s_at(const Lay& l, const float* __restrict__ mom, int64_t m) {
    if (m < l.sym_lo || m >= l.sym_lo + l.sym_cnt) return 0.f;
    return mom[(int64_t)1 * l.sym_cnt + (m - l.sym_lo)] + mom[(int64_t)5 * l.sym_cnt + (m - l.sym_lo)];
}

// sum over the owned, valid windows t of phase r of s[t + a] s[t + b]
__device__ __forceinline__ float psum_value(const Lay& l, const float* __restrict__ ws, int r, int a, int b) {
    const float* __restrict__ mom = ws + l.o_mom;
    const int kk = a < b ? a : b, d = a < b ? b - a : a - b;
    const int64_t lo = l.own_lo, hi = l.own_lo + l.own_cnt;
    const int64_t Tr = (l.W - 1 - r >= 0) ? (l.W - 1 - r) / l.S + 1 : 0;
    const int64_t br = hi < Tr ? hi : Tr;
    if (br <= lo) return 0.f;
    double v = reinterpret_cast<const double*>(ws + l.o_psum_tot)[d];
    auto c = [&](int64_t m) { return (double)(cx_s_at(l, mom, m) * cx_s_at(l, mom, m + d)); };
    for (int64_t m = lo; m < lo + kk; ++m) v -= c(m);
    const int64_t last = br - 1 + kk;
    if (last >= hi) { for (int64_t m = hi; m <= last; ++m) v += c(m); }
    else { for (int64_t m = last + 1; m < hi; ++m) v -= c(m); }
    return (float)v;
}


}  // namespace v2vae
#endif
