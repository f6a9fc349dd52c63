// Pieces shared by the matrix-core ELBO kernels of the complex model (v2vae_elbo_cx.hip, v2vae_elbo_cxb.hip).
#ifndef V2VAE_CX_COMMON_H
#define V2VAE_CX_COMMON_H
#include <hip/hip_runtime.h>

#include "v2vae_layout.h"

namespace v2vae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short bfx8 __attribute__((ext_vector_type(8)));   // eight bf16 values: one operand of a 16-deep bf16 matrix instruction

// tap (within a block of 16) held in accumulator register q of lane half h of a 32x32 MFMA result
typedef float f2 __attribute__((ext_vector_type(2)));
// two neighbouring floats as a register pair (one ds_read2_b32 / global_load_dwordx2; no alignment requirement)
__device__ __forceinline__ f2 ld2(const float* p) { return f2{p[0], p[1]}; }

__device__ __forceinline__ int cx_kk(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }

// Sum along the anti-diagonals of a 32-window x 16-tap tile held "windows on lanes, taps in registers" in the MFMA accumulator
// layout: lane = window + 32 h, register q holds tap (q & 3) + 8 (q >> 2) + 4 h.  Returns, on lane m (0 <= m < 47), the sum of
// v[window][tap] over window + tap = m; lanes from 47 on return 0.  No LDS: each half of the wave is moved to lanes 0..31 of its
// own register set (v_permlane32_swap against zero), reduced by a Horner scheme of one-lane shifts fused into the additions
// (v_add_f32 ... wave_shr:1), and the two halves are combined with the four-lane offset between their taps.
__device__ __forceinline__ float cx_shr1(float s) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s), 0x138 /* wave_shr:1 */, 0xF, 0xF, false));
}
__device__ __forceinline__ float cx_horner8(const float (&t)[8]) {  // lane m: sum_q t[q][m - ((q & 3) + 8 (q >> 2))]
    float s = t[7];
    s = cx_shr1(s) + t[6];
    s = cx_shr1(s) + t[5];
    s = cx_shr1(s) + t[4];
    s = cx_shr1(cx_shr1(cx_shr1(cx_shr1(s))));
    s = cx_shr1(s) + t[3];
    s = cx_shr1(s) + t[2];
    s = cx_shr1(s) + t[1];
    s = cx_shr1(s) + t[0];
    return s;
}
__device__ __forceinline__ float cx_antidiag(const float (&v)[8]) {
    float lo[8], hi[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        // after the swap: lo = (v of lanes 0..31 | 0), hi = (v of lanes 32..63 moved to lanes 0..31 | 0).  Written as inline
        // assembly because the compiler's builtin returns its first result twice when the second operand is a zero; the
        // s_nop cover the VALU-write -> permlane-swap hazards the compiler cannot see inside the statement.
        float x = v[q], z = 0.f;
        asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(z));  // not volatile: free to interleave
        lo[q] = x;
        hi[q] = z;
    }
    const float a = cx_horner8(lo), b = cx_horner8(hi);   // b belongs to taps + 4
    return a + cx_shr1(cx_shr1(cx_shr1(cx_shr1(b))));
}

// x(lane) + x(lane ^ 32) on every lane, on the vector pipe: after the swap `a` holds the lower half's value in both halves and
// `b` the upper half's.
__device__ __forceinline__ float cx_sum_halves(float x) {
    float a = x, b = x;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    return a + b;
}

// Sum of x over the 64 lanes, valid on lane 63 only: shifts by 1, 2, 4, 8 inside the rows of 16 lanes (lanes without a source
// add zero), then the row totals are passed on to the following rows (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows
// 2 and 3).  Six additions on the vector pipe, no LDS round trip.
__device__ __forceinline__ float cx_wave_sum_to_last(float x) {
#define CX_DPP_ADD(ctrl, rmask)                                                                                                  \
    x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), (ctrl), (rmask), 0xF, (rmask) == 0xF))
    CX_DPP_ADD(0x111, 0xF);   // row_shr:1
    CX_DPP_ADD(0x112, 0xF);   // row_shr:2
    CX_DPP_ADD(0x114, 0xF);   // row_shr:4
    CX_DPP_ADD(0x118, 0xF);   // row_shr:8
    CX_DPP_ADD(0x142, 0xA);   // row_bcast:15 -> rows 1 and 3 (the other rows keep old = 0, i.e. add nothing)
    CX_DPP_ADD(0x143, 0xC);   // row_bcast:31 -> rows 2 and 3
#undef CX_DPP_ADD
    return x;
}

#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Lagged products of s = v_a + v_b over the owned symbols: C[d] = sum_m s[m] s[m + d], d = 0 .. number of lags - 1.
__device__ __forceinline__ float cx_s_at(const Lay& l, const float* __restrict__ mom, int64_t m) {
    if (m < l.sym_lo || m >= l.sym_lo + l.sym_cnt) return 0.f;
    return mom[(int64_t)1 * l.sym_cnt + (m - l.sym_lo)] + mom[(int64_t)5 * l.sym_cnt + (m - l.sym_lo)];
}

// sum over the owned, valid windows t of phase r of s[t + a] s[t + b]
// `tot` = the lagged products over all owned symbols (doubles, one per lag)
__device__ __forceinline__ float psum_value(const Lay& l, const float* __restrict__ ws, const double* tot, int r, int a, int b) {
    const float* __restrict__ mom = ws + l.o_mom;
    const int kk = a < b ? a : b, d = a < b ? b - a : a - b;
    const int64_t lo = l.own_lo, hi = l.own_lo + l.own_cnt;
    const int64_t Tr = (l.W - 1 - r >= 0) ? (l.W - 1 - r) / l.S + 1 : 0;
    const int64_t br = hi < Tr ? hi : Tr;
    if (br <= lo) return 0.f;
    double v = tot[d];
    auto c = [&](int64_t m) { return (double)(cx_s_at(l, mom, m) * cx_s_at(l, mom, m + d)); };
    for (int64_t m = lo; m < lo + kk; ++m) v -= c(m);
    const int64_t last = br - 1 + kk;
    if (last >= hi) { for (int64_t m = hi; m <= last; ++m) v += c(m); }
    else { for (int64_t m = last + 1; m < hi; ++m) v -= c(m); }
    return (float)v;
}


}  // namespace v2vae
#endif
