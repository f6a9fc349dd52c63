// Forward of the complex second-order Volterra equaliser on the bf16 matrix pipe in split arithmetic (one or two samples per
// symbol, up to 95 lags).
//
//   xhat = sum_a y_a (k1w_a + u_a),   u_a = sum_{b >= a} K'_ab y_b,   K'_ab = K_ab (a = b), 2 K_ab (a < b),   y_x = y[n S + p2 - x]
//
// (the symmetric form: K = k2 + k2^T is complex symmetric).  The row sums u_a - all but T of the T (T + 3) / 2 complex
// multiply-adds per symbol - are a matrix product of the constant table with the sliding sample windows,
//   [Re u; Im u] (32 rows = 16 lags) = [[K'r, -K'i], [K'i, K'r]] . [Ya; Yb],
// formed by v_mfma_f32_32x32x16_bf16 with the symbols on the lanes: every float is the exact sum of three bf16 pieces (8 + 8 + 8
// significand bits, each rounded to nearest) and the six piece products whose weight is not below 2^-24 of the whole are accumulated in
// float, which is what the packed-float row sums of the vector-pipe kernel (k_eq_fwd_sym) amount to.  An f32 matrix instruction
// would run at the vector rate and hold the vector pipe; a bf16 one runs at sixteen times that rate beside it.
//   * lag rows in blocks of 16 (32 matrix rows: Re | Im); block row R needs the lag blocks g >= R only (upper triangle);
//   * B operand: k-slot (lane half h, element j) of chunk (stream c, lag block g) is lag 16 g + 8 h + 7 - j, i.e. eight
//     CONSECUTIVE samples of stream c in ascending order: the samples are split once per tile into planes of bf16 and a lane
//     reads four aligned words per piece (the tile origin is chosen so that they are aligned at two samples per symbol; at one
//     sample per symbol an odd symbol shifts by one piece);
//   * A operand: the table pieces, LDS-resident for the whole launch, one 16-byte read per lane and matrix instruction pair:
//     a wave works on two column tiles of 32 symbols at a time, so every table fragment read serves two matrix instructions
//     and every sample fragment all block rows R <= g;
//   * a lag count of the form 16 NBK + 1 (65 lags: the configuration of the long-channel benchmark) would spend a whole block
//     row and column - NBK + 1 of the (NBK + 1)(NBK + 2) / 2 block pairs - on one lag.  The XL = 1 instances keep NBK blocks on the
//     matrix pipe and carry lag e = 16 NBK on the side, in the closing stage: u_a += K'_ae y_e for every row a < e (one complex
//     multiply-add on the vector pipe per row), and row e itself is the single product K'_ee y_e;
//   * the closing stage stays what it was: first-order weight added to the finished row sum, the row's product with y_a in
//     float, the sum over the rows in double (the soft demapper amplifies an error of xhat by 2 |xhat - a| / noise variance on
//     its way into the equaliser gradients), rounded to float once.
#include <hip/hip_runtime.h>

#include "v2vae_bf16_split.h"
#include "v2vae_cx_common.h"
#include "v2vae_layout.h"

namespace v2vae {

// waves per workgroup (64 symbols each): four up to two lag blocks (the table is small: several workgroups per CU), else eight
// (one workgroup per CU holds the table)
V2VAE_HD constexpr int eqf_nw(int nbk) { return nbk <= 2 ? 4 : 8; }
// column tiles of 32 symbols a wave works on at a time: two while their accumulators fit the registers
V2VAE_HD constexpr int eqf_ct(int nbk) { return nbk <= 4 ? 2 : 1; }

// extra left halo of the sample tile: the lag count padded to whole blocks, and one more sample where that makes the first
// sample of every operand even (so that at two samples per symbol every operand is a run of aligned words)
V2VAE_HD inline int eqf_xh(const Lay& l, int nbk) { return 16 * nbk + ((l.hl + l.p2 + 1) & 1); }
V2VAE_HD inline int eqf_tlen(const Lay& l, int nbk) { return 64 * eqf_nw(nbk) * l.S + l.hl + l.hr + 1 + eqf_xh(l, nbk); }
V2VAE_HD inline int eqf_nwd(const Lay& l, int nbk) { return (eqf_tlen(l, nbk) + 1) / 2 + 4; }
// floats of the side lag's table: K'_ae for the 16 NBK rows in accumulator-slot order, then (K'_ee, first-order weight of lag e)
V2VAE_HD constexpr int eqf_xl_floats(int nbk, int xl) { return xl ? 32 * nbk + 4 : 0; }
V2VAE_HD inline int eqf_lds_floats(const Lay& l, int nbk) {
    return nbk * (nbk + 1) / 2 * 1536 + 32 * nbk + eqf_xl_floats(nbk, l.eqfbf_xl) + 2 * eqf_tlen(l, nbk) + 6 * eqf_nwd(l, nbk);
}

// LDS (floats): table images [pairs (R, g >= R)][2 streams][3 pieces][64 lanes][4 words] | first-order weights [16 NBK][2] |
// XL: side-lag column [16 NBK][2] and four scalars | sample tile [tlen][2] | bf16 planes [2 streams][3 pieces][nwd words]
template <int NBK, int SS, int XL>
__global__ __launch_bounds__(eqf_nw(NBK) * 64, NBK <= 2 ? 3 : 2) void k_eq_fwd_bf(Lay l, const float* __restrict__ y, const float* __restrict__ ws,
                                                                          float* __restrict__ xhat) {
    constexpr int NP = NBK * (NBK + 1) / 2, EQF_NW = eqf_nw(NBK), NTH = EQF_NW * 64, EQF_NSYM = 64 * EQF_NW, S = SS, CT = eqf_ct(NBK);
    typedef const unsigned __attribute__((address_space(3))) * lword_p;
    extern __shared__ __align__(16) float sm[];
    const u32x4* s_A = reinterpret_cast<const u32x4*>(sm);
    // first-order weights in the order of the accumulator slots: [lag block R][lane half h][16] = (Re of lags 16 R + cx_kk(q, h),
    // q = 0..7 | Im of the same lags)
    const f32x4* s_k1 = reinterpret_cast<const f32x4*>(sm + NP * 1536);
    const int tlen = eqf_tlen(l, NBK), nwd = eqf_nwd(l, NBK), xh = eqf_xh(l, NBK);
    constexpr int XLF = eqf_xl_floats(NBK, XL);
    // XL: column e = 16 NBK of K' in the order of the accumulator slots, [lag block R][lane half h][16] like the first-order
    // weights, then K'_ee (Re, Im) and the first-order weight of lag e (Re, Im)
    const f32x4* s_xl = reinterpret_cast<const f32x4*>(sm + NP * 1536 + 32 * NBK);
    f2* tile = reinterpret_cast<f2*>(sm + NP * 1536 + 32 * NBK + XLF);
    unsigned* Yw = reinterpret_cast<unsigned*>(sm + NP * 1536 + 32 * NBK + XLF + 2 * tlen);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, i = lane & 31, h = lane >> 5;
    const int64_t sym_hi = l.sym_lo + l.sym_cnt;
    const int64_t ntile = (l.sym_cnt + EQF_NSYM - 1) / EQF_NSYM;
    const int C0 = l.hl + xh + l.p2;   // tile-local sample of symbol n at lag x: n S + C0 - x

    for (int idx = tid; idx < NP * 1536; idx += NTH) sm[idx] = ws[l.o_eqfbf + idx];
    for (int idx = tid; idx < 32 * NBK; idx += NTH) {
        const int q = idx & 15, hh = (idx >> 4) & 1, R = idx >> 5;
        sm[NP * 1536 + idx] = ws[l.o_eqfbf + NP * 1536 + 2 * (16 * R + cx_kk(q & 7, hh)) + (q >> 3)];
    }
    if (XL) {
        const float* src = ws + l.o_eqfbf + NP * 1536 + 32 * NBK;   // [16 NBK][2], then the four scalars
        for (int idx = tid; idx < 32 * NBK; idx += NTH) {
            const int q = idx & 15, hh = (idx >> 4) & 1, R = idx >> 5;
            sm[NP * 1536 + 32 * NBK + idx] = src[2 * (16 * R + cx_kk(q & 7, hh)) + (q >> 3)];
        }
        if (tid < 4) sm[NP * 1536 + 64 * NBK + tid] = src[32 * NBK + tid];
    }
    for (int idx = tid; idx < 6 * nwd; idx += NTH) Yw[idx] = 0u;

    // The samples of a tile travel through registers: they are requested before the tile in front of them is computed and
    // written (as float pairs for the closing products and as bf16 pieces for the matrix operands) when it is done, so the
    // memory latency of a tile lies behind the arithmetic of the one before.  A tile inside the sequence and the local sample
    // block (all but the ends) is fetched without per-element checks.
    constexpr int NPRE = 4;   // samples per thread and tile (tlen <= 4 NTH at every supported size)
    f2 pre[NPRE];
    auto fetch = [&](int64_t t) {
        const int64_t g0 = (l.sym_lo + t * EQF_NSYM) * l.S - l.hl - xh, loc0 = g0 - l.y_lo;
        const bool inside = g0 >= 0 && g0 + tlen <= l.n_glob && loc0 >= 0 && loc0 + tlen <= l.y_len;   // uniform
        const float* __restrict__ sa = y + loc0;
        const float* __restrict__ sb = y + l.y_stride + loc0;
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int e = tid + k * NTH;
            const int64_t g = g0 + e, loc = loc0 + e;
            const bool in = e < tlen && (inside || (g >= 0 && g < l.n_glob && loc >= 0 && loc < l.y_len));
            pre[k] = in ? f2{sa[e], sb[e]} : f2{0.f, 0.f};
        }
    };
    auto stage = [&]() {
        unsigned short* Yh = reinterpret_cast<unsigned short*>(Yw);
#pragma unroll
        for (int k = 0; k < NPRE; ++k) {
            const int e = tid + k * NTH;
            if (e < tlen) {
                const f2 v = pre[k];
                tile[e] = v;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    unsigned b1, b2, b3;
                    bf16_split3_rne(c ? v.y : v.x, &b1, &b2, &b3);   // (split once per tile: pieces rounded to nearest)
                    unsigned short* d = Yh + 2 * (c * 3 * nwd) + e;
                    d[0] = (unsigned short)(b1 >> 16);
                    d[2 * nwd] = (unsigned short)(b2 >> 16);
                    d[4 * nwd] = (unsigned short)(b3 >> 16);
                }
            }
        }
    };
    if ((int64_t)blockIdx.x < ntile) fetch(blockIdx.x);
    for (int64_t t = blockIdx.x; t < ntile; t += gridDim.x) {
        const int64_t b0 = l.sym_lo + t * EQF_NSYM;
        __syncthreads();
        stage();
        __syncthreads();
        if (t + gridDim.x < ntile) fetch(t + gridDim.x);

        // ---- row sums of the wave's 64 symbols, CT column tiles of 32 at a time (symbols 64 wv + 32 (ct0 + ct) + i)
#pragma unroll 1
        for (int ct0 = 0; ct0 < 2; ct0 += CT) {
        // (the table address is made opaque per pass: the table does not depend on the pass, and hoisted out of the loops its
        // fragments would be held in - and spilled from - registers)
        typedef const u32x4 __attribute__((address_space(3))) * lvec_p;
        lvec_p At = (lvec_p)s_A + lane;
        asm volatile("" : "+v"(At));
        f32x16 acc[CT][NBK];
#pragma unroll
        for (int R = 0; R < NBK; ++R)
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[ct][R][q] = 0.f;
        const int t00 = (64 * wv + 32 * ct0 + i) * S + C0 - 8 * h - 7;   // first sample of the lane's operand of lag block 0, first column tile
        const unsigned sh = (t00 & 1) * 16;                    // (zero at two samples per symbol: the tile origin sees to that)
        // table fragments of step n + 1 (a step = one (lag block g, stream c, block row R <= g)) are requested before the matrix
        // instructions of step n are issued: two register sets, alternating
        u32x4 Abuf[2][3];
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) Abuf[0][pc] = At[pc * 64];   // step 0: pair (0, 0), stream 0
        int step = 0;
#pragma unroll
        for (int g = 0; g < NBK; ++g) {
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                u32x4 B[CT][3];
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) {
                    lword_p xw = (lword_p)(Yw + c * 3 * nwd) + ((t00 + 32 * ct * S - 16 * g) >> 1);
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) {
                        if (SS == 2) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) B[ct][pc][q] = xw[pc * nwd + q];
                        } else {
                            unsigned w[5];
#pragma unroll
                            for (int q = 0; q < 5; ++q) w[q] = xw[pc * nwd + q];
#pragma unroll
                            for (int q = 0; q < 4; ++q) B[ct][pc][q] = __builtin_amdgcn_alignbit(w[q + 1], w[q], sh);
                        }
                    }
                }
#pragma unroll
                for (int R = 0; R <= g; ++R) {
                    {   // the step after this one
                        const int R2 = R < g ? R + 1 : 0, c2 = R < g ? c : (c ^ 1), g2 = (R < g || c == 0) ? g : g + 1;
                        if (g2 < NBK) {
                            const int pi2 = R2 * NBK - R2 * (R2 - 1) / 2 + (g2 - R2);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int pc = 0; pc < 3; ++pc) Abuf[(step + 1) & 1][pc] = At[((pi2 * 2 + c2) * 3 + pc) * 64];
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                    const u32x4 (&A)[3] = Abuf[step & 1];
                    ++step;
                    // table piece x sample piece, smallest products first: (1,3) (2,2) (3,1) (1,2) (2,1) (1,1)
                    constexpr int TP6[6] = {0, 1, 2, 0, 1, 0}, DP6[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
                    for (int k = 0; k < 6; ++k)
#pragma unroll
                        for (int ct = 0; ct < CT; ++ct)
                            acc[ct][R] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfx8, A[TP6[k]]),
                                                                                 __builtin_bit_cast(bfx8, B[ct][DP6[k]]), acc[ct][R], 0, 0, 0);
                }
            }
        }

        // ---- closing stage: accumulator slot q of lane half h holds lag 16 R + cx_kk(q, h) (Re in slot q, Im in slot q + 8)
        double outr[CT], outi[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) { outr[ct] = 0.0; outi[ct] = 0.0; }
#pragma unroll
        for (int R = 0; R < NBK; ++R) {
            // (the first-order weight - the centre one is the largest number of the whole sum - joins the FINISHED row sum, one
            // rounding at its size; as the start value of the accumulator it would take one rounding per matrix instruction of
            // the row, 12 per lag block, and left 2e-7 of relative error in xhat)
            f32x4 kw[4], kx[4];
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                kw[v] = s_k1[(R * 2 + h) * 4 + v];
                if (XL) kx[v] = s_xl[(R * 2 + h) * 4 + v];
            }
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const f2* ys = tile + (64 * wv + 32 * (ct0 + ct) + i) * S + C0;
                const f2 ye = XL ? ys[-16 * NBK] : f2{0.f, 0.f};
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int a = 16 * R + cx_kk(q, h);
                    const f2 yv = ys[-a];
                    float sr = acc[ct][R][q], si = acc[ct][R][q + 8];
                    if (XL) {   // the side lag's term of the row sum: K'_ae y_e
                        const float xr = kx[q >> 2][q & 3], xi = kx[2 + (q >> 2)][q & 3];
                        sr = fmaf(-xi, ye.y, fmaf(xr, ye.x, sr));
                        si = fmaf(xi, ye.x, fmaf(xr, ye.y, si));
                    }
                    const float ur = sr + kw[q >> 2][q & 3], ui = si + kw[2 + (q >> 2)][q & 3];
                    const float tr = fmaf(-ui, yv.y, ur * yv.x), ti = fmaf(ui, yv.x, ur * yv.y);
                    outr[ct] += (double)tr;
                    outi[ct] += (double)ti;
                }
            }
        }
        if (XL && h == 0) {   // row e: u_e = K'_ee y_e, its first-order weight, its product with y_e
            const f32x4 ke = *reinterpret_cast<const f32x4*>(sm + NP * 1536 + 64 * NBK);
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                const f2 ye = tile[(64 * wv + 32 * (ct0 + ct) + i) * S + C0 - 16 * NBK];
                const float ur = fmaf(-ke[1], ye.y, ke[0] * ye.x) + ke[2], ui = fmaf(ke[1], ye.x, ke[0] * ye.y) + ke[3];
                outr[ct] += (double)fmaf(-ui, ye.y, ur * ye.x);
                outi[ct] += (double)fmaf(ui, ye.x, ur * ye.y);
            }
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) {
            const double sr = outr[ct] + __shfl_xor(outr[ct], 32), si = outi[ct] + __shfl_xor(outi[ct], 32);
            const int64_t b = b0 + 64 * wv + 32 * (ct0 + ct) + i;
            if (h == 0 && b < sym_hi) {
                xhat[b - l.sym_lo] = (float)sr;
                xhat[l.sym_cnt + (b - l.sym_lo)] = (float)si;
            }
        }
        }
    }
}

template <int NBK, int XL>
static int launch_eqf(const Lay& l, const float* y, const float* ws, float* xhat, hipStream_t st) {
    constexpr int EQF_NW = eqf_nw(NBK), EQF_NSYM = 64 * EQF_NW;
    const int lds = eqf_lds_floats(l, NBK) * 4;
    const int64_t ntile = (l.sym_cnt + EQF_NSYM - 1) / EQF_NSYM;
    int per_cu = 160 * 1024 / (lds + 512);
    const int cap = NBK <= 2 ? 3 : 1;   // (registers: three waves per SIMD up to two lag blocks)
    per_cu = per_cu < 1 ? 1 : per_cu > cap ? cap : per_cu;
    const int nb = (int)(ntile < (int64_t)per_cu * 256 ? ntile : (int64_t)per_cu * 256);
#define V2VAE_EQF(SS)                                                                                                       \
    do {                                                                                                                    \
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(k_eq_fwd_bf<NBK, SS, XL>),                      \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)          \
            return V2VAE_ELAUNCH;                                                                                           \
        hipLaunchKernelGGL((k_eq_fwd_bf<NBK, SS, XL>), dim3(nb), dim3(EQF_NW * 64), lds, st, l, y, ws, xhat);                    \
    } while (0)
    if (l.S == 1) V2VAE_EQF(1);
    else V2VAE_EQF(2);
#undef V2VAE_EQF
    return V2VAE_OK;
}

int eq_fwd_bf_lds_bytes(const Lay& l) { return eqf_lds_floats(l, l.eqfbf_nbk) * 4; }

int launch_eq_fwd_bf(const Lay& l, const float* y, const float* ws, float* xhat, hipStream_t st) {
    if (l.eqfbf_xl) {   // 16 NBK + 1 lags: NBK blocks and the lag on the side
        switch (l.eqfbf_nbk) {
            case 2: return launch_eqf<2, 1>(l, y, ws, xhat, st);
            case 3: return launch_eqf<3, 1>(l, y, ws, xhat, st);
            case 4: return launch_eqf<4, 1>(l, y, ws, xhat, st);
            case 5: return launch_eqf<5, 1>(l, y, ws, xhat, st);
            default: return V2VAE_ESIZE;
        }
    }
    switch (l.eqfbf_nbk) {
        case 3: return launch_eqf<3, 0>(l, y, ws, xhat, st);
        case 4: return launch_eqf<4, 0>(l, y, ws, xhat, st);
        case 5: return launch_eqf<5, 0>(l, y, ws, xhat, st);
        case 6: return launch_eqf<6, 0>(l, y, ws, xhat, st);
        default: return V2VAE_ESIZE;
    }
}

}  // namespace v2vae
