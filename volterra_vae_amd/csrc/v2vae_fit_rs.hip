// Small-shape real (PAM) model: the whole mini-batch loop of GenericTorchBlindEqualizer.fit
// (lib/real_equalization.py:482-502) for one model inside ONE workgroup, with every intermediate on chip.
//
// The reference's default configuration (500-symbol batches, 15-tap channel, 25-lag equaliser) is ~2 Mflop per
// step with a strict step-to-step dependency, far too little for a grid of workgroups.  Here a workgroup of 8 waves
// keeps the batch, the posterior moments, their gradients and all tables in LDS, the parameters and the optimiser
// state in registers, and touches global memory only for the samples in and the equalised symbols / loss out:
//   equaliser  x = k1 * y + X^T (k2 + k2^T) X      matrix cores: Z = K X per 32-symbol tile, then the row dot
//   soft demapper, moments, KL                     one symbol per lane
//   ELBO expectations + their backward             one window per lane, the <= 8 active taps of its phase in
//                                                  registers (wavefront FMA), per-lane outer-product accumulators
//                                                  reduced across the wave by a transposing butterfly
//   sum_n v v^T                                    lagged products (Toeplitz up to edge terms)
//   equaliser backward  X^T diag(g) X              matrix cores, symbols as the contraction index
//   Adam-AMSGrad                                   registers
// Mathematics and conventions are those of the generic kernels (v2vae_kernels.hip); only the summation order differs.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdlib.h>

#include "v2vae_layout.h"
#include "v2vae_posterior.h"

namespace v2vae {

constexpr int RS_NT = 256;          // threads per workgroup
constexpr int RS_SLOTS = 512;       // window slots per chunk of the ELBO stage (each thread takes RS_SLOTS / RS_NT of them)
constexpr int RS_PARTS = RS_NT / 128;  // threads sharing one element of the cross-lane sum
constexpr int RS_NW = RS_NT / 64;   // waves
constexpr int RS_HL = 32;           // zero samples kept left of the batch in LDS
constexpr int RS_APT = 8;           // optimiser elements per thread (parameter pack <= RS_APT * RS_NT)
constexpr int RS_YPT = 8;           // samples per thread fetched one batch ahead (batches up to RS_YPT * RS_NT samples)
constexpr int RS_QCACHE = 4;        // posteriors of up to this many levels are kept in LDS between forward and backward
constexpr int RS_GREC = 96;         // per-wave, per-phase record: G 8x8 | f1 8 | f2 8 | f3 8 | pad 8

typedef float rs_f16 __attribute__((ext_vector_type(16)));
typedef float rs_f2 __attribute__((ext_vector_type(2)));

struct RsOff {  // LDS carve in floats
    int y, x, gx, q, mom, MP, gm, GP, D, vl, H, HH, h1c, d, dq, par, grad, gtot, P, misc, dbl, total;
    int nchunks, TS, packp;
    int c0, o1, o2;  // equaliser sample window: X_b[i] = y[b S + c0 - i], i < 32; first-order tap j at o1 - j, lag j at o2 + j
};

static inline int rs_rup(int x, int m) { return (x + m - 1) / m * m; }

// Fills `o` and returns 1 if the shape fits this kernel.
int rs_plan(const Lay& l, int64_t pack, RsOff* o) {
    if (l.flags & V2VAE_FLAG_GENERIC) return 0;
    if (l.mode != 0 || l.ns != 1 || l.LE > 8 || l.S > 4 || l.l2 > 31 || l.l1 > 64 || l.pad1 > RS_HL || l.C > MAX_C) return 0;
    if (pack > RS_APT * RS_NT || l.b_glob > 4096 || l.n_glob > 8192) return 0;
    const int B = (int)l.b_glob, N = (int)l.n_glob;
    RsOff r{};
    r.TS = RS_SLOTS - l.kmax;
    r.nchunks = (B + r.TS - 1) / r.TS;
    r.packp = rs_rup((int)pack, 64);
    {
        const int right1 = l.l1 - 1 - l.pad1;
        r.c0 = l.l2 > 0 && l.p2 > right1 ? l.p2 : right1;
        r.o1 = r.c0 + l.pad1;
        r.o2 = r.c0 - l.p2;
        if (r.o1 > 31 || (l.l2 > 0 && r.o2 + l.l2 > 32)) return 0;  // both kernels must fit one 32-sample window
    }
    int off = 0;
    auto take = [&](int n) { int at = off; off += rs_rup(n, 4); return at; };
    r.y = take(RS_HL + N + 32 * l.S + 64);
    r.x = take(rs_rup(B, 64) + 64);
    r.gx = take(rs_rup(B, 64) + 64);
    r.MP = rs_rup(8 + (r.nchunks - 1) * r.TS + RS_SLOTS + 8, 4);
    r.mom = take(4 * r.MP);
    r.GP = rs_rup(B, 4);
    r.gm = take(4 * r.GP);
    r.q = take(l.C <= RS_QCACHE ? RS_QCACHE * r.GP : 4);
    {   // per-window gradient image of the ELBO stage; later the cross-lane reduction image (64 slots, pitch 97) and the
        // cross-wave image of the equaliser-gradient tiles
        int nd = 3 * 8 * RS_SLOTS;
        if (nd < RS_NW * 16 * 97) nd = RS_NW * 16 * 97;
        if (nd < RS_NW * 1024 + RS_NW * 32) nd = RS_NW * 1024 + RS_NW * 32;
        r.D = take(nd);
    }
    r.vl = take(RS_SLOTS);
    r.H = take(l.S * 64);
    r.HH = take(l.S * 64);
    r.h1c = take(l.S * 8);
    r.d = take(l.S * 8);
    r.dq = take(l.S * 8);
    r.par = take(r.packp);
    r.grad = take(r.packp);
    r.gtot = take(l.S * RS_GREC);
    r.P = take(l.S * 64);
    r.misc = take(RS_NT + 64 * RS_NW);
    r.dbl = take(2 * 16);
    r.total = off;
    if ((int64_t)r.total * 4 > LDS_BUDGET) return 0;
    *o = r;
    return 1;
}

// Lane exchanges and wave sums on the vector pipe (DPP modifiers inside the rows of 16 lanes, half / quarter swaps across them):
// a shuffle (`__shfl_xor`) is an LDS-pipeline instruction that the wave waits for, and this kernel runs one wave per SIMD.
template <int CTRL>
__device__ __forceinline__ float rs_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}
// the values of lanes ^ 16 (Q = 16) or ^ 32 (Q = 32): after the swap `a` holds, in both rows (halves) of a pair, the value of the even
// one and `b` that of the odd one
template <int Q>
__device__ __forceinline__ void rs_swap_pair(float v, float& a, float& b) {
    a = v; b = v;
    if (Q == 16) asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    else asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
}
template <int OFF>
__device__ __forceinline__ float rs_lane_xor(float v, int lane) {   // value of lane ^ OFF
    if (OFF == 1) return rs_dpp<0xB1>(v);                       // quad_perm [1,0,3,2]
    if (OFF == 2) return rs_dpp<0x4E>(v);                       // quad_perm [2,3,0,1]
    if (OFF == 4) return rs_dpp<0x1B>(rs_dpp<0x141>(v));        // row_half_mirror (^ 7), then quad_perm [3,2,1,0] (^ 3)
    if (OFF == 8) return rs_dpp<0x141>(rs_dpp<0x140>(v));       // row_mirror (^ 15), then row_half_mirror (^ 7)
    float a, b;
    rs_swap_pair<OFF>(v, a, b);
    return (lane & OFF) ? a : b;
}
template <int OFF>
__device__ __forceinline__ float rs_add_xor(float v) {      // v + (value of lane ^ OFF)
    if (OFF < 16) return v + rs_lane_xor<OFF>(v, 0);
    float a, b;
    rs_swap_pair<OFF>(v, a, b);
    return a + b;
}
__device__ __forceinline__ float rs_wave_sum(float v) {      // total over the 64 lanes, on every lane
    v = rs_add_xor<1>(v); v = rs_add_xor<2>(v); v = rs_add_xor<4>(v); v = rs_add_xor<8>(v);
    v = rs_add_xor<16>(v);
    return rs_add_xor<32>(v);
}
__device__ __forceinline__ double rs_wave_sum(double v) {
    auto step = [&](auto mover) {
        const int lo = mover(__double2loint(v)), hi = mover(__double2hiint(v));
        v += __hiloint2double(hi, lo);
    };
    auto ix = [](float f) { return __float_as_int(f); };
    auto fx = [](int i) { return __int_as_float(i); };
    step([&](int w) { return ix(rs_lane_xor<1>(fx(w), 0)); });
    step([&](int w) { return ix(rs_lane_xor<2>(fx(w), 0)); });
    step([&](int w) { return ix(rs_lane_xor<4>(fx(w), 0)); });
    step([&](int w) { return ix(rs_lane_xor<8>(fx(w), 0)); });
    const int lane = (int)(threadIdx.x & 63);
    step([&](int w) { return ix(rs_lane_xor<16>(fx(w), lane)); });
    step([&](int w) { return ix(rs_lane_xor<32>(fx(w), lane)); });
    return v;
}

// Sum over the 64 lanes of N per-lane values each, N a power of two <= 64.  Returns in lane `ln` the total of element
// e(ln) = sum over s < log2(N) of bit_s(ln) * (N >> (s + 1)); costs N - 1 exchanges instead of N * 6.
template <int N>
__device__ __forceinline__ float rs_wave_reduce_transposed(float (&x)[N], int lane) {
    int sft = 0;
#pragma unroll
    for (int n = N; n > 1; n >>= 1, ++sft) {
        const bool up = (lane >> sft) & 1;
#pragma unroll
        for (int i = 0; i < n / 2; ++i) {
            const float lo = x[i], hi = x[i + n / 2];
            const float send = up ? lo : hi, keep = up ? hi : lo;
            float got;
            switch (sft) {
                case 0: got = rs_lane_xor<1>(send, lane); break;
                case 1: got = rs_lane_xor<2>(send, lane); break;
                case 2: got = rs_lane_xor<4>(send, lane); break;
                case 3: got = rs_lane_xor<8>(send, lane); break;
                case 4: got = rs_lane_xor<16>(send, lane); break;
                default: got = rs_lane_xor<32>(send, lane); break;
            }
            x[i] = keep + got;
        }
    }
    float v = x[0];
    if (N <= 1) v = rs_add_xor<1>(v);
    if (N <= 2) v = rs_add_xor<2>(v);
    if (N <= 4) v = rs_add_xor<4>(v);
    if (N <= 8) v = rs_add_xor<8>(v);
    if (N <= 16) v = rs_add_xor<16>(v);
    if (N <= 32) v = rs_add_xor<32>(v);
    return v;
}

__device__ __forceinline__ float rs_quad_xor1(float v) {  // value of lane ^ 1 (DPP quad_perm [1,0,3,2])
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float rs_quad_xor2(float v) {  // value of lane ^ 2 (DPP quad_perm [2,3,0,1])
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
}

template <int CTRL>
__device__ __forceinline__ float rs_quad_perm(float v) {  // DPP quad_perm: 0xB1 = lane ^ 1, 0x4E = lane ^ 2, 0x1B = lane ^ 3
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true));
}

// Outer-product accumulation G += a mu^T summed over the four lanes of a quad, kept in folded form: lane q = lane & 3 owns
// rows 2q, 2q + 1 (16 of the 64 elements), so every lane fetches from its three quad partners the two `a` values of
// ITS rows (the sender selects them) and the partner's eight mu values, all through DPP -- 64 FMAs per lane as for a
// private 8x8 accumulator, but 16 registers instead of 64 and no fold afterwards.
template <int CTRL>
__device__ __forceinline__ void rs_quad_outer(float (&G2)[2][8], const float (&ak)[8], const float (&mu)[8], int q) {
    const int idx = CTRL == 0 ? q : (CTRL == 0xB1 ? q ^ 1 : (CTRL == 0x4E ? q ^ 2 : q ^ 3));  // receiver's q, seen from the sender
    float r[2];
#pragma unroll
    for (int odd = 0; odd < 2; ++odd) {
        const float lo = (idx & 2) ? ak[4 + odd] : ak[odd];
        const float hi = (idx & 2) ? ak[6 + odd] : ak[2 + odd];
        const float v = (idx & 1) ? hi : lo;
        r[odd] = CTRL == 0 ? v : rs_quad_perm<CTRL == 0 ? 0xB1 : CTRL>(v);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float m = CTRL == 0 ? mu[j] : rs_quad_perm<CTRL == 0 ? 0xB1 : CTRL>(mu[j]);
        G2[0][j] = fmaf(r[0], m, G2[0][j]);
        G2[1][j] = fmaf(r[1], m, G2[1][j]);
    }
}

// Two transposing butterfly steps inside each quad of lanes: afterwards x[0 .. N/4) hold, summed over the lane's quad,
// the elements e = bit0(lane) * N/2 + bit1(lane) * N/4 + i.  Register-only (DPP), no LDS traffic.
template <int N>
__device__ __forceinline__ void rs_quad_fold(float (&x)[N], int lane) {
    const bool up0 = lane & 1, up1 = lane & 2;
#pragma unroll
    for (int i = 0; i < N / 2; ++i) {
        const float lo = x[i], hi = x[i + N / 2];
        x[i] = (up0 ? hi : lo) + rs_quad_xor1(up0 ? lo : hi);
    }
#pragma unroll
    for (int i = 0; i < N / 4; ++i) {
        const float lo = x[i], hi = x[i + N / 4];
        x[i] = (up1 ? hi : lo) + rs_quad_xor2(up1 ? lo : hi);
    }
}

template <int N>
__device__ __forceinline__ int rs_transposed_index(int lane) {
    int e = 0, sft = 0;
#pragma unroll
    for (int n = N; n > 1; n >>= 1, ++sft) e += ((lane >> sft) & 1) * (n >> 1);
    return e;
}

// total over the workgroup, returned to every thread; `red` holds RS_NW doubles
__device__ __forceinline__ double rs_block_sum(double v, double* red, int tid) {
    v = rs_wave_sum(v);
    __syncthreads();
    if ((tid & 63) == 0) red[tid >> 6] = v;
    __syncthreads();
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < RS_NW; ++i) s += red[i];
    return s;
}

// compact tap k of phase r -> channel tap index or -1
__device__ __forceinline__ int rs_tap(const Lay& l, int r, int k) {
    if (k >= l.LE) return -1;
    const int i = r + l.L - 1 - k * l.S;
    return (i >= 0 && i < l.L) ? i : -1;
}

// Diagnostic builds (-DV2VAE_STAMPS) record the constant-rate wall clock (10 ns ticks) of thread 0 of model 0 at the
// stage boundaries of the last batch into the (otherwise unused) workspace.
#ifdef V2VAE_STAMPS
#define RS_STAMP(i)                                                                                        \
    do {                                                                                                   \
        if (tid == 0 && model == 0 && n == a.n_batches - 1) reinterpret_cast<unsigned long long*>(a.ws)[i] = wall_clock64(); \
    } while (0)
#else
#define RS_STAMP(i) do { } while (0)
#endif
#define RS_SUB(i, cond) do { if (cond) RS_STAMP(i); } while (0)

template <int CB>
__global__ __launch_bounds__(RS_NT) void k_fit_rs(Lay l, RsOff o, v2vae_fit_args a, int n0, int n1, int n2, int n3, int n4) {
    extern __shared__ __align__(16) float sm[];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int model = blockIdx.x;
    const int B = (int)l.b_glob, N = (int)l.n_glob, S = l.S, L = l.L, l1 = l.l1, l2 = l.l2, C = l.C;
    const int kmax = l.kmax, LE = l.LE, MP = o.MP, GP = o.GP;
    const int W = (int)l.W;
    const int pack = n0 + n1 + n2 + n3 + n4;
    const int nadam = n0 + n1 + n2 + n3 + (a.learn_scaling ? n4 : 0);
    float* s_y = sm + o.y;
    float* s_x = sm + o.x;
    float* s_gx = sm + o.gx;
    float* s_mom = sm + o.mom;
    float* s_gm = sm + o.gm;
    float* s_q = sm + o.q;
    float* s_D = sm + o.D;
    float* s_vl = sm + o.vl;
    float* s_H = sm + o.H;
    float* s_HH = sm + o.HH;
    float* s_h1c = sm + o.h1c;
    float* s_d = sm + o.d;
    float* s_dq = sm + o.dq;
    float* s_par = sm + o.par;
    float* s_grad = sm + o.grad;
    float* s_gtot = sm + o.gtot;
    float* s_P = sm + o.P;
    float* s_misc = sm + o.misc;
    double* s_dbl = reinterpret_cast<double*>(sm + o.dbl);
    const float* p_h1 = s_par;
    const float* p_h2 = s_par + n0;
    const float* p_k1 = s_par + n0 + n1;
    const float* p_k2 = s_par + n0 + n1 + n2;
    const float* p_sc = s_par + n0 + n1 + n2 + n3;

    // ---- state into registers; LDS cleared once (halos and paddings stay zero for the whole fit)
    float pr[RS_APT], am[RS_APT], av[RS_APT], avm[RS_APT];
    const int64_t gbase = (int64_t)model * pack;
#pragma unroll
    for (int q = 0; q < RS_APT; ++q) {
        const int idx = tid + q * RS_NT;
        const bool in = idx < pack;
        pr[q] = in ? a.params[gbase + idx] : 0.f;
        am[q] = in ? a.adam_m[gbase + idx] : 0.f;
        av[q] = in ? a.adam_v[gbase + idx] : 0.f;
        avm[q] = in ? a.adam_vmax[gbase + idx] : 0.f;
    }
    for (int i = tid; i < o.total; i += RS_NT) sm[i] = 0.f;
    float nv = a.noise_var[model];
    double E = 0.0, KL = 0.0, loss = 0.0, lamd = 0.0;
    int bad_steps = 0;   // batches whose E was not a positive finite number or whose KL was not finite (failure detection)
    const float w1 = (float)(1.0 - (double)a.beta1), w2 = (float)(1.0 - (double)a.beta2);
    const float* ym = a.y + (int64_t)model * a.y_model_stride;
    const float* levels = a.levels;
    const float* prior = a.prior;
    __syncthreads();
    const int chan = l.chan;
    // a noise-scaling gradient that is never reset (lib/real_equalization.py:1169-1178, :1356-1364) starts from the caller's sum
    if (a.accumulate_scaling_grad && tid < n4) s_grad[n0 + n1 + n2 + n3 + tid] = a.grads[gbase + n0 + n1 + n2 + n3 + tid];
    const bool y_ahead = N <= RS_YPT * RS_NT;
    if (y_ahead)
        for (int i = tid; i < N; i += RS_NT) s_y[RS_HL + i] = ym[i];

    for (int n = 0; n < a.n_batches; ++n) {
        const float* yb = ym + (int64_t)n * N;
        const float step_size = a.step_size[n], bc2_sqrt = a.bc2_sqrt[n];  // requested early: used by the last stage
        RS_STAMP(0);
#ifdef V2VAE_STAMPS
        if (tid == 0 && model == 0 && n == a.n_batches - 1) reinterpret_cast<unsigned long long*>(a.ws)[12] = clock64();
#endif
        // ---- parameters and samples to LDS
#pragma unroll
        for (int q = 0; q < RS_APT; ++q) {
            const int idx = tid + q * RS_NT;
            if (idx < pack) s_par[idx] = pr[q];
        }
        // samples: short batches are fetched one step ahead into registers (the fetch latency hides behind the whole step)
        float ynx[RS_YPT];
        if (!y_ahead) {
            for (int i = tid; i < N; i += RS_NT) s_y[RS_HL + i] = yb[i];
        } else if (n + 1 < a.n_batches) {
#pragma unroll
            for (int q = 0; q < RS_YPT; ++q) {
                const int i = tid + q * RS_NT;
                ynx[q] = i < N ? yb[N + i] : 0.f;
            }
        }
        __syncthreads();

        RS_STAMP(1);
        // ---- channel tables of every phase (compact active taps)
        for (int idx = tid; idx < S * 64; idx += RS_NT) {
            const int r = idx >> 6, k = (idx >> 3) & 7, j = idx & 7;
            const int ta = rs_tap(l, r, k), tb = rs_tap(l, r, j);
            float hv = 0.f;
            if (ta >= 0 && tb >= 0) {
                if (chan == 0) hv = p_h2[ta * L + tb] + p_h2[tb * L + ta];
                else if (ta == tb) hv = chan == 2 ? p_h2[1] * p_h1[ta] : p_h2[ta];  // diagonal kernel of the polynomial channels
            }
            s_H[idx] = hv;
            s_HH[idx] = hv * hv;
        }
        for (int idx = tid; idx < S * 8; idx += RS_NT) {
            const int r = idx >> 3, k = idx & 7;
            const int ta = rs_tap(l, r, k);
            if (chan == 0) {
                const float dg = ta >= 0 ? p_h2[ta * L + ta] : 0.f;
                s_h1c[idx] = ta >= 0 ? p_h1[ta] : 0.f;
                s_d[idx] = 2.f * dg;
                s_dq[idx] = dg;  // lib/real_equalization.py:869 uses diag(h2), not diag(h2 + h2^T)
            } else {
                // memory polynomial (:1333-1334): a = h1, b = h2; Hammerstein (:1146-1147): a = alpha1 h, b = alpha2 h
                const float av_ = ta < 0 ? 0.f : (chan == 2 ? p_h2[0] * p_h1[ta] : p_h1[ta]);
                const float bv_ = ta < 0 ? 0.f : (chan == 2 ? p_h2[1] * p_h1[ta] : p_h2[ta]);
                s_h1c[idx] = av_;
                s_d[idx] = bv_;
                s_dq[idx] = bv_;
            }
        }

        RS_STAMP(2);
        // ---- equaliser forward (lib/torch_components.py:79-96): per 32-symbol tile Z = Kw X on the matrix cores over a
        // 32-sample window that holds both kernels, then x = sum_i (Z_i + k1w_i) X_i
        {
            const int irow = lane & 31, half = lane >> 5;
            float Areg[16], k1w[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int ia = irow - o.o2, ib = 2 * s + half - o.o2;  // lags of (row, contraction) positions
                const bool in2 = ia >= 0 && ia < l2 && ib >= 0 && ib < l2;
                const float ka = p_k2[in2 ? ia * l2 + ib : 0] + p_k2[in2 ? ib * l2 + ia : 0];  // index 0 is always inside s_par
                Areg[s] = in2 ? ka : 0.f;
                const int row = (s & 3) + 8 * (s >> 2) + 4 * half;
                const int j1 = o.o1 - row;
                const bool in1 = j1 >= 0 && j1 < l1;
                const float kb = p_k1[in1 ? j1 : 0];
                k1w[s] = in1 ? kb : 0.f;
            }
            const int g_lo = l2 > 0 ? o.o2 / 8 : 4, g_hi = l2 > 0 ? (o.o2 + l2 - 1) / 8 : -1;  // groups of 4 contraction steps
            const int ntile = (B + 31) / 32;
            for (int tile = wv; tile < ntile; tile += RS_NW) {
                const int b = tile * 32 + irow;
                const int base = RS_HL + b * S + o.c0;
                float xb[16], xr[16];
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    xb[s] = s_y[base - (2 * s + half)];
                    xr[s] = s_y[base - ((s & 3) + 8 * (s >> 2) + 4 * half)];
                }
                rs_f16 acc;
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    if (g >= g_lo && g <= g_hi) {
#pragma unroll
                        for (int s = 4 * g; s < 4 * g + 4; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Areg[s], xb[s], acc, 0, 0, 0);
                    }
                float part = 0.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) part = fmaf(acc[q] + k1w[q], xr[q], part);
                part = rs_add_xor<32>(part);
                if (half == 0 && b < B) s_x[b] = part;
            }
        }
        __syncthreads();

        RS_STAMP(3);
        // ---- soft demapper, moments, KL (lib/real_equalization.py:816-823, 851-866, 915-918)
        {
            double kl = 0.0;
            double inv_den[MAX_C];
            posterior_temperatures<CB>(p_sc, C, nv, inv_den);
            float prk[MAX_C];
#pragma unroll
            for (int k = 0; k < CB; ++k) prk[k] = k < C ? 1.f / prior[k] : 1.f;  // inverse priors
            for (int b = tid; b < B; b += RS_NT) {
                const float x = s_x[b];
                Posterior p;
                posterior<CB>(x, levels, inv_den, C, p);
                float* mo = s_mom + 8 + b;
                mo[0] = p.mu; mo[MP] = p.var; mo[2 * MP] = p.kap; mo[3 * MP] = p.gam;
                if (CB <= RS_QCACHE) {
#pragma unroll
                    for (int k = 0; k < CB; ++k) s_q[k * GP + b] = p.q[k];
                }
                float klb = 0.f;
#pragma unroll
                for (int k = 0; k < CB; ++k)
                    if (k < C) klb += p.q[k] * __logf(fmaf(p.q[k], prk[k], 1e-12f));
                if (b >= l.kl_lo && b < l.kl_hi) kl += (double)klb;
                if (a.y_eq) a.y_eq[(int64_t)model * a.n_batches * B + (int64_t)n * B + b] = x;
            }
            KL = rs_block_sum(kl, s_dbl, tid);  // its barriers also publish the moments
        }

        RS_STAMP(4);
        // ---- lagged products c_d(m) = v(m) v(m + d): totals over the batch
        {
            float c[8];
#pragma unroll
            for (int d = 0; d < 8; ++d) c[d] = 0.f;
            for (int b = tid; b < B; b += RS_NT) {
                const float* vb = s_mom + MP + 8 + b;
                const float v0 = vb[0];
#pragma unroll
                for (int d = 0; d < 8; ++d) c[d] = fmaf(v0, vb[d], c[d]);
            }
            const float tot = rs_wave_reduce_transposed<8>(c, lane);
            if (lane < 8) s_misc[RS_NT + wv * 8 + rs_transposed_index<8>(lane)] = tot;
        }
        __syncthreads();

        // ---- sum_n v v^T per phase from the lag totals minus the edge terms
        for (int idx = tid; idx < S * 64; idx += RS_NT) {
            const int r = idx >> 6, k = (idx >> 3) & 7, j = idx & 7;
            const int d = k > j ? k - j : j - k, mn = k < j ? k : j;
            const int Tr = W > r ? (W - r + S - 1) / S : 0;  // windows of this phase
            float tw[RS_NW];
#pragma unroll
            for (int w = 0; w < RS_NW; ++w) tw[w] = s_misc[RS_NT + w * 8 + d];
            const float* vb = s_mom + MP + 8;
            // edge terms: m < mn at the head (mn <= 7), m >= Tr + mn at the tail (a handful: B - Tr is about (L - 1) / S);
            // fixed-length predicated loops keep the LDS reads independent of each other
            float hd[8], tl[16];
#pragma unroll
            for (int m = 0; m < 8; ++m) hd[m] = vb[m] * vb[m + d];
            const int t0 = Tr + mn;
#pragma unroll
            for (int i = 0; i < 16; ++i) tl[i] = vb[t0 + i] * vb[t0 + i + d];   // beyond the batch the moments read as zero
            double tot = 0.0, edge = 0.0;
#pragma unroll
            for (int w = 0; w < RS_NW; ++w) tot += (double)tw[w];
#pragma unroll
            for (int m = 0; m < 8; ++m) edge += m < mn ? (double)hd[m] : 0.0;
#pragma unroll
            for (int i = 0; i < 16; ++i) edge += t0 + i < B ? (double)tl[i] : 0.0;
            for (int m = t0 + 16; m < B; ++m) edge += (double)(vb[m] * vb[m + d]);
            s_P[idx] = (float)(tot - edge);
        }

        RS_STAMP(5);
        // ---- ELBO expectations, one window per lane (lib/real_equalization.py:868-913 in window form)
        double Eacc = 0.0;
        for (int r = 0; r < S; ++r) {
            float G2[2][8], F[32];
#pragma unroll
            for (int j = 0; j < 8; ++j) G2[0][j] = G2[1][j] = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) F[i] = 0.f;
            // tables of this phase in registers, as pairs: H is symmetric, so "row j" is also "column j" and
            // u += H[:, j] mu[j] runs as packed two-wide FMAs over adjacent elements of row j
            rs_f2 Hr[8][4], h1p[4], dp[4], dqp[4];
            const rs_f2* gq = reinterpret_cast<const rs_f2*>(s_HH + r * 64);  // (H.H) rows are read where used
            {
                const rs_f2* hq = reinterpret_cast<const rs_f2*>(s_H + r * 64);
#pragma unroll
                for (int j = 0; j < 8; ++j)
#pragma unroll
                    for (int kp = 0; kp < 4; ++kp) Hr[j][kp] = hq[j * 4 + kp];
                const rs_f2* t1 = reinterpret_cast<const rs_f2*>(s_h1c + r * 8);
                const rs_f2* t2 = reinterpret_cast<const rs_f2*>(s_d + r * 8);
                const rs_f2* t3 = reinterpret_cast<const rs_f2*>(s_dq + r * 8);
#pragma unroll
                for (int kp = 0; kp < 4; ++kp) { h1p[kp] = t1[kp]; dp[kp] = t2[kp]; dqp[kp] = t3[kp]; }
            }
            for (int c = 0; c < o.nchunks; ++c) {
                RS_SUB(14, r == 0 && c == 0);
                const int s0 = c * o.TS;
                const int s1 = s0 + o.TS < B ? s0 + o.TS : B;
#pragma unroll 1
                for (int sub = 0; sub < RS_SLOTS / RS_NT; ++sub) {
                const int slot_w = tid + sub * RS_NT;
                const int t = s0 - kmax + slot_w;
                const int nn = t * S + r;
                const bool valid = (t < s1) && nn >= 0 && nn < W;
                const bool own = valid && t >= s0;
                const float vl = valid ? 1.f : 0.f;
                const float yraw = s_y[RS_HL + l.pre + (valid ? nn : 0)];
                const float yv = valid ? yraw : 0.f;
                const float* mb = s_mom + 8 + t;
                float mu[8], vv[8], kp[8], gm[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { mu[k] = mb[k]; vv[k] = mb[MP + k]; kp[k] = mb[2 * MP + k]; gm[k] = mb[3 * MP + k]; }
                // pass 1: u = H mu, tt = (H.H) v
                rs_f2 u2[4], t2[4];
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) { u2[kq] = Hr[0][kq] * mu[0]; t2[kq] = gq[kq] * vv[0]; }
#pragma unroll
                for (int j = 1; j < 8; ++j)
#pragma unroll
                    for (int kq = 0; kq < 4; ++kq) { u2[kq] += Hr[j][kq] * mu[j]; t2[kq] += gq[j * 4 + kq] * vv[j]; }
                float z[8], tt[8], wq[8];
                float pa = 0.f, pq = 0.f, ps = 0.f, psq = 0.f, pvs = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float u = u2[k >> 1][k & 1], tk = t2[k >> 1][k & 1];
                    const float h1k = h1p[k >> 1][k & 1], dk = dp[k >> 1][k & 1], dqk = dqp[k >> 1][k & 1];
                    const float zk = fmaf(2.f, u, h1k);
                    pa = fmaf(h1k, mu[k], pa);
                    pq = fmaf(mu[k], u, pq);
                    ps = fmaf(vv[k], dk, ps);
                    psq = fmaf(vv[k], dqk, psq);
                    pvs += vv[k] * zk * zk + 2.f * vv[k] * tk + 2.f * kp[k] * dk * zk + gm[k] * dk * dk;
                    z[k] = zk; tt[k] = tk;
                    wq[k] = 2.f * vv[k] * zk + 2.f * kp[k] * dk;
                }
                const float A = pa + pq + ps;
                const float e = A - yv;
                const float En = e * e + 2.f * yv * (ps - psq) + pvs;
                if (own) Eacc += (double)En;
                RS_SUB(15, r == 0 && c == 0);
                const float wo = own ? 1.f : 0.f;
                // pass 2: H w, per-(window, tap) gradients of the moments, parameter-gradient integrands of owned windows
                rs_f2 hw2[4];
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) hw2[kq] = Hr[0][kq] * wq[0];
#pragma unroll
                for (int j = 1; j < 8; ++j)
#pragma unroll
                    for (int kq = 0; kq < 4; ++kq) hw2[kq] += Hr[j][kq] * wq[j];
                float ak[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float hw = hw2[k >> 1][k & 1];
                    const float dk = dp[k >> 1][k & 1], dqk = dqp[k >> 1][k & 1];
                    s_D[(0 * 8 + k) * RS_SLOTS + slot_w] = vl * (2.f * e * z[k] + 2.f * hw);
                    s_D[(1 * 8 + k) * RS_SLOTS + slot_w] = vl * (2.f * A * dk - 2.f * yv * dqk + z[k] * z[k] + 4.f * tt[k]);
                    s_D[(2 * 8 + k) * RS_SLOTS + slot_w] = vl * (2.f * dk * z[k]);
                    ak[k] = wo * fmaf(e, mu[k], wq[k]);
                    F[k] += wo * (2.f * e * mu[k] + wq[k]);
                    F[8 + k] += wo * (2.f * A * vv[k] + 2.f * kp[k] * z[k] + 2.f * gm[k] * dk);
                    F[16 + k] += wo * (-2.f * yv * vv[k]);
                }
                RS_SUB(16, r == 0 && c == 0);
                rs_quad_outer<0>(G2, ak, mu, lane & 3);
                rs_quad_outer<0xB1>(G2, ak, mu, lane & 3);
                rs_quad_outer<0x4E>(G2, ak, mu, lane & 3);
                rs_quad_outer<0x1B>(G2, ak, mu, lane & 3);
                s_vl[slot_w] = vl;
                }
                RS_SUB(17, r == 0 && c == 0);
                __syncthreads();
                RS_SUB(18, r == 0 && c == 0);
                // gather: symbol m = s0 + ml collects window t = m - kk at compact tap kk
                for (int sub = 0; sub < RS_SLOTS / RS_NT; ++sub) {
                const int ml = tid + sub * RS_NT;
                if (ml < o.TS && s0 + ml < s1) {
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                    // four taps at a time: their sixteen reads are requested together (with one wave per SIMD a read waited for
                    // where it is used is idle time; the register budget does not allow all eight taps at once)
#pragma unroll
                    for (int k4 = 0; k4 < 8; k4 += 4) {
                        float d0[4], d1[4], d2[4], vs[4], dk[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int kk = k4 + u;
                            const int slot = kk < LE ? ml + kmax - kk : 0;
                            d0[u] = s_D[(0 * 8 + kk) * RS_SLOTS + slot]; d1[u] = s_D[(1 * 8 + kk) * RS_SLOTS + slot];
                            d2[u] = s_D[(2 * 8 + kk) * RS_SLOTS + slot]; vs[u] = s_vl[slot];
                            dk[u] = s_d[r * 8 + kk];
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const bool act = k4 + u < LE;
                            a0 += act ? d0[u] : 0.f;
                            a1 += act ? d1[u] : 0.f;
                            a2 += act ? d2[u] : 0.f;
                            a3 = fmaf(act ? vs[u] : 0.f, dk[u] * dk[u], a3);
                        }
                        __builtin_amdgcn_sched_group_barrier(0x100, 20, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 24, 0);
                    }
                    float* gp = s_gm + s0 + ml;
                    if (r == 0) { gp[0] = a0; gp[GP] = a1; gp[2 * GP] = a2; gp[3 * GP] = a3; }
                    else {   // the four reads before the first write: one LDS round trip instead of four (one wave per SIMD)
                        const float o0 = gp[0], o1 = gp[GP], o2 = gp[2 * GP], o3 = gp[3 * GP];
                        gp[0] = o0 + a0; gp[GP] = o1 + a1; gp[2 * GP] = o2 + a2; gp[3 * GP] = o3 + a3;
                    }
                }
                }
                __syncthreads();
            }
            RS_SUB(19, r == 0);
            // cross-lane sums: G is already folded over each quad, F takes two butterfly steps in registers; then
            // [slot = quad][element] through LDS (pitch 97) and a fixed-order sum over the RS_NW * 16 slots
            rs_quad_fold<32>(F, lane);
            {
                float* R = s_D + (wv * 16 + (lane >> 2)) * 97;
                const int eg = (lane & 3) * 16, ef = 64 + (lane & 1) * 16 + ((lane >> 1) & 1) * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) { R[eg + j] = G2[0][j]; R[eg + 8 + j] = G2[1][j]; }
#pragma unroll
                for (int i = 0; i < 8; ++i) R[ef + i] = F[i];
            }
            __syncthreads();
            {
                constexpr int PER = RS_NW * 16 / RS_PARTS;  // slots per thread
                const int e = tid / RS_PARTS, part = tid % RS_PARTS;
                const float* Rp = s_D + part * PER * 97 + (e < RS_GREC ? e : 0);
                float rv[PER];
#pragma unroll
                for (int i = 0; i < PER; ++i) rv[i] = Rp[i * 97];   // all loads in flight before the first add
                float sacc = 0.f;
#pragma unroll
                for (int i = 0; i < PER; ++i) sacc += rv[i];
                sacc += rs_quad_xor1(sacc);
                if (RS_PARTS == 4) sacc += rs_quad_xor2(sacc);
                if (e < RS_GREC && part == 0) s_gtot[r * RS_GREC + e] = sacc;
            }
            __syncthreads();
            RS_SUB(20, r == 0);
        }
        RS_STAMP(6);
        E = rs_block_sum(Eacc, s_dbl, tid);
        loss = KL + (double)W * log(E);
        lamd = (double)W / E;
        if (!(E > 0.0) || !isfinite(E) || !isfinite(KL)) ++bad_steps;
        const float lam = (float)lamd;

        RS_STAMP(7);
        // ---- per-symbol backward: moments -> posterior -> equaliser output, noise scaling.
        // Softmax backward in the pairwise form dz_k = q_k sum_j q_j (v_k - v_j): every term is a product of a (possibly
        // tiny) probability with an O(1) difference, so there is no cancellation against the posterior mean of v and
        // single precision suffices.
        {
            float gs[MAX_C];
#pragma unroll
            for (int k = 0; k < CB; ++k) gs[k] = 0.f;
            double inv_den[MAX_C];
            posterior_temperatures<CB>(p_sc, C, nv, inv_den);
            float idn[MAX_C], idsc[MAX_C], lvk[MAX_C], prk[MAX_C], epr[MAX_C];
#pragma unroll
            for (int k = 0; k < CB; ++k) {
                idn[k] = (float)inv_den[k];
                idsc[k] = k < C ? (float)(inv_den[k] / (double)p_sc[k]) : 0.f;
                lvk[k] = k < C ? levels[k] : 0.f;
                prk[k] = k < C ? 1.f / prior[k] : 1.f;    // inverse prior
                epr[k] = k < C ? 1e-12f * prior[k] : 1.f;
            }
            for (int b = tid; b < B; b += RS_NT) {
                const float x = s_x[b];
                float q[MAX_C], pmu, pvar, pkap;
                if (CB <= RS_QCACHE) {  // posterior kept from the forward stage
#pragma unroll
                    for (int k = 0; k < CB; ++k) q[k] = s_q[k * GP + b];
                    const float* mo = s_mom + 8 + b;
                    pmu = mo[0]; pvar = mo[MP]; pkap = mo[2 * MP];
                } else {
                    Posterior p;
                    posterior<CB>(x, levels, inv_den, C, p);
#pragma unroll
                    for (int k = 0; k < CB; ++k) q[k] = p.q[k];
                    pmu = p.mu; pvar = p.var; pkap = p.kap;
                }
                const float* g = s_gm + b;
                const float gmu = g[0], gv = g[GP], gk = g[2 * GP], gg = g[3 * GP];
                const bool in_kl = b >= l.kl_lo && b < l.kl_hi;
                float v[MAX_C];
#pragma unroll
                for (int k = 0; k < CB; ++k) {
                    const float d = lvk[k] - pmu, d2 = d * d;
                    float vk = gmu * d + gv * d2 + gk * (d2 * d - 3.f * pvar * d) + gg * (d2 * d2 - 4.f * pkap * d - 6.f * pvar * d2);
                    vk *= lam;
                    if (in_kl) vk += __logf(fmaf(q[k], prk[k], 1e-12f)) + q[k] * __builtin_amdgcn_rcpf(q[k] + epr[k]);
                    v[k] = vk;
                }
                float gxa = 0.f;
#pragma unroll
                for (int k = 0; k < CB; ++k)
                    if (k < C) {
                        float sk = 0.f;
#pragma unroll
                        for (int j = 0; j < CB; ++j)
                            if (j < C && j != k) sk = fmaf(q[j], v[k] - v[j], sk);
                        const float dz = q[k] * sk;
                        const float df = x - lvk[k];
                        gxa = fmaf(dz, -2.f * df * idn[k], gxa);
                        gs[k] = fmaf(dz * df, df * idsc[k], gs[k]);
                    }
                s_gx[b] = gxa;
            }
#pragma unroll
            for (int k = 0; k < CB; ++k) {
                const float v = rs_wave_sum(gs[k]);
                if (lane == 0 && k < C) s_misc[wv * MAX_C + k] = v;
            }
        }
        __syncthreads();

        RS_STAMP(8);
        // ---- channel gradients: chain from compact per-phase totals to h1, h2
        for (int idx = tid; idx < n0 + n1; idx += RS_NT) {
            auto compact = [&](int t, int& r, int& k) {
                int rr = (t - (L - 1)) % S;
                if (rr < 0) rr += S;
                r = rr;
                k = (rr + L - 1 - t) / S;
            };
            // polynomial channels: dE/da_t and dE/db_t (b = diagonal of the kernel, exact diagonal in the mean)
            auto ga = [&](int t) { int r, k; compact(t, r, k); return s_gtot[r * RS_GREC + 64 + k]; };
            auto gb = [&](int t) {
                int r, k;
                compact(t, r, k);
                const float* T = s_gtot + r * RS_GREC;
                return 2.f * T[k * 8 + k] + 4.f * s_H[r * 64 + k * 8 + k] * s_P[r * 64 + k * 8 + k] + T[64 + 8 + k] + T[64 + 16 + k];
            };
            if (chan == 1) {
                s_grad[idx] = lam * (idx < n0 ? ga(idx) : gb(idx - n0));
            } else if (chan == 2) {
                if (idx < n0) s_grad[idx] = lam * (p_h2[0] * ga(idx) + p_h2[1] * gb(idx));
                else {
                    float acc = 0.f;
                    for (int t = 0; t < L; ++t) acc = fmaf(p_h1[t], idx == n0 ? ga(t) : gb(t), acc);
                    s_grad[idx] = lam * acc;
                }
            } else if (idx < n0) {
                int r, k;
                compact(idx, r, k);
                s_grad[idx] = lam * s_gtot[r * RS_GREC + 64 + k];
            } else {
                const int e = idx - n0;
                const int i = e / L, j = e - i * L;
                int ri, ki, rj, kj;
                compact(i, ri, ki);
                compact(j, rj, kj);
                float v = 0.f;
                if (ri == rj) {
                    const float* T = s_gtot + ri * RS_GREC;
                    const float hp = 4.f * s_H[ri * 64 + ki * 8 + kj] * s_P[ri * 64 + ki * 8 + kj];
                    v = (2.f * T[ki * 8 + kj] + hp) + (2.f * T[kj * 8 + ki] + hp);
                    if (i == j) v += 2.f * T[64 + 8 + ki] + T[64 + 16 + ki];
                }
                s_grad[idx] = lam * v;
            }
        }
        if (tid < C) {
            float s = 0.f;
            for (int w = 0; w < RS_NW; ++w) s += s_misc[w * MAX_C + tid];
            s_grad[n0 + n1 + n2 + n3 + tid] = a.accumulate_scaling_grad ? s_grad[n0 + n1 + n2 + n3 + tid] + s : s;
        }

        RS_STAMP(9);
        // ---- equaliser backward: X^T diag(g) X over the 32-sample window on the matrix cores, symbols as the contraction
        // index; the first-order gradient is the row sum of the same A operand
        {
            rs_f16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
            float asum = 0.f;
            const int i = lane & 31, half = lane >> 5;
            const int npair = (B + 1) / 2;
            const float* yw = s_y + RS_HL + o.c0 - i;
            for (int pq = wv; pq < npair; pq += 4 * RS_NW) {
                float xv[4], av[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    // pairs past the end read the zero padding of both arrays (up to 8 RS_NW symbols beyond the batch)
                    const int b = 2 * (pq + u * RS_NW) + half;
                    xv[u] = yw[b * S];
                    av[u] = s_gx[b] * xv[u];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    asum += av[u];
                    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], xv[u], acc, 0, 0, 0);
                }
            }
            // the ELBO stage is over: its per-window image is free for the cross-wave sums
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = (q & 3) + 8 * (q >> 2) + 4 * half;
                s_D[wv * 1024 + row * 32 + i] = acc[q];
            }
            asum = rs_add_xor<32>(asum);
            if (half == 0) s_D[RS_NW * 1024 + wv * 32 + i] = asum;
        }
        __syncthreads();
        {
            if (tid < l1) {
                float sacc = 0.f;
                for (int w = 0; w < RS_NW; ++w) sacc += s_D[RS_NW * 1024 + w * 32 + o.o1 - tid];
                s_grad[n0 + n1 + tid] = sacc;
            }
            for (int e = tid; e < n3; e += RS_NT) {
                const int ia = e / l2 + o.o2, ib = e % l2 + o.o2;
                float sacc = 0.f;
                for (int w = 0; w < RS_NW; ++w) sacc += s_D[w * 1024 + ia * 32 + ib] + s_D[w * 1024 + ib * 32 + ia];
                s_grad[n0 + n1 + n2 + e] = sacc;
            }
        }
        __syncthreads();

        RS_STAMP(10);
        // ---- Adam with AMSGrad (torch.optim.Adam single-tensor arithmetic), noise variance, loss
        {
#pragma unroll
            for (int q = 0; q < RS_APT; ++q) {
                const int idx = tid + q * RS_NT;
                if (idx < nadam && !(a.freeze_h2 && idx >= n0 && idx < n0 + n1)) {
                    const float gi = s_grad[idx];
                    const float mi = fmaf(w1, gi - am[q], am[q]);
                    const float vi = fmaf(w2 * gi, gi, av[q] * a.beta2);
                    const float vm = fmaxf(avm[q], vi);
                    const float denom = sqrtf(vm) / bc2_sqrt + a.eps;
                    am[q] = mi; av[q] = vi; avm[q] = vm;
                    pr[q] = pr[q] - step_size * (mi / denom);
                }
            }
            if (y_ahead && n + 1 < a.n_batches) {
#pragma unroll
                for (int q = 0; q < RS_YPT; ++q) {
                    const int i = tid + q * RS_NT;
                    if (i < N) s_y[RS_HL + i] = ynx[q];
                }
            }
            if (a.adaptive_noise) nv = (float)(E / (double)W);
            if (tid == 0 && a.loss_curve) a.loss_curve[(int64_t)model * a.n_batches + n] = (float)loss;
        }
        __syncthreads();
        RS_STAMP(11);
#ifdef V2VAE_STAMPS
        if (tid == 0 && model == 0 && n == a.n_batches - 1) reinterpret_cast<unsigned long long*>(a.ws)[13] = clock64();
#endif
    }

    // ---- state back to global memory
#pragma unroll
    for (int q = 0; q < RS_APT; ++q) {
        const int idx = tid + q * RS_NT;
        if (idx < pack) {
            a.params[gbase + idx] = pr[q];
            a.adam_m[gbase + idx] = am[q];
            a.adam_v[gbase + idx] = av[q];
            a.adam_vmax[gbase + idx] = avm[q];
            a.grads[gbase + idx] = s_grad[idx];
        }
    }
    if (tid == 0) {
        a.noise_var[model] = nv;
        double* sc = a.scal + (int64_t)model * V2VAE_SCAL_COUNT;
        sc[V2VAE_SCAL_E] = E; sc[V2VAE_SCAL_KL] = KL; sc[V2VAE_SCAL_LOSS] = loss; sc[V2VAE_SCAL_LAMBDA] = lamd;
        sc[V2VAE_SCAL_W] = (double)W;
        sc[V2VAE_SCAL_BAD_STEPS] += (double)bad_steps;
    }
}

int fit_rs_applicable(const Lay& l, int64_t pack) {
    RsOff o;
    return rs_plan(l, pack, &o);
}

// Launches the on-chip fit kernel if the shape fits it; returns 1 if launched, 0 if not applicable, < 0 on error.
int launch_fit_rs(const Lay& l, const v2vae_fit_args* a, const int64_t n[5], hipStream_t st) {
    RsOff o;
    const int64_t pack = n[0] + n[1] + n[2] + n[3] + n[4];
    if (!rs_plan(l, pack, &o)) return 0;
    const int lds = o.total * 4;
#define V2VAE_LAUNCH_RS(CB)                                                                                             \
    do {                                                                                                                \
        if (lds > 64 * 1024 &&                                                                                          \
            hipFuncSetAttribute(reinterpret_cast<const void*>(k_fit_rs<CB>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                lds) != hipSuccess)                                                                     \
            return V2VAE_ELAUNCH;                                                                                       \
        hipLaunchKernelGGL(k_fit_rs<CB>, dim3(a->n_models), dim3(RS_NT), lds, st, l, o, *a, (int)n[0], (int)n[1],        \
                           (int)n[2], (int)n[3], (int)n[4]);                                                            \
    } while (0)
    if (l.C <= 2) V2VAE_LAUNCH_RS(2);
    else if (l.C <= 4) V2VAE_LAUNCH_RS(4);
    else if (l.C <= 8) V2VAE_LAUNCH_RS(8);
    else V2VAE_LAUNCH_RS(16);
#undef V2VAE_LAUNCH_RS
    return 1;
}

}  // namespace v2vae
