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
constexpr int RS_NW = RS_NT / 64;   // waves
constexpr int RS_HL = 32;           // zero samples kept left of the batch in LDS
constexpr int RS_APT = 8;           // optimiser elements per thread (parameter pack <= RS_APT * RS_NT)
constexpr int RS_GREC = 96;         // per-wave, per-phase record: G 8x8 | f1 8 | f2 8 | f3 8 | pad 8

typedef float rs_f16 __attribute__((ext_vector_type(16)));

struct RsOff {  // LDS carve in floats
    int y, x, gx, mom, MP, gm, GP, D, vl, H, HH, h1c, d, dq, par, grad, gred, gtot, P, misc, dbl, total;
    int nchunks, TS, packp;
};

static inline int rs_rup(int x, int m) { return (x + m - 1) / m * m; }

// Fills `o` and returns 1 if the shape fits this kernel.
int rs_plan(const Lay& l, int64_t pack, RsOff* o) {
    if (getenv("V2VAE_FORCE_GENERIC")) return 0;
    if (l.mode != 0 || l.ns != 1 || l.LE > 8 || l.S > 4 || l.l2 > 31 || l.l1 > 64 || l.pad1 > RS_HL || l.C > MAX_C) return 0;
    if (pack > RS_APT * RS_NT || l.b_glob > 4096 || l.n_glob > 8192) return 0;
    const int B = (int)l.b_glob, N = (int)l.n_glob;
    RsOff r{};
    r.TS = RS_NT - l.kmax;
    r.nchunks = (B + r.TS - 1) / r.TS;
    r.packp = rs_rup((int)pack, 64);
    int off = 0;
    auto take = [&](int n) { int at = off; off += rs_rup(n, 4); return at; };
    r.y = take(RS_HL + N + 32 * l.S + 64);
    r.x = take(rs_rup(B, 64) + 64);
    r.gx = take(rs_rup(B, 64) + 64);
    r.MP = rs_rup(8 + (r.nchunks - 1) * r.TS + RS_NT + 8, 4);
    r.mom = take(4 * r.MP);
    r.GP = rs_rup(B, 4);
    r.gm = take(4 * r.GP);
    r.D = take(3 * 8 * RS_NT);  // also the cross-wave image of the equaliser-gradient tiles (8 x 1024 floats)
    r.vl = take(RS_NT);
    r.H = take(l.S * 64);
    r.HH = take(l.S * 64);
    r.h1c = take(l.S * 8);
    r.d = take(l.S * 8);
    r.dq = take(l.S * 8);
    r.par = take(r.packp);
    r.grad = take(r.packp);
    r.gred = take(RS_NW * l.S * RS_GREC);
    r.gtot = take(l.S * RS_GREC);
    r.P = take(l.S * 64);
    r.misc = take(RS_NT + 64 * RS_NW);
    r.dbl = take(2 * 16);
    r.total = off;
    if ((int64_t)r.total * 4 > LDS_BUDGET) return 0;
    *o = r;
    return 1;
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
            x[i] = keep + __shfl_xor(send, 1 << sft, 64);
        }
    }
    float v = x[0];
#pragma unroll
    for (int off = N; off < 64; off <<= 1) v += __shfl_xor(v, off, 64);
    return v;
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
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
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

template <int CB>
__global__ __launch_bounds__(RS_NT) void k_fit_rs(Lay l, RsOff o, v2vae_fit_args a, int n0, int n1, int n2, int n3, int n4) {
    extern __shared__ float sm[];
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
    float* s_D = sm + o.D;
    float* s_vl = sm + o.vl;
    float* s_H = sm + o.H;
    float* s_HH = sm + o.HH;
    float* s_h1c = sm + o.h1c;
    float* s_d = sm + o.d;
    float* s_dq = sm + o.dq;
    float* s_par = sm + o.par;
    float* s_grad = sm + o.grad;
    float* s_gred = sm + o.gred;
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
    const float w1 = (float)(1.0 - (double)a.beta1), w2 = (float)(1.0 - (double)a.beta2);
    const float* ym = a.y + (int64_t)model * a.y_model_stride;
    const float* levels = a.levels;
    const float* prior = a.prior;
    __syncthreads();

    for (int n = 0; n < a.n_batches; ++n) {
        const float* yb = ym + (int64_t)n * N;
        // ---- parameters and samples to LDS
#pragma unroll
        for (int q = 0; q < RS_APT; ++q) {
            const int idx = tid + q * RS_NT;
            if (idx < pack) s_par[idx] = pr[q];
        }
        for (int i = tid; i < N; i += RS_NT) s_y[RS_HL + i] = yb[i];
        __syncthreads();

        // ---- channel tables of every phase (compact active taps)
        for (int idx = tid; idx < S * 64; idx += RS_NT) {
            const int r = idx >> 6, k = (idx >> 3) & 7, j = idx & 7;
            const int ta = rs_tap(l, r, k), tb = rs_tap(l, r, j);
            const float hv = (ta >= 0 && tb >= 0) ? p_h2[ta * L + tb] + p_h2[tb * L + ta] : 0.f;
            s_H[idx] = hv;
            s_HH[idx] = hv * hv;
        }
        for (int idx = tid; idx < S * 8; idx += RS_NT) {
            const int r = idx >> 3, k = idx & 7;
            const int ta = rs_tap(l, r, k);
            const float dg = ta >= 0 ? p_h2[ta * L + ta] : 0.f;
            s_h1c[idx] = ta >= 0 ? p_h1[ta] : 0.f;
            s_d[idx] = 2.f * dg;
            s_dq[idx] = dg;  // lib/real_equalization.py:869 uses diag(h2), not diag(h2 + h2^T)
        }

        // ---- equaliser forward (lib/torch_components.py:79-96)
        {
            const int nks = (l2 + 1) / 2;
            const int jrow = lane & 31, half = lane >> 5;
            float Areg[16];
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int i = 2 * s + half;
                Areg[s] = (i < l2 && jrow < l2) ? p_k2[jrow * l2 + i] + p_k2[i * l2 + jrow] : 0.f;
            }
            const int ntile = (B + 31) / 32;
            for (int tile = wv; tile < ntile; tile += RS_NW) {
                const int b = tile * 32 + jrow;
                const int base = RS_HL + b * S + l.p2;
                float part = 0.f;
                if (l2 > 0) {
                    rs_f16 acc;
#pragma unroll
                    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
                    for (int s = 0; s < 16; ++s)
                        if (s < nks) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Areg[s], s_y[base - (2 * s + half)], acc, 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int j = (q & 3) + 8 * (q >> 2) + 4 * half;
                        part = fmaf(acc[q], s_y[base - j], part);
                    }
                }
                const int b1 = RS_HL + b * S - l.pad1;
                for (int j = half; j < l1; j += 2) part = fmaf(p_k1[j], s_y[b1 + j], part);
                part += __shfl_xor(part, 32, 64);
                if (half == 0 && b < B) s_x[b] = part;
            }
        }
        __syncthreads();

        // ---- soft demapper, moments, KL (lib/real_equalization.py:816-823, 851-866, 915-918)
        {
            double kl = 0.0;
            for (int b = tid; b < B; b += RS_NT) {
                const float x = s_x[b];
                Posterior p;
                posterior<CB>(x, levels, p_sc, C, nv, p);
                float* mo = s_mom + 8 + b;
                mo[0] = p.mu; mo[MP] = p.var; mo[2 * MP] = p.kap; mo[3 * MP] = p.gam;
                float klb = 0.f;
#pragma unroll
                for (int k = 0; k < CB; ++k)
                    if (k < C) klb += p.q[k] * logf(p.q[k] / prior[k] + 1e-12f);
                if (b >= l.kl_lo && b < l.kl_hi) kl += (double)klb;
                if (a.y_eq) a.y_eq[(int64_t)model * a.n_batches * B + (int64_t)n * B + b] = x;
            }
            KL = rs_block_sum(kl, s_dbl, tid);  // its barriers also publish the moments
        }

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
            double tot = 0.0;
            for (int w = 0; w < RS_NW; ++w) tot += (double)s_misc[RS_NT + w * 8 + d];
            const float* vb = s_mom + MP + 8;
            double edge = 0.0;
            for (int m = 0; m < mn; ++m) edge += (double)(vb[m] * vb[m + d]);
            for (int m = Tr + mn; m < B; ++m) edge += (double)(vb[m] * vb[m + d]);
            s_P[idx] = (float)(tot - edge);
        }

        // ---- ELBO expectations, one window per lane (lib/real_equalization.py:868-913 in window form)
        double Eacc = 0.0;
        for (int r = 0; r < S; ++r) {
            float G[64], F[32];
#pragma unroll
            for (int i = 0; i < 64; ++i) G[i] = 0.f;
#pragma unroll
            for (int i = 0; i < 32; ++i) F[i] = 0.f;
            const float4* Hq = reinterpret_cast<const float4*>(s_H + r * 64);
            const float4* HHq = reinterpret_cast<const float4*>(s_HH + r * 64);
            for (int c = 0; c < o.nchunks; ++c) {
                const int s0 = c * o.TS;
                const int s1 = s0 + o.TS < B ? s0 + o.TS : B;
                const int t = s0 - kmax + tid;
                const int nn = t * S + r;
                const bool valid = (t < s1) && nn >= 0 && nn < W;
                const bool own = valid && t >= s0;
                const float vl = valid ? 1.f : 0.f;
                const float yv = valid ? s_y[RS_HL + l.pre + nn] : 0.f;
                const float* mb = s_mom + 8 + t;
                float mu[8], vv[8], kp[8], gm[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) { mu[k] = mb[k]; vv[k] = mb[MP + k]; kp[k] = mb[2 * MP + k]; gm[k] = mb[3 * MP + k]; }
                float z[8], tt[8], wq[8];
                float pa = 0.f, pq = 0.f, ps = 0.f, psq = 0.f, pvs = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float4 h0 = Hq[2 * k], h1v = Hq[2 * k + 1], g0 = HHq[2 * k], g1 = HHq[2 * k + 1];
                    float u = h0.x * mu[0];
                    u = fmaf(h0.y, mu[1], u); u = fmaf(h0.z, mu[2], u); u = fmaf(h0.w, mu[3], u);
                    u = fmaf(h1v.x, mu[4], u); u = fmaf(h1v.y, mu[5], u); u = fmaf(h1v.z, mu[6], u); u = fmaf(h1v.w, mu[7], u);
                    float tk = g0.x * vv[0];
                    tk = fmaf(g0.y, vv[1], tk); tk = fmaf(g0.z, vv[2], tk); tk = fmaf(g0.w, vv[3], tk);
                    tk = fmaf(g1.x, vv[4], tk); tk = fmaf(g1.y, vv[5], tk); tk = fmaf(g1.z, vv[6], tk); tk = fmaf(g1.w, vv[7], tk);
                    const float h1k = s_h1c[r * 8 + k], dk = s_d[r * 8 + k], dqk = s_dq[r * 8 + k];
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
                const float wo = own ? 1.f : 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float4 h0 = Hq[2 * k], h1v = Hq[2 * k + 1];
                    float hw = h0.x * wq[0];
                    hw = fmaf(h0.y, wq[1], hw); hw = fmaf(h0.z, wq[2], hw); hw = fmaf(h0.w, wq[3], hw);
                    hw = fmaf(h1v.x, wq[4], hw); hw = fmaf(h1v.y, wq[5], hw); hw = fmaf(h1v.z, wq[6], hw); hw = fmaf(h1v.w, wq[7], hw);
                    const float dk = s_d[r * 8 + k], dqk = s_dq[r * 8 + k];
                    s_D[(0 * 8 + k) * RS_NT + tid] = vl * (2.f * e * z[k] + 2.f * hw);
                    s_D[(1 * 8 + k) * RS_NT + tid] = vl * (2.f * A * dk - 2.f * yv * dqk + z[k] * z[k] + 4.f * tt[k]);
                    s_D[(2 * 8 + k) * RS_NT + tid] = vl * (2.f * dk * z[k]);
                    // parameter-gradient integrands of owned windows
                    const float ak = wo * fmaf(e, mu[k], wq[k]);
#pragma unroll
                    for (int j = 0; j < 8; ++j) G[k * 8 + j] = fmaf(ak, mu[j], G[k * 8 + j]);
                    F[k] += wo * (2.f * e * mu[k] + wq[k]);
                    F[8 + k] += wo * (2.f * A * vv[k] + 2.f * kp[k] * z[k] + 2.f * gm[k] * dk);
                    F[16 + k] += wo * (-2.f * yv * vv[k]);
                }
                s_vl[tid] = vl;
                __syncthreads();
                // gather: symbol m = s0 + tid collects window t = m - kk at compact tap kk
                if (tid < o.TS && s0 + tid < s1) {
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                    for (int kk = 0; kk < LE; ++kk) {
                        const int slot = tid + kmax - kk;
                        a0 += s_D[(0 * 8 + kk) * RS_NT + slot];
                        a1 += s_D[(1 * 8 + kk) * RS_NT + slot];
                        a2 += s_D[(2 * 8 + kk) * RS_NT + slot];
                        const float dk = s_d[r * 8 + kk];
                        a3 = fmaf(s_vl[slot], dk * dk, a3);
                    }
                    float* gp = s_gm + s0 + tid;
                    if (r == 0) { gp[0] = a0; gp[GP] = a1; gp[2 * GP] = a2; gp[3 * GP] = a3; }
                    else { gp[0] += a0; gp[GP] += a1; gp[2 * GP] += a2; gp[3 * GP] += a3; }
                }
                __syncthreads();
            }
            const float gt = rs_wave_reduce_transposed<64>(G, lane);
            s_gred[(wv * S + r) * RS_GREC + rs_transposed_index<64>(lane)] = gt;
            const float ft = rs_wave_reduce_transposed<32>(F, lane);
            if (lane < 32) s_gred[(wv * S + r) * RS_GREC + 64 + rs_transposed_index<32>(lane)] = ft;
        }
        E = rs_block_sum(Eacc, s_dbl, tid);
        loss = KL + (double)W * log(E);
        lamd = (double)W / E;
        const float lam = (float)lamd;

        // ---- totals of the per-wave records (the barriers of the block sum published them)
        for (int idx = tid; idx < S * RS_GREC; idx += RS_NT) {
            const int r = idx / RS_GREC, e = idx - r * RS_GREC;
            double s = 0.0;
            for (int w = 0; w < RS_NW; ++w) s += (double)s_gred[(w * S + r) * RS_GREC + e];
            s_gtot[idx] = (float)s;
        }

        // ---- per-symbol backward: moments -> posterior -> equaliser output, noise scaling (softmax backward in double)
        {
            float gs[MAX_C];
#pragma unroll
            for (int k = 0; k < CB; ++k) gs[k] = 0.f;
            for (int b = tid; b < B; b += RS_NT) {
                const float x = s_x[b];
                Posterior p;
                posterior<CB>(x, levels, p_sc, C, nv, p);
                const float* g = s_gm + b;
                const float gmu = g[0], gv = g[GP], gk = g[2 * GP], gg = g[3 * GP];
                const bool in_kl = b >= l.kl_lo && b < l.kl_hi;
                double dq[MAX_C];
                double qdq = 0.0;
#pragma unroll
                for (int k = 0; k < CB; ++k)
                    if (k < C) {
                        const double d = (double)levels[k] - (double)p.mu, d2 = d * d;
                        double v = (double)gmu * d + (double)gv * d2 + (double)gk * (d2 * d - 3.0 * (double)p.var * d) +
                                   (double)gg * (d2 * d2 - 4.0 * (double)p.kap * d - 6.0 * (double)p.var * d2);
                        v *= (double)lam;
                        if (in_kl) {
                            const float prk = prior[k];
                            v += (double)(logf(p.q[k] / prk + 1e-12f) + p.q[k] / (p.q[k] + 1e-12f * prk));
                        }
                        dq[k] = v;
                        qdq += (double)p.q[k] * v;
                    }
                double gxd = 0.0;
#pragma unroll
                for (int k = 0; k < CB; ++k)
                    if (k < C) {
                        const double dz = (double)p.q[k] * (dq[k] - qdq);
                        const double df = (double)x - (double)levels[k];
                        const double den = (double)nv * (double)p_sc[k];
                        gxd += dz * (-2.0 * df / den);
                        gs[k] += (float)(dz * df * df / (den * (double)p_sc[k]));
                    }
                s_gx[b] = (float)gxd;
            }
#pragma unroll
            for (int k = 0; k < CB; ++k) {
                float v = gs[k];
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
                if (lane == 0 && k < C) s_misc[wv * MAX_C + k] = v;
            }
        }
        __syncthreads();

        // ---- channel gradients: chain from compact per-phase totals to h1, h2
        for (int idx = tid; idx < n0 + n1; idx += RS_NT) {
            auto compact = [&](int t, int& r, int& k) {
                int rr = (t - (L - 1)) % S;
                if (rr < 0) rr += S;
                r = rr;
                k = (rr + L - 1 - t) / S;
            };
            if (idx < n0) {
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
            s_grad[n0 + n1 + n2 + n3 + tid] = s;
        }

        // ---- equaliser backward: first-order kernel, then X^T diag(g) X on the matrix cores
        {
            const int ngrp = RS_NT / l1;
            const int j = tid % l1, grp = tid / l1;
            float acc = 0.f;
            if (grp < ngrp)
                for (int b = grp; b < B; b += ngrp) acc = fmaf(s_gx[b], s_y[RS_HL + b * S - l.pad1 + j], acc);
            // the ELBO stage is over: its per-window image is free for the partial sums
            if (grp < ngrp) s_D[RS_NW * 1024 + grp * l1 + j] = acc;
        }
        if (l2 > 0) {
            rs_f16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
            const int i = lane & 31, half = lane >> 5;
            const int npair = (B + 1) / 2;
            for (int pq = wv; pq < npair; pq += RS_NW) {
                const int b = 2 * pq + half;
                const float xv = s_y[RS_HL + b * S + l.p2 - i];
                const float gv = s_gx[b];  // zero beyond the batch
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(gv * xv, xv, acc, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const int row = (q & 3) + 8 * (q >> 2) + 4 * half;
                s_D[wv * 1024 + row * 32 + i] = acc[q];
            }
        }
        __syncthreads();
        {
            const int ngrp = RS_NT / l1;
            if (tid < l1) {
                float s = 0.f;
                for (int g = 0; g < ngrp; ++g) s += s_D[RS_NW * 1024 + g * l1 + tid];
                s_grad[n0 + n1 + tid] = s;
            }
            for (int e = tid; e < n3; e += RS_NT) {
                const int i = e / l2, j = e - i * l2;
                float s = 0.f;
                for (int w = 0; w < RS_NW; ++w) s += s_D[w * 1024 + i * 32 + j] + s_D[w * 1024 + j * 32 + i];
                s_grad[n0 + n1 + n2 + e] = s;
            }
        }
        __syncthreads();

        // ---- Adam with AMSGrad (torch.optim.Adam single-tensor arithmetic), noise variance, loss
        {
            const float step_size = a.step_size[n], bc2_sqrt = a.bc2_sqrt[n];
#pragma unroll
            for (int q = 0; q < RS_APT; ++q) {
                const int idx = tid + q * RS_NT;
                if (idx < nadam) {
                    const float gi = s_grad[idx];
                    const float mi = fmaf(w1, gi - am[q], am[q]);
                    const float vi = fmaf(w2 * gi, gi, av[q] * a.beta2);
                    const float vm = fmaxf(avm[q], vi);
                    const float denom = sqrtf(vm) / bc2_sqrt + a.eps;
                    am[q] = mi; av[q] = vi; avm[q] = vm;
                    pr[q] = pr[q] - step_size * (mi / denom);
                }
            }
            if (a.adaptive_noise) nv = (float)(E / (double)W);
            if (tid == 0 && a.loss_curve) a.loss_curve[(int64_t)model * a.n_batches + n] = (float)loss;
        }
        __syncthreads();
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
    }
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
