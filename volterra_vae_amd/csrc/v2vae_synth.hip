// Device-side synthetic nonlinear-ISI channel: the received sequence of a benchmark / test run is produced in HBM instead of
// being simulated on the host and shipped over PCIe.  Same chain as the host generator of this package (and, in structure, the
// reference's simulator, lib/data_generation.py:206-241 and :349-372): uniform PAM / square-QAM symbols, zero-stuffed to sps
// samples per symbol, a low-pass pulse, a Wiener-Hammerstein system (FIR -> 2nd-order polynomial -> FIR, every stage a 'same'
// convolution with zeros outside the global sequence), normalisation, additive white Gaussian noise.
//
// Every random number is a pure function of (seed, kind, global index), so any rank can produce any sample range of one
// global sequence and overlapping ranges agree bit for bit.  One workgroup produces SY_TB symbols: symbols, the pulse-shaped
// signal and the polynomial output are staged through LDS tiles with halos, so each hash / FIR is evaluated once per tile.
#include <hip/hip_runtime.h>

#include "../../include/v2vae_hip.h"

namespace v2vae {

constexpr int SY_NT = 256;
constexpr int SY_TB = 512;  // symbols per workgroup

__host__ __device__ inline uint64_t sy_mix(uint64_t z) {  // splitmix64 finaliser
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// 64 random bits for (seed, kind, index); kind 0 = symbols, 1 = noise
__host__ __device__ inline uint64_t sy_bits(uint64_t seed, uint64_t kind, uint64_t idx) {
    return sy_mix(sy_mix(seed ^ (kind << 56)) ^ idx);
}

__global__ __launch_bounds__(SY_NT) void k_synth_channel(v2vae_synth_args a, float* __restrict__ y, int32_t* __restrict__ sym_idx) {
    extern __shared__ float sm[];
    const int S = a.sps, ns = a.ns, np_ = a.n_pulse, n1 = a.n_fir1, n2 = a.n_fir2;
    const int hp = (np_ - 1) / 2, h1 = (n1 - 1) / 2, h2 = (n2 - 1) / 2;
    const int64_t N = a.b_glob * S;
    const int64_t m0 = a.sym_lo + (int64_t)blockIdx.x * SY_TB;          // first symbol of this tile
    int64_t m1 = m0 + SY_TB;
    if (m1 > a.sym_lo + a.sym_cnt) m1 = a.sym_lo + a.sym_cnt;
    const int64_t s0 = m0 * S, s1 = m1 * S;                             // output samples [s0, s1)
    // w (polynomial output) is needed on [s0 - h2, s1 + h2), x on [s0 - h2 - h1, s1 + h2 + h1), symbols such that every
    // pulse tap of those x samples is covered
    const int64_t w_lo = s0 - h2, x_lo = w_lo - h1;
    const int w_len = (int)(s1 - s0) + 2 * h2, x_len = w_len + 2 * h1;
    const int64_t b_lo_raw = (x_lo - hp) / S - 1;                       // generous lower symbol bound
    const int b_len = (x_len + 2 * hp) / S + 3;
    float* s_sym = sm;                       // [ns][b_len] amplitudes (0 outside the global sequence)
    float* s_x = s_sym + ns * b_len;         // [ns][x_len]
    float* s_w = s_x + ns * x_len;           // [ns][w_len]
    const int tid = threadIdx.x;
    for (int i = tid; i < b_len; i += SY_NT) {
        const int64_t m = b_lo_raw + i;
        float va = 0.f, vb = 0.f;
        if (m >= 0 && m < a.b_glob) {
            const uint64_t z = sy_bits(a.seed, 0, (uint64_t)m);
            const int ia = (int)(((z >> 32) * (uint64_t)a.C) >> 32);
            const int ib = (int)(((z & 0xFFFFFFFFull) * (uint64_t)a.C) >> 32);
            va = a.levels[ia];
            vb = a.levels[ib];
            if (sym_idx && m >= m0 && m < m1) {
                sym_idx[m - a.sym_lo] = ia;
                if (ns == 2) sym_idx[a.sym_cnt + (m - a.sym_lo)] = ib;
            }
        }
        s_sym[i] = va;
        if (ns == 2) s_sym[b_len + i] = vb;
    }
    __syncthreads();
    // pulse shaping of the zero-stuffed symbols: x[n] = sum_m pulse[n - m S + hp] a_m
    for (int i = tid; i < x_len; i += SY_NT) {
        const int64_t n = x_lo + i;
        float xa = 0.f, xb = 0.f;
        if (n >= 0 && n < N) {
            // symbols m with |n - m S| <= hp
            int64_t mlo = (n - hp + S - 1) / S;
            if (n - hp < 0) mlo = -((hp - n) / S);
            const int64_t mhi = (n + hp) / S;
            for (int64_t m = mlo; m <= mhi; ++m) {
                const float p = a.pulse[(int)(n - m * S) + hp];
                const int j = (int)(m - b_lo_raw);
                xa = fmaf(p, s_sym[j], xa);
                if (ns == 2) xb = fmaf(p, s_sym[b_len + j], xb);
            }
        }
        s_x[i] = xa;
        if (ns == 2) s_x[x_len + i] = xb;
    }
    __syncthreads();
    // first FIR + polynomial (complex arithmetic for two streams)
    for (int i = tid; i < w_len; i += SY_NT) {
        const int64_t n = w_lo + i;
        float wa = 0.f, wb = 0.f;
        if (n >= 0 && n < N) {
            float za = 0.f, zb = 0.f;
            for (int k = 0; k < n1; ++k) {  // 'same': z[n] = sum_k fir1[k] x[n + h1 - k]
                const int j = i + 2 * h1 - k;   // index of sample n + h1 - k in s_x (x_lo = w_lo - h1)
                za = fmaf(a.fir1[k], s_x[j], za);
                if (ns == 2) zb = fmaf(a.fir1[k], s_x[x_len + j], zb);
            }
            if (ns == 2) {
                wa = a.poly1 * za + a.poly2 * (za * za - zb * zb);
                wb = a.poly1 * zb + a.poly2 * (2.f * za * zb);
            } else {
                wa = a.poly1 * za + a.poly2 * za * za;
            }
        }
        s_w[i] = wa;
        if (ns == 2) s_w[w_len + i] = wb;
    }
    __syncthreads();
    // second FIR, normalisation, noise
    const int out_len = (int)(s1 - s0);
    const int64_t ylen = a.sym_cnt * S;
    for (int i = tid; i < out_len; i += SY_NT) {
        const int64_t n = s0 + i;
        float oa = 0.f, ob = 0.f;
        for (int k = 0; k < n2; ++k) {
            const int j = i + 2 * h2 - k;
            oa = fmaf(a.fir2[k], s_w[j], oa);
            if (ns == 2) ob = fmaf(a.fir2[k], s_w[w_len + j], ob);
        }
        oa = (oa - a.mean[0]) * a.inv_std;
        ob = (ob - a.mean[1]) * a.inv_std;
        if (a.noise_std > 0.f) {
            const uint64_t z = sy_bits(a.seed, 1, (uint64_t)n);
            const float u1 = ((float)(z >> 40) + 0.5f) * (1.f / 16777216.f);
            const float u2 = ((float)((z >> 16) & 0xFFFFFFull) + 0.5f) * (1.f / 16777216.f);
            const float rad = sqrtf(-2.f * logf(u1)) * a.noise_std;
            float sn, cs;
            sincosf(6.28318530717958647692f * u2, &sn, &cs);
            oa = fmaf(rad, cs, oa);
            ob = fmaf(rad, sn, ob);
        }
        y[n - a.sym_lo * S] = oa;
        if (ns == 2) y[ylen + (n - a.sym_lo * S)] = ob;
    }
}

}  // namespace v2vae

using namespace v2vae;

extern "C" int v2vae_synth_channel(const v2vae_synth_args* a, float* y, int32_t* sym_idx, void* stream) {
    if (!a || !y) return V2VAE_EINVAL;
    if (a->ns < 1 || a->ns > 2 || a->sps < 1 || a->sps > 8 || a->C < 1 || a->C > 16) return V2VAE_EINVAL;
    if (a->n_pulse < 1 || a->n_pulse > 65 || !(a->n_pulse & 1) || a->n_fir1 < 1 || a->n_fir1 > 9 || !(a->n_fir1 & 1) ||
        a->n_fir2 < 1 || a->n_fir2 > 9 || !(a->n_fir2 & 1))
        return V2VAE_EINVAL;
    if (a->b_glob < 1 || a->sym_lo < 0 || a->sym_cnt < 1 || a->sym_lo + a->sym_cnt > a->b_glob) return V2VAE_EINVAL;
    const int S = a->sps, h1 = (a->n_fir1 - 1) / 2, h2 = (a->n_fir2 - 1) / 2, hp = (a->n_pulse - 1) / 2;
    const int w_len = SY_TB * S + 2 * h2, x_len = w_len + 2 * h1, b_len = (x_len + 2 * hp) / S + 3;
    const int lds = a->ns * (b_len + x_len + w_len) * 4;
    if (lds > 64 * 1024) return V2VAE_ESIZE;
    const int64_t nblk = (a->sym_cnt + SY_TB - 1) / SY_TB;
    if (nblk > 0x7FFFFFFF) return V2VAE_ESIZE;
    hipLaunchKernelGGL(k_synth_channel, dim3((unsigned)nblk), dim3(SY_NT), lds, (hipStream_t)stream, *a, y, sym_idx);
    return hipGetLastError() == hipSuccess ? V2VAE_OK : V2VAE_ELAUNCH;
}

extern "C" uint64_t v2vae_synth_bits(uint64_t seed, uint64_t kind, uint64_t idx) { return sy_bits(seed, kind, idx); }
