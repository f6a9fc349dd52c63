// Complex second-order Volterra equaliser on the f32 matrix cores: backward for kernels of up to 31 lags (the forward of every
// length and the backward of longer kernels live in v2vae_eq_cxb.hip).
//
// xhat_b = sum_j k1[j] y[bS + j - p1] + Y_b^T K Y_b,  K = k2 + k2^T complex symmetric, Y_b[x] = y[bS + p2 - x],
// Y = Ya + j Yb.
//
// Backward: with rho = g0 Ya + g1 Yb, sigma = g1 Ya - g0 Yb (g = dloss/dxhat) the gradient planes are
// dK_r = sum rho^T Ya + sigma^T Yb and dK_i = sum sigma^T Ya - rho^T Yb: two 32x32 accumulators per wave, two MFMAs into
// each per k-step, two symbols per k-step; a ones column at index l2 of the Ya operand (zero in the Yb operand) yields
// the first-order gradient in the same products.
//
// Applies when mode = complex, l2 odd <= 31, l1 odd <= l2 and pad1 = l1 / 2.
#include <hip/hip_runtime.h>

#include "v2vae_layout.h"

namespace v2vae {

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void eqcx_load_tile(const Lay& l, const float* __restrict__ y, int64_t g0, float* tile) {
    const int n = l.ns * l.tile_len;
    for (int idx = threadIdx.x; idx < n; idx += NT) {
        const int c = idx / l.tile_len, i = idx - c * l.tile_len;
        const int64_t g = g0 + i;
        float v = 0.f;
        if (g >= 0 && g < l.n_glob) {
            const int64_t loc = g - l.y_lo;
            if (loc >= 0 && loc < l.y_len) v = y[(int64_t)c * l.y_stride + loc];
        }
        tile[idx] = v;
    }
}

__global__ __launch_bounds__(NT, 4) void k_eq_bwd_cx(Lay l, const float* __restrict__ y, const float* __restrict__ gx,
                                                  float* __restrict__ ws) {
    extern __shared__ float sm[];
    float* tile = sm;
    float* gt = tile + l.ns * l.tile_len;  // [2][EQ_TB]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    f32x16 Rr, Ri;
#pragma unroll
    for (int g = 0; g < 16; ++g) { Rr[g] = 0.f; Ri[g] = 0.f; }
    const float* tr = tile;
    const float* ti = tile + l.tile_len;
    const bool ones_col = (j == l.l2);
    const int jc = j <= l.l2 ? j : l.l2;  // rows / columns beyond l2 are never read back; keep their loads in range
    for (int t = blockIdx.x; t < l.ntile_eq; t += gridDim.x) {
        const int64_t b0 = l.own_lo + (int64_t)t * EQ_TB;
        const int64_t rem = l.own_lo + l.own_cnt - b0;
        const int nbs = rem < EQ_TB ? (int)rem : EQ_TB;
        __syncthreads();
        eqcx_load_tile(l, y, b0 * l.S - l.hl, tile);
        for (int idx = tid; idx < 2 * EQ_TB; idx += NT) {
            const int o = idx / EQ_TB, i = idx - o * EQ_TB;
            gt[idx] = i < nbs ? gx[(int64_t)o * l.sym_cnt + (b0 + i - l.sym_lo)] : 0.f;
        }
        __syncthreads();
        const int wbase = wv * 64;
#pragma unroll 4
        for (int s = 0; s < 32; ++s) {
            const int bloc = wbase + 2 * s + h;
            const int base = bloc * l.S + l.hl + l.p2 - jc;
            const float xa = tr[base], xb = ti[base];
            const float g0 = gt[bloc], g1 = gt[EQ_TB + bloc];
            const float rho = g0 * xa + g1 * xb, sig = g1 * xa - g0 * xb;
            const float ca = ones_col ? 1.f : xa, cb = ones_col ? 0.f : xb;
            Rr = MFMA32(rho, ca, Rr);
            Ri = MFMA32(sig, ca, Ri);
            Rr = MFMA32(sig, cb, Rr);
            Ri = MFMA32(-rho, cb, Ri);
        }
    }
    // fixed-order reduction of the four waves through LDS, then one record per block
    __syncthreads();
    float* img = sm;  // [2][1024]
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int e = ((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + j;
                if (w == 0) { img[e] = Rr[g]; img[1024 + e] = Ri[g]; }
                else { img[e] += Rr[g]; img[1024 + e] += Ri[g]; }
            }
        }
        __syncthreads();
    }
    float* rec = ws + l.o_part_eqcx + (size_t)blockIdx.x * 2048;
    for (int e = tid; e < 2048; e += NT) rec[e] = img[e];
}

__global__ __launch_bounds__(NT) void k_chain_equaliser_cx(Lay l, const float* __restrict__ ws, float* __restrict__ g_k1,
                                                           float* __restrict__ g_k2) {
    const float* T = ws + l.o_tot_eqcx;
    const int l1 = l.l1, l2 = l.l2;
    const int n1 = 2 * l1, n2 = 2 * l2 * l2;
    auto GK = [&](int p, int a, int b) {  // dE/dK (effective symmetric kernel), plane p
        return T[p * 1024 + a * 32 + b];
    };
    for (int idx = blockIdx.x * NT + threadIdx.x; idx < n1 + n2; idx += gridDim.x * NT) {
        if (idx < n1) {
            if (!g_k1) continue;
            const int p = idx / l1, jj = idx - p * l1;
            const int x = l.pad1 + l.p2 - jj;  // lag row that first-order tap jj multiplies
            g_k1[idx] = T[p * 1024 + x * 32 + l2];
        } else {
            if (!g_k2) continue;
            const int e = idx - n1;
            const int p = e / (l2 * l2), rem = e - p * l2 * l2;
            const int a = rem / l2, b = rem - a * l2;
            g_k2[e] = GK(p, a, b) + GK(p, b, a);
        }
    }
}

int launch_eq_bwd_cx(const Lay& l, const float* y, const float* gx, float* ws, hipStream_t st) {
    hipLaunchKernelGGL(k_eq_bwd_cx, dim3(l.nblk_eq), dim3(NT), l.lds_eq, st, l, y, gx, ws);
    return V2VAE_OK;
}

void launch_chain_equaliser_cx(const Lay& l, float* ws, float* g_k1, float* g_k2, hipStream_t st) {
    const int n = 2 * l.l1 + 2 * l.l2 * l.l2;
    hipLaunchKernelGGL(k_chain_equaliser_cx, dim3((n + NT - 1) / NT), dim3(NT), 0, st, l, (const float*)ws, g_k1, g_k2);
}

}  // namespace v2vae
