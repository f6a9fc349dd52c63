// Complex second-order Volterra equaliser on the f32 matrix cores: backward for kernels of up to 31 lags (the forward of every
// length and the backward of longer kernels live in v2vae_eq_cxb.hip).
//
// xhat_b = sum_j k1[j] y[bS + j - p1] + Y_b^T K Y_b,  K = k2 + k2^T complex symmetric, Y_b[x] = y[bS + p2 - x],
// Y = Ya + j Yb.
//
// Backward: with rho = g0 Ya + g1 Yb, sigma = g1 Ya - g0 Yb (g = dloss/dxhat) the gradient planes are
// dK_r = sum rho^T Ya + sigma^T Yb and dK_i = sum sigma^T Ya - rho^T Yb, both symmetric in the lag pair: the three
// upper-triangular 16x16 blocks per wave, as three real products each (see the kernel), 16x16x4 MFMAs, four symbols per k-step; a
// ones column at index l2 of the Ya operand (zero in the Yb operand) yields the first-order gradient in the same products.
//
// Applies when mode = complex, l2 odd <= 31, l1 odd <= l2 and pad1 = l1 / 2.
#include <hip/hip_runtime.h>

#include "v2vae_layout.h"

namespace v2vae {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

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
    const int i16 = lane & 15, kq = lane >> 4;
    // Both gradient planes are symmetric in the lag pair, so of the 2 x 2 blocks of 16 lags only (0,0), (0,1), (1,1) are
    // accumulated (16x16x4 MFMAs, four symbols per k-step); the chain rule reads block (1,0) from (0,1).
    // The complex accumulation dK_r + i dK_i = sum (rho + i sigma)^T (Ya - i Yb) is formed from THREE real products per block
    // instead of four:  K1 = (rho + sigma)^T Ya,  K2 = -rho^T (Ya + Yb),  K3 = sigma^T (Ya - Yb);  dK_r = K1 - K3, dK_i = K1 + K2.
    // The sums are linear, so K1, K2, K3 are accumulated over all symbols and combined once at the end: 9 MFMAs per k-step
    // instead of 12.  The ones column (first-order gradient) sits in Ya only; its Yb entry is zero in all three operands.
    f32x4 A00, A01, A11, B00, B01, B11, C00, C01, C11;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        A00[g] = 0.f; A01[g] = 0.f; A11[g] = 0.f; B00[g] = 0.f; B01[g] = 0.f; B11[g] = 0.f; C00[g] = 0.f; C01[g] = 0.f; C11[g] = 0.f;
    }
    const float* tr = tile;
    const float* ti = tile + l.tile_len;
    const int j0 = i16, j1 = 16 + i16;
    const bool ones0 = (j0 == l.l2), ones1 = (j1 == l.l2);
    const int jc0 = j0 <= l.l2 ? j0 : l.l2, jc1 = j1 <= l.l2 ? j1 : l.l2;  // rows / columns beyond l2 are never read back
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
        for (int s = 0; s < 16; ++s) {
            const int bloc = wbase + 4 * s + kq;
            const int base = bloc * l.S + l.hl + l.p2;
            const float xa0 = tr[base - jc0], xb0 = ti[base - jc0], xa1 = tr[base - jc1], xb1 = ti[base - jc1];
            const float g0 = gt[bloc], g1 = gt[EQ_TB + bloc];
            const float rho0 = g0 * xa0 + g1 * xb0, sig0 = g1 * xa0 - g0 * xb0;
            const float rho1 = g0 * xa1 + g1 * xb1, sig1 = g1 * xa1 - g0 * xb1;
            const float ca0 = ones0 ? 1.f : xa0, cb0 = ones0 ? 0.f : xb0;
            const float ca1 = ones1 ? 1.f : xa1, cb1 = ones1 ? 0.f : xb1;
            const float rs0 = rho0 + sig0, rs1 = rho1 + sig1;
            const float cs0 = ca0 + cb0, cs1 = ca1 + cb1, cd0 = ca0 - cb0, cd1 = ca1 - cb1;
            A00 = MFMA16(rs0, ca0, A00); B00 = MFMA16(-rho0, cs0, B00); C00 = MFMA16(sig0, cd0, C00);
            A01 = MFMA16(rs0, ca1, A01); B01 = MFMA16(-rho0, cs1, B01); C01 = MFMA16(sig0, cd1, C01);
            A11 = MFMA16(rs1, ca1, A11); B11 = MFMA16(-rho1, cs1, B11); C11 = MFMA16(sig1, cd1, C11);
        }
    }
    const f32x4 Rr00 = A00 - C00, Rr01 = A01 - C01, Rr11 = A11 - C11;
    const f32x4 Ri00 = A00 + B00, Ri01 = A01 + B01, Ri11 = A11 + B11;
    // fixed-order reduction of the four waves through LDS, then one record per block: [2 planes][32][32], block (1,0) left zero
    __syncthreads();
    float* img = sm;  // [2][1024]
    for (int e = tid; e < 2048; e += NT) img[e] = 0.f;
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int a = 4 * kq + g, b = i16;
                img[a * 32 + b] += Rr00[g];             img[1024 + a * 32 + b] += Ri00[g];
                img[a * 32 + 16 + b] += Rr01[g];        img[1024 + a * 32 + 16 + b] += Ri01[g];
                img[(16 + a) * 32 + 16 + b] += Rr11[g]; img[1024 + (16 + a) * 32 + 16 + b] += Ri11[g];
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
        return (a >= 16 && b < 16) ? T[p * 1024 + b * 32 + a] : T[p * 1024 + a * 32 + b];   // block (1,0) = (0,1) transposed
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
