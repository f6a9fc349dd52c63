// Complex second-order Volterra equaliser on the f32 matrix cores: backward for kernels of up to 79 lags (the forward of every
// length and the backward of still longer kernels live in v2vae_eq_cxb.hip).
//
// xhat_b = sum_j k1[j] y[bS + j - p1] + Y_b^T K Y_b,  K = k2 + k2^T complex symmetric, Y_b[x] = y[bS + p2 - x],
// Y = Ya + j Yb.
//
// Backward: with rho = g0 Ya + g1 Yb, sigma = g1 Ya - g0 Yb (g = dloss/dxhat) the gradient planes are
// dK_r = sum rho^T Ya + sigma^T Yb and dK_i = sum sigma^T Ya - rho^T Yb, both symmetric in the lag pair: the three
// upper-triangular 16x16 blocks per wave, as three real products each (see the kernel), 16x16x4 MFMAs, four symbols per k-step; a
// ones column at index l2 of the Ya operand (zero in the Yb operand) yields the first-order gradient in the same products.
//
// Applies when mode = complex, l2 odd <= 79, l1 odd <= l2 and pad1 = l1 / 2 (lag rows in 1 .. 5 blocks of 16).
#include <hip/hip_runtime.h>

#include "v2vae_chain_cx.h"
#include "v2vae_layout.h"

namespace v2vae {

typedef float f32x4 __attribute__((ext_vector_type(4)));
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

// Sample tile from global sample g0 on, the two streams INTERLEAVED ([tile_len][2]: one 8-byte LDS read gives both parts of
// a sample).  A tile that lies inside both the global sequence and the local sample block (all but the first and last few) is
// copied without per-element checks; indices stay 32-bit inside the tile.
// REAL: the real model (one stream): the quadrature part of every sample is zero.
template <bool REAL>
__device__ __forceinline__ void eqcx_load_tile(const Lay& l, const float* __restrict__ y, int64_t g0, float* tile) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const int tl = l.tile_len;
    const int64_t loc0 = g0 - l.y_lo;
    const bool inside = g0 >= 0 && g0 + tl <= l.n_glob && loc0 >= 0 && loc0 + tl <= l.y_len;   // uniform
    const float* __restrict__ sa = y + loc0;
    const float* __restrict__ sb = REAL ? sa : y + l.y_stride + loc0;
    f2* dst = reinterpret_cast<f2*>(tile);
    if (inside) {
        // four samples per thread requested before the first is written (one memory latency per tile, not one per sample)
        for (int i0 = threadIdx.x; i0 < tl; i0 += 4 * NT) {
            f2 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * NT;
                v[k] = i < tl ? f2{sa[i], REAL ? 0.f : sb[i]} : f2{0.f, 0.f};
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int i = i0 + k * NT;
                if (i < tl) dst[i] = v[k];
            }
        }
    } else {
        for (int i = threadIdx.x; i < tl; i += NT) {
            const int64_t g = g0 + i, loc = loc0 + i;
            const bool in = g >= 0 && g < l.n_glob && loc >= 0 && loc < l.y_len;
            dst[i] = in ? f2{sa[i], REAL ? 0.f : sb[i]} : f2{0.f, 0.f};
        }
    }
}

// NB16 = blocks of 16 lag rows (lags 0 .. l2, the ones column at index l2 included): 2 for up to 31 lags, ... 5 for up to 79.
// SS = samples per symbol when it is 1 or 2 (the sample offsets of the unrolled symbol loop are then instruction immediates),
// 0 = read from the layout.
// LDS: sample tile [tile_len][2] | per-symbol gradient combinations, 4-vectors [EQ_TB] and a plane [EQ_TB] | constant (1, 0) samples [64 S][2].
// REAL: the real model through the same kernel: Yb = 0 and g1 = 0, so rho = g0 Ya, sigma = 0 and the gradient plane dK = sum rho^T Ya
// is the first of the three products alone (one MFMA per block and k-step instead of three; the second plane of the record stays zero).
template <int NB16, int SS, bool REAL>
__global__ __launch_bounds__(NT, NB16 <= 3 ? 4 : 2) void k_eq_bwd_cx(Lay l, const float* __restrict__ y, const float* __restrict__ gx,
                                                                  float* __restrict__ ws) {
    constexpr int NBLK = NB16 * (NB16 + 1) / 2, LD = 16 * NB16;
    extern __shared__ __align__(16) float sm[];
    const int S = SS ? SS : l.S;
    float* tile = sm;
    const int ntl = (2 * l.tile_len + 3) & ~3;   // interleaved (Ya, Yb) pairs, also for the real model
    f32x4* g4 = reinterpret_cast<f32x4*>(sm + ntl);   // (g0 + g1, g1, g1 - g0, -g0) per symbol of the tile
    float* ng1 = sm + ntl + 4 * EQ_TB;                 // -g1
    const int o_ones = ntl + 5 * EQ_TB;   // (1, 0) pairs
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    // Both gradient planes are symmetric in the lag pair, so only the upper-triangular 16x16 blocks (ib <= jb) are
    // accumulated (16x16x4 MFMAs, four symbols per k-step); the chain rule reads a lower block from its transpose.
    // The complex accumulation dK_r + i dK_i = sum (rho + i sigma)^T (Ya - i Yb) is formed from THREE real products per block
    // instead of four:  K1 = (rho + sigma)^T Ya,  K2 = -rho^T (Ya + Yb),  K3 = sigma^T (Ya - Yb);  dK_r = K1 - K3, dK_i = K1 + K2.
    // The sums are linear, so K1, K2, K3 are accumulated over all symbols and combined once at the end.  The ones column
    // (first-order gradient) sits in Ya only; its Yb entry is zero in all three operands.
    // f32 MFMAs and vector instructions do not overlap on this part, so the operand preparation is kept to six vector
    // instructions per (k-step, block): the three row operands are linear in (Ya, Yb) with per-symbol coefficients that are formed
    // once per tile, and the lanes of the ones column (and beyond) read constant 1 / 0 samples from LDS instead of being patched.
    f32x4 A[NBLK], B[NBLK], C[NBLK];
#pragma unroll
    for (int q = 0; q < NBLK; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) { A[q][g] = 0.f; B[q][g] = 0.f; C[q][g] = 0.f; }
    for (int idx = tid; idx < 128 * S; idx += NT) sm[o_ones + idx] = (idx & 1) ? 0.f : 1.f;
    // per-lane sample positions of symbol (wave base + kq) for each block; rows / columns from l2 on read the constants (their rows
    // are never read back, column l2 is the ones column)
    typedef float f2 __attribute__((ext_vector_type(2)));
    const f2* pa[NB16];
    {
        const int base0 = (wv * 64 + kq) * S + l.hl + l.p2;
#pragma unroll
        for (int ib = 0; ib < NB16; ++ib) {
            const int j = 16 * ib + i16;
            pa[ib] = reinterpret_cast<const f2*>(sm + (j >= l.l2 ? o_ones : 2 * (base0 - j)));
        }
    }
    for (int t = blockIdx.x; t < l.ntile_eq; t += gridDim.x) {
        const int64_t b0 = l.own_lo + (int64_t)t * EQ_TB;
        const int64_t rem = l.own_lo + l.own_cnt - b0;
        const int nbs = rem < EQ_TB ? (int)rem : EQ_TB;
        __syncthreads();
        eqcx_load_tile<REAL>(l, y, b0 * l.S - l.hl, tile);
        {   // EQ_TB == NT: one symbol per thread
            const float g0 = tid < nbs ? gx[b0 + tid - l.sym_lo] : 0.f;
            const float g1 = (!REAL && tid < nbs) ? gx[l.sym_cnt + (b0 + tid - l.sym_lo)] : 0.f;
            g4[tid] = f32x4{g0 + g1, g1, g1 - g0, -g0};
            ng1[tid] = -g1;
        }
        __syncthreads();
        const int wbase = wv * 64;
#pragma unroll((NB16 <= 2 && SS > 0) ? 16 : 2)
        for (int s = 0; s < 16; ++s) {
            const f32x4 gg = g4[wbase + 4 * s + kq];
            const float n1 = ng1[wbase + 4 * s + kq];
            float rs[NB16], nrho[NB16], sig[NB16], ca[NB16], cs[NB16], cd[NB16];
#pragma unroll
            for (int ib = 0; ib < NB16; ++ib) {
                const f2 x = pa[ib][4 * s * S];
                const float xa = x.x, xb = x.y;
                ca[ib] = xa;
                if (REAL) { rs[ib] = gg.x * xa; continue; }
                const f2 t = f2{gg.x, gg.y} * xa + f2{gg.z, gg.w} * xb;   // (rho + sigma, sigma = g1 Ya - g0 Yb)
                rs[ib] = t.x;
                sig[ib] = t.y;
                nrho[ib] = fmaf(gg.w, xa, n1 * xb);  // -(g0 Ya + g1 Yb)
                cs[ib] = xa + xb;
                cd[ib] = xa - xb;
            }
#pragma unroll
            for (int ib = 0; ib < NB16; ++ib)
#pragma unroll
                for (int jb = ib; jb < NB16; ++jb) {
                    const int q = ib * NB16 - ib * (ib - 1) / 2 + (jb - ib);
                    A[q] = MFMA16(rs[ib], ca[jb], A[q]);
                    if (!REAL) {
                        B[q] = MFMA16(nrho[ib], cs[jb], B[q]);
                        C[q] = MFMA16(sig[ib], cd[jb], C[q]);
                    }
                }
        }
    }
    // fixed-order reduction of the four waves through LDS, then one record per block: [2 planes][LD][LD], lower blocks left zero
    __syncthreads();
    float* img = sm;  // [2][LD * LD]
    for (int e = tid; e < 2 * LD * LD; e += NT) img[e] = 0.f;
    __syncthreads();
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int ib = 0; ib < NB16; ++ib)
#pragma unroll
                for (int jb = ib; jb < NB16; ++jb) {
                    const int q = ib * NB16 - ib * (ib - 1) / 2 + (jb - ib);
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int e = (16 * ib + 4 * kq + g) * LD + 16 * jb + i16;
                        img[e] += REAL ? A[q][g] : A[q][g] - C[q][g];
                        if (!REAL) img[LD * LD + e] += A[q][g] + B[q][g];
                    }
                }
        }
        __syncthreads();
    }
    float* rec = ws + l.o_part_eqcx + (size_t)blockIdx.x * (2 * LD * LD);
    for (int e = tid; e < 2 * LD * LD; e += NT) rec[e] = img[e];
}

// chain rule of the equaliser gradients (body: v2vae_chain_cx.h)
__global__ __launch_bounds__(NT) void k_chain_equaliser_cx(Lay l, const float* __restrict__ ws, float* __restrict__ g_k1,
                                                           float* __restrict__ g_k2) {
    d_chain_equaliser_cx(l, ws, g_k1, g_k2, (int)blockIdx.x, (int)gridDim.x);
}

int launch_eq_bwd_cx(const Lay& l, const float* y, const float* gx, float* ws, hipStream_t st) {
    const int lds = l.lds_eqb16;
#define V2VAE_LAUNCH_EQB16_R(NB, SS, RR)                                                                                         \
    do {                                                                                                                         \
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(k_eq_bwd_cx<NB, SS, RR>),                        \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)               \
            return V2VAE_ELAUNCH;                                                                                                \
        hipLaunchKernelGGL((k_eq_bwd_cx<NB, SS, RR>), dim3(l.nblk_eq), dim3(NT), lds, st, l, y, gx, ws);                          \
    } while (0)
#define V2VAE_LAUNCH_EQB16_S(NB, SS)                     \
    do {                                                 \
        if (l.ns == 1) V2VAE_LAUNCH_EQB16_R(NB, SS, true); \
        else V2VAE_LAUNCH_EQB16_R(NB, SS, false);        \
    } while (0)
#define V2VAE_LAUNCH_EQB16(NB)                       \
    do {                                             \
        if (l.S == 1) V2VAE_LAUNCH_EQB16_S(NB, 1);   \
        else if (l.S == 2) V2VAE_LAUNCH_EQB16_S(NB, 2); \
        else V2VAE_LAUNCH_EQB16_S(NB, 0);            \
    } while (0)
    switch (l.eqb16) {
        case 1: V2VAE_LAUNCH_EQB16(1); break;
        case 2: V2VAE_LAUNCH_EQB16(2); break;
        case 3: V2VAE_LAUNCH_EQB16(3); break;
        case 4: V2VAE_LAUNCH_EQB16(4); break;
        case 5: V2VAE_LAUNCH_EQB16(5); break;
        default: return V2VAE_ESIZE;
    }
#undef V2VAE_LAUNCH_EQB16
#undef V2VAE_LAUNCH_EQB16_S
#undef V2VAE_LAUNCH_EQB16_R
    return V2VAE_OK;
}

void launch_chain_equaliser_cx(const Lay& l, float* ws, float* g_k1, float* g_k2, hipStream_t st) {
    const int n = l.ns * (l.l1 + l.l2 * l.l2);
    hipLaunchKernelGGL(k_chain_equaliser_cx, dim3((n + NT - 1) / NT), dim3(NT), 0, st, l, (const float*)ws, g_k1, g_k2);
}

}  // namespace v2vae
