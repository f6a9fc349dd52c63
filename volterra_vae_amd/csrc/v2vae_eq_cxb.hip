// Complex second-order Volterra equaliser.  Two forward kernels and the backward for the longest kernels:
//  * k_eq_fwd_sym<T>: the forward in symmetric form on the vector pipe, up to 95 lags (the default; described at the kernel);
//  * k_eq_fwd_cxb<NBL> / k_eq_bwd_cxb<NBL>: the f32 matrix cores with the lag rows in NBL blocks of 32: forward on request
//    (V2VAE_FLAG_EQ_MFMA_FWD, the A/B switch of the symmetric form), backward for 81 .. 95 lags
//    (v2vae_eq_cx.hip has the backward on 16-lag blocks up to 79 lags).
//
// xhat_b = sum_j k1[j] y[bS + j - p1] + Y_b^T K Y_b,  K = k2 + k2^T complex symmetric, Y_b[x] = y[bS + p2 - x]
// (lib/torch_components.py:79-96 on complex samples).
//
// Forward: [Zr | Zi] = [Ya | Yb] . [[Kr, Ki], [-Ki, Kr]] with symbols on the lanes and lag rows in the accumulator
// registers: 2 * NBL tiles of 32 rows per 32 symbols, all fed by the same B operand per k-step, so every k-step is 2 * NBL
// independent v_mfma_f32_32x32x2_f32.  The constant A operands (the kernel table, up to 147 KB) stay in LDS for the whole
// workgroup.  The first-order kernel, mapped to lag indices, is the INITIAL value of the accumulators (xhat = sum_j Y_j (Z_j +
// k1'_j)), so no row needs special treatment and xhat = Y^T Z is a packed complex dot product over row pairs, lane-local plus
// one exchange with lane ^ 32.
//
// Backward: with rho = g0 Ya + g1 Yb, sigma = g1 Ya - g0 Yb (g = dloss/dxhat) the gradient of the effective kernel is
//     dKr = sum_b rho^T Ya + sigma^T Yb,   dKi = sum_b sigma^T Ya - rho^T Yb,
// both SYMMETRIC in the lag pair, so only the NBL (NBL + 1) / 2 upper-triangular 32x32 blocks are accumulated (two MFMAs into
// each of the two planes per block and k-step, two symbols per k-step).  A ones column at lag index l2 of the Ya operand (zero in
// the Yb operand) yields the first-order gradient in the same products.
//
// Applies when mode = complex, l2 odd <= 95, l1 odd <= l2 and pad1 = l1 / 2.
#include <hip/hip_runtime.h>
#include <stdlib.h>

#include "v2vae_cx_common.h"
#include "v2vae_layout.h"

namespace v2vae {

__device__ __forceinline__ void eqcxb_load_tile(const Lay& l, const float* __restrict__ y, int64_t g0, float* tile) {
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

// (the tables of the forward kernels are built by d_prep_eqcxb, v2vae_prep_eq.h, inside the launch that prepares every table)

template <int NBL>
__global__ __launch_bounds__(NT) void k_eq_fwd_cxb(Lay l, const float* __restrict__ y, const float* __restrict__ ws,
                                                   float* __restrict__ xhat) {
    extern __shared__ float sm[];
    float* tile = sm;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int bl = lane & 31, h = lane >> 5;
    const int steps = l.eqcx_steps, LK = l.eqcx_lk;
    float* s_A = tile + ((l.ns * l.tile_len + 63) & ~63);
    {   // the kernel table is staged once per (persistent) workgroup
        const float4* src = reinterpret_cast<const float4*>(ws + l.o_eqcxbA);
        float4* dst = reinterpret_cast<float4*>(s_A);
        for (int idx = tid; idx < steps * NBL * 32 + NBL * 16; idx += NT) dst[idx] = src[idx];
    }
    const float4* s_k1w = reinterpret_cast<const float4*>(s_A + (size_t)steps * NBL * 128);  // [2][NBL * 8] float4
    const float* tr = tile;
    const float* ti = tile + l.tile_len;
    const float* Al = s_A + lane;
    const int ntile = (int)((l.sym_cnt + EQ_TB - 1) / EQ_TB);
    for (int t = blockIdx.x; t < ntile; t += gridDim.x) {
    const int64_t b0 = l.sym_lo + (int64_t)t * EQ_TB;
    __syncthreads();
    eqcxb_load_tile(l, y, b0 * l.S - l.hl, tile);
    __syncthreads();
#pragma unroll 1
    for (int sub = 0; sub < 2; ++sub) {
        const int bloc = wv * 64 + sub * 32 + bl;           // symbol within the block tile
        const int base = bloc * l.S + l.hl + l.p2;          // tile index of lag 0
        // accumulators start from the first-order kernel of their rows (rows (g & 3) + 8 (g >> 2) + 4 h: four float4 per tile)
        f32x16 Z[NBL][2];
#pragma unroll
        for (int jb = 0; jb < NBL; ++jb)
#pragma unroll
            for (int pl = 0; pl < 2; ++pl)
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    const float4 v = s_k1w[pl * NBL * 8 + jb * 8 + 2 * g4 + h];
                    Z[jb][pl][4 * g4 + 0] = v.x; Z[jb][pl][4 * g4 + 1] = v.y;
                    Z[jb][pl][4 * g4 + 2] = v.z; Z[jb][pl][4 * g4 + 3] = v.w;
                }
        // stream 0 lags then stream 1 lags: k = 2 s + h runs over [0, 2 LK); LK is even, so a step never straddles the streams.
        // One wave per SIMD and the vector unit shares its pipe with the f32 MFMAs, so the loop is kept to MFMAs and LDS reads:
        // operands of the next step are fetched before the MFMAs of this one issue (two register sets, loop unrolled by two,
        // no copies), addresses are a per-lane base plus wave-uniform offsets.
        const int half = LK >> 1;                        // steps per stream
        const float* y0 = tr + base - h;                 // lag x = 2 s + h of stream 0 is y0[-2 s]
        const float* y1 = ti + base - h;
        auto fetch = [&](int s, float (&av)[2 * NBL], float& bv) {
            const int in1 = s >= half ? 1 : 0;           // wave-uniform
            bv = (in1 ? y1 : y0)[-2 * (s - in1 * half)];
            const float* As = Al + (size_t)s * NBL * 128;
#pragma unroll
            for (int q = 0; q < 2 * NBL; ++q) av[q] = As[q * 64];
        };
        auto mfmas = [&](const float (&av)[2 * NBL], float bv) {
#pragma unroll
            for (int jb = 0; jb < NBL; ++jb) {
                Z[jb][0] = MFMA32(av[2 * jb], bv, Z[jb][0]);
                Z[jb][1] = MFMA32(av[2 * jb + 1], bv, Z[jb][1]);
            }
        };
        float a0[2 * NBL], a1[2 * NBL], b0v, b1v;
        fetch(0, a0, b0v);
        for (int s = 0; s < steps; s += 2) {             // steps is even
            fetch(s + 1, a1, b1v);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(a0, b0v);
            __builtin_amdgcn_sched_barrier(0);
            fetch(s + 2 < steps ? s + 2 : s, a0, b0v);
            __builtin_amdgcn_sched_barrier(0);
            mfmas(a1, b1v);
            __builtin_amdgcn_sched_barrier(0);
        }
        // xhat = sum_j Y_j Z_j as a packed complex dot product over row pairs (j, j + 1), j even.  Rows from l2 on hold
        // zeros in Z; their samples only have to be finite, so the pair start is clamped to l2 - 1 (lag l2 is inside the tile).
        f2 re2 = 0.f, im2 = 0.f;
#pragma unroll
        for (int jb = 0; jb < NBL; ++jb)
#pragma unroll
            for (int g = 0; g < 16; g += 2) {
                const int j = jb * 32 + (g & 3) + 8 * (g >> 2) + 4 * h;
                const int a = j < l.l2 - 1 ? j : l.l2 - 1;
                const f2 yar = ld2(tr + base - a - 1), ybr = ld2(ti + base - a - 1);   // (y[a + 1], y[a])
                const f2 ya = __builtin_shufflevector(yar, yar, 1, 0), yb = __builtin_shufflevector(ybr, ybr, 1, 0);
                const f2 zr = f2{Z[jb][0][g], Z[jb][0][g + 1]}, zi = f2{Z[jb][1][g], Z[jb][1][g + 1]};
                re2 += ya * zr - yb * zi;
                im2 += ya * zi + yb * zr;
            }
        float re = re2.x + re2.y, im = im2.x + im2.y;
        re += __shfl_xor(re, 32);
        im += __shfl_xor(im, 32);
        const int64_t b = b0 + bloc;
        if (h == 0 && b < l.sym_lo + l.sym_cnt) {
            xhat[b - l.sym_lo] = re;
            xhat[l.sym_cnt + (b - l.sym_lo)] = im;
        }
    }
    }
}

// upper-triangular block index of (ib <= jb)
__host__ __device__ constexpr int eqcxb_blk(int nbl, int ib, int jb) { return ib * nbl - ib * (ib - 1) / 2 + (jb - ib); }

template <int NBL>
__global__ __launch_bounds__(NT) void k_eq_bwd_cxb(Lay l, const float* __restrict__ y, const float* __restrict__ gx,
                                                   float* __restrict__ ws) {
    constexpr int NBLK = NBL * (NBL + 1) / 2;
    extern __shared__ float sm[];
    float* tile = sm;
    float* gt = tile + l.ns * l.tile_len;  // [2][EQ_TB]
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    f32x16 G0[NBLK], G1[NBLK];
#pragma unroll
    for (int q = 0; q < NBLK; ++q)
#pragma unroll
        for (int g = 0; g < 16; ++g) { G0[q][g] = 0.f; G1[q][g] = 0.f; }
    const float* tr = tile;
    const float* ti = tile + l.tile_len;
    // lag index of this lane in every block; lags beyond l2 are clamped for the loads and masked in the column operands
    int jc[NBL];
    float mA[NBL], mB[NBL], oneA[NBL];
#pragma unroll
    for (int ib = 0; ib < NBL; ++ib) {
        const int a = ib * 32 + j;
        jc[ib] = a <= l.l2 ? a : l.l2;
        mA[ib] = a < l.l2 ? 1.f : 0.f;       // Ya column: the lag itself below l2 ...
        oneA[ib] = a == l.l2 ? 1.f : 0.f;    // ... a ones column at l2 (first-order gradient), zero beyond
        mB[ib] = a < l.l2 ? 1.f : 0.f;       // Yb column: zero from l2 on
    }
    for (int t = blockIdx.x; t < l.ntile_eq; t += gridDim.x) {
        const int64_t b0 = l.own_lo + (int64_t)t * EQ_TB;
        const int64_t rem = l.own_lo + l.own_cnt - b0;
        const int nbs = rem < EQ_TB ? (int)rem : EQ_TB;
        __syncthreads();
        eqcxb_load_tile(l, y, b0 * l.S - l.hl, tile);
        for (int idx = tid; idx < 2 * EQ_TB; idx += NT) {
            const int o = idx / EQ_TB, i = idx - o * EQ_TB;
            gt[idx] = i < nbs ? gx[(int64_t)o * l.sym_cnt + (b0 + i - l.sym_lo)] : 0.f;
        }
        __syncthreads();
        const int wbase = wv * 64;
#pragma unroll 2
        for (int s = 0; s < 32; ++s) {
            const int bloc = wbase + 2 * s + h;
            const int base = bloc * l.S + l.hl + l.p2;
            const float g0 = gt[bloc], g1 = gt[EQ_TB + bloc];
            float rho[NBL], sig[NBL], nrho[NBL], ca[NBL], cb[NBL];
#pragma unroll
            for (int ib = 0; ib < NBL; ++ib) {
                const float xa = tr[base - jc[ib]], xb = ti[base - jc[ib]];
                rho[ib] = g0 * xa + g1 * xb;
                sig[ib] = g1 * xa - g0 * xb;
                nrho[ib] = -rho[ib];
                ca[ib] = fmaf(xa, mA[ib], oneA[ib]);
                cb[ib] = xb * mB[ib];
            }
#pragma unroll
            for (int ib = 0; ib < NBL; ++ib)
#pragma unroll
                for (int jb = ib; jb < NBL; ++jb) {
                    const int q = eqcxb_blk(NBL, ib, jb);
                    G0[q] = MFMA32(rho[ib], ca[jb], G0[q]);
                    G1[q] = MFMA32(sig[ib], ca[jb], G1[q]);
                    G0[q] = MFMA32(sig[ib], cb[jb], G0[q]);
                    G1[q] = MFMA32(nrho[ib], cb[jb], G1[q]);
                }
        }
    }
    // fixed-order reduction of the four waves through LDS, then one record per block: [plane][block][32][32]
    __syncthreads();
    float* img = sm;
    for (int w = 0; w < 4; ++w) {
        if (wv == w) {
#pragma unroll
            for (int q = 0; q < NBLK; ++q)
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const int e = q * 1024 + ((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + j;
                    if (w == 0) {
                        img[e] = G0[q][g]; img[NBLK * 1024 + e] = G1[q][g];
                    } else {
                        img[e] += G0[q][g]; img[NBLK * 1024 + e] += G1[q][g];
                    }
                }
        }
        __syncthreads();
    }
    float* rec = ws + l.o_part_eqcxb + (size_t)blockIdx.x * (2 * NBLK * 1024);
    for (int e = tid; e < 2 * NBLK * 1024; e += NT) rec[e] = img[e];
}

__global__ __launch_bounds__(NT) void k_chain_equaliser_cxb(Lay l, const float* __restrict__ ws, float* __restrict__ g_k1,
                                                            float* __restrict__ g_k2) {
    const float* T = ws + l.o_tot_eqcxb;
    const int l1 = l.l1, l2 = l.l2, NBL = l.eqcxb_nbl, NBLK = NBL * (NBL + 1) / 2;
    const int n1 = 2 * l1, n2 = 2 * l2 * l2;
    auto GK = [&](int p, int a, int b) {  // d loss / d K (effective symmetric kernel) of plane p, any (a, b)
        if (a > b) { const int t = a; a = b; b = t; }
        const int ib = a >> 5, jb = b >> 5;
        return T[((size_t)p * NBLK + eqcxb_blk(NBL, ib, jb)) * 1024 + (a & 31) * 32 + (b & 31)];
    };
    for (int idx = blockIdx.x * NT + threadIdx.x; idx < n1 + n2; idx += gridDim.x * NT) {
        if (idx < n1) {
            if (!g_k1) continue;
            const int p = idx / l1, jj = idx - p * l1;
            const int x = l.pad1 + l.p2 - jj;  // lag row that first-order tap jj multiplies (x < l2: always above the ones column)
            g_k1[idx] = T[((size_t)p * NBLK + eqcxb_blk(NBL, x >> 5, l2 >> 5)) * 1024 + (x & 31) * 32 + (l2 & 31)];
        } else {
            if (!g_k2) continue;
            const int e = idx - n1;
            const int p = e / (l2 * l2), rem = e - p * l2 * l2;
            const int a = rem / l2, b = rem - a * l2;
            // inside a diagonal block both orientations were accumulated; across blocks the symmetric partner is the same entry
            const bool same = (a >> 5) == (b >> 5);
            const float v = same ? T[((size_t)p * NBLK + eqcxb_blk(NBL, a >> 5, a >> 5)) * 1024 + (a & 31) * 32 + (b & 31)] +
                                       T[((size_t)p * NBLK + eqcxb_blk(NBL, a >> 5, a >> 5)) * 1024 + (b & 31) * 32 + (a & 31)]
                                 : 2.f * GK(p, a, b);
            g_k2[e] = v;
        }
    }
}

// Forward equaliser in symmetric form on the vector pipe, for kernels of up to 95 lags (instances 9, 17, 25, 31, 47, 65, 79, 95):
//   xhat = sum_a y_a (k1w_a + sum_{b >= a} K'_ab y_b),  K'_ab = K_ab (a = b), 2 K_ab (a < b),  y_x = y[b S + p2 - x],
// T (T + 1) / 2 + T complex multiply-adds per symbol instead of the T^2 of the matrix product (the f32 matrix rate equals the
// packed vector rate on this part, so half the multiply-adds and no padding of the lag count to 32 is what counts).  One
// symbol per thread: its T samples sit in registers, the kernel entries are wave-uniform (scalar loads), every multiply-add is
// two packed FMAs.
typedef float eq_f2 __attribute__((ext_vector_type(2)));
typedef unsigned long long eq_k8 __attribute__((ext_vector_type(8)));   // eight (Re, Im) table entries

// Kernels of up to 31 lags take NSY = 2 symbols per thread (a tile of 512 symbols): every kernel entry fetched serves both, which
// halves the scalar-load waits the instruction stream is bound by.
// acc += k * y for complex numbers held as (Re, Im) register pairs: two packed FMAs on the pairs as they are -- the first takes
// Re k for both halves (op_sel), the second Im k against (-Im y, Re y) (op_sel swaps the halves of y, neg_lo negates the low one).
// k comes straight from the scalar registers a scalar load filled (_s) or from vector registers (_v); written as inline assembly
// because the compiler builds the broadcast and the rotated pair with scalar moves instead (two to four per entry).
__device__ __forceinline__ void eq_cmac_s(eq_f2& acc, unsigned long long k, eq_f2 y) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "+v"(acc) : "s"(k), "v"(y));
}
__device__ __forceinline__ void eq_cmac_v(eq_f2& acc, eq_f2 k, eq_f2 y) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
        "v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "+v"(acc) : "v"(k), "v"(y));
}

// End of a row of the symmetric form: u += w (the first-order weight, a scalar pair as loaded), t = u * y (packed complex
// product, one rounding per component), then (outr + i outi) += t in double.
__device__ __forceinline__ void eq_row_close(double& outr, double& outi, eq_f2 u, unsigned long long w, eq_f2 y) {
    eq_f2 t;
    asm("v_pk_add_f32 %1, %1, %2\n\t"
        "v_pk_mul_f32 %0, %1, %3 op_sel:[0,0] op_sel_hi:[0,1]\n\t"
        "v_pk_fma_f32 %0, %1, %3, %0 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,1,0]"
        : "=&v"(t), "+v"(u) : "s"(w), "v"(y));
    outr += (double)t.x;
    outi += (double)t.y;
}

template <int T>
__global__ __launch_bounds__(NT, T <= 25 ? 4 : T <= 65 ? 3 : 2) void k_eq_fwd_sym(Lay l, const float* __restrict__ y,
                                                                              const float* __restrict__ ws, float* __restrict__ xhat) {
    extern __shared__ float sm[];
    float* tile = sm;
    constexpr int NSY = T <= 31 ? 2 : 1, TB = NSY * EQ_TB;
    const int tid = threadIdx.x;
    const float* tab0 = ws + l.o_eqsym;
    constexpr int NE = T * (T + 1) / 2 + T;
    const int64_t sym_hi = l.sym_lo + l.sym_cnt;
    const int64_t ntile = (l.sym_cnt + TB - 1) / TB;
    // the instance reads T lags; table entries of lags from l2 on are zero, and T - l2 extra samples of left halo keep those
    // reads inside the tile (finite values), so no lag needs a bounds check
    const int xh = T - l.l2;
    const int tlen = l.tile_len + xh + (NSY - 1) * EQ_TB * l.S;
    for (int64_t t = blockIdx.x; t < ntile; t += gridDim.x) {
        const int64_t b0 = l.sym_lo + t * TB;
        // The table base is re-declared opaque (a scalar register pair) per tile: the table does not depend on the tile, and
        // hoisted out of this loop its T (T + 3) / 2 scalar loads would be held in (and spilled from) scalar registers.  The
        // opaque copy is cast to the constant address space so that the loads stay scalar loads; entries are (Re, Im) pairs.
        typedef const unsigned long long __attribute__((address_space(4))) * ctab_p;
        const float* tab_generic = tab0;
        asm volatile("" : "+s"(tab_generic));
        const ctab_p tab = (ctab_p)tab_generic;
        __syncthreads();
        {   // a tile inside the sequence and the local sample block (all but the ends) is copied without per-element checks
            const int64_t g0 = b0 * l.S - l.hl - xh, loc0 = g0 - l.y_lo;
            const bool inside = g0 >= 0 && g0 + tlen <= l.n_glob && loc0 >= 0 && loc0 + tlen <= l.y_len;   // uniform
            // the two streams interleaved: one 8-byte LDS read per sample
            const float* __restrict__ sa = y + loc0;
            const float* __restrict__ sb = y + l.y_stride + loc0;
            eq_f2* dst = reinterpret_cast<eq_f2*>(tile);
            if (inside) {
                // six samples per thread requested before the first is written (one memory latency per tile, not one per sample)
                for (int i0 = tid; i0 < tlen; i0 += 6 * NT) {
                    eq_f2 v[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const int i = i0 + k * NT;
                        v[k] = i < tlen ? eq_f2{sa[i], sb[i]} : eq_f2{0.f, 0.f};
                    }
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const int i = i0 + k * NT;
                        if (i < tlen) dst[i] = v[k];
                    }
                }
            } else {
                for (int i = tid; i < tlen; i += NT) {
                    const int64_t g = g0 + i, loc = loc0 + i;
                    const bool in = g >= 0 && g < l.n_glob && loc >= 0 && loc < l.y_len;
                    dst[i] = in ? eq_f2{sa[i], sb[i]} : eq_f2{0.f, 0.f};
                }
            }
        }
        __syncthreads();
        eq_f2 yv[NSY][T];
#pragma unroll
        for (int n = 0; n < NSY; ++n) {
            const int base = (tid + n * EQ_TB) * l.S + l.hl + xh + l.p2;
#pragma unroll
            for (int x = 0; x < T; ++x) yv[n][x] = reinterpret_cast<const eq_f2*>(tile)[base - x];
        }
        // The row sums u_a = sum_{b >= a} K'_ab y_b (small terms) are formed in packed float arithmetic, the first-order weight is
        // added to a finished row sum and the row's product with y_a is formed in float (one rounding each at the size of the
        // term), and the outer sum over the rows - whose partial sums are of the size of the result from the dominant
        // first-order tap on - runs in double, rounded to float once at the end: the soft demapper amplifies an error of xhat by 2 |xhat - a| / noise variance on its way into the equaliser
        // gradients, and a float accumulation over the rows alone left 2 - 4e-7 of relative error in xhat.
        double outr[NSY], outi[NSY];
        eq_f2 u[NSY];
#pragma unroll
        for (int n = 0; n < NSY; ++n) { outr[n] = 0.0; outi[n] = 0.0; }
        // the table is one stream of entries, fetched eight at a time (one 64-byte scalar load, one wait per eight entries)
        // wherever the flat index crosses a multiple of eight; all indices are compile-time constants after unrolling
        eq_k8 kk;
#pragma unroll
        for (int a = 0; a < T; ++a) {
            const int off = a * (T + 1) - a * (a - 1) / 2;   // flat index of K'_aa, which opens row a; the first-order weight closes it
#pragma unroll
            for (int n = 0; n < NSY; ++n) u[n] = eq_f2{0.f, 0.f};
#pragma unroll
            for (int b = a; b < T; ++b) {
                const int idx = off + (b - a);
                if ((idx & 7) == 0) kk = *reinterpret_cast<const eq_k8 __attribute__((address_space(4)))*>(tab + idx);
#pragma unroll
                for (int n = 0; n < NSY; ++n) eq_cmac_s(u[n], kk[idx & 7], yv[n][b]);
            }
            const int iw = off + (T - a);
            if ((iw & 7) == 0) kk = *reinterpret_cast<const eq_k8 __attribute__((address_space(4)))*>(tab + iw);
#pragma unroll
            for (int n = 0; n < NSY; ++n) eq_row_close(outr[n], outi[n], u[n], kk[iw & 7], yv[n][a]);
        }
#pragma unroll
        for (int n = 0; n < NSY; ++n) {
            const int64_t b = b0 + tid + n * EQ_TB;
            if (b < sym_hi) {
                xhat[b - l.sym_lo] = (float)outr[n];
                xhat[l.sym_cnt + (b - l.sym_lo)] = (float)outi[n];
            }
        }
    }
}

// The REAL model's forward in the same symmetric form, with every sum in double (v_fma_f64 issues at the rate of an unpacked f32
// FMA on this part): xhat = sum_a y_a (k1w_a + sum_{b >= a} K'_ab y_b) - T (T + 1) / 2 + T multiply-adds per symbol against the
// T^2 of the generic kernel, samples in registers as doubles, table entries (doubles; one row-interleaved stream, here with the first-order weight opening its row) by
// scalar loads, eight per load.  The result leaves as a float plus its float remainder (planes o_xhi / o_xlo), exactly like the
// generic real forward: the soft demapper amplifies an error of xhat by 2 |xhat - a| / noise variance.
typedef double eq_d8 __attribute__((ext_vector_type(8)));

__global__ __launch_bounds__(NT) void k_prep_eqsymr(Lay l, const float* __restrict__ k1, const float* __restrict__ k2, float* __restrict__ ws) {
    const int T = l.eqsym_t, ne = T * (T + 1) / 2 + T, l1 = l.l1, l2 = l.l2;
    double* tab = reinterpret_cast<double*>(ws + l.o_eqsym);
    for (int idx = blockIdx.x * NT + threadIdx.x; idx < ne + 8; idx += gridDim.x * NT) {
        double v = 0.0;
        if (idx < ne) {
            int a = 0, rem = idx;
            while (rem >= T - a + 1) { rem -= T - a + 1; ++a; }
            if (rem == 0) {
                const int jj = l.pad1 + l.p2 - a;  // first-order tap that multiplies lag a
                if (a < l2 && jj >= 0 && jj < l1) v = (double)k1[jj];
            } else {
                const int b = a + rem - 1;
                // the symmetrised entry as the generic kernel's table holds it (a float sum), doubled off the diagonal
                if (b < l2) v = (a == b ? 1.0 : 2.0) * (double)(k2[a * l2 + b] + k2[b * l2 + a]);
            }
        }
        tab[idx] = v;
    }
}

template <int T>
__global__ __launch_bounds__(NT, 2) void k_eq_fwd_symr(Lay l, const float* __restrict__ y, float* __restrict__ ws, float* __restrict__ xhat) {
    extern __shared__ float sm[];
    float* tile = sm;
    constexpr int NSY = T <= 31 ? 2 : 1, TB = NSY * EQ_TB;   // (the samples are held as doubles: two symbols per thread up to 31 lags)
    const int tid = threadIdx.x;
    const float* tab0 = ws + l.o_eqsym;
    const int64_t sym_hi = l.sym_lo + l.sym_cnt;
    const int64_t ntile = (l.sym_cnt + TB - 1) / TB;
    const int xh = T - l.l2;   // extra left halo: table entries of lags from l2 on are zero, their samples only have to be finite
    const int tlen = l.tile_len + xh + (NSY - 1) * EQ_TB * l.S;
    for (int64_t t = blockIdx.x; t < ntile; t += gridDim.x) {
        const int64_t b0 = l.sym_lo + t * TB;
        // (opaque per tile, constant address space: see k_eq_fwd_sym)
        typedef const double __attribute__((address_space(4))) * ctab_p;
        const float* tab_generic = tab0;
        asm volatile("" : "+s"(tab_generic));
        const ctab_p tab = (ctab_p)tab_generic;
        __syncthreads();
        {
            const int64_t g0 = b0 * l.S - l.hl - xh, loc0 = g0 - l.y_lo;
            const bool inside = g0 >= 0 && g0 + tlen <= l.n_glob && loc0 >= 0 && loc0 + tlen <= l.y_len;   // uniform
            const float* __restrict__ sa = y + loc0;
            if (inside) {
                // six samples per thread requested before the first is written (one memory latency per tile, not one per sample)
                for (int i0 = tid; i0 < tlen; i0 += 6 * NT) {
                    float v[6];
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const int i = i0 + k * NT;
                        v[k] = i < tlen ? sa[i] : 0.f;
                    }
#pragma unroll
                    for (int k = 0; k < 6; ++k) {
                        const int i = i0 + k * NT;
                        if (i < tlen) tile[i] = v[k];
                    }
                }
            } else {
                for (int i = tid; i < tlen; i += NT) {
                    const int64_t g = g0 + i, loc = loc0 + i;
                    tile[i] = (g >= 0 && g < l.n_glob && loc >= 0 && loc < l.y_len) ? sa[i] : 0.f;
                }
            }
        }
        __syncthreads();
        double yv[NSY][T];
#pragma unroll
        for (int n = 0; n < NSY; ++n) {
            const int base = (tid + n * EQ_TB) * l.S + l.hl + xh + l.p2;
#pragma unroll
            for (int x = 0; x < T; ++x) yv[n][x] = (double)tile[base - x];
        }
        double out[NSY], u[NSY];
#pragma unroll
        for (int n = 0; n < NSY; ++n) { out[n] = 0.0; u[n] = 0.0; }
        // Two alternating register sets of eight entries.  Scalar loads return out of order, so the only wait there is waits for
        // ALL of them: the load of the next eight entries must be issued AFTER the wait for the current eight, or that wait would
        // cover it too.  Left to the compiler the loads (invariant, constant address space) are all hoisted to the top of the tile
        // and hundreds of scalar registers are spilled; so load and wait are written out here - the wait names the register set
        // as in / out operand (its consumers stay behind it), the compiler, which does not know of a load in flight, adds no
        // wait of its own, and a scheduling fence at every boundary keeps the multiply-adds of a set between its wait and the
        // load that overwrites it.
        constexpr int NE = T * (T + 1) / 2 + T, NCH = (NE + 7) / 8;
        eq_d8 kb[2];
        auto issue = [&](int c) { asm volatile("s_load_dwordx16 %0, %1, %2" : "=s"(kb[c & 1]) : "s"(tab), "s"(c * 64)); };
        auto arrive = [&](int c) {
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(kb[c & 1]));
            if (c + 1 < NCH) issue(c + 1);
            __builtin_amdgcn_sched_barrier(0);
        };
        __builtin_amdgcn_s_waitcnt(0xC07F);   // the samples have arrived (a wait the compiler knows of: it adds none behind the loads)
        issue(0);
#pragma unroll
        for (int a = 0; a < T; ++a) {
            const int off = a * (T + 1) - a * (a - 1) / 2;   // flat index of the first-order weight that opens row a
            if ((off & 7) == 0) arrive(off >> 3);
#pragma unroll
            for (int n = 0; n < NSY; ++n) u[n] = kb[(off >> 3) & 1][off & 7];
#pragma unroll
            for (int b = a; b < T; ++b) {
                const int idx = off + 1 + (b - a);
                if ((idx & 7) == 0) arrive(idx >> 3);
#pragma unroll
                for (int n = 0; n < NSY; ++n)   // (volatile: keeps its place between the hand-placed loads and waits)
                    asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(u[n]) : "s"(kb[(idx >> 3) & 1][idx & 7]), "v"(yv[n][b]));
            }
#pragma unroll
            for (int n = 0; n < NSY; ++n) out[n] = fma(u[n], yv[n][a], out[n]);
        }
#pragma unroll
        for (int n = 0; n < NSY; ++n) {
            const int64_t b = b0 + tid + n * EQ_TB;
            if (b < sym_hi) {
                const float hi = (float)out[n];
                xhat[b - l.sym_lo] = hi;
                ws[l.o_xhi + (b - l.sym_lo)] = hi;
                ws[l.o_xlo + (b - l.sym_lo)] = (float)(out[n] - (double)hi);
            }
        }
    }
}

int launch_prep_eqsymr(const Lay& l, const float* k1, const float* k2, float* ws, hipStream_t st) {
    hipLaunchKernelGGL(k_prep_eqsymr, dim3(2), dim3(NT), 0, st, l, k1, k2, ws);
    return V2VAE_OK;
}

int launch_eq_fwd_symr(const Lay& l, const float* y, float* ws, float* xhat, hipStream_t st) {
    const int nsy = l.eqsym_t <= 31 ? 2 : 1;   // symbols per thread
    const int nt = (int)((l.sym_cnt + nsy * EQ_TB - 1) / (nsy * EQ_TB));
    const int nb = nt < 2 * CX_BLOCKS ? nt : 2 * CX_BLOCKS;   // resident workgroups: two per CU (launch bounds)
    const int lds = (l.tile_len + (l.eqsym_t - l.l2) + (nsy - 1) * EQ_TB * l.S) * 4;
    switch (l.eqsym_t) {
        case 9: hipLaunchKernelGGL(k_eq_fwd_symr<9>, dim3(nb), dim3(NT), lds, st, l, y, ws, xhat); break;
        case 17: hipLaunchKernelGGL(k_eq_fwd_symr<17>, dim3(nb), dim3(NT), lds, st, l, y, ws, xhat); break;
        case 25: hipLaunchKernelGGL(k_eq_fwd_symr<25>, dim3(nb), dim3(NT), lds, st, l, y, ws, xhat); break;
        case 31: hipLaunchKernelGGL(k_eq_fwd_symr<31>, dim3(nb), dim3(NT), lds, st, l, y, ws, xhat); break;
        case 47: hipLaunchKernelGGL(k_eq_fwd_symr<47>, dim3(nb), dim3(NT), lds, st, l, y, ws, xhat); break;
        case 65: hipLaunchKernelGGL(k_eq_fwd_symr<65>, dim3(nb), dim3(NT), lds, st, l, y, ws, xhat); break;
        default: return V2VAE_ESIZE;
    }
    return V2VAE_OK;
}

template <typename K>
static int eqcxb_allow_lds(K kernel, int bytes) {
    if (bytes <= 64 * 1024) return V2VAE_OK;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess
               ? V2VAE_OK : V2VAE_ELAUNCH;
}

int launch_eq_fwd_cxb(const Lay& l, const float* y, const float* ws, float* xhat, hipStream_t st) {
    int nblk = (int)((l.sym_cnt + EQ_TB - 1) / EQ_TB);
    if (l.eqsym_t && !(l.flags & V2VAE_FLAG_EQ_MFMA_FWD)) {
        const int per_cu = l.eqsym_t <= 25 ? 4 : l.eqsym_t <= 65 ? 3 : 2;  // resident workgroups per CU (launch bounds)
        const int nsy = l.eqsym_t <= 31 ? 2 : 1;                            // symbols per thread
        nblk = (nblk + nsy - 1) / nsy;
        const int nb = nblk < per_cu * CX_BLOCKS ? nblk : per_cu * CX_BLOCKS;
        const int lds = l.ns * (l.tile_len + (l.eqsym_t - l.l2) + (nsy - 1) * EQ_TB * l.S) * 4;
        if (l.eqsym_t == 9) hipLaunchKernelGGL(k_eq_fwd_sym<9>, dim3(nb), dim3(NT), lds, st, l, y, ws, xhat);
        else if (l.eqsym_t == 17) hipLaunchKernelGGL(k_eq_fwd_sym<17>, dim3(nb), dim3(NT), lds, st, l, y, ws, xhat);
        else if (l.eqsym_t == 25) hipLaunchKernelGGL(k_eq_fwd_sym<25>, dim3(nb), dim3(NT), lds, st, l, y, ws, xhat);
        else if (l.eqsym_t == 31) hipLaunchKernelGGL(k_eq_fwd_sym<31>, dim3(nb), dim3(NT), lds, st, l, y, ws, xhat);
        else if (l.eqsym_t == 47) hipLaunchKernelGGL(k_eq_fwd_sym<47>, dim3(nb), dim3(NT), lds, st, l, y, ws, xhat);
        else if (l.eqsym_t == 65) hipLaunchKernelGGL(k_eq_fwd_sym<65>, dim3(nb), dim3(NT), lds, st, l, y, ws, xhat);
        else if (l.eqsym_t == 79) hipLaunchKernelGGL(k_eq_fwd_sym<79>, dim3(nb), dim3(NT), lds, st, l, y, ws, xhat);
        else hipLaunchKernelGGL(k_eq_fwd_sym<95>, dim3(nb), dim3(NT), lds, st, l, y, ws, xhat);
        return V2VAE_OK;
    }
    // persistent workgroups: as many per CU as their LDS image (sample tile + kernel table) allows, at most four
    int per_cu = 160 * 1024 / (l.lds_eqcxb_fwd + 1024);
    per_cu = per_cu < 1 ? 1 : per_cu > 4 ? 4 : per_cu;
    if (l.eqcxb_nbl >= 2 && per_cu > 2) per_cu = 2;   // register use allows two workgroups per CU from two lag blocks on
    if (nblk > CX_BLOCKS * per_cu) nblk = CX_BLOCKS * per_cu;
    int rc;
#define V2VAE_LAUNCH_EQF(NBL)                                                                                     \
    do {                                                                                                          \
        if ((rc = eqcxb_allow_lds(k_eq_fwd_cxb<NBL>, l.lds_eqcxb_fwd))) return rc;                                \
        hipLaunchKernelGGL(k_eq_fwd_cxb<NBL>, dim3(nblk), dim3(NT), l.lds_eqcxb_fwd, st, l, y, ws, xhat);          \
    } while (0)
    if (l.eqcxb_nbl == 1) V2VAE_LAUNCH_EQF(1);
    else if (l.eqcxb_nbl == 2) V2VAE_LAUNCH_EQF(2);
    else if (l.eqcxb_nbl == 3) V2VAE_LAUNCH_EQF(3);
    else return V2VAE_ESIZE;
#undef V2VAE_LAUNCH_EQF
    return V2VAE_OK;
}

int launch_eq_bwd_cxb(const Lay& l, const float* y, const float* gx, float* ws, hipStream_t st) {
    int rc;
#define V2VAE_LAUNCH_EQB(NBL)                                                                                     \
    do {                                                                                                          \
        if ((rc = eqcxb_allow_lds(k_eq_bwd_cxb<NBL>, l.lds_eqcxb_bwd))) return rc;                                \
        hipLaunchKernelGGL(k_eq_bwd_cxb<NBL>, dim3(l.nblk_eq), dim3(NT), l.lds_eqcxb_bwd, st, l, y, gx, ws);       \
    } while (0)
    if (l.eqcxb_nbl == 2) V2VAE_LAUNCH_EQB(2);
    else if (l.eqcxb_nbl == 3) V2VAE_LAUNCH_EQB(3);
    else return V2VAE_ESIZE;
#undef V2VAE_LAUNCH_EQB
    return V2VAE_OK;
}

void launch_chain_equaliser_cxb(const Lay& l, float* ws, float* g_k1, float* g_k2, hipStream_t st) {
    const int n = 2 * l.l1 + 2 * l.l2 * l.l2;
    hipLaunchKernelGGL(k_chain_equaliser_cxb, dim3((n + NT - 1) / NT), dim3(NT), 0, st, l, (const float*)ws, g_k1, g_k2);
}

}  // namespace v2vae
