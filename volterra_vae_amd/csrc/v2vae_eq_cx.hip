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

#include "v2vae_bf16_split.h"
#include "v2vae_chain_cx.h"
#include "v2vae_layout.h"

namespace v2vae {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
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

// ------------------------------------------------------------------------------------------------------------
// The same sums on the bf16 matrix pipe in split arithmetic (one or two samples per symbol).  An f32 matrix instruction
// holds the SIMD's vector pipe for all of its cycles and runs at the vector rate; a bf16 one runs at sixteen times that rate
// and leaves the vector pipe free.  Every float is the exact sum of three bf16 pieces (8 + 8 + 8 significand bits, by
// truncation); the six piece products whose weight is not below 2^-24 of the whole, accumulated in float by
// v_mfma_f32_16x16x32_bf16 (32 symbols per instruction), give the product to float accuracy.
//   B operand (columns = lags, k = symbols): the samples themselves.  They are split once per tile when it is staged, into
//   planes [copy][parity][piece plane][word] of bf16 PAIRS: the eight symbols a lane needs for its lag are eight consecutive
//   elements of one parity plane (sample index = symbol * S + const), and the copy whose pairs start at the odd elements
//   serves the lags whose first element is odd, so that an operand is four aligned consecutive words.
//   A operand (rows = lags, k = symbols): rho = g0 Ya + g1 Yb, sigma = g1 Ya - g0 Yb of the lane's lag and eight symbols,
//   formed in float from the float tile and split in registers (plain vector instructions: they run beside the matrix pipe).
//   dK_r = sum rho^T Ya + sigma^T Yb,  dK_i = sum sigma^T Ya + (-rho)^T Yb: four products of 6 matrix instructions per
//   16x16 block pair and 32 symbols.
//   Columns on the side (XS = 1): with 16 k + 1 lags (the lag count is odd) the one lag beyond the k blocks and the column of
//   ones behind the lags (whose sums are the first-order gradient) would take a whole block row and column, k + 1 of the
//   (k + 1)(k + 2) / 2 block pairs.  Those instances keep NB16 = k blocks on the matrix pipe and form the two extra columns per
//   lane on the vector pipe from the rho / sigma it holds in float anyway (two additions, four multiply-adds per symbol and
//   row), and row e of the extra lag likewise; the record keeps its k + 1 blocks.
// ------------------------------------------------------------------------------------------------------------
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short bfx8 __attribute__((ext_vector_type(8)));
#define MFMA16BF(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfx8, (a)), __builtin_bit_cast(bfx8, (b)), (c), 0, 0, 0)

// rho / sigma are split in registers, per (symbol, lag): by truncation (one AND per piece)
__device__ __forceinline__ void bf_split3(float x, unsigned* p1, unsigned* p2, unsigned* p3) { bf16_split3_trunc(x, p1, p2, p3); }
// (high half of lo, high half of hi) as one word of two bf16
__device__ __forceinline__ unsigned bf_pack_hi(unsigned lo, unsigned hi) { return __builtin_amdgcn_perm(hi, lo, 0x07060302u); }
// Words between the (copy, parity) regions of the planes of bf16 pairs: NPL piece planes and a pad of eight words.  The lanes
// of one 4-byte read (32 banks of four bytes) cover about nine consecutive words in each of the four regions, so the regions are
// put eight banks apart.
template <bool REAL> constexpr int eqbf_rs() { return (REAL ? 3 : 6) * EQBF_NWD + 8; }

// Sample tile from global sample g0 on, the two streams INTERLEAVED ([tile_len][2] floats: the A side forms rho and sigma from
// them), and its bf16 pieces in the planes Yh (the B side).  Piece planes of a (copy, parity): piece dp of stream c (0: Ya,
// 1: Yb) is plane 2 dp + c (the two streams of a piece side by side: one address register reaches both), dp alone for the
// real model.  A tile that lies inside both the global sequence and the local sample block (all but the first and last few) is
// copied without per-element checks.
template <int SS, bool REAL>
__device__ __forceinline__ void eqbf_load_tile(const Lay& l, const float* __restrict__ y, int64_t g0, float* tile,
                                               unsigned short* Yh) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    constexpr int NPL = REAL ? 3 : 6, nwd = EQBF_NWD, RS = eqbf_rs<REAL>();
    const int tl = l.tile_len;
    const int64_t loc0 = g0 - l.y_lo;
    const bool inside = g0 >= 0 && g0 + tl <= l.n_glob && loc0 >= 0 && loc0 + tl <= l.y_len;   // uniform
    const float* __restrict__ sa = y + loc0;
    const float* __restrict__ sb = REAL ? sa : y + l.y_stride + loc0;
    f2* dst = reinterpret_cast<f2*>(tile);
    auto put = [&](int i, f2 v) {
        dst[i] = v;
        const int e = SS == 1 ? i : i >> 1, par = SS == 1 ? 0 : i & 1;
        unsigned pc[6];
        // (the samples are split once per tile: pieces rounded to nearest)
        if (REAL) bf16_split3_rne(v.x, &pc[0], &pc[1], &pc[2]);
        else { bf16_split3_rne(v.x, &pc[0], &pc[2], &pc[4]); bf16_split3_rne(v.y, &pc[1], &pc[3], &pc[5]); }
        // copy 0: element e is half e & 1 of word e >> 1, i.e. halfword e; copy 1 (pairs start at the odd elements): halfword e - 1
        unsigned short* d0 = Yh + 2 * ((0 * SS + par) * RS) + e;
        unsigned short* d1 = Yh + 2 * ((1 * SS + par) * RS) + e - 1;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
            d0[2 * pl * nwd] = (unsigned short)(pc[pl] >> 16);
            if (e >= 1) d1[2 * pl * nwd] = (unsigned short)(pc[pl] >> 16);
        }
    };
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
                if (i < tl) put(i, v[k]);
            }
        }
    } else {
        for (int i = threadIdx.x; i < tl; i += NT) {
            const int64_t g = g0 + i, loc = loc0 + i;
            const bool in = g >= 0 && g < l.n_glob && loc >= 0 && loc < l.y_len;
            put(i, in ? f2{sa[i], REAL ? 0.f : sb[i]} : f2{0.f, 0.f});
        }
    }
}

// LDS (floats): sample tile [tile_len][2] | (g0, g1) per symbol [EQ_TB] | constant (1, 0) samples [64 S][2] | planes of bf16
// pairs [2 copies][S parities] regions of [NPL][EQBF_NWD] words (+ pad) | one constant region (lag columns from l2 on: Ya
// pieces (1, 0, 0), Yb pieces 0)
// XS: -1 = every column in the blocks; 1 = 16 NB16 + 1 lags, the last lag e = 16 NB16 and the column of ones on the side
// (0 = the column of ones alone on the side: no supported lag count needs it, the lag count is odd)
// (waves per SIMD by register need; the extra lag's samples and sums take about 20 + 4 NB16 registers)
__host__ __device__ constexpr int eqbf_waves(int nb16, int xs) { return xs == 1 ? (nb16 <= 2 ? 3 : 2) : (nb16 <= 2 ? 4 : nb16 == 3 ? 3 : 2); }
template <int NB16, int SS, bool REAL, int XS>
__global__ __launch_bounds__(NT, eqbf_waves(NB16, XS)) void k_eq_bwd_bf(Lay l, const float* __restrict__ y, const float* __restrict__ gx,
                                                                  float* __restrict__ ws) {
    constexpr int NBLK = NB16 * (NB16 + 1) / 2, LD = 16 * (NB16 + (XS >= 0 ? 1 : 0)), S = SS, NPL = REAL ? 3 : 6, nwd = EQBF_NWD, RS = eqbf_rs<REAL>();
    constexpr int EX = 16 * NB16;   // XS = 1: the lag on the side
    static_assert(SS == 1 || SS == 2, "parity planes for one or two samples per symbol");
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef const unsigned __attribute__((address_space(3))) * lword_p;
    extern __shared__ __align__(16) float sm[];
    float* tile = sm;
    const int ntl = (2 * l.tile_len + 3) & ~3;
    f2* g2 = reinterpret_cast<f2*>(sm + ntl);
    const int o_ones = ntl + 2 * EQ_TB;
    unsigned* Yw = reinterpret_cast<unsigned*>(sm + o_ones + 128 * S);
    unsigned* Yc = Yw + 2 * S * RS;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i16 = lane & 15, kq = lane >> 4;
    f32x4 Ar[NBLK], Ai[NBLK];
#pragma unroll
    for (int q = 0; q < NBLK; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) { Ar[q][g] = 0.f; Ai[q][g] = 0.f; }
    // columns on the side: per lane and row block the sums over this lane's symbols of (rho, sigma) - the column of ones - and
    // of (rho + i sigma) paired with the samples of lag e; row e: its (e, e) and (e, ones) entries
    // Plain floats, and after every update each sum is made opaque to the compiler, so that it stays a chain of single-width
    // instructions of its own: left alone the compiler pairs these loop-carried sums into packed instructions (v_pk_fma_f32 /
    // v_pk_add_f32 with operand selects) between the bf16 matrix instructions, and the sums so formed came out different from run
    // to run (the entries of column e, nothing else; every variant that kept them unpacked was bitwise reproducible and right).
    float s1r[XS >= 0 ? NB16 : 1], s1i[XS >= 0 ? NB16 : 1], ser[XS == 1 ? NB16 : 1], sei[XS == 1 ? NB16 : 1];
    float reer = 0.f, reei = 0.f, re1r = 0.f, re1i = 0.f;
#pragma unroll
    for (int ib = 0; ib < (XS >= 0 ? NB16 : 1); ++ib) { s1r[ib] = 0.f; s1i[ib] = 0.f; }
#pragma unroll
    for (int ib = 0; ib < (XS == 1 ? NB16 : 1); ++ib) { ser[ib] = 0.f; sei[ib] = 0.f; }
    for (int idx = tid; idx < 128 * S; idx += NT) sm[o_ones + idx] = (idx & 1) ? 0.f : 1.f;
    for (int idx = tid; idx < 2 * S * RS; idx += NT) Yw[idx] = 0u;
    for (int idx = tid; idx < RS; idx += NT) Yc[idx] = idx < nwd ? 0x3F803F80u : 0u;
    // per-lane operand positions.  A side: the float samples of symbols (wave base + 8 kq + k) at this lane's lag of block ib.
    // B side: four words from `pb` on (+ 16 words per 32 symbols) in each of the NPL planes of the lane's (copy, parity).
    const f2* pa[NB16];
    lword_p pb[NB16];
    {
        const int base0 = (wv * 64 + 8 * kq) * S + l.hl + l.p2;
#pragma unroll
        for (int ib = 0; ib < NB16; ++ib) {
            const int j = 16 * ib + i16;
            pa[ib] = reinterpret_cast<const f2*>(sm + (j >= l.l2 ? o_ones : 2 * (base0 - j)));
            const int c = l.hl + l.p2 - j;                          // sample index of symbol 0 of the tile at this lag
            const int off = S == 1 ? c : c >> 1, par = S == 1 ? 0 : c & 1;
            pb[ib] = (lword_p)(j >= l.l2 ? Yc : Yw + ((off & 1) * S + par) * RS + (32 * wv + 4 * kq + (off >> 1)));
        }
    }
    for (int t = blockIdx.x; t < l.ntile_eq; t += gridDim.x) {
        const int64_t b0 = l.own_lo + (int64_t)t * EQ_TB;
        const int64_t rem = l.own_lo + l.own_cnt - b0;
        const int nbs = rem < EQ_TB ? (int)rem : EQ_TB;
        __syncthreads();
        eqbf_load_tile<SS, REAL>(l, y, b0 * l.S - l.hl, tile, reinterpret_cast<unsigned short*>(Yw));
        {   // EQ_TB == NT: one symbol per thread
            const float g0 = tid < nbs ? gx[b0 + tid - l.sym_lo] : 0.f;
            const float g1 = (!REAL && tid < nbs) ? gx[l.sym_cnt + (b0 + tid - l.sym_lo)] : 0.f;
            g2[tid] = f2{g0, g1};
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const f32x4* g4 = reinterpret_cast<const f32x4*>(g2 + wv * 64 + 32 * s + 8 * kq);   // (g0, g1) of two symbols per read
            float yex[XS == 1 ? 8 : 1], yey[XS == 1 ? 8 : 1];
            if (XS == 1) {   // the samples of the lane's eight symbols at lag e, and row e's own sums
                const f2* pe = reinterpret_cast<const f2*>(tile) + (wv * 64 + 8 * kq + 32 * s) * S + l.hl + l.p2 - EX;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const f2 v = pe[k * S];
                    yex[k] = v.x; yey[k] = v.y;
                }
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    const f32x4 gg = g4[k2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const float xx = yex[2 * k2 + u], xy = yey[2 * k2 + u];
                        const float g0 = gg[2 * u], g1 = gg[2 * u + 1];
                        const float r = REAL ? g0 * xx : fmaf(g1, xy, g0 * xx);
                        const float sg = REAL ? 0.f : fmaf(g1, xx, -(g0 * xy));
                        re1r += r;
                        re1i += sg;
                        reer = fmaf(sg, xy, fmaf(r, xx, reer));
                        if (!REAL) reei = fmaf(-r, xy, fmaf(sg, xx, reei));
                        asm volatile("" : "+v"(reer));
                        asm volatile("" : "+v"(reei));
                        asm volatile("" : "+v"(re1r));
                        asm volatile("" : "+v"(re1i));
                    }
                }
            }
#pragma unroll
            for (int ib = 0; ib < NB16; ++ib) {
                // ---- A operands of block ib: rho, sigma, -rho in three pieces each, two symbols per word (plain vector
                //      instructions throughout: packed f32 arithmetic stalls beside the bf16 matrix instructions)
                u32x4 rho[3], sig[3], nrho[3];
#pragma unroll
                for (int k2 = 0; k2 < 4; ++k2) {
                    unsigned pr[2][3], ps[2][3];
                    const f32x4 gg = g4[k2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int k = 2 * k2 + u;
                        const f2 x = pa[ib][(32 * s + k) * S];
                        const float g0 = gg[2 * u], g1 = gg[2 * u + 1];
                        const float r = REAL ? g0 * x.x : fmaf(g1, x.y, g0 * x.x);
                        bf_split3(r, &pr[u][0], &pr[u][1], &pr[u][2]);
                        const float sg = REAL ? 0.f : fmaf(g1, x.x, -(g0 * x.y));
                        if (!REAL) bf_split3(sg, &ps[u][0], &ps[u][1], &ps[u][2]);
                        if (XS >= 0) {
                            s1r[ib] += r; s1i[ib] += sg;
                            if (XS == 0) { asm volatile("" : "+v"(s1r[ib])); asm volatile("" : "+v"(s1i[ib])); }
                        }
                        if (XS == 1) {
                            const float vx = yex[k], vy = yey[k];
                            ser[ib] = fmaf(sg, vy, fmaf(r, vx, ser[ib]));
                            if (!REAL) sei[ib] = fmaf(-r, vy, fmaf(sg, vx, sei[ib]));
                            // (each sum kept a single-width chain of its own: see the note at the declarations)
                            asm volatile("" : "+v"(ser[ib]));
                            asm volatile("" : "+v"(sei[ib]));
                            asm volatile("" : "+v"(s1r[ib]));
                            asm volatile("" : "+v"(s1i[ib]));
                        }
                    }
#pragma unroll
                    for (int p = 0; p < 3; ++p) {
                        rho[p][k2] = bf_pack_hi(pr[0][p], pr[1][p]);
                        if (!REAL) {
                            sig[p][k2] = bf_pack_hi(ps[0][p], ps[1][p]);
                            nrho[p][k2] = rho[p][k2] ^ 0x80008000u;
                        }
                    }
                }
#pragma unroll
                for (int jb = ib; jb < NB16; ++jb) {
                    const int q = ib * NB16 - ib * (ib - 1) / 2 + (jb - ib);
                    // piece dp of the B operand with the pieces tp <= 2 - dp of the A operand: the six products whose weight
                    // is not below 2^-24 of the whole, the small pieces first
#pragma unroll
                    for (int dp = 2; dp >= 0; --dp) {
                        // (one opaque address register per piece: the offset field of the reads reaches the plane of stream b)
                        lword_p pw = pb[jb] + (REAL ? dp : 2 * dp) * nwd + 16 * s;
                        asm volatile("" : "+v"(pw));
                        u32x4 ya, yb;
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            ya[w] = pw[w];
                            if (!REAL) yb[w] = pw[nwd + w];
                        }
#pragma unroll
                        for (int tp = 2 - dp; tp >= 0; --tp) {
                            Ar[q] = MFMA16BF(rho[tp], ya, Ar[q]);
                            if (!REAL) {
                                Ai[q] = MFMA16BF(sig[tp], ya, Ai[q]);
                                Ar[q] = MFMA16BF(sig[tp], yb, Ar[q]);
                                Ai[q] = MFMA16BF(nrho[tp], yb, Ai[q]);
                            }
                        }
                    }
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
                        img[e] += Ar[q][g];
                        if (!REAL) img[LD * LD + e] += Ai[q][g];
                    }
                }
            if (XS >= 0) {   // the columns on the side: the four symbol groups of the wave added on the vector pipe, lanes 0-15 write
#pragma unroll
                for (int ib = 0; ib < NB16; ++ib) {
                    float ax = s1r[ib], ay = s1i[ib], bx = XS == 1 ? ser[ib] : 0.f, by = XS == 1 ? sei[ib] : 0.f;
                    ax += __shfl_xor(ax, 16); ay += __shfl_xor(ay, 16); bx += __shfl_xor(bx, 16); by += __shfl_xor(by, 16);
                    ax += __shfl_xor(ax, 32); ay += __shfl_xor(ay, 32); bx += __shfl_xor(bx, 32); by += __shfl_xor(by, 32);
                    if (lane < 16) {
                        const int row = (16 * ib + lane) * LD;
                        img[row + EX + XS] += ax;
                        if (!REAL) img[LD * LD + row + EX + XS] += ay;
                        if (XS == 1) {
                            img[row + EX] += bx;
                            if (!REAL) img[LD * LD + row + EX] += by;
                        }
                    }
                }
                if (XS == 1) {   // row e (the sixteen lanes of a symbol group hold the same sums: one of them counts)
                    float ax = re1r, ay = re1i, bx = reer, by = reei;
                    ax += __shfl_xor(ax, 16); ay += __shfl_xor(ay, 16); bx += __shfl_xor(bx, 16); by += __shfl_xor(by, 16);
                    ax += __shfl_xor(ax, 32); ay += __shfl_xor(ay, 32); bx += __shfl_xor(bx, 32); by += __shfl_xor(by, 32);
                    if (lane == 0) {
                        img[EX * LD + EX + 1] += ax;
                        img[EX * LD + EX] += bx;
                        if (!REAL) { img[LD * LD + EX * LD + EX + 1] += ay; img[LD * LD + EX * LD + EX] += by; }
                    }
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
    if (l.eqbf) {
#define V2VAE_LAUNCH_EQBF_R(NB, SS, RR, XS)                                                                                      \
    do {                                                                                                                         \
        if (lds > 64 * 1024 && hipFuncSetAttribute(reinterpret_cast<const void*>(k_eq_bwd_bf<NB, SS, RR, XS>),                    \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)               \
            return V2VAE_ELAUNCH;                                                                                                \
        hipLaunchKernelGGL((k_eq_bwd_bf<NB, SS, RR, XS>), dim3(l.nblk_eq), dim3(NT), lds, st, l, y, gx, ws);                      \
    } while (0)
#define V2VAE_LAUNCH_EQBF(NB, XS)                                            \
    do {                                                                     \
        if (l.S == 1) { if (l.ns == 1) V2VAE_LAUNCH_EQBF_R(NB, 1, true, XS); else V2VAE_LAUNCH_EQBF_R(NB, 1, false, XS); } \
        else { if (l.ns == 1) V2VAE_LAUNCH_EQBF_R(NB, 2, true, XS); else V2VAE_LAUNCH_EQBF_R(NB, 2, false, XS); }          \
    } while (0)
#define V2VAE_LAUNCH_EQBF_X(NB)                                  \
    do {   /* (complex model only) */                            \
        if (l.S == 1) V2VAE_LAUNCH_EQBF_R(NB, 1, false, 1);      \
        else V2VAE_LAUNCH_EQBF_R(NB, 2, false, 1);               \
    } while (0)
        if (l.eqbf_xs == 1) {   // 16 k + 1 lags: k blocks, the last lag and the column of ones on the side (at 81 the sixth block stays)
            switch (l.eqbf_nb) {
                case 1: V2VAE_LAUNCH_EQBF_X(1); break;
                case 2: V2VAE_LAUNCH_EQBF_X(2); break;
                case 3: V2VAE_LAUNCH_EQBF_X(3); break;
                case 4: V2VAE_LAUNCH_EQBF_X(4); break;
                default: return V2VAE_ESIZE;
            }
            return V2VAE_OK;
        }
        switch (l.eqb16) {
            case 1: V2VAE_LAUNCH_EQBF(1, -1); break;
            case 2: V2VAE_LAUNCH_EQBF(2, -1); break;
            case 3: V2VAE_LAUNCH_EQBF(3, -1); break;
            case 4: V2VAE_LAUNCH_EQBF(4, -1); break;
            case 5: V2VAE_LAUNCH_EQBF(5, -1); break;
            case 6: V2VAE_LAUNCH_EQBF(6, -1); break;
            default: return V2VAE_ESIZE;
        }
#undef V2VAE_LAUNCH_EQBF_X
#undef V2VAE_LAUNCH_EQBF
#undef V2VAE_LAUNCH_EQBF_R
        return V2VAE_OK;
    }
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
