// Fused ELBO forward + backward for the complex V2VAE with LONG channel memory on the f32 matrix cores:
// 16 < Le <= 48 active taps per stream and phase (e.g. L = 65 at two samples per symbol: Le = 33), where the
// window-by-kernel contraction is a genuine dense product.
//
// Same mathematics and orientation as the <= 16-tap kernel (v2vae_elbo_cx.hip: windows on the lanes, taps in the
// accumulator registers), with the taps cut into NB blocks of 16:
//     [P | Q](xb)   = sum_jb [mu_a | mu_b](jb) . [[Hr, Hi], [-Hi, Hr]](jb, xb)       16 MFMA 32x32x2 per block pair
//     [HWa|HWb](xb) = sum_jb [alpha | beta](jb) . [[Hr, -Hi], [Hi, Hr]](jb, xb)       16
//     (no (H.H) s product: that term of E and its variance gradient are lagged-product sums formed by k_g2_cx)
//     dE/dHr(xb,jb) += rho^T mu_a + sigma^T mu_b,  dE/dHi(xb,jb) += sigma^T mu_a - rho^T mu_b
//                                                                   4 x 8 MFMA 16x16x4 (4 accumulator registers each)
// The NB^2 x 40 constant A-operand images (92 KB per phase at NB = 3) do not fit LDS next to the tiles; they stay in
// global memory (L2-resident, every wave streams the same sequence) and are fetched ahead of the MFMA chains.
// One workgroup = 4 waves, ONE per SIMD: the per-block register images of z, w, ... and the NB^2 gradient tiles need the
// full 512-register budget - except with one block and a tap on the side (17 taps), which fits 256 registers: that instance
// runs eight waves, two per SIMD, on 256-window tiles (1.15 against 1.38 ms per 2^22 symbols).  The per-(window, tap) algebra runs on tap PAIRS (accumulator registers 2 p, 2 p + 1 hold
// neighbouring taps) in explicit two-wide arithmetic over structure-of-arrays moment planes.
//
// Tap counts of the form 16 NB + 1 (L = 65 at two samples per symbol: 33) would waste a whole block row and column on one
// tap.  The XT = 1 instances keep NB blocks on the matrix cores and carry tap e = 16 NB on the side: as a contraction index
// it is ONE more MFMA step per output block (both streams of tap e fill the two k-slots); as an output row it is a dot
// product per window on the vector ALU; its row and column of dE/dH are accumulated per lane in the transposed layout the
// 16x16 MFMAs read.  Tables and records keep the stride 16 (NB + 1).
#include <hip/hip_runtime.h>

#include "v2vae_bf16_split.h"
#include "v2vae_cx_common.h"
#include "v2vae_layout.h"

namespace v2vae {

#ifndef CXB_HOLDB
#define CXB_HOLDB 1
#endif
// compact tap k of phase r -> channel tap index or -1
__device__ __forceinline__ int cxb_tap(const Lay& l, int r, int k) {
    if (k >= l.LE) return -1;
    const int i = r + l.L - 1 - k * l.S;
    return (i >= 0 && i < l.L) ? i : -1;
}

// symmetrised complex second-order kernel on compact taps (a, b) of phase r: plane 0 = Re, 1 = Im
__device__ __forceinline__ float cxb_H(const Lay& l, const float* __restrict__ h2, int r, int plane, int a, int b) {
    const int ta = cxb_tap(l, r, a), tb = cxb_tap(l, r, b);
    if (ta < 0 || tb < 0) return 0.f;
    const float* hp = h2 + (size_t)plane * l.L * l.L;
    return hp[ta * l.L + tb] + hp[tb * l.L + ta];
}

__global__ __launch_bounds__(NT) void k_prep_cxb(Lay l, const float* __restrict__ h1, const float* __restrict__ h2,
                                                 float* __restrict__ ws) {
    const int NB = l.cxb_nbk, LEB = l.cxb_leb;
    const int tid = blockIdx.x * NT + threadIdx.x, nth = gridDim.x * NT;
    for (int idx = tid; idx < l.S * 2 * LEB * LEB; idx += nth) {
        const int b = idx % LEB, a = (idx / LEB) % LEB, p = (idx / (LEB * LEB)) % 2, r = idx / (2 * LEB * LEB);
        ws[l.o_cxbH + idx] = cxb_H(l, h2, r, p, a, b);
    }
    for (int idx = tid; idx < l.S * 4 * LEB; idx += nth) {
        const int x = idx % LEB, q = (idx / LEB) % 4, r = idx / (4 * LEB);
        const int t = cxb_tap(l, r, x);
        float v = 0.f;
        if (t >= 0) v = q == 0 ? h1[t] : q == 1 ? h1[l.L + t] : cxb_H(l, h2, r, q - 2, x, x);
        ws[l.o_cxbc + idx] = v;
    }
    // Constant operand images [r][xb][jb][2 tables][3 pieces][2 chunks][64 lanes][4 words of two bf16]: the tables of the H X and
    // H w products as three exact bf16 pieces (split arithmetic on the bf16 matrix pipe).  Lane (row jj = lane & 31 -> (stream
    // cjj, tap 16 xb + (jj & 15)) of the output, half hh = lane >> 5); element j of chunk c is the entry of the k-slot
    //   H X: (stream c, tap 16 jb + 8 hh + j)           [[Hr, Hi], [-Hi, Hr]]
    //   H w: (stream c, tap 16 jb + cx_kk(j, hh))       [[Hr, -Hi], [Hi, Hr]]   (the tap accumulator slot j of that half holds)
    const int64_t total = (int64_t)l.S * NB * NB * CXB_STEPS * 64;
    for (int64_t idx = tid; idx < total; idx += nth) {
        const int f = (int)(idx % (CXB_STEPS * 64));
        const int pair = (int)((idx / (CXB_STEPS * 64)) % (NB * NB)), r = (int)(idx / ((int64_t)CXB_STEPS * 64 * NB * NB));
        const int xb = pair / NB, jb = pair % NB;
        const int w = f & 3, ln = (f >> 2) & 63, ck = (f >> 8) & 1, pc = (f >> 9) % 3, tb = f / 1536;
        const int jj = ln & 31, hh = ln >> 5, cjj = jj >> 4, yo = 16 * xb + (jj & 15);
        unsigned word = 0;
        for (int u = 0; u < 2; ++u) {
            const int j = 2 * w + u;
            const int xi = 16 * jb + (tb ? cx_kk(j, hh) : 8 * hh + j);
            const float a = cxb_H(l, h2, r, 0, xi, yo), b = cxb_H(l, h2, r, 1, xi, yo);
            float t = tb ? (ck == 0 ? (cjj == 0 ? a : -b) : (cjj == 0 ? b : a)) : (ck == 0 ? (cjj == 0 ? a : b) : (cjj == 0 ? -b : a));
            word |= bf16_piece_rne(t, pc) << (16 * u);   // (split once per step: pieces rounded to nearest)
        }
        ws[l.o_cxbA + idx] = __uint_as_float(word);
    }
    // extra-tap steps [r][xb][kind][64 lanes] behind the block-pair images: the two k-slots are the two streams of tap e = 16 NB;
    // kind 0 = [[Hr, Hi], [-Hi, Hr]] (pass A), kind 1 = [[Hr, -Hi], [Hi, Hr]] (pass B)
    if (l.cxb_xt) {
        const int e = 16 * NB;
        for (int idx = tid; idx < l.S * NB * 2 * 64; idx += nth) {
            const int ln = idx & 63, kind = (idx >> 6) & 1, xb = (idx >> 7) % NB, r = (idx >> 7) / NB;
            const int jj = ln & 31, ck = ln >> 5, cjj = jj >> 4, yo = 16 * xb + (jj & 15);
            const float a = cxb_H(l, h2, r, 0, e, yo), b = cxb_H(l, h2, r, 1, e, yo);
            float v;
            if (kind == 0) v = ck == 0 ? (cjj == 0 ? a : b) : (cjj == 0 ? -b : a);
            else v = ck == 0 ? (cjj == 0 ? a : -b) : (cjj == 0 ? b : a);
            ws[l.o_cxbA + total + idx] = v;
        }
    }
}

template <int S, int NB, int XT>
__global__ __launch_bounds__(cxb_nw(NB, XT) * 64) void k_elbo_cxb(Lay l, const float* __restrict__ y, float* __restrict__ ws) {
    extern __shared__ __align__(16) float sm[];
    constexpr int TWB = cxb_twb(NB, XT), NU = TWB / 32, TP = XT ? CXB_TPX : CX_TP, NW = cxb_nw(NB, XT), CXB_NT = NW * 64;
    constexpr int LEB = 16 * (NB + XT), EX = 16 * NB, MS = TWB + LEB + 16, NSIG = 32 + LEB, NWP = NW / S;
    constexpr int REC = 2 * LEB * LEB + 4 * LEB;
    constexpr int STG = 7 * MS + 2 * S * TWB;
    // seven planes [MS] of per-symbol quantities, as the pair arithmetic reads them (two neighbouring taps per register pair):
    // mean_a, mean_b, v_a + v_b, v_a - v_b, 3rd moment a, 3rd moment b, 4th cumulant a + b; then y' [2][S][TWB]
    float* s_stage = sm;
    float* s_c = s_stage + STG;                // [S][4][LEB]  h1r, h1i, diag Hr, diag Hi
    float* s_T = s_c + S * 4 * LEB;            // [NW][32 * TP]  transposing buffer of rho / sigma
    float* s_P = s_T + NW * 32 * TP;           // [S][NU][4][NSIG] anti-diagonal partial sums
    constexpr int TVP = 81;                    // row pitch of the second transposing buffer (odd multiple of 1 mod 64 spreads the banks)
    constexpr int RECA = S * REC > NW * 32 * TVP ? S * REC : NW * 32 * TVP;
    float* s_rec = s_P + S * NU * 4 * NSIG;    // [S][REC] after the last tile; before that [NW][32 windows][TVP]: per wave the four
                                               // per-(window, tap) terms of the vector gradients of one tap block, to be transposed
    float* s_xe = s_rec + RECA;                // XT: [S][2][LEB] row e of the symmetrised kernel (Re, Im)
    // the means in three exact bf16 pieces each, [2 streams][3 pieces][MS] (two bytes each): B operands of the H X product
    unsigned short* s_xb = reinterpret_cast<unsigned short*>(s_xe + (XT ? S * 2 * LEB : 0));
    __shared__ double dred[NW];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    // wave-uniform by construction; saying so lets the table loads use scalar base + lane offset addressing (one 64-bit
    // base per wave instead of one 64-bit address per lane and load)
    const int r = __builtin_amdgcn_readfirstlane(wv / NWP), wsub = __builtin_amdgcn_readfirstlane(wv % NWP);
    float* Tb = s_T + wv * 32 * TP;
    const float* __restrict__ mom = ws + l.o_mom;
    float* __restrict__ gmom = ws + l.o_gmom;
    const int64_t own_hi = l.own_lo + l.own_cnt, sym_hi = l.sym_lo + l.sym_cnt;
    const int TS = TWB - l.kmax;

    for (int idx = tid; idx < S * 4 * LEB; idx += CXB_NT) s_c[idx] = ws[l.o_cxbc + idx];
    if (XT)
        for (int idx = tid; idx < S * 2 * LEB; idx += CXB_NT) {
            const int j = idx % LEB, rp = idx / LEB;
            s_xe[idx] = ws[l.o_cxbH + ((size_t)rp * LEB + EX) * LEB + j];
        }
    const float* xe = s_xe + r * 2 * LEB;   // xe[j] = Re H(j, e), xe[LEB + j] = Im H(j, e)
    const float* __restrict__ Abase0 = ws + l.o_cxbA + (size_t)r * NB * NB * CXB_STEPS * 64;
    const unsigned ulane = lane;
    const float* cst = s_c + r * 4 * LEB;
    __syncthreads();
    // dE/dH block (xb, jb) = sum over windows of (rho + i sigma)^T conj(X), a complex 16x16 accumulation, from three real products
    // instead of four:  K1 = (rho + sigma)^T mu_a,  K2 = -rho^T (mu_a + mu_b),  K3 = sigma^T (mu_a - mu_b);  Re = K1 - K3,
    // Im = K1 + K2 (combined once at the end: the sums are linear)
    // BFK (the two-wave instance): the same accumulation on the bf16 matrix pipe in split arithmetic (v_mfma_f32_16x16x32_bf16,
    // the slab's 32 windows one k-step): Re += rho^T mu_a + sigma^T mu_b, Im += sigma^T mu_a - rho^T mu_b, rho / sigma split in
    // registers, the means read from the bf16 planes of the H X product - an f32 matrix instruction holds the vector issue of
    // the SIMD's other wave.  K1 holds Re and K2 Im then, K3 is not used.
    constexpr bool BFK = cxb_bfk(NB, XT);
    f32x4 K1[NB][NB], K2[NB][NB], K3[BFK ? 1 : NB][BFK ? 1 : NB];
    // vector gradients (d/dh1 Re, Im; d/d diag H Re, Im), accumulated in the TRANSPOSED layout the 16x16 MFMAs read (lane = (tap
    // lane & 15, window group lane >> 4)): one register per block and quantity instead of eight in the window-on-lane layout
    float vacc[NB][4];
    float* Tv = s_rec + wv * 32 * TVP;
#pragma unroll
    for (int a = 0; a < NB; ++a) {
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int g = 0; g < 4; ++g) { K1[a][b][g] = 0.f; K2[a][b][g] = 0.f; if (!BFK) K3[a][b][g] = 0.f; }
#pragma unroll
        for (int q = 0; q < 4; ++q) vacc[a][q] = 0.f;
    }
    // extra tap e (XT): vector gradients, the (e, e) element, and row / column e of dE/dH in the transposed layout
    // (lane = (tap lane & 15, window group lane >> 4))
    float ghr_e = 0.f, ghi_e = 0.f, gdr_e = 0.f, gdi_e = 0.f, gee_r = 0.f, gee_i = 0.f;
    f2 col[NB], row[NB];   // (Re, Im) of this lane's share of column e / row e of dE/dH
#pragma unroll
    for (int a = 0; a < NB; ++a) { col[a] = 0.f; row[a] = 0.f; }
    const float* __restrict__ Xbase0 = ws + l.o_cxbA + (size_t)S * NB * NB * CXB_STEPS * 64 + (size_t)r * NB * 2 * 64;
    double Eacc = 0.0;
#ifdef V2VAE_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#define BSTAMP(k)                                                       \
    do {                                                                \
        __builtin_amdgcn_sched_barrier(0);                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
        __builtin_amdgcn_s_waitcnt(0xC07F);                             \
        st_acc[k] += now_ - st_last;                                    \
        st_last = now_;                                                 \
        __builtin_amdgcn_sched_barrier(0);                              \
    } while (0)
#else
#define BSTAMP(k)
#endif

    // staging of one block tile through registers: the loads of tile n + 1 are in flight while tile n is computed
    constexpr int NSL = (MS + CXB_NT - 1) / CXB_NT;
    constexpr int NYW = (2 * S * TWB + CXB_NT - 1) / CXB_NT;
    float4 prea[NSL], preb[NSL];
    float prey[NYW];
    auto stage_load = [&](int tile) {
        const int64_t s0 = l.own_lo + (int64_t)tile * TS;
        const int64_t s1 = (s0 + TS < own_hi) ? s0 + TS : own_hi;
        const int64_t tbase = s0 - l.kmax;
#pragma unroll
        for (int u = 0; u < NSL; ++u) {
            const int ms = tid + u * CXB_NT;
            const int64_t m = tbase + ms;
            float4 a = make_float4(0.f, 0.f, 0.f, 0.f), b = a;
            if (ms < MS && m >= l.sym_lo && m < sym_hi) {
                const float* g = mom + (m - l.sym_lo);
                const int64_t st = l.sym_cnt;
                a = make_float4(g[0], g[st], g[2 * st], g[3 * st]);
                b = make_float4(g[4 * st], g[5 * st], g[6 * st], g[7 * st]);
            }
            prea[u] = a; preb[u] = b;
        }
#pragma unroll
        for (int u = 0; u < NYW; ++u) {
            const int e = tid + u * CXB_NT;
            const int o = e / (S * TWB), rem = e - o * S * TWB;
            const int rr = rem / TWB, slot = rem - rr * TWB;
            const int64_t t = tbase + slot, n = t * S + rr;
            float v = 0.f;
            if (e < 2 * S * TWB && t < s1 && n >= 0 && n < l.W) {
                const int64_t loc = l.pre + n - l.y_lo;
                if (loc >= 0 && loc < l.y_len) v = y[(int64_t)o * l.y_stride + loc];
            }
            prey[u] = v;
        }
    };
    auto stage_store = [&](float* dst) {
#pragma unroll
        for (int u = 0; u < NSL; ++u) {
            const int ms = tid + u * CXB_NT;
            if (ms < MS) {
                dst[ms] = prea[u].x;
                dst[MS + ms] = preb[u].x;
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    unsigned b1, b2, b3;
                    bf16_split3_rne(c ? preb[u].x : prea[u].x, &b1, &b2, &b3);   // (split once per symbol: rounded to nearest)
                    unsigned short* d = s_xb + c * 3 * MS + ms;
                    d[0] = (unsigned short)(b1 >> 16);
                    d[MS] = (unsigned short)(b2 >> 16);
                    d[2 * MS] = (unsigned short)(b3 >> 16);
                }
                dst[2 * MS + ms] = prea[u].y + preb[u].y;
                dst[3 * MS + ms] = prea[u].y - preb[u].y;
                dst[4 * MS + ms] = prea[u].z;
                dst[5 * MS + ms] = preb[u].z;
                dst[6 * MS + ms] = prea[u].w + preb[u].w;
            }
        }
#pragma unroll
        for (int u = 0; u < NYW; ++u) {
            const int e = tid + u * CXB_NT;
            if (e < 2 * S * TWB) dst[7 * MS + e] = prey[u];
        }
    };
    if ((int)blockIdx.x < l.ntile_cxb) { stage_load(blockIdx.x); stage_store(s_stage); }
    __syncthreads();

    for (int tile = blockIdx.x; tile < l.ntile_cxb; tile += gridDim.x) {
        const int64_t s0 = l.own_lo + (int64_t)tile * TS;
        const int64_t s1 = (s0 + TS < own_hi) ? s0 + TS : own_hi;
        const int64_t tbase = s0 - l.kmax;
        const float* s_mu = s_stage;            // [2][MS]
        const float* s_S = s_stage + 2 * MS;
        const float* s_dv = s_stage + 3 * MS;
        const float* s_ka = s_stage + 4 * MS;
        const float* s_kb = s_stage + 5 * MS;
        const float* s_gs = s_stage + 6 * MS;
        const float* s_yw = s_stage + 7 * MS;
        const bool more = tile + (int)gridDim.x < l.ntile_cxb;

        BSTAMP(0);
        for (int sub = wsub; sub < NU; sub += NWP) {
            const int tsl = sub * 32;
            const int64_t t = tbase + tsl + i, n = t * S + r;
            const bool valid = (t < s1) && n >= 0 && n < l.W;
            const bool own = valid && t >= s0;
            const float mv = valid ? 1.f : 0.f, mo = own ? 1.f : 0.f;
            float* Pu = s_P + ((size_t)(r * NU + sub) * 4) * NSIG;
            for (int e = lane; e < 4 * NSIG; e += 64) Pu[e] = 0.f;
            // The table base is re-declared opaque (a scalar register pair) in every iteration: otherwise the compiler hoists
            // the several hundred per-lane 64-bit load addresses of the unrolled chains out of the loop and spills them all.
            // The opaque copy loses the pointer's address space; it is cast back to GLOBAL explicitly: a generic (flat) load
            // also counts on the LDS counter, so waiting for an LDS operand would wait for every table load in flight and
            // the prefetch of the next block pair would never overlap the MFMA chain of this one.
            typedef const float __attribute__((address_space(1))) * gfloat_p;
            const float* Abase_generic = Abase0;
            asm volatile("" : "+s"(Abase_generic));
            const gfloat_p Abase = (gfloat_p)Abase_generic;

            // ---- pass A: per output block H X and (H.H) s, then the per-(window, tap) algebra that does not need e
            f2 zP[NB][4], zQ[NB][4], al[NB][4], be[NB][4];   // tap pairs (2 p, 2 p + 1): neighbouring accumulator registers, taps x, x + 1
            f2 a0v = 0.f, a1v = 0.f, q0v = 0.f, q1v = 0.f, sd0v = 0.f, sd1v = 0.f, vsv = 0.f;
            // the A operands of block pair p + 1 are requested before the MFMA chain of pair p is issued; sched_barrier keeps
            // the compiler from hoisting more than that one pair of loads (register budget)
            // block pairs the table fragments are requested ahead of their use: the chains wait for these loads (L2) more than for
            // anything else; two pairs ahead where the registers reach (two tap blocks: 6.08 -> 5.97 ms; three ahead: 6.10)
            constexpr int PFD = NB == 2 ? 2 : 1;
            u32x4 abuf[PFD + 1][6];    // register sets rotating by block pair: no copies; [piece][chunk] images of 16 bytes per lane
            typedef const u32x4 __attribute__((address_space(1))) * gvec_p;
#pragma unroll
            for (int q = 0; q < PFD && q < NB * NB; ++q)
#pragma unroll
                for (int s = 0; s < 6; ++s) abuf[q][s] = ((gvec_p)(Abase + (size_t)q * CXB_STEPS * 64))[s * 64 + ulane];
            // B operands: k-slot (chunk c, lane half h, element j) = (stream c, tap 8 h + j of block jb), i.e. eight consecutive bf16
            // of stream c from staged symbol tsl + i + 16 jb + 8 h on: five aligned words, an odd window shifts by one piece
            typedef const unsigned __attribute__((address_space(3))) * lword_p;
            const unsigned sh = (i & 1) * 16;
            const float* Xbase_generic = Xbase0;
            asm volatile("" : "+s"(Xbase_generic));
            const gfloat_p Xbase = (gfloat_p)Xbase_generic;
            float xbufA[NB], xbufB[NB];
            if (XT) {
#pragma unroll
                for (int xb = 0; xb < NB; ++xb) {
                    xbufA[xb] = Xbase[(xb * 2 + 0) * 64 + ulane];
                    xbufB[xb] = Xbase[(xb * 2 + 1) * 64 + ulane];
                }
            }
            // the B operand of a tap block serves every output block: with two blocks both are formed once per slab and held
            auto load_b = [&](int jb, u32x4 (&d)[3][2]) {
                lword_p xw = (lword_p)(reinterpret_cast<const unsigned*>(s_xb) + ((tsl + i + 16 * jb + 8 * h) >> 1));
                asm volatile("" : "+v"(xw));
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc) {
                        unsigned w[5];
#pragma unroll
                        for (int q = 0; q < 5; ++q) w[q] = xw[(c * 3 + pc) * (MS / 2) + q];
#pragma unroll
                        for (int q = 0; q < 4; ++q) d[pc][c][q] = __builtin_amdgcn_alignbit(w[q + 1], w[q], sh);
                    }
            };
            constexpr bool HOLDB = CXB_HOLDB && NB == 2;
            u32x4 bd[HOLDB ? NB : 1][3][2];
            if (HOLDB) {
#pragma unroll
                for (int jb = 0; jb < NB; ++jb) load_b(jb, bd[jb]);
            }
#pragma unroll
            for (int xb = 0; xb < NB; ++xb) {
                f32x16 U;
#pragma unroll
                for (int g = 0; g < 16; ++g) U[g] = 0.f;
#pragma unroll
                for (int jb = 0; jb < NB; ++jb) {
                    const int pr = xb * NB + jb;
                    __builtin_amdgcn_sched_barrier(0);
                    if (pr + PFD < NB * NB) {
                        const gvec_p A = (gvec_p)(Abase + (size_t)(pr + PFD) * CXB_STEPS * 64);
#pragma unroll
                        for (int s = 0; s < 6; ++s) abuf[(pr + PFD) % (PFD + 1)][s] = A[s * 64 + ulane];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    {
                        u32x4 dl[3][2];
                        if (!HOLDB) load_b(jb, dl);
                        const u32x4 (&d)[3][2] = HOLDB ? bd[jb] : dl;
                        // table piece x operand piece, smallest products first: (1,3) (2,2) (3,1) (1,2) (2,1) (1,1)
                        constexpr int TP6[6] = {0, 1, 2, 0, 1, 0}, DP6[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
                        for (int k = 0; k < 6; ++k)
#pragma unroll
                            for (int c = 0; c < 2; ++c)
                                U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfx8, abuf[pr % (PFD + 1)][TP6[k] * 2 + c]),
                                                                            __builtin_bit_cast(bfx8, d[DP6[k]][c]), U, 0, 0, 0);
                    }
                    // (no (H.H) s product: the term of E that holds it and its variance gradient come from k_g2_cx)
                }
                if (XT) U = MFMA32(xbufA[xb], s_mu[h * MS + tsl + i + EX], U);
                BSTAMP(1);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const int x = 16 * xb + cx_kk(2 * p, h);
                    const int slot = tsl + i + x;
                    const f2 P = f2{U[2 * p], U[2 * p + 1]}, Q = f2{U[2 * p + 8], U[2 * p + 9]};
                    const f2 ma = ld2(s_mu + slot), mb = ld2(s_mu + MS + slot), Sv = ld2(s_S + slot), dv = ld2(s_dv + slot);
                    const f2 ka = ld2(s_ka + slot), kb = ld2(s_kb + slot), gs = ld2(s_gs + slot);
                    const f2 hr = ld2(cst + x), hi = ld2(cst + LEB + x), dr = ld2(cst + 2 * LEB + x), di = ld2(cst + 3 * LEB + x);
                    const f2 zp = 2.f * P + hr, zq = 2.f * Q + hi;
                    a0v += hr * ma - hi * mb;
                    a1v += hi * ma + hr * mb;
                    q0v += ma * P - mb * Q;
                    q1v += ma * Q + mb * P;
                    sd0v += dv * dr;
                    sd1v += dv * di;
                    const f2 z2 = zp * zp + zq * zq;
                    const f2 ca = dr * zp + di * zq, cb = dr * zq - di * zp;
                    vsv += Sv * z2 + 2.f * (ka * ca + kb * cb) + gs * (dr * dr + di * di);
                    al[xb][p] = 2.f * (Sv * zp + ka * dr - kb * di);
                    be[xb][p] = 2.f * (Sv * zq + ka * di + kb * dr);
                    zP[xb][p] = zp; zQ[xb][p] = zq;
                }
                BSTAMP(2);
            }
            // the two lane halves hold disjoint taps of the window: the per-window sums are completed by one half swap each (vector
            // pipe; a shuffle would travel through the LDS pipeline and be waited for).  a0 / a1 carry the whole residual sums
            // from here on (first-order + quadratic + diagonal parts); q and sd stay zero and take the extra tap's terms.
            float a0 = cx_sum_halves(((a0v.x + a0v.y) + (q0v.x + q0v.y)) + (sd0v.x + sd0v.y));
            float a1 = cx_sum_halves(((a1v.x + a1v.y) + (q1v.x + q1v.y)) + (sd1v.x + sd1v.y));
            float vs = cx_sum_halves(vsv.x + vsv.y);
            float q0 = 0.f, q1 = 0.f, sd0 = 0.f, sd1 = 0.f;
            // extra tap e: its row of H X as a dot product over all taps (each half of the wave takes the taps it holds, the
            // halves are added), then the same per-(window, tap) algebra; both halves hold identical values from here on
            float zp_e = 0.f, zq_e = 0.f, al_e = 0.f, be_e = 0.f, z2_e = 0.f, dv_e = 0.f, ma_e = 0.f, mb_e = 0.f;
            if (XT) {
                f2 pe2 = 0.f, qe2 = 0.f;
#pragma unroll
                for (int jb = 0; jb < NB; ++jb)
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int j = 16 * jb + cx_kk(2 * p, h);
                        const f2 ma = ld2(s_mu + tsl + i + j), mb = ld2(s_mu + MS + tsl + i + j);
                        const f2 cr = ld2(xe + j), ci = ld2(xe + LEB + j);
                        pe2 += ma * cr - mb * ci;
                        qe2 += ma * ci + mb * cr;
                    }
                float pe = cx_sum_halves(pe2.x + pe2.y), qe = cx_sum_halves(qe2.x + qe2.y);
                const int slot = tsl + i + EX;
                const float ma = s_mu[slot], mb = s_mu[MS + slot], Sv = s_S[slot], dv = s_dv[slot];
                const float ka = s_ka[slot], kb = s_kb[slot], gs = s_gs[slot];
                pe += ma * xe[EX] - mb * xe[LEB + EX];
                qe += ma * xe[LEB + EX] + mb * xe[EX];
                const float hr = cst[EX], hi = cst[LEB + EX], dr = cst[2 * LEB + EX], di = cst[3 * LEB + EX];
                const float zp = fmaf(2.f, pe, hr), zq = fmaf(2.f, qe, hi);
                a0 += hr * ma - hi * mb;
                a1 += hi * ma + hr * mb;
                q0 += ma * pe - mb * qe;
                q1 += ma * qe + mb * pe;
                sd0 = fmaf(dv, dr, sd0);
                sd1 = fmaf(dv, di, sd1);
                const float z2 = zp * zp + zq * zq;
                const float ca = dr * zp + di * zq, cb = dr * zq - di * zp;
                vs += Sv * z2 + 2.f * (ka * ca + kb * cb) + gs * (dr * dr + di * di);
                al_e = 2.f * Sv * zp + 2.f * (ka * dr - kb * di);
                be_e = 2.f * Sv * zq + 2.f * (ka * di + kb * dr);
                zp_e = zp; zq_e = zq; z2_e = z2; dv_e = dv; ma_e = ma; mb_e = mb;
                gdr_e += mo * (2.f * (ka * zp + kb * zq) + 2.f * gs * dr);
                gdi_e += mo * (2.f * (ka * zq - kb * zp) + 2.f * gs * di);
            }
            const float y0 = s_yw[(0 * S + r) * TWB + tsl + i], y1 = s_yw[(1 * S + r) * TWB + tsl + i];
            float e0 = a0 + q0 + sd0 - y0, e1 = a1 + q1 + sd1 - y1;
            if (own && h == 0) Eacc += (double)(e0 * e0 + e1 * e1 + vs);
            e0 *= mv; e1 *= mv;

            // ---- extra tap e, second half: its row of H w from the registers, its anti-diagonal terms (window i -> symbol slot
            // i + e, no reduction needed), vector gradients, and rho / sigma of tap e for row e of dE/dH
            if (XT) {
                f2 hwa2 = 0.f, hwb2 = 0.f;
#pragma unroll
                for (int jb = 0; jb < NB; ++jb)
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int j = 16 * jb + cx_kk(2 * p, h);
                        const f2 cr = ld2(xe + j), ci = ld2(xe + LEB + j);
                        hwa2 += al[jb][p] * cr + be[jb][p] * ci;
                        hwb2 += be[jb][p] * cr - al[jb][p] * ci;
                    }
                float hwa = cx_sum_halves(hwa2.x + hwa2.y), hwb = cx_sum_halves(hwb2.x + hwb2.y);
                hwa += al_e * xe[EX] + be_e * xe[LEB + EX];
                hwb += be_e * xe[EX] - al_e * xe[LEB + EX];
                const float dr = cst[2 * LEB + EX], di = cst[3 * LEB + EX];
                const float ed = 2.f * (e0 * dr + e1 * di);
                const float gr = 2.f * (e0 * ma_e + e1 * mb_e), gi = 2.f * (e1 * ma_e - e0 * mb_e);
                ghr_e += mo * (gr + al_e);
                ghi_e += mo * (gi + be_e);
                gdr_e += mo * 2.f * e0 * dv_e;
                gdi_e += mo * 2.f * e1 * dv_e;
                const float Re = mo * (gr + 2.f * al_e), Se = mo * (gi + 2.f * be_e);
                gee_r += Re * ma_e + Se * mb_e;
                gee_i += Se * ma_e - Re * mb_e;
                if (h == 0) {
                    Pu[0 * NSIG + EX + i] += mv * (2.f * (e0 * zp_e + e1 * zq_e) + 2.f * hwa);
                    Pu[1 * NSIG + EX + i] += mv * (2.f * (e1 * zp_e - e0 * zq_e) + 2.f * hwb);
                    Pu[2 * NSIG + EX + i] += mv * (z2_e + ed);
                    Pu[3 * NSIG + EX + i] += mv * (z2_e - ed);
                    Tb[i * TP + 32] = Re;
                    Tb[i * TP + 33] = Se;
                }
            }

            // ---- pass B: per output block H w, the anti-diagonal sums of the moment gradients, dE/dH
            u32x4 hbuf[PFD + 1][6];
#pragma unroll
            for (int q = 0; q < PFD && q < NB * NB; ++q)
#pragma unroll
                for (int s = 0; s < 6; ++s) hbuf[q][s] = ((gvec_p)(Abase + (size_t)q * CXB_STEPS * 64))[(6 + s) * 64 + ulane];
            // B operands of H w: (alpha, beta) of the lane's eight taps of block jb in three bf16 pieces each (plain vector
            // instructions: they run beside the matrix pipe), chunk 0 = alpha, chunk 1 = beta.  Up to two blocks they are formed
            // once per slab; with three the registers do not reach and every output block forms them again.
            auto split_w = [&](int jb, u32x4 (&d)[3][2]) {
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const f2 v = c ? be[jb][p] : al[jb][p];
                        const unsigned a1 = __float_as_uint(v.x) & 0xFFFF0000u, b1 = __float_as_uint(v.y) & 0xFFFF0000u;
                        const float ra = v.x - __uint_as_float(a1), rb = v.y - __uint_as_float(b1);
                        const unsigned a2 = __float_as_uint(ra) & 0xFFFF0000u, b2 = __float_as_uint(rb) & 0xFFFF0000u;
                        const float sa = ra - __uint_as_float(a2), sb = rb - __uint_as_float(b2);
                        d[0][c][p] = __builtin_amdgcn_perm(b1, a1, 0x07060302u);
                        d[1][c][p] = __builtin_amdgcn_perm(b2, a2, 0x07060302u);
                        d[2][c][p] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
                    }
            };
            constexpr bool HOLDW = NB <= 2;
            u32x4 wd[HOLDW ? NB : 1][3][2];
            if (HOLDW) {
#pragma unroll
                for (int jb = 0; jb < NB; ++jb) split_w(jb, wd[jb]);
            }
#pragma unroll
            for (int xb = 0; xb < NB; ++xb) {
                f32x16 HW;
#pragma unroll
                for (int g = 0; g < 16; ++g) HW[g] = 0.f;
#pragma unroll
                for (int jb = 0; jb < NB; ++jb) {
                    const int pr = xb * NB + jb;
                    __builtin_amdgcn_sched_barrier(0);
                    if (pr + PFD < NB * NB) {
                        const gvec_p A = (gvec_p)(Abase + (size_t)(pr + PFD) * CXB_STEPS * 64);
#pragma unroll
                        for (int s = 0; s < 6; ++s) hbuf[(pr + PFD) % (PFD + 1)][s] = A[(6 + s) * 64 + ulane];
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    {
                        if (!HOLDW) split_w(jb, wd[0]);
                        const u32x4 (&d)[3][2] = wd[HOLDW ? jb : 0];
                        constexpr int TP6[6] = {0, 1, 2, 0, 1, 0}, DP6[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
                        for (int k = 0; k < 6; ++k)
#pragma unroll
                            for (int c = 0; c < 2; ++c)
                                HW = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfx8, hbuf[pr % (PFD + 1)][TP6[k] * 2 + c]),
                                                                             __builtin_bit_cast(bfx8, d[DP6[k]][c]), HW, 0, 0, 0);
                    }
                }
                if (XT) HW = MFMA32(xbufB[xb], h == 0 ? al_e : be_e, HW);
                BSTAMP(3);
                // anti-diagonal sums (symbol = window + tap) of the gradients of means and variances through the wave's buffer
                // (the memory is the second transposing buffer's, used after it): a (window, tap pair) writes its register pairs as
                // they are - plane A holds (d mean_a, d mean_b), plane B (d var_a, d var_b), each as (even tap, odd tap) pairs, with a
                // zero vector behind either plane that every out-of-range read is pointed at; lane m sums the eight pair rows at
                // window m - even tap (reads requested in batches), the odd taps' sums, which belong to the next symbol, move up one
                // lane at the end, and the partial of this tap block is added to the slab's record.
                // The gradients of the 3rd moments and 4th cumulants do not need the residual: k_g2_cx has written them.
                {
                    constexpr int ZV = 8 * 32, PB = ZV + 1;
                    f32x4* Bd = reinterpret_cast<f32x4*>(Tv);
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const int xl = cx_kk(2 * p, h), x = 16 * xb + xl, pr = xl >> 1;
                        const f2 zp = zP[xb][p], zq = zQ[xb][p];
                        const f2 dr = ld2(cst + 2 * LEB + x), di = ld2(cst + 3 * LEB + x);
                        const f2 ed = 2.f * (e0 * dr + e1 * di), zzp = zp * zp + zq * zq;
                        const f2 d0 = (2.f * mv) * (e0 * zp + e1 * zq + f2{HW[2 * p], HW[2 * p + 1]});
                        const f2 d1 = (2.f * mv) * (e1 * zp - e0 * zq + f2{HW[2 * p + 8], HW[2 * p + 9]});
                        const f2 d2 = mv * (zzp + ed), d3 = mv * (zzp - ed);
                        Bd[pr * 32 + i] = f32x4{d0.x, d0.y, d1.x, d1.y};
                        Bd[PB + pr * 32 + i] = f32x4{d2.x, d2.y, d3.x, d3.y};
                    }
                    if (lane < 2) Bd[ZV + lane * PB] = f32x4{0.f, 0.f, 0.f, 0.f};
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_sched_barrier(0);
                    f32x4 ra[8], rb[8];
#pragma unroll
                    for (int pr = 0; pr < 8; ++pr) {
                        const int idx = (unsigned)(lane - 2 * pr) < 32u ? pr * 30 + lane : ZV;
                        ra[pr] = Bd[idx];
                        rb[pr] = Bd[idx + PB];
                    }
                    f32x4 sa = ra[0], sb = rb[0];
#pragma unroll
                    for (int pr = 1; pr < 8; ++pr) { sa += ra[pr]; sb += rb[pr]; }
                    const float g0 = sa[0] + cx_shr1(sa[1]), g1 = sa[2] + cx_shr1(sa[3]);
                    const float g2 = sb[0] + cx_shr1(sb[1]), g3 = sb[2] + cx_shr1(sb[3]);
                    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
                    for (int k = 0; k < 4; ++k) {                         // a pair summed, the next pair requested
                        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x002, 24, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    if (lane < 48) {
                        Pu[0 * NSIG + 16 * xb + lane] += g0;
                        Pu[1 * NSIG + 16 * xb + lane] += g1;
                        Pu[2 * NSIG + 16 * xb + lane] += g2;
                        Pu[3 * NSIG + 16 * xb + lane] += g3;
                    }
                    __builtin_amdgcn_wave_barrier();
                }
                BSTAMP(4);
                // rho / sigma of this block (transposed through LDS), vector gradients
                // (two tap pairs at a time, all their reads before their first write: the compiler cannot tell the LDS regions
                // apart, so a read placed after a write is not moved above it and each pair would wait for its own reads)
#pragma unroll
                for (int pp = 0; pp < 4; pp += 2) {
                    f2 ma[2], mb[2], dvq[2], ka[2], kb[2], gs[2], dr[2], di[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int xl = cx_kk(2 * (pp + u), h);
                        const int slot = tsl + i + 16 * xb + xl;
                        ma[u] = ld2(s_mu + slot); mb[u] = ld2(s_mu + MS + slot); dvq[u] = ld2(s_dv + slot);
                        ka[u] = ld2(s_ka + slot); kb[u] = ld2(s_kb + slot); gs[u] = ld2(s_gs + slot);
                        dr[u] = ld2(cst + 2 * LEB + 16 * xb + xl); di[u] = ld2(cst + 3 * LEB + 16 * xb + xl);
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const int p = pp + u;
                        const int xl = cx_kk(2 * p, h);
                        const f2 zp = zP[xb][p], zq = zQ[xb][p];
                        const f2 gr = 2.f * (e0 * ma[u] + e1 * mb[u]), gi = 2.f * (e1 * ma[u] - e0 * mb[u]);
                        const f2 v0 = mo * (gr + al[xb][p]), v1 = mo * (gi + be[xb][p]);
                        const f2 v2 = (2.f * mo) * (ka[u] * zp + kb[u] * zq + gs[u] * dr[u] + e0 * dvq[u]);
                        const f2 v3 = (2.f * mo) * (ka[u] * zq - kb[u] * zp + gs[u] * di[u] + e1 * dvq[u]);
                        const f2 t0 = mo * (gr + 2.f * al[xb][p]), t1 = mo * (gi + 2.f * be[xb][p]);
                        Tb[i * TP + xl] = t0.x;
                        Tb[i * TP + xl + 1] = t0.y;
                        Tb[i * TP + 16 + xl] = t1.x;
                        Tb[i * TP + 16 + xl + 1] = t1.y;
                        float* tv = Tv + i * TVP + xl;
                        tv[0] = v0.x; tv[1] = v0.y;
                        tv[16] = v1.x; tv[17] = v1.y;
                        tv[32] = v2.x; tv[33] = v2.y;
                        tv[48] = v3.x; tv[49] = v3.y;
                    }
                }
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_sched_barrier(0);
                {
                    const int xa = lane & 15, kq = lane >> 4;
                    // this lane's eight windows of the slab: 4 st + kq on the f32 matrix instructions (four windows per k-step),
                    // 8 kq + st on the bf16 ones (32 windows per instruction, eight consecutive ones per lane group)
                    auto win = [&](int st) { return BFK ? 8 * kq + st : 4 * st + kq; };
                    float ar[8], as[8];
#pragma unroll
                    for (int st = 0; st < 8; ++st) {
                        ar[st] = Tb[win(st) * TP + xa];
                        as[st] = Tb[win(st) * TP + 16 + xa];
                    }
#pragma unroll
                    for (int st = 0; st < 8; ++st) {
                        const float* tv = Tv + win(st) * TVP + xa;
                        vacc[xb][0] += tv[0];
                        vacc[xb][1] += tv[16];
                        vacc[xb][2] += tv[32];
                        vacc[xb][3] += tv[48];
                    }
                    if (XT) {   // column e of dE/dH: this lane's share of sum_w (rho, sigma)(w, x) mu(w + e)
#pragma unroll
                        for (int st = 0; st < 8; ++st) {
                            const int w = tsl + win(st) + EX;
                            const f2 me = {s_mu[w], s_mu[MS + w]};
                            col[xb] += ar[st] * f2{me.x, -me.y} + as[st] * f2{me.y, me.x};
                        }
                    }
                    if (BFK) {
                        u32x4 rho[3], sig[3], nrho[3];
#pragma unroll
                        for (int k2 = 0; k2 < 4; ++k2) {
                            unsigned pr[2][3], ps[2][3];
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                bf16_split3_trunc(ar[2 * k2 + u], &pr[u][0], &pr[u][1], &pr[u][2]);
                                bf16_split3_trunc(as[2 * k2 + u], &ps[u][0], &ps[u][1], &ps[u][2]);
                            }
#pragma unroll
                            for (int pc = 0; pc < 3; ++pc) {
                                rho[pc][k2] = __builtin_amdgcn_perm(pr[1][pc], pr[0][pc], 0x07060302u);
                                sig[pc][k2] = __builtin_amdgcn_perm(ps[1][pc], ps[0][pc], 0x07060302u);
                                nrho[pc][k2] = rho[pc][k2] ^ 0x80008000u;
                            }
                        }
#pragma unroll
                        for (int jb = 0; jb < NB; ++jb) {
                            // B operands: the means of windows 8 kq .. 8 kq + 7 at tap 16 jb + xa: eight consecutive bf16 of either
                            // stream from staged symbol tsl + 8 kq + 16 jb + xa on (five aligned words; an odd tap shifts by one piece)
                            lword_p xw = (lword_p)(reinterpret_cast<const unsigned*>(s_xb) + ((tsl + 8 * kq + 16 * jb + xa) >> 1));
                            asm volatile("" : "+v"(xw));
                            const unsigned shk = (xa & 1) * 16;
#pragma unroll
                            for (int dp = 2; dp >= 0; --dp) {   // the small pieces first
                                u32x4 ya, yb;
                                unsigned wa[5], wb[5];
#pragma unroll
                                for (int q = 0; q < 5; ++q) { wa[q] = xw[dp * (MS / 2) + q]; wb[q] = xw[(3 + dp) * (MS / 2) + q]; }
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    ya[q] = __builtin_amdgcn_alignbit(wa[q + 1], wa[q], shk);
                                    yb[q] = __builtin_amdgcn_alignbit(wb[q + 1], wb[q], shk);
                                }
#pragma unroll
                                for (int tp = 2 - dp; tp >= 0; --tp) {
                                    K1[xb][jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfx8, rho[tp]), __builtin_bit_cast(bfx8, ya), K1[xb][jb], 0, 0, 0);
                                    K2[xb][jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfx8, sig[tp]), __builtin_bit_cast(bfx8, ya), K2[xb][jb], 0, 0, 0);
                                    K1[xb][jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfx8, sig[tp]), __builtin_bit_cast(bfx8, yb), K1[xb][jb], 0, 0, 0);
                                    K2[xb][jb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfx8, nrho[tp]), __builtin_bit_cast(bfx8, yb), K2[xb][jb], 0, 0, 0);
                                }
                            }
                            if (XT && xb == 0) {   // row e of dE/dH, once per slab: sum_w (rho, sigma)(w, e) mu(w + j)
#pragma unroll
                                for (int st = 0; st < 8; ++st) {
                                    const int w = tsl + win(st) + 16 * jb + xa;
                                    const float ma = s_mu[w], mb = s_mu[MS + w];
                                    const f2 rs = ld2(Tb + win(st) * TP + 32);   // (rho, sigma) of tap e at this window
                                    row[jb] += ma * rs + mb * f2{rs.y, -rs.x};
                                }
                            }
                        }
                    } else {
#pragma unroll
                    for (int jb = 0; jb < NB; ++jb) {
        #pragma unroll
                        for (int st = 0; st < 8; ++st) {
                            const int w = tsl + 4 * st + kq + 16 * jb + xa;
                            const float ma = s_mu[w], mb = s_mu[MS + w];
                            K1[xb][jb] = MFMA16(ar[st] + as[st], ma, K1[xb][jb]);
                            K2[xb][jb] = MFMA16(-ar[st], ma + mb, K2[xb][jb]);
                            K3[xb][jb] = MFMA16(as[st], ma - mb, K3[xb][jb]);
                            if (XT && xb == 0) {   // row e of dE/dH, once per slab: sum_w (rho, sigma)(w, e) mu(w + j)
                                const f2 rs = ld2(Tb + (4 * st + kq) * TP + 32);   // (rho, sigma) of tap e at this window
                                row[jb] += ma * rs + mb * f2{rs.y, -rs.x};
                            }
                        }
                    }
                    // with one wave per SIMD every operand read waited for where it is used is idle time: the first reads are
                    // requested before the chain starts, the others while it runs
                    __builtin_amdgcn_sched_group_barrier(0x100, 16, 0);
#pragma unroll
                    for (int k = 0; k < 8 * NB; ++k) {
                        __builtin_amdgcn_sched_group_barrier(0x002, 8, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
                    }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_wave_barrier();
                BSTAMP(5);
            }
        }
        __syncthreads();
        BSTAMP(6);
        // the next tile is fetched here, not during the slabs: the 36 registers it would occupy across them are worth more than the
        // load latency exposed once per tile (four slabs per wave)
        if (more) { stage_load(tile + gridDim.x); stage_store(s_stage); }
        // ---- gather: symbol slot ms = kmax + ml collects the partials of every slab whose 32 + Le - 1 slots cover it
        for (int idx = tid; idx < 2 * TS; idx += CXB_NT) {
            const int c = idx / TS, ml = idx - c * TS;
            const int64_t m = s0 + ml;
            if (m < s1) {
                const int ms = ml + l.kmax;
                float g0 = 0.f, g1 = 0.f;
                for (int rr = 0; rr < S; ++rr) {
                    for (int d = (NSIG - 1) / 32; d >= 0; --d) {
                        const int sub = (ms >> 5) - d, sg = (ms & 31) + 32 * d;
                        if (sub >= 0 && sub < NU && sg < NSIG) {
                            const float* P = s_P + ((size_t)(rr * NU + sub) * 4) * NSIG + sg;
                            g0 += P[(0 + c) * NSIG]; g1 += P[(2 + c) * NSIG];
                        }
                    }
                }
                float* gp = gmom + (int64_t)c * 4 * l.sym_cnt + (m - l.sym_lo);
                gp[0] = g0; gp[l.sym_cnt] = g1 + ws[l.o_gT + (m - l.sym_lo)];   // + the part k_g2_cx formed; planes 2 and 3: k_g2_cx
            }
        }
        __syncthreads();
        BSTAMP(7);
    }
#ifdef V2VAE_STAMPS
    if (lane == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(ws + l.o_sumstage);  // free until the record sums run
        for (int k = 0; k < 8; ++k) dbg[((size_t)blockIdx.x * NW + wv) * 8 + k] = st_acc[k];
    }
#endif

    // ---- one record per (block, phase): the waves of a phase add their tiles in a fixed order
    for (int idx = tid; idx < S * REC; idx += CXB_NT) s_rec[idx] = 0.f;   // (the last tile ended with a workgroup barrier)
    __syncthreads();
    float* img = s_rec + (size_t)r * REC;
    for (int w = 0; w < NWP; ++w) {
        if (wsub == w) {
#pragma unroll
            for (int xb = 0; xb < NB; ++xb) {
#pragma unroll
                for (int jb = 0; jb < NB; ++jb)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int a = 16 * xb + 4 * (lane >> 4) + g, b = 16 * jb + (lane & 15);
                        img[a * LEB + b] += BFK ? K1[xb][jb][g] : K1[xb][jb][g] - K3[xb][jb][g];
                        img[(LEB + a) * LEB + b] += BFK ? K2[xb][jb][g] : K1[xb][jb][g] + K2[xb][jb][g];
                    }
                {
                    float v0 = vacc[xb][0], v1 = vacc[xb][1], v2 = vacc[xb][2], v3 = vacc[xb][3];
#pragma unroll
                    for (int off = 16; off < 64; off <<= 1) {   // over the four window groups
                        v0 += __shfl_xor(v0, off); v1 += __shfl_xor(v1, off);
                        v2 += __shfl_xor(v2, off); v3 += __shfl_xor(v3, off);
                    }
                    if (lane < 16) {
                        const int x = 16 * xb + lane;
                        img[2 * LEB * LEB + 0 * LEB + x] += v0;
                        img[2 * LEB * LEB + 1 * LEB + x] += v1;
                        img[2 * LEB * LEB + 2 * LEB + x] += v2;
                        img[2 * LEB * LEB + 3 * LEB + x] += v3;
                    }
                }
            }
            if (XT) {
#pragma unroll
                for (int xb = 0; xb < NB; ++xb) {
                    float c0 = col[xb].x, c1 = col[xb].y, r0 = row[xb].x, r1 = row[xb].y;
                    c0 += __shfl_xor(c0, 16); c1 += __shfl_xor(c1, 16); r0 += __shfl_xor(r0, 16); r1 += __shfl_xor(r1, 16);
                    c0 += __shfl_xor(c0, 32); c1 += __shfl_xor(c1, 32); r0 += __shfl_xor(r0, 32); r1 += __shfl_xor(r1, 32);
                    if (lane < 16) {
                        const int x = 16 * xb + lane;
                        img[x * LEB + EX] += c0;
                        img[(LEB + x) * LEB + EX] += c1;
                        img[EX * LEB + x] += r0;
                        img[(LEB + EX) * LEB + x] += r1;
                    }
                }
                float v[6] = {ghr_e, ghi_e, gdr_e, gdi_e, gee_r, gee_i};
#pragma unroll
                for (int k = 0; k < 6; ++k)
#pragma unroll
                    for (int off = 1; off < 32; off <<= 1) v[k] += __shfl_xor(v[k], off);
                if (lane == 0) {
#pragma unroll
                    for (int k = 0; k < 4; ++k) img[2 * LEB * LEB + k * LEB + EX] += v[k];
                    img[EX * LEB + EX] += v[4];
                    img[(LEB + EX) * LEB + EX] += v[5];
                }
            }
        }
        __syncthreads();
    }
    float* rec = ws + l.o_part_cxb + (size_t)blockIdx.x * S * REC;
    for (int e = tid; e < S * REC; e += CXB_NT) rec[e] = s_rec[e];
    double v = Eacc;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) dred[wv] = v;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < NW; ++w) s += dred[w];
        reinterpret_cast<double*>(ws + l.o_Epart)[blockIdx.x] = s;
    }
}

// raw gradients of the complex channel kernels from the per-phase totals
__global__ __launch_bounds__(NT) void k_chain_channel_cxb(Lay l, const float* __restrict__ ws, const double* __restrict__ scal,
                                                          const float* __restrict__ grad_scale, float* __restrict__ g_h1,
                                                          float* __restrict__ g_h2) {
    const float lam = (float)scal[V2VAE_SCAL_LAMBDA] * (grad_scale ? grad_scale[0] : 1.f);
    const int L = l.L, n1 = 2 * L, n2 = 2 * L * L, LEB = l.cxb_leb;
    auto compact = [&](int tap, int* r, int* x) {
        int rr = (tap - (L - 1)) % l.S;
        if (rr < 0) rr += l.S;
        *r = rr;
        *x = (rr + L - 1 - tap) / l.S;
    };
    for (int idx = blockIdx.x * NT + threadIdx.x; idx < n1 + n2; idx += gridDim.x * NT) {
        if (idx < n1) {
            if (!g_h1) continue;
            const int p = idx / L, tap = idx - p * L;
            int r, x;
            compact(tap, &r, &x);
            g_h1[idx] = lam * ws[l.o_tot_cxb + (size_t)r * l.cxb_rec + 2 * LEB * LEB + p * LEB + x];
        } else {
            if (!g_h2) continue;
            const int e = idx - n1;
            const int p = e / (L * L), rem = e - p * L * L;
            const int ti = rem / L, tj = rem - ti * L;
            int ri, xi, rj, xj;
            compact(ti, &ri, &xi);
            compact(tj, &rj, &xj);
            float v = 0.f;
            if (ri == rj) {
                const float* tot = ws + l.o_tot_cxb + (size_t)ri * l.cxb_rec;
                const float* Hp = ws + l.o_cxbH + (size_t)(ri * 2 + p) * LEB * LEB;
                const float ps = ws[l.o_psv + ((size_t)ri * l.LE + xi) * l.LE + xj];   // lagged sum of the tap pair (k_g2_cx)
                auto G = [&](int a, int b) {
                    float g = tot[(p * LEB + a) * LEB + b] + 4.f * Hp[a * LEB + b] * ps;
                    if (a == b) g += tot[2 * LEB * LEB + (2 + p) * LEB + a];
                    return g;
                };
                v = G(xi, xj) + G(xj, xi);
            }
            g_h2[e] = lam * v;
        }
    }
}

void launch_prep_cxb(const Lay& l, const float* h1, const float* h2, float* ws, hipStream_t st) {
    hipLaunchKernelGGL(k_prep_cxb, dim3(64), dim3(NT), 0, st, l, h1, h2, ws);
}

int launch_elbo_cxb(const Lay& l, const float* y, float* ws, hipStream_t st) {
    const int lds = l.lds_cxb;
#define V2VAE_LAUNCH_CXB(SS, NBB, XTT)                                                                                        \
    do {                                                                                                                   \
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_elbo_cxb<SS, NBB, XTT>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                lds) != hipSuccess)                                                                        \
            return V2VAE_ELAUNCH;                                                                                          \
        hipLaunchKernelGGL((k_elbo_cxb<SS, NBB, XTT>), dim3(l.nblk_cxb), dim3(cxb_nw(NBB, XTT) * 64), lds, st, l, y, ws);                  \
    } while (0)
    if (l.cxb_xt) {
        if (l.cxb_nbk == 1) { if (l.S == 1) V2VAE_LAUNCH_CXB(1, 1, 1); else V2VAE_LAUNCH_CXB(2, 1, 1); }
        else { if (l.S == 1) V2VAE_LAUNCH_CXB(1, 2, 1); else V2VAE_LAUNCH_CXB(2, 2, 1); }
    } else if (l.cxb_nb == 2) { if (l.S == 1) V2VAE_LAUNCH_CXB(1, 2, 0); else V2VAE_LAUNCH_CXB(2, 2, 0); }
    else { if (l.S == 1) V2VAE_LAUNCH_CXB(1, 3, 0); else V2VAE_LAUNCH_CXB(2, 3, 0); }
#undef V2VAE_LAUNCH_CXB
    return V2VAE_OK;
}

void launch_chain_channel_cxb(const Lay& l, float* ws, const double* scal, const float* grad_scale, float* g_h1, float* g_h2,
                              hipStream_t st) {
    const int n = 2 * l.L + 2 * l.L * l.L;
    hipLaunchKernelGGL(k_chain_channel_cxb, dim3((n + NT - 1) / NT), dim3(NT), 0, st, l, (const float*)ws, scal, grad_scale, g_h1,
                       g_h2);
}

}  // namespace v2vae
