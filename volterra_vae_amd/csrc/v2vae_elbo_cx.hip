// Fused ELBO forward + backward for the complex (I/Q) V2VAE on the f32 matrix cores.
//
// Structure exploited (holomorphic second-order kernel H = Hr + j Hi, symmetric, per phase compact,
// Le <= 16 active taps per stream): with X = mu_a + j mu_b the two real outputs share one complex
// mat-vec, so per window the O(Le^2) work is four real products on 32-wide tiles
//     [P | Q]   = [mu_a | mu_b] . [[Hr, Hi], [-Hi, Hr]]          (H X)
//     T         = (v_a + v_b) . (Hr.Hr + Hi.Hi)                    (sum over outputs of (H.H) v)
//     [HWa|HWb] = [alpha | beta] . [[Hr, -Hi], [Hi, Hr]]           (sum over outputs of H w)
//     R        += [rho | sigma]^T [mu_a | mu_b]                    (dE/dH accumulation, over windows)
// plus Ps += s s^T (s = v_a + v_b).  Each is a chain of v_mfma_f32_32x32x2_f32.
//
// Orientation: windows sit on the lanes (column index of the MFMA result), taps in the accumulator
// registers.  Lane (i = lane & 31, h = lane >> 5) of a wave then owns window i and the eight taps
// kk(q, h) = (q & 3) + 8 (q >> 2) + 4 h, q = 0..7, for BOTH streams (registers q and q + 8 of the
// result), so the whole per-(window, tap) algebra is lane-local and per-window sums need one exchange
// with lane ^ 32.  The H w product takes its B operand straight from those registers (k-slots are
// permuted to match); only the dE/dH accumulation, which sums over windows (= lanes), goes through a
// per-wave transposing LDS buffer.
//
// A block tile is 256 windows per phase (TS = 256 - kmax owned symbols plus the kmax halo windows recomputed on the
// left), so every per-symbol gradient is completed inside the block and gathered without atomics.
#include <hip/hip_runtime.h>

#include "v2vae_bf16_split.h"
#include "v2vae_cx_common.h"
#include "v2vae_chain_cx.h"
#include "v2vae_layout.h"

namespace v2vae {


// ------------------------------------------------------------------------------------------------------------
// 8 waves per workgroup (two per SIMD, so one wave's per-tap algebra overlaps the other's MFMA
// chain), constant A operands in LDS, block tiles of 256 windows per phase, and an LDS-lean gather: each wave sums
// its own 32 x Le tile of per-(window, tap) gradients along the anti-diagonals (symbol = window + tap) into a
// 47-entry partial per (phase, subtile); the block then adds the <= 2 x S partials that cover a symbol in a fixed order.
// The s s^T accumulation is not done here at all: it is a sum of lagged products along the symbol axis
// (k_psum_cx + psum_value below), O(Le) instead of O(Le^2) per window.
// ------------------------------------------------------------------------------------------------------------
// Which (stream, tap) of the H w operand sits on k-slot (step s, lane half h) of the run kernel.  NK = 16: the accumulator
// layout as it is (steps 0-7 alpha, 8-15 beta, tap cx_kk(s & 7, h)).  NK = 13 (taps 13-15 do not exist): steps 0-4 alpha and
// 5-9 beta of the slots q = s % 5 (taps 0-3, 8 on the lower half, 4-7, 12 on the upper), steps 10-12: lower half alpha, upper half
// beta, both of tap 9 + (s - 10), a lower-half tap.
template <int NK>
__host__ __device__ __forceinline__ void cx3_hw_slot(int s, int h, int* stream, int* tap) {
    if (NK == 16) { *stream = s >> 3; *tap = ((s & 7) & 3) + 8 * ((s & 7) >> 2) + 4 * h; }
    else if (s < 10) { const int q = s % 5; *stream = s / 5; *tap = (q & 3) + 8 * (q >> 2) + 4 * h; }
    else { *stream = h; *tap = 9 + (s - 10); }
}

constexpr int CX2_TWB = 256;
constexpr int CX2_MS = CX2_TWB + 48;
constexpr int CX2_NU = CX2_TWB / 32;
constexpr int CX2_NSIG = 64;  // anti-diagonal entries kept per unit (47 used; 64 keeps the store branch-free)

template <int S, int NW>
__global__ __launch_bounds__(NW * 64) void k_elbo_cx2(Lay l, const float* __restrict__ y, float* __restrict__ ws) {
    extern __shared__ __align__(16) float sm[];
    constexpr int TWB = CX2_TWB, MS = CX2_MS, TP = CX_TP, NU = CX2_NU, NSIG = CX2_NSIG;
    constexpr int NWP = NW / S;
    constexpr int CX2_NT = NW * 64;
    static_assert(CX2_TWB <= NW * 64, "the gather takes one thread per symbol of a tile");
    // one staging buffer of per-symbol planes [MS] each (unit stride: MFMA operands, and pairs of neighbouring taps load
    // as register pairs for packed arithmetic): mean_a, mean_b, v_a + v_b, v_a - v_b, 3rd central moments k_a, k_b,
    // sum of the 4th cumulants g_a + g_b; then y' [2][S][TWB]
    constexpr int STG = 7 * MS + 2 * S * TWB;
    float* s_stage = sm;                      // [STG]; the next tile travels in registers until this one is consumed
    float* s_A = s_stage + STG;           // [S][40][64]  A operands: steps 0-15 H X, 16-23 (H.H) s, 24-39 H w
    float* s_c = s_A + S * 40 * 64;           // [S][4][16]   h1r, h1i, diag Hr, diag Hi
    float* s_T = s_c + S * 64;                // [8 waves][32 * TP]  transposing buffer of rho / sigma
    // anti-diagonal buffer, one per wave, of 4-vectors (the four gradient quantities of a (window, tap) pair travel
    // together): 16 rows (taps) of 32 windows, then one zero vector that every out-of-range read is pointed at
    constexpr int ADB = 16 * 32 + 4;           // vectors per wave
    float* s_B = s_T + NW * 32 * TP;           // [NW waves][ADB][4]
    float* s_P = s_B + NW * 4 * ADB;           // [S][NU][NSIG][4]
    __shared__ double dred[CX2_NT / 64];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int r = wv / NWP, wsub = wv % NWP;
    float* Tb = s_T + wv * 32 * TP;
    f32x4* Bd = reinterpret_cast<f32x4*>(s_B) + wv * ADB;
    const float* __restrict__ mom = ws + l.o_mom;
    float* __restrict__ gmom = ws + l.o_gmom;
    const int64_t own_hi = l.own_lo + l.own_cnt, sym_hi = l.sym_lo + l.sym_cnt;
    const int TS = TWB - l.kmax;
    const int cj = i >> 4, yj = i & 15;

    // ---- constants of every phase into LDS
    for (int idx = tid; idx < S * 40 * 64; idx += CX2_NT) {
        const int ln = idx & 63, s = (idx >> 6) % 40, rr = idx / (40 * 64);
        const int jj = ln & 31, hh = ln >> 5, cjj = jj >> 4, yjj = jj & 15;
        const float* Hr = ws + l.o_cxH + (size_t)(rr * 2 + 0) * 256;
        const float* Hi = ws + l.o_cxH + (size_t)(rr * 2 + 1) * 256;
        float v;
        if (s < 16) {          // [[Hr, Hi], [-Hi, Hr]], k = 2 s + h
            const int k = 2 * s + hh, ck = k >> 4, xk = k & 15;
            const float a = Hr[xk * 16 + yjj], b = Hi[xk * 16 + yjj];
            v = ck == 0 ? (cjj == 0 ? a : b) : (cjj == 0 ? -b : a);
        } else if (s < 24) {   // |H|^2 rows 0..15, k = 2 (s - 16) + h
            const int x = 2 * (s - 16) + hh;
            const float a = Hr[yjj * 16 + x], b = Hi[yjj * 16 + x];
            v = cjj == 0 ? a * a + b * b : 0.f;
        } else {               // [[Hr, -Hi], [Hi, Hr]], k-slot <-> row ((s-24) >> 3, kk((s-24) & 7, h))
            const int s2 = s - 24, ck = s2 >> 3, xk = cx_kk(s2 & 7, hh);
            const float a = Hr[xk * 16 + yjj], b = Hi[xk * 16 + yjj];
            v = ck == 0 ? (cjj == 0 ? a : -b) : (cjj == 0 ? b : a);
        }
        s_A[idx] = v;
    }
    for (int idx = tid; idx < S * 64; idx += CX2_NT) {
        const int rr = idx / 64, q = (idx >> 4) & 3, x = idx & 15;
        const float* Hr = ws + l.o_cxH + (size_t)(rr * 2 + 0) * 256;
        const float* Hi = ws + l.o_cxH + (size_t)(rr * 2 + 1) * 256;
        s_c[idx] = q == 0 ? ws[l.o_cxh1 + (rr * 2 + 0) * 16 + x] : q == 1 ? ws[l.o_cxh1 + (rr * 2 + 1) * 16 + x]
                 : q == 2 ? Hr[x * 16 + x] : Hi[x * 16 + x];
    }
    for (int idx = tid; idx < NW * 4 * ADB; idx += CX2_NT) s_B[idx] = 0.f;
    const float* Aop = s_A + (size_t)r * 40 * 64 + lane;
    // per-tap constants stay in LDS (all lanes of a half read one address: a broadcast, no bank conflict)
    const float* cst = s_c + r * 64 + 4 * h;
#define C_HR(q) cst[0 * 16 + ((q) & 3) + 8 * ((q) >> 2)]
#define C_HI(q) cst[1 * 16 + ((q) & 3) + 8 * ((q) >> 2)]
#define C_DR(q) cst[2 * 16 + ((q) & 3) + 8 * ((q) >> 2)]
#define C_DI(q) cst[3 * 16 + ((q) & 3) + 8 * ((q) >> 2)]
    // the same for the tap pair (2 p, 2 p + 1): neighbouring taps, neighbouring addresses
#define C2(row, p) ld2(cst + (row) * 16 + ((2 * (p)) & 3) + 8 * ((2 * (p)) >> 2))
    __syncthreads();

#ifdef V2VAE_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#define STAMP(k)                                                        \
    do {                                                                \
        __builtin_amdgcn_sched_barrier(0);                              \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();   \
        __builtin_amdgcn_s_waitcnt(0xC07F);                             \
        st_acc[k] += now_ - st_last;                                    \
        st_last = now_;                                                 \
        __builtin_amdgcn_sched_barrier(0);                              \
    } while (0)
#else
#define STAMP(k)
#endif
    f32x16 R;
#pragma unroll
    for (int g = 0; g < 16; ++g) R[g] = 0.f;
    f2 ghr[4], ghi[4], gdr[4], gdi[4];  // vector gradients of the tap pairs (2 p, 2 p + 1)
#pragma unroll
    for (int p = 0; p < 4; ++p) { ghr[p] = 0.f; ghi[p] = 0.f; gdr[p] = 0.f; gdi[p] = 0.f; }
    double Eacc = 0.0;

    // staging of one block tile through registers, so that the loads of tile n + 1 are in flight while tile n is
    // being computed: thread t carries symbol slots t, t + NT, ... (all eight moments of a slot) and y' slots likewise
    constexpr int NSL = (MS + CX2_NT - 1) / CX2_NT;
    constexpr int NYW = (2 * S * TWB + CX2_NT - 1) / CX2_NT;
    float4 prea[NSL], preb[NSL];
    float prey[NYW];
    auto stage_load = [&](int tile) {
        const int64_t s0 = l.own_lo + (int64_t)tile * TS;
        const int64_t s1 = (s0 + TS < own_hi) ? s0 + TS : own_hi;
        const int64_t tbase = s0 - l.kmax;
#pragma unroll
        for (int u = 0; u < NSL; ++u) {
            const int ms = tid + u * CX2_NT;
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
            const int e = tid + u * CX2_NT;
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
            const int ms = tid + u * CX2_NT;
            if (ms < MS) {
                dst[0 * MS + ms] = prea[u].x;
                dst[1 * MS + ms] = preb[u].x;
                dst[2 * MS + ms] = prea[u].y + preb[u].y;
                dst[3 * MS + ms] = prea[u].y - preb[u].y;
                dst[4 * MS + ms] = prea[u].z;
                dst[5 * MS + ms] = preb[u].z;
                dst[6 * MS + ms] = prea[u].w + preb[u].w;
            }
        }
#pragma unroll
        for (int u = 0; u < NYW; ++u) {
            const int e = tid + u * CX2_NT;
            if (e < 2 * S * TWB) dst[7 * MS + e] = prey[u];
        }
    };
    if (blockIdx.x < l.ntile_cx) { stage_load(blockIdx.x); stage_store(s_stage); }
    __syncthreads();

    for (int tile = blockIdx.x; tile < l.ntile_cx; tile += gridDim.x) {
        const int64_t s0 = l.own_lo + (int64_t)tile * TS;
        const int64_t s1 = (s0 + TS < own_hi) ? s0 + TS : own_hi;
        const int64_t tbase = s0 - l.kmax;
        const float* s_mu = s_stage;                                      // [2][MS] means
        const float* s_S = s_stage + 2 * MS;                              // v_a + v_b
        const float* s_dv = s_stage + 3 * MS;                             // v_a - v_b
        const float* s_ka = s_stage + 4 * MS;
        const float* s_kb = s_stage + 5 * MS;
        const float* s_gs = s_stage + 6 * MS;                             // g_a + g_b
        const float* s_yw = s_stage + 7 * MS;
        const bool more = tile + (int)gridDim.x < l.ntile_cx;
        if (more) stage_load(tile + gridDim.x);
        // the part of the variance gradient that k_g2_cx formed, for the symbol this thread gathers at the end of the tile:
        // requested now, so that its latency is not paid between the two barriers of the gather
        float gT_pre = 0.f;
        if (tid < TS && s0 + tid < s1) gT_pre = ws[l.o_gT + (s0 + tid - l.sym_lo)];
        STAMP(0);

        for (int sub = wsub; sub < NU; sub += NWP) {
            const int tsl = sub * 32;
            const int64_t t = tbase + tsl + i, n = t * S + r;
            const bool valid = (t < s1) && n >= 0 && n < l.W;
            const bool own = valid && t >= s0;
            const float mv = valid ? 1.f : 0.f, mo = own ? 1.f : 0.f;

            f32x16 U;
#pragma unroll
            for (int g = 0; g < 16; ++g) U[g] = 0.f;
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int k = 2 * s + h;
                const float b = s_mu[(k >> 4) * MS + tsl + i + (k & 15)];
#ifdef ABLATE_U
                U[s] += Aop[s * 64] * b;
#else
                U = MFMA32(Aop[s * 64], b, U);
#endif
            }
            // (no (H.H) s product: the term of E that holds it and its variance gradient come from k_g2_cx)

            STAMP(1);
            // per-tap algebra on tap PAIRS (2 p, 2 p + 1): their accumulator rows, moments and constants sit in neighbouring
            // registers / addresses, so every operation below is one packed instruction for two taps
            f2 zP[4], zQ[4], al[4], be[4], zz[4], dvv[4];
            f2 a0v = 0.f, a1v = 0.f, q0v = 0.f, q1v = 0.f, sd0v = 0.f, sd1v = 0.f, vsv = 0.f;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int slot = tsl + i + cx_kk(2 * p, h);
                const f2 P = f2{U[2 * p], U[2 * p + 1]}, Q = f2{U[2 * p + 8], U[2 * p + 9]};
                const f2 ma = ld2(s_mu + slot), mb = ld2(s_mu + MS + slot), Sv = ld2(s_S + slot), dv = ld2(s_dv + slot);
                const f2 ka = ld2(s_ka + slot), kb = ld2(s_kb + slot), gs = ld2(s_gs + slot);
                const f2 hr = C2(0, p), hi = C2(1, p), dr = C2(2, p), di = C2(3, p);
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
                al[p] = 2.f * (Sv * zp + ka * dr - kb * di);
                be[p] = 2.f * (Sv * zq + ka * di + kb * dr);
                zP[p] = zp; zQ[p] = zq;
                zz[p] = z2;
                dvv[p] = dv;
                gdr[p] += (2.f * mo) * (ka * zp + kb * zq + gs * dr);
                gdi[p] += (2.f * mo) * (ka * zq - kb * zp + gs * di);
            }
            float a0 = a0v.x + a0v.y, a1 = a1v.x + a1v.y, q0 = q0v.x + q0v.y, q1 = q1v.x + q1v.y;
            float sd0 = sd0v.x + sd0v.y, sd1 = sd1v.x + sd1v.y, vs = vsv.x + vsv.y;
            a0 += __shfl_xor(a0, 32); a1 += __shfl_xor(a1, 32);
            q0 += __shfl_xor(q0, 32); q1 += __shfl_xor(q1, 32);
            sd0 += __shfl_xor(sd0, 32); sd1 += __shfl_xor(sd1, 32);
            vs += __shfl_xor(vs, 32);
            const float y0 = s_yw[(0 * S + r) * TWB + tsl + i], y1 = s_yw[(1 * S + r) * TWB + tsl + i];
            float e0 = a0 + q0 + sd0 - y0, e1 = a1 + q1 + sd1 - y1;
            if (own && h == 0) Eacc += (double)(e0 * e0 + e1 * e1 + vs);
            e0 *= mv; e1 *= mv;

            STAMP(2);
            f32x16 HW;
#pragma unroll
            for (int g = 0; g < 16; ++g) HW[g] = 0.f;
#ifdef ABLATE_HW
#pragma unroll
            for (int s = 0; s < 16; ++s) HW[s] = (s < 8 ? al[(s & 7) >> 1][s & 1] : be[(s & 7) >> 1][s & 1]);
#else
#pragma unroll
            for (int s = 0; s < 16; ++s)
                HW = MFMA32(Aop[(24 + s) * 64], s < 8 ? al[(s & 7) >> 1][s & 1] : be[(s & 7) >> 1][s & 1], HW);
#endif

            STAMP(3);
            // ---- anti-diagonal sums of the moment gradients (symbol = window + tap): the gradients of the two means and the
            //      two variances of a (window, tap) pair go through the wave's zero-padded buffer as one 4-vector (the other
            //      two pairs: k_g2_cx).  LDS instructions of one wave execute in program order, so lanes may read what other
            //      lanes of the same wave wrote earlier; wave_barrier only pins the compiler's ordering.
            f32x4* Pu = reinterpret_cast<f32x4*>(s_P) + (size_t)(r * NU + sub) * NSIG;
#ifdef ABLATE_ANTIDIAG
#pragma unroll
            for (int q = 0; q < 8; ++q) asm volatile("" ::"v"(HW[q]), "v"(HW[q + 8]), "v"(zz[q >> 1][q & 1]));
#else
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int x = cx_kk(2 * p, h);
                const f2 zp = zP[p], zq = zQ[p], dr = C2(2, p), di = C2(3, p);
                const f2 ed = 2.f * (e0 * dr + e1 * di);
                const f2 da0 = (2.f * mv) * (e0 * zp + e1 * zq + f2{HW[2 * p], HW[2 * p + 1]});
                const f2 db0 = (2.f * mv) * (e1 * zp - e0 * zq + f2{HW[2 * p + 8], HW[2 * p + 9]});
                const f2 da1 = mv * (zz[p] + ed), db1 = mv * (zz[p] - ed);
                Bd[x * 32 + i] = f32x4{da0.x, db0.x, da1.x, db1.x};
                Bd[(x + 1) * 32 + i] = f32x4{da0.y, db0.y, da1.y, db1.y};
            }
            __builtin_amdgcn_wave_barrier();
            {
                f32x4 sv = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int x = 0; x < 16; ++x)   // row x, window lane - x (the zero vector when that is not a window of the slab)
                    sv += Bd[(unsigned)(lane - x) < 32u ? x * 31 + lane : 16 * 32];
                Pu[lane] = sv;
            }
            __builtin_amdgcn_wave_barrier();
#endif

            STAMP(4);
            // ---- rho / sigma (transposed through LDS), vector gradients
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int x = cx_kk(2 * p, h);
                const int slot = tsl + i + x;
                const f2 ma = ld2(s_mu + slot), mb = ld2(s_mu + MS + slot);
                const f2 gr = 2.f * (e0 * ma + e1 * mb), gi = 2.f * (e1 * ma - e0 * mb);
                ghr[p] += mo * (gr + al[p]);
                ghi[p] += mo * (gi + be[p]);
                gdr[p] += (mo * 2.f * e0) * dvv[p];
                gdi[p] += (mo * 2.f * e1) * dvv[p];
                const f2 t0 = mo * (gr + 2.f * al[p]), t1 = mo * (gi + 2.f * be[p]);
                Tb[i * TP + x] = t0.x;
                Tb[i * TP + x + 1] = t0.y;
                Tb[i * TP + 16 + x] = t1.x;
                Tb[i * TP + 16 + x + 1] = t1.y;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int w = 2 * s + h;
                const float a = Tb[w * TP + i];
                const float b = s_mu[cj * MS + tsl + w + yj];
#ifdef ABLATE_G
                R[s] += a * b;
#else
                R = MFMA32(a, b, R);
#endif
            }
            __builtin_amdgcn_wave_barrier();
            STAMP(5);
        }
        __syncthreads();
        if (more) stage_store(s_stage);
        // ---- gather: symbol slot ms = kmax + ml is covered by subtile ms / 32 and, if ms % 32 <= Le - 2, by the one before
        //      (one thread per symbol: the 4-vector holds the gradients of both streams' means and variances)
        for (int ml = tid; ml < TS; ml += CX2_NT) {
            const int64_t m = s0 + ml;
            if (m < s1) {
                const int ms = ml + l.kmax;
                const int sub_hi = ms >> 5, sg_hi = ms & 31;
                const bool two = (sg_hi + 32 < 32 + l.LE - 1) && sub_hi > 0;
                const f32x4* P4 = reinterpret_cast<const f32x4*>(s_P);
                f32x4 g = {0.f, 0.f, 0.f, 0.f};
                for (int rr = 0; rr < S; ++rr) {
                    if (two) g += P4[(rr * NU + sub_hi - 1) * NSIG + sg_hi + 32];
                    if (sub_hi < NU) g += P4[(rr * NU + sub_hi) * NSIG + sg_hi];
                }
                float* gp = gmom + (m - l.sym_lo);   // planes 2 and 3 of either stream: k_g2_cx
                const int64_t st = l.sym_cnt;
                const float gT = gT_pre;   // TS <= threads per workgroup: ml == tid
                gp[0] = g.x; gp[st] = g.z + gT; gp[4 * st] = g.y; gp[5 * st] = g.w + gT;
            }
        }
        __syncthreads();
        STAMP(6);
    }
#ifdef V2VAE_STAMPS
    if (lane == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(ws + l.o_sumstage);  // free until the record sums run
        for (int k = 0; k < 8; ++k) dbg[((size_t)blockIdx.x * NW + wv) * 8 + k] = st_acc[k];
    }
#endif

    // ---- one record per (block, phase); Ps block of the record is left zero (filled from the lagged sums later)
    float* img = s_T + (size_t)r * CX_IMG;  // NW * 32 * TP >= 4224 floats >= S * CX_IMG
    float vsum[8][4];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float v0 = ghr[q >> 1][q & 1], v1 = ghi[q >> 1][q & 1], v2 = gdr[q >> 1][q & 1], v3 = gdi[q >> 1][q & 1];
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            v0 += __shfl_xor(v0, off); v1 += __shfl_xor(v1, off);
            v2 += __shfl_xor(v2, off); v3 += __shfl_xor(v3, off);
        }
        vsum[q][0] = v0; vsum[q][1] = v1; vsum[q][2] = v2; vsum[q][3] = v3;
    }
    for (int w = 0; w < NWP; ++w) {
        if (wsub == w) {
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const int e = ((g & 3) + 8 * (g >> 2) + 4 * h) * 32 + i;
                if (w == 0) { img[e] = R[g]; img[1024 + e] = 0.f; }
                else img[e] += R[g];
            }
            if (i == 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int x = cx_kk(q, h);
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        if (w == 0) img[2048 + v * 16 + x] = vsum[q][v];
                        else img[2048 + v * 16 + x] += vsum[q][v];
                    }
                }
            }
        }
        __syncthreads();
    }
    // record: Re = R00 + R11, Im = R10 - R01 of the 32x32 tile R = [rho | sigma]^T [mu_a | mu_b], then the vectors
    float* rec = ws + l.o_part_cx + (size_t)blockIdx.x * S * CX_REC;
    for (int e = tid; e < S * CX_REC; e += CX2_NT) {
        const int rr = e / CX_REC, q = e - rr * CX_REC;
        const float* im = s_T + (size_t)rr * CX_IMG;
        float v;
        if (q < 256) { const int a = q >> 4, b = q & 15; v = im[a * 32 + b] + im[(16 + a) * 32 + 16 + b]; }
        else if (q < 512) { const int a = (q - 256) >> 4, b = q & 15; v = im[(16 + a) * 32 + b] - im[a * 32 + 16 + b]; }
        else v = im[2048 + (q - 512)];
        rec[e] = v;
    }
    double v = Eacc;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if (lane == 0) dred[wv] = v;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < CX2_NT / 64; ++w) s += dred[w];
        reinterpret_cast<double*>(ws + l.o_Epart)[blockIdx.x] = s;
    }
}

// ------------------------------------------------------------------------------------------------------------
// The same mathematics without a workgroup barrier in the main loop: every wave works alone on a RUN of consecutive windows
// of one phase (the run's symbols [a, b) plus the kmax windows before a, which it recomputes un-owned, exactly as a block tile
// does above).  A run is cut into strips of cx3_wt(S) windows whose moments the wave stages in its own LDS planes; the per-symbol
// gradients (symbol = window + tap) are accumulated with LDS float adds into the wave's own per-symbol arrays - one array per
// lane half, so that no two lanes of one instruction ever meet at an address and the result is bit-reproducible - and carried
// from strip to strip; a strip's finished symbols go straight to global memory.  With two phases (S = 2) the two waves that cover
// the same symbols write separate planes (phase 0: the gradient planes proper, phase 1: a second set) and the per-symbol
// backward adds them, together with the correlation part of the variance gradient.
// NK = k-steps of the H X and H w products: 16, or 13 when no phase has more than 13 active taps (the k index then runs over
// [stream][13 taps] and three values of the H w operand change lane halves).
// ------------------------------------------------------------------------------------------------------------
// H X on the bf16 matrix pipe in split arithmetic (complex model, one or two phases): every float is the exact sum of three
// bf16 pieces (8 + 8 + 8 significand bits, by truncation), and the six piece products whose weight is not below 2^-24 of the
// whole are accumulated in float by v_mfma_f32_32x32x16_bf16 - twelve matrix instructions of 32 cycles instead of thirteen of
// 64, which moreover leave the vector pipe free while they run (an f32 matrix instruction holds it for its 64 cycles).  The
// means are split once per symbol when they are staged, the table once per workgroup.
__host__ __device__ constexpr bool cx3_bfx(int S, bool REAL) { return !REAL && S <= 2; }
// windows per strip: 192 with one or two phases (128 where the bf16 pieces of the means take their share of the wave's area);
// 96 with four samples per symbol, whose four per-phase tables leave the eight wave areas less of the 160 KB
__host__ __device__ constexpr int cx3_wt(int S, bool BFX) { return S <= 2 ? (BFX ? 128 : 192) : 96; }
// staged symbols per strip (16 used beyond the windows; 32 keeps the gradient update of a slab, 64 lanes wide, inside the arrays)
__host__ __device__ constexpr int cx3_ms(int S, bool BFX) { return cx3_wt(S, BFX) + 32; }
constexpr int CX3_NW = 8;                   // waves per workgroup
#ifndef V2VAE_CX3_BFK
#define V2VAE_CX3_BFK 1
#endif
constexpr bool CX3_BFK = V2VAE_CX3_BFK != 0;   // (with the split-bf16 products) dE/dH on the bf16 matrix pipe as well
constexpr int CX3_ADB = 16 * 32 + 4;        // 4-vectors of the anti-diagonal buffer (two planes of 8 tap-pair rows of 32 windows, a zero vector behind each)
// floats of one wave's private LDS: 7 moment planes, the per-symbol gradient 4-vectors, the anti-diagonal buffer (which the
// transposing buffer of the rho / sigma rows shares: the two are used one after the other within a slab), and with BFX the
// bf16 pieces of the means [2 streams][3 pieces][MS] (two bytes each)
__host__ __device__ constexpr int cx3_wave_f(int S, bool BFX) {
    return 7 * cx3_ms(S, BFX) + 4 * cx3_ms(S, BFX) + 4 * CX3_ADB + (BFX ? 3 * cx3_ms(S, BFX) : 0);
}
// floats of one phase's constant operands: f32 images [steps][64] of H X and H w, or with BFX the bf16 images of the two
// tables, [2 tables][3 pieces][2 chunks][64 lanes][8] (two bytes each)
__host__ __device__ constexpr int cx3_tab_f32(int NK, bool BFX) { return BFX ? 0 : 2 * NK * 64; }
constexpr int CX3_TAB_BF = 2 * 3 * 2 * 64 * 8 / 2;

__host__ __device__ constexpr int cx3_lds_floats(int S, int NK, bool BFX) {
    return S * cx3_tab_f32(NK, BFX) + S * 64 + (BFX ? S * CX3_TAB_BF : 0) + CX3_NW * cx3_wave_f(S, BFX);
}

// REAL: the real model (one stream).  Everything that is identically zero with a zero quadrature stream is a compile-time zero in
// the REAL instances (Q = H X's second half, the quadrature means / 3rd moments, Im of the tables, the second residual, beta,
// sigma), so its arithmetic folds away: H X takes 7 (8) instead of 13 (16) matrix steps, H w 8 instead of 13 (16), dE/dH one
// product instead of three.
// The quadrature stream is identically zero - its samples read as zero, its moment planes are
// zero-filled by the soft demapper, the imaginary planes of the tables are zero - and the three terms of the reference's
// diagonal quirk (lib/real_equalization.py:869: the mean uses diag(h2) where the second moment uses the diagonal d = 2 diag(h2) of
// the symmetrised kernel) are added: with T_n = y'_n sum_x v_{n+x} d_x,  E += T_n,  dE/dv_{n+x} += y'_n d_x,  dE/dd_x += y'_n v_{n+x}.
template <int S, int NK, bool REAL, bool BFX>
__global__ __launch_bounds__(CX3_NW * 64) void k_elbo_cx3(Lay l, const float* __restrict__ y, float* __restrict__ ws) {
    extern __shared__ __align__(16) float sm[];
    static_assert(!BFX || cx3_bfx(S, REAL), "the split-bf16 H X product serves the complex model with one or two phases");
    constexpr int WT = cx3_wt(S, BFX), MS = cx3_ms(S, BFX), CX3_WAVE_F = cx3_wave_f(S, BFX), TP = CX_TP, NW = CX3_NW;
    constexpr int NWP = NW / S;
    constexpr int CX3_NT = NW * 64;
    static_assert(NK == 16 || NK == 13, "k-step counts with an operand schedule");
    constexpr int NA = cx3_tab_f32(NK, BFX) / 64;   // f32 A-operand steps per phase
    constexpr int HW0 = NK;                    // first step of the H w operands
    float* s_A = sm;                           // [S][NA][64]    A operands (not with BFX): steps 0..NK-1 H X, then NK steps of H w
    float* s_c = s_A + S * NA * 64;            // [S][4][16]     h1r, h1i, diag Hr, diag Hi
    // BFX: the two tables in bf16 pieces, [S][2 tables][3 pieces][2 chunks][64 lanes][8]: lane (row, half), element j of chunk c
    // is the entry of k-slot (stream = half, tap = 8 c + j) of H X, of k-slot (stream = c, tap = cx_kk(j, half)) of H w - the
    // tap that accumulator slot j of that lane half holds, so the per-tap algebra's (alpha, beta) are the B operand as they are
    unsigned short* s_Ab = reinterpret_cast<unsigned short*>(s_c + S * 64);
    float* s_W = s_c + S * 64 + (BFX ? S * CX3_TAB_BF : 0);   // [NW][CX3_WAVE_F]
    __shared__ double dred[CX3_NT / 64];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = lane & 31, h = lane >> 5;
    const int r = wv / NWP, wsub = wv % NWP;
    float* stg = s_W + wv * CX3_WAVE_F;        // [7][MS]
    f32x4* G4 = reinterpret_cast<f32x4*>(stg + 7 * MS);   // [MS] per-symbol gradients (d/d mean_a, mean_b, var_a, var_b) of the strip
    f32x4* Bd = G4 + MS;                       // [ADB] anti-diagonal buffer
    float* Tr = reinterpret_cast<float*>(Bd);  // [32 windows][17] rho rows and, after them, sigma rows: the transposing buffers of
    float* Ts = Tr + 32 * 17;                  // the dE/dH accumulation, in the same memory as the anti-diagonal buffer
    unsigned short* Xb = reinterpret_cast<unsigned short*>(Bd + CX3_ADB);   // BFX: [2 streams][3 pieces][MS] bf16 pieces of the means
    static_assert(2 * 32 * 17 <= 4 * 16 * 32 && S * CX_REC <= CX3_WAVE_F, "shared regions");
    const int i16 = lane & 15, kq = lane >> 4;
    const float* __restrict__ mom = ws + l.o_mom;
    const int64_t own_hi = l.own_lo + l.own_cnt, sym_hi = l.sym_lo + l.sym_cnt, st = l.sym_cnt;

    // ---- constants of every phase into LDS (compile-time trip count: unrolled, so that the loads of all passes are in flight together)
#pragma unroll
    for (int idx = tid; idx < S * NA * 64; idx += CX3_NT) {
        const int ln = idx & 63, s = (idx >> 6) % (BFX ? 1 : NA), rr = idx / ((BFX ? 1 : NA) * 64);
        const int jj = ln & 31, hh = ln >> 5, cjj = jj >> 4, yjj = jj & 15;
        const float* Hr = ws + l.o_cxH + (size_t)(rr * 2 + 0) * 256;
        const float* Hi = ws + l.o_cxH + (size_t)(rr * 2 + 1) * 256;
        float v;
        if (s < NK) {          // [[Hr, Hi], [-Hi, Hr]], row k = 2 s + h over [stream][NK taps]
            const int k = 2 * s + hh, ck = k / NK, xk = k - ck * NK;
            float a = Hr[xk * 16 + yjj], b = Hi[xk * 16 + yjj];
            // NK = 13: output row 13 (no such tap) computes the first-order term h1 . X of the window instead
            if (NK == 13 && yjj == 13) { a = ws[l.o_cxh1 + (rr * 2 + 0) * 16 + xk]; b = ws[l.o_cxh1 + (rr * 2 + 1) * 16 + xk]; }
            v = ck == 0 ? (cjj == 0 ? a : b) : (cjj == 0 ? -b : a);
        } else {               // [[Hr, -Hi], [Hi, Hr]], row = the (stream, tap) that cx3_hw_slot puts on k-slot (s - NK, h)
            int ck, xk;
            cx3_hw_slot<NK>(s - NK, hh, &ck, &xk);
            // (doubled: the operand the chain takes from the registers is w / 2, see the per-tap algebra)
            const float a = 2.f * Hr[xk * 16 + yjj], b = 2.f * Hi[xk * 16 + yjj];
            v = ck == 0 ? (cjj == 0 ? a : -b) : (cjj == 0 ? b : a);
        }
        s_A[idx] = v;
    }
    if (BFX) {
#pragma unroll
        for (int idx = tid; idx < S * 2 * 2 * 64 * 8; idx += CX3_NT) {
            const int j = idx & 7, ln = (idx >> 3) & 63, c = (idx >> 9) & 1, tb = (idx >> 10) & 1, rr = idx >> 11;
            const int jj = ln & 31, hh = ln >> 5, cjj = jj >> 4, yjj = jj & 15;
            const int ck = tb ? c : hh, xk = tb ? cx_kk(j, hh) : 8 * c + j;
            const float* Hr = ws + l.o_cxH + (size_t)(rr * 2 + 0) * 256;
            const float* Hi = ws + l.o_cxH + (size_t)(rr * 2 + 1) * 256;
            float a = Hr[xk * 16 + yjj], b = Hi[xk * 16 + yjj];
            // NK = 13: output row 13 of H X (no such tap) computes the first-order term h1 . X of the window instead
            if (!tb && NK == 13 && yjj == 13) { a = ws[l.o_cxh1 + (rr * 2 + 0) * 16 + xk]; b = ws[l.o_cxh1 + (rr * 2 + 1) * 16 + xk]; }
            // H X: [[Hr, Hi], [-Hi, Hr]];  H w: 2 [[Hr, -Hi], [Hi, Hr]] (doubled: the operand the chain takes is w / 2)
            float v = tb ? 2.f * (ck == 0 ? (cjj == 0 ? a : -b) : (cjj == 0 ? b : a))
                         : (ck == 0 ? (cjj == 0 ? a : b) : (cjj == 0 ? -b : a));
            if (xk >= NK) v = 0.f;
            unsigned b1, b2, b3;
            bf16_split3_rne(v, &b1, &b2, &b3);   // (split once per workgroup: pieces rounded to nearest)
            unsigned short* d = s_Ab + ((size_t)((rr * 2 + tb) * 3) * 2 + c) * 512 + ln * 8 + j;
            d[0] = (unsigned short)(b1 >> 16);
            d[2 * 512] = (unsigned short)(b2 >> 16);
            d[4 * 512] = (unsigned short)(b3 >> 16);
        }
    }
    for (int idx = tid; idx < S * 64; idx += CX3_NT) {
        const int rr = idx / 64, q = (idx >> 4) & 3, x = idx & 15;
        const float* Hr = ws + l.o_cxH + (size_t)(rr * 2 + 0) * 256;
        const float* Hi = ws + l.o_cxH + (size_t)(rr * 2 + 1) * 256;
        s_c[idx] = q == 0 ? ws[l.o_cxh1 + (rr * 2 + 0) * 16 + x] : q == 1 ? ws[l.o_cxh1 + (rr * 2 + 1) * 16 + x]
                 : q == 2 ? Hr[x * 16 + x] : Hi[x * 16 + x];
    }
    const float* Aop = s_A + (size_t)r * NA * 64 + lane;
    const u32x4* Abf = reinterpret_cast<const u32x4*>(s_Ab) + (size_t)r * 12 * 64 + lane;   // [table][piece][chunk][64]
    const float* cst = s_c + r * 64 + 4 * h;
    __syncthreads();

#ifdef V2VAE_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
#endif
    // dE/dH = sum over windows of (rho + i sigma)^T conj(X), a complex 16x16 accumulation, from three real products instead of
    // four:  K1 = (rho + sigma)^T mu_a,  K2 = -rho^T (mu_a + mu_b),  K3 = sigma^T (mu_a - mu_b);  Re = K1 - K3, Im = K1 + K2
    // (combined once at the end: the sums are linear).  16x16x4 MFMAs, four windows per k-step.
    f32x4 K1 = {0.f, 0.f, 0.f, 0.f}, K2 = K1, K3 = K1;
    f2 ghr[4], ghi[4], gdr[4], gdi[4];  // vector gradients of the tap pairs (2 p, 2 p + 1)
    f2 cdr[4], cdi[4], cdd[4];          // diagonal of the kernel at this lane's tap pairs (both passes of a slab use it), its squared modulus
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        ghr[p] = 0.f; ghi[p] = 0.f; gdr[p] = 0.f; gdi[p] = 0.f; cdr[p] = C2(2, p); cdi[p] = C2(3, p);
        cdd[p] = REAL ? cdr[p] * cdr[p] : cdr[p] * cdr[p] + cdi[p] * cdi[p];
        asm volatile("" : "+v"(cdd[p]));   // (kept in registers: recomputed per slab it costs two instructions per tap pair)
    }
    double Eacc = 0.0;

    // ---- this wave's run: symbols [a, b) of phase r
    const int64_t run = (int64_t)blockIdx.x * NWP + wsub;
    const int64_t a = l.own_lo + run * l.cx_rl;
    const int64_t b = (a + l.cx_rl < own_hi) ? a + l.cx_rl : own_hi;
    // this phase's array of per-symbol 4-vectors (gradients of mean_a, mean_b, var_a, var_b)
    f32x4* __restrict__ gout = reinterpret_cast<f32x4*>(ws + l.o_gpk) + (int64_t)r * st;

    auto load_y = [&](int64_t t, int o) -> float {
        const int64_t n = t * S + r;
        float v = 0.f;
        if ((!REAL || o == 0) && t < b && n >= 0 && n < l.W) {
            const int64_t loc = l.pre + n - l.y_lo;
            if (loc >= 0 && loc < l.y_len) v = y[(int64_t)o * l.y_stride + loc];
        }
        return v;
    };

    // Moments travel from global memory to the wave's planes 32 symbols at a time, one slab ahead of their use: lane (j, g) holds
    // the four moments of stream g of symbol j of the group; the sums / differences across the streams are formed on the way in.
    float pm[4];
    auto fetch32 = [&](int64_t m0) {
        const int64_t m = m0 + i;
        pm[0] = 0.f; pm[1] = 0.f; pm[2] = 0.f; pm[3] = 0.f;
        if (m >= l.sym_lo && m < sym_hi) {
            const float* g = mom + (int64_t)h * 4 * st + (m - l.sym_lo);
            pm[0] = g[0]; pm[1] = g[st]; pm[2] = g[2 * st]; pm[3] = g[3 * st];
        }
    };
    auto store32 = [&](int e0, int jmin) {   // members j >= jmin of the held group become staged symbols e0 + j
        const float ov = __shfl_xor(pm[1], 32), og = __shfl_xor(pm[3], 32);
        float* d = stg + e0 + i;
        if (i >= jmin) {
            d[h * MS] = pm[0];                                  // mean_a | mean_b
            if (BFX) {   // and its three bf16 pieces (exact sum, each rounded to nearest: split once per symbol)
                unsigned b1, b2, b3;
                bf16_split3_rne(pm[0], &b1, &b2, &b3);
                unsigned short* x = Xb + h * 3 * MS + e0 + i;
                x[0] = (unsigned short)(b1 >> 16);
                x[MS] = (unsigned short)(b2 >> 16);
                x[2 * MS] = (unsigned short)(b3 >> 16);
            }
            d[(2 + h) * MS] = h ? ov - pm[1] : pm[1] + ov;      // v_a + v_b | v_a - v_b
            d[(4 + h) * MS] = pm[2];                            // 3rd central moments
            if (h == 0) d[6 * MS] = pm[3] + og;                 // g_a + g_b
        }
    };

    bool first = true;
    for (int64_t ts = a - l.kmax; a < own_hi && ts < b; ts += WT, first = false) {
        const int64_t left = b - ts;
        const int nsl = left >= WT ? WT / 32 : (int)((left + 31) / 32);   // slabs of this strip
        // ---- per-symbol gradients: keep what the previous strip left for this strip's first 16 symbols, clear the rest;
        //      the last 16 staged symbols of the previous strip are this strip's first 16
        {
            f32x4 carry = {0.f, 0.f, 0.f, 0.f};
            float c0 = 0.f, c1 = 0.f;
            unsigned cw = 0;
            unsigned* Xw = reinterpret_cast<unsigned*>(Xb);   // (pairs of bf16 pieces)
            if (!first) {
                if (lane < 16) carry = G4[WT + lane];
                c0 = stg[(lane >> 4) * MS + WT + (lane & 15)];
                if (lane < 48) c1 = stg[(4 + (lane >> 4)) * MS + WT + (lane & 15)];
                if (BFX && lane < 48) cw = Xw[(lane >> 3) * (MS / 2) + WT / 2 + (lane & 7)];
            }
            __builtin_amdgcn_wave_barrier();
            for (int e = lane; e < MS; e += 64) G4[e] = f32x4{0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_wave_barrier();
            if (lane < 16) G4[lane] = carry;
            if (lane < 2) Bd[8 * 32 + lane * (8 * 32 + 1)] = f32x4{0.f, 0.f, 0.f, 0.f};   // the zero vectors behind the two planes of the anti-diagonal buffer
            if (first) {   // start of the run: symbols [ts, ts + 16) directly, the first group of 32 into the registers
                fetch32(ts - 16);
                store32(-16, 16);
            } else {
                stg[(lane >> 4) * MS + (lane & 15)] = c0;
                if (lane < 48) stg[(4 + (lane >> 4)) * MS + (lane & 15)] = c1;
                if (BFX && lane < 48) Xw[(lane >> 3) * (MS / 2) + (lane & 7)] = cw;
            }
        }
        if (first) fetch32(ts + 16);
        __builtin_amdgcn_wave_barrier();
        const float* s_mu = stg;
        const float* s_S = stg + 2 * MS;
        const float* s_dv = stg + 3 * MS;
        const float* s_ka = stg + 4 * MS;
        const float* s_kb = stg + 5 * MS;
        const float* s_gs = stg + 6 * MS;
        float yn0 = load_y(ts + i, 0), yn1 = load_y(ts + i, 1);
        STAMP(0);

        for (int sub = 0; sub < nsl; ++sub) {
            const int tsl = sub * 32;
            const int64_t t = ts + tsl + i, n = t * S + r;
            const bool valid = (t < b) && n >= 0 && n < l.W;
            const bool own = valid && t >= a;
            const float mv = valid ? 1.f : 0.f, mo = own ? 1.f : 0.f;
            const float y0 = yn0, y1 = REAL ? 0.f : yn1;
            if (sub + 1 < nsl) { yn0 = load_y(t + 32, 0); yn1 = load_y(t + 32, 1); }
            store32(tsl + 16, 0);                    // symbols [tsl + 16, tsl + 48) complete what this slab reads
            fetch32(ts + tsl + 48);                  // the group the next slab (of this or of the next strip) will add
            __builtin_amdgcn_wave_barrier();

            f32x16 U = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (BFX) {
                // B operand: k-slot (chunk c, lane half h, element j) = (stream h, tap 8 c + j), i.e. the 16 consecutive bf16
                // of stream h from staged symbol tsl + i on.  Nine aligned words cover them; an odd window shifts by one piece.
                // (one opaque LDS address register: the 27 words sit within the offset field of the reads)
                typedef const unsigned __attribute__((address_space(3))) * lword_p;
                lword_p xw = (lword_p)(reinterpret_cast<const unsigned*>(Xb) + h * 3 * (MS / 2) + ((tsl + i) >> 1));
                asm volatile("" : "+v"(xw));
                const unsigned sh = (i & 1) * 16;
                u32x4 d[3][2];
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    unsigned w[9];
#pragma unroll
                    for (int q = 0; q < 9; ++q) w[q] = xw[p * (MS / 2) + q];
#pragma unroll
                    for (int q = 0; q < 8; ++q) d[p][q >> 2][q & 3] = __builtin_amdgcn_alignbit(w[q + 1], w[q], sh);
                }
                // table piece x operand piece, smallest products first: (1,3) (2,2) (3,1) (1,2) (2,1) (1,1)
                constexpr int TP6[6] = {0, 1, 2, 0, 1, 0}, DP6[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
                for (int k = 0; k < 6; ++k)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        U = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfx8, Abf[(TP6[k] * 2 + c) * 64]),
                                                                    __builtin_bit_cast(bfx8, d[DP6[k]][c]), U, 0, 0, 0);
            } else {
#pragma unroll
            for (int s = 0; s < NK; ++s) {
                // k = 2 s + h over [stream][NK taps]
                int off;
                if (2 * s + 1 < NK) off = 2 * s + h;
                else if (2 * s >= NK) off = MS + 2 * s - NK + h;
                else off = h ? MS : NK - 1;
                if (REAL && 2 * s >= NK) continue;   // both k-slots of the step belong to the quadrature stream
                U = MFMA32(Aop[s * 64], s_mu[tsl + i + off], U);
            }
            // operands of the first steps requested before the chain starts, then two reads per step (the constant A operands
            // and the means first, the moments of the per-tap algebra behind them)
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);
#pragma unroll
            for (int s = 0; s < NK; ++s) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            }

            STAMP(1);
            // NK = 13: rows 13 of the product (upper lane half, accumulator slots 5 and 13) hold the first-order term of the window
            float a0 = 0.f, a1 = 0.f;
            if (NK == 13) {
                a0 = h ? U[5] : 0.f; a1 = (!REAL && h) ? U[13] : 0.f;
                U[5] = h ? 0.f : U[5]; U[13] = h ? 0.f : U[13];
            }
            // per-tap algebra on tap PAIRS (2 p, 2 p + 1).  One address register per PAIR of planes (the offset field of a
            // two-float LDS read reaches the second plane of a pair but not the third): made opaque, or the compiler forms every
            // plane address of every tap pair from one base with an addition of its own
            // (pointers typed as LDS addresses: an opaque generic pointer would turn the reads into flat loads)
            typedef const float __attribute__((address_space(3))) * lfloat_p;
            lfloat_p b0 = (lfloat_p)(stg + tsl + i + 4 * h);
            lfloat_p b2 = b0 + 2 * MS, b4 = b0 + 4 * MS, b6 = b0 + 6 * MS;
            asm volatile("" : "+v"(b0), "+v"(b2), "+v"(b4), "+v"(b6));
            auto ldl = [](lfloat_p q) { return f2{q[0], q[1]}; };
            f2 zP[4], zQ[4], al[4], be[4], zz[4], dvv[4];
            f2 a0v = 0.f, a1v = 0.f, q0v = 0.f, q1v = 0.f, sd0v = 0.f, sd1v = 0.f, vsv = 0.f;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int xo = cx_kk(2 * p, 0);
                const f2 zero2 = {0.f, 0.f};
                const f2 P = f2{U[2 * p], U[2 * p + 1]}, Q = REAL ? zero2 : f2{U[2 * p + 8], U[2 * p + 9]};
                const f2 ma = ldl(b0 + xo), mb = REAL ? zero2 : ldl(b0 + MS + xo), Sv = ldl(b2 + xo), dv = ldl(b2 + MS + xo);
                const f2 ka = ldl(b4 + xo), kb = REAL ? zero2 : ldl(b4 + MS + xo), gs = ldl(b6 + xo);
                const f2 hr = C2(0, p), hi = REAL ? zero2 : C2(1, p), dr = cdr[p], di = REAL ? zero2 : cdi[p];
                const f2 zp = 2.f * P + hr, zq = 2.f * Q + hi;
                if (NK != 13) {
                    a0v += hr * ma - hi * mb;
                    a1v += hi * ma + hr * mb;
                }
                q0v += ma * P - mb * Q;
                q1v += ma * Q + mb * P;
                sd0v += dv * dr;
                sd1v += dv * di;
                const f2 z2 = zp * zp + zq * zq;
                const f2 ca = dr * zp + di * zq, cb = dr * zq - di * zp;
                vsv += Sv * z2 + 2.f * (ka * ca + kb * cb) + gs * cdd[p];
                // (alpha, beta) / 2: the factor two rides in the constant operand of the H w product, which is exact
                al[p] = Sv * zp + ka * dr - kb * di;
                be[p] = Sv * zq + ka * di + kb * dr;
                zP[p] = zp; zQ[p] = zq;
                zz[p] = z2;
                dvv[p] = dv;
                gdr[p] += (2.f * mo) * (ka * zp + kb * zq + gs * dr);
                if (REAL) gdr[p] += (mo * y0) * Sv;
                gdi[p] += (2.f * mo) * (ka * zq - kb * zp + gs * di);
            }
            if (NK != 13) { a0 = a0v.x + a0v.y; a1 = a1v.x + a1v.y; }
            // the two lane halves hold disjoint taps of the window: the per-window sums are completed by one half swap each
            // (vector pipe; a shuffle would travel through the LDS pipeline and be waited for)
            float sd0 = sd0v.x + sd0v.y;
            const float vs = cx_sum_halves(vsv.x + vsv.y);
            float e0 = cx_sum_halves((a0 + (q0v.x + q0v.y)) + sd0) - y0;
            float e1 = REAL ? 0.f : cx_sum_halves((a1 + (q1v.x + q1v.y)) + (sd1v.x + sd1v.y)) - y1;
            if (REAL) sd0 = cx_sum_halves(sd0);
            if (own && h == 0) Eacc += (double)(e0 * e0 + e1 * e1 + vs + (REAL ? y0 * sd0 : 0.f));
            e0 *= mv; e1 *= mv;
            const float E0 = 2.f * e0, E1 = 2.f * e1;   // masked and doubled once, instead of per tap pair

            STAMP(2);
            // ---- H w: k-slot (s, h) carries the (stream, tap) of cx3_hw_slot
            f32x16 HW = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (BFX) {
                // (alpha, beta) / 2 of the lane's eight taps in three bf16 pieces each (plain vector instructions: they run
                // beside the matrix pipe), chunk 0 = alpha, chunk 1 = beta
                u32x4 d[3][2];
#pragma unroll
                for (int c = 0; c < 2; ++c)
#pragma unroll
                    for (int p = 0; p < 4; ++p) {
                        const f2 v = c ? be[p] : al[p];
                        const unsigned a1 = __float_as_uint(v.x) & 0xFFFF0000u, b1 = __float_as_uint(v.y) & 0xFFFF0000u;
                        const float ra = v.x - __uint_as_float(a1), rb = v.y - __uint_as_float(b1);
                        const unsigned a2 = __float_as_uint(ra) & 0xFFFF0000u, b2 = __float_as_uint(rb) & 0xFFFF0000u;
                        const float sa = ra - __uint_as_float(a2), sb = rb - __uint_as_float(b2);
                        d[0][c][p] = __builtin_amdgcn_perm(b1, a1, 0x07060302u);
                        d[1][c][p] = __builtin_amdgcn_perm(b2, a2, 0x07060302u);
                        d[2][c][p] = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
                    }
                constexpr int TP6[6] = {0, 1, 2, 0, 1, 0}, DP6[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
                for (int k = 0; k < 6; ++k)
#pragma unroll
                    for (int c = 0; c < 2; ++c)
                        HW = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bfx8, Abf[((3 + TP6[k]) * 2 + c) * 64]),
                                                                     __builtin_bit_cast(bfx8, d[DP6[k]][c]), HW, 0, 0, 0);
            } else if (NK == 16) {
#pragma unroll
                for (int s = 0; s < (REAL ? 8 : 16); ++s)   // (REAL: beta is zero)
                    HW = MFMA32(Aop[(HW0 + s) * 64], s < 8 ? al[(s & 7) >> 1][s & 1] : be[(s & 7) >> 1][s & 1], HW);
            } else {
                // accumulator slots q = 0..4 of either lane half hold valid taps (0-3, 8 | 4-7, 12): ten steps take them as they
                // are; taps 9, 10, 11 (slots 5-7 of the lower half, whose upper-half partners 13-15 do not exist) take three
                // steps in which the upper half supplies the lower half's beta value, moved over by a lane-half swap
#pragma unroll
                for (int s = 0; s < 5; ++s) HW = MFMA32(Aop[(HW0 + s) * 64], al[s >> 1][s & 1], HW);
#pragma unroll
                for (int s = 0; s < (REAL ? 0 : 5); ++s) HW = MFMA32(Aop[(HW0 + 5 + s) * 64], be[s >> 1][s & 1], HW);
#pragma unroll
                for (int s = 5; s < 8; ++s) {
                    float x = al[s >> 1][s & 1], z = be[s >> 1][s & 1];
                    if (REAL) x = h ? 0.f : x;   // (alpha of lanes 0-31 | beta = 0)
                    else asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(z));   // x = (alpha of lanes 0-31 | beta of lanes 0-31)
                    HW = MFMA32(Aop[(HW0 + 5 + s) * 64], x, HW);
                }
            }

            STAMP(3);
            // ---- gradients of the means and variances of symbol (window + tap): anti-diagonal sums of the slab through the
            //      wave's zero-padded buffer, then one update of the strip's per-symbol array (47 symbols).  A (window, tap pair)
            //      writes its register pairs as they are: plane A holds (d mean_a, d mean_b) and plane B (d var_a, d var_b), each as
            //      (even tap, odd tap) pairs; lane m sums the eight pair rows at window m - even tap, and the odd taps' sums, which
            //      belong to the next symbol, move up one lane at the end.
            constexpr int ZV = 8 * 32, PB = ZV + 1;   // the zero vector of plane A; plane B starts behind it and ends in its own
            const float mv2 = 2.f * mv;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int pr = cx_kk(2 * p, h) >> 1;
                const f2 zero2 = {0.f, 0.f};
                const f2 zp = zP[p], zq = zQ[p], dr = cdr[p], di = REAL ? zero2 : cdi[p];
                const f2 ed = E0 * dr + E1 * di;
                const f2 da0 = E0 * zp + (E1 * zq + mv2 * f2{HW[2 * p], HW[2 * p + 1]});
                // (REAL: the gradients with respect to the quadrature moments are not read by anyone)
                const f2 db0 = REAL ? zero2 : E1 * zp - (E0 * zq - mv2 * f2{HW[2 * p + 8], HW[2 * p + 9]});
                f2 da1 = mv * zz[p] + ed;          // (E0, E1 carry the mask already)
                const f2 db1 = REAL ? zero2 : mv * zz[p] - ed;
                if (REAL) da1 += (mv * y0) * dr;
                Bd[pr * 32 + i] = f32x4{da0.x, da0.y, db0.x, db0.y};
                Bd[PB + pr * 32 + i] = f32x4{da1.x, da1.y, db1.x, db1.y};
            }
            Bd[ZV] = f32x4{0.f, 0.f, 0.f, 0.f};   // (the transposing buffers of the previous slab reached over it)
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_sched_barrier(0);
            {
                // the row reads are requested in batches and summed pairwise: two LDS round trips instead of the sixteen
                // dependent ones the register-pressure heuristics of the compiler would leave
                f32x4 ra[8], rb[8];
                const f32x4 g0 = G4[tsl + lane];
#pragma unroll
                for (int pr = 0; pr < 8; ++pr) {   // pair row pr, window lane - 2 pr (the zero vectors when that is not a window of the slab)
                    const int idx = (unsigned)(lane - 2 * pr) < 32u ? pr * 30 + lane : ZV;
                    ra[pr] = Bd[idx];
                    rb[pr] = Bd[idx + PB];
                }
                f32x4 sa = ra[0], sb = rb[0];
#pragma unroll
                for (int pr = 1; pr < 8; ++pr) { sa += ra[pr]; sb += rb[pr]; }
                G4[tsl + lane] = g0 + f32x4{sa[0] + cx_shr1(sa[1]), sa[2] + cx_shr1(sa[3]), sb[0] + cx_shr1(sb[1]), sb[2] + cx_shr1(sb[3])};
                __builtin_amdgcn_sched_group_barrier(0x100, 9, 0);    // G4 + four pair rows
#pragma unroll
                for (int k = 0; k < 4; ++k) {                         // a pair summed, the next pair requested
                    __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x002, 24, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_wave_barrier();

            STAMP(4);
            // ---- rho / sigma (transposed through LDS), vector gradients; the means are read again, all before the first
            //      write (the compiler cannot tell the two LDS regions apart, so a read after a write would wait for it)
            f2 mav[4], mbv[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                mav[p] = ldl(b0 + cx_kk(2 * p, 0));
                mbv[p] = REAL ? f2{0.f, 0.f} : ldl(b0 + MS + cx_kk(2 * p, 0));
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int x = cx_kk(2 * p, h);
                const f2 ma = mav[p], mb = mbv[p];
                const f2 gr = E0 * ma + E1 * mb, gi = E1 * ma - E0 * mb;
                ghr[p] += mo * (gr + 2.f * al[p]);
                ghi[p] += mo * (gi + 2.f * be[p]);
                gdr[p] += (mo * E0) * dvv[p];
                gdi[p] += (mo * E1) * dvv[p];
                const f2 t0 = mo * (gr + 4.f * al[p]), t1 = mo * (gi + 4.f * be[p]);
                Tr[i * 17 + x] = t0.x;
                Tr[i * 17 + x + 1] = t0.y;
                if (!REAL) {   // (sigma is zero)
                    Ts[i * 17 + x] = t1.x;
                    Ts[i * 17 + x + 1] = t1.y;
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);   // the eight pairs of means requested together
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                __builtin_amdgcn_sched_group_barrier(0x002, 16, 0);
                __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (BFX && CX3_BFK) {
                // dE/dH on the bf16 matrix pipe too (v_mfma_f32_16x16x32_bf16: the slab's 32 windows are ONE k-step; lane
                // (tap i16, group kq) carries windows 8 kq .. 8 kq + 7): Re += rho^T mu_a + sigma^T mu_b, Im += sigma^T mu_a - rho^T mu_b
                // (K1 holds Re, K2 Im).  rho / sigma are split in registers; the means are the bf16 planes of the H X product.
                float ar[8], as[8];
#pragma unroll
                for (int st = 0; st < 8; ++st) { ar[st] = Tr[(8 * kq + st) * 17 + i16]; as[st] = Ts[(8 * kq + st) * 17 + i16]; }
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
                typedef const unsigned __attribute__((address_space(3))) * lword_p;
                lword_p xw = (lword_p)(reinterpret_cast<const unsigned*>(Xb) + ((tsl + 8 * kq + i16) >> 1));
                asm volatile("" : "+v"(xw));
                const unsigned shk = (i16 & 1) * 16;
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
                        K1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfx8, rho[tp]), __builtin_bit_cast(bfx8, ya), K1, 0, 0, 0);
                        K2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfx8, sig[tp]), __builtin_bit_cast(bfx8, ya), K2, 0, 0, 0);
                        K1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfx8, sig[tp]), __builtin_bit_cast(bfx8, yb), K1, 0, 0, 0);
                        K2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bfx8, nrho[tp]), __builtin_bit_cast(bfx8, yb), K2, 0, 0, 0);
                    }
                }
            } else {
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int w = 4 * s + kq;   // window of this lane's k-slot; tap i16
                if (REAL) {   // sigma = 0, mu_b = 0: dE/dH = rho^T mu_a (K2 and K3 stay zero: Re = K1 - K3, Im = K1 + K2 is not read)
                    K1 = MFMA16(Tr[w * 17 + i16], s_mu[tsl + w + i16], K1);
                    continue;
                }
                const float rho = Tr[w * 17 + i16], sig = Ts[w * 17 + i16];
                const float ma = s_mu[tsl + w + i16], mb = s_mu[MS + tsl + w + i16];
                K1 = MFMA16(rho + sig, ma, K1);
                K2 = MFMA16(-rho, ma + mb, K2);
                K3 = MFMA16(sig, ma - mb, K3);
            }
            // operands of the first k-steps requested before the chain starts, the others while it runs
            __builtin_amdgcn_sched_group_barrier(0x100, 12, 0);
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 3, 0);
            }
            }
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_wave_barrier();
            STAMP(5);
        }

        // ---- the strip's finished symbols (every window that holds them has been processed) go to global memory
        {
            const int64_t lo = ts > a ? ts : a;
            const int64_t hi2 = ts + nsl * 32 < b ? ts + nsl * 32 : b;
            for (int e = lane; e < nsl * 32; e += 64) {
                const int64_t m = ts + e;
                if (m >= lo && m < hi2) gout[m - l.sym_lo] = G4[e];
            }
        }
        STAMP(6);
        __builtin_amdgcn_wave_barrier();
    }

#ifdef V2VAE_STAMPS
    if (lane == 0) {
        unsigned long long* dbg = reinterpret_cast<unsigned long long*>(ws + l.o_sumstage);  // free until the record sums run
        for (int k = 0; k < 8; ++k) dbg[((size_t)blockIdx.x * NW + wv) * 8 + k] = st_acc[k];
    }
#endif
    // ---- one record per (block, phase); Ps block of the record is left zero (filled from the lagged sums later)
    __syncthreads();
    float* img = s_W + (size_t)r * CX_REC;  // wave 0's private area holds the S records (every wave is past its run here)
    float vsum[8][4];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        float v0 = ghr[q >> 1][q & 1], v1 = ghi[q >> 1][q & 1], v2 = gdr[q >> 1][q & 1], v3 = gdi[q >> 1][q & 1];
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            v0 += __shfl_xor(v0, off); v1 += __shfl_xor(v1, off);
            v2 += __shfl_xor(v2, off); v3 += __shfl_xor(v3, off);
        }
        vsum[q][0] = v0; vsum[q][1] = v1; vsum[q][2] = v2; vsum[q][3] = v3;
    }
    const f32x4 Re = (BFX && CX3_BFK) ? K1 : K1 - K3, Im = (BFX && CX3_BFK) ? K2 : K1 + K2;
    for (int w = 0; w < NWP; ++w) {
        if (wsub == w) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int e = (4 * kq + g) * 16 + i16;
                if (w == 0) { img[e] = Re[g]; img[256 + e] = Im[g]; }
                else { img[e] += Re[g]; img[256 + e] += Im[g]; }
            }
            if (i == 0) {
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int x = cx_kk(q, h);
#pragma unroll
                    for (int v = 0; v < 4; ++v) {
                        if (w == 0) img[512 + v * 16 + x] = vsum[q][v];
                        else img[512 + v * 16 + x] += vsum[q][v];
                    }
                }
            }
        }
        __syncthreads();
    }
    float* rec = ws + l.o_part_cx + (size_t)blockIdx.x * S * CX_REC;
    for (int e = tid; e < S * CX_REC; e += CX3_NT) rec[e] = s_W[e];
    double v = Eacc;
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    __syncthreads();
    if (lane == 0) dred[wv] = v;
    __syncthreads();
    if (tid == 0) {
        double s = 0.0;
        for (int w = 0; w < CX3_NT / 64; ++w) s += dred[w];
        reinterpret_cast<double*>(ws + l.o_Epart)[blockIdx.x] = s;
    }
}

constexpr int PS_PER = 8;               // symbols per thread and tile
constexpr int PS_TILE = NT * PS_PER;    // symbols per tile; the tile's s values plus NLAG of halo are staged in LDS

template <int NLAG>
__global__ __launch_bounds__(NT) void k_psum_cx(Lay l, float* __restrict__ ws) {
    static_assert(NLAG % 4 == 0, "a thread reads its values as 4-vectors");
    __shared__ double red[NT / 64][NLAG];
    __shared__ __align__(16) float s_s[PS_TILE + NLAG];
    constexpr int NLD = (PS_TILE + NLAG + NT - 1) / NT;   // staged values per thread
    constexpr int NV = (PS_PER + NLAG) / 4;               // 4-vectors a thread reads: its PS_PER consecutive symbols and the lags beyond
    const float* __restrict__ mom = ws + l.o_mom;
    double acc[NLAG];
#pragma unroll
    for (int d = 0; d < NLAG; ++d) acc[d] = 0.0;
    const int64_t ntile = (l.own_cnt + PS_TILE - 1) / PS_TILE;
    // the next tile's values travel in registers while this tile is summed (one exposed load latency per workgroup, not per tile)
    float pre[NLD];
    auto fetch = [&](int64_t m0) {
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = threadIdx.x + k * NT;
            pre[k] = idx < PS_TILE + NLAG ? cx_s_at(l, mom, m0 + idx) : 0.f;
        }
    };
    if ((int64_t)blockIdx.x < ntile) fetch(l.own_lo + (int64_t)blockIdx.x * PS_TILE);
    for (int64_t tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
        const int64_t m0 = l.own_lo + tile * PS_TILE;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < NLD; ++k) {
            const int idx = threadIdx.x + k * NT;
            if (idx < PS_TILE + NLAG) s_s[idx] = pre[k];
        }
        __syncthreads();
        if (tile + gridDim.x < ntile) fetch(l.own_lo + (tile + gridDim.x) * PS_TILE);
        // a thread owns PS_PER consecutive symbols: their values and the NLAG - 1 beyond them come as 4-vectors, the products
        // are formed from registers
        float sv[4 * NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(s_s + PS_PER * threadIdx.x + 4 * v);
            sv[4 * v] = t[0]; sv[4 * v + 1] = t[1]; sv[4 * v + 2] = t[2]; sv[4 * v + 3] = t[3];
        }
        float part[NLAG];
#pragma unroll
        for (int d = 0; d < NLAG; ++d) part[d] = 0.f;
#pragma unroll
        for (int q = 0; q < PS_PER; ++q) {
            const float s0 = (m0 + PS_PER * threadIdx.x + q < l.own_lo + l.own_cnt) ? sv[q] : 0.f;
#pragma unroll
            for (int d = 0; d < NLAG; ++d) part[d] = fmaf(s0, sv[q + d], part[d]);
        }
        // the wave's total of the tile per lag (float, on the vector pipe: six lane-shift additions; lane 63 holds it), added to
        // that lane's double accumulator
#pragma unroll
        for (int d = 0; d < NLAG; ++d) acc[d] += (double)cx_wave_sum_to_last(part[d]);
    }
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (lane == 63) {
#pragma unroll
        for (int d = 0; d < NLAG; ++d) red[wv][d] = d <= l.kmax ? acc[d] : 0.0;
    }
    __syncthreads();
    if (threadIdx.x < NLAG) {
        double s = 0.0;
        for (int w = 0; w < NT / 64; ++w) s += red[w][threadIdx.x];
        reinterpret_cast<double*>(ws + l.o_psum_part)[(size_t)blockIdx.x * NLAG + threadIdx.x] = s;
    }
}

// What the backward needs of E that does not depend on the residual, right after the soft demapper:
//  * the gradient with respect to the 3rd central moments is LINEAR in the means with constant coefficients,
//      g_a + i g_b at symbol m = 2 sum_{r,x} conj(d_x) z_{m-x,x},   z_{n,x} = h_x + 2 sum_j H_xj mu_{n+j}
//                             = K0 + 4 sum_lag C[lag] mu_{m+lag},    C[lag] = sum_r sum_{j-x=lag} conj(d_x) H_xj,
//    a (2 Le - 1)-tap complex correlation per symbol; the gradient with respect to the 4th cumulants is the constant sum |d_x|^2;
//  * the term 2 sum_n sum_x s_{n+x} T_{n,x}, T_{n,x} = sum_j |H_xj|^2 s_{n+j}, s = v_a + v_b, of E:
//    its gradient with respect to either variance is the real correlation 4 sum_lag D[lag] s_{m+lag}, D[lag] = sum_r sum_x
//    |H_{x,x+lag}|^2, written to a plane of its own (the ELBO kernel adds it to its part), and its value is
//    2 sum_{r,x,j} |H_xj|^2 (sum_n s_{n+x} s_{n+j}) from the lagged products - one extra workgroup adds it to the partial sums
//    of E - so the ELBO kernel forms no (H.H) s product at all.
// All of this for symbols all of whose windows exist; the few symbols at the two ends of the sequence take the explicit
// sums over the windows that do exist.
constexpr int G2_TILE = 4 * NT;         // symbols per tile: four consecutive symbols per thread

template <int NLAG>
__global__ __launch_bounds__(NT, 5) void k_g2_cx(Lay l, float* __restrict__ ws) {
    constexpr int HALO = NLAG + 8;                     // staged symbols on either side of the tile
    constexpr int PL = (G2_TILE + 2 * HALO) / 4;       // the staged values are kept as four planes by (index mod 4): a
                                                       // thread owns four consecutive symbols and every read is unit-stride
    __shared__ float s_mq[3][4][PL];                   // mean_a, mean_b, v_a + v_b
    __shared__ float s_cf[4 + 2 * (2 * NLAG + 8)];
    __shared__ float s_cd[2 * NLAG + 8];
    __shared__ float s_e[NT / 64][4];
    __shared__ double s_d[NT / 64];
    __shared__ float s_out[3][4][NT];                  // results, for coalesced stores
    const float* __restrict__ mom = ws + l.o_mom;
    float* __restrict__ gmom = ws + l.o_gmom;
    const int64_t own_hi = l.own_lo + l.own_cnt, sym_hi = l.sym_lo + l.sym_cnt, st = l.sym_cnt;
    const int S = l.S;
    const int K4 = (l.kmax + 3) & ~3, ntap = 2 * K4 + 4;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int LE = l.LE, LTp = l.LTp;
    // the first G2_NE workgroups form the value term (first, so that their long chains of dependent loads start at once and
    // overlap the symbol work instead of trailing it); the rest work on symbols
    const int nb = (int)gridDim.x - G2_NE;
    const int bid = (int)blockIdx.x - G2_NE;

    if (bid < 0) {
        // totals of the lagged products over the workgroups of k_psum_cx, fixed order (thread (lag d, chunk c) sums workgroups
        // c, c + nchunk, ...; then the chunks in order); every one of these few workgroups forms them for itself
        __shared__ double s_pp[NT];
        __shared__ double s_tot[NLAG];
        {
            const int nchunk = NT / NLAG;
            const int d = threadIdx.x % NLAG, c = threadIdx.x / NLAG;
            const double* part = reinterpret_cast<const double*>(ws + l.o_psum_part);
            double s = 0.0;
            if (c < nchunk) {
                // eight loads in flight, added in workgroup order
                int b = c;
                for (; b + 7 * nchunk < l.nblk_psum; b += 8 * nchunk) {
                    double v[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(b + u * nchunk) * NLAG + d];
#pragma unroll
                    for (int u = 0; u < 8; ++u) s += v[u];
                }
                for (; b < l.nblk_psum; b += nchunk) s += part[(size_t)b * NLAG + d];
            }
            s_pp[threadIdx.x] = s;
            __syncthreads();
            if (threadIdx.x < NLAG) {
                double t = 0.0;
                for (int k = 0; k < nchunk; ++k) t += s_pp[k * NLAG + threadIdx.x];
                s_tot[threadIdx.x] = t;
            }
            __syncthreads();
        }
        double v = 0.0;
        for (int it = (int)blockIdx.x + G2_NE * (int)threadIdx.x; it < S * LE * LE; it += G2_NE * NT) {
            const int r = it / (LE * LE), rem = it - r * LE * LE;
            const int x = rem / LE, j = rem - x * LE;
            // |H_xj|^2 from the squared stacked tables: [phase][output 0 / 1][tap][tap] (Re^2 + Im^2); the real model has one output
            const float hh = l.cx_real ? ws[l.o_HHc + ((size_t)r * LTp + x) * LTp + j]
                                       : ws[l.o_HHc + ((size_t)(r * 2 + 0) * LTp + x) * LTp + j] + ws[l.o_HHc + ((size_t)(r * 2 + 1) * LTp + x) * LTp + j];
            const float ps = psum_value(l, ws, s_tot, r, x, j);
            ws[l.o_psv + it] = ps;   // kept for the chain rule of dE/dH
            v += 2.0 * (double)hh * (double)ps;
        }
        for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
        if (lane == 0) s_d[wv] = v;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < NT / 64; ++w) t += s_d[w];
            reinterpret_cast<double*>(ws + l.o_Epart)[l.nblk_E - G2_NE + (int)blockIdx.x] = t;
        }
        return;
    }

    for (int i = threadIdx.x; i < 4 + 2 * ntap; i += NT) s_cf[i] = ws[l.o_g2c + i];
    for (int i = threadIdx.x; i < ntap; i += NT) s_cd[i] = ws[l.o_g2c + G2C_D + i];
    const int64_t ntile = (l.own_cnt + G2_TILE - 1) / G2_TILE;
    for (int64_t tile = bid; tile < ntile; tile += nb) {
        const int64_t m0 = l.own_lo + tile * G2_TILE;
        __syncthreads();
        {   // all of a thread's staged values are requested before the first is written (one memory latency per tile)
            constexpr int NST = (G2_TILE + 2 * HALO + NT - 1) / NT;
            float va[NST], vb[NST], vs[NST], vt[NST];
#pragma unroll
            for (int k = 0; k < NST; ++k) {
                const int e = threadIdx.x + k * NT;
                const int64_t m = m0 - HALO + e;
                const bool in = e < G2_TILE + 2 * HALO && m >= l.sym_lo && m < sym_hi;
                const float* g = mom + (m - l.sym_lo);
                va[k] = in ? g[0] : 0.f;
                vb[k] = in ? g[4 * st] : 0.f;
                vs[k] = in ? g[st] : 0.f;
                vt[k] = in ? g[5 * st] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < NST; ++k) {
                const int e = threadIdx.x + k * NT;
                if (e < G2_TILE + 2 * HALO) {
                    s_mq[0][e & 3][e >> 2] = va[k];
                    s_mq[1][e & 3][e >> 2] = vb[k];
                    s_mq[2][e & 3][e >> 2] = vs[k] + vt[k];
                }
            }
        }
        __syncthreads();
        // thread t owns symbols m0 + 4 t .. + 3; the windows of means and of s slide through registers
        const int q0 = (HALO - K4) / 4 + threadIdx.x;    // first quad of the window (HALO and K4 are multiples of 4)
        float wa[8], wb[8], wsv[8], ar[4], ai[4], at[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            wa[k] = s_mq[0][k][q0]; wb[k] = s_mq[1][k][q0]; wsv[k] = s_mq[2][k][q0];
            ar[k] = 0.f; ai[k] = 0.f; at[k] = 0.f;
        }
        for (int g = 0; g < ntap / 4; ++g) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                wa[4 + k] = s_mq[0][k][q0 + g + 1]; wb[4 + k] = s_mq[1][k][q0 + g + 1];
                wsv[4 + k] = s_mq[2][k][q0 + g + 1];
            }
#pragma unroll
            for (int tt = 0; tt < 4; ++tt) {
                const f2 cc = ld2(s_cf + 4 + 2 * (4 * g + tt));
                const float cd = s_cd[4 * g + tt];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    ar[k] = fmaf(cc.x, wa[tt + k], ar[k]); ar[k] = fmaf(-cc.y, wb[tt + k], ar[k]);
                    ai[k] = fmaf(cc.x, wb[tt + k], ai[k]); ai[k] = fmaf(cc.y, wa[tt + k], ai[k]);
                    at[k] = fmaf(cd, wsv[tt + k], at[k]);
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) { wa[k] = wa[4 + k]; wb[k] = wb[4 + k]; wsv[k] = wsv[4 + k]; }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) { s_out[0][k][threadIdx.x] = ar[k]; s_out[1][k][threadIdx.x] = ai[k]; s_out[2][k][threadIdx.x] = at[k]; }
        __syncthreads();
        const float k0r = s_cf[0], k0i = s_cf[1];   // (the constant 4th-cumulant gradient s_cf[2] is read by k_sym_bwd directly)
        for (int i = threadIdx.x; i < G2_TILE; i += NT) {
            const int64_t m = m0 + i;
            // every window that holds this symbol exists: (m - kmax) * S >= 0 and m * S + S - 1 < W
            if (m < own_hi && m >= l.kmax && m * S + S - 1 < l.W) {
                const float gka = fmaf(4.f, s_out[0][i & 3][i >> 2], k0r), gkb = fmaf(4.f, s_out[1][i & 3][i >> 2], k0i);
                const float gt = 4.f * s_out[2][i & 3][i >> 2];
                if (l.cx_split) {
                    reinterpret_cast<f32x4*>(ws + l.o_gpk)[(int64_t)S * st + (m - l.sym_lo)] = f32x4{gka, gkb, gt, s_cf[2]};
                } else {
                    float* gp = gmom + (m - l.sym_lo);
                    gp[2 * st] = gka;
                    gp[6 * st] = gkb;
                    ws[l.o_gT + (m - l.sym_lo)] = gt;
                }
            }
        }
    }

    // ---- symbols at the ends of the sequence: m < kmax, or m >= W / S (some window (m - x) S + r does not exist); one
    //      workgroup per symbol, threads over (phase, tap, tap)
    const int64_t tail0 = l.W > 0 ? l.W / S : 0;
    const int64_t head_hi = own_hi < l.kmax ? own_hi : l.kmax;
    const int64_t nh = head_hi > l.own_lo ? head_hi - l.own_lo : 0;
    int64_t t0 = tail0 > l.kmax ? tail0 : l.kmax;
    if (t0 < l.own_lo) t0 = l.own_lo;
    const int64_t nt = own_hi > t0 ? own_hi - t0 : 0;
    for (int64_t e = bid; e < nh + nt; e += nb) {
        const int64_t m = e < nh ? l.own_lo + e : t0 + (e - nh);
        float gr = 0.f, gi = 0.f, g3 = 0.f, gt = 0.f;
        for (int it = threadIdx.x; it < S * LE * LE; it += NT) {
            const int r = it / (LE * LE), rem = it - r * LE * LE;
            const int x = rem / LE, j = rem - x * LE;
            const int64_t n = (m - x) * S + r;
            if (n < 0 || n >= l.W) continue;
            // stacked tables [phase][output][tap][tap]: outputs 0 / 1 restricted to the taps of stream 0 are Re / Im; the real model
            // has one output and no imaginary part
            const int no = l.cx_real ? 1 : 2;
            const float* Hr = ws + l.o_Hc + (size_t)(r * no + 0) * LTp * LTp;
            const float* Hi = ws + l.o_Hc + (size_t)(r * no + 1) * LTp * LTp;
            const float dr = Hr[x * LTp + x], di = l.cx_real ? 0.f : Hi[x * LTp + x];
            const float hr = Hr[x * LTp + j], hi = l.cx_real ? 0.f : Hi[x * LTp + j];
            const int64_t mm = m - x + j;
            float ua = 0.f, ub = 0.f, sv = 0.f;
            if (mm >= l.sym_lo && mm < sym_hi) {
                const float* g = mom + (mm - l.sym_lo);
                ua = g[0]; ub = g[4 * st]; sv = g[st] + g[5 * st];
            }
            const float cr = hr * ua - hi * ub, ci = hi * ua + hr * ub;
            gr += 4.f * (dr * cr + di * ci);
            gi += 4.f * (dr * ci - di * cr);
            gt += 4.f * (hr * hr + hi * hi) * sv;
            if (j == 0) {
                const float h1r = ws[l.o_h1c + (size_t)(r * no + 0) * LTp + x];
                const float h1i = l.cx_real ? 0.f : ws[l.o_h1c + (size_t)(r * no + 1) * LTp + x];
                gr += 2.f * (dr * h1r + di * h1i);
                gi += 2.f * (dr * h1i - di * h1r);
                g3 += dr * dr + di * di;
            }
        }
        for (int off = 32; off > 0; off >>= 1) {
            gr += __shfl_down(gr, off, 64); gi += __shfl_down(gi, off, 64); g3 += __shfl_down(g3, off, 64);
            gt += __shfl_down(gt, off, 64);
        }
        __syncthreads();
        if (lane == 0) { s_e[wv][0] = gr; s_e[wv][1] = gi; s_e[wv][2] = g3; s_e[wv][3] = gt; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float a = 0.f, b = 0.f, c = 0.f, d = 0.f;
            for (int w = 0; w < NT / 64; ++w) { a += s_e[w][0]; b += s_e[w][1]; c += s_e[w][2]; d += s_e[w][3]; }
            if (l.cx_split) {
                reinterpret_cast<f32x4*>(ws + l.o_gpk)[(int64_t)S * st + (m - l.sym_lo)] = f32x4{a, b, d, c};
            } else {
                float* gp = gmom + (m - l.sym_lo);
                gp[2 * st] = a; gp[6 * st] = b; gp[3 * st] = c; gp[7 * st] = c;
                ws[l.o_gT + (m - l.sym_lo)] = d;
            }
        }
    }
}

// raw gradients of the complex channel kernels from the per-phase totals (body: v2vae_chain_cx.h)
__global__ __launch_bounds__(NT) void k_chain_channel_cx(Lay l, const float* __restrict__ ws, const double* __restrict__ scal,
                                                         const float* __restrict__ grad_scale, float* __restrict__ g_h1,
                                                         float* __restrict__ g_h2) {
    d_chain_channel_cx(l, ws, scal, grad_scale, g_h1, g_h2, (int)blockIdx.x, (int)gridDim.x);
}

template <typename K>
static int launch_cx(K kernel, const Lay& l, int lds, const float* y, float* ws, hipStream_t st) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
        return V2VAE_ELAUNCH;
    hipLaunchKernelGGL(kernel, dim3(l.nblk_cx), dim3(8 * 64), lds, st, l, y, ws);
    return V2VAE_OK;
}

int launch_elbo_cx(const Lay& l, const float* y, float* ws, hipStream_t st) {
    if (!l.cx_split) {   // the block-tile kernel (V2VAE_FLAG_ELBO_TILES)
        if (l.S == 1) return launch_cx(k_elbo_cx2<1, 8>, l, l.lds_cx, y, ws, st);
        return launch_cx(k_elbo_cx2<2, 8>, l, l.lds_cx, y, ws, st);
    }
    const bool k13 = l.LE <= 13;
    const bool bfx = cx3_bfx(l.S, l.cx_real) && !(l.flags & V2VAE_FLAG_F32_MATRIX);
    const int lds = cx3_lds_floats(l.S, k13 ? 13 : 16, bfx) * 4;
#define V2VAE_CX3(SS, RR, BB) (k13 ? launch_cx(k_elbo_cx3<SS, 13, RR, BB>, l, lds, y, ws, st) : launch_cx(k_elbo_cx3<SS, 16, RR, BB>, l, lds, y, ws, st))
    if (l.cx_real) return l.S == 4 ? V2VAE_CX3(4, true, false) : l.S == 1 ? V2VAE_CX3(1, true, false) : V2VAE_CX3(2, true, false);
    if (l.S == 4) return V2VAE_CX3(4, false, false);
    if (bfx) return l.S == 1 ? V2VAE_CX3(1, false, true) : V2VAE_CX3(2, false, true);
    return l.S == 1 ? V2VAE_CX3(1, false, false) : V2VAE_CX3(2, false, false);
#undef V2VAE_CX3
}

// what needs only the moments and runs right after the soft demapper: lagged products of s = v_a + v_b, and the gradients
// with respect to 3rd moments and 4th cumulants
void launch_psum_cx(const Lay& l, float* ws, hipStream_t st) {
    const int64_t nt = (l.own_cnt + G2_TILE - 1) / G2_TILE;
    const int ng = (int)(nt < 4096 ? (nt > 0 ? nt : 1) : 4096);
    if (l.use_cxb) {
        hipLaunchKernelGGL(k_psum_cx<CX_MAXLAG>, dim3(l.nblk_psum), dim3(NT), 0, st, l, ws);
        hipLaunchKernelGGL(k_g2_cx<CX_MAXLAG>, dim3(ng + G2_NE), dim3(NT), 0, st, l, ws);
    } else {
        hipLaunchKernelGGL(k_psum_cx<16>, dim3(l.nblk_psum), dim3(NT), 0, st, l, ws);
        hipLaunchKernelGGL(k_g2_cx<16>, dim3(ng + G2_NE), dim3(NT), 0, st, l, ws);   // + G2_NE: the lagged-product term of E
    }
}

void launch_chain_channel_cx(const Lay& l, float* ws, const double* scal, const float* grad_scale, float* g_h1, float* g_h2,
                             hipStream_t st) {
    const int n = (l.cx_real ? 1 : 2) * (l.L + l.L * l.L);
    hipLaunchKernelGGL(k_chain_channel_cx, dim3((n + NT - 1) / NT), dim3(NT), 0, st, l, (const float*)ws, scal, grad_scale, g_h1, g_h2);
}

}  // namespace v2vae
