// Tables of the complex equaliser forward kernels from the raw kernels, as a device function of (block id, block count): it has
// a kernel of its own, and the launch that prepares the other tables of a step runs it in extra workgroups.
#ifndef V2VAE_PREP_EQ_H
#define V2VAE_PREP_EQ_H
#include <hip/hip_runtime.h>

#include "v2vae_bf16_split.h"
#include "v2vae_layout.h"

namespace v2vae {

constexpr int PREP_EQCXB_BLOCKS = 32;   // workgroups of the equaliser-table part

// A-operand table of the forward: tab[step][row block][plane (Zr, Zi)][64 lanes], followed by the first-order kernel on lag
// indices: k1w[plane][NBL * 32] with k1w[x] = k1[pad1 + p2 - x]
__device__ __forceinline__ void d_prep_eqcxb(const Lay& l, const float* __restrict__ k1, const float* __restrict__ k2,
                                             float* __restrict__ ws, int bid, int nb) {
    const int l1 = l.l1, l2 = l.l2, p1 = l.pad1, p2 = l.p2, LK = l.eqcx_lk, NBL = l.eqcxb_nbl;
    const float* k1r = k1;
    const float* k1i = k1 + l1;
    const float* k2r = k2;
    const float* k2i = k2 + (size_t)l2 * l2;
    const int n = l.eqcx_steps * NBL * 2 * 64;
    for (int idx = bid * NT + threadIdx.x; idx < n; idx += nb * NT) {
        const int lane = idx & 63, plane = (idx >> 6) & 1, jb = (idx >> 7) % NBL, s = idx / (128 * NBL);
        const int j = jb * 32 + (lane & 31), h = lane >> 5;
        const int k = 2 * s + h, ck = k >= LK ? 1 : 0, x = k - ck * LK;  // stacked input lag (stream ck, lag x)
        float v = 0.f;
        if (x < l2) {
            if (j < l2) {
                const float kr = k2r[x * l2 + j] + k2r[j * l2 + x], ki = k2i[x * l2 + j] + k2i[j * l2 + x];
                v = ck == 0 ? (plane == 0 ? kr : ki) : (plane == 0 ? -ki : kr);  // [[Kr, Ki], [-Ki, Kr]]
            }
        }
        ws[l.o_eqcxbA + idx] = v;
    }
    for (int idx = bid * NT + threadIdx.x; idx < 2 * NBL * 32; idx += nb * NT) {
        const int plane = idx / (NBL * 32), x = idx - plane * NBL * 32;
        const int jj = p1 + p2 - x;  // first-order tap that multiplies lag x
        ws[l.o_eqcxbA + n + idx] = (x < l2 && jj >= 0 && jj < l1) ? (plane == 0 ? k1r[jj] : k1i[jj]) : 0.f;
    }
    if (l.eqfbf) {
        // table images of the forward on the bf16 matrix pipe (k_eq_fwd_bf): [pairs (R, g >= R)][stream c][piece][64 lanes][4 words
        // of two bf16].  Lane (jj, hh): matrix row jj = (Re | Im of u, lag a = 16 R + (jj & 15)); element j of the lane is the
        // k-slot of lag b = 16 g + 8 hh + 7 - j of stream c.  Entry: [[K'r, -K'i], [K'i, K'r]] with K'_ab = K_ab (a = b),
        // 2 K_ab (a < b), 0 (a > b, or a lag from l2 on), K = k2 + k2^T; each float as three exact bf16 pieces (rounded to nearest).
        const int NBK = l.eqfbf_nbk, NPW = NBK * (NBK + 1) / 2 * 1536;
        for (int idx = bid * NT + threadIdx.x; idx < NPW; idx += nb * NT) {
            const int w = idx & 3, ln = (idx >> 2) & 63, pc = (idx >> 8) % 3, c = (idx / 768) & 1;
            int pi = idx / 1536, R = 0;
            while (pi >= NBK - R) { pi -= NBK - R; ++R; }
            const int g = R + pi;
            const int jj = ln & 31, hh = ln >> 5, cjj = jj >> 4, a = 16 * R + (jj & 15);
            unsigned word = 0;
            for (int u = 0; u < 2; ++u) {
                const int b = 16 * g + 8 * hh + 7 - (2 * w + u);
                float t = 0.f;
                if (a < l2 && b < l2 && b >= a) {
                    const float sc = a == b ? 1.f : 2.f;
                    const float kr = sc * (k2r[a * l2 + b] + k2r[b * l2 + a]), ki = sc * (k2i[a * l2 + b] + k2i[b * l2 + a]);
                    t = c == 0 ? (cjj == 0 ? kr : ki) : (cjj == 0 ? -ki : kr);
                }
                word |= bf16_piece_rne(t, pc) << (16 * u);   // (split once per step: pieces rounded to nearest)
            }
            ws[l.o_eqfbf + idx] = __uint_as_float(word);
        }
        for (int idx = bid * NT + threadIdx.x; idx < 32 * NBK; idx += nb * NT) {
            const int a = idx >> 1, plane = idx & 1;
            const int jj = p1 + p2 - a;  // first-order tap that multiplies lag a
            ws[l.o_eqfbf + NPW + idx] = (a < l2 && jj >= 0 && jj < l1) ? (plane == 0 ? k1r[jj] : k1i[jj]) : 0.f;
        }
        if (l.eqfbf_xl) {   // the lag on the side, e = 16 NBK = l2 - 1: K'_ae = 2 K_ae for the rows a < e, then K'_ee and lag e's weight
            const int e = 16 * NBK;
            for (int idx = bid * NT + threadIdx.x; idx < 32 * NBK + 4; idx += nb * NT) {
                float v;
                if (idx < 32 * NBK) {
                    const int a = idx >> 1;
                    v = 2.f * ((idx & 1) ? k2i[a * l2 + e] + k2i[e * l2 + a] : k2r[a * l2 + e] + k2r[e * l2 + a]);
                } else {
                    const int q = idx - 32 * NBK, jj = p1 + p2 - e;
                    if (q < 2) v = q ? 2.f * k2i[e * l2 + e] : 2.f * k2r[e * l2 + e];   // K_ee = k2_ee + k2_ee
                    else v = (jj >= 0 && jj < l1) ? (q == 2 ? k1r[jj] : k1i[jj]) : 0.f;
                }
                ws[l.o_eqfbf + NPW + 32 * NBK + idx] = v;
            }
        }
    }
    if (l.eqsym_t) {   // table of the symmetric-form forward kernel (k_eq_fwd_sym): one stream of (Re, Im) entries, row after row,
                       // row a = [K'_aa, K'_a,a+1, ..., K'_a,T-1, first-order weight of lag a]; eight zero entries pad the end
        const int T = l.eqsym_t, ne = T * (T + 1) / 2 + T;
        for (int idx = bid * NT + threadIdx.x; idx < ne + 8; idx += nb * NT) {
            float vr = 0.f, vi = 0.f;
            if (idx < ne) {
                int a = 0, rem = idx;
                while (rem >= T - a + 1) { rem -= T - a + 1; ++a; }
                if (rem == T - a) {
                    const int jj = p1 + p2 - a;  // first-order tap that multiplies lag a
                    if (a < l2 && jj >= 0 && jj < l1) { vr = k1r[jj]; vi = k1i[jj]; }
                } else {
                    const int b = a + rem;
                    if (b < l2) {
                        const float sc = a == b ? 1.f : 2.f;   // K_ab y_a y_b + K_ba y_b y_a
                        vr = sc * (k2r[a * l2 + b] + k2r[b * l2 + a]);
                        vi = sc * (k2i[a * l2 + b] + k2i[b * l2 + a]);
                    }
                }
            }
            ws[l.o_eqsym + 2 * idx] = vr;
            ws[l.o_eqsym + 2 * idx + 1] = vi;
        }
    }
}

}  // namespace v2vae
#endif
