// Chain rules from the summed records to the raw parameter gradients on the matrix-core paths, as device functions of
// (block id, block count): each has a kernel of its own, and one launch runs both when a step needs both.
#ifndef V2VAE_CHAIN_CX_H
#define V2VAE_CHAIN_CX_H
#include <hip/hip_runtime.h>

#include "../../include/v2vae_hip.h"
#include "v2vae_layout.h"

namespace v2vae {

// raw gradients of the complex channel kernels from the per-phase totals
__device__ __forceinline__ void d_chain_channel_cx(const Lay& l, const float* __restrict__ ws, const double* __restrict__ scal,
                                                   const float* __restrict__ grad_scale, float* __restrict__ g_h1,
                                                   float* __restrict__ g_h2, int bid, int nb) {
    const float lam = (float)scal[V2VAE_SCAL_LAMBDA] * (grad_scale ? grad_scale[0] : 1.f);
    const int np = l.cx_real ? 1 : 2;   // parameter planes (Re, Im; the real model has one)
    const int L = l.L, n1 = np * L, n2 = np * L * L;
    auto compact = [&](int tap, int* r, int* x) {
        int rr = (tap - (L - 1)) % l.S;
        if (rr < 0) rr += l.S;
        *r = rr;
        *x = (rr + L - 1 - tap) / l.S;
    };
    for (int idx = bid * NT + threadIdx.x; idx < n1 + n2; idx += nb * NT) {
        if (idx < n1) {
            if (!g_h1) continue;
            const int p = idx / L, tap = idx - p * L;
            int r, x;
            compact(tap, &r, &x);
            g_h1[idx] = lam * ws[l.o_tot_cx + (size_t)r * CX_REC + 512 + p * 16 + x];
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
                const float* tot = ws + l.o_tot_cx + (size_t)ri * CX_REC;
                const float* Hp = ws + l.o_cxH + (size_t)(ri * 2 + p) * 256;
                const float ps = ws[l.o_psv + ((size_t)ri * l.LE + xi) * l.LE + xj];   // lagged sum of the tap pair (k_g2_cx)
                auto G = [&](int a, int b) {
                    float g = tot[p * 256 + a * 16 + b];
                    g += 4.f * Hp[a * 16 + b] * ps;
                    if (a == b) g += tot[512 + (2 + p) * 16 + a];
                    return g;
                };
                v = G(xi, xj) + G(xj, xi);
            }
            g_h2[e] = lam * v;
        }
    }
}

// raw gradients of the equaliser kernels from the summed gradient planes of k_eq_bwd_cx
__device__ __forceinline__ void d_chain_equaliser_cx(const Lay& l, const float* __restrict__ ws, float* __restrict__ g_k1,
                                                     float* __restrict__ g_k2, int bid, int nb) {
    const float* T = ws + l.o_tot_eqcx;
    const int l1 = l.l1, l2 = l.l2, LD = 16 * l.eqb16, PL = LD * LD;
    const int n1 = l.ns * l1, n2 = l.ns * l2 * l2;   // two parameter planes (Re, Im); the real model has one
    auto GK = [&](int p, int a, int b) {  // dE/dK (effective symmetric kernel), plane p: a lower block is its partner transposed
        return (a >> 4) > (b >> 4) ? T[p * PL + b * LD + a] : T[p * PL + a * LD + b];
    };
    for (int idx = bid * NT + threadIdx.x; idx < n1 + n2; idx += nb * NT) {
        if (idx < n1) {
            if (!g_k1) continue;
            const int p = idx / l1, jj = idx - p * l1;
            const int x = l.pad1 + l.p2 - jj;  // lag row that first-order tap jj multiplies (x < l2)
            g_k1[idx] = T[p * PL + x * LD + l2];
        } else {
            if (!g_k2) continue;
            const int e = idx - n1;
            const int p = e / (l2 * l2), rem = e - p * l2 * l2;
            const int a = rem / l2, b = rem - a * l2;
            g_k2[e] = GK(p, a, b) + GK(p, b, a);
        }
    }
}

}  // namespace v2vae
#endif
