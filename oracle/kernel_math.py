"""numpy emulation of the arithmetic the HIP kernels perform.  TEST INFRASTRUCTURE ONLY.

The torch oracle (v2vae_oracle.py) gives values and autograd gradients; this file restates the
*hand-derived* fused forward+backward in the exact data layout the kernels use (polyphase
compact tap tables, per-window vectors, gather of per-symbol gradients, chain rules from the
real-stacked effective kernels back to the raw parameters), so that the derivation itself is
checked on the CPU against autograd before it is trusted on the GPU.

Layout conventions (shared with volterra_vae_amd/csrc/v2vae_kernels.hip):
  * NS real streams (1 = PAM, 2 = I/Q); outputs o in [0, NS).
  * window n = S*t + r (phase r); compact tap kk in [0, LE) of stream c reads symbol t + kk and
    stands for channel tap i = r + L - 1 - kk*S (coefficient zero when i is outside [0, L)).
  * stacked compact index k = c*LEp + kk, LEp = LE rounded up to a multiple of 4.
"""
from __future__ import annotations

import numpy as np

EPS = 1e-12


def round_up(x, m):
    return (x + m - 1) // m * m


def delays(L):
    return (L - 1) // 2, -(L - 1) // 2


# ------------------------------------------------------------------------------------------
# raw parameters -> effective real-stacked kernels
# ------------------------------------------------------------------------------------------
def stacked_channel(mode, h1, h2):
    """mode 0: real (h1 [L], h2 [L,L]);  mode 1: complex holomorphic (h1 [2,L], h2 [2,L,L]).
    Returns h1s [NS, NS*L], Hs [NS, NS*L, NS*L], dqs [NS, NS*L]."""
    if mode == 0:
        H = h2 + h2.T
        return h1[None, :].copy(), H[None], np.diag(h2)[None, :].copy()
    hr, hi = h1
    Hr, Hi = h2[0] + h2[0].T, h2[1] + h2[1].T
    h1s = np.stack((np.concatenate((hr, -hi)), np.concatenate((hi, hr))))
    Hs = np.stack((np.block([[Hr, -Hi], [-Hi, -Hr]]), np.block([[Hi, Hr], [Hr, -Hi]])))
    dqs = np.stack([np.diag(Hs[0]), np.diag(Hs[1])])
    return h1s, Hs, dqs


def stacked_equaliser_kernels(mode, k1, k2):
    """Returns K1s [NS, NS, l1], K2s [NS, NS*l2, NS*l2] (l2 may be 0)."""
    if mode == 0:
        K2 = (k2 + k2.T)[None] if k2 is not None and k2.size else np.zeros((1, 0, 0))
        return k1[None, None, :].copy(), K2
    kr, ki = k1
    K1s = np.stack((np.stack((kr, -ki)), np.stack((ki, kr))))
    if k2 is None or k2.size == 0:
        return K1s, np.zeros((2, 0, 0))
    Kr, Ki = k2[0] + k2[0].T, k2[1] + k2[1].T
    K2s = np.stack((np.block([[Kr, -Ki], [-Ki, -Kr]]), np.block([[Ki, Kr], [Kr, -Ki]])))
    return K1s, K2s


def polyphase_tables(h1s, Hs, dqs, L, S):
    """Compact per-phase tables.  Returns dict with LE, LEp, LTp and arrays
    h1c [S,NS,LTp], Hc [S,NS,LTp,LTp], dc [S,NS,LTp], dqc [S,NS,LTp]."""
    NS = h1s.shape[0]
    kmax = (S - 1 + L - 1) // S
    LE = kmax + 1
    LEp = round_up(LE, 4)
    LTp = NS * LEp
    h1c = np.zeros((S, NS, LTp))
    Hc = np.zeros((S, NS, LTp, LTp))
    dc = np.zeros((S, NS, LTp))
    dqc = np.zeros((S, NS, LTp))
    tap = np.full((S, LTp), -1, dtype=np.int64)  # stacked original index c*L+i or -1
    for r in range(S):
        for c in range(NS):
            for kk in range(LE):
                i = r + L - 1 - kk * S
                if 0 <= i < L:
                    tap[r, c * LEp + kk] = c * L + i
        for o in range(NS):
            for k in range(LTp):
                if tap[r, k] < 0:
                    continue
                h1c[r, o, k] = h1s[o, tap[r, k]]
                dc[r, o, k] = Hs[o, tap[r, k], tap[r, k]]
                dqc[r, o, k] = dqs[o, tap[r, k]]
                for j in range(LTp):
                    if tap[r, j] >= 0:
                        Hc[r, o, k, j] = Hs[o, tap[r, k], tap[r, j]]
    return dict(LE=LE, LEp=LEp, LTp=LTp, kmax=kmax, h1c=h1c, Hc=Hc, dc=dc, dqc=dqc, tap=tap)


# ------------------------------------------------------------------------------------------
# per-symbol soft demapping, moments, and its backward
# ------------------------------------------------------------------------------------------
def demap_moments(xhat, levels, prior, scaling, nv, pre, end_excl):
    """xhat [NS,B]; levels/prior/scaling [NS,C].  Returns q [NS,C,B], mom [NS,4,B] (mean, variance,
    3rd central moment, 4th cumulant) and KL (sum over symbols b in [pre, end_excl))."""
    NS, B = xhat.shape
    logits = -(xhat[:, None, :] - levels[:, :, None]) ** 2 / (nv * scaling[:, :, None])
    logits = logits - logits.max(axis=1, keepdims=True)
    ex = np.exp(logits)
    q = ex / ex.sum(axis=1, keepdims=True)
    mu = (q * levels[:, :, None]).sum(axis=1)
    dl = levels[:, :, None] - mu[:, None, :]
    var = (q * dl ** 2).sum(axis=1)
    kap = (q * dl ** 3).sum(axis=1)
    gam = (q * dl ** 4).sum(axis=1) - 3 * var ** 2
    mom = np.stack((mu, var, kap, gam), axis=1)
    sl = slice(pre, end_excl)
    kl = np.sum(q[:, :, sl] * np.log(q[:, :, sl] / prior[:, :, None] + EPS))
    return q, mom, kl


def symbol_backward(xhat, levels, prior, scaling, nv, gmom, lam, pre, end_excl, gscale=1.0):
    """Gradient of loss = KL + W log E w.r.t. xhat and scaling given gmom = dE/dmom [NS,4,B]
    and lam = W / E."""
    NS, B = xhat.shape
    q, mom, _ = demap_moments(xhat, levels, prior, scaling, nv, pre, end_excl)
    mu, var, kap = mom[:, 0], mom[:, 1], mom[:, 2]
    dl = levels[:, :, None] - mu[:, None, :]
    gmu, gv, gk, gg = (gmom[:, i][:, None, :] for i in range(4))
    dq = lam * (gmu * dl + gv * dl ** 2 + gk * (dl ** 3 - 3 * var[:, None, :] * dl)
                + gg * (dl ** 4 - 4 * kap[:, None, :] * dl - 6 * var[:, None, :] * dl ** 2))
    inslice = np.zeros(B)
    inslice[pre:end_excl] = 1.0
    dq = dq + inslice[None, None, :] * (np.log(q / prior[:, :, None] + EPS) + q / (q + EPS * prior[:, :, None]))
    dq = dq * gscale
    dz = q * (dq - (q * dq).sum(axis=1, keepdims=True))
    diff = xhat[:, None, :] - levels[:, :, None]
    den = nv * scaling[:, :, None]
    gx = (dz * (-2 * diff / den)).sum(axis=1)
    gs = (dz * diff ** 2 / (den * scaling[:, :, None])).sum(axis=2)
    return gx, gs


# ------------------------------------------------------------------------------------------
# fused ELBO forward + backward in polyphase window form
# ------------------------------------------------------------------------------------------
def elbo_fused(y, mom, tabs, L, S, own=None):
    """y [NS,N], mom [NS,4,B].  Returns E, gmom [NS,4,B], and per-phase parameter partials
    G [S,NS,LTp,LTp] (dE/dHc incl. the H.H(v v^T) term), gh1 [S,NS,LTp], gd, gdq.
    `own` = (lo, hi): only windows with t in [lo, hi) are counted in E and the parameter partials,
    while gmom is complete for symbols in [lo, hi) (the halo-sharding contract)."""
    NS, N = y.shape
    B = mom.shape[2]
    W = N - L + 1
    pre, _ = delays(L)
    LE, LEp, LTp, kmax = tabs["LE"], tabs["LEp"], tabs["LTp"], tabs["kmax"]
    lo, hi = (0, B) if own is None else own
    E = 0.0
    gmom = np.zeros_like(mom)
    G = np.zeros((S, NS, LTp, LTp))
    gh1 = np.zeros((S, NS, LTp))
    gd = np.zeros((S, NS, LTp))
    gdq = np.zeros((S, NS, LTp))
    mpad = np.zeros((NS, 4, B + 2 * kmax + 8))  # symbol m stored at m + kmax
    mpad[:, :, kmax:kmax + B] = mom
    for r in range(S):
        ts = np.arange(lo - kmax, hi)
        n = S * ts + r
        valid = (n >= 0) & (n < W)
        ownw = valid & (ts >= lo)
        T = len(ts)
        M = np.zeros((4, T, LTp))
        for c in range(NS):
            for kk in range(LE):
                M[:, :, c * LEp + kk] = mpad[c, :, ts + kk + kmax].T
        MU, V, KP, GM = M
        yw = np.zeros((NS, T))
        yw[:, valid] = y[:, pre + n[valid]]
        Dmu = np.zeros((T, LTp))
        Dv = np.zeros((T, LTp))
        Dk = np.zeros((T, LTp))
        Dg = np.zeros((T, LTp))
        P = np.einsum("t,tk,tj->kj", ownw.astype(float), V, V)
        for o in range(NS):
            H = tabs["Hc"][r, o]
            h1 = tabs["h1c"][r, o]
            d = tabs["dc"][r, o]
            dq = tabs["dqc"][r, o]
            u = MU @ H
            tt = V @ (H * H)
            z = h1[None, :] + 2 * u
            a = MU @ h1
            quad = np.sum(MU * u, axis=1)
            s = V @ d
            sq = V @ dq
            A = a + quad + s
            e = A - yw[o]
            vs = np.sum(V * z * z + 2 * V * tt + 2 * KP * d[None, :] * z + GM * d[None, :] ** 2, axis=1)
            En = e * e + 2 * yw[o] * (s - sq) + vs
            E += np.sum(En[ownw])
            w = 2 * V * z + 2 * KP * d[None, :]
            hw = w @ H
            vm = valid.astype(float)[:, None]
            Dmu += vm * (2 * e[:, None] * z + 2 * hw)
            Dv += vm * (2 * A[:, None] * d[None, :] - 2 * yw[o][:, None] * dq[None, :] + z * z + 4 * tt)
            Dk += vm * (2 * d[None, :] * z)
            Dg += vm * (d[None, :] ** 2)
            om = ownw.astype(float)
            bvec = 2 * (e[:, None] * MU + w)
            G[r, o] = np.einsum("t,tk,tj->kj", om, bvec, MU) + 4 * H * P
            gh1[r, o] = om @ (2 * e[:, None] * MU + w)
            gd[r, o] = om @ (2 * A[:, None] * V + 2 * KP * z + 2 * GM * d[None, :])
            gdq[r, o] = om @ (-2 * yw[o][:, None] * V)
        # gather per-symbol gradients: symbol m = t + kk
        for c in range(NS):
            for kk in range(LE):
                m = ts + kk
                ok = (m >= lo) & (m < hi)
                for qi, D in enumerate((Dmu, Dv, Dk, Dg)):
                    np.add.at(gmom[c, qi], m[ok], D[ok, c * LEp + kk])
    return E, gmom, dict(G=G, gh1=gh1, gd=gd, gdq=gdq)


def finish_channel_grads(mode, parts, tabs, L, S, lam):
    """Chain per-phase compact partials back to the raw channel parameters; returns (g_h1, g_h2)
    in the raw layout of `stacked_channel`, already multiplied by lam = W / E."""
    NS = parts["G"].shape[1]
    tap = tabs["tap"]
    LTp = tabs["LTp"]
    GS = np.zeros((NS, NS * L, NS * L))
    g1 = np.zeros((NS, NS * L))
    gdq = np.zeros((NS, NS * L))
    for r in range(S):
        for k in range(LTp):
            if tap[r, k] < 0:
                continue
            ik = tap[r, k]
            for o in range(NS):
                g1[o, ik] += parts["gh1"][r, o, k]
                GS[o, ik, ik] += parts["gd"][r, o, k]
                gdq[o, ik] += parts["gdq"][r, o, k]
                for j in range(LTp):
                    if tap[r, j] >= 0:
                        GS[o, ik, tap[r, j]] += parts["G"][r, o, k, j]
    if mode == 0:
        g_h2 = GS[0] + GS[0].T + np.diag(gdq[0])
        return lam * g1[0], lam * g_h2
    for o in range(NS):
        GS[o] += np.diag(gdq[o])
    a, b = slice(0, L), slice(L, 2 * L)
    gHr = GS[0][a, a] - GS[0][b, b] + GS[1][a, b] + GS[1][b, a]
    gHi = -GS[0][a, b] - GS[0][b, a] + GS[1][a, a] - GS[1][b, b]
    g_h2 = np.stack((gHr + gHr.T, gHi + gHi.T))
    g_h1 = np.stack((g1[0][a] + g1[1][b], -g1[0][b] + g1[1][a]))
    return lam * g_h1, lam * g_h2


# ------------------------------------------------------------------------------------------
# equaliser forward / backward on stacked kernels
# ------------------------------------------------------------------------------------------
def _lag_vectors(y, l2, S, B):
    NS, N = y.shape
    p2 = l2 // 2
    ypad = np.zeros((NS, N + 2 * l2 + S * B))
    ypad[:, l2:l2 + N] = y
    X = np.zeros((B, NS * l2))
    for c in range(NS):
        for j in range(l2):
            X[:, c * l2 + j] = ypad[c, l2 + np.arange(B) * S + p2 - j]
    return X


def equaliser_forward(y, K1s, K2s, S, pad1, B):
    NS, N = y.shape
    l1 = K1s.shape[2]
    ypad = np.zeros((NS, N + 2 * l1 + S * B))
    ypad[:, l1:l1 + N] = y
    out = np.zeros((NS, B))
    for o in range(NS):
        for c in range(NS):
            for j in range(l1):
                out[o] += K1s[o, c, j] * ypad[c, l1 + np.arange(B) * S + j - pad1]
    if K2s.shape[1]:
        l2 = K2s.shape[1] // NS
        X = _lag_vectors(y, l2, S, B)
        out += np.einsum("bi,oij,bj->ob", X, K2s, X)
    return out


def equaliser_backward(mode, y, gx, l1, l2, S, pad1):
    """gx [NS,B] -> raw gradients (g_k1, g_k2) in the layout of `stacked_equaliser_kernels`."""
    NS, N = y.shape
    B = gx.shape[1]
    ypad = np.zeros((NS, N + 2 * l1 + S * B))
    ypad[:, l1:l1 + N] = y
    GK1 = np.zeros((NS, NS, l1))
    for o in range(NS):
        for c in range(NS):
            for j in range(l1):
                GK1[o, c, j] = gx[o] @ ypad[c, l1 + np.arange(B) * S + j - pad1]
    GK2 = np.zeros((NS, NS * l2, NS * l2))
    if l2:
        X = _lag_vectors(y, l2, S, B)
        GK2 = np.einsum("ob,bi,bj->oij", gx, X, X)
    if mode == 0:
        return GK1[0, 0], (GK2[0] + GK2[0].T if l2 else None)
    g_k1 = np.stack((GK1[0, 0] + GK1[1, 1], -GK1[0, 1] + GK1[1, 0]))
    if not l2:
        return g_k1, None
    a, b = slice(0, l2), slice(l2, 2 * l2)
    gKr = GK2[0][a, a] - GK2[0][b, b] + GK2[1][a, b] + GK2[1][b, a]
    gKi = -GK2[0][a, b] - GK2[0][b, a] + GK2[1][a, a] - GK2[1][b, b]
    return g_k1, np.stack((gKr + gKr.T, gKi + gKi.T))


def full_step_grads(mode, y, params, levels, prior, scaling, nv, S, pad1=None):
    """Everything end to end: returns dict(loss, E, KL, xhat, grads...) for one batch."""
    h1, h2, k1, k2 = params
    NS, N = y.shape
    L = h1.shape[-1]
    l1 = k1.shape[-1]
    l2 = 0 if k2 is None else k2.shape[-1]
    B = N // S
    W = N - L + 1
    pre, end = delays(L)
    if pad1 is None:
        pad1 = l1 // 2
    K1s, K2s = stacked_equaliser_kernels(mode, k1, k2)
    xhat = equaliser_forward(y, K1s, K2s, S, pad1, B)
    q, mom, kl = demap_moments(xhat, levels, prior, scaling, nv, pre, B + end)
    h1s, Hs, dqs = stacked_channel(mode, h1, h2)
    tabs = polyphase_tables(h1s, Hs, dqs, L, S)
    E, gmom, parts = elbo_fused(y, mom, tabs, L, S)
    lam = W / E
    g_h1, g_h2 = finish_channel_grads(mode, parts, tabs, L, S, lam)
    gx, gs = symbol_backward(xhat, levels, prior, scaling, nv, gmom, lam, pre, B + end)
    g_k1, g_k2 = equaliser_backward(mode, y, gx, l1, l2, S, pad1)
    return dict(loss=kl + W * np.log(E), E=E, KL=kl, xhat=xhat, q=q, mom=mom, gmom=gmom, gx=gx,
                g_h1=g_h1, g_h2=g_h2, g_k1=g_k1, g_k2=g_k2, g_scaling=gs)
