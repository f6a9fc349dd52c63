// V2VAE training hot path for MI355X (gfx950): equaliser forward/backward, soft demapping,
// closed-form ELBO expectations over sliding windows (fused forward + backward), reductions,
// chain rules back to raw parameters and Adam-AMSGrad.
//
// Generic path ("g" kernels): correct for every supported shape; the per-window vectors live in
// LDS, coefficient tables are read through the scalar cache (wave-uniform addresses), per-block
// partial sums are written to the workspace and reduced in a fixed order, so results are
// bit-reproducible run to run.
#include <hip/hip_runtime.h>

#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include <mutex>

#include "v2vae_chain_cx.h"
#include "v2vae_layout.h"
#include "v2vae_posterior.h"
#include "v2vae_prep_eq.h"

namespace v2vae {

static inline int64_t rup(int64_t x, int64_t m) { return (x + m - 1) / m * m; }

int make_layout(const v2vae_problem* p, Lay* out) {
    if (!p || !out) return V2VAE_EINVAL;
    Lay l{};
    l.ns = p->ns; l.S = p->sps;
    if (p->flags & ~(V2VAE_FLAG_GENERIC | V2VAE_FLAG_EQ_MFMA_FWD | V2VAE_FLAG_ELBO_TILES | V2VAE_FLAG_REAL_MFMA | V2VAE_FLAG_F32_MATRIX) || p->reserved != 0) return V2VAE_EINVAL;
    l.flags = p->flags;
    if (p->mode < 0 || p->mode > 3) return V2VAE_EINVAL;
    l.chan = p->mode >= 2 ? p->mode - 1 : 0;  // modes 2, 3: real path with a polynomial channel model
    l.mode = p->mode >= 2 ? 0 : p->mode; l.l1 = p->l1; l.pad1 = p->pad1; l.l2 = p->l2;
    l.L = p->L; l.C = p->C;
    l.n_glob = p->n_glob; l.y_lo = p->y_lo; l.y_len = p->y_len; l.sym_lo = p->sym_lo;
    l.sym_cnt = p->sym_cnt; l.own_lo = p->own_lo; l.own_cnt = p->own_cnt;
    l.y_stride = l.y_len;
    if (l.ns < 1 || l.ns > 2) return V2VAE_EINVAL;
    if ((l.mode == 0) != (l.ns == 1)) return V2VAE_EINVAL;
    if (l.S < 1 || l.S > 8) return V2VAE_EINVAL;
    if (l.L < 2 || l.S > l.L) return V2VAE_EINVAL;  // assert S <= L: lib/real_equalization.py:751
    if (l.l1 < 1 || l.l1 > 255 || l.pad1 < 0 || l.pad1 >= l.l1) return V2VAE_EINVAL;
    if (l.l2 < 0 || l.l2 > 129 || (l.l2 > 0 && (l.l2 % 2) == 0)) return V2VAE_EINVAL;  // odd lags: torch_components.py:65-66
    if (l.C < 1 || l.C > MAX_C) return V2VAE_EINVAL;
    if (l.n_glob < l.L || l.n_glob % l.S != 0) return V2VAE_EINVAL;
    l.b_glob = l.n_glob / l.S;
    l.W = l.n_glob - l.L + 1;
    l.pre = (l.L - 1) / 2;
    l.kl_lo = l.pre;
    l.kl_hi = l.b_glob - l.L / 2;  // Python -(L-1)//2 == -(L/2)
    if (l.kl_hi < l.kl_lo) l.kl_hi = l.kl_lo;
    l.kmax = (l.S - 1 + l.L - 1) / l.S;
    l.LE = l.kmax + 1;
    l.LEp = (int)rup(l.LE, 4);
    l.LTp = l.ns * l.LEp;
    l.p2 = l.l2 / 2;
    l.LT2 = l.ns * l.l2;
    l.LT2p = (int)rup(l.LT2, 4);
    {
        int left = l.pad1 > l.p2 ? l.pad1 : l.p2;
        int right1 = l.l1 - 1 - l.pad1;
        int right = right1 > l.p2 ? right1 : l.p2;
        l.hl = left + 4;  // +4: padded stacked lags read (finite) samples just left of the halo
        l.hr = right;
        l.tile_len = EQ_TB * l.S + l.hl + l.hr + 1;
    }
    // ---- range checks (halo contract)
    if (l.sym_cnt < 1 || l.own_cnt < 1 || l.y_len < 1) return V2VAE_EINVAL;
    if (l.sym_lo < 0 || l.sym_lo + l.sym_cnt > l.b_glob) return V2VAE_EINVAL;
    if (l.own_lo < l.sym_lo || l.own_lo + l.own_cnt > l.sym_lo + l.sym_cnt) return V2VAE_EINVAL;
    {
        int64_t need_lo = l.own_lo - l.kmax; if (need_lo < 0) need_lo = 0;
        int64_t need_hi = l.own_lo + l.own_cnt + l.kmax; if (need_hi > l.b_glob) need_hi = l.b_glob;
        if (l.sym_lo > need_lo || l.sym_lo + l.sym_cnt < need_hi) return V2VAE_EHALO;
        int64_t ys_lo = l.sym_lo * l.S - (l.hl - 4); if (ys_lo < 0) ys_lo = 0;
        int64_t ys_hi = (l.sym_lo + l.sym_cnt - 1) * l.S + l.hr + 1; if (ys_hi > l.n_glob) ys_hi = l.n_glob;
        // y' of every window this rank evaluates: n in [(own_lo-kmax)*S, (own_lo+own_cnt)*S), sample pre+n
        int64_t yw_lo = (l.own_lo - l.kmax) * l.S + l.pre; if (yw_lo < 0) yw_lo = 0;
        int64_t yw_hi = (l.own_lo + l.own_cnt) * l.S + l.pre; if (yw_hi > l.n_glob) yw_hi = l.n_glob;
        if (yw_lo < ys_lo) ys_lo = yw_lo;
        if (yw_hi > ys_hi) ys_hi = yw_hi;
        if (l.y_lo > ys_lo || l.y_lo + l.y_len < ys_hi) return V2VAE_EHALO;
        if (l.y_lo < 0 || l.y_lo + l.y_len > l.n_glob) return V2VAE_EINVAL;
    }
    // ---- ELBO tiling: largest window tile whose LDS image fits
    l.TWIN = 0;
    for (int tw = 256; tw >= 64; tw >>= 1) {
        int ts = tw - l.kmax;
        if (ts < tw / 4) continue;
        int tsym = tw + l.kmax + 4;
        int64_t fl = (int64_t)l.ns * 4 * tsym + 3 * l.ns * tw + 2 * tw + 2 * (int64_t)l.ns * l.LTp * tw +
                     3 * (int64_t)l.LTp * tw + (int64_t)l.ns * 5 * NT + SCRATCH_F;
        if (fl * 4 <= LDS_BUDGET) { l.TWIN = tw; l.TS = ts; l.TSYM = tsym; l.lds_elbo = (int)(fl * 4); break; }
    }
    // shapes beyond the generic ELBO kernel's tiling are still fine if a matrix-core kernel covers them (decided below)
    int generic_elbo_ok = 1;
    if (l.TWIN == 0) { generic_elbo_ok = 0; l.TWIN = 64; l.TS = 64 - l.kmax > 0 ? 64 - l.kmax : 1; l.TSYM = 64 + l.kmax + 4; l.lds_elbo = 0; }
    l.NG = NT / l.TWIN;
    l.ntile_elbo = (int)((l.own_cnt + l.TS - 1) / l.TS);
    l.nblk_elbo = l.ntile_elbo < MAX_BLOCKS ? l.ntile_elbo : MAX_BLOCKS;
    {
        int nb = l.LTp / 4;
        l.nG_elbo = l.ns * nb * nb;
        l.nitems_elbo = l.nG_elbo + nb * nb;
        if (l.nitems_elbo <= NT) { l.nchunk_elbo = NT / l.nitems_elbo; l.maxi_elbo = 1; }
        else { l.nchunk_elbo = 1; l.maxi_elbo = (l.nitems_elbo + NT - 1) / NT; }
        if (l.maxi_elbo > 4) generic_elbo_ok = 0;
        if (l.nchunk_elbo > 16) l.nchunk_elbo = 16;
        if (l.ns * l.LTp > NT) generic_elbo_ok = 0;
    }
    // ---- equaliser tiling
    int generic_eq_ok = 1;
    l.ntile_eq = (int)((l.own_cnt + EQ_TB - 1) / EQ_TB);
    l.nblk_eq = l.ntile_eq < MAX_BLOCKS ? l.ntile_eq : MAX_BLOCKS;
    {
        int nb = l.LT2p / 4;
        l.nitems_eq = l.ns * nb * nb;
        if (l.nitems_eq == 0) { l.nchunk_eq = 1; l.maxi_eq = 1; }
        else if (l.nitems_eq <= NT) { l.nchunk_eq = NT / l.nitems_eq; l.maxi_eq = 1; }
        else { l.nchunk_eq = 1; l.maxi_eq = (l.nitems_eq + NT - 1) / NT; }
        if (l.maxi_eq > 10) generic_eq_ok = 0;
        if (l.nchunk_eq > 16) l.nchunk_eq = 16;
        l.nk1 = l.ns * l.ns * l.l1;
        if (l.nk1 > 4 * NT) return V2VAE_ESIZE;
        int64_t fl = (int64_t)l.ns * l.tile_len + (int64_t)l.ns * EQ_TB + SCRATCH_F;
        if (fl < 4096 + 64) fl = 4096 + 64;  // the matrix-core backward reduces its four waves through a 16 KB image
        if (fl * 4 > LDS_BUDGET) return V2VAE_ESIZE;
        l.lds_eq = (int)(fl * 4);
    }
    {
        int64_t nb = (l.sym_cnt + NT - 1) / NT;
        // grid-stride kernels (soft demapper, per-symbol backward): as many workgroups as are resident at once - 256 CUs x
        // the waves per SIMD of their launch bounds - so that no second, partly filled round of workgroups trails the first
        const int cap = 256 * (l.C <= 2 ? 4 : l.C <= 4 ? 3 : l.C <= 8 ? 2 : 1);
        l.nblk_sym = (int)(nb < cap ? nb : cap);
        const int cap_dm = 256 * (l.C <= 4 ? 4 : l.C <= 8 ? 2 : 1);
        l.nblk_dm = (int)(nb < cap_dm ? nb : cap_dm);
    }
    // ---- matrix-core ELBO path: complex model, <= 16 active taps per stream and phase, S in {1, 2}; the run kernel also
    //      serves four samples per symbol (two waves per phase)
    const bool run_s = l.S <= 2 || (l.S == 4 && !(l.flags & V2VAE_FLAG_ELBO_TILES));
    l.use_cx = (l.mode == 1 && l.LE <= 16 && run_s && !(l.flags & V2VAE_FLAG_GENERIC)) ? 1 : 0;
    // the real model runs the same kernel on a zero quadrature stream (large batches; any batch on request)
    l.cx_real = (l.mode == 0 && l.ns == 1 && l.chan == 0 && l.LE <= 16 && run_s &&
                 !(l.flags & (V2VAE_FLAG_GENERIC | V2VAE_FLAG_ELBO_TILES)) &&
                 (l.own_cnt >= 4096 || (l.flags & V2VAE_FLAG_REAL_MFMA))) ? 1 : 0;
    if (l.cx_real) l.use_cx = 1;
    l.cx_split = (l.use_cx && !(l.flags & V2VAE_FLAG_ELBO_TILES)) ? 1 : 0;
    {
        const int ts = 256 - l.kmax;
        l.ntile_cx = (int)((l.own_cnt + ts - 1) / ts);
        l.nblk_cx = l.ntile_cx < CX_BLOCKS ? l.ntile_cx : CX_BLOCKS;
        if (l.cx_split) {
            // runs: CX_BLOCKS workgroups x (8 / S) waves per phase; a run is a multiple of 32 symbols
            const int nwp = 8 / l.S;
            int64_t rl = (l.own_cnt + (int64_t)CX_BLOCKS * nwp - 1) / ((int64_t)CX_BLOCKS * nwp);
            rl = rup(rl < 64 ? 64 : rl, 32);
            l.cx_rl = rl;
            const int64_t nruns = (l.own_cnt + rl - 1) / rl;
            l.nblk_cx = (int)((nruns + nwp - 1) / nwp);
        }
        l.lds_cx = ((7 * (256 + 48) + 2 * l.S * 256) + l.S * 40 * 64 + l.S * 64 + 8 * 32 * CX_TP + 8 * 4 * (16 * 32 + 4) + l.S * 8 * 4 * 64) * 4;
        // blocked variant for longer channels
        l.use_cxb = (l.mode == 1 && l.LE > 16 && l.LE <= 48 && l.S <= 2 && !(l.flags & V2VAE_FLAG_GENERIC)) ? 1 : 0;
        l.cxb_nb = (l.LE + 15) / 16;
        l.cxb_xt = (l.LE % 16 == 1 && !(l.flags & V2VAE_FLAG_ELBO_TILES)) ? 1 : 0;
        l.cxb_nbk = l.cxb_nb - l.cxb_xt;
        l.cxb_leb = 16 * l.cxb_nb;
        l.cxb_rec = 2 * l.cxb_leb * l.cxb_leb + 4 * l.cxb_leb;
        {
            const int CXB_TWB = cxb_twb(l.cxb_nbk, l.cxb_xt), CXB_NW = cxb_nw(l.cxb_nbk, l.cxb_xt);
            l.cxb_twbv = CXB_TWB;
            l.cxb_nwv = CXB_NW;
            const int tsb = CXB_TWB - l.kmax;
            l.ntile_cxb = l.use_cxb ? (int)((l.own_cnt + tsb - 1) / tsb) : 0;
            l.nblk_cxb = l.ntile_cxb < CX_BLOCKS ? l.ntile_cxb : CX_BLOCKS;
            const int ms = CXB_TWB + l.cxb_leb + 16, nsig = 32 + l.cxb_leb;
            l.lds_cxb = ((7 * ms + 2 * l.S * CXB_TWB) + l.S * 4 * l.cxb_leb + CXB_NW * 32 * (l.cxb_xt ? CXB_TPX : CX_TP) +
                         l.S * (CXB_TWB / 32) * 4 * nsig + (l.S * l.cxb_rec > CXB_NW * 32 * 81 ? l.S * l.cxb_rec : CXB_NW * 32 * 81) +
                         (l.cxb_xt ? l.S * 2 * l.cxb_leb : 0) + 3 * ms) * 4;   // records / per-wave [32][81] transposing buffers share memory; bf16 pieces of the means
        }
        l.nblk_E = l.use_cx ? l.nblk_cx + G2_NE : (l.use_cxb ? l.nblk_cxb + G2_NE : l.nblk_elbo);   // + 1: the lagged-product term of E (k_g2_cx)
        if (!l.use_cx && !l.use_cxb && !generic_elbo_ok) return V2VAE_ESIZE;
        l.generic_elbo_ok = generic_elbo_ok;
        int64_t nb = (l.own_cnt + NT - 1) / NT;
        l.nblk_psum = (int)(nb < 2048 ? nb : 2048);
    }
    // ---- matrix-core equaliser path: complex, second-order lags odd <= 31, centred first-order window inside it
    l.use_eqcx = (l.mode == 1 && l.l2 >= 1 && l.l2 <= 31 && (l.l1 & 1) && l.l1 <= l.l2 && l.pad1 == l.l1 / 2 &&
                  !(l.flags & V2VAE_FLAG_GENERIC)) ? 1 : 0;
    l.eqcx_lk = (l.l2 + 2) & ~1;           // lags per stream, even and > l2 (the spare lag carries a zero coefficient)
    l.eqcx_steps = l.eqcx_lk;              // 2 streams * lk lags / 2 per MFMA step
    // blocked variant for longer kernels
    l.use_eqcxb = (l.mode == 1 && l.l2 > 31 && l.l2 <= 95 && (l.l1 & 1) && l.l1 <= l.l2 && l.pad1 == l.l1 / 2 &&
                   !(l.flags & V2VAE_FLAG_GENERIC)) ? 1 : 0;
    l.eqcxb_nbl = (l.l2 + 32) / 32;        // 32-row blocks holding lags 0 .. l2 (row l2 carries the first-order kernel)
    l.eqcxb_fwd = (l.use_eqcxb || l.use_eqcx) ? 1 : 0;  // the blocked forward kernel also serves the one-block case
    l.eqcxb_rec = 2 * (l.eqcxb_nbl * (l.eqcxb_nbl + 1) / 2) * 1024;
    l.lds_eqcxb_fwd = (int)((rup((int64_t)l.ns * l.tile_len, 64) + (int64_t)l.eqcx_steps * l.eqcxb_nbl * 128 + l.eqcxb_nbl * 64) * 4);
    {
        const int64_t a = (int64_t)l.ns * l.tile_len + 2 * EQ_TB, b = l.eqcxb_rec;
        l.lds_eqcxb_bwd = (int)((a > b ? a : b) * 4);
    }
    if (l.use_eqcxb && l.lds_eqcxb_fwd > LDS_BUDGET) l.use_eqcxb = 0;
    // persistent backward kernels: exactly the workgroups that are resident at once (register use: 4 per CU for the
    // single-block kernel, 2 with two lag blocks, 1 with three), so that no partly filled second round trails the first
    // (the instance on the bf16 matrix pipe reaches 95 lags - six blocks -, the f32 one 79)
    const bool eqbf_ok = l.S <= 2 && !(l.flags & V2VAE_FLAG_F32_MATRIX);
    l.eqb16 = (l.use_eqcx || (l.use_eqcxb && l.l2 <= (eqbf_ok ? 95 : 79))) ? (l.l2 + 1 + 15) / 16 : 0;
    // the real model's equaliser BACKWARD runs the same kernel on a zero quadrature stream wherever its ELBO does (large batches;
    // any batch on request); the forward keeps the generic kernel, whose sums in double carry the remainder the demapper needs
    if (l.cx_real && l.l2 >= 1 && l.l2 <= 79 && (l.l1 & 1) && l.l1 <= l.l2 && l.pad1 == l.l1 / 2)
        l.eqb16 = (l.l2 + 1 + 15) / 16;
    l.eqb16_rec = 2 * (16 * l.eqb16) * (16 * l.eqb16);
    l.eqbf = (l.eqb16 && eqbf_ok) ? 1 : 0;
    // (16 k + 1 lags: the last block would hold one lag and the column of ones alone; they go on the side)
    l.eqbf_xs = (l.eqbf && l.ns == 2 && l.l2 > 16 && l.l2 % 16 == 1 && l.l2 != 81) ? 1 : -1;   // (complex model; 81 lags: registers)
    l.eqbf_nb = l.eqbf_xs >= 0 ? l.l2 / 16 : l.eqb16;
    {
        // sample tile (interleaved pairs), per-symbol gradient combinations (5 per symbol), constant ones / zeros (64 S each); afterwards the record
        int64_t a = (((int64_t)2 * l.tile_len + 3) & ~(int64_t)3) + 5 * EQ_TB + 128 * l.S;
        const int64_t b = l.eqb16_rec;
        // bf16 instance: (g0, g1) per symbol instead of the five combinations, then the (copy, parity) regions of bf16 pairs and the constant region
        if (l.eqbf) a = (((int64_t)2 * l.tile_len + 3) & ~(int64_t)3) + 2 * EQ_TB + 128 * l.S + ((int64_t)2 * l.S + 1) * ((l.ns == 1 ? 3 : 6) * EQBF_NWD + 8);
        l.lds_eqb16 = (int)((a > b ? a : b) * 4);
    }
    if (l.eqb16) {
        // (the bf16 instance with three lag blocks holds three workgroups per CU)
        const int nbm = l.eqbf ? l.eqbf_nb : l.eqb16;   // blocks whose accumulators the instance holds
        const int res = 256 * (l.eqbf_xs == 1 ? (nbm <= 2 ? 3 : 2) : nbm <= 2 ? 4 : nbm == 3 ? (l.eqbf ? 3 : 4) : 2);
        l.nblk_eq = l.ntile_eq < res ? l.ntile_eq : res;
    }
    else if (l.use_eqcxb && l.eqcxb_nbl >= 3) l.nblk_eq = l.ntile_eq < 256 ? l.ntile_eq : 256;
    if (!generic_eq_ok && !l.use_eqcxb) return V2VAE_ESIZE;
    l.rec_elbo = l.S * rec_elbo_phase(l);
    l.rec_eq = l.ns * l.LT2p * l.LT2p + l.nk1;
    l.rec_sym = l.ns * l.C;
    // ---- workspace carve (floats, every region 64-float aligned)
    int64_t o = 0;
    auto take = [&](int64_t n) { int64_t r = o; o += rup(n, 64); return r; };
    l.o_eqK1 = take(l.nk1);
    l.o_eqK2 = take((int64_t)l.ns * l.LT2p * l.LT2p);
    l.o_h1c = take((int64_t)l.S * l.ns * l.LTp);
    l.o_Hc = take((int64_t)l.S * l.ns * l.LTp * l.LTp);
    l.o_HHc = take((int64_t)l.S * l.ns * l.LTp * l.LTp);
    l.o_dc = take((int64_t)l.S * l.ns * l.LTp);
    l.o_dqc = take((int64_t)l.S * l.ns * l.LTp);
    l.o_chanraw = take(l.chan ? 2 * (int64_t)l.L : 0);
    // real path: the equaliser output is formed in double and kept as a (float, float remainder) pair; the per-symbol kernels
    // add the remainder back when the xhat they are handed is bitwise the one the forward kernel produced
    l.xlo = l.ns == 1 ? 1 : 0;
    l.o_xhi = take(l.xlo ? l.sym_cnt : 0);
    l.o_xlo = take(l.xlo ? l.sym_cnt : 0);
    l.o_mom = take((int64_t)(l.cx_real ? 2 : l.ns) * 4 * l.sym_cnt);
    l.o_gmom = take((int64_t)(l.cx_real ? 2 : l.ns) * 4 * l.sym_cnt);
    l.o_gpk = take(l.cx_split ? 4 * (int64_t)(l.S + 1) * l.sym_cnt : 0);
    l.o_part_elbo = take((int64_t)l.nblk_elbo * l.rec_elbo);
    l.o_part_eq = take((int64_t)l.nblk_eq * l.rec_eq);
    l.o_part_sym = take((int64_t)l.nblk_sym * l.rec_sym);
    l.o_tot_elbo = take(l.rec_elbo);
    l.o_tot_eq = take(l.rec_eq);
    l.o_tot_sym = take(l.rec_sym);
    l.o_cxH = take((int64_t)l.S * 2 * 256);
    l.o_cxh1 = take((int64_t)l.S * 2 * 16);
    l.o_part_cx = take(l.use_cx ? (int64_t)l.nblk_cx * l.S * CX_REC : 0);
    l.o_tot_cx = take((int64_t)l.S * CX_REC);
    l.o_psum_part = take(2 * CX_MAXLAG * (int64_t)l.nblk_psum);
    l.o_psum_tot = take(2 * CX_MAXLAG);
    l.o_g2c = take((l.use_cx || l.use_cxb) ? G2C_FLOATS : 0);
    l.o_gT = take((l.use_cx || l.use_cxb) ? l.sym_cnt : 0);
    l.o_psv = take((l.use_cx || l.use_cxb) ? (int64_t)l.S * l.LE * l.LE : 0);
    l.o_cxbA = take(l.use_cxb ? (int64_t)l.S * l.cxb_nb * l.cxb_nb * CXB_STEPS * 64 : 0);
    l.o_cxbH = take(l.use_cxb ? (int64_t)l.S * 2 * l.cxb_leb * l.cxb_leb : 0);
    l.o_cxbc = take(l.use_cxb ? (int64_t)l.S * 4 * l.cxb_leb : 0);
    l.o_part_cxb = take(l.use_cxb ? (int64_t)l.nblk_cxb * l.S * l.cxb_rec : 0);
    l.o_tot_cxb = take(l.use_cxb ? (int64_t)l.S * l.cxb_rec : 0);
    l.o_part_eqcx = take(l.eqb16 ? (int64_t)l.nblk_eq * l.eqb16_rec : 0);
    l.o_tot_eqcx = take(l.eqb16 ? l.eqb16_rec : 0);
    l.o_eqcxbA = take(l.eqcxb_fwd ? (int64_t)l.eqcx_steps * l.eqcxb_nbl * 128 + l.eqcxb_nbl * 64 : 0);
    // symmetric-form forward on the vector pipe: half the multiply-adds of the matrix product and no padding of the lag count
    l.eqsym_t = l.use_eqcx ? (l.l2 <= 9 ? 9 : l.l2 <= 17 ? 17 : l.l2 <= 25 ? 25 : 31)
                           : l.use_eqcxb ? (l.l2 <= 47 ? 47 : l.l2 <= 65 ? 65 : l.l2 <= 79 ? 79 : 95) : 0;
    // real model on the matrix-core path (see eqb16 above): forward in symmetric form, sums in double, up to 65 lags
    l.eqsymr = (l.cx_real && l.eqb16 && l.l2 <= 65) ? 1 : 0;
    if (l.eqsymr) l.eqsym_t = l.l2 <= 9 ? 9 : l.l2 <= 17 ? 17 : l.l2 <= 25 ? 25 : l.l2 <= 31 ? 31 : l.l2 <= 47 ? 47 : 65;
    l.o_eqsym = take(l.eqsym_t ? l.eqsym_t * (l.eqsym_t + 1) + 2 * l.eqsym_t + 16 : 0);
    // (from three blocks of 16 lags on: up to 31 lags the symmetric form on the vector pipe is ahead, 0.130 against 0.150 ms per
    // 2^22 symbols at 25 lags - there the closing stage, which the two share, is as long as the row sums)
    // (16 NBK + 1 lags: NBK blocks on the matrix pipe and the last lag on the side, in the closing stage)
    l.eqfbf_xl = (l.l2 > 32 && l.l2 % 16 == 1) ? 1 : 0;
    l.eqfbf_nbk = l.eqfbf_xl ? l.l2 / 16 : (l.l2 + 15) / 16;
    l.eqfbf = (l.mode == 1 && l.eqsym_t && !l.eqsymr && l.S <= 2 && l.eqfbf_nbk + l.eqfbf_xl >= 3 &&
               !(l.flags & (V2VAE_FLAG_F32_MATRIX | V2VAE_FLAG_EQ_MFMA_FWD))) ? 1 : 0;
    l.o_eqfbf = take(l.eqfbf ? (int64_t)(l.eqfbf_nbk * (l.eqfbf_nbk + 1) / 2) * 1536 + 32 * l.eqfbf_nbk +
                                   (l.eqfbf_xl ? 32 * l.eqfbf_nbk + 4 : 0) : 0);
    l.o_part_eqcxb = take((l.use_eqcxb && !l.eqb16) ? (int64_t)l.nblk_eq * l.eqcxb_rec : 0);
    l.o_tot_eqcxb = take((l.use_eqcxb && !l.eqb16) ? l.eqcxb_rec : 0);
    {
        int64_t mx = l.rec_elbo;
        if (l.rec_eq > mx) mx = l.rec_eq;
        if (l.rec_sym > mx) mx = l.rec_sym;
        if (l.S * CX_REC > mx) mx = l.S * CX_REC;
        if (l.use_cxb && l.S * l.cxb_rec > mx) mx = l.S * l.cxb_rec;
        if (4096 > mx) mx = 4096;
        if (l.use_eqcxb && l.eqcxb_rec > mx) mx = l.eqcxb_rec;
        if (l.eqb16_rec > mx) mx = l.eqb16_rec;
        l.sumstage_stride = (int64_t)SUM_CHUNKS * mx;
        l.o_sumstage = take(3 * l.sumstage_stride);
    }
    {
        int nbE = l.nblk_elbo > l.nblk_cx ? l.nblk_elbo : l.nblk_cx;
        if (l.nblk_cxb > nbE) nbE = l.nblk_cxb;
        l.o_Epart = take(2 * (int64_t)(nbE + G2_NE));
    }
    l.o_KLpart = take(2 * (int64_t)l.nblk_dm);
    l.total_floats = o;
    *out = l;
    return V2VAE_OK;
}

// ------------------------------------------------------------------------------------------
// device helpers
// ------------------------------------------------------------------------------------------
// Every stage is written as a device function of (block id, block count) so that the same code runs either as
// its own launch (one call per stage, many workgroups) or back to back inside the persistent fit kernel
// (one workgroup per model, bid = 0, nb = 1).
struct GridCtx {
    int bid, nb;
};

__device__ __forceinline__ float stacked_quad(int mode, const float* __restrict__ h2, int L, int o, int a, int b) {
    // effective symmetric second-order kernel of output o on the real-stacked tap vector
    if (mode == 0) return h2[a * L + b] + h2[b * L + a];
    const int ca = a >= L, ia = a - ca * L, cb = b >= L, ib = b - cb * L;
    const float* hr = h2;
    const float* hi = h2 + (size_t)L * L;
    const float Hr = hr[ia * L + ib] + hr[ib * L + ia];
    const float Hi = hi[ia * L + ib] + hi[ib * L + ia];
    if (o == 0) return (ca == cb) ? (ca == 0 ? Hr : -Hr) : -Hi;
    return (ca == cb) ? (ca == 0 ? Hi : -Hi) : Hr;
}

__device__ __forceinline__ float stacked_lin(int mode, const float* __restrict__ h1, int L, int o, int a) {
    if (mode == 0) return h1[a];
    const int ca = a >= L, ia = a - ca * L;
    if (o == 0) return ca == 0 ? h1[ia] : -h1[L + ia];
    return ca == 0 ? h1[L + ia] : h1[ia];
}

// Polynomial channel models (real path): coefficient of x and of x^2 at tap a.  h2 is [L] (memory polynomial,
// lib/real_equalization.py:1333-1334) or [alpha1, alpha2] (Hammerstein, :1146-1147).
__device__ __forceinline__ float chan_lin(int chan, const float* __restrict__ h1, const float* __restrict__ h2, int a) {
    return chan == 2 ? h2[0] * h1[a] : h1[a];
}
__device__ __forceinline__ float chan_sq(int chan, const float* __restrict__ h1, const float* __restrict__ h2, int a) {
    return chan == 2 ? h2[1] * h1[a] : h2[a];
}

// compact tap k of phase r -> stacked original tap index c*L + i, or -1
__device__ __forceinline__ int tap_index(const Lay& l, int r, int k) {
    const int c = k / l.LEp, kk = k - c * l.LEp;
    if (kk >= l.LE) return -1;
    const int i = r + l.L - 1 - kk * l.S;
    if (i < 0 || i >= l.L) return -1;
    return c * l.L + i;
}

// original tap (stream c, index i) -> phase r and compact index k
__device__ __forceinline__ void tap_to_compact(const Lay& l, int a, int* r, int* k) {
    const int c = a / l.L, i = a - c * l.L;
    int rr = (i - (l.L - 1)) % l.S;
    if (rr < 0) rr += l.S;
    *r = rr;
    *k = c * l.LEp + (rr + l.L - 1 - i) / l.S;
}

__device__ __forceinline__ double block_sum_double(double v, double* red /* >= NT/64 doubles of LDS */) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wv] = v;
    __syncthreads();
    double s = 0.0;
    if (threadIdx.x == 0) for (int i = 0; i < NT / 64; ++i) s += red[i];
    return s;  // valid in thread 0
}

// ------------------------------------------------------------------------------------------
// tables
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void d_prep_tables(const GridCtx gc, Lay l, const float* __restrict__ h1, const float* __restrict__ h2,
                                                    const float* __restrict__ k1, const float* __restrict__ k2,
                                                    float* __restrict__ ws) {
    const int tid = gc.bid * NT + threadIdx.x, nth = gc.nb * NT;
    for (int idx = tid; k1 && idx < l.nk1; idx += nth) {
        const int o = idx / (l.ns * l.l1), rem = idx - o * l.ns * l.l1;
        ws[l.o_eqK1 + idx] = stacked_lin(l.mode, k1, l.l1, o, rem);
    }
    const int n2 = l.ns * l.LT2p * l.LT2p;
    for (int idx = tid; k1 && idx < n2; idx += nth) {
        const int o = idx / (l.LT2p * l.LT2p), rem = idx - o * l.LT2p * l.LT2p;
        const int a = rem / l.LT2p, b = rem - a * l.LT2p;
        ws[l.o_eqK2 + idx] = (a < l.LT2 && b < l.LT2) ? stacked_quad(l.mode, k2, l.l2, o, a, b) : 0.f;
    }
    const int nv = l.S * l.ns * l.LTp;
    for (int idx = tid; h1 && idx < nv; idx += nth) {
        const int r = idx / (l.ns * l.LTp), rem = idx - r * l.ns * l.LTp;
        const int o = rem / l.LTp, k = rem - o * l.LTp;
        const int a = tap_index(l, r, k);
        float hv = 0.f, dv = 0.f, dqv = 0.f;
        if (a >= 0 && l.chan) {
            hv = chan_lin(l.chan, h1, h2, a);
            dv = dqv = chan_sq(l.chan, h1, h2, a);  // exact expectation of the square term (:1334, :1147)
        } else if (a >= 0) {
            hv = stacked_lin(l.mode, h1, l.L, o, a);
            dv = stacked_quad(l.mode, h2, l.L, o, a, a);
            dqv = (l.mode == 0) ? h2[a * l.L + a] : dv;  // lib/real_equalization.py:869 uses diag(h2)
        }
        ws[l.o_h1c + idx] = hv;
        ws[l.o_dc + idx] = dv;
        ws[l.o_dqc + idx] = dqv;
    }
    if (l.use_cx && h1) {
        for (int idx = tid; idx < l.S * 2 * 256; idx += nth) {
            const int rp = idx / 256, rem = idx - rp * 256;
            const int r = rp / 2, pl = rp - r * 2;
            const int a = tap_index(l, r, rem / 16), b = tap_index(l, r, rem % 16);  // compact taps of stream 0
            // (real model on this path: the imaginary planes of every table are zero)
            ws[l.o_cxH + idx] = (rem / 16 < l.LE && rem % 16 < l.LE && a >= 0 && b >= 0 && !(l.cx_real && pl)) ? stacked_quad(l.mode, h2, l.L, pl, a, b) : 0.f;
        }
        for (int idx = tid; idx < l.S * 2 * 16; idx += nth) {
            const int rp = idx / 16, x = idx - rp * 16;
            const int r = rp / 2, pl = rp - r * 2;
            const int a = tap_index(l, r, x);
            ws[l.o_cxh1 + idx] = (x < l.LE && a >= 0 && !(l.cx_real && pl)) ? stacked_lin(l.mode, h1, l.L, pl, a) : 0.f;
        }
    }
    if ((l.use_cx || l.use_cxb) && h1) {
        // One wave per coefficient of the correlation form of the third-moment gradient (see k_g2_cx): with d the diagonal
        // of the symmetrised kernel H of phase r,  C[lag] = sum_r sum_x conj(d_x) H_{x, x + lag},  K0 = 2 sum conj(d_x) h_x,
        // the fourth-cumulant gradient of a symbol all of whose windows exist, sum |d_x|^2, and the coefficients of the part of
        // the variance gradient that is a correlation of s = v_a + v_b,  D[lag] = sum_r sum_x |H_{x, x + lag}|^2.
        const int K4 = (l.kmax + 3) & ~3, ntap = 2 * K4 + 4;
        const int lane = threadIdx.x & 63;
        for (int c = tid >> 6; c <= ntap; c += nth >> 6) {
            float cr = 0.f, ci = 0.f, g3 = 0.f, dd = 0.f;
            for (int it = lane; it < l.S * l.LE; it += 64) {
                const int r = it / l.LE, x = it - r * l.LE;
                const int a = tap_index(l, r, x);
                if (a < 0) continue;
                const float im = l.cx_real ? 0.f : 1.f;
                const float dr = stacked_quad(l.mode, h2, l.L, 0, a, a), di = im * stacked_quad(l.mode, h2, l.L, 1, a, a);
                float ar, ai;
                if (c == ntap) {
                    ar = 2.f * stacked_lin(l.mode, h1, l.L, 0, a);
                    ai = im * 2.f * stacked_lin(l.mode, h1, l.L, 1, a);
                    g3 += dr * dr + di * di;
                } else {
                    const int j = x + c - K4;
                    if (j < 0 || j >= l.LE) continue;
                    const int b = tap_index(l, r, j);
                    if (b < 0) continue;
                    ar = stacked_quad(l.mode, h2, l.L, 0, a, b);
                    ai = im * stacked_quad(l.mode, h2, l.L, 1, a, b);
                    dd += ar * ar + ai * ai;
                }
                cr += dr * ar + di * ai;   // conj(d) * a
                ci += dr * ai - di * ar;
            }
            for (int off = 32; off > 0; off >>= 1) {
                cr += __shfl_down(cr, off, 64); ci += __shfl_down(ci, off, 64); g3 += __shfl_down(g3, off, 64);
                dd += __shfl_down(dd, off, 64);
            }
            if (lane == 0) {
                float* t = ws + l.o_g2c;
                if (c == ntap) { t[0] = cr; t[1] = ci; t[2] = g3; t[3] = 0.f; }
                else { t[4 + 2 * c] = cr; t[5 + 2 * c] = ci; t[G2C_D + c] = dd; }
            }
        }
    }
    const int nm = l.S * l.ns * l.LTp * l.LTp;
    for (int idx = tid; h1 && idx < nm; idx += nth) {
        const int ro = idx / (l.LTp * l.LTp), rem = idx - ro * l.LTp * l.LTp;
        const int r = ro / l.ns, o = ro - r * l.ns;
        const int k = rem / l.LTp, j = rem - k * l.LTp;
        const int a = tap_index(l, r, k), b = tap_index(l, r, j);
        float hv = 0.f;
        if (a >= 0 && b >= 0) hv = l.chan ? (a == b ? chan_sq(l.chan, h1, h2, a) : 0.f) : stacked_quad(l.mode, h2, l.L, o, a, b);
        ws[l.o_Hc + idx] = hv;
        ws[l.o_HHc + idx] = hv * hv;
    }
    if (l.chan && h1) {
        const int n2 = l.chan == 2 ? 2 : l.L;
        for (int idx = tid; idx < l.L + n2; idx += nth) ws[l.o_chanraw + idx] = idx < l.L ? h1[idx] : h2[idx - l.L];
    }
}

__global__ __launch_bounds__(NT) void k_prep_tables(Lay l, const float* __restrict__ h1, const float* __restrict__ h2,
                                                    const float* __restrict__ k1, const float* __restrict__ k2,
                                                    float* __restrict__ ws) {
    d_prep_tables(GridCtx{(int)blockIdx.x, (int)gridDim.x}, l, h1, h2, k1, k2, ws);
}

// the same plus, in PREP_EQCXB_BLOCKS further workgroups, the tables of the complex forward equaliser kernels (one launch)
__global__ __launch_bounds__(NT) void k_prep_tables_eq(Lay l, const float* __restrict__ h1, const float* __restrict__ h2,
                                                       const float* __restrict__ k1, const float* __restrict__ k2,
                                                       float* __restrict__ ws, int nb_tab) {
    if ((int)blockIdx.x < nb_tab) d_prep_tables(GridCtx{(int)blockIdx.x, nb_tab}, l, h1, h2, k1, k2, ws);
    else d_prep_eqcxb(l, k1, k2, ws, (int)blockIdx.x - nb_tab, (int)gridDim.x - nb_tab);
}

// ------------------------------------------------------------------------------------------
// equaliser forward
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void load_sample_tile(const Lay& l, const float* __restrict__ y, int64_t g0, float* tile) {
    // tile[c][i] = y_c[g0 + i] (0 outside the batch), i in [0, tile_len)
    // four samples per thread are requested before the first is written (one memory latency per group, not per sample)
    const int n = l.ns * l.tile_len;
    for (int idx0 = threadIdx.x; idx0 < n; idx0 += 4 * NT) {
        float v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = idx0 + k * NT;
            const int c = idx / l.tile_len, i = idx - c * l.tile_len;
            const int64_t g = g0 + i, loc = g - l.y_lo;
            const bool in = idx < n && g >= 0 && g < l.n_glob && loc >= 0 && loc < l.y_len;
            v[k] = in ? y[(int64_t)c * l.y_stride + loc] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = idx0 + k * NT;
            if (idx < n) tile[idx] = v[k];
        }
    }
}

// one tile of EQ_TB symbols per block id.  The sums are formed in double (ACC; exact to double rounding) because the soft
// demapper amplifies an error of xhat by 2 |xhat - a| / noise variance (~1e2) on its way into the equaliser and noise-scaling
// gradients; on the real path the result leaves as a float plus its float remainder (workspace planes o_xhi / o_xlo), with two
// streams as the correctly rounded float.
template <typename ACC>
__device__ __forceinline__ void d_eq_fwd_t(const GridCtx gc, Lay l, const float* __restrict__ y, float* __restrict__ ws,
                                           float* __restrict__ xhat) {
    extern __shared__ float sm[];
    float* tile = sm;
    const int64_t b0 = l.sym_lo + (int64_t)gc.bid * EQ_TB;
    load_sample_tile(l, y, b0 * l.S - l.hl, tile);
    __syncthreads();
    const int64_t b = b0 + threadIdx.x;
    if (b >= l.sym_lo + l.sym_cnt) return;
    const int base = threadIdx.x * l.S + l.hl;
    const float* __restrict__ K1 = ws + l.o_eqK1;
    const float* __restrict__ K2 = ws + l.o_eqK2;
    ACC out[2] = {0, 0};
    for (int c = 0; c < l.ns; ++c)
        for (int j = 0; j < l.l1; ++j) {
            const ACC x = tile[c * l.tile_len + base - l.pad1 + j];
#pragma unroll
            for (int o = 0; o < 2; ++o)
                if (o < l.ns) out[o] = fma((ACC)K1[(o * l.ns + c) * l.l1 + j], x, out[o]);
        }
    if (l.l2 > 0) {
        const int nb = l.LT2p / 4;
        for (int ib = 0; ib < nb; ++ib) {
            ACC acc[2][4] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
            for (int c = 0; c < l.ns; ++c)
                for (int j = 0; j < l.l2; ++j) {
                    const ACC x = tile[c * l.tile_len + base + l.p2 - j];
                    const int ip = c * l.l2 + j;
#pragma unroll
                    for (int o = 0; o < 2; ++o)
                        if (o < l.ns) {
                            const float4 kv = *reinterpret_cast<const float4*>(K2 + ((size_t)o * l.LT2p + ip) * l.LT2p + ib * 4);
                            acc[o][0] = fma((ACC)kv.x, x, acc[o][0]);
                            acc[o][1] = fma((ACC)kv.y, x, acc[o][1]);
                            acc[o][2] = fma((ACC)kv.z, x, acc[o][2]);
                            acc[o][3] = fma((ACC)kv.w, x, acc[o][3]);
                        }
                }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int ii = ib * 4 + i;
                if (ii < l.LT2) {
                    const int c = ii / l.l2, j = ii - c * l.l2;
                    const ACC xi = tile[c * l.tile_len + base + l.p2 - j];
#pragma unroll
                    for (int o = 0; o < 2; ++o)
                        if (o < l.ns) out[o] = fma(xi, acc[o][i], out[o]);
                }
            }
        }
    }
    for (int o = 0; o < l.ns; ++o) {
        const float hi = (float)out[o];
        xhat[(int64_t)o * l.sym_cnt + (b - l.sym_lo)] = hi;
        if (sizeof(ACC) == sizeof(double) && l.xlo) {   // ns == 1
            ws[l.o_xhi + (b - l.sym_lo)] = hi;
            ws[l.o_xlo + (b - l.sym_lo)] = (float)(out[o] - (ACC)hi);
        }
    }
}

__device__ __forceinline__ void d_eq_fwd(const GridCtx gc, Lay l, const float* __restrict__ y, float* __restrict__ ws,
                                         float* __restrict__ xhat) {
    d_eq_fwd_t<double>(gc, l, y, ws, xhat);
}

__global__ __launch_bounds__(NT) void k_eq_fwd(Lay l, const float* __restrict__ y, float* __restrict__ ws,
                                               float* __restrict__ xhat) {
    d_eq_fwd(GridCtx{(int)blockIdx.x, (int)gridDim.x}, l, y, ws, xhat);
}

// The equaliser output at local symbol i in double: the float the caller holds plus, if that float is bitwise the one the
// forward kernel produced, the remainder the kernel kept (real path only; otherwise just the float).
// `hi`, `lo`: planes o_xhi / o_xlo at the symbol, fetched ahead by the caller (anything when the layout has no remainder).
__device__ __forceinline__ double xhat_full(const Lay& l, float x, float hi, float lo) {
    if (!l.xlo) return (double)x;
    return __float_as_uint(hi) == __float_as_uint(x) ? (double)x + (double)lo : (double)x;
}

// ------------------------------------------------------------------------------------------
// soft demapping, moments, KL
// ------------------------------------------------------------------------------------------
// EX = 1 / 2: exactly CB levels per stream and one / two streams, both known at compile time (no bounds checks, levels in
// registers); EX = 0: any level count up to CB, stream count from the layout
template <int CB, int EX = 0>
__device__ __forceinline__ void d_demap_moments(const GridCtx gc, Lay l, const float* __restrict__ xhat, const float* __restrict__ levels,
                                                      const float* __restrict__ prior, const float* __restrict__ scaling,
                                                      const float* __restrict__ noise_var, float* __restrict__ ws,
                                                      float* __restrict__ q_out) {
    __shared__ double red[NT / 64];
    const int C = EX ? CB : l.C, NS = EX ? EX : l.ns;
    const float nv = noise_var[0];
    float* __restrict__ mom = ws + l.o_mom;
    double kl = 0.0;
    double inv_den[2][MAX_C];
    posterior_temperatures<CB>(scaling, C, nv, inv_den[0]);
    posterior_temperatures<CB>(scaling + (NS - 1) * C, C, nv, inv_den[1]);
    float inv_pr[2][MAX_C];  // 1 / prior, once per thread instead of a division per symbol and level
    float lvr[2][MAX_C];     // the levels, in registers
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            const bool in = c < NS && k < C;
            inv_pr[c][k] = in ? 1.f / prior[c * C + k] : 0.f;
            lvr[c][k] = in ? levels[c * C + k] : 0.f;
        }
    const int64_t stride = (int64_t)gc.nb * NT;
    float nx[2] = {0.f, 0.f};   // the next symbol's samples, requested one iteration ahead
    float nhi = 0.f, nlo = 0.f; // and, one-stream model, the copy and the float remainder of its equaliser output
    {
        const int64_t i0 = (int64_t)gc.bid * NT + threadIdx.x;
        if (i0 < l.sym_cnt) {
            nx[0] = xhat[i0]; if (NS > 1) nx[1] = xhat[l.sym_cnt + i0];
            if (l.xlo) { nhi = ws[l.o_xhi + i0]; nlo = ws[l.o_xlo + i0]; }
        }
    }
    for (int64_t i = (int64_t)gc.bid * NT + threadIdx.x; i < l.sym_cnt; i += stride) {
        const int64_t b = l.sym_lo + i;
        const bool in_kl = b >= l.kl_lo && b < l.kl_hi && b >= l.own_lo && b < l.own_lo + l.own_cnt;
        const float cx[2] = {nx[0], nx[1]};
        const float chi = nhi, clo = nlo;
        if (i + stride < l.sym_cnt) {
            nx[0] = xhat[i + stride]; if (NS > 1) nx[1] = xhat[l.sym_cnt + i + stride];
            if (l.xlo) { nhi = ws[l.o_xhi + i + stride]; nlo = ws[l.o_xlo + i + stride]; }
        }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            if (c >= NS) break;
            Posterior p;
            posterior<CB, (EX != 0)>(xhat_full(l, cx[c], chi, clo), lvr[c], inv_den[c], C, p);
            float* m = mom + (int64_t)c * 4 * l.sym_cnt + i;
            m[0] = p.mu; m[l.sym_cnt] = p.var; m[2 * l.sym_cnt] = p.kap; m[3 * l.sym_cnt] = p.gam;
            float klb = 0.f;
#pragma unroll
            for (int k = 0; k < CB; ++k)
                if (k < C) {
                    if (q_out) q_out[((int64_t)c * C + k) * l.sym_cnt + i] = p.q[k];
                    klb += p.q[k] * __logf(fmaf(p.q[k], inv_pr[c][k], 1e-12f));  // argument >= 1e-12: the hardware log is exact enough
                }
            if (in_kl) kl += (double)klb;
        }
        if (l.cx_real) {   // the zero quadrature stream the matrix-core ELBO kernels read
            float* m = mom + (int64_t)4 * l.sym_cnt + i;
            m[0] = 0.f; m[l.sym_cnt] = 0.f; m[2 * l.sym_cnt] = 0.f; m[3 * l.sym_cnt] = 0.f;
        }
    }
    const double s = block_sum_double(kl, red);
    if (threadIdx.x == 0) reinterpret_cast<double*>(ws + l.o_KLpart)[gc.bid] = s;
}

template <int CB, int EX>
__global__ __launch_bounds__(NT, CB <= 4 ? 4 : CB <= 8 ? 2 : 1) void k_demap_moments(Lay l, const float* __restrict__ xhat, const float* __restrict__ levels,
                                                      const float* __restrict__ prior, const float* __restrict__ scaling,
                                                      const float* __restrict__ noise_var, float* __restrict__ ws,
                                                      float* __restrict__ q_out) {
    d_demap_moments<CB, EX>(GridCtx{(int)blockIdx.x, (int)gridDim.x}, l, xhat, levels, prior, scaling, noise_var, ws, q_out);
}

// hard decisions: arg-max over the levels of -(x - level)^2 / (noise variance x scaling), stream by stream; four symbols
// per thread as one 16-byte load and one 16-byte store where the planes allow it
template <int CB>
__global__ __launch_bounds__(NT) void k_decide(Lay l, const float* __restrict__ xhat, const float* __restrict__ levels,
                                               const float* __restrict__ scaling, const float* __restrict__ noise_var,
                                               int32_t* __restrict__ dec) {
    const float nv = noise_var[0];
    const int64_t n = l.sym_cnt;
    const bool vec = (n % 4 == 0) && (((reinterpret_cast<uintptr_t>(xhat) | reinterpret_cast<uintptr_t>(dec)) & 15) == 0);
    const int64_t first = (int64_t)blockIdx.x * NT + threadIdx.x, stride = (int64_t)gridDim.x * NT;
    for (int c = 0; c < l.ns; ++c) {
        float lv[CB], den[CB];
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            const bool in = k < l.C;
            lv[k] = in ? levels[c * l.C + k] : 0.f;
            den[k] = in ? nv * scaling[c * l.C + k] : 1.f;
        }
        auto pick = [&](float x) {
            float best = -INFINITY;
            int arg = 0;
#pragma unroll
            for (int k = 0; k < CB; ++k) {
                const float d = x - lv[k];
                const float z = -(d * d) / den[k];
                if (k < l.C && z > best) { best = z; arg = k; }
            }
            return arg;
        };
        const float* __restrict__ xs = xhat + c * n;
        int32_t* __restrict__ ds = dec + c * n;
        if (vec) {
            const float4* __restrict__ xv = reinterpret_cast<const float4*>(xs);
            int4* __restrict__ dv = reinterpret_cast<int4*>(ds);
            for (int64_t i = first; i < n / 4; i += stride) {
                const float4 x = xv[i];
                dv[i] = make_int4(pick(x.x), pick(x.y), pick(x.z), pick(x.w));
            }
        } else {
            for (int64_t i = first; i < n; i += stride) ds[i] = pick(xs[i]);
        }
    }
}

// ------------------------------------------------------------------------------------------
// per-symbol backward
// ------------------------------------------------------------------------------------------
// EX = 1 / 2: exactly CB levels per stream and one / two streams, both known at compile time (no bounds checks, levels in
// registers); EX = 0: any level count up to CB, stream count from the layout
template <int CB, int EX = 0>
__device__ __forceinline__ void d_sym_bwd(const GridCtx gc, Lay l, const float* __restrict__ xhat, const float* __restrict__ levels,
                                                const float* __restrict__ prior, const float* __restrict__ scaling,
                                                const float* __restrict__ noise_var, const double* __restrict__ scal,
                                                const float* __restrict__ grad_scale, float* __restrict__ ws,
                                                float* __restrict__ gxhat) {
    __shared__ float red[NT / 64][2 * MAX_C];
    const int C = EX ? CB : l.C, NS = EX ? EX : l.ns;
    const float nv = noise_var[0];
    const float lam = (float)scal[V2VAE_SCAL_LAMBDA];
    const float gsc = grad_scale ? grad_scale[0] : 1.f;
    const float* __restrict__ gmom = ws + l.o_gmom;
    float gs[2][MAX_C];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int k = 0; k < CB; ++k) gs[c][k] = 0.f;
    double inv_den[2][MAX_C];
    posterior_temperatures<CB>(scaling, C, nv, inv_den[0]);
    posterior_temperatures<CB>(scaling + (NS - 1) * C, C, nv, inv_den[1]);
    // per-level constants, once per thread: 1 / prior, 1e-12 * prior, 1 / (nv sc) and 1 / (nv sc^2) in float
    float inv_pr[2][MAX_C], eps_pr[2][MAX_C], idn_f[2][MAX_C], idn_sc[2][MAX_C], lvr[2][MAX_C];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            const bool in = c < NS && k < C;
            const float pr = in ? prior[c * C + k] : 1.f;
            lvr[c][k] = in ? levels[c * C + k] : 0.f;
            inv_pr[c][k] = 1.f / pr;
            eps_pr[c][k] = 1e-12f * pr;
            idn_f[c][k] = (float)inv_den[c][k];
            idn_sc[c][k] = in ? idn_f[c][k] / scaling[c * C + k] : 0.f;
        }
    // the loads of the next symbol are issued before the arithmetic of this one (the arithmetic is long enough to cover them)
    float nx[2], ng[2][4];
    float nhi = 0.f, nlo = 0.f;   // one-stream model: the copy and the float remainder of the equaliser output
    const bool cx_path = l.use_cx || l.use_cxb;
    const float g4c = cx_path ? ws[l.o_g2c + 2] : 0.f;
    auto fetch = [&](int64_t i) {
        const int64_t b = l.sym_lo + i;
        const bool own = i < l.sym_cnt && b >= l.own_lo && b < l.own_lo + l.own_cnt;
        if (own && l.xlo) { nhi = ws[l.o_xhi + i]; nlo = ws[l.o_xlo + i]; }
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            nx[c] = 0.f; ng[c][0] = 0.f; ng[c][1] = 0.f; ng[c][2] = 0.f; ng[c][3] = 0.f;
            if (own && c < NS) {
                nx[c] = xhat[(int64_t)c * l.sym_cnt + i];
                const float* g = gmom + (int64_t)c * 4 * l.sym_cnt + i;
                if (l.cx_split) continue;
                ng[c][0] = g[0]; ng[c][1] = g[l.sym_cnt]; ng[c][2] = g[2 * l.sym_cnt];
                // blocked matrix-core ELBO path: the 4th-cumulant gradient of a symbol all of whose windows exist is one constant
                // (kept in the coefficient table); only the symbols at the ends of the sequence have it in the plane
                ng[c][3] = (cx_path && b >= l.kmax && b * l.S + l.S - 1 < l.W) ? g4c : g[3 * l.sym_cnt];
            }
        }
        if (own && l.cx_split) {   // run kernel: S + 1 per-symbol 4-vectors (one per phase, then k_g2_cx's)
            typedef float v4 __attribute__((ext_vector_type(4)));
            const v4* pk = reinterpret_cast<const v4*>(ws + l.o_gpk) + i;
            v4 p0 = pk[0];
            const v4 p2 = pk[(int64_t)l.S * l.sym_cnt];
            if (l.S >= 2) p0 += pk[l.sym_cnt];
            if (l.S == 4) { p0 += pk[2 * l.sym_cnt]; p0 += pk[3 * l.sym_cnt]; }
            ng[0][0] = p0.x; ng[0][1] = p0.z + p2.z; ng[0][2] = p2.x; ng[0][3] = p2.w;
            ng[1][0] = p0.y; ng[1][1] = p0.w + p2.z; ng[1][2] = p2.y; ng[1][3] = p2.w;
        }
    };
    const int64_t stride = (int64_t)gc.nb * NT;
    fetch((int64_t)gc.bid * NT + threadIdx.x);
    for (int64_t i = (int64_t)gc.bid * NT + threadIdx.x; i < l.sym_cnt; i += stride) {
        const int64_t b = l.sym_lo + i;
        const bool own = b >= l.own_lo && b < l.own_lo + l.own_cnt;
        const bool in_kl = own && b >= l.kl_lo && b < l.kl_hi;
        float cx[2], cg[2][4];
#pragma unroll
        for (int c = 0; c < 2; ++c) { cx[c] = nx[c]; cg[c][0] = ng[c][0]; cg[c][1] = ng[c][1]; cg[c][2] = ng[c][2]; cg[c][3] = ng[c][3]; }
        const float chi = nhi, clo = nlo;
        fetch(i + stride);
#pragma unroll
        for (int c = 0; c < 2; ++c)
            if (c < NS) {
                float gx = 0.f;
                if (own) {
                    const double x = xhat_full(l, cx[c], chi, clo);
                    const float* lv = lvr[c];
                    Posterior p;
                    posterior<CB, (EX != 0)>(x, lv, inv_den[c], C, p);
                    const float gmu = cg[c][0], gv = cg[c][1], gk = cg[c][2], gg = cg[c][3];
                    // Softmax backward in the pairwise form dz_k = q_k sum_j q_j (v_k - v_j): each term is a product of a
                    // (possibly tiny) probability with an O(1) difference, so nothing cancels against the posterior mean
                    // of v (which for peaked posteriors would be many orders above the result) and fp32 suffices.
                    float v[MAX_C];
#pragma unroll
                    for (int k = 0; k < CB; ++k)
                        if (k < C) {
                            const float d = lv[k] - p.mu, d2 = d * d;
                            float vk = gmu * d + gv * d2 + gk * (d2 * d - 3.f * p.var * d) +
                                       gg * (d2 * d2 - 4.f * p.kap * d - 6.f * p.var * d2);
                            vk *= lam;
                            if (in_kl)
                                vk += __logf(fmaf(p.q[k], inv_pr[c][k], 1e-12f)) +
                                      p.q[k] * __builtin_amdgcn_rcpf(p.q[k] + eps_pr[c][k]);
                            v[k] = vk * gsc;
                        }
                    float gxa = 0.f;
#pragma unroll
                    for (int k = 0; k < CB; ++k)
                        if (k < C) {
                            float sk = 0.f;
#pragma unroll
                            for (int j = 0; j < CB; ++j)
                                if (j < C && j != k) sk = fmaf(p.q[j], v[k] - v[j], sk);
                            const float dz = p.q[k] * sk;
                            const float df = (float)(x - (double)lv[k]);
                            gxa = fmaf(dz, -2.f * df * idn_f[c][k], gxa);
                            gs[c][k] = fmaf(dz * df, df * idn_sc[c][k], gs[c][k]);
                        }
                    gx = gxa;
                }
                gxhat[(int64_t)c * l.sym_cnt + i] = gx;
            }
    }
    // deterministic block reduction of the noise-scaling gradient
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int k = 0; k < CB; ++k) {
            float v = gs[c][k];
            for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
            if (lane == 0) red[wv][c * MAX_C + k] = v;
        }
    __syncthreads();
    if (threadIdx.x < NS * C) {
        const int c = threadIdx.x / C, k = threadIdx.x - c * C;
        float s = 0.f;
        for (int w = 0; w < NT / 64; ++w) s += red[w][c * MAX_C + k];
        ws[l.o_part_sym + (int64_t)gc.bid * l.rec_sym + threadIdx.x] = s;
    }
}

template <int CB, int EX>
__global__ __launch_bounds__(NT, CB <= 2 ? 4 : CB <= 4 ? 3 : CB <= 8 ? 2 : 1) void k_sym_bwd(Lay l, const float* __restrict__ xhat, const float* __restrict__ levels,
                                                const float* __restrict__ prior, const float* __restrict__ scaling,
                                                const float* __restrict__ noise_var, const double* __restrict__ scal,
                                                const float* __restrict__ grad_scale, float* __restrict__ ws,
                                                float* __restrict__ gxhat) {
    d_sym_bwd<CB, EX>(GridCtx{(int)blockIdx.x, (int)gridDim.x}, l, xhat, levels, prior, scaling, noise_var, scal, grad_scale, ws, gxhat);
}

// ------------------------------------------------------------------------------------------
// equaliser backward (persistent workgroups; partial sums per workgroup)
// ------------------------------------------------------------------------------------------
template <int MAXI>
__device__ __forceinline__ void d_eq_bwd(const GridCtx gc, Lay l, const float* __restrict__ y, const float* __restrict__ gx,
                                               float* __restrict__ ws) {
    extern __shared__ float sm[];
    float* tile = sm;
    float* gt = tile + l.ns * l.tile_len;
    float* scratch = gt + l.ns * EQ_TB;
    const int tid = threadIdx.x;
    const int nb = l.LT2p / 4;
    // work assignment
    int item0, chunk, nchunk = l.nchunk_eq;
    if (MAXI == 1 && l.nitems_eq > 0 && l.nitems_eq <= NT) { item0 = tid % l.nitems_eq; chunk = tid / l.nitems_eq; }
    else { item0 = tid; chunk = 0; }
    const bool active = chunk < nchunk;
    float acc[MAXI][16];
#pragma unroll
    for (int s = 0; s < MAXI; ++s)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][e] = 0.f;
    float acc1[4] = {0.f, 0.f, 0.f, 0.f};
    for (int t = gc.bid; t < l.ntile_eq; t += gc.nb) {
        const int64_t b0 = l.own_lo + (int64_t)t * EQ_TB;
        const int64_t rem = l.own_lo + l.own_cnt - b0;
        const int nbs = rem < EQ_TB ? (int)rem : EQ_TB;
        __syncthreads();
        load_sample_tile(l, y, b0 * l.S - l.hl, tile);
        for (int idx = tid; idx < l.ns * EQ_TB; idx += NT) {
            const int o = idx / EQ_TB, i = idx - o * EQ_TB;
            gt[idx] = i < nbs ? gx[(int64_t)o * l.sym_cnt + (b0 + i - l.sym_lo)] : 0.f;
        }
        __syncthreads();
        if (active && l.nitems_eq > 0) {
#pragma unroll
            for (int s = 0; s < MAXI; ++s) {
                const int it = item0 + s * NT;
                if (it < l.nitems_eq) {
                    const int o = it / (nb * nb), r2 = it - o * nb * nb;
                    const int ib = r2 / nb, jb = r2 - ib * nb;
                    int offr[4], offc[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        int ii = ib * 4 + e;
                        int c = ii < l.LT2 ? ii / l.l2 : l.ns - 1;
                        int j = ii < l.LT2 ? ii - c * l.l2 : l.l2 + (ii - l.LT2);
                        offr[e] = c * l.tile_len + l.hl + l.p2 - j;
                        ii = jb * 4 + e;
                        c = ii < l.LT2 ? ii / l.l2 : l.ns - 1;
                        j = ii < l.LT2 ? ii - c * l.l2 : l.l2 + (ii - l.LT2);
                        offc[e] = c * l.tile_len + l.hl + l.p2 - j;
                    }
                    for (int bl = chunk; bl < nbs; bl += nchunk) {
                        const float g = gt[o * EQ_TB + bl];
                        const int sb = bl * l.S;
                        float xr[4], xc[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) { xr[e] = g * tile[offr[e] + sb]; xc[e] = tile[offc[e] + sb]; }
#pragma unroll
                        for (int a = 0; a < 4; ++a)
#pragma unroll
                            for (int b = 0; b < 4; ++b) acc[s][a * 4 + b] = fmaf(xr[a], xc[b], acc[s][a * 4 + b]);
                    }
                }
            }
        }
        // first-order kernel gradient
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int it = tid + s * NT;
            if (it < l.nk1) {
                const int o = it / (l.ns * l.l1), r2 = it - o * l.ns * l.l1;
                const int c = r2 / l.l1, j = r2 - c * l.l1;
                const int off = c * l.tile_len + l.hl - l.pad1 + j;
                float a1 = acc1[s];
                for (int bl = 0; bl < nbs; ++bl) a1 = fmaf(gt[o * EQ_TB + bl], tile[off + bl * l.S], a1);
                acc1[s] = a1;
            }
        }
    }
    // flush
    float* part = ws + l.o_part_eq + (int64_t)gc.bid * l.rec_eq;
    __syncthreads();
    if (l.nitems_eq > 0) {
        if (MAXI == 1 && l.nitems_eq <= NT) {
            for (int base = 0; base < nchunk; base += 1) {
                // one chunk at a time through scratch keeps the LDS need at nitems*16 floats
                __syncthreads();
                if (active && chunk == base)
#pragma unroll
                    for (int e = 0; e < 16; ++e) scratch[item0 * 16 + e] = (base == 0 ? 0.f : scratch[item0 * 16 + e]) + acc[0][e];
            }
            __syncthreads();
            for (int idx = tid; idx < l.nitems_eq * 16; idx += NT) {
                const int it = idx / 16, e = idx - it * 16;
                const int o = it / (nb * nb), r2 = it - o * nb * nb;
                const int ib = r2 / nb, jb = r2 - ib * nb;
                part[((size_t)o * l.LT2p + ib * 4 + e / 4) * l.LT2p + jb * 4 + (e & 3)] = scratch[idx];
            }
        } else {
#pragma unroll
            for (int s = 0; s < MAXI; ++s) {
                const int it = item0 + s * NT;
                if (it < l.nitems_eq) {
                    const int o = it / (nb * nb), r2 = it - o * nb * nb;
                    const int ib = r2 / nb, jb = r2 - ib * nb;
#pragma unroll
                    for (int e = 0; e < 16; ++e)
                        part[((size_t)o * l.LT2p + ib * 4 + e / 4) * l.LT2p + jb * 4 + (e & 3)] = acc[s][e];
                }
            }
        }
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int it = tid + s * NT;
        if (it < l.nk1) part[(size_t)l.ns * l.LT2p * l.LT2p + it] = acc1[s];
    }
}

template <int MAXI>
__global__ __launch_bounds__(NT) void k_eq_bwd(Lay l, const float* __restrict__ y, const float* __restrict__ gx,
                                               float* __restrict__ ws) {
    d_eq_bwd<MAXI>(GridCtx{(int)blockIdx.x, (int)gridDim.x}, l, y, gx, ws);
}

// ------------------------------------------------------------------------------------------
// fused ELBO forward + backward over sliding windows (polyphase form)
// ------------------------------------------------------------------------------------------
// DIAG: the second-order kernel is diagonal (memory-polynomial and Hammerstein channel models,
// lib/real_equalization.py:1304-1354, :1117-1167): H mu, (H.H) v and H w are element-wise products and of dE/dH only the
// diagonal is formed, so the work per window is O(taps) instead of O(taps^2).
template <int MAXI, bool DIAG>
__device__ __forceinline__ void d_elbo(const GridCtx gc, Lay l, const float* __restrict__ y, float* __restrict__ ws) {
    extern __shared__ float sm[];
    const int TW = l.TWIN, TSYM = l.TSYM, LTp = l.LTp, NS = l.ns, NG = l.NG;
    float* s_mom = sm;                       // [NS][4][TSYM]
    float* s_yw = s_mom + NS * 4 * TSYM;     // [NS][TW]  y' (0 if window invalid)
    float* s_ev = s_yw + NS * TW;            // [NS][TW]  residual e = A - y' (0 if invalid)
    float* s_Av = s_ev + NS * TW;            // [NS][TW]  A (0 if invalid)
    float* s_ow = s_Av + NS * TW;            // [TW] 1 if the window is owned and valid
    float* s_vl = s_ow + TW;                 // [TW] 1 if the window is valid
    float* s_z = s_vl + TW;                  // [NS][LTp][TW]
    float* s_w = s_z + NS * LTp * TW;        // [NS][LTp][TW]
    float* s_Dmu = s_w + NS * LTp * TW;      // [LTp][TW]
    float* s_Dv = s_Dmu + LTp * TW;
    float* s_Dk = s_Dv + LTp * TW;
    float* s_red = s_Dk + LTp * TW;          // [NS][5][NG][TW]
    float* scratch = s_red + NS * 5 * NT;    // SCRATCH_F
    __shared__ double dred[NT / 64];

    const int tid = threadIdx.x;
    const int ts = tid % TW;
    const int g = __builtin_amdgcn_readfirstlane(tid / TW);  // TW >= 64: uniform per wave
    const int nb = LTp / 4;
    const float* __restrict__ mom = ws + l.o_mom;
    float* __restrict__ gmom = ws + l.o_gmom;
    const int64_t own_hi = l.own_lo + l.own_cnt;
    const int64_t sym_hi = l.sym_lo + l.sym_cnt;

    int item0, chunk;
    const int nchunk = l.nchunk_elbo;
    if (MAXI == 1 && l.nitems_elbo <= NT) { item0 = tid % l.nitems_elbo; chunk = tid / l.nitems_elbo; }
    else { item0 = tid; chunk = 0; }
    const bool active = chunk < nchunk;

    double Eacc = 0.0;
    for (int r = 0; r < l.S; ++r) {
        const float* __restrict__ Hc = ws + l.o_Hc + (size_t)r * NS * LTp * LTp;
        const float* __restrict__ HHc = ws + l.o_HHc + (size_t)r * NS * LTp * LTp;
        const float* __restrict__ h1c = ws + l.o_h1c + (size_t)r * NS * LTp;
        const float* __restrict__ dc = ws + l.o_dc + (size_t)r * NS * LTp;
        const float* __restrict__ dqc = ws + l.o_dqc + (size_t)r * NS * LTp;
        float acc[MAXI][16];
#pragma unroll
        for (int s = 0; s < MAXI; ++s)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[s][e] = 0.f;
        double v1 = 0.0, v2 = 0.0, v3 = 0.0;  // gh1, gd, gdq of vector item (o,k) = tid
        double v4 = 0.0, v5 = 0.0;            // DIAG: diagonal entries (k, k) of the two matrix gradients

        for (int tile = gc.bid; tile < l.ntile_elbo; tile += gc.nb) {
            const int64_t s0 = l.own_lo + (int64_t)tile * l.TS;
            const int64_t s1 = (s0 + l.TS < own_hi) ? s0 + l.TS : own_hi;
            const int64_t tbase = s0 - l.kmax;
            __syncthreads();
            // ---- A: stage moments and window data
            for (int idx0 = tid; idx0 < NS * 4 * TSYM; idx0 += 4 * NT) {   // four loads in flight per thread
                float v[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int idx = idx0 + k * NT;
                    const int cq = idx / TSYM, ms = idx - cq * TSYM;
                    const int64_t m = tbase + ms;
                    v[k] = (idx < NS * 4 * TSYM && m >= l.sym_lo && m < sym_hi) ? mom[(int64_t)cq * l.sym_cnt + (m - l.sym_lo)] : 0.f;
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int idx = idx0 + k * NT;
                    if (idx < NS * 4 * TSYM) s_mom[idx] = v[k];
                }
            }
            if (tid < TW) {
                const int64_t t = tbase + tid;
                const int64_t n = t * l.S + r;
                const bool valid = (t < s1) && n >= 0 && n < l.W;
                s_vl[tid] = valid ? 1.f : 0.f;
                s_ow[tid] = (valid && t >= s0) ? 1.f : 0.f;
                for (int o = 0; o < NS; ++o) {
                    float v = 0.f;
                    if (valid) {
                        const int64_t loc = l.pre + n - l.y_lo;
                        if (loc >= 0 && loc < l.y_len) v = y[(int64_t)o * l.y_stride + loc];
                    }
                    s_yw[o * TW + tid] = v;
                }
            }
            __syncthreads();
            // ---- B: pass 1 -- u = H mu, tt = (H.H) v for this thread's rows; per-window partial scalars
            for (int o = 0; o < NS; ++o) {
                float pa = 0.f, pq = 0.f, ps = 0.f, psq = 0.f, pvs = 0.f;
                const float* __restrict__ Ho = Hc + (size_t)o * LTp * LTp;
                const float* __restrict__ HHo = HHc + (size_t)o * LTp * LTp;
                for (int u = g; u < nb; u += NG) {
                    const int k0 = u * 4;
                    float uu[4] = {0.f, 0.f, 0.f, 0.f}, tt[4] = {0.f, 0.f, 0.f, 0.f};
                    if (DIAG) {   // NS == 1
#pragma unroll
                        for (int i = 0; i < 4; ++i) {
                            const int k = k0 + i;
                            const float hd = Ho[(size_t)k * LTp + k];
                            uu[i] = hd * s_mom[ts + k];
                            tt[i] = hd * hd * s_mom[TSYM + ts + k];
                        }
                    } else
                    for (int c = 0; c < NS; ++c) {
                        const float* mrow = s_mom + (c * 4 + 0) * TSYM + ts;
                        const float* vrow = s_mom + (c * 4 + 1) * TSYM + ts;
                        for (int kk = 0; kk < l.LE; ++kk) {
                            const int j = c * l.LEp + kk;
                            const float4 hv = *reinterpret_cast<const float4*>(Ho + (size_t)j * LTp + k0);
                            const float4 hh = *reinterpret_cast<const float4*>(HHo + (size_t)j * LTp + k0);
                            const float mj = mrow[kk], vj = vrow[kk];
                            uu[0] = fmaf(hv.x, mj, uu[0]); uu[1] = fmaf(hv.y, mj, uu[1]);
                            uu[2] = fmaf(hv.z, mj, uu[2]); uu[3] = fmaf(hv.w, mj, uu[3]);
                            tt[0] = fmaf(hh.x, vj, tt[0]); tt[1] = fmaf(hh.y, vj, tt[1]);
                            tt[2] = fmaf(hh.z, vj, tt[2]); tt[3] = fmaf(hh.w, vj, tt[3]);
                        }
                    }
                    const int c = k0 / l.LEp, kk0 = k0 - c * l.LEp;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int k = k0 + i;
                        const float* mb = s_mom + c * 4 * TSYM + ts + kk0 + i;
                        const float mk = mb[0], vk = mb[TSYM], kp = mb[2 * TSYM], gm = mb[3 * TSYM];
                        const float h1k = h1c[o * LTp + k], dk = dc[o * LTp + k], dqk = dqc[o * LTp + k];
                        const float z = fmaf(2.f, uu[i], h1k);
                        pa = fmaf(h1k, mk, pa);
                        pq = fmaf(mk, uu[i], pq);
                        ps = fmaf(vk, dk, ps);
                        psq = fmaf(vk, dqk, psq);
                        pvs += vk * z * z + 2.f * vk * tt[i] + 2.f * kp * dk * z + gm * dk * dk;
                        s_z[(o * LTp + k) * TW + ts] = z;
                        s_w[(o * LTp + k) * TW + ts] = 2.f * vk * z + 2.f * kp * dk;
                        const float dv = z * z + 4.f * tt[i], dkp = 2.f * dk * z;
                        if (o == 0) { s_Dv[k * TW + ts] = dv; s_Dk[k * TW + ts] = dkp; }
                        else { s_Dv[k * TW + ts] += dv; s_Dk[k * TW + ts] += dkp; }
                    }
                }
                float* rd = s_red + (o * 5 * NG + g) * TW + ts;
                rd[0] = pa; rd[NG * TW] = pq; rd[2 * NG * TW] = ps; rd[3 * NG * TW] = psq; rd[4 * NG * TW] = pvs;
            }
            __syncthreads();
            // ---- C: per-window scalars
            if (tid < TW) {
                const float vl = s_vl[tid];
                float En = 0.f;
                for (int o = 0; o < NS; ++o) {
                    float q5[5];
#pragma unroll
                    for (int q = 0; q < 5; ++q) {
                        float s = 0.f;
                        for (int gg = 0; gg < NG; ++gg) s += s_red[((o * 5 + q) * NG + gg) * TW + tid];
                        q5[q] = s;
                    }
                    const float yv = s_yw[o * TW + tid];
                    const float A = q5[0] + q5[1] + q5[2];
                    const float e = A - yv;
                    En += e * e + 2.f * yv * (q5[2] - q5[3]) + q5[4];
                    s_ev[o * TW + tid] = vl * e;
                    s_Av[o * TW + tid] = vl * A;
                }
                if (s_ow[tid] != 0.f) Eacc += (double)En;
            }
            __syncthreads();
            // ---- D: pass 2 -- H w, per-(window, tap) gradients of the moments
            for (int u = g; u < nb; u += NG) {
                const int k0 = u * 4;
                float dmu[4] = {0.f, 0.f, 0.f, 0.f}, dvx[4] = {0.f, 0.f, 0.f, 0.f};
                for (int o = 0; o < NS; ++o) {
                    const float* __restrict__ Ho = Hc + (size_t)o * LTp * LTp;
                    float hw[4] = {0.f, 0.f, 0.f, 0.f};
                    if (DIAG) {
#pragma unroll
                        for (int i = 0; i < 4; ++i) hw[i] = Ho[(size_t)(k0 + i) * LTp + k0 + i] * s_w[(o * LTp + k0 + i) * TW + ts];
                    } else
                    for (int c = 0; c < NS; ++c)
                        for (int kk = 0; kk < l.LE; ++kk) {
                            const int j = c * l.LEp + kk;
                            const float4 hv = *reinterpret_cast<const float4*>(Ho + (size_t)j * LTp + k0);
                            const float wj = s_w[(o * LTp + j) * TW + ts];
                            hw[0] = fmaf(hv.x, wj, hw[0]); hw[1] = fmaf(hv.y, wj, hw[1]);
                            hw[2] = fmaf(hv.z, wj, hw[2]); hw[3] = fmaf(hv.w, wj, hw[3]);
                        }
                    const float e = s_ev[o * TW + ts], A = s_Av[o * TW + ts], yv = s_yw[o * TW + ts];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int k = k0 + i;
                        dmu[i] += 2.f * e * s_z[(o * LTp + k) * TW + ts] + 2.f * hw[i];
                        dvx[i] += 2.f * A * dc[o * LTp + k] - 2.f * yv * dqc[o * LTp + k];
                    }
                }
                const float vl = s_vl[ts];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int k = k0 + i;
                    s_Dmu[k * TW + ts] = vl * dmu[i];
                    s_Dv[k * TW + ts] = vl * (s_Dv[k * TW + ts] + dvx[i]);
                    s_Dk[k * TW + ts] = vl * s_Dk[k * TW + ts];
                }
            }
            __syncthreads();
            // ---- E1: gather per-symbol gradients (symbol m = t + kk)
            for (int idx = tid; idx < NS * l.TS; idx += NT) {
                const int c = idx / l.TS, ml = idx - c * l.TS;
                const int64_t m = s0 + ml;
                if (m < s1) {
                    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
                    for (int kk = 0; kk < l.LE; ++kk) {
                        const int k = c * l.LEp + kk;
                        const int tsl = ml + l.kmax - kk;  // window slot of t = m - kk
                        a0 += s_Dmu[k * TW + tsl];
                        a1 += s_Dv[k * TW + tsl];
                        a2 += s_Dk[k * TW + tsl];
                        float dd = 0.f;
                        for (int o = 0; o < NS; ++o) { const float d = dc[o * LTp + k]; dd = fmaf(d, d, dd); }
                        a3 = fmaf(s_vl[tsl], dd, a3);
                    }
                    float* gp = gmom + (int64_t)c * 4 * l.sym_cnt + (m - l.sym_lo);
                    if (r == 0) { gp[0] = a0; gp[l.sym_cnt] = a1; gp[2 * l.sym_cnt] = a2; gp[3 * l.sym_cnt] = a3; }
                    else { gp[0] += a0; gp[l.sym_cnt] += a1; gp[2 * l.sym_cnt] += a2; gp[3 * l.sym_cnt] += a3; }
                }
            }
            // ---- E2: parameter-gradient outer products over this tile's owned windows (DIAG: only their diagonals, in E3)
            if (!DIAG && active) {
#pragma unroll
                for (int s = 0; s < MAXI; ++s) {
                    const int it = item0 + s * NT;
                    if (it < l.nitems_elbo) {
                        const bool isG = it < l.nG_elbo;
                        const int itx = isG ? it : it - l.nG_elbo;
                        const int o = isG ? itx / (nb * nb) : 0;
                        const int r2 = itx - o * nb * nb;
                        const int kb = r2 / nb, jb = r2 - kb * nb;
                        const int ck = (kb * 4) / l.LEp, kk0 = kb * 4 - ck * l.LEp;
                        const int cj = (jb * 4) / l.LEp, jj0 = jb * 4 - cj * l.LEp;
                        const int q = isG ? 0 : 1;
                        const float* rowm = s_mom + (ck * 4 + q) * TSYM + kk0;
                        const float* colm = s_mom + (cj * 4 + q) * TSYM + jj0;
                        for (int t = chunk; t < TW; t += nchunk) {
                            const float wgt = s_ow[t];
                            float xr[4], xc[4];
                            if (isG) {
                                const float e = s_ev[o * TW + t];
#pragma unroll
                                for (int i = 0; i < 4; ++i)
                                    xr[i] = wgt * fmaf(e, rowm[t + i], s_w[(o * LTp + kb * 4 + i) * TW + t]);
                            } else {
#pragma unroll
                                for (int i = 0; i < 4; ++i) xr[i] = wgt * rowm[t + i];
                            }
#pragma unroll
                            for (int i = 0; i < 4; ++i) xc[i] = colm[t + i];
#pragma unroll
                            for (int a = 0; a < 4; ++a)
#pragma unroll
                                for (int b = 0; b < 4; ++b) acc[s][a * 4 + b] = fmaf(xr[a], xc[b], acc[s][a * 4 + b]);
                        }
                    }
                }
            }
            // ---- E3: vector gradients (h1, diag, diag-of-mean), one (o,k) per thread
            if (tid < NS * LTp) {
                const int o = tid / LTp, k = tid - o * LTp;
                const int c = k / l.LEp, kk = k - c * l.LEp;
                const float dk = dc[o * LTp + k];
                const float* mb = s_mom + c * 4 * TSYM + kk;
                float f1 = 0.f, f2 = 0.f, f3 = 0.f, f4 = 0.f, f5 = 0.f;
                for (int t = 0; t < TW; ++t) {
                    const float wgt = s_ow[t];
                    const float mk = mb[t], vk = mb[TSYM + t], kp = mb[2 * TSYM + t], gm = mb[3 * TSYM + t];
                    const float e = s_ev[o * TW + t], A = s_Av[o * TW + t], yv = s_yw[o * TW + t];
                    const float z = s_z[(o * LTp + k) * TW + t], w = s_w[(o * LTp + k) * TW + t];
                    f1 += wgt * (2.f * e * mk + w);
                    f2 += wgt * (2.f * A * vk + 2.f * kp * z + 2.f * gm * dk);
                    f3 += wgt * (-2.f * yv * vk);
                    if (DIAG) {   // diagonal of the two outer-product accumulations of E2
                        f4 += wgt * fmaf(e, mk, w) * mk;
                        f5 += wgt * vk * vk;
                    }
                }
                v1 += (double)f1; v2 += (double)f2; v3 += (double)f3;
                if (DIAG) { v4 += (double)f4; v5 += (double)f5; }
            }
        }
        // ---- flush this phase's partial sums
        float* part = ws + l.o_part_elbo + (int64_t)gc.bid * l.rec_elbo + (size_t)r * rec_elbo_phase(l);
        float* partG = part;
        float* partP = part + NS * LTp * LTp;
        float* partV = partP + LTp * LTp;
        __syncthreads();
        if (DIAG) {
            if (tid < NS * LTp) {
                const int o = tid / LTp, k = tid - o * LTp;
                partG[((size_t)o * LTp + k) * LTp + k] = 2.f * (float)v4;
                if (o == 0) partP[(size_t)k * LTp + k] = (float)v5;
            }
        } else if (MAXI == 1 && l.nitems_elbo <= NT) {
            for (int base = 0; base < nchunk; ++base) {
                __syncthreads();
                if (active && chunk == base)
#pragma unroll
                    for (int e = 0; e < 16; ++e) scratch[item0 * 16 + e] = (base == 0 ? 0.f : scratch[item0 * 16 + e]) + acc[0][e];
            }
            __syncthreads();
            for (int idx = tid; idx < l.nitems_elbo * 16; idx += NT) {
                const int it = idx / 16, e = idx - it * 16;
                const bool isG = it < l.nG_elbo;
                const int itx = isG ? it : it - l.nG_elbo;
                const int o = isG ? itx / (nb * nb) : 0;
                const int r2 = itx - o * nb * nb;
                const int kb = r2 / nb, jb = r2 - kb * nb;
                const int row = kb * 4 + e / 4, col = jb * 4 + (e & 3);
                if (isG) partG[((size_t)o * LTp + row) * LTp + col] = 2.f * scratch[idx];
                else partP[(size_t)row * LTp + col] = scratch[idx];
            }
        } else {
#pragma unroll
            for (int s = 0; s < MAXI; ++s) {
                const int it = item0 + s * NT;
                if (it < l.nitems_elbo) {
                    const bool isG = it < l.nG_elbo;
                    const int itx = isG ? it : it - l.nG_elbo;
                    const int o = isG ? itx / (nb * nb) : 0;
                    const int r2 = itx - o * nb * nb;
                    const int kb = r2 / nb, jb = r2 - kb * nb;
#pragma unroll
                    for (int e = 0; e < 16; ++e) {
                        const int row = kb * 4 + e / 4, col = jb * 4 + (e & 3);
                        if (isG) partG[((size_t)o * LTp + row) * LTp + col] = 2.f * acc[s][e];
                        else partP[(size_t)row * LTp + col] = acc[s][e];
                    }
                }
            }
        }
        if (tid < NS * LTp) {
            const int o = tid / LTp, k = tid - o * LTp;
            partV[(o * 3 + 0) * LTp + k] = (float)v1;
            partV[(o * 3 + 1) * LTp + k] = (float)v2;
            partV[(o * 3 + 2) * LTp + k] = (float)v3;
        }
    }
    const double s = block_sum_double(Eacc, dred);
    if (tid == 0) reinterpret_cast<double*>(ws + l.o_Epart)[gc.bid] = s;
}

template <int MAXI, bool DIAG>
__global__ __launch_bounds__(NT) void k_elbo(Lay l, const float* __restrict__ y, float* __restrict__ ws) {
    d_elbo<MAXI, DIAG>(GridCtx{(int)blockIdx.x, (int)gridDim.x}, l, y, ws);
}

// ------------------------------------------------------------------------------------------
// reductions, chain rules, optimiser
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void d_reduce_scalars(const GridCtx gc, Lay l, const float* __restrict__ ws, double* __restrict__ scal) {
    __shared__ double red[NT / 64];
    const double* Ep = reinterpret_cast<const double*>(ws + l.o_Epart);
    const double* Kp = reinterpret_cast<const double*>(ws + l.o_KLpart);
    double e = 0.0, k = 0.0;
    // up to four partial sums of either kind per thread in flight at once; added in index order
    for (int i0 = threadIdx.x; i0 < l.nblk_E || i0 < l.nblk_dm; i0 += 4 * NT) {
        double ve[4], vk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int i = i0 + u * NT;
            ve[u] = i < l.nblk_E ? Ep[i] : 0.0;
            vk[u] = i < l.nblk_dm ? Kp[i] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { e += ve[u]; k += vk[u]; }
    }
    const double es = block_sum_double(e, red);
    const double ks = block_sum_double(k, red);
    if (threadIdx.x == 0) { scal[V2VAE_SCAL_E] = es; scal[V2VAE_SCAL_KL] = ks; }
}

__global__ __launch_bounds__(NT) void k_reduce_scalars(Lay l, const float* __restrict__ ws, double* __restrict__ scal) {
    d_reduce_scalars(GridCtx{(int)blockIdx.x, (int)gridDim.x}, l, ws, scal);
}

__device__ __forceinline__ void d_finalize_scalars(const GridCtx gc, double W, double* __restrict__ scal) {
    if (threadIdx.x == 0 && gc.bid == 0) {
        const double E = scal[V2VAE_SCAL_E], KL = scal[V2VAE_SCAL_KL];
        scal[V2VAE_SCAL_LOSS] = KL + W * log(E);
        scal[V2VAE_SCAL_LAMBDA] = W / E;
        scal[V2VAE_SCAL_W] = W;
        // failure detection: steps whose exponent term was not a positive finite number or whose KL term was not finite (the
        // loss and every gradient of such a step are meaningless); a running count the caller reads when it next synchronises
        if (!(E > 0.0) || !isfinite(E) || !isfinite(KL)) scal[V2VAE_SCAL_BAD_STEPS] += 1.0;
    }
}

// single-GPU step: both in one launch (thread 0 writes E and KL and then reads them back itself)
__global__ __launch_bounds__(NT) void k_reduce_finalize_scalars(Lay l, const float* ws, double* scal) {
    const GridCtx one{0, 1};
    d_reduce_scalars(one, l, ws, scal);
    d_finalize_scalars(one, (double)l.W, scal);
}

__global__ void k_finalize_scalars(double W, double* __restrict__ scal) {
    d_finalize_scalars(GridCtx{(int)blockIdx.x, (int)gridDim.x}, W, scal);
}

// records [first, first + per) of `part` summed into out[blockIdx.y][.] (fixed order, double accumulation)
__device__ __forceinline__ void d_sum_records(const GridCtx gc, int chunk, const float* __restrict__ part, int rec, int nrec,
                                              int per, float* __restrict__ out) {
    const int first = chunk * per;
    const int last = first + per < nrec ? first + per : nrec;
    for (int i = gc.bid * NT + threadIdx.x; i < rec; i += gc.nb * NT) {
        // the records are requested sixteen (then four) at a time and added in record order: one memory latency per group
        // instead of one per record, the same sum bit for bit
        double s = 0.0;
        int b = first;
        for (; b + 16 <= last; b += 16) {
            float v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = part[(size_t)(b + u) * rec + i];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += (double)v[u];
        }
        for (; b + 4 <= last; b += 4) {
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = part[(size_t)(b + u) * rec + i];
#pragma unroll
            for (int u = 0; u < 4; ++u) s += (double)v[u];
        }
        for (; b < last; ++b) s += (double)part[(size_t)b * rec + i];
        out[(size_t)chunk * rec + i] = (float)s;
    }
}

__global__ __launch_bounds__(NT) void k_sum_records(const float* __restrict__ part, int rec, int nrec, int per,
                                                    float* __restrict__ out) {
    d_sum_records(GridCtx{(int)blockIdx.x, (int)gridDim.x}, (int)blockIdx.y, part, rec, nrec, per, out);
}

// same two-stage order as launch_sum_records, executed by one workgroup
__device__ __forceinline__ void d_sum_records_all(const float* part, int rec, int nrec, float* stage, float* tot) {
    if (rec <= 0) return;
    const GridCtx one{0, 1};
    if (nrec <= SUM_CHUNKS) {
        d_sum_records(one, 0, part, rec, nrec, nrec, tot);
        return;
    }
    const int per = (nrec + SUM_CHUNKS - 1) / SUM_CHUNKS;
    const int chunks = (nrec + per - 1) / per;
    for (int c = 0; c < chunks; ++c) d_sum_records(one, c, part, rec, nrec, per, stage);
    __syncthreads();
    d_sum_records(one, 0, stage, rec, chunks, chunks, tot);
}

void launch_sum_records(const Lay& l, const float* part, int rec, int nrec, float* stage, float* tot, hipStream_t st) {
    (void)l;
    if (rec <= 0) return;
    const int gx = (rec + NT - 1) / NT;
    if (nrec <= SUM_CHUNKS) {
        hipLaunchKernelGGL(k_sum_records, dim3(gx, 1), dim3(NT), 0, st, part, rec, nrec, nrec, tot);
        return;
    }
    const int per = (nrec + SUM_CHUNKS - 1) / SUM_CHUNKS;
    const int chunks = (nrec + per - 1) / per;
    hipLaunchKernelGGL(k_sum_records, dim3(gx, chunks), dim3(NT), 0, st, part, rec, nrec, per, stage);
    hipLaunchKernelGGL(k_sum_records, dim3(gx, 1), dim3(NT), 0, st, (const float*)stage, rec, chunks, chunks, tot);
}

// Up to three record sets summed by one launch per stage (same chunking and order as launch_sum_records).
struct SumSet {
    const float* part;
    float* stage;
    float* tot;
    float* copy_to;   // the totals also go here (set or added, `copy_acc`) by the stage that forms them: no copy launch
    int rec, nrec, per, chunks, copy_acc;
};
struct SumSets {
    SumSet s[3];
    int n;
};

__global__ __launch_bounds__(NT) void k_sum_multi(SumSets ss, int second) {
    const SumSet q = ss.s[blockIdx.z];
    const GridCtx gc{(int)blockIdx.x, (int)gridDim.x};
    bool final_stage = false;
    if (!second) {
        if ((int)blockIdx.y >= q.chunks) return;
        d_sum_records(gc, (int)blockIdx.y, q.part, q.rec, q.nrec, q.per, q.chunks > 1 ? q.stage : q.tot);
        final_stage = q.chunks == 1;
    } else if (q.chunks > 1) {
        d_sum_records(gc, 0, q.stage, q.rec, q.chunks, q.chunks, q.tot);
        final_stage = true;
    }
    // every thread re-reads only the totals it has just written itself (same index mapping as d_sum_records)
    if (final_stage && q.copy_to)
        for (int i = gc.bid * NT + threadIdx.x; i < q.rec; i += gc.nb * NT) q.copy_to[i] = q.copy_acc ? q.copy_to[i] + q.tot[i] : q.tot[i];
}

static void sum_sets_add(SumSets& ss, const float* part, int rec, int nrec, float* stage, float* tot, float* copy_to = nullptr,
                         int copy_acc = 0) {
    if (rec <= 0) return;
    SumSet& q = ss.s[ss.n++];
    q.part = part; q.stage = stage; q.tot = tot; q.rec = rec; q.nrec = nrec; q.copy_to = copy_to; q.copy_acc = copy_acc;
    if (nrec <= SUM_CHUNKS) { q.per = nrec; q.chunks = 1; }
    else { q.per = (nrec + SUM_CHUNKS - 1) / SUM_CHUNKS; q.chunks = (nrec + q.per - 1) / q.per; }
}

static void sum_sets_launch(const SumSets& ss, hipStream_t st) {
    if (ss.n == 0) return;
    int gx = 1, gy = 1;
    bool two = false;
    for (int i = 0; i < ss.n; ++i) {
        const int g = (ss.s[i].rec + NT - 1) / NT;
        if (g > gx) gx = g;
        if (ss.s[i].chunks > gy) gy = ss.s[i].chunks;
        two = two || ss.s[i].chunks > 1;
    }
    hipLaunchKernelGGL(k_sum_multi, dim3(gx, gy, ss.n), dim3(NT), 0, st, ss, 0);
    if (two) hipLaunchKernelGGL(k_sum_multi, dim3(gx, 1, ss.n), dim3(NT), 0, st, ss, 1);
}

// dE/dHs[o][a][b] on original stacked taps from the per-phase compact totals
__device__ __forceinline__ float GS_value(const Lay& l, const float* __restrict__ ws, int o, int a, int b) {
    int ra, ka, rb, kb;
    tap_to_compact(l, a, &ra, &ka);
    tap_to_compact(l, b, &rb, &kb);
    if (ra != rb) return 0.f;
    const float* tot = ws + l.o_tot_elbo + (size_t)ra * rec_elbo_phase(l);
    const float* G = tot;
    const float* P = tot + l.ns * l.LTp * l.LTp;
    const float* V = P + l.LTp * l.LTp;
    const float Hv = ws[l.o_Hc + (((size_t)ra * l.ns + o) * l.LTp + ka) * l.LTp + kb];
    float v = G[((size_t)o * l.LTp + ka) * l.LTp + kb] + 4.f * Hv * P[(size_t)ka * l.LTp + kb];
    if (a == b) {
        v += V[(o * 3 + 1) * l.LTp + ka];
        if (l.mode != 0 || l.chan != 0) v += V[(o * 3 + 2) * l.LTp + ka];
    }
    return v;
}

__device__ __forceinline__ float g1_value(const Lay& l, const float* __restrict__ ws, int o, int a, int which) {
    int r, k;
    tap_to_compact(l, a, &r, &k);
    const float* V = ws + l.o_tot_elbo + (size_t)r * rec_elbo_phase(l) + l.ns * l.LTp * l.LTp + l.LTp * l.LTp;
    return V[(o * 3 + which) * l.LTp + k];
}

__device__ __forceinline__ void d_chain_channel(const GridCtx gc, Lay l, const float* __restrict__ ws, const double* __restrict__ scal,
                                                      const float* __restrict__ grad_scale, float* __restrict__ g_h1,
                                                      float* __restrict__ g_h2) {
    const float lam = (float)scal[V2VAE_SCAL_LAMBDA] * (grad_scale ? grad_scale[0] : 1.f);
    const int L = l.L;
    if (l.chan) {
        // polynomial channel models: dE/da[i] is the first-order record, dE/db[i] the diagonal of the kernel gradient
        const float* raw = ws + l.o_chanraw;
        const int n2c = l.chan == 2 ? 2 : L;
        for (int idx = gc.bid * NT + threadIdx.x; idx < L + n2c; idx += gc.nb * NT) {
            if (l.chan == 1) {
                if (idx < L) { if (g_h1) g_h1[idx] = lam * g1_value(l, ws, 0, idx, 0); }
                else if (g_h2) g_h2[idx - L] = lam * GS_value(l, ws, 0, idx - L, idx - L);
            } else if (idx < L) {
                // h enters as a = alpha1 h and b = alpha2 h (lib/real_equalization.py:1146-1147)
                if (g_h1) g_h1[idx] = lam * (raw[L] * g1_value(l, ws, 0, idx, 0) + raw[L + 1] * GS_value(l, ws, 0, idx, idx));
            } else if (g_h2) {
                float acc = 0.f;
                for (int i = 0; i < L; ++i)
                    acc = fmaf(raw[i], idx == L ? g1_value(l, ws, 0, i, 0) : GS_value(l, ws, 0, i, i), acc);
                g_h2[idx - L] = lam * acc;
            }
        }
        return;
    }
    const int n1 = l.ns * L, n2 = l.ns * L * L;
    for (int idx = gc.bid * NT + threadIdx.x; idx < n1 + n2; idx += gc.nb * NT) {
        if (idx < n1) {
            if (!g_h1) continue;
            float v;
            if (l.mode == 0) v = g1_value(l, ws, 0, idx, 0);
            else {
                const int p = idx / L, i = idx - p * L;
                v = (p == 0) ? g1_value(l, ws, 0, i, 0) + g1_value(l, ws, 1, L + i, 0)
                             : -g1_value(l, ws, 0, L + i, 0) + g1_value(l, ws, 1, i, 0);
            }
            g_h1[idx] = lam * v;
        } else {
            if (!g_h2) continue;
            const int e = idx - n1;
            const int p = e / (L * L), rem = e - p * L * L;
            const int i = rem / L, j = rem - i * L;
            float v;
            if (l.mode == 0) {
                v = GS_value(l, ws, 0, i, j) + GS_value(l, ws, 0, j, i);
                if (i == j) v += g1_value(l, ws, 0, i, 2);
            } else {
                const int ai = i, bi = L + i, aj = j, bj = L + j;
                if (p == 0)
                    v = (GS_value(l, ws, 0, ai, aj) - GS_value(l, ws, 0, bi, bj) + GS_value(l, ws, 1, ai, bj) + GS_value(l, ws, 1, bi, aj)) +
                        (GS_value(l, ws, 0, aj, ai) - GS_value(l, ws, 0, bj, bi) + GS_value(l, ws, 1, aj, bi) + GS_value(l, ws, 1, bj, ai));
                else
                    v = (-GS_value(l, ws, 0, ai, bj) - GS_value(l, ws, 0, bi, aj) + GS_value(l, ws, 1, ai, aj) - GS_value(l, ws, 1, bi, bj)) +
                        (-GS_value(l, ws, 0, aj, bi) - GS_value(l, ws, 0, bj, ai) + GS_value(l, ws, 1, aj, ai) - GS_value(l, ws, 1, bj, bi));
            }
            g_h2[e] = lam * v;
        }
    }
}

__global__ __launch_bounds__(NT) void k_chain_channel(Lay l, const float* __restrict__ ws, const double* __restrict__ scal,
                                                      const float* __restrict__ grad_scale, float* __restrict__ g_h1,
                                                      float* __restrict__ g_h2) {
    d_chain_channel(GridCtx{(int)blockIdx.x, (int)gridDim.x}, l, ws, scal, grad_scale, g_h1, g_h2);
}

__device__ __forceinline__ void d_chain_equaliser(const GridCtx gc, Lay l, const float* __restrict__ ws, float* __restrict__ g_k1,
                                                        float* __restrict__ g_k2) {
    const float* K2t = ws + l.o_tot_eq;
    const float* K1t = K2t + (size_t)l.ns * l.LT2p * l.LT2p;
    const int l1 = l.l1, l2 = l.l2, P = l.LT2p;
    const int n1 = l.ns * l1, n2 = l.ns * l2 * l2;
    auto K2 = [&](int o, int a, int b) { return K2t[((size_t)o * P + a) * P + b]; };
    auto K1 = [&](int o, int c, int j) { return K1t[(o * l.ns + c) * l1 + j]; };
    for (int idx = gc.bid * NT + threadIdx.x; idx < n1 + n2; idx += gc.nb * NT) {
        if (idx < n1) {
            if (!g_k1) continue;
            float v;
            if (l.mode == 0) v = K1(0, 0, idx);
            else {
                const int p = idx / l1, j = idx - p * l1;
                v = (p == 0) ? K1(0, 0, j) + K1(1, 1, j) : -K1(0, 1, j) + K1(1, 0, j);
            }
            g_k1[idx] = v;
        } else {
            if (!g_k2) continue;
            const int e = idx - n1;
            const int p = e / (l2 * l2), rem = e - p * l2 * l2;
            const int i = rem / l2, j = rem - i * l2;
            float v;
            if (l.mode == 0) v = K2(0, i, j) + K2(0, j, i);
            else {
                const int ai = i, bi = l2 + i, aj = j, bj = l2 + j;
                if (p == 0)
                    v = (K2(0, ai, aj) - K2(0, bi, bj) + K2(1, ai, bj) + K2(1, bi, aj)) +
                        (K2(0, aj, ai) - K2(0, bj, bi) + K2(1, aj, bi) + K2(1, bj, ai));
                else
                    v = (-K2(0, ai, bj) - K2(0, bi, aj) + K2(1, ai, aj) - K2(1, bi, bj)) +
                        (-K2(0, aj, bi) - K2(0, bj, ai) + K2(1, aj, ai) - K2(1, bj, bi));
            }
            g_k2[e] = v;
        }
    }
}

__global__ __launch_bounds__(NT) void k_chain_equaliser(Lay l, const float* __restrict__ ws, float* __restrict__ g_k1,
                                                        float* __restrict__ g_k2) {
    d_chain_equaliser(GridCtx{(int)blockIdx.x, (int)gridDim.x}, l, ws, g_k1, g_k2);
}

// accumulate != 0: dst += src (a gradient the caller never resets, lib/real_equalization.py:1169-1178, :1356-1364)
__device__ __forceinline__ void d_copy(const GridCtx gc, const float* __restrict__ src, float* __restrict__ dst, int n,
                                       int accumulate = 0) {
    for (int i = gc.bid * NT + threadIdx.x; i < n; i += gc.nb * NT) dst[i] = accumulate ? dst[i] + src[i] : src[i];
}

__device__ __forceinline__ void d_adam(const GridCtx gc, int64_t n, float* __restrict__ p, const float* __restrict__ g,
                                       float* __restrict__ m, float* __restrict__ v, float* __restrict__ vmax, float w1,
                                       float beta2, float w2, float step_size, float bc2_sqrt, float eps,
                                       const double* __restrict__ scal, float* __restrict__ noise_var,
                                       float* __restrict__ loss_out) {
    for (int64_t i = (int64_t)gc.bid * NT + threadIdx.x; i < n; i += (int64_t)gc.nb * NT) {
        const float gi = g[i];
        const float mi = fmaf(w1, gi - m[i], m[i]);          // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = fmaf(w2 * gi, gi, v[i] * beta2);    // mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        const float vm = fmaxf(vmax[i], vi);
        const float denom = sqrtf(vm) / bc2_sqrt + eps;
        m[i] = mi; v[i] = vi; vmax[i] = vm;
        p[i] = p[i] - step_size * (mi / denom);
    }
    if (gc.bid == 0 && threadIdx.x == 0 && scal) {
        if (noise_var) noise_var[0] = (float)(scal[V2VAE_SCAL_E] / scal[V2VAE_SCAL_W]);
        if (loss_out) loss_out[0] = (float)scal[V2VAE_SCAL_LOSS];
    }
}

__global__ __launch_bounds__(NT) void k_adam(int64_t n, float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                             float* __restrict__ v, float* __restrict__ vmax, float w1, float beta2,
                                             float w2, float step_size, float bc2_sqrt, float eps,
                                             const double* __restrict__ scal, float* __restrict__ noise_var,
                                             float* __restrict__ loss_out) {
    d_adam(GridCtx{(int)blockIdx.x, (int)gridDim.x}, n, p, g, m, v, vmax, w1, beta2, w2, step_size, bc2_sqrt, eps, scal, noise_var,
           loss_out);
}


// ------------------------------------------------------------------------------------------
// persistent fit: one workgroup per model runs the whole mini-batch loop
// ------------------------------------------------------------------------------------------
// Everything a model's workgroup reads back after writing it (tables, moments, partial records, scalars,
// parameters, optimiser state) goes through pointers that are made opaque per-lane values here, so those reads
// are ordinary vector loads ordered by the workgroup barriers between stages; only the never-written inputs
// (samples, levels, prior, schedule) keep wave-uniform addresses.
// The opaque copy is made of the pointer typed as GLOBAL, so the accesses stay global_load / global_store (a generic "flat"
// access also counts on the LDS counter, so every wait for an LDS operand would wait for it too).
template <typename T>
__device__ __forceinline__ T* per_lane(T* p) {
    typedef T __attribute__((address_space(1))) * gptr;
    gptr g = (gptr)p;
    asm volatile("" : "+v"(g));
    return (T*)g;
}

template <int CB, int MAXI_EQ>
__global__ __launch_bounds__(NT) void k_fit_persistent(Lay l, v2vae_fit_args a, int64_t ws_floats, int64_t n0, int64_t n1,
                                                       int64_t n2, int64_t n3, int64_t n4) {
    const int m = blockIdx.x;
    const int64_t pack = n0 + n1 + n2 + n3 + n4;
    const int64_t B = l.sym_cnt;
    float* ws = per_lane(reinterpret_cast<float*>(a.ws) + (int64_t)m * ws_floats);
    float* prm = per_lane(a.params + (int64_t)m * pack);
    float* g = per_lane(a.grads + (int64_t)m * pack);
    float* am = per_lane(a.adam_m + (int64_t)m * pack);
    float* av = per_lane(a.adam_v + (int64_t)m * pack);
    float* avm = per_lane(a.adam_vmax + (int64_t)m * pack);
    float* nv = per_lane(a.noise_var + m);
    double* scal = per_lane(a.scal + (int64_t)m * V2VAE_SCAL_COUNT);
    float* xh = per_lane(a.xhat + (int64_t)m * l.ns * B);
    float* gx = per_lane(a.gxhat + (int64_t)m * l.ns * B);
    const float* ym = a.y + (int64_t)m * a.y_model_stride;
    float* h1 = prm;
    float* h2 = h1 + n0;
    float* k1 = h2 + n1;
    float* k2 = n3 ? k1 + n2 : nullptr;
    float* sc = k1 + n2 + n3;
    const GridCtx one{0, 1};
    const int nt_eqf = (int)((B + EQ_TB - 1) / EQ_TB);
    const float w1 = (float)(1.0 - (double)a.beta1), w2 = (float)(1.0 - (double)a.beta2);
    const int64_t nadam = n0 + n1 + n2 + n3 + (a.learn_scaling ? n4 : 0);
    for (int n = 0; n < a.n_batches; ++n) {
        const float* yb = ym + (int64_t)n * B * l.S;
        d_prep_tables(one, l, h1, h2, k1, k2, ws);
        __syncthreads();
        for (int t = 0; t < nt_eqf; ++t) {
            d_eq_fwd(GridCtx{t, nt_eqf}, l, yb, ws, xh);
            __syncthreads();
        }
        for (int b = 0; b < l.nblk_dm; ++b) {
            // the same instantiation as the per-batch launch takes (bit-identical results)
            if (l.C == CB && l.ns == 1) d_demap_moments<CB, 1>(GridCtx{b, l.nblk_dm}, l, xh, a.levels, a.prior, sc, nv, ws, nullptr);
            else if (l.C == CB && l.ns == 2) d_demap_moments<CB, 2>(GridCtx{b, l.nblk_dm}, l, xh, a.levels, a.prior, sc, nv, ws, nullptr);
            else d_demap_moments<CB, 0>(GridCtx{b, l.nblk_dm}, l, xh, a.levels, a.prior, sc, nv, ws, nullptr);
            __syncthreads();
        }
        for (int b = 0; b < l.nblk_elbo; ++b) {
            if (l.chan) d_elbo<1, true>(GridCtx{b, l.nblk_elbo}, l, yb, ws);
            else d_elbo<1, false>(GridCtx{b, l.nblk_elbo}, l, yb, ws);
            __syncthreads();
        }
        d_reduce_scalars(one, l, ws, scal);
        __syncthreads();
        d_finalize_scalars(one, (double)l.W, scal);
        __syncthreads();
        for (int b = 0; b < l.nblk_sym; ++b) {
            if (l.C == CB && l.ns == 1) d_sym_bwd<CB, 1>(GridCtx{b, l.nblk_sym}, l, xh, a.levels, a.prior, sc, nv, scal, nullptr, ws, gx);
            else if (l.C == CB && l.ns == 2) d_sym_bwd<CB, 2>(GridCtx{b, l.nblk_sym}, l, xh, a.levels, a.prior, sc, nv, scal, nullptr, ws, gx);
            else d_sym_bwd<CB, 0>(GridCtx{b, l.nblk_sym}, l, xh, a.levels, a.prior, sc, nv, scal, nullptr, ws, gx);
            __syncthreads();
        }
        for (int b = 0; b < l.nblk_eq; ++b) {
            d_eq_bwd<MAXI_EQ>(GridCtx{b, l.nblk_eq}, l, yb, gx, ws);
            __syncthreads();
        }
        d_sum_records_all(ws + l.o_part_elbo, l.rec_elbo, l.nblk_elbo, ws + l.o_sumstage, ws + l.o_tot_elbo);
        __syncthreads();
        d_chain_channel(one, l, ws, scal, nullptr, g, g + n0);
        d_sum_records_all(ws + l.o_part_sym, l.rec_sym, l.nblk_sym, ws + l.o_sumstage, ws + l.o_tot_sym);
        __syncthreads();
        d_copy(one, ws + l.o_tot_sym, g + n0 + n1 + n2 + n3, l.rec_sym, a.accumulate_scaling_grad);
        d_sum_records_all(ws + l.o_part_eq, l.rec_eq, l.nblk_eq, ws + l.o_sumstage, ws + l.o_tot_eq);
        __syncthreads();
        d_chain_equaliser(one, l, ws, g + n0 + n1, n3 ? g + n0 + n1 + n2 : nullptr);
        __syncthreads();
        int64_t first = 0;
        if (a.freeze_h2) {
            d_adam(one, n0, prm, g, am, av, avm, w1, a.beta2, w2, a.step_size[n], a.bc2_sqrt[n], a.eps, nullptr, nullptr, nullptr);
            first = n0 + n1;
        }
        d_adam(one, nadam - first, prm + first, g + first, am + first, av + first, avm + first, w1, a.beta2, w2, a.step_size[n],
               a.bc2_sqrt[n], a.eps, scal, a.adaptive_noise ? nv : nullptr,
               a.loss_curve ? a.loss_curve + (int64_t)m * a.n_batches + n : nullptr);
        if (a.y_eq) {
            const int64_t tot = (int64_t)a.n_batches * B;
            for (int idx = threadIdx.x; idx < l.ns * B; idx += NT) {
                const int c = idx / (int)B, i = idx - c * (int)B;
                a.y_eq[((int64_t)m * l.ns + c) * tot + (int64_t)n * B + i] = xh[idx];
            }
        }
        __syncthreads();
    }
}

static thread_local char g_last_error[256] = "";

static inline int launch_status_at(const char* where) {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return V2VAE_OK;
    snprintf(g_last_error, sizeof(g_last_error), "%s: %s", where, hipGetErrorString(e));
    return V2VAE_ELAUNCH;
}
#define launch_status() launch_status_at(__func__)

template <typename K>
static inline int allow_lds(K kernel, int bytes) {
    if (bytes <= 64 * 1024) return V2VAE_OK;
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) return V2VAE_OK;
    snprintf(g_last_error, sizeof(g_last_error), "hipFuncSetAttribute(%d bytes of LDS): %s", bytes, hipGetErrorString(e));
    return V2VAE_ELAUNCH;
}

}  // namespace v2vae

using namespace v2vae;

extern "C" {

const char* v2vae_version(void) { return "v2vae-hip 0.1 (gfx950)"; }

const char* v2vae_last_error(void) { return g_last_error; }

int v2vae_check_problem(const v2vae_problem* p) {
    Lay l;
    return make_layout(p, &l);
}

size_t v2vae_workspace_bytes(const v2vae_problem* p) {
    Lay l;
    if (make_layout(p, &l) != V2VAE_OK) return 0;
    return (size_t)l.total_floats * sizeof(float);
}

#ifdef V2VAE_STAMPS
int64_t v2vae_dbg_offset(const v2vae_problem* p) {
    Lay l;
    return make_layout(p, &l) == V2VAE_OK ? l.o_sumstage : -1;
}
#endif

int64_t v2vae_ws_offset_moments(const v2vae_problem* p) {
    Lay l;
    return make_layout(p, &l) == V2VAE_OK ? l.o_mom : -1;
}

int64_t v2vae_ws_offset_gmoments(const v2vae_problem* p) {
    Lay l;
    return make_layout(p, &l) == V2VAE_OK ? l.o_gmom : -1;
}

int64_t v2vae_ws_offset_xhat_float(const v2vae_problem* p) {
    Lay l;
    return (make_layout(p, &l) == V2VAE_OK && l.xlo) ? l.o_xhi : -1;
}

int64_t v2vae_ws_offset_xhat_remainder(const v2vae_problem* p) {
    Lay l;
    return (make_layout(p, &l) == V2VAE_OK && l.xlo) ? l.o_xlo : -1;
}

int v2vae_prep_tables(const v2vae_problem* p, const float* h1, const float* h2, const float* k1, const float* k2,
                      void* ws, void* stream) {
    Lay l;
    int rc = make_layout(p, &l);
    if (rc) return rc;
    // either group may be absent (NULL): channel tables need h1 and h2, equaliser tables k1 (and k2 if l2 > 0)
    if (!ws || (!h1 && !k1) || (h1 && !h2) || (k1 && l.l2 > 0 && !k2)) return V2VAE_EINVAL;
    if (k1 && l.eqcxb_fwd) {
        hipLaunchKernelGGL(k_prep_tables_eq, dim3(16 + PREP_EQCXB_BLOCKS), dim3(NT), 0, (hipStream_t)stream, l, h1, h2, k1, k2,
                           (float*)ws, 16);
    } else {
        hipLaunchKernelGGL(k_prep_tables, dim3(16), dim3(NT), 0, (hipStream_t)stream, l, h1, h2, k1, k2, (float*)ws);
    }
    if (k1 && l.eqsymr) launch_prep_eqsymr(l, k1, k2, (float*)ws, (hipStream_t)stream);
    if (h1 && l.use_cxb) launch_prep_cxb(l, h1, h2, (float*)ws, (hipStream_t)stream);
    return launch_status();
}

int v2vae_eq_fwd(const v2vae_problem* p, const float* y, void* ws, float* xhat, void* stream) {
    Lay l;
    int rc = make_layout(p, &l);
    if (rc) return rc;
    if (!y || !ws || !xhat) return V2VAE_EINVAL;
    if (l.eqfbf) {
        rc = launch_eq_fwd_bf(l, y, (const float*)ws, xhat, (hipStream_t)stream);
        return rc ? rc : launch_status();
    }
    if (l.eqcxb_fwd) {
        rc = launch_eq_fwd_cxb(l, y, (const float*)ws, xhat, (hipStream_t)stream);
        return rc ? rc : launch_status();
    }
    if (l.eqsymr) {
        rc = launch_eq_fwd_symr(l, y, (float*)ws, xhat, (hipStream_t)stream);
        return rc ? rc : launch_status();
    }
    rc = allow_lds(k_eq_fwd, l.lds_eq);
    if (rc) return rc;
    const int nblk = (int)((l.sym_cnt + EQ_TB - 1) / EQ_TB);
    hipLaunchKernelGGL(k_eq_fwd, dim3(nblk), dim3(NT), l.lds_eq, (hipStream_t)stream, l, y, (float*)ws, xhat);
    return launch_status();
}

int v2vae_demap_moments(const v2vae_problem* p, const float* xhat, const float* levels, const float* prior,
                        const float* scaling, const float* noise_var, void* ws, float* q_out, void* stream) {
    Lay l;
    int rc = make_layout(p, &l);
    if (rc) return rc;
    if (!xhat || !levels || !prior || !scaling || !noise_var || !ws) return V2VAE_EINVAL;
#define V2VAE_LAUNCH_DEMAP_X(CB, EX)                                                                                    \
    hipLaunchKernelGGL((k_demap_moments<CB, EX>), dim3(l.nblk_dm),  dim3(NT), 0, (hipStream_t)stream, l, xhat, levels, prior, \
                       scaling, noise_var, (float*)ws, q_out)
#define V2VAE_LAUNCH_DEMAP(CB)                                                     \
    do {                                                                           \
        if (l.C == CB && l.ns == 2) V2VAE_LAUNCH_DEMAP_X(CB, 2);                   \
        else if (l.C == CB && l.ns == 1) V2VAE_LAUNCH_DEMAP_X(CB, 1);              \
        else V2VAE_LAUNCH_DEMAP_X(CB, 0);                                          \
    } while (0)
    if (l.C <= 2) V2VAE_LAUNCH_DEMAP(2);
    else if (l.C <= 4) V2VAE_LAUNCH_DEMAP(4);
    else if (l.C <= 8) V2VAE_LAUNCH_DEMAP(8);
    else V2VAE_LAUNCH_DEMAP(16);
#undef V2VAE_LAUNCH_DEMAP
#undef V2VAE_LAUNCH_DEMAP_X
    return launch_status();
}

int v2vae_elbo_prepass(const v2vae_problem* p, void* ws, void* stream) {
    Lay l;
    int rc = make_layout(p, &l);
    if (rc) return rc;
    if (!ws) return V2VAE_EINVAL;
    if (l.use_cx || l.use_cxb) launch_psum_cx(l, (float*)ws, (hipStream_t)stream);
    return launch_status();
}

int v2vae_decide(const v2vae_problem* p, const float* xhat, const float* levels, const float* scaling,
                 const float* noise_var, int32_t* dec, void* stream) {
    Lay l;
    int rc = make_layout(p, &l);
    if (rc) return rc;
    if (!xhat || !levels || !scaling || !noise_var || !dec) return V2VAE_EINVAL;
#define V2VAE_LAUNCH_DECIDE(CB) \
    hipLaunchKernelGGL(k_decide<CB>, dim3(l.nblk_sym), dim3(NT), 0, (hipStream_t)stream, l, xhat, levels, scaling, noise_var, dec)
    if (l.C <= 2) V2VAE_LAUNCH_DECIDE(2);
    else if (l.C <= 4) V2VAE_LAUNCH_DECIDE(4);
    else if (l.C <= 8) V2VAE_LAUNCH_DECIDE(8);
    else V2VAE_LAUNCH_DECIDE(16);
#undef V2VAE_LAUNCH_DECIDE
    return launch_status();
}

int v2vae_elbo_fused(const v2vae_problem* p, const float* y, void* ws, void* stream) {
    Lay l;
    int rc = make_layout(p, &l);
    if (rc) return rc;
    if (!y || !ws) return V2VAE_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (l.use_cx) {
        rc = launch_elbo_cx(l, y, (float*)ws, st);
        return rc ? rc : launch_status();
    }
    if (l.use_cxb) {
        rc = launch_elbo_cxb(l, y, (float*)ws, st);
        return rc ? rc : launch_status();
    }
#define V2VAE_LAUNCH_ELBO(M, D)                                                                               \
    do {                                                                                                      \
        rc = allow_lds((k_elbo<M, D>), l.lds_elbo);                                                           \
        if (rc) return rc;                                                                                    \
        hipLaunchKernelGGL((k_elbo<M, D>), dim3(l.nblk_elbo), dim3(NT), l.lds_elbo, st, l, y, (float*)ws);   \
    } while (0)
    if (l.chan) V2VAE_LAUNCH_ELBO(1, true);   // diagonal kernel (polynomial channel models): O(taps) per window
    else if (l.maxi_elbo == 1) V2VAE_LAUNCH_ELBO(1, false);
    else if (l.maxi_elbo == 2) V2VAE_LAUNCH_ELBO(2, false);
    else V2VAE_LAUNCH_ELBO(4, false);
#undef V2VAE_LAUNCH_ELBO
    return launch_status();
}

int v2vae_reduce_scalars(const v2vae_problem* p, const void* ws, double* scal, void* stream) {
    Lay l;
    int rc = make_layout(p, &l);
    if (rc) return rc;
    if (!ws || !scal) return V2VAE_EINVAL;
    hipLaunchKernelGGL(k_reduce_scalars, dim3(1), dim3(NT), 0, (hipStream_t)stream, l, (const float*)ws, scal);
    return launch_status();
}

int v2vae_finalize_scalars(const v2vae_problem* p, double* scal, void* stream) {
    Lay l;
    int rc = make_layout(p, &l);
    if (rc) return rc;
    if (!scal) return V2VAE_EINVAL;
    hipLaunchKernelGGL(k_finalize_scalars, dim3(1), dim3(64), 0, (hipStream_t)stream, (double)l.W, scal);
    return launch_status();
}

int v2vae_sym_bwd(const v2vae_problem* p, const float* xhat, const float* levels, const float* prior,
                  const float* scaling, const float* noise_var, const double* scal, const float* grad_scale,
                  void* ws, float* gxhat, void* stream) {
    Lay l;
    int rc = make_layout(p, &l);
    if (rc) return rc;
    if (!xhat || !levels || !prior || !scaling || !noise_var || !scal || !ws || !gxhat) return V2VAE_EINVAL;
#define V2VAE_LAUNCH_SYMB_X(CB, EX)                                                                                \
    hipLaunchKernelGGL((k_sym_bwd<CB, EX>), dim3(l.nblk_sym), dim3(NT), 0, (hipStream_t)stream, l, xhat, levels, prior,   \
                       scaling, noise_var, scal, grad_scale, (float*)ws, gxhat)
#define V2VAE_LAUNCH_SYMB(CB)                                                      \
    do {                                                                           \
        if (l.C == CB && l.ns == 2) V2VAE_LAUNCH_SYMB_X(CB, 2);                    \
        else if (l.C == CB && l.ns == 1) V2VAE_LAUNCH_SYMB_X(CB, 1);               \
        else V2VAE_LAUNCH_SYMB_X(CB, 0);                                           \
    } while (0)
    if (l.C <= 2) V2VAE_LAUNCH_SYMB(2);
    else if (l.C <= 4) V2VAE_LAUNCH_SYMB(4);
    else if (l.C <= 8) V2VAE_LAUNCH_SYMB(8);
    else V2VAE_LAUNCH_SYMB(16);
#undef V2VAE_LAUNCH_SYMB
#undef V2VAE_LAUNCH_SYMB_X
    return launch_status();
}

int v2vae_eq_bwd(const v2vae_problem* p, const float* y, const float* gxhat, void* ws, void* stream) {
    Lay l;
    int rc = make_layout(p, &l);
    if (rc) return rc;
    if (!y || !gxhat || !ws) return V2VAE_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    if (l.eqb16) {
        rc = launch_eq_bwd_cx(l, y, gxhat, (float*)ws, st);
        return rc ? rc : launch_status();
    }
    if (l.use_eqcxb) {
        rc = launch_eq_bwd_cxb(l, y, gxhat, (float*)ws, st);
        return rc ? rc : launch_status();
    }
#define V2VAE_LAUNCH_EQB(M)                                                                                   \
    do {                                                                                                      \
        rc = allow_lds(k_eq_bwd<M>, l.lds_eq);                                                                \
        if (rc) return rc;                                                                                    \
        hipLaunchKernelGGL(k_eq_bwd<M>, dim3(l.nblk_eq), dim3(NT), l.lds_eq, st, l, y, gxhat, (float*)ws);    \
    } while (0)
    if (l.maxi_eq == 1) V2VAE_LAUNCH_EQB(1);
    else if (l.maxi_eq == 2) V2VAE_LAUNCH_EQB(2);
    else if (l.maxi_eq <= 4) V2VAE_LAUNCH_EQB(4);
    else V2VAE_LAUNCH_EQB(10);
#undef V2VAE_LAUNCH_EQB
    return launch_status();
}

// both chain rules of the matrix-core paths in one launch: workgroups [0, nb_ch) the channel kernels, the rest the equaliser's
__global__ __launch_bounds__(NT) void k_chain_both_cx(Lay l, const float* __restrict__ ws, const double* __restrict__ scal,
                                                      const float* __restrict__ grad_scale, float* __restrict__ g_h1,
                                                      float* __restrict__ g_h2, float* __restrict__ g_k1, float* __restrict__ g_k2,
                                                      int nb_ch) {
    if ((int)blockIdx.x < nb_ch) d_chain_channel_cx(l, ws, scal, grad_scale, g_h1, g_h2, (int)blockIdx.x, nb_ch);
    else d_chain_equaliser_cx(l, ws, g_k1, g_k2, (int)blockIdx.x - nb_ch, (int)gridDim.x - nb_ch);
}

int v2vae_finish_grads(const v2vae_problem* p, void* ws, const double* scal, const float* grad_scale, int which,
                       float* g_h1, float* g_h2, float* g_k1, float* g_k2, float* g_scaling, void* stream) {
    Lay l;
    int rc = make_layout(p, &l);
    if (rc) return rc;
    if (!ws || !scal) return V2VAE_EINVAL;
    hipStream_t st = (hipStream_t)stream;
    float* w = (float*)ws;
    SumSets ss{};
    float* stage = w + l.o_sumstage;
    if ((which & 1) && l.use_cx) sum_sets_add(ss, w + l.o_part_cx, l.S * CX_REC, l.nblk_cx, stage, w + l.o_tot_cx);
    else if ((which & 1) && l.use_cxb) sum_sets_add(ss, w + l.o_part_cxb, l.S * l.cxb_rec, l.nblk_cxb, stage, w + l.o_tot_cxb);
    else if (which & 1) sum_sets_add(ss, w + l.o_part_elbo, l.rec_elbo, l.nblk_elbo, stage, w + l.o_tot_elbo);
    if (which & 2) sum_sets_add(ss, w + l.o_part_sym, l.rec_sym, l.nblk_sym, stage + l.sumstage_stride, w + l.o_tot_sym, g_scaling,
                                (which & 8) ? 1 : 0);
    if ((which & 4) && l.eqb16) sum_sets_add(ss, w + l.o_part_eqcx, l.eqb16_rec, l.nblk_eq, stage + 2 * l.sumstage_stride, w + l.o_tot_eqcx);
    else if ((which & 4) && l.use_eqcxb) sum_sets_add(ss, w + l.o_part_eqcxb, l.eqcxb_rec, l.nblk_eq, stage + 2 * l.sumstage_stride, w + l.o_tot_eqcxb);
    else if (which & 4) sum_sets_add(ss, w + l.o_part_eq, l.rec_eq, l.nblk_eq, stage + 2 * l.sumstage_stride, w + l.o_tot_eq);
    sum_sets_launch(ss, st);
    if ((which & 1) && l.use_cx && (which & 4) && l.eqb16) {
        const int n_ch = (l.cx_real ? 1 : 2) * (l.L + l.L * l.L), n_eq = l.ns * (l.l1 + l.l2 * l.l2);
        const int nb_ch = (n_ch + NT - 1) / NT, nb_eq = (n_eq + NT - 1) / NT;
        hipLaunchKernelGGL(k_chain_both_cx, dim3(nb_ch + nb_eq), dim3(NT), 0, st, l, (const float*)w, scal, grad_scale, g_h1, g_h2,
                           g_k1, g_k2, nb_ch);
        return launch_status();
    }
    if ((which & 1) && l.use_cx) {
        launch_chain_channel_cx(l, w, scal, grad_scale, g_h1, g_h2, st);
    } else if ((which & 1) && l.use_cxb) {
        launch_chain_channel_cxb(l, w, scal, grad_scale, g_h1, g_h2, st);
    } else if (which & 1) {
        const int n = l.ns * l.L + l.ns * l.L * l.L;
        hipLaunchKernelGGL(k_chain_channel, dim3((n + NT - 1) / NT), dim3(NT), 0, st, l, (const float*)w, scal, grad_scale, g_h1, g_h2);
    }
    if ((which & 4) && l.eqb16) {
        launch_chain_equaliser_cx(l, w, g_k1, g_k2, st);
    } else if ((which & 4) && l.use_eqcxb) {
        launch_chain_equaliser_cxb(l, w, g_k1, g_k2, st);
    } else if (which & 4) {
        const int n = l.ns * l.l1 + l.ns * l.l2 * l.l2;
        hipLaunchKernelGGL(k_chain_equaliser, dim3((n + NT - 1) / NT), dim3(NT), 0, st, l, (const float*)w, g_k1, g_k2);
    }
    return launch_status();
}

int v2vae_adam_amsgrad(int64_t n, float* params, const float* grads, float* m, float* v, float* vmax, float lr,
                       float beta1, float beta2, float eps, int64_t step, const double* scal, float* noise_var,
                       float* loss_out, void* stream) {
    if (n < 1 || !params || !grads || !m || !v || !vmax || step < 1) return V2VAE_EINVAL;
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const float step_size = (float)((double)lr / bc1);
    const float bc2_sqrt = (float)sqrt(bc2);
    const float w1 = (float)(1.0 - (double)beta1);
    const float w2 = (float)(1.0 - (double)beta2);
    hipLaunchKernelGGL(k_adam, dim3((unsigned)((n + NT - 1) / NT)), dim3(NT), 0, (hipStream_t)stream, n, params, grads, m, v,
                       vmax, w1, beta2, w2, step_size, bc2_sqrt, eps, scal, noise_var, loss_out);
    return launch_status();
}

void v2vae_pack_sizes(const v2vae_problem* p, int64_t out[5]) {
    out[0] = (int64_t)p->ns * p->L;
    out[1] = p->mode == 2 ? (int64_t)p->L : p->mode == 3 ? 2 : (int64_t)p->ns * p->L * p->L;
    out[2] = (int64_t)p->ns * p->l1;
    out[3] = (int64_t)p->ns * p->l2 * p->l2;
    out[4] = (int64_t)p->ns * p->C;
}

namespace {
struct PackPtrs {
    int64_t n[5];
    float *h1, *h2, *k1, *k2, *sc;
    float *g_h1, *g_h2, *g_k1, *g_k2, *g_sc;
};
PackPtrs pack_ptrs(const v2vae_problem* p, const v2vae_train_args* a) {
    PackPtrs q;
    v2vae_pack_sizes(p, q.n);
    q.h1 = a->params;
    q.h2 = q.h1 + q.n[0];
    q.k1 = q.h2 + q.n[1];
    q.k2 = q.n[3] ? q.k1 + q.n[2] : nullptr;
    q.sc = q.k1 + q.n[2] + q.n[3];
    q.g_h1 = a->grads;
    q.g_h2 = q.g_h1 + q.n[0];
    q.g_k1 = q.g_h2 + q.n[1];
    q.g_k2 = q.n[3] ? q.g_k1 + q.n[2] : nullptr;
    q.g_sc = q.g_k1 + q.n[2] + q.n[3];
    return q;
}
}  // namespace

// forward + ELBO partial sums of this rank; leaves the local E and KL in scal[0..1]
int v2vae_train_step_sharded_pre(const v2vae_problem* p, const v2vae_train_args* a, void* stream) {
    if (!p || !a || !a->params || !a->ws || !a->scal) return V2VAE_EINVAL;
    const PackPtrs q = pack_ptrs(p, a);
    int rc;
    if ((rc = v2vae_prep_tables(p, q.h1, q.h2, q.k1, q.k2, a->ws, stream))) return rc;
    if ((rc = v2vae_eq_fwd(p, a->y, a->ws, a->xhat, stream))) return rc;
    if ((rc = v2vae_demap_moments(p, a->xhat, a->levels, a->prior, q.sc, a->noise_var, a->ws, nullptr, stream))) return rc;
    if ((rc = v2vae_elbo_prepass(p, a->ws, stream))) return rc;
    if ((rc = v2vae_elbo_fused(p, a->y, a->ws, stream))) return rc;
    return v2vae_reduce_scalars(p, a->ws, a->scal, stream);
}

// given the GLOBAL E and KL in scal[0..1]: loss, W / E, backward, this rank's gradient sums into a->grads
int v2vae_train_step_sharded_post(const v2vae_problem* p, const v2vae_train_args* a, void* stream) {
    if (!p || !a || !a->params || !a->grads || !a->ws || !a->scal) return V2VAE_EINVAL;
    const PackPtrs q = pack_ptrs(p, a);
    int rc;
    if ((rc = v2vae_finalize_scalars(p, a->scal, stream))) return rc;
    if ((rc = v2vae_sym_bwd(p, a->xhat, a->levels, a->prior, q.sc, a->noise_var, a->scal, nullptr, a->ws, a->gxhat, stream))) return rc;
    if ((rc = v2vae_eq_bwd(p, a->y, a->gxhat, a->ws, stream))) return rc;
    return v2vae_finish_grads(p, a->ws, a->scal, nullptr, 7, q.g_h1, q.g_h2, q.g_k1, q.g_k2, q.g_sc, stream);
}

// Adam-AMSGrad on the sections the model trains, from a->grads (after the caller's all-reduce in the sharded step)
int v2vae_train_step_update(const v2vae_problem* p, const v2vae_train_args* a, void* stream) {
    if (!p || !a || !a->params || !a->grads) return V2VAE_EINVAL;
    int64_t n[5];
    v2vae_pack_sizes(p, n);
    float* g = a->grads;
    const int64_t nadam = n[0] + n[1] + n[2] + n[3] + (a->learn_scaling ? n[4] : 0);
    int64_t first = 0;
    int rc;
    if (a->freeze_h2) {  // update [h1] and [k1 | k2 | scaling] only
        if ((rc = v2vae_adam_amsgrad(n[0], a->params, g, a->adam_m, a->adam_v, a->adam_vmax, a->lr, a->beta1, a->beta2, a->eps,
                                     a->step, nullptr, nullptr, nullptr, stream))) return rc;
        first = n[0] + n[1];
    }
    return v2vae_adam_amsgrad(nadam - first, a->params + first, g + first, a->adam_m + first, a->adam_v + first,
                              a->adam_vmax + first, a->lr, a->beta1, a->beta2, a->eps, a->step, a->scal,
                              a->adaptive_noise ? a->noise_var : nullptr, a->loss_out, stream);
}

int v2vae_train_step(const v2vae_problem* p, const v2vae_train_args* a, void* stream) {
    if (!p || !a) return V2VAE_EINVAL;
    const PackPtrs q = pack_ptrs(p, a);
    int rc;
    if ((rc = v2vae_prep_tables(p, q.h1, q.h2, q.k1, q.k2, a->ws, stream))) return rc;
    if ((rc = v2vae_eq_fwd(p, a->y, a->ws, a->xhat, stream))) return rc;
    if ((rc = v2vae_demap_moments(p, a->xhat, a->levels, a->prior, q.sc, a->noise_var, a->ws, nullptr, stream))) return rc;
    if ((rc = v2vae_elbo_prepass(p, a->ws, stream))) return rc;
    if ((rc = v2vae_elbo_fused(p, a->y, a->ws, stream))) return rc;
    {
        Lay l;
        if ((rc = make_layout(p, &l))) return rc;
        if (!a->ws || !a->scal) return V2VAE_EINVAL;
        hipLaunchKernelGGL(k_reduce_finalize_scalars, dim3(1), dim3(NT), 0, (hipStream_t)stream, l, (const float*)a->ws, a->scal);
    }
    if ((rc = v2vae_sym_bwd(p, a->xhat, a->levels, a->prior, q.sc, a->noise_var, a->scal, nullptr, a->ws, a->gxhat, stream))) return rc;
    if ((rc = v2vae_eq_bwd(p, a->y, a->gxhat, a->ws, stream))) return rc;
    if ((rc = v2vae_finish_grads(p, a->ws, a->scal, nullptr, a->accumulate_scaling_grad ? 15 : 7, q.g_h1, q.g_h2, q.g_k1, q.g_k2,
                                 q.g_sc, stream))) return rc;
    return v2vae_train_step_update(p, a, stream);
}

// ------------------------------------------------------------------------------------------
// v2vae_train_step_static: the whole step with no allocation by anyone.  Workspace, scalar block and the per-symbol
// gradient buffer of v2vae_train_step are carved out of one static device array.  `static_scratch` below is only the
// host-side handle of that array: host code must never pass it to a kernel or a memset; the per-device address comes from
// hipGetSymbolAddress (each GPU has its own copy of the array at its own address) and only that pointer is used.
// ------------------------------------------------------------------------------------------
static __device__ __align__(256) unsigned char static_scratch[V2VAE_STATIC_SCRATCH_BYTES];

namespace {
constexpr int STATIC_MAX_DEVICES = 64;
std::mutex g_static_mutex;
void* g_static_ptr[STATIC_MAX_DEVICES] = {};  // device address of static_scratch per device ordinal, filled at first use

// Address of the copy of the array on the device that owns `stream` (the current device for the default stream), resolved
// once per device with that device current.
int static_scratch_ptr(hipStream_t stream, unsigned char** out) {
    int cur = -1, dev = -1;
    hipError_t e = hipGetDevice(&cur);
    if (e == hipSuccess) {
        dev = cur;
        if (stream) {
            hipDevice_t sd;
            e = hipStreamGetDevice(stream, &sd);
            if (e == hipSuccess) dev = (int)sd;
        }
    }
    if (e != hipSuccess || dev < 0 || dev >= STATIC_MAX_DEVICES) {
        snprintf(g_last_error, sizeof(g_last_error), "v2vae_train_step_static: device of the stream: %s (device %d)",
                 hipGetErrorString(e), dev);
        return V2VAE_ELAUNCH;
    }
    if (dev != cur) {   // kernels are launched with the current device's code object: the stream must belong to it
        snprintf(g_last_error, sizeof(g_last_error), "v2vae_train_step_static: the stream belongs to device %d, the current "
                 "device is %d", dev, cur);
        return V2VAE_EINVAL;
    }
    std::lock_guard<std::mutex> lock(g_static_mutex);
    if (!g_static_ptr[dev]) {
        void* ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(static_scratch));
        if (e != hipSuccess || !ptr) {
            snprintf(g_last_error, sizeof(g_last_error), "v2vae_train_step_static: hipGetSymbolAddress: %s", hipGetErrorString(e));
            return V2VAE_ELAUNCH;
        }
        g_static_ptr[dev] = ptr;
    }
    *out = static_cast<unsigned char*>(g_static_ptr[dev]);
    return V2VAE_OK;
}

// carve of the static array: [scal: 8 doubles | gxhat: ns * sym_cnt floats | workspace], every part 256-byte aligned
struct StaticCarve {
    size_t o_scal, o_gxhat, o_ws, ws_bytes, total;
};
int static_carve(const v2vae_problem* p, StaticCarve* c) {
    Lay l;
    const int rc = make_layout(p, &l);
    if (rc) return rc;
    c->o_scal = 0;
    c->o_gxhat = (size_t)rup(V2VAE_SCAL_COUNT * sizeof(double), 256);
    c->o_ws = c->o_gxhat + (size_t)rup((int64_t)l.ns * l.sym_cnt * sizeof(float), 256);
    c->ws_bytes = (size_t)l.total_floats * sizeof(float);
    c->total = c->o_ws + (size_t)rup((int64_t)c->ws_bytes, 256);
    return V2VAE_OK;
}
}  // namespace

size_t v2vae_static_scratch_bytes(const v2vae_problem* p) {
    StaticCarve c;
    if (!p || static_carve(p, &c) != V2VAE_OK) return 0;
    return c.total;
}

int v2vae_train_step_static(const v2vae_problem* p, const v2vae_train_args* a, void* stream) {
    if (!p || !a) return V2VAE_EINVAL;
    if (a->ws || a->scal || a->gxhat) return V2VAE_EINVAL;  // these three live in the static array
    StaticCarve c;
    int rc = static_carve(p, &c);
    if (rc) return rc;
    if (c.total > (size_t)V2VAE_STATIC_SCRATCH_BYTES) {
        snprintf(g_last_error, sizeof(g_last_error),
                 "v2vae_train_step_static: this problem needs %zu bytes of scratch, the static array holds %zu; use "
                 "v2vae_train_step with a caller-owned workspace", c.total, (size_t)V2VAE_STATIC_SCRATCH_BYTES);
        return V2VAE_ESIZE;
    }
    unsigned char* base = nullptr;
    if ((rc = static_scratch_ptr((hipStream_t)stream, &base))) return rc;
    // the array keeps the previous call's contents: start every call from the all-zero state a fresh workspace has
    const hipError_t e = hipMemsetAsync(base, 0, c.total, (hipStream_t)stream);
    if (e != hipSuccess) {
        snprintf(g_last_error, sizeof(g_last_error), "v2vae_train_step_static: hipMemsetAsync: %s", hipGetErrorString(e));
        return V2VAE_ELAUNCH;
    }
    v2vae_train_args b = *a;
    b.scal = reinterpret_cast<double*>(base + c.o_scal);
    b.gxhat = reinterpret_cast<float*>(base + c.o_gxhat);
    b.ws = base + c.o_ws;
    return v2vae_train_step(p, &b, stream);
}

int v2vae_adam_schedule(int64_t step0, int64_t n, const float* lr, float beta1, float beta2, float* step_size,
                        float* bc2_sqrt) {
    if (step0 < 0 || n < 1 || !lr || !step_size || !bc2_sqrt) return V2VAE_EINVAL;
    for (int64_t i = 0; i < n; ++i) {
        const double step = (double)(step0 + i + 1);
        const double bc1 = 1.0 - pow((double)beta1, step);
        const double bc2 = 1.0 - pow((double)beta2, step);
        step_size[i] = (float)((double)lr[i] / bc1);
        bc2_sqrt[i] = (float)sqrt(bc2);
    }
    return V2VAE_OK;
}

int v2vae_fit_persistent_kind(const v2vae_problem* p) {
    if (!p) return V2VAE_EINVAL;
    Lay l;
    int rc = make_layout(p, &l);
    if (rc) return rc;
    if (l.y_lo != 0 || l.sym_lo != 0 || l.own_lo != 0 || l.y_len != l.n_glob || l.sym_cnt != l.b_glob || l.own_cnt != l.b_glob)
        return V2VAE_EINVAL;
    int64_t n[5];
    v2vae_pack_sizes(p, n);
    if (fit_rs_applicable(l, n[0] + n[1] + n[2] + n[3] + n[4])) return 2;
    if (!l.generic_elbo_ok || l.maxi_elbo != 1 || l.maxi_eq > 2) return V2VAE_ESIZE;
    return 1;
}

int v2vae_fit_persistent(const v2vae_problem* p, const v2vae_fit_args* a, void* stream) {
    if (!p || !a) return V2VAE_EINVAL;
    Lay l;
    int rc = make_layout(p, &l);
    if (rc) return rc;
    // one batch, described as a whole sequence
    if (l.y_lo != 0 || l.sym_lo != 0 || l.own_lo != 0 || l.y_len != l.n_glob || l.sym_cnt != l.b_glob || l.own_cnt != l.b_glob)
        return V2VAE_EINVAL;
    if (a->n_models < 1 || a->n_batches < 1 || !a->y || !a->levels || !a->prior || !a->params || !a->grads || !a->adam_m ||
        !a->adam_v || !a->adam_vmax || !a->noise_var || !a->ws || !a->scal || !a->xhat || !a->gxhat || !a->step_size ||
        !a->bc2_sqrt)
        return V2VAE_EINVAL;
    if (a->y_stream_stride < (int64_t)a->n_batches * l.n_glob || a->y_model_stride < 0) return V2VAE_EINVAL;
    int64_t n[5];
    v2vae_pack_sizes(p, n);
    hipStream_t st = (hipStream_t)stream;
    l.y_stride = a->y_stream_stride;
    {
        const int took = launch_fit_rs(l, a, n, st);  // small real shapes: everything on chip
        if (took < 0) return took;
        if (took == 1) return launch_status();
    }
    if (!l.generic_elbo_ok || l.maxi_elbo != 1 || l.maxi_eq > 2) return V2VAE_ESIZE;
    // the stage-by-stage kernel runs the generic stages only: every switch of the matrix-core paths off
    l.use_cx = 0;
    l.cx_real = 0;
    l.cx_split = 0;
    l.use_cxb = 0;
    l.use_eqcx = 0;
    l.eqb16 = 0;
    l.use_eqcxb = 0;
    l.eqcxb_fwd = 0;
    l.eqsymr = 0;
    l.eqsym_t = 0;
    l.nblk_E = l.nblk_elbo;
    const int lds = l.lds_elbo > l.lds_eq ? l.lds_elbo : l.lds_eq;
#define V2VAE_LAUNCH_FIT(CB, ME)                                                                                       \
    do {                                                                                                               \
        rc = allow_lds(k_fit_persistent<CB, ME>, lds);                                                                 \
        if (rc) return rc;                                                                                             \
        hipLaunchKernelGGL((k_fit_persistent<CB, ME>), dim3(a->n_models), dim3(NT), lds, st, l, *a, l.total_floats,    \
                           n[0], n[1], n[2], n[3], n[4]);                                                              \
    } while (0)
#define V2VAE_LAUNCH_FIT_C(ME)                   \
    do {                                         \
        if (l.C <= 2) V2VAE_LAUNCH_FIT(2, ME);   \
        else if (l.C <= 4) V2VAE_LAUNCH_FIT(4, ME); \
        else if (l.C <= 8) V2VAE_LAUNCH_FIT(8, ME); \
        else V2VAE_LAUNCH_FIT(16, ME);           \
    } while (0)
    if (l.maxi_eq == 1) V2VAE_LAUNCH_FIT_C(1);
    else V2VAE_LAUNCH_FIT_C(2);
#undef V2VAE_LAUNCH_FIT_C
#undef V2VAE_LAUNCH_FIT
    return launch_status();
}

}  // extern "C"
