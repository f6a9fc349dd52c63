// Problem geometry shared by the host entry points and the device kernels.
// Everything a kernel needs besides pointers travels in one `Lay` value (kernel argument).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <hip/hip_runtime.h>

#include "../../include/v2vae_hip.h"

#if defined(__HIPCC__)
#define V2VAE_HD __host__ __device__
#else
#define V2VAE_HD
#endif

namespace v2vae {

constexpr int NT = 256;         // threads per workgroup (4 waves of 64)
constexpr int MAX_C = 16;       // amplitude levels per stream
constexpr int EQ_TB = 256;      // symbols per equaliser tile
constexpr int EQBF_NWD = 192;   // k_eq_bwd_bf: words per plane of bf16 pairs: the elements of one parity of the longest tile
                                // (256 S + 99 samples at 95 lags), in pairs, with slack for the last operand
constexpr int LDS_BUDGET = 163000;  // one workgroup may use (almost) all 160 KiB of a CU
constexpr int SCRATCH_F = 4096; // floats of LDS scratch for cross-chunk reductions
constexpr int MAX_BLOCKS = 512; // persistent workgroups per kernel (2 per CU)
// matrix-core ELBO kernel for the complex model (v2vae_elbo_cx.hip)
constexpr int CX_TP = 33;            // row pitch of the per-wave transposing buffer
constexpr int CX_REC = 576;          // floats per (workgroup, phase) record of the matrix-core ELBO kernels (<= 16 taps): the complex
                                     // 16x16 accumulation sum (rho + i sigma)^T conj(X) as Re [16][16], Im [16][16], then 4 vectors of 16
constexpr int CX_IMG = 2112;         // floats of the block-tile kernel's LDS image of its 32x32 accumulator tile (+ vectors)
constexpr int SUM_CHUNKS = 32;       // first-stage groups of the two-stage partial-record sums
constexpr int CX_BLOCKS = 256;       // one workgroup per CU (the LDS image takes most of a CU)
// Blocked ELBO kernel: waves per workgroup and windows per phase per block tile.  One block plus a tap on the side (17 active
// taps) fits 256 registers per lane without spills: eight waves (two per SIMD) on 256-window tiles; the longer tap counts need
// the 512-register budget of one wave per SIMD: four waves on 512-window tiles.
__host__ __device__ constexpr int cxb_nw(int nbk, int xt) { return (nbk == 1 && xt) ? 8 : 4; }
__host__ __device__ constexpr int cxb_twb(int nbk, int xt) { return (nbk == 1 && xt) ? 256 : 512; }
// the instance of the blocked ELBO kernel that accumulates dE/dH on the bf16 matrix pipe in split arithmetic: the one at two
// waves per SIMD (with one wave there is no second wave whose vector issue an f32 matrix instruction would hold up: no gain)
__host__ __device__ constexpr bool cxb_bfk(int nbk, int xt) { return nbk == 1 && xt; }
constexpr int CXB_STEPS = 48;        // 256-byte rows of one block pair's constant operand image: the bf16 images of the H X and
                                     // H w tables, [2 tables][3 pieces][2 chunks][64 lanes][8 bf16]
constexpr int CXB_TPX = 35;          // row pitch of its transposing buffer when an extra tap rides along (columns 32, 33)
constexpr int CX_MAXLAG = 48;        // lagged products kept for sum s s^T (largest blocked tap count)
constexpr int G2C_D = 4 + 2 * (2 * CX_MAXLAG + 8);   // offset of the real coefficients D[lag] in the table at Lay::o_g2c
constexpr int G2C_FLOATS = G2C_D + 2 * CX_MAXLAG + 8;
constexpr int G2_NE = 8;   // workgroups of k_g2_cx that form the lagged-product term of E (one partial sum each)

struct Lay {
    int ns, mode, S, l1, pad1, l2, L, C;
    // channel model of the real path: 0 full second-order kernel h2 [L][L]; 1 memory polynomial (h2 [L] weighs x^2 per tap:
    // diagonal second-order kernel); 2 Hammerstein (one filter h [L] on alpha1 x + alpha2 x^2; the h2 section is
    // [alpha1, alpha2]).  1 and 2 run the same kernels on tables built with a diagonal kernel.
    int chan;
    int flags;               // v2vae_problem::flags (V2VAE_FLAG_*): A/B switches between two HIP kernels for one shape
    int64_t n_glob, y_lo, y_len, sym_lo, sym_cnt, own_lo, own_cnt;
    int64_t y_stride;  // floats between the ns sample streams (y_len unless a batch is a slice of a longer sequence)
    int pre;                 // (L-1)/2: y' = y[pre + n]
    int64_t b_glob, W, kl_lo, kl_hi;
    int kmax, LE, LEp, LTp;  // compact channel taps per phase
    int p2, LT2, LT2p;       // stacked equaliser lags
    int hl, hr, tile_len;    // equaliser sample tile: [b0*S - hl, b0*S + EQ_TB*S + hr)
    // ELBO tiling
    int TWIN, TS, NG, TSYM, ntile_elbo, nblk_elbo, nitems_elbo, nG_elbo, nchunk_elbo, maxi_elbo;
    int lds_elbo;            // bytes
    int generic_elbo_ok;     // 0: the shape exceeds the generic ELBO kernel's tiling (only the matrix-core kernels cover it)
    // equaliser backward tiling
    int ntile_eq, nblk_eq, nitems_eq, nchunk_eq, maxi_eq, nk1;
    int lds_eq;              // bytes (forward and backward use the same carve)
    int nblk_sym, nblk_dm;   // workgroups of the per-symbol backward and of the soft demapper
    // matrix-core ELBO path
    int use_cx, ntile_cx, nblk_cx, lds_cx, nblk_E, nblk_psum;
    int64_t o_psum_part, o_psum_tot;
    // coefficients of the third-moment gradient as a correlation of the means (k_g2_cx): K0 (Re, Im), the constant
    // fourth-cumulant gradient, 0, then (Re, Im) of C[lag] for lag = t - K4, t = 0 .. 2 K4 + 3, K4 = kmax rounded up to 4;
    // from G2C_D on the real D[lag] of the variance-gradient term that is a correlation of v_a + v_b
    int64_t o_g2c;
    int64_t o_psv;  // [S][LE][LE] lagged sums of v_a + v_b per (phase, tap, tap) over the owned windows (k_g2_cx; read by the chain rule)
    int64_t o_gT;   // [sym_cnt] the part of dE/dv (same for both streams) that is a correlation of v_a + v_b (matrix-core ELBO kernels)
    int64_t o_cxH, o_cxh1, o_part_cx, o_tot_cx;
    // run kernel (k_elbo_cx3): cx_split = 1, every wave owns a run of cx_rl symbols of one phase.  dE/d(moments) then lives in
    // S + 1 arrays of per-symbol 4-vectors at o_gpk, [S + 1][sym_cnt][4]: [r] the gradients of (mean_a, mean_b, var_a, var_b)
    // from the windows of phase r, [S] = (3rd-moment gradients a, b; the part of the variance gradient that is a
    // correlation of v_a + v_b; the 4th-cumulant gradient) from k_g2_cx; the per-symbol backward adds them up
    int cx_split;
    // real model on the run kernel: the quadrature stream is identically zero (its moment planes are allocated and zero-filled by the
    // soft demapper, its samples read as zero) and the kernel adds the terms of the diagonal quirk (dq = diag(h2), d = 2 diag(h2))
    int cx_real;
    int64_t cx_rl, o_gpk;
    int cxb_nwv, cxb_twbv;   // waves per workgroup / windows per tile of the blocked ELBO kernel instance that serves the shape
    // blocked matrix-core ELBO path (complex, 16 < active taps per phase <= 48): taps in blocks of 16
    // cxb_nb blocks give the stride cxb_leb of tables and records; the kernel runs cxb_nbk blocks on the matrix cores plus cxb_xt
    // (0 or 1) extra tap on the side (tap counts 16 cxb_nbk + 1), else cxb_nbk = cxb_nb
    int use_cxb, cxb_nb, cxb_nbk, cxb_xt, cxb_leb, ntile_cxb, nblk_cxb, lds_cxb, cxb_rec;
    int64_t o_cxbA, o_cxbH, o_cxbc, o_part_cxb, o_tot_cxb;
    // matrix-core equaliser path
    int use_eqcx, eqcx_lk, eqcx_steps;
    // backward of the complex equaliser on 16-lag blocks (k_eq_bwd_cx<NB16>): eqb16 = number of blocks (0: not used; l2 <= 79),
    // records [2 planes][16 eqb16][16 eqb16] at o_part_eqcx / o_tot_eqcx
    int eqb16, eqb16_rec, lds_eqb16;
    int eqbf_nb, eqbf_xs;    // bf16 instance: lag blocks on the matrix pipe, and -1 (all of them) / 1 (16 eqbf_nb + 1 lags: the last lag and
                             // the column of ones on the side); the record keeps eqb16 blocks
    int eqbf;                // 1: its instance on the bf16 matrix pipe in split arithmetic (k_eq_bwd_bf: one or two samples per symbol)
    int64_t o_part_eqcx, o_tot_eqcx;
    // blocked matrix-core equaliser path (complex, 31 < l2 <= 95): lag rows in blocks of 32
    int use_eqcxb, eqcxb_fwd, eqcxb_nbl, eqcxb_rec, lds_eqcxb_fwd, lds_eqcxb_bwd;
    int64_t o_eqcxbA, o_part_eqcxb, o_tot_eqcxb;
    // symmetric-form forward equaliser (complex, l2 <= 31): table of the upper-triangular kernel pairs, (Re, Im) each, off-diagonal
    // entries doubled, rows a = 0 .. eqsym_t - 1 of columns b = a .. eqsym_t - 1, then the first-order kernel per lag (Re, Im)
    int eqsym_t;             // compile-time lag count of the kernel instance that serves l2 (0: not used)
    int eqsymr;              // 1: the real model's forward in symmetric form with sums in double (k_eq_fwd_symr); the table then holds doubles
    int64_t o_eqsym;
    // forward of the complex equaliser on the bf16 matrix pipe in split arithmetic (k_eq_fwd_bf: one or two samples per symbol,
    // up to 95 lags): eqfbf_nbk blocks of 16 lags; at o_eqfbf the table images [pairs (R, g >= R)][2 streams][3 pieces][64 lanes]
    // [8 bf16], then the first-order weights on lag indices [16 eqfbf_nbk][2]; eqfbf_xl: the lag count is 16 eqfbf_nbk + 1 and
    // the last lag e is carried on the side - behind the weights then column e of K' [16 eqfbf_nbk][2], K'_ee and the first-order
    // weight of lag e
    int eqfbf, eqfbf_nbk, eqfbf_xl;
    int64_t o_eqfbf;
    int64_t o_sumstage;      // 3 x [SUM_CHUNKS][largest record] staging of the two-stage record sums (one per record set)
    int64_t sumstage_stride; // floats between the three staging areas
    // record sizes of the per-workgroup partial sums (floats)
    int rec_elbo, rec_eq, rec_sym;
    // workspace offsets in floats
    int64_t o_eqK1, o_eqK2, o_h1c, o_Hc, o_HHc, o_dc, o_dqc, o_mom, o_gmom;
    int xlo;                 // 1: the equaliser output carries a float remainder (real path), planes o_xhi (copy of xhat) / o_xlo
    int64_t o_xhi, o_xlo;
    int64_t o_chanraw;       // chan != 0: copy of the raw channel sections [h1 (L) | h2 section] for the chain rule
    int64_t o_part_elbo, o_part_eq, o_part_sym, o_tot_elbo, o_tot_eq, o_tot_sym;
    int64_t o_Epart, o_KLpart;  // doubles, offsets still in floats (8-byte aligned)
    int64_t total_floats;
};

// offsets inside one ELBO record: [S][ G: ns*LTp*LTp | P: LTp*LTp | vec: ns*3*LTp ]
V2VAE_HD inline int rec_elbo_phase(const Lay& l) { return l.ns * l.LTp * l.LTp + l.LTp * l.LTp + l.ns * 3 * l.LTp; }

int make_layout(const v2vae_problem* p, Lay* out);
int launch_eq_fwd_cxb(const Lay& l, const float* y, const float* ws, float* xhat, hipStream_t st);
int launch_eq_fwd_bf(const Lay& l, const float* y, const float* ws, float* xhat, hipStream_t st);   // v2vae_eq_fwd_bf.hip
int launch_prep_eqsymr(const Lay& l, const float* k1, const float* k2, float* ws, hipStream_t st);
int launch_eq_fwd_symr(const Lay& l, const float* y, float* ws, float* xhat, hipStream_t st);
int launch_eq_bwd_cxb(const Lay& l, const float* y, const float* gx, float* ws, hipStream_t st);
void launch_chain_equaliser_cxb(const Lay& l, float* ws, float* g_k1, float* g_k2, hipStream_t st);

// v2vae_elbo_cx.hip
int launch_elbo_cx(const Lay& l, const float* y, float* ws, hipStream_t st);
void launch_psum_cx(const Lay& l, float* ws, hipStream_t st);
void launch_chain_channel_cx(const Lay& l, float* ws, const double* scal, const float* grad_scale, float* g_h1, float* g_h2,
                             hipStream_t st);

// v2vae_elbo_cxb.hip
void launch_prep_cxb(const Lay& l, const float* h1, const float* h2, float* ws, hipStream_t st);
int launch_elbo_cxb(const Lay& l, const float* y, float* ws, hipStream_t st);
void launch_chain_channel_cxb(const Lay& l, float* ws, const double* scal, const float* grad_scale, float* g_h1, float* g_h2,
                              hipStream_t st);

// two-stage deterministic sum of nrec records of `rec` floats: tot[e] = sum_b part[b][e]  (v2vae_kernels.hip)
void launch_sum_records(const Lay& l, const float* part, int rec, int nrec, float* stage, float* tot, hipStream_t st);

// v2vae_eq_cx.hip
int launch_eq_bwd_cx(const Lay& l, const float* y, const float* gx, float* ws, hipStream_t st);
void launch_chain_equaliser_cx(const Lay& l, float* ws, float* g_k1, float* g_k2, hipStream_t st);

// v2vae_fit_rs.hip: whole fit loop on chip for small real shapes; 1 = launched, 0 = shape not covered, < 0 = error
int launch_fit_rs(const Lay& l, const v2vae_fit_args* a, const int64_t n[5], hipStream_t st);
int fit_rs_applicable(const Lay& l, int64_t pack);

}  // namespace v2vae
