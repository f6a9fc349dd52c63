/*
 * C ABI of libv2vae_hip.so: the V2VAE training hot path on MI355X (gfx950).
 *
 * Every entry point takes borrowed device pointers, plain sizes and a hipStream_t (passed as
 * void*; NULL = the default stream), enqueues work asynchronously on that stream and returns
 * 0 on success or a negative V2VAE_E* code.  Nothing here allocates, frees or synchronises;
 * the caller owns every buffer, including the workspace (size from v2vae_workspace_bytes).
 * Calls on different streams / devices with different workspaces are independent; the only process-wide state is the
 * calling thread's last-error text, the per-kernel LDS-size attribute the launchers set, and the per-device address of the
 * static scratch of v2vae_train_step_static.
 *
 * The reference has no FFI; the functions it exposes for this path are Python methods.  Each
 * entry point cites the reference code whose arithmetic it replaces (paths under the reference
 * checkout).
 *
 * Data model ("real-stacked"): ns real streams (ns = 1 real PAM, ns = 2 I/Q of complex QAM).
 *   y        [ns][y_len]    received samples, row c = stream c (complex: Re, Im;
 *                           lib/complex_equalization.py:677-681)
 *   xhat     [ns][sym_cnt]  equalised symbols
 *   levels, prior, scaling [ns][C]  amplitude levels / prior / learnable noise scaling per axis
 * Raw parameters, mode 0 (real, lib/real_equalization.py:753-764, lib/torch_components.py:69-76):
 *   h1 [L], h2 [L][L], k1 [l1], k2 [l2][l2]
 * mode 1 (complex, holomorphic second-order kernels; Re plane then Im plane):
 *   h1 [2][L], h2 [2][L][L], k1 [2][l1], k2 [2][l2][l2]
 * mode 2 (real, memory-polynomial channel, lib/real_equalization.py:1238-1246, ELBO :1304-1354):
 *   h1 [L], h2 [L] (per-tap weight of x^2), k1, k2 as mode 0
 * mode 3 (real, Hammerstein channel, lib/real_equalization.py:1053-1058, ELBO :1117-1167):
 *   h1 [L] = channel_h, h2 [2] = (alpha_1, alpha_2), k1, k2 as mode 0
 * Modes 2 and 3 are the second-order model with a diagonal kernel and run on the mode-0 kernels.
 * l2 = 0 selects a linear FIR equaliser (lib/real_equalization.py:971-982) and k2 may be NULL.
 */
#ifndef V2VAE_HIP_H
#define V2VAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define V2VAE_OK 0
#define V2VAE_EINVAL (-1)   /* bad size / NULL pointer / unsupported combination */
#define V2VAE_EHALO (-2)    /* local y / symbol range does not cover what the owned range needs */
#define V2VAE_ELAUNCH (-3)  /* HIP reported a launch error */
#define V2VAE_ESIZE (-4)    /* problem too large for the kernels' on-chip tiling */

#define V2VAE_FLAG_GENERIC 1     /* the generic FMA kernels for every shape (no matrix-core / on-chip kernels) */
#define V2VAE_FLAG_EQ_MFMA_FWD 2 /* complex, l2 <= 65: the matrix-core forward equaliser instead of the symmetric-form one */
#define V2VAE_FLAG_REAL_MFMA 8   /* real model (mode 0): the matrix-core ELBO kernel and equaliser backward at every batch size (by default
                                    only from 2^12 owned symbols on; smaller batches run the generic kernels) */
#define V2VAE_FLAG_ELBO_TILES 4  /* complex, <= 16 active taps per phase: the block-tile matrix-core ELBO kernel (workgroup barriers
                                    per tile) instead of the run kernel (every wave on its own run of windows); 17 or 33 active
                                    taps: whole blocks of 16 taps (2 or 3) instead of 1 or 2 blocks plus one tap on the side */

#define V2VAE_FLAG_F32_MATRIX 16 /* complex model, one or two samples per symbol: the kernels that by default run on bf16 matrix
                                    instructions in three-piece split arithmetic - equaliser backward (l2 <= 95), equaliser forward
                                    from 33 lags on, the H X and H w products of the run ELBO kernel (<= 16 active taps per phase) -
                                    on f32 matrix instructions / the vector pipe instead (backward then l2 <= 79 on 16-lag blocks) */

typedef struct v2vae_problem {
    int32_t ns;      /* 1 or 2 */
    int32_t mode;    /* 0 real (bug-compatible with lib/real_equalization.py:869), 1 complex,
                        2 real with memory-polynomial channel, 3 real with Hammerstein channel */
    int32_t sps;     /* samples per symbol S */
    int32_t l1;      /* first-order equaliser taps */
    int32_t pad1;    /* left zero padding of the first-order part (l1/2 for the Volterra equaliser,
                        (l1-1)/2 for the Conv1d equaliser) */
    int32_t l2;      /* second-order equaliser lags (odd) or 0 */
    int32_t L;       /* channel memory */
    int32_t C;       /* amplitude levels per stream (<= 16) */
    int32_t flags;   /* 0, or V2VAE_FLAG_* bits: A/B switches for measurements between two HIP kernels of this library that
                        cover the same shape.  Kernel dispatch and the workspace carve are functions of this struct alone. */
    int32_t reserved; /* must be 0 */
    int64_t n_glob;  /* samples of the whole batch; zero padding applies outside [0, n_glob) */
    int64_t y_lo;    /* global sample index of y[.][0] */
    int64_t y_len;   /* samples per stream held locally */
    int64_t sym_lo;  /* first symbol for which xhat / moments are held locally */
    int64_t sym_cnt; /* number of such symbols */
    int64_t own_lo;  /* first owned symbol: owned symbols carry the KL term, per-symbol and equaliser
                        gradients; windows n in [own_lo*sps, (own_lo+own_cnt)*sps) carry E and the
                        channel-model gradients */
    int64_t own_cnt;
} v2vae_problem;

/* Slots of the double-precision scalar block `scal` (8 doubles, device memory). */
#define V2VAE_SCAL_E 0      /* exponent term E (lib/real_equalization.py:918), local then global sum */
#define V2VAE_SCAL_KL 1     /* KL term (:847) */
#define V2VAE_SCAL_LOSS 2   /* KL + W log E (:919) */
#define V2VAE_SCAL_LAMBDA 3 /* W / E */
#define V2VAE_SCAL_W 4      /* number of windows W = n_glob - L + 1 */
#define V2VAE_SCAL_BAD_STEPS 5 /* failure detection (the reference has none on this path: a non-finite loss just propagates,
                                  lib/real_equalization.py:919): running count of steps whose E was not a positive finite number or
                                  whose KL term was not finite, since the caller last zeroed the block.  The library only ever adds
                                  to it (v2vae_train_step_static starts every call from a zeroed block: there it is 0 or 1) */
#define V2VAE_SCAL_COUNT 8

/* Bytes of workspace the calls below need for this problem (0 if the problem is invalid). */
size_t v2vae_workspace_bytes(const v2vae_problem* p);

/* Validate a problem description (sizes, halo coverage).  Returns a V2VAE_* code. */
int v2vae_check_problem(const v2vae_problem* p);

/* Raw parameters -> on-device coefficient tables in the workspace (symmetrised second-order
 * kernels, lib/torch_components.py:94 and lib/real_equalization.py:841; per-phase compact channel
 * taps).  Must precede the calls below whenever a parameter changed. */
int v2vae_prep_tables(const v2vae_problem* p, const float* h1, const float* h2, const float* k1,
                      const float* k2, void* ws, void* stream);

/* Equaliser forward: SecondOrderVolterraSeries.forward (lib/torch_components.py:79-96) or the
 * Conv1d equaliser (lib/real_equalization.py:980-982).  xhat [ns][sym_cnt].  Real model: the sums are formed in double and
 * the part of the result that does not fit the float is kept in the workspace; v2vae_demap_moments and v2vae_sym_bwd add it
 * back when they are handed exactly this xhat (compared bitwise per symbol), which takes the fp32 rounding of xhat, amplified
 * by 2 |xhat - a| / noise variance in the soft demapper, out of the equaliser and noise-scaling gradients. */
int v2vae_eq_fwd(const v2vae_problem* p, const float* y, void* ws, float* xhat, void* stream);

/* Soft demapping + per-symbol posterior moments + KL partial sums
 * (lib/real_equalization.py:816-823, :847, :850-865).  q_out [ns][C][sym_cnt] may be NULL.  Needs no coefficient table:
 * this alone (after v2vae_eq_fwd) is the inference path. */
int v2vae_demap_moments(const v2vae_problem* p, const float* xhat, const float* levels,
                        const float* prior, const float* scaling, const float* noise_var, void* ws,
                        float* q_out, void* stream);

/* The pieces of E and of dE/d(moments) that do not involve the residual y' - mean (terms of lib/real_equalization.py:887,
 * :904-914 whose coefficients are constants of the channel kernels): lagged products of the posterior variances and per-symbol
 * correlations.  Order: after v2vae_demap_moments AND v2vae_prep_tables(h1, h2), before v2vae_elbo_fused.  Launches nothing for
 * shapes served by the generic ELBO kernel, which forms these terms itself. */
int v2vae_elbo_prepass(const v2vae_problem* p, void* ws, void* stream);

/* Closed-form ELBO expectations over sliding windows, forward and (unscaled) backward in one pass
 * (lib/real_equalization.py:868-918): partial sums of E, dE/d(channel kernels), dE/d(moments).  Needs the moments of
 * v2vae_demap_moments, the channel tables and the output of v2vae_elbo_prepass. */
int v2vae_elbo_fused(const v2vae_problem* p, const float* y, void* ws, void* stream);

/* Deterministic reduction of the E / KL partial sums of this rank into scal[E], scal[KL]. */
int v2vae_reduce_scalars(const v2vae_problem* p, const void* ws, double* scal, void* stream);

/* scal[LOSS] = KL + W log E, scal[LAMBDA] = W / E (lib/real_equalization.py:919), from global sums. */
int v2vae_finalize_scalars(const v2vae_problem* p, double* scal, void* stream);

/* d loss / d xhat for owned symbols (0 elsewhere) and noise-scaling gradient partials: backward
 * of :816-865 given dE/d(moments) in the workspace.  grad_scale (device float, may be NULL = 1)
 * multiplies everything (upstream gradient of the loss). */
int v2vae_sym_bwd(const v2vae_problem* p, const float* xhat, const float* levels, const float* prior,
                  const float* scaling, const float* noise_var, const double* scal,
                  const float* grad_scale, void* ws, float* gxhat, void* stream);

/* Equaliser backward: gradient partial sums of the first- and second-order kernels given gxhat
 * (autograd of lib/torch_components.py:79-96). */
int v2vae_eq_bwd(const v2vae_problem* p, const float* y, const float* gxhat, void* ws, void* stream);

/* Reduce all partial sums and chain them to raw-parameter gradients.  Any output may be NULL.
 * which: bit 0 = channel (g_h1, g_h2: scaled by scal[LAMBDA] and grad_scale), bit 1 = noise scaling
 * (g_scaling [ns][C]), bit 2 = equaliser (g_k1, g_k2), bit 3 = ADD the noise-scaling gradient to g_scaling instead of
 * overwriting it (lib/real_equalization.py:1169-1178 and :1356-1364 never reset that gradient). */
int v2vae_finish_grads(const v2vae_problem* p, void* ws, const double* scal, const float* grad_scale,
                       int which, float* g_h1, float* g_h2, float* g_k1, float* g_k2,
                       float* g_scaling, void* stream);

/* Adam with AMSGrad on a flat parameter pack, torch.optim.Adam(amsgrad=True) arithmetic
 * (lib/real_equalization.py:457-459); `step` is the 1-based step count.  If noise_var is not NULL
 * it is set to E / W (:921-923); if loss_out is not NULL the loss is stored there as a float. */
int v2vae_adam_amsgrad(int64_t n, float* params, const float* grads, float* m, float* v, float* vmax,
                       float lr, float beta1, float beta2, float eps, int64_t step,
                       const double* scal, float* noise_var, float* loss_out, void* stream);

/* Hard decisions from posteriors: arg max over levels per stream (lib/utility.py:60-69).
 * dec [ns][sym_cnt] int32 level indices, computed from xhat directly. */
int v2vae_decide(const v2vae_problem* p, const float* xhat, const float* levels, const float* scaling,
                 const float* noise_var, int32_t* dec, void* stream);

/* Element counts of the five sections of the flat parameter pack [h1 | h2 | k1 | k2 | scaling]. */
void v2vae_pack_sizes(const v2vae_problem* p, int64_t out[5]);

typedef struct v2vae_train_args {
    const float* y;       /* [ns][y_len] */
    const float* levels;  /* [ns][C] */
    const float* prior;   /* [ns][C] */
    float* params;        /* flat pack, sections as in v2vae_pack_sizes */
    float* grads;         /* same layout, overwritten */
    float* adam_m;        /* Adam first moment, second moment, running max of the second moment */
    float* adam_v;
    float* adam_vmax;
    float* noise_var;     /* device scalar, read by the soft demapper, then set to E / W if adaptive_noise */
    void* ws;
    double* scal;         /* V2VAE_SCAL_COUNT doubles */
    float* xhat;          /* [ns][sym_cnt] out */
    float* gxhat;         /* [ns][sym_cnt] scratch */
    float* loss_out;      /* device float, may be NULL */
    float lr, beta1, beta2, eps;
    int64_t step;         /* 1-based Adam step count */
    int32_t learn_scaling;   /* 0: the scaling section is not updated (lib/real_equalization.py:1014-1037) */
    int32_t adaptive_noise;  /* lib/real_equalization.py:921-923 */
    int32_t freeze_h2;       /* 1: the second-order channel kernel is not updated; with h2 = 0 and unit scaling this is the
                                linear-channel VAE of lib/real_equalization.py:564-690 (same ELBO, :629-664) */
    int32_t accumulate_scaling_grad;  /* 1: the scaling section of `grads` is a running sum over steps (the memory-polynomial
                                and Hammerstein VAEs never reset that gradient: lib/real_equalization.py:1169-1178, :1356-1364) */
} v2vae_train_args;

/* One whole single-GPU training step of GenericTorchBlindEqualizer.fit's loop body
 * (lib/real_equalization.py:488-496): forward, ELBO, backward, Adam-AMSGrad update, all enqueued
 * on `stream` with no host synchronisation. */
int v2vae_train_step(const v2vae_problem* p, const v2vae_train_args* a, void* stream);

/* The halo-sharded step (one rank's part of one long batch; the reference has no distributed code, its fit loop is
 * lib/real_equalization.py:484-502) as three calls around the caller's two collectives:
 *   v2vae_train_step_sharded_pre    forward and ELBO partial sums; local E, KL in scal[0..1]
 *   -- all-reduce (sum) scal[0..1] over the ranks --
 *   v2vae_train_step_sharded_post   loss and W / E from the global sums, backward, this rank's gradient sums in a->grads
 *   -- all-reduce (sum) a->grads over the ranks --
 *   v2vae_train_step_update         Adam-AMSGrad from a->grads (sections per learn_scaling / freeze_h2), noise variance, loss
 * With one rank (no collectives) the three calls equal v2vae_train_step.  accumulate_scaling_grad is not supported by the
 * sharded pair (the running sum would be all-reduced again every step). */
int v2vae_train_step_sharded_pre(const v2vae_problem* p, const v2vae_train_args* a, void* stream);
int v2vae_train_step_sharded_post(const v2vae_problem* p, const v2vae_train_args* a, void* stream);
int v2vae_train_step_update(const v2vae_problem* p, const v2vae_train_args* a, void* stream);

/* The same step for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops, harnesses
 * that forbid hipMalloc).  No counterpart in the reference.  Workspace, scalar block and per-symbol gradient buffer come from
 * one static device array of V2VAE_STATIC_SCRATCH_BYTES inside the library (every GPU the library runs on holds its own copy
 * until the process ends); a->ws, a->scal and a->gxhat must be NULL, every other field is as for v2vae_train_step and the
 * results are bit-identical to it.  Calls no hipMalloc / hipFree or any other allocating routine.  V2VAE_ESIZE (and a
 * v2vae_last_error text with both sizes) if v2vae_static_scratch_bytes(p) exceeds the array.  All calls on one GPU share that
 * GPU's array: it is cleared at the start of every call, and two calls that overlap in time on one GPU (different streams or
 * threads) race - serialise them.  The stream must belong to the calling thread's current device (V2VAE_EINVAL otherwise).
 * The first call on a device resolves the array's address (hipGetSymbolAddress, which may load the code object): make it
 * outside a graph capture.  Adam's bias corrections and the learning rate are computed on the host from a->step and a->lr and
 * travel as kernel arguments, so a captured step replays with the step count and rate it was captured with. */
#define V2VAE_STATIC_SCRATCH_BYTES (32u << 20)
size_t v2vae_static_scratch_bytes(const v2vae_problem* p); /* bytes of the array this problem would use; 0 if invalid */
int v2vae_train_step_static(const v2vae_problem* p, const v2vae_train_args* a, void* stream);

typedef struct v2vae_fit_args {
    const float* y;           /* model m, stream c, sample i of batch n: y[m*y_model_stride + c*y_stream_stride + n*B*sps + i] */
    int64_t y_model_stride;   /* 0: every model trains on the same sequence */
    int64_t y_stream_stride;  /* >= n_batches * B * sps */
    const float* levels;      /* [ns][C], shared by all models */
    const float* prior;       /* [ns][C], shared */
    float* params;            /* [n_models][pack], pack = sum of v2vae_pack_sizes */
    float* grads;             /* [n_models][pack] scratch (holds the last batch's gradients on return) */
    float* adam_m;            /* [n_models][pack] each */
    float* adam_v;
    float* adam_vmax;
    float* noise_var;         /* [n_models] */
    void* ws;                 /* n_models * v2vae_workspace_bytes(p), zero-initialised once */
    double* scal;             /* [n_models][V2VAE_SCAL_COUNT] */
    float* xhat;              /* [n_models][ns][B] scratch */
    float* gxhat;             /* [n_models][ns][B] scratch */
    float* y_eq;              /* [n_models][ns][n_batches*B] equaliser output of every batch, before its update; may be NULL */
    float* loss_curve;        /* [n_models][n_batches]; may be NULL */
    const float* step_size;   /* [n_batches] device array from v2vae_adam_schedule */
    const float* bc2_sqrt;    /* [n_batches] device array from v2vae_adam_schedule */
    float beta1, beta2, eps;
    int32_t n_models;
    int32_t n_batches;
    int32_t learn_scaling;
    int32_t adaptive_noise;
    int32_t freeze_h2;        /* as in v2vae_train_args */
    int32_t accumulate_scaling_grad;  /* as in v2vae_train_args; the scaling section of `grads` is read on entry */
} v2vae_fit_args;

/* Host-side helper (no GPU work): per-step Adam constants of v2vae_adam_amsgrad for steps
 * step0+1 .. step0+n with learning rates lr[0..n) (the StepLR schedule of
 * lib/real_equalization.py:476-480,499 evaluated by the caller): step_size[i] = lr[i] / (1 - beta1^step),
 * bc2_sqrt[i] = sqrt(1 - beta2^step).  Both outputs are HOST arrays the caller copies to the device. */
int v2vae_adam_schedule(int64_t step0, int64_t n, const float* lr, float beta1, float beta2, float* step_size,
                        float* bc2_sqrt);

/* The whole mini-batch loop of GenericTorchBlindEqualizer.fit (lib/real_equalization.py:482-502) for n_models
 * independent models in ONE launch: one workgroup per model runs n_batches sequential training steps, so the
 * launch-bound small-batch configuration costs one launch per fit instead of ~17 per batch, and a sweep of
 * models (channels / SNR points / seeds) fills the GPU.  `p` describes ONE batch as a whole sequence
 * (n_glob = y_len = B*sps, y_lo = sym_lo = own_lo = 0, sym_cnt = own_cnt = B).
 * Two kernels (v2vae_fit_persistent_kind tells which): the on-chip kernel keeps batch, moments, tables, parameters
 * and optimiser state in LDS / registers (same mathematics as v2vae_train_step, its own summation order: results
 * agree to fp32 rounding); the stage-by-stage kernel runs the generic stage code of v2vae_train_step in the same
 * order (bit-identical results) out of the workspace.  xhat, gxhat and ws are only touched by the latter.
 * V2VAE_ESIZE if neither covers the shape (then use v2vae_train_step). */
int v2vae_fit_persistent(const v2vae_problem* p, const v2vae_fit_args* a, void* stream);

/* Which kernel v2vae_fit_persistent would run for this one-batch problem: 2 = the on-chip kernel (real model,
 * <= 8 active channel taps per phase, both equaliser kernels inside one 32-sample window, batch and parameters
 * fit LDS/registers): fastest route for small batches; 1 = the stage-by-stage persistent kernel (meant for sweeps of
 * many models); negative = status code (shape not supported by either). */
int v2vae_fit_persistent_kind(const v2vae_problem* p);

/* Device-side synthetic channel (no counterpart function in the reference: its simulator, lib/data_generation.py:206-241 and
 * :349-372, runs on the host with numpy; this produces the same kind of sequence directly in HBM).  Chain: uniform symbols
 * (levels per axis) -> zero-stuffing to sps samples -> pulse -> fir1 -> w = poly1 z + poly2 z^2 (complex arithmetic for ns = 2)
 * -> fir2 -> (. - mean) * inv_std -> + white Gaussian noise of standard deviation noise_std per stream.  Every stage is a
 * 'same' convolution with zeros outside the global sequence of b_glob symbols.  All random numbers are functions of
 * (seed, global index), so overlapping ranges produced by different ranks are identical. */
typedef struct v2vae_synth_args {
    int32_t ns;        /* 1 real PAM, 2 complex square QAM (independent I / Q levels) */
    int32_t sps;       /* samples per symbol, 1..8 */
    int32_t C;         /* levels per axis, <= 16 */
    int32_t n_pulse;   /* odd, <= 65 */
    int32_t n_fir1;    /* odd, <= 9 */
    int32_t n_fir2;    /* odd, <= 9 */
    float levels[16];
    float pulse[65];
    float fir1[9];
    float fir2[9];
    float poly1, poly2;
    float mean[2];     /* subtracted per stream before scaling */
    float inv_std;
    float noise_std;   /* per stream; 0 = no noise */
    uint64_t seed;
    int64_t b_glob;    /* symbols of the whole sequence */
    int64_t sym_lo;    /* first symbol produced */
    int64_t sym_cnt;   /* symbols produced: y [ns][sym_cnt * sps], sym_idx [ns][sym_cnt] level indices (may be NULL) */
} v2vae_synth_args;

int v2vae_synth_channel(const v2vae_synth_args* a, float* y, int32_t* sym_idx, void* stream);

/* The 64 random bits behind symbol (kind 0: high / low 32 bits pick the I / Q level) or noise sample (kind 1) `idx`;
 * host function, for checkers. */
uint64_t v2vae_synth_bits(uint64_t seed, uint64_t kind, uint64_t idx);

/* Read-only views into the workspace for tests and tooling: float offsets (in floats from the
 * workspace base) of the per-symbol moments [ns][4][sym_cnt] and of dE/d(moments), same shape.
 * Complex model on the matrix-core ELBO kernels: the dE/d(moments) planes are not (<= 16 active taps per phase) or only
 * partly (longer channels: not plane 3 away from the two ends of the sequence) filled; those kernels hand the per-symbol
 * backward their gradients in a layout of their own. */
int64_t v2vae_ws_offset_moments(const v2vae_problem* p);
int64_t v2vae_ws_offset_gmoments(const v2vae_problem* p);

/* Real models (one stream): float offsets of the two workspace planes [sym_cnt] in which v2vae_eq_fwd keeps its output as a float
 * (a copy of xhat) and the float remainder of the double-precision sum behind it; -1 when the layout has none.  A caller that
 * runs the equaliser on one workspace and the soft demapper / per-symbol backward on another (the reference's module structure:
 * SecondOrderVolterraSeries.forward, lib/torch_components.py:79-96, is a module of its own; _calculate_loss,
 * lib/real_equalization.py:836-925, belongs to the estimator) copies the two planes across so that the remainder takes effect
 * there as well. */
int64_t v2vae_ws_offset_xhat_float(const v2vae_problem* p);
int64_t v2vae_ws_offset_xhat_remainder(const v2vae_problem* p);

const char* v2vae_version(void);

/* Text of the last HIP error seen by the calling thread inside this library (debugging aid). */
const char* v2vae_last_error(void);

#ifdef __cplusplus
}
#endif
#endif
