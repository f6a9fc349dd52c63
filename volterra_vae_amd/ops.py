"""Thin torch layer over the C ABI: tensors in, kernels enqueued on torch's current HIP stream.

`Plan` binds one problem geometry (sizes + which slice of the batch this rank owns) to its
workspace; the autograd Functions below make the equaliser and the ELBO differentiable torch
ops whose forward and backward are the hand-written kernels.
"""
from __future__ import annotations

import ctypes as C
import itertools
import weakref
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import FitArgs, Problem, TrainArgs, check, lib


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _require_gpu_f32(t: torch.Tensor, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU: the V2VAE hot path has no CPU implementation")
    if t.dtype != torch.float32:
        raise NotImplementedError(f"{name}: the HIP kernels compute in float32 (got {t.dtype})")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")


class Plan:
    """Geometry + workspace for one (batch length, model shape, owned slice) combination."""

    def __init__(self, *, ns, sps, l1, pad1, l2, L, C_levels, n_glob, device,
                 y_lo=0, y_len=None, sym_lo=0, sym_cnt=None, own_lo=None, own_cnt=None, mode=None, flags=None):
        b_glob = n_glob // sps
        p = Problem()
        p.flags = _lib.default_flags() if flags is None else flags
        mode = (0 if ns == 1 else 1) if mode is None else mode  # 2 / 3: memory-polynomial / Hammerstein channel (real)
        p.ns, p.mode, p.sps, p.l1, p.pad1, p.l2, p.L, p.C = ns, mode, sps, l1, pad1, l2, L, C_levels
        p.n_glob, p.y_lo = n_glob, y_lo
        p.y_len = n_glob if y_len is None else y_len
        p.sym_lo = sym_lo
        p.sym_cnt = b_glob if sym_cnt is None else sym_cnt
        p.own_lo = p.sym_lo if own_lo is None else own_lo
        p.own_cnt = p.sym_cnt if own_cnt is None else own_cnt
        check(lib.v2vae_check_problem(C.byref(p)), "v2vae_check_problem")
        self.p = p
        self.device = device
        nbytes = lib.v2vae_workspace_bytes(C.byref(p))
        self.ws = torch.zeros((nbytes + 3) // 4, dtype=torch.float32, device=device)
        self.scal = torch.zeros(_lib.SCAL_COUNT, dtype=torch.float64, device=device)
        self.gxhat = torch.zeros((ns, p.sym_cnt), dtype=torch.float32, device=device)
        self.sizes = _lib.pack_sizes(p)
        self.W = n_glob - L + 1
        # Counts the calls that overwrite what the ELBO's backward reads from the workspace (channel tables, moments,
        # dE/d(moments), partial records, the scalar block).  ElboLossFn notes it at forward and re-runs its forward stages
        # in backward if anything ran on this plan in between.
        self.elbo_generation = 0
        self.handle = _register_plan(self)   # what the registered torch ops (torch.ops.v2vae.*) take in place of the object

    # --- individual stages ---------------------------------------------------------------
    def prep_tables(self, h1=None, h2=None, k1=None, k2=None):
        if h1 is not None:
            self.elbo_generation += 1
        check(lib.v2vae_prep_tables(C.byref(self.p), _ptr(h1), _ptr(h2), _ptr(k1), _ptr(k2), _ptr(self.ws), _stream()),
              "v2vae_prep_tables")

    def eq_fwd(self, y, xhat):
        check(lib.v2vae_eq_fwd(C.byref(self.p), _ptr(y), _ptr(self.ws), _ptr(xhat), _stream()), "v2vae_eq_fwd")

    def demap_moments(self, xhat, levels, prior, scaling, noise_var, q_out=None):
        self.elbo_generation += 1
        check(lib.v2vae_demap_moments(C.byref(self.p), _ptr(xhat), _ptr(levels), _ptr(prior), _ptr(scaling), _ptr(noise_var),
                                      _ptr(self.ws), _ptr(q_out), _stream()), "v2vae_demap_moments")

    def elbo_prepass(self):
        self.elbo_generation += 1
        check(lib.v2vae_elbo_prepass(C.byref(self.p), _ptr(self.ws), _stream()), "v2vae_elbo_prepass")

    def elbo_fused(self, y, prepass=True):
        """ELBO forward + unscaled backward; `prepass=False` if elbo_prepass() was already called for these moments."""
        if prepass:
            self.elbo_prepass()
        self.elbo_generation += 1
        check(lib.v2vae_elbo_fused(C.byref(self.p), _ptr(y), _ptr(self.ws), _stream()), "v2vae_elbo_fused")

    def reduce_scalars(self):
        self.elbo_generation += 1
        check(lib.v2vae_reduce_scalars(C.byref(self.p), _ptr(self.ws), _ptr(self.scal), _stream()), "v2vae_reduce_scalars")

    def finalize_scalars(self):
        self.elbo_generation += 1
        check(lib.v2vae_finalize_scalars(C.byref(self.p), _ptr(self.scal), _stream()), "v2vae_finalize_scalars")

    def sym_bwd(self, xhat, levels, prior, scaling, noise_var, grad_scale=None, gxhat=None):
        gx = self.gxhat if gxhat is None else gxhat
        check(lib.v2vae_sym_bwd(C.byref(self.p), _ptr(xhat), _ptr(levels), _ptr(prior), _ptr(scaling), _ptr(noise_var),
                                _ptr(self.scal), _ptr(grad_scale), _ptr(self.ws), _ptr(gx), _stream()), "v2vae_sym_bwd")
        return gx

    def eq_bwd(self, y, gxhat):
        check(lib.v2vae_eq_bwd(C.byref(self.p), _ptr(y), _ptr(gxhat), _ptr(self.ws), _stream()), "v2vae_eq_bwd")

    def finish_grads(self, which, g_h1=None, g_h2=None, g_k1=None, g_k2=None, g_scaling=None, grad_scale=None):
        check(lib.v2vae_finish_grads(C.byref(self.p), _ptr(self.ws), _ptr(self.scal), _ptr(grad_scale), which,
                                     _ptr(g_h1), _ptr(g_h2), _ptr(g_k1), _ptr(g_k2), _ptr(g_scaling), _stream()),
              "v2vae_finish_grads")

    def decide(self, xhat, levels, scaling, noise_var):
        dec = torch.empty((self.p.ns, self.p.sym_cnt), dtype=torch.int32, device=self.device)
        check(lib.v2vae_decide(C.byref(self.p), _ptr(xhat), _ptr(levels), _ptr(scaling), _ptr(noise_var), _ptr(dec), _stream()),
              "v2vae_decide")
        return dec

    def moments_view(self):
        off = lib.v2vae_ws_offset_moments(C.byref(self.p))
        n = self.p.ns * 4 * self.p.sym_cnt
        return self.ws[off:off + n].view(self.p.ns, 4, self.p.sym_cnt)

    def xhat_remainder(self):
        """Real model: the float remainder [sym_cnt] of the last eq_fwd on this plan (a copy), else None.  Together with the
        xhat that eq_fwd returned it is the equaliser output to double precision."""
        off = lib.v2vae_ws_offset_xhat_remainder(C.byref(self.p))
        return None if off < 0 else self.ws[off:off + self.p.sym_cnt].clone()

    def set_xhat_remainder(self, xhat, xlo):
        """Hands this plan an equaliser output that another plan computed: xhat [1, sym_cnt] (float) and its remainder xlo
        [sym_cnt] go into the planes v2vae_demap_moments and v2vae_sym_bwd read."""
        hi, lo = lib.v2vae_ws_offset_xhat_float(C.byref(self.p)), lib.v2vae_ws_offset_xhat_remainder(C.byref(self.p))
        if hi < 0:
            return
        n = self.p.sym_cnt
        self.ws[hi:hi + n].copy_(xhat.reshape(-1))
        self.ws[lo:lo + n].copy_(xlo.reshape(-1))

    def bad_steps(self) -> int:
        """Failure detection: how many steps on this plan had an exponent term E that was not a positive finite number or a
        non-finite KL term since the plan was created (synchronises; the reference has no such check)."""
        return int(self.scal[_lib.SCAL_BAD_STEPS].item())

    def gmoments_view(self):
        off = lib.v2vae_ws_offset_gmoments(C.byref(self.p))
        n = self.p.ns * 4 * self.p.sym_cnt
        return self.ws[off:off + n].view(self.p.ns, 4, self.p.sym_cnt)

    # --- whole fused step ------------------------------------------------------------------
    def train_args(self, *, y, levels, prior, params, grads, adam_m, adam_v, adam_vmax, noise_var, xhat, loss_out,
                   lr, step, beta1=0.9, beta2=0.999, eps=1e-8, learn_scaling=True, adaptive_noise=True, freeze_h2=False,
                   accumulate_scaling_grad=False) -> TrainArgs:
        """v2vae_train_args for this plan's workspace (the tensors must outlive the enqueued work)."""
        a = TrainArgs()
        a.y, a.levels, a.prior = y.data_ptr(), levels.data_ptr(), prior.data_ptr()
        a.params, a.grads = params.data_ptr(), grads.data_ptr()
        a.adam_m, a.adam_v, a.adam_vmax = adam_m.data_ptr(), adam_v.data_ptr(), adam_vmax.data_ptr()
        a.noise_var, a.ws, a.scal = noise_var.data_ptr(), self.ws.data_ptr(), self.scal.data_ptr()
        a.xhat, a.gxhat = xhat.data_ptr(), self.gxhat.data_ptr()
        a.loss_out = 0 if loss_out is None else loss_out.data_ptr()
        a.lr, a.beta1, a.beta2, a.eps, a.step = lr, beta1, beta2, eps, step
        a.learn_scaling, a.adaptive_noise, a.freeze_h2 = int(learn_scaling), int(adaptive_noise), int(freeze_h2)
        a.accumulate_scaling_grad = int(accumulate_scaling_grad)
        return a

    def train_step(self, *, static_scratch=False, **kw):
        """One whole single-GPU training step (v2vae_train_step); keyword arguments as for train_args."""
        self.elbo_generation += 1
        a = self.train_args(**kw)
        if static_scratch:  # workspace, scalars and gxhat from the library's static device array: nothing allocated by anyone
            a.ws = a.scal = a.gxhat = None
            check(lib.v2vae_train_step_static(C.byref(self.p), C.byref(a), _stream()), "v2vae_train_step_static")
            return
        check(lib.v2vae_train_step(C.byref(self.p), C.byref(a), _stream()), "v2vae_train_step")

    # the halo-sharded step: three calls around the caller's two all-reduces (scal[:2], then the gradient pack)
    def sharded_pre(self, a: TrainArgs):
        self.elbo_generation += 1
        check(lib.v2vae_train_step_sharded_pre(C.byref(self.p), C.byref(a), _stream()), "v2vae_train_step_sharded_pre")

    def sharded_post(self, a: TrainArgs):
        self.elbo_generation += 1
        check(lib.v2vae_train_step_sharded_post(C.byref(self.p), C.byref(a), _stream()), "v2vae_train_step_sharded_post")

    def update(self, a: TrainArgs):
        check(lib.v2vae_train_step_update(C.byref(self.p), C.byref(a), _stream()), "v2vae_train_step_update")


def adam_amsgrad_(params, grads, m, v, vmax, lr, step, beta1=0.9, beta2=0.999, eps=1e-8, scal=None, noise_var=None,
                  loss_out=None, n=None):
    n = params.numel() if n is None else n
    check(lib.v2vae_adam_amsgrad(n, _ptr(params), _ptr(grads), _ptr(m), _ptr(v), _ptr(vmax), lr, beta1, beta2, eps, step,
                                 _ptr(scal), _ptr(noise_var), _ptr(loss_out), _stream()), "v2vae_adam_amsgrad")


def adam_schedule(lrs, step0=0, beta1=0.9, beta2=0.999):
    """Per-step (step_size, bc2_sqrt) float32 host arrays for steps step0+1.. with learning rates `lrs`."""
    import numpy as np
    lr = np.ascontiguousarray(lrs, dtype=np.float32)
    ss, bc = np.empty_like(lr), np.empty_like(lr)
    check(lib.v2vae_adam_schedule(step0, lr.size, lr.ctypes.data, beta1, beta2, ss.ctypes.data, bc.ctypes.data),
          "v2vae_adam_schedule")
    return ss, bc


class PersistentFit:
    """n_models independent models, each trained over n_batches sequential mini-batches in ONE kernel launch
    (one workgroup per model): the mini-batch loop of lib/real_equalization.py:482-502 without a launch or a
    host round trip per batch.  All state tensors carry a leading model dimension."""

    def __init__(self, *, ns, sps, l1, pad1, l2, L, C_levels, batch_symbols, n_models, device, mode=None, flags=None):
        p = Problem()
        p.flags = _lib.default_flags() if flags is None else flags
        mode = (0 if ns == 1 else 1) if mode is None else mode
        p.ns, p.mode, p.sps, p.l1, p.pad1, p.l2, p.L, p.C = ns, mode, sps, l1, pad1, l2, L, C_levels
        p.n_glob = p.y_len = batch_symbols * sps
        p.y_lo = p.sym_lo = p.own_lo = 0
        p.sym_cnt = p.own_cnt = batch_symbols
        kind = lib.v2vae_fit_persistent_kind(C.byref(p))
        if kind < 0:
            check(kind, "v2vae_fit_persistent_kind")
        self.on_chip = kind == 2  # the whole step stays in LDS/registers (small real shapes)
        self.p, self.device, self.n_models, self.B = p, device, n_models, batch_symbols
        self.sizes = _lib.pack_sizes(p)
        self.pack = sum(self.sizes)
        nfl = (lib.v2vae_workspace_bytes(C.byref(p)) + 3) // 4
        self.ws = torch.zeros((n_models, nfl), dtype=torch.float32, device=device)
        self.scal = torch.zeros((n_models, _lib.SCAL_COUNT), dtype=torch.float64, device=device)
        self.xhat = torch.zeros((n_models, ns, batch_symbols), dtype=torch.float32, device=device)
        self.gxhat = torch.zeros_like(self.xhat)

    def run(self, *, y, levels, prior, params, grads, adam_m, adam_v, adam_vmax, noise_var, lrs, step0=0,
            y_eq=None, loss_curve=None, beta1=0.9, beta2=0.999, eps=1e-8, learn_scaling=True, adaptive_noise=True,
            freeze_h2=False, accumulate_scaling_grad=False):
        """y: [ns, T] (shared by all models) or [n_models, ns, T], T >= n_batches * B * sps; lrs: one learning
        rate per batch (its length is the number of batches); params & co: [n_models, pack]."""
        nb = len(lrs)
        p = self.p
        for name, t in (("y", y), ("params", params), ("grads", grads), ("adam_m", adam_m), ("adam_v", adam_v),
                        ("adam_vmax", adam_vmax), ("noise_var", noise_var), ("levels", levels), ("prior", prior)):
            _require_gpu_f32(t, name)
        if y.dim() == 3 and y.shape[0] != self.n_models:
            raise ValueError("y: leading dimension must be n_models")
        if y.shape[-2] != p.ns or y.shape[-1] < nb * self.B * p.sps:
            raise ValueError(f"y: expected [..., {p.ns}, >= {nb * self.B * p.sps}], got {tuple(y.shape)}")
        for name, t in (("params", params), ("grads", grads), ("adam_m", adam_m), ("adam_v", adam_v), ("adam_vmax", adam_vmax)):
            if tuple(t.shape) != (self.n_models, self.pack):
                raise ValueError(f"{name}: expected {(self.n_models, self.pack)}, got {tuple(t.shape)}")
        if noise_var.numel() != self.n_models:
            raise ValueError("noise_var: one value per model")
        if y_eq is not None and tuple(y_eq.shape) != (self.n_models, p.ns, nb * self.B):
            raise ValueError(f"y_eq: expected {(self.n_models, p.ns, nb * self.B)}")
        if loss_curve is not None and tuple(loss_curve.shape) != (self.n_models, nb):
            raise ValueError(f"loss_curve: expected {(self.n_models, nb)}")
        ss, bc = adam_schedule(lrs, step0, beta1, beta2)
        sched = torch.from_numpy(__import__("numpy").stack((ss, bc))).to(self.device)
        a = FitArgs()
        a.y = y.data_ptr()
        a.y_model_stride = y.shape[-2] * y.shape[-1] if y.dim() == 3 else 0
        a.y_stream_stride = y.shape[-1]
        a.levels, a.prior = levels.data_ptr(), prior.data_ptr()
        a.params, a.grads = params.data_ptr(), grads.data_ptr()
        a.adam_m, a.adam_v, a.adam_vmax = adam_m.data_ptr(), adam_v.data_ptr(), adam_vmax.data_ptr()
        a.noise_var, a.ws, a.scal = noise_var.data_ptr(), self.ws.data_ptr(), self.scal.data_ptr()
        a.xhat, a.gxhat = self.xhat.data_ptr(), self.gxhat.data_ptr()
        a.y_eq = 0 if y_eq is None else y_eq.data_ptr()
        a.loss_curve = 0 if loss_curve is None else loss_curve.data_ptr()
        a.step_size, a.bc2_sqrt = sched[0].data_ptr(), sched[1].data_ptr()
        a.beta1, a.beta2, a.eps = beta1, beta2, eps
        a.n_models, a.n_batches = self.n_models, nb
        a.learn_scaling, a.adaptive_noise, a.freeze_h2 = int(learn_scaling), int(adaptive_noise), int(freeze_h2)
        a.accumulate_scaling_grad = int(accumulate_scaling_grad)
        check(lib.v2vae_fit_persistent(C.byref(p), C.byref(a), _stream()), "v2vae_fit_persistent")
        self._keep = sched  # the launch is asynchronous: keep the schedule alive until the next call


# ------------------------------------------------------------------------------------------------
# registered torch custom ops (torch.ops.v2vae.*) over the C ABI, with autograd formulas
# ------------------------------------------------------------------------------------------------
# A Plan (geometry + workspace) is not a tensor, so the ops take its integer handle; `plan_from_handle` resolves it.
# The ops are eager-only and stateful: forward and backward communicate through the plan's workspace (moments, dE/d(moments),
# partial records, the scalar block), which no argument declares, so they must not be traced into a compiled graph that
# reorders or deduplicates them; the generation counter of the plan (see _elbo_setup) covers every interleaving eager autograd
# allows.
_PLAN_COUNTER = itertools.count(1)
_PLANS = weakref.WeakValueDictionary()


def plan_from_handle(handle: int) -> Plan:
    try:
        return _PLANS[handle]
    except KeyError:
        raise RuntimeError(f"v2vae: plan handle {handle} is not alive (the Plan object was released)") from None


def _register_plan(plan: Plan) -> int:
    h = next(_PLAN_COUNTER)
    _PLANS[h] = plan
    return h


@torch.library.custom_op("v2vae::equaliser", mutates_args=(), device_types="cuda")
def _op_equaliser(y: torch.Tensor, k1: torch.Tensor, k2: Optional[torch.Tensor], plan: int) -> torch.Tensor:
    """xhat [ns, sym_cnt] = equaliser(y [ns, N]; k1, k2) through v2vae_prep_tables + v2vae_eq_fwd."""
    pl = plan_from_handle(plan)
    _require_gpu_f32(y, "y")
    pl.prep_tables(k1=k1.detach().contiguous(), k2=None if k2 is None else k2.detach().contiguous())
    xhat = torch.empty((pl.p.ns, pl.p.sym_cnt), dtype=torch.float32, device=y.device)
    pl.eq_fwd(y, xhat)
    return xhat


@_op_equaliser.register_fake
def _(y, k1, k2, plan):
    pl = plan_from_handle(plan)
    return y.new_empty((pl.p.ns, pl.p.sym_cnt), dtype=torch.float32)


@torch.library.custom_op("v2vae::equaliser_backward", mutates_args=(), device_types="cuda")
def _op_equaliser_backward(y: torch.Tensor, gx: torch.Tensor, plan: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """(g_k1, g_k2) flat, through v2vae_eq_bwd + v2vae_finish_grads; g_k2 is empty for a first-order equaliser."""
    pl = plan_from_handle(plan)
    n = pl.sizes
    pl.eq_bwd(y, gx.contiguous())
    g_k1 = torch.empty(n[2], dtype=torch.float32, device=y.device)
    g_k2 = torch.empty(n[3], dtype=torch.float32, device=y.device)
    pl.finish_grads(4, g_k1=g_k1, g_k2=g_k2 if n[3] else None)
    return g_k1, g_k2


@_op_equaliser_backward.register_fake
def _(y, gx, plan):
    n = plan_from_handle(plan).sizes
    return y.new_empty(n[2], dtype=torch.float32), y.new_empty(n[3], dtype=torch.float32)


def _equaliser_setup(ctx, inputs, output):
    y, k1, k2, plan = inputs
    ctx.save_for_backward(y)
    ctx.plan = plan
    ctx.plan_object = plan_from_handle(plan)   # the graph keeps the plan (its workspace) alive until the backward has run
    ctx.k_shapes = (k1.shape, None if k2 is None else k2.shape)


def _equaliser_backward(ctx, gx):
    (y,) = ctx.saved_tensors
    g_k1, g_k2 = torch.ops.v2vae.equaliser_backward(y, gx, ctx.plan)
    s1, s2 = ctx.k_shapes
    return None, g_k1.view(s1), None if s2 is None else g_k2.view(s2), None


_op_equaliser.register_autograd(_equaliser_backward, setup_context=_equaliser_setup)


def _elbo_stages(pl: Plan, xh, y, snap, levels, prior, xlo=None):
    n = pl.sizes
    h1c, h2c, scc, nv = snap[:n[0]], snap[n[0]:n[0] + n[1]], snap[n[0] + n[1]:n[0] + n[1] + n[4]], snap[-1:]
    pl.prep_tables(h1=h1c, h2=h2c)
    if xlo is not None and xlo.numel() == xh.numel():
        pl.set_xhat_remainder(xh, xlo)
    pl.demap_moments(xh, levels, prior, scc, nv)
    pl.elbo_fused(y)
    pl.reduce_scalars()
    pl.finalize_scalars()
    return scc, nv


@torch.library.custom_op("v2vae::elbo_loss", mutates_args=(), device_types="cuda")
def _op_elbo_loss(xhat: torch.Tensor, y: torch.Tensor, h1: torch.Tensor, h2: torch.Tensor, scaling: torch.Tensor,
                  noise_var: torch.Tensor, levels: torch.Tensor, prior: torch.Tensor,
                  plan: int, xlo: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """(loss = KL + W log E, E, snapshot) through the fused ELBO kernels.  `snapshot` = [h1 | h2 | scaling | noise variance] as
    they were at this call: the backward op works from it, so later in-place updates of the parameters or of the noise
    variance (lib/real_equalization.py:921-923) do not reach an earlier loss's gradients.  `xlo` (real model, optional): the
    float remainder of xhat from the equaliser's plan (Plan.xhat_remainder)."""
    pl = plan_from_handle(plan)
    _require_gpu_f32(y, "y")
    snap = torch.cat([h1.detach().reshape(-1), h2.detach().reshape(-1), scaling.detach().reshape(-1),
                      noise_var.detach().reshape(-1)[:1]])
    _elbo_stages(pl, xhat.detach().contiguous(), y, snap, levels, prior, xlo)
    sc = pl.scal.clone()
    return sc[_lib.SCAL_LOSS].to(torch.float32), sc[_lib.SCAL_E].to(torch.float32), snap


@_op_elbo_loss.register_fake
def _(xhat, y, h1, h2, scaling, noise_var, levels, prior, plan, xlo=None):
    return (xhat.new_empty((), dtype=torch.float32), xhat.new_empty((), dtype=torch.float32),
            xhat.new_empty(h1.numel() + h2.numel() + scaling.numel() + 1, dtype=torch.float32))


@torch.library.custom_op("v2vae::elbo_loss_backward", mutates_args=(), device_types="cuda")
def _op_elbo_loss_backward(gloss: torch.Tensor, xhat: torch.Tensor, y: torch.Tensor, snap: torch.Tensor, levels: torch.Tensor,
                           prior: torch.Tensor, plan: int, restage: bool,
                           xlo: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """(d loss / d xhat, g_h1, g_h2, g_scaling) flat.  The backward reads dE/d(moments), the partial records and W / E from the
    plan's workspace; `restage` re-runs the forward stages first (needed if anything else ran on the plan since the forward)."""
    pl = plan_from_handle(plan)
    n = pl.sizes
    xh = xhat.detach().contiguous()
    if restage:
        scc, nv = _elbo_stages(pl, xh, y, snap, levels, prior, xlo)
    else:
        scc, nv = snap[n[0] + n[1]:n[0] + n[1] + n[4]], snap[-1:]
    gs = gloss.detach().to(torch.float32).reshape(1).contiguous()
    gx = torch.empty_like(xh)
    pl.sym_bwd(xh, levels, prior, scc, nv, grad_scale=gs, gxhat=gx)
    g_h1 = torch.empty(n[0], dtype=torch.float32, device=xh.device)
    g_h2 = torch.empty(n[1], dtype=torch.float32, device=xh.device)
    g_sc = torch.empty(n[4], dtype=torch.float32, device=xh.device)
    pl.finish_grads(3, g_h1=g_h1, g_h2=g_h2, g_scaling=g_sc, grad_scale=gs)
    return gx, g_h1, g_h2, g_sc


@_op_elbo_loss_backward.register_fake
def _(gloss, xhat, y, snap, levels, prior, plan, restage, xlo=None):
    n = plan_from_handle(plan).sizes
    f = lambda k: xhat.new_empty(k, dtype=torch.float32)
    return torch.empty_like(xhat), f(n[0]), f(n[1]), f(n[4])


def _elbo_setup(ctx, inputs, output):
    xhat, y, h1, h2, scaling, noise_var, levels, prior, plan, xlo = inputs
    ctx.save_for_backward(xhat.detach(), y, output[2], levels, prior, xlo)
    ctx.plan = plan
    ctx.plan_object = plan_from_handle(plan)   # the graph keeps the plan (its workspace) alive until the backward has run
    ctx.shapes = (h1.shape, h2.shape, scaling.shape, xhat.shape)
    ctx.generation = [plan_from_handle(plan).elbo_generation]   # mutable: a re-staging backward brings it up to date


def _elbo_backward(ctx, gloss, _gE, _gsnap):
    xh, y, snap, levels, prior, xlo = ctx.saved_tensors
    pl = plan_from_handle(ctx.plan)
    restage = pl.elbo_generation != ctx.generation[0]
    gx, g_h1, g_h2, g_sc = torch.ops.v2vae.elbo_loss_backward(gloss, xh, y, snap, levels, prior, ctx.plan, restage, xlo)
    if restage:
        ctx.generation[0] = pl.elbo_generation
    s1, s2, s3, sx = ctx.shapes
    return gx.view(sx), None, g_h1.view(s1), g_h2.view(s2), g_sc.view(s3), None, None, None, None, None


_op_elbo_loss.register_autograd(_elbo_backward, setup_context=_elbo_setup)


def equaliser(plan: Plan, y, k1, k2):
    """xhat = equaliser(y; k1, k2) as the registered op torch.ops.v2vae.equaliser (differentiable in k1, k2)."""
    return torch.ops.v2vae.equaliser(y, k1, k2, plan.handle)


def elbo_loss(plan: Plan, xhat, y, h1, h2, scaling, noise_var, levels, prior, xlo=None):
    """(loss, E) as the registered op torch.ops.v2vae.elbo_loss (differentiable in xhat, h1, h2, scaling)."""
    loss, E, _ = torch.ops.v2vae.elbo_loss(xhat, y, h1, h2, scaling, noise_var, levels, prior, plan.handle, xlo)
    return loss, E.detach()


class EqualiserFn:
    """Kept for callers of the earlier autograd.Function interface: EqualiserFn.apply(plan, y, k1, k2)."""
    apply = staticmethod(equaliser)


class ElboLossFn:
    """Kept for callers of the earlier autograd.Function interface: ElboLossFn.apply(plan, xhat, y, h1, h2, ...)."""
    apply = staticmethod(elbo_loss)
