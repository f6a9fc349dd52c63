"""Batched sweep: many independent V2VAE models (channels / SNR points / seeds) trained concurrently.

The reference runs such sweeps as a Python loop over `SecondVolterraV2VAE(...).fit(rx)` calls
(main_v2vae.py:75-92 once per point).  Here all models of one shape train in ONE kernel launch, one workgroup per
model (ops.PersistentFit); with several GPUs each rank takes a contiguous slice of the models and there is no
collective at all.
"""
from __future__ import annotations

import numpy as np
import torch

from .ops import PersistentFit, Plan
from .torch_components import _check_dtype, _resolve_device


def shard_models(rank: int, world: int, n_models: int):
    """Contiguous [lo, hi) slice of the model dimension owned by `rank` (sizes differ by at most one)."""
    base, extra = divmod(n_models, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


class SecondVolterraV2VAESweep:
    """n_models real-PAM SecondVolterraV2VAE models (lib/real_equalization.py:988-1011) with identical shape and
    hyper-parameters, each with its own received sequence, parameters, optimiser state and noise variance."""

    def __init__(self, n_models, channel_memory, equaliser_n_lags1, equaliser_n_lags2, learning_rate, constellation,
                 samples_per_symbol, batch_size, dtype=torch.float32, noise_variance=1.0, adaptive_noise_variance=True,
                 noise_scaling=True, torch_device=None, lr_schedule='step'):
        _check_dtype(dtype)
        self.torch_device = _resolve_device(torch_device)
        assert samples_per_symbol <= channel_memory
        self.n_models, self.channel_memory = n_models, channel_memory
        self.l1, self.l2 = equaliser_n_lags1, equaliser_n_lags2
        self.samples_per_symbol, self.batch_size = samples_per_symbol, batch_size
        self.learning_rate, self.lr_schedule = learning_rate, lr_schedule
        self.adaptive_noise_variance, self.learnable_noise_scaling = adaptive_noise_variance, noise_scaling
        dev = self.torch_device
        self.constellation = torch.from_numpy(np.asarray(constellation)).to(torch.float32).to(dev).contiguous()
        C = len(self.constellation)
        self.constellation_prior = torch.ones((C,), dtype=torch.float32, device=dev) / C
        self._fit = PersistentFit(ns=1, sps=samples_per_symbol, l1=self.l1, pad1=self.l1 // 2, l2=self.l2, L=channel_memory,
                                  C_levels=C, batch_symbols=batch_size, n_models=n_models, device=dev)
        n = self._fit.sizes
        self._o = np.cumsum([0] + n)
        self.params = torch.zeros((n_models, self._fit.pack), dtype=torch.float32, device=dev)
        self.params[:, channel_memory // 2] = 1.0                    # Dirac channel (:753-756)
        self.params[:, self._o[2] + self.l1 // 2] = 1.0              # Dirac equaliser (torch_components.py:69-72)
        self.params[:, self._o[4]:] = 1.0                            # noise scaling (:779-781)
        self._grads, self._m, self._v, self._vmax = (torch.zeros_like(self.params) for _ in range(4))
        nv = np.broadcast_to(np.asarray(noise_variance, dtype=np.float32), (n_models,)).copy()
        self.noise_variance = torch.from_numpy(nv).to(dev)
        self.step_count = 0
        self._current_lr = learning_rate   # the optimiser's rate; StepLR decays it in place across fit() calls

    # views with the reference's attribute names, leading dimension = model
    @property
    def channel_h1(self):
        return self.params[:, self._o[0]:self._o[1]]

    @property
    def channel_h2(self):
        L = self.channel_memory
        return self.params[:, self._o[1]:self._o[2]].view(self.n_models, L, L)

    @property
    def kernel1(self):
        return self.params[:, self._o[2]:self._o[3]]

    @property
    def kernel2(self):
        return self.params[:, self._o[3]:self._o[4]].view(self.n_models, self.l2, self.l2)

    @property
    def noise_scaling(self):
        return self.params[:, self._o[4]:self._o[5]]

    def _learning_rates(self, n_batches):
        """One learning rate per batch of this fit() call.  As in the reference, every fit() creates a new StepLR (10 steps
        down to 1 % over the call, lib/real_equalization.py:476-480) on the SAME optimiser, and a new StepLR starts from the
        optimiser's current - already decayed - rate, so the rate keeps falling across calls."""
        if not self.lr_schedule:
            return [self.learning_rate] * n_batches
        gamma = np.exp(np.log(0.01) / 10)
        every = n_batches // 10
        if every == 0:
            raise ZeroDivisionError("StepLR needs at least 10 batches per fit (as in the reference)")
        lrs, cur = [], self._current_lr
        for n in range(n_batches):
            lrs.append(cur)
            if (n + 1) % every == 0:   # StepLR.step() after batch n: last_epoch = n + 1
                cur *= gamma
        self._current_lr = cur
        return lrs

    def fit(self, y_input):
        """y_input: [n_models, N] (one received sequence per model) or [N] (shared).  Returns
        (y_eq [n_models, N // sps], loss_curves [n_models, n_batches])."""
        y_np = np.ascontiguousarray(y_input)
        S, B = self.samples_per_symbol, self.batch_size
        n_batches = y_np.shape[-1] // S // B
        if n_batches * S * B != y_np.shape[-1]:
            print(f"Warning! Number of samples in receiver signal does not match with a multiplum of the symbol length ({S})")
        y = torch.from_numpy(y_np).to(torch.float32).to(self.torch_device)
        y = y.view(1, -1) if y.dim() == 1 else y.view(self.n_models, 1, -1)
        y_eq = torch.zeros((self.n_models, 1, n_batches * B), dtype=torch.float32, device=self.torch_device)
        losses = torch.zeros((self.n_models, max(n_batches, 1)), dtype=torch.float32, device=self.torch_device)
        if n_batches:
            self._fit.run(y=y, levels=self.constellation.view(1, -1), prior=self.constellation_prior.view(1, -1),
                          params=self.params, grads=self._grads, adam_m=self._m, adam_v=self._v, adam_vmax=self._vmax,
                          noise_var=self.noise_variance, lrs=self._learning_rates(n_batches), step0=self.step_count,
                          y_eq=y_eq, loss_curve=losses[:, :n_batches], learn_scaling=self.learnable_noise_scaling,
                          adaptive_noise=self.adaptive_noise_variance)
            self.step_count += n_batches
        out = np.zeros((self.n_models, y_np.shape[-1] // S), dtype=np.float32)
        out[:, :n_batches * B] = y_eq[:, 0].cpu().numpy()
        return out, losses[:, :n_batches].cpu().numpy().astype(np.float64)

    def decide(self, y_input):
        """Hard symbol decisions (level indices) of every model on [n_models, N] or shared [N] samples:
        equaliser + arg max of the posterior (lib/real_equalization.py:506-510, lib/utility.py:60-69)."""
        y_np = np.ascontiguousarray(y_input)
        y = torch.from_numpy(y_np).to(torch.float32).to(self.torch_device)
        n = y.shape[-1]
        plan = Plan(ns=1, sps=self.samples_per_symbol, l1=self.l1, pad1=self.l1 // 2, l2=self.l2, L=self.channel_memory,
                    C_levels=len(self.constellation), n_glob=n, device=self.torch_device)
        out = torch.empty((self.n_models, n // self.samples_per_symbol), dtype=torch.int32, device=self.torch_device)
        xhat = torch.empty((1, n // self.samples_per_symbol), dtype=torch.float32, device=self.torch_device)
        for m in range(self.n_models):
            ym = (y if y.dim() == 1 else y[m]).reshape(1, -1).contiguous()
            k2 = self.kernel2[m].contiguous() if self.l2 else None
            plan.prep_tables(k1=self.kernel1[m].contiguous(), k2=k2)
            plan.eq_fwd(ym, xhat)
            out[m] = plan.decide(xhat, self.constellation.view(1, -1), self.noise_scaling[m].contiguous().view(1, -1),
                                 self.noise_variance[m:m + 1])[0]
        return out.cpu().numpy()
