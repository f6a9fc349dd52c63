"""Drop-in blind V2VAE equalisers for real PAM, interface of lib/real_equalization.py.

Class names, constructor signatures, attributes, method names, printed messages and error
conventions follow the reference (`GenericTorchBlindEqualizer` :434-542,
`GenericTorchBlindProbabilisticEqualizer` :545-560, `VAESecondVolterraForward` :739-958,
`LinearV2VAE` :961-985, `SecondVolterraV2VAE` :988-1011, `SecondVolterraV2VAENoNoiseScaling`
:1014-1037).  The arithmetic runs in the HIP kernels of libv2vae_hip.so:

* `fit` drives one fused kernel sequence per mini-batch (forward, ELBO, backward, Adam-AMSGrad)
  with no host synchronisation except where the reference prints the loss;
* `forward`, `_calculate_loss` and `_update_model` remain separately callable and autograd
  compatible, so user code that steps the model by hand keeps working.

All trainable tensors are views into one flat float32 buffer [h1 | h2 | equaliser | noise scaling],
which is what lets the optimiser be a single kernel launch.
"""
from __future__ import annotations

import numpy as np
import torch

from . import ops
from ._lib import V2VAEError
from .ops import ElboLossFn, PersistentFit, Plan
from .torch_components import (SecondOrderVolterraSeries, StridedFIREqualiser, _check_dtype, _resolve_device,
                               carry_remainder)


class PackedAdamAMSGrad(torch.optim.Optimizer):
    """torch.optim.Adam(amsgrad=True) semantics (what lib/real_equalization.py:457-459 constructs) on
    parameters that are views of one flat buffer; state lives in flat buffers of the same layout."""

    def __init__(self, param_groups, pack, offsets, lr, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(param_groups, dict(lr=lr, betas=betas, eps=eps, amsgrad=True))
        self.pack = pack
        self.offsets = offsets  # {id(param): (start, numel)}
        self.exp_avg = torch.zeros_like(pack)
        self.exp_avg_sq = torch.zeros_like(pack)
        self.max_exp_avg_sq = torch.zeros_like(pack)
        self.flat_grad = torch.zeros_like(pack)
        self.step_count = 0
        self.n_trainable = max(s + n for s, n in offsets.values())

    def current_lr(self):
        return float(self.param_groups[0]["lr"])

    @torch.no_grad()
    def step(self, closure=None):
        self.step_count += 1
        segs = []
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                s, n = self.offsets[id(p)]
                self.flat_grad[s:s + n].copy_(p.grad.reshape(-1))
                segs.append((s, n, float(group["lr"]), group["betas"], group["eps"]))
        segs.sort()
        merged = []
        for seg in segs:  # merge adjacent sections with identical hyper-parameters into one launch
            if merged and merged[-1][0] + merged[-1][1] == seg[0] and merged[-1][2:] == seg[2:]:
                merged[-1] = (merged[-1][0], merged[-1][1] + seg[1]) + merged[-1][2:]
            else:
                merged.append(seg)
        for s, n, lr, betas, eps in merged:
            ops.adam_amsgrad_(self.pack[s:], self.flat_grad[s:], self.exp_avg[s:], self.exp_avg_sq[s:],
                              self.max_exp_avg_sq[s:], lr, self.step_count, betas[0], betas[1], eps, n=n)


class GenericTorchBlindEqualizer(object):
    """Parent class for blind equalisers with torch-style optimisation (lib/real_equalization.py:434-542)."""
    IS_PROBABILISTIC = False

    def __init__(self, samples_per_symbol, batch_size, learning_rate, dtype=torch.float32, torch_device=None,
                 lr_schedule='step') -> None:
        _check_dtype(dtype)
        self.samples_per_symbol = samples_per_symbol
        self.batch_size = batch_size
        self.torch_device = _resolve_device(torch_device)
        self.dtype = dtype
        self.loss_print_interval = 100
        self.learning_rate = learning_rate
        self.optimizer = None
        self.lr_schedule = lr_schedule

    def _check_input(self, input_array):
        n_batches = len(input_array) // self.samples_per_symbol // self.batch_size
        if n_batches * self.samples_per_symbol * self.batch_size != len(input_array):
            print(f"Warning! Number of samples in receiver signal does not match with a multiplum of the symbol length ({self.samples_per_symbol})")
        return n_batches

    def initialize_optimizer(self):
        raise NotImplementedError

    def _make_lr_schedule(self, n_batches):
        # StepLR: 10 steps down to 1 % of the initial rate over one fit() call (:476-480)
        steps = 10
        reduction = 0.01
        gamma = np.exp(np.log(reduction) / steps)
        return torch.optim.lr_scheduler.StepLR(self.optimizer, step_size=n_batches // steps, gamma=gamma)

    def fit(self, y_input):
        raise NotImplementedError

    def apply(self, y_input):
        y = torch.from_numpy(np.ascontiguousarray(y_input)).type(self.dtype).to(device=self.torch_device)
        with torch.set_grad_enabled(False):
            y_eq = self.forward(y)
        return y_eq.cpu().numpy()

    def get_parameters(self):
        raise NotImplementedError

    def _update_model(self, loss):
        raise NotImplementedError

    def forward(self, y):
        raise NotImplementedError

    def _calculate_loss(self, xhat, y):
        raise NotImplementedError

    def print_model_parameters(self):
        raise NotImplementedError

    def train_mode(self):
        raise NotImplementedError

    def eval_mode(self):
        raise NotImplementedError

    def __repr__(self) -> str:
        raise NotImplementedError


class GenericTorchBlindProbabilisticEqualizer(GenericTorchBlindEqualizer):
    """Adds probabilistic outputs (lib/real_equalization.py:545-560)."""
    IS_PROBABILISTIC = True

    def estimate_symbol_probs(self, y_input):
        y = torch.from_numpy(np.ascontiguousarray(y_input)).type(self.dtype).to(device=self.torch_device)
        with torch.set_grad_enabled(False):
            q_eq = self.forward_probs(y)
        return q_eq.cpu().numpy()

    def forward_probs(self, y_in):
        raise NotImplementedError


class VAESecondVolterraForward(GenericTorchBlindProbabilisticEqualizer):
    """Blind VAE equaliser with a second-order Volterra channel model (lib/real_equalization.py:739-958)."""

    def __init__(self, channel_memory, learning_rate, constellation: np.ndarray,
                 samples_per_symbol, batch_size, dtype=torch.float32, noise_variance=1.0,
                 adaptive_noise_variance=True, noise_scaling=True,
                 torch_device=None, lr_schedule='step', **equaliser_kwargs) -> None:
        super().__init__(samples_per_symbol, batch_size, learning_rate, dtype, torch_device, lr_schedule)
        assert (samples_per_symbol <= channel_memory)
        self.channel_memory = L = channel_memory
        self.constellation = torch.from_numpy(np.asarray(constellation)).type(self.dtype).to(self.torch_device).contiguous()
        self.constellation_size = C = len(self.constellation)
        self._eq_sizes = self._equaliser_sizes(**equaliser_kwargs)  # (l1, l2)
        l1, l2 = self._eq_sizes
        h2_shape = self._channel_h2_shape(L)
        n = [L, int(np.prod(h2_shape)), l1, l2 * l2, C]
        o = np.cumsum([0] + n)
        self._pack = torch.zeros(int(o[-1]), dtype=self.dtype, device=self.torch_device)
        self._sections = [(int(o[i]), n[i]) for i in range(5)]

        # channel model (:753-761): Dirac first-order kernel, zero second-order kernel
        self.channel_h1 = self._pack[o[0]:o[1]]
        self.channel_h1.data[L // 2] = 1.0
        self.channel_h1.requires_grad = True
        self.channel_h2 = self._pack[o[1]:o[2]].view(*h2_shape)
        self.channel_h2.requires_grad = True

        # equaliser (:764)
        self._eq_storage = (self._pack[o[2]:o[3]], self._pack[o[3]:o[4]].view(l2, l2) if l2 else None)
        self.equaliser = self.initialize_equaliser(**equaliser_kwargs)
        self.equaliser._host = (L, C)

        # noise variance (:767-769): a device scalar that the kernels read and (if adaptive) update in place
        self._noise_var = torch.full((1,), float(noise_variance), dtype=self.dtype, device=self.torch_device)
        self.adaptive_noise_variance = adaptive_noise_variance
        self.learning_rate_second_order = learning_rate

        # per-symbol noise scaling (:779-781)
        self.noise_scaling = self._pack[o[4]:o[5]]
        self.noise_scaling.data.fill_(1.0)
        self.learnable_noise_scaling = noise_scaling
        self.noise_scaling.requires_grad = True if self.learnable_noise_scaling else False

        self.constellation_prior = torch.ones((C,), dtype=self.dtype, device=self.torch_device) / C
        self.sm = torch.nn.Softmax(dim=-1)
        self.epsilon = 1e-12
        self._plans = {}

    # ---- attribute with the reference's name and tensor type
    @property
    def noise_variance(self):
        return self._noise_var[0]

    @noise_variance.setter
    def noise_variance(self, value):
        with torch.no_grad():
            self._noise_var[0] = float(value) if not torch.is_tensor(value) else value.detach().to(self._noise_var)

    # ---- plumbing
    _channel_mode = None              # C-ABI `mode` of the channel model (None: 0, full second-order kernel)
    _accumulate_scaling_grad = False  # True where the reference never resets the noise-scaling gradient

    def _channel_h2_shape(self, L):
        return (L, L)

    def _equaliser_sizes(self, **equaliser_kwargs):
        raise NotImplementedError

    def _pad1(self):
        l1, l2 = self._eq_sizes
        return l1 // 2 if l2 else (l1 - 1) // 2

    def _plan(self, n_samples) -> Plan:
        if n_samples not in self._plans:
            l1, l2 = self._eq_sizes
            self._plans[n_samples] = Plan(ns=1, sps=self.samples_per_symbol, l1=l1, pad1=self._pad1(), l2=l2,
                                          L=self.channel_memory, C_levels=self.constellation_size, n_glob=n_samples,
                                          device=self.torch_device, mode=self._channel_mode)
        return self._plans[n_samples]

    def _levels2d(self):
        return self.constellation.view(1, -1)

    def initialize_optimizer(self):
        groups = self.get_parameters()
        offsets = {}
        for g in groups:
            g["params"] = list(g["params"]) if not torch.is_tensor(g["params"]) else [g["params"]]
        named = {"channel_h1": 0, "channel_filter": 0, "channel_h": 0, "channel_h2": 1, "noise_scale": 4}
        for g in groups:
            if g["name"] == "equaliser":
                s = self._sections[2][0]
                for p in g["params"]:
                    offsets[id(p)] = (s, p.numel())
                    s += p.numel()
            else:
                sec = self._sections[named[g["name"]]] if g["name"] in named else self._extra_sections()[g["name"]]
                offsets[id(g["params"][0])] = sec
        self.optimizer = PackedAdamAMSGrad(groups, self._pack, offsets, lr=self.learning_rate)

    def get_parameters(self):
        params = [{'params': self.channel_h1, 'name': 'channel_h1'},
                  {'params': self.channel_h2, "lr": self.learning_rate_second_order, 'name': 'channel_h2'},
                  {'params': self.equaliser.parameters(), 'name': 'equaliser'}]
        if self.learnable_noise_scaling:
            params += [{'params': self.noise_scaling, 'name': 'noise_scale'}]
        return params

    def initialize_equaliser(self, **equaliser_kwargs) -> torch.nn.Module:
        raise NotImplementedError

    def forward(self, y) -> torch.Tensor:
        raise NotImplementedError

    def forward_probs(self, y_in):
        y = self.forward(y_in)
        return self._soft_demapping(y)

    def _soft_demapping(self, xhat):
        """[C, B] posteriors (:816-823) through v2vae_demap_moments."""
        xh = xhat.detach().to(self.dtype).reshape(1, -1).contiguous()
        B = xh.shape[1]
        plan = self._plan(B * self.samples_per_symbol)
        q = torch.empty((1, self.constellation_size, B), dtype=self.dtype, device=self.torch_device)
        xlo = getattr(xhat, "_v2vae_xlo", None)
        if xlo is not None and xlo.numel() == B:   # the equaliser's own plan computed xhat: its remainder comes along
            plan.set_xhat_remainder(xh, xlo)
        plan.demap_moments(xh, self._levels2d(), self.constellation_prior.view(1, -1), self.noise_scaling.detach().view(1, -1),
                           self._noise_var, q_out=q)
        return q[0]

    def _calculate_loss(self, xhat, y):
        """ELBO (:836-925) incl. the noise-variance side effect (:921-923); 0-d tensor with grad_fn."""
        n = y.shape[-1]
        plan = self._plan(n)
        loss, E = ElboLossFn.apply(plan, xhat.reshape(1, -1), y.reshape(1, -1).to(self.dtype).contiguous(), self.channel_h1,
                                   self._h2_argument(), self.noise_scaling.view(1, -1), self._noise_var,
                                   self._levels2d(), self.constellation_prior.view(1, -1).contiguous(),
                                   getattr(xhat, "_v2vae_xlo", None))
        if self.adaptive_noise_variance:
            with torch.no_grad():
                self._noise_var[0] = E / (n - self.channel_memory + 1)
        return loss

    def _h2_argument(self):
        return self.channel_h2

    def _extra_sections(self):
        return {}

    def _update_model(self, loss):
        loss.backward(retain_graph=True)
        self.optimizer.step()
        self.channel_h1.grad = None
        self.channel_h2.grad = None
        self.equaliser.zero_grad()
        if self.learnable_noise_scaling:
            self.noise_scaling.grad = None

    # None (default): fit() runs the whole mini-batch loop as ONE launch when the on-chip kernel covers the shape (small
    # real shapes: 42 us per 500-symbol step instead of 171 us), else one fused kernel sequence per batch.
    # True: always one launch (the stage-by-stage persistent kernel for shapes the on-chip one does not cover; it is
    # bitwise equal to the per-batch sequence but slower for a single model).  False: always per batch.
    persistent_fit = None
    _freeze_h2 = False  # True in the linear-channel classes: the second-order channel kernel stays at zero

    def _persistent(self, B):
        key = ("persistent", B)
        if key not in self._plans:
            l1, l2 = self._eq_sizes
            try:
                self._plans[key] = PersistentFit(ns=1, sps=self.samples_per_symbol, l1=l1, pad1=self._pad1(), l2=l2,
                                                 L=self.channel_memory, C_levels=self.constellation_size,
                                                 batch_symbols=B, n_models=1, device=self.torch_device,
                                                 mode=self._channel_mode)
            except V2VAEError:
                self._plans[key] = None
        return self._plans[key]

    def fit(self, y_input):
        """Mini-batch training loop of lib/real_equalization.py:461-504.  Small batches: the whole loop is one
        persistent kernel launch (the per-batch loss lines are then printed after the launch has finished);
        otherwise one fused kernel sequence per batch."""
        n_batches = self._check_input(y_input)
        if self.optimizer is None:
            raise Exception("Optimizer has not been initialized yet. Please call 'initialize_optimizer' before proceeding to 'fit'.")
        y = torch.from_numpy(np.ascontiguousarray(y_input)).type(self.dtype).to(device=self.torch_device)
        S, B = self.samples_per_symbol, self.batch_size
        y_eq = torch.zeros((y.shape[-1] // S,), dtype=y.dtype, device=self.torch_device)
        lr_schedule = self._make_lr_schedule(n_batches) if self.lr_schedule else None
        losses = torch.zeros(max(n_batches, 1), dtype=torch.float32, device=self.torch_device)
        opt = self.optimizer
        opt._opt_called = True  # the fused step below is the optimizer step
        levels, prior = self._levels2d(), self.constellation_prior.view(1, -1).contiguous()
        acc = self._accumulate_scaling_grad and self.learnable_noise_scaling
        if acc:  # the running gradient sum lives in noise_scaling.grad between calls, as in the reference
            s0, nsc = self._sections[4]
            if self.noise_scaling.grad is None:
                opt.flat_grad[s0:s0 + nsc].zero_()
            else:
                opt.flat_grad[s0:s0 + nsc].copy_(self.noise_scaling.grad)
        pf = None
        if self.persistent_fit is not False and n_batches and B <= self.PERSISTENT_MAX_BATCH:
            pf = self._persistent(B)
            if pf is not None and self.persistent_fit is None and not pf.on_chip:
                pf = None
        if pf is not None:
            lrs, shown = [], []
            for n in range(n_batches):
                lrs.append(opt.current_lr())
                shown.append(lr_schedule.get_last_lr()[-1] if self.lr_schedule else self.learning_rate)
                if lr_schedule:
                    lr_schedule.step()
            yq = torch.zeros((1, 1, n_batches * B), dtype=y.dtype, device=self.torch_device)
            pf.run(y=y.view(1, -1), levels=levels, prior=prior, params=self._pack.view(1, -1), grads=opt.flat_grad.view(1, -1),
                   adam_m=opt.exp_avg.view(1, -1), adam_v=opt.exp_avg_sq.view(1, -1), adam_vmax=opt.max_exp_avg_sq.view(1, -1),
                   noise_var=self._noise_var, lrs=lrs, step0=opt.step_count, y_eq=yq, loss_curve=losses[:n_batches].view(1, -1),
                   learn_scaling=self.learnable_noise_scaling, adaptive_noise=self.adaptive_noise_variance,
                   freeze_h2=self._freeze_h2, accumulate_scaling_grad=acc)
            if acc:
                self.noise_scaling.grad = opt.flat_grad[s0:s0 + nsc].clone()
            opt.step_count += n_batches
            y_eq[:n_batches * B] = yq.view(-1)
            loss_curve = losses[:n_batches].cpu().numpy().astype(np.float64)
            for n in range(0, n_batches, self.loss_print_interval):
                print(f"Batch {n}, Loss: {loss_curve[n]:.3f} (LR: {shown[n]:.3e})")
            return y_eq.cpu().numpy(), loss_curve
        plan = self._plan(B * S) if n_batches else None
        for n in range(n_batches):
            ysl = y[n * B * S:(n + 1) * B * S].view(1, -1)
            lr_now = opt.current_lr()
            opt.step_count += 1
            plan.train_step(y=ysl, levels=levels, prior=prior, params=self._pack, grads=opt.flat_grad, adam_m=opt.exp_avg,
                            adam_v=opt.exp_avg_sq, adam_vmax=opt.max_exp_avg_sq, noise_var=self._noise_var,
                            xhat=y_eq[n * B:(n + 1) * B], loss_out=losses[n:n + 1], lr=lr_now, step=opt.step_count,
                            learn_scaling=self.learnable_noise_scaling, adaptive_noise=self.adaptive_noise_variance,
                            freeze_h2=self._freeze_h2, accumulate_scaling_grad=acc)
            if n % self.loss_print_interval == 0:
                lrprint = lr_schedule.get_last_lr()[-1] if self.lr_schedule else self.learning_rate
                print(f"Batch {n}, Loss: {losses[n].item():.3f} (LR: {lrprint:.3e})")
            if lr_schedule:
                lr_schedule.step()
        if acc and n_batches:
            self.noise_scaling.grad = opt.flat_grad[s0:s0 + nsc].clone()
        loss_curve = losses[:n_batches].cpu().numpy().astype(np.float64)
        return y_eq.cpu().numpy(), loss_curve

    # largest batch (symbols) trained by one workgroup; above it the many-workgroup kernels win
    PERSISTENT_MAX_BATCH = 4096

    def print_model_parameters(self):
        print(f"Equaliser: {self.equaliser}")
        print(f"Channel (1st order): {self.channel_h1}")
        print(f"Channel (2nd order): {self.channel_h2}")
        print(f"Noise variance: {self.noise_variance}")
        if self.learnable_noise_scaling:
            print(f"Noise scaling: {self.noise_scaling}")

    def train_mode(self):
        self.channel_h1.requires_grad = True
        self.channel_h2.requires_grad = True
        self.equaliser.train()

    def eval_mode(self):
        self.channel_h1.requires_grad = False
        self.channel_h2.requires_grad = False
        self.equaliser.eval()


class LinearV2VAE(VAESecondVolterraForward):
    """V2VAE with a linear (strided FIR) equaliser, lib/real_equalization.py:961-985."""

    def __init__(self, channel_memory, learning_rate, constellation: np.ndarray, samples_per_symbol,
                 batch_size, dtype=torch.float32, noise_variance=1, adaptive_noise_variance=True,
                 torch_device=None, lr_schedule='step', **equaliser_kwargs) -> None:
        super().__init__(channel_memory=channel_memory, learning_rate=learning_rate, constellation=constellation,
                         samples_per_symbol=samples_per_symbol, batch_size=batch_size, dtype=dtype,
                         noise_variance=noise_variance, adaptive_noise_variance=adaptive_noise_variance,
                         noise_scaling=True, torch_device=torch_device, lr_schedule=lr_schedule,
                         **equaliser_kwargs)

    def _equaliser_sizes(self, **equaliser_kwargs):
        return equaliser_kwargs['equaliser_n_taps'], 0

    def initialize_equaliser(self, **equaliser_kwargs):
        return StridedFIREqualiser(equaliser_kwargs['equaliser_n_taps'], self.samples_per_symbol, dtype=self.dtype,
                                   torch_device=self.torch_device, _storage=self._eq_storage[0])

    def forward(self, y):
        out = self.equaliser.forward(y[None, None, :])
        return carry_remainder(out, out.squeeze())

    def __repr__(self) -> str:
        return "LinV2VAE"


class SecondVolterraV2VAE(VAESecondVolterraForward):
    """V2VAE with a second-order Volterra equaliser, lib/real_equalization.py:988-1011."""

    def __init__(self, channel_memory, learning_rate, constellation: np.ndarray, samples_per_symbol,
                 batch_size, dtype=torch.float32, noise_variance=1, adaptive_noise_variance=True,
                 torch_device=None, lr_schedule='step', **equaliser_kwargs) -> None:
        super().__init__(channel_memory=channel_memory, learning_rate=learning_rate, constellation=constellation,
                         samples_per_symbol=samples_per_symbol, batch_size=batch_size, dtype=dtype,
                         noise_variance=noise_variance, adaptive_noise_variance=adaptive_noise_variance,
                         noise_scaling=True, torch_device=torch_device, lr_schedule=lr_schedule,
                         **equaliser_kwargs)

    def _equaliser_sizes(self, **equaliser_kwargs):
        return equaliser_kwargs['equaliser_n_lags1'], equaliser_kwargs['equaliser_n_lags2']

    def initialize_equaliser(self, **equaliser_kwargs):
        return SecondOrderVolterraSeries(n_lags1=equaliser_kwargs['equaliser_n_lags1'],
                                         n_lags2=equaliser_kwargs['equaliser_n_lags2'],
                                         samples_per_symbol=self.samples_per_symbol,
                                         dtype=self.dtype, torch_device=self.torch_device, _storage=self._eq_storage)

    def forward(self, y):
        return self.equaliser.forward(y)

    def __repr__(self) -> str:
        return "SecondVolterraV2VAE"


class SecondVolterraV2VAENoNoiseScaling(VAESecondVolterraForward):
    """Same with the per-symbol noise scaling frozen at 1, lib/real_equalization.py:1014-1037."""

    def __init__(self, channel_memory, learning_rate, constellation: np.ndarray, samples_per_symbol,
                 batch_size, dtype=torch.float32, noise_variance=1, adaptive_noise_variance=True,
                 torch_device=None, lr_schedule='step', **equaliser_kwargs) -> None:
        super().__init__(channel_memory=channel_memory, learning_rate=learning_rate, constellation=constellation,
                         samples_per_symbol=samples_per_symbol, batch_size=batch_size, dtype=dtype,
                         noise_variance=noise_variance, adaptive_noise_variance=adaptive_noise_variance,
                         noise_scaling=False, torch_device=torch_device, lr_schedule=lr_schedule,
                         **equaliser_kwargs)

    def _equaliser_sizes(self, **equaliser_kwargs):
        return equaliser_kwargs['equaliser_n_lags1'], equaliser_kwargs['equaliser_n_lags2']

    def initialize_equaliser(self, **equaliser_kwargs):
        return SecondOrderVolterraSeries(n_lags1=equaliser_kwargs['equaliser_n_lags1'],
                                         n_lags2=equaliser_kwargs['equaliser_n_lags2'],
                                         samples_per_symbol=self.samples_per_symbol,
                                         dtype=self.dtype, torch_device=self.torch_device, _storage=self._eq_storage)

    def forward(self, y):
        return self.equaliser.forward(y)

    def __repr__(self) -> str:
        return "SecondVolterraV2VAE(Beta=1.0)"


class VAELinearForward(VAESecondVolterraForward):
    """Blind VAE equaliser with a LINEAR channel model (lib/real_equalization.py:564-690).  Its ELBO (:629-664) is the
    second-order one with the quadratic channel kernel identically zero and unit noise scaling, so it runs on the same
    kernels with `channel_h2` frozen at zero; `channel_filter` is the first-order channel kernel."""
    _freeze_h2 = True

    def __init__(self, channel_n_taps, learning_rate, constellation: np.ndarray,
                 samples_per_symbol, batch_size, dtype=torch.float32, noise_variance=1.0,
                 adaptive_noise_variance=True, torch_device=None, lr_schedule='step', **equaliser_kwargs) -> None:
        super().__init__(channel_memory=channel_n_taps, learning_rate=learning_rate, constellation=constellation,
                         samples_per_symbol=samples_per_symbol, batch_size=batch_size, dtype=dtype,
                         noise_variance=noise_variance, adaptive_noise_variance=adaptive_noise_variance,
                         noise_scaling=False, torch_device=torch_device, lr_schedule=lr_schedule, **equaliser_kwargs)
        self.channel_n_taps = channel_n_taps
        self.channel_filter = self.channel_h1
        self.channel_h2.requires_grad = False

    def get_parameters(self):
        return [{'params': self.channel_filter, 'name': 'channel_filter'},
                {'params': self.equaliser.parameters(), 'name': 'equaliser'}]

    def _update_model(self, loss):
        loss.backward(retain_graph=True)
        self.optimizer.step()
        self.channel_filter.grad = None
        self.equaliser.zero_grad()

    def print_model_parameters(self):
        print("Equaliser")
        print(self.equaliser)
        print(f"Channel: {self.channel_filter}")
        print(f"Noise variance: {self.noise_variance}")

    def train_mode(self):
        self.channel_filter.requires_grad = True
        self.equaliser.train()

    def eval_mode(self):
        self.channel_filter.requires_grad = False
        self.equaliser.eval()


class LinearVAE(VAELinearForward):
    """Linear channel, linear (strided FIR) equaliser, lib/real_equalization.py:693-713."""

    def _equaliser_sizes(self, **equaliser_kwargs):
        return equaliser_kwargs['equaliser_n_taps'], 0

    def initialize_equaliser(self, **equaliser_kwargs):
        return StridedFIREqualiser(equaliser_kwargs['equaliser_n_taps'], self.samples_per_symbol, dtype=self.dtype,
                                   torch_device=self.torch_device, _storage=self._eq_storage[0])

    def forward(self, y):
        out = self.equaliser.forward(y[None, None, :])
        return carry_remainder(out, out.squeeze())

    def __repr__(self) -> str:
        return "LinVAE"


class SecondVolterraVAE(VAELinearForward):
    """Linear channel, second-order Volterra equaliser, lib/real_equalization.py:716-736."""

    def __init__(self, channel_memory, learning_rate, constellation: np.ndarray, samples_per_symbol,
                 batch_size, dtype=torch.float32, noise_variance=1, adaptive_noise_variance=True,
                 torch_device=None, lr_schedule='step', **equaliser_kwargs) -> None:
        super().__init__(channel_memory, learning_rate, constellation, samples_per_symbol, batch_size, dtype,
                         noise_variance, adaptive_noise_variance, torch_device, lr_schedule, **equaliser_kwargs)

    def _equaliser_sizes(self, **equaliser_kwargs):
        return equaliser_kwargs['equaliser_n_lags1'], equaliser_kwargs['equaliser_n_lags2']

    def initialize_equaliser(self, **equaliser_kwargs):
        return SecondOrderVolterraSeries(n_lags1=equaliser_kwargs['equaliser_n_lags1'],
                                         n_lags2=equaliser_kwargs['equaliser_n_lags2'],
                                         samples_per_symbol=self.samples_per_symbol,
                                         dtype=self.dtype, torch_device=self.torch_device, _storage=self._eq_storage)

    def forward(self, y):
        return self.equaliser.forward(y)

    def __repr__(self) -> str:
        return "SecondVolterraVAE"


class VAEMemoryPolynomialForward(VAESecondVolterraForward):
    """Blind VAE equaliser with a memory-polynomial channel model y = h1 * x + h2 * x^2 (lib/real_equalization.py:1224-1384;
    ELBO :1304-1354).  That model is the second-order Volterra one with the DIAGONAL kernel diag(channel_h2), so it runs on
    the same kernels (C-ABI mode 2); `channel_h2` is the [L] vector of the reference."""
    _channel_mode = 2
    _accumulate_scaling_grad = True  # :1356-1364 resets every gradient except noise_scaling's, which keeps accumulating

    def _channel_h2_shape(self, L):
        return (L,)

    def get_parameters(self):
        return [{'params': self.channel_h1, 'name': 'channel_h1'},
                {'params': self.channel_h2, 'name': 'channel_h2'},
                {'params': self.equaliser.parameters(), 'name': 'equaliser'},
                {'params': self.noise_scaling, 'name': 'noise_scale'}]

    def _update_model(self, loss):
        loss.backward(retain_graph=True)
        self.optimizer.step()
        self.channel_h1.grad = None
        self.channel_h2.grad = None
        self.equaliser.zero_grad()

    def print_model_parameters(self):
        print("Equaliser")
        self.equaliser.print_parameters()
        print(f"Channel (h1): {self.channel_h1}")
        print(f"Channel (h2): {self.channel_h2}")
        print(f"Noise variance: {self.noise_variance}")


class SecondVolterraMPVAE(VAEMemoryPolynomialForward):
    """Memory-polynomial channel model, second-order Volterra equaliser (lib/real_equalization.py:1387-1410)."""

    def __init__(self, channel_memory, learning_rate, constellation: np.ndarray, samples_per_symbol,
                 batch_size, dtype=torch.float32, noise_variance=1, adaptive_noise_variance=True, noise_scaling=True,
                 torch_device=None, lr_schedule='step', **equaliser_kwargs) -> None:
        super().__init__(channel_memory, learning_rate, constellation, samples_per_symbol, batch_size, dtype, noise_variance,
                         adaptive_noise_variance, noise_scaling, torch_device, lr_schedule, **equaliser_kwargs)

    def _equaliser_sizes(self, **equaliser_kwargs):
        return equaliser_kwargs['equaliser_n_lags1'], equaliser_kwargs['equaliser_n_lags2']

    def initialize_equaliser(self, **equaliser_kwargs):
        return SecondOrderVolterraSeries(n_lags1=equaliser_kwargs['equaliser_n_lags1'],
                                         n_lags2=equaliser_kwargs['equaliser_n_lags2'],
                                         samples_per_symbol=self.samples_per_symbol,
                                         dtype=self.dtype, torch_device=self.torch_device, _storage=self._eq_storage)

    def forward(self, y):
        return self.equaliser.forward(y)

    def __repr__(self) -> str:
        return "SecondVolterraMPVAE"


class VAEHammersteinForward(VAESecondVolterraForward):
    """Blind VAE equaliser with a Hammerstein channel model y = h * (alpha_1 x + alpha_2 x^2)
    (lib/real_equalization.py:1040-1199; ELBO :1117-1167): the memory polynomial with h1 = alpha_1 h, h2 = alpha_2 h, run on the
    same kernels (C-ABI mode 3).  `channel_h`, `alpha_1`, `alpha_2` are views into the flat parameter pack."""
    _channel_mode = 3
    _accumulate_scaling_grad = True  # :1169-1178 never resets noise_scaling.grad

    def __init__(self, channel_memory, learning_rate, constellation: np.ndarray,
                 samples_per_symbol, batch_size, dtype=torch.float32, noise_variance=1.0,
                 adaptive_noise_variance=True, torch_device=None, lr_schedule='step', **equaliser_kwargs) -> None:
        super().__init__(channel_memory=channel_memory, learning_rate=learning_rate, constellation=constellation,
                         samples_per_symbol=samples_per_symbol, batch_size=batch_size, dtype=dtype,
                         noise_variance=noise_variance, adaptive_noise_variance=adaptive_noise_variance,
                         noise_scaling=True, torch_device=torch_device, lr_schedule=lr_schedule, **equaliser_kwargs)
        self.channel_h = self.channel_h1
        s = self._sections[1][0]
        self.channel_h2.requires_grad = False
        self.alpha_1 = self._pack[s]       # 0-d views (:1057-1058)
        self.alpha_2 = self._pack[s + 1]
        self.alpha_1.data.fill_(1.0)
        self.alpha_1.requires_grad = True
        self.alpha_2.requires_grad = True

    def _channel_h2_shape(self, L):
        return (2,)

    def _h2_argument(self):
        return torch.stack((self.alpha_1, self.alpha_2))

    def _extra_sections(self):
        s = self._sections[1][0]
        return {"alpha1": (s, 1), "alpha2": (s + 1, 1)}

    def get_parameters(self):
        return [{'params': self.channel_h, 'name': 'channel_h'},
                {'params': self.alpha_1, 'name': 'alpha1'},
                {'params': self.alpha_2, 'name': 'alpha2'},
                {'params': self.equaliser.parameters(), 'name': 'equaliser'},
                {'params': self.noise_scaling, 'name': 'noise_scale'}]

    def _update_model(self, loss):
        loss.backward(retain_graph=True)
        self.optimizer.step()
        self.channel_h.grad = None
        self.alpha_1.grad = None
        self.alpha_2.grad = None
        self.equaliser.zero_grad()

    def print_model_parameters(self):
        print(f"Equaliser: {self.equaliser}")
        print(f"Channel: {self.channel_h}")
        print(f"Channel (non-linearity): {self.alpha_1:.4f}x + {self.alpha_2:.4f}x^2")
        print(f"Noise variance: {self.noise_variance}")

    def train_mode(self):
        self.channel_h.requires_grad = True
        self.alpha_1.requires_grad = True
        self.alpha_2.requires_grad = True
        self.equaliser.train()

    def eval_mode(self):
        self.channel_h.requires_grad = False
        self.alpha_1.requires_grad = False
        self.alpha_2.requires_grad = False
        self.equaliser.eval()


class SecondVolterraHVAE(VAEHammersteinForward):
    """Hammerstein channel model, second-order Volterra equaliser (lib/real_equalization.py:1202-1221)."""

    def __init__(self, channel_memory, learning_rate, constellation: np.ndarray, samples_per_symbol,
                 batch_size, dtype=torch.float32, noise_variance=1, adaptive_noise_variance=True, torch_device=None,
                 lr_schedule='step', **equaliser_kwargs) -> None:
        super().__init__(channel_memory, learning_rate, constellation, samples_per_symbol, batch_size, dtype, noise_variance,
                         adaptive_noise_variance, torch_device, lr_schedule, **equaliser_kwargs)

    def _equaliser_sizes(self, **equaliser_kwargs):
        return equaliser_kwargs['equaliser_n_lags1'], equaliser_kwargs['equaliser_n_lags2']

    def initialize_equaliser(self, **equaliser_kwargs):
        return SecondOrderVolterraSeries(n_lags1=equaliser_kwargs['equaliser_n_lags1'],
                                         n_lags2=equaliser_kwargs['equaliser_n_lags2'],
                                         samples_per_symbol=self.samples_per_symbol,
                                         dtype=self.dtype, torch_device=self.torch_device, _storage=self._eq_storage)

    def forward(self, y):
        return self.equaliser.forward(y)

    def __repr__(self) -> str:
        return "SecondVolterraHVAE"
