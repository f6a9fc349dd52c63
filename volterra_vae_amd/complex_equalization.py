"""Blind V2VAE equalisers for complex QAM.

The reference's complex file stops at a linear channel model (lib/complex_equalization.py:746-916);
these classes extend its conventions to the second-order Volterra channel model of the real path:

* tensors are real-stacked [2, N] (Re row, Im row; `to_real_stacked` / `to_complex`, :677-681);
* the constellation is factorised into its unique real and imaginary amplitude levels with
  independent I and Q soft-max posteriors, prior 1/M per axis (:778-787, :808-817);
* the optimiser (Adam, amsgrad) is built in the constructor and there is no learning-rate schedule
  (:790-792); `fit` returns only the complex equalised signal (:683-709).

Model (no reference counterpart -- parity is pinned by this repository's CPU oracle only):
  channel    yhat_n = sum_i h1[i] x[n+L-1-i] + x_w^T (h2 + h2^T) x_w   with complex h1 [L], h2 [L, L]
             (plain products, no conjugation; reduces to the real model for real data),
  equaliser  xhat_b = sum_j k1[j] y[bS + j - l1//2] + Y_b^T (k2 + k2^T) Y_b  (second-order) or the
             first term alone (linear), complex k1, k2.
Parameters are stored as [2, ...] real tensors (Re plane, Im plane).
"""
from __future__ import annotations

import numpy as np
import torch

from .ops import ElboLossFn, EqualiserFn, Plan
from .real_equalization import PackedAdamAMSGrad
from .torch_components import _check_dtype, _resolve_device


class GenericTorchBlindEqualizer(object):
    """Complex-signal fit/apply loop, lib/complex_equalization.py:658-743."""

    def __init__(self, samples_per_symbol, batch_size, dtype=torch.float32, torch_device=None, flex_update_interval=None) -> None:
        _check_dtype(dtype)
        self.samples_per_symbol = samples_per_symbol
        self.batch_size = batch_size
        self.torch_device = _resolve_device(torch_device)
        self.dtype = dtype
        self.complex_dtype = torch.complex64
        self.loss_print_interval = 100
        self.needs_cpr = False

    def _check_input(self, input_array):
        n_batches = len(input_array) // self.samples_per_symbol // self.batch_size
        if n_batches * self.samples_per_symbol * self.batch_size != len(input_array):
            print(f"Warning! Number of samples in receiver signal does not match with a multiplum of the symbol length ({self.samples_per_symbol})")
        return n_batches

    def to_real_stacked(self, x_cmplx_array):
        return np.concatenate((np.real(x_cmplx_array[np.newaxis, :]), np.imag(x_cmplx_array[np.newaxis, :])), axis=0)

    def to_complex(self, x_stacked_array):
        return np.squeeze(x_stacked_array[0, :] + 1j * x_stacked_array[1, :])

    def apply(self, y_input):
        y = torch.from_numpy(self.to_real_stacked(y_input)).type(self.dtype).to(device=self.torch_device)
        with torch.set_grad_enabled(False):
            y_eq = self.forward(y)
        return self.to_complex(y_eq.cpu().numpy())


class VAESecondVolterraForward(GenericTorchBlindEqualizer):
    """Complex V2VAE parent class: second-order Volterra channel model, pluggable equaliser."""
    IS_PROBABILISTIC = True
    _freeze_h2 = False  # True in the linear-channel class: the second-order channel kernel stays at zero

    def __init__(self, channel_memory, learning_rate, constellation: np.ndarray,
                 samples_per_symbol, batch_size, dtype=torch.float32, noise_variance=1.0,
                 adaptive_noise_variance=True, noise_scaling=True, torch_device=None, flex_update_interval=None,
                 **equaliser_kwargs) -> None:
        super().__init__(samples_per_symbol, batch_size, dtype, torch_device, flex_update_interval)
        assert (samples_per_symbol <= channel_memory)
        self.channel_memory = L = channel_memory
        const = np.asarray(constellation).astype(np.complex128)
        re, im = np.unique(const.real), np.unique(const.imag)
        assert len(re) == len(im)
        M = len(re)
        self.constellation_real = torch.from_numpy(re).to(self.dtype).to(self.torch_device)
        self.constellation_imag = torch.from_numpy(im).to(self.dtype).to(self.torch_device)
        self.constellation_size = 2 * M
        self.constellation_amplitudes = torch.concatenate((self.constellation_real, self.constellation_imag))
        self._levels = torch.stack((self.constellation_real, self.constellation_imag)).contiguous()
        self.constellation_prior = torch.ones((2 * M,), dtype=self.dtype, device=self.torch_device) / M
        self._prior = self.constellation_prior.view(2, M).contiguous()
        self._eq_sizes = self._equaliser_sizes(**equaliser_kwargs)
        l1, l2 = self._eq_sizes
        n = [2 * L, 2 * L * L, 2 * l1, 2 * l2 * l2, 2 * M]
        o = np.cumsum([0] + n)
        self._pack = torch.zeros(int(o[-1]), dtype=self.dtype, device=self.torch_device)
        self._sections = [(int(o[i]), n[i]) for i in range(5)]
        self.channel_h1 = self._pack[o[0]:o[1]].view(2, L)
        self.channel_h1.data[0, L // 2] = 1.0
        self.channel_h2 = self._pack[o[1]:o[2]].view(2, L, L)
        self.equaliser_k1 = self._pack[o[2]:o[3]].view(2, l1)
        self.equaliser_k1.data[0, l1 // 2] = 1.0
        self.equaliser_k2 = self._pack[o[3]:o[4]].view(2, l2, l2) if l2 else None
        self.noise_scaling = self._pack[o[4]:o[5]].view(2, M)
        self.noise_scaling.data.fill_(1.0)
        self.learnable_noise_scaling = noise_scaling
        for p in self._trainable():
            p.requires_grad = True
        self._noise_var = torch.full((1,), float(noise_variance), dtype=self.dtype, device=self.torch_device)
        self.adaptive_noise_variance = adaptive_noise_variance
        self.learning_rate = learning_rate
        self.epsilon = 1e-12
        self._plans = {}
        groups = [{'params': [self.channel_h1] + ([] if self._freeze_h2 else [self.channel_h2]), 'name': 'channel'},
                  {'params': [p for p in (self.equaliser_k1, self.equaliser_k2) if p is not None], 'name': 'equaliser'}]
        offsets = {id(self.channel_h1): self._sections[0], id(self.channel_h2): self._sections[1],
                   id(self.equaliser_k1): self._sections[2]}
        if self.equaliser_k2 is not None:
            offsets[id(self.equaliser_k2)] = self._sections[3]
        if noise_scaling:
            groups.append({'params': [self.noise_scaling], 'name': 'noise_scale'})
            offsets[id(self.noise_scaling)] = self._sections[4]
        self.optimizer = PackedAdamAMSGrad(groups, self._pack, offsets, lr=self.learning_rate)

    def _trainable(self):
        ps = [self.channel_h1, self.equaliser_k1] if self._freeze_h2 else [self.channel_h1, self.channel_h2, self.equaliser_k1]
        if self.equaliser_k2 is not None:
            ps.append(self.equaliser_k2)
        if self.learnable_noise_scaling:
            ps.append(self.noise_scaling)
        return ps

    @property
    def noise_variance(self):
        return self._noise_var[0]

    @noise_variance.setter
    def noise_variance(self, value):
        with torch.no_grad():
            self._noise_var[0] = float(value) if not torch.is_tensor(value) else value.detach().to(self._noise_var)

    def _equaliser_sizes(self, **equaliser_kwargs):
        raise NotImplementedError

    def _plan(self, n_samples) -> Plan:
        if n_samples not in self._plans:
            l1, l2 = self._eq_sizes
            self._plans[n_samples] = Plan(ns=2, sps=self.samples_per_symbol, l1=l1, pad1=l1 // 2, l2=l2,
                                          L=self.channel_memory, C_levels=self.constellation_size // 2,
                                          n_glob=n_samples, device=self.torch_device)
        return self._plans[n_samples]

    def forward(self, y):
        """y [2, N] real-stacked -> xhat [2, ceil(N / sps)]."""
        y = y.to(self.dtype).contiguous()
        pad = (-y.shape[-1]) % self.samples_per_symbol
        if pad:
            y = torch.nn.functional.pad(y, (0, pad))
        return EqualiserFn.apply(self._plan(y.shape[-1]), y, self.equaliser_k1, self.equaliser_k2)

    def _soft_demapping(self, xhat):
        """[2M, B]: rows 0..M-1 the I-axis posterior, rows M..2M-1 the Q-axis posterior (:808-817)."""
        xh = xhat.detach().to(self.dtype).contiguous()
        B = xh.shape[1]
        M = self.constellation_size // 2
        q = torch.empty((2, M, B), dtype=self.dtype, device=self.torch_device)
        self._plan(B * self.samples_per_symbol).demap_moments(xh, self._levels, self._prior, self.noise_scaling.detach().contiguous(),
                                                              self._noise_var, q_out=q)
        return q.view(2 * M, B)

    def forward_probs(self, y_in):
        return self._soft_demapping(self.forward(y_in))

    def estimate_symbol_probs(self, y_input):
        y = torch.from_numpy(self.to_real_stacked(y_input)).type(self.dtype).to(device=self.torch_device)
        with torch.set_grad_enabled(False):
            q = self.forward_probs(y)
        return q.cpu().numpy()

    def _calculate_loss(self, xhat, y):
        n = y.shape[-1]
        loss, E = ElboLossFn.apply(self._plan(n), xhat, y.to(self.dtype).contiguous(), self.channel_h1, self.channel_h2,
                                   self.noise_scaling, self._noise_var, self._levels, self._prior)
        if self.adaptive_noise_variance:
            with torch.no_grad():
                self._noise_var[0] = E / (n - self.channel_memory + 1)
        return loss

    def _update_model(self, loss):
        loss.backward(retain_graph=True)
        self.optimizer.step()
        for p in self._trainable():
            p.grad = None

    def fit(self, y_input):
        n_batches = self._check_input(y_input)
        y = torch.from_numpy(self.to_real_stacked(y_input)).type(self.dtype).to(device=self.torch_device)
        S, B = self.samples_per_symbol, self.batch_size
        y_eq = torch.zeros((2, y.shape[-1] // S), dtype=y.dtype, device=self.torch_device)
        self.loss_curve = torch.zeros(max(n_batches, 1), dtype=torch.float32, device=self.torch_device)
        opt = self.optimizer
        plan = self._plan(B * S) if n_batches else None
        xhat = torch.empty((2, B), dtype=self.dtype, device=self.torch_device)
        for n in range(n_batches):
            ysl = y[:, n * B * S:(n + 1) * B * S].contiguous()
            opt.step_count += 1
            plan.train_step(y=ysl, levels=self._levels, prior=self._prior, params=self._pack, grads=opt.flat_grad,
                            adam_m=opt.exp_avg, adam_v=opt.exp_avg_sq, adam_vmax=opt.max_exp_avg_sq,
                            noise_var=self._noise_var, xhat=xhat, loss_out=self.loss_curve[n:n + 1], lr=opt.current_lr(),
                            step=opt.step_count, learn_scaling=self.learnable_noise_scaling,
                            adaptive_noise=self.adaptive_noise_variance, freeze_h2=self._freeze_h2)
            if n % self.loss_print_interval == 0:
                print(f"Batch {n}, Loss: {self.loss_curve[n].item():.3f}")
            y_eq[:, n * B:(n + 1) * B] = xhat
        return self.to_complex(y_eq.cpu().numpy())

    def print_model_parameters(self):
        print(f"Channel (1st order): {self.channel_h1}")
        print(f"Channel (2nd order): {self.channel_h2}")
        print(f"Equaliser (1st order): {self.equaliser_k1}")
        print(f"Equaliser (2nd order): {self.equaliser_k2}")
        print(f"Noise variance: {self.noise_variance}")

    def train_mode(self):
        for p in self._trainable():
            p.requires_grad = True

    def eval_mode(self):
        for p in self._trainable():
            p.requires_grad = False


class SecondVolterraV2VAE(VAESecondVolterraForward):
    """Complex V2VAE with a complex second-order Volterra equaliser
    (kwargs equaliser_n_lags1, equaliser_n_lags2 as in the real class, lib/real_equalization.py:998-1004)."""

    def _equaliser_sizes(self, **equaliser_kwargs):
        l1, l2 = equaliser_kwargs['equaliser_n_lags1'], equaliser_kwargs['equaliser_n_lags2']
        assert (l1 + 1) % 2 == 0 and (l2 + 1) % 2 == 0
        return l1, l2

    def __repr__(self) -> str:
        return "ComplexSecondVolterraV2VAE"


class LinearV2VAE(VAESecondVolterraForward):
    """Complex V2VAE with a complex FIR equaliser (kwarg equaliser_n_taps, odd)."""

    def _equaliser_sizes(self, **equaliser_kwargs):
        n = equaliser_kwargs['equaliser_n_taps']
        assert (n + 1) % 2 == 0
        return n, 0

    def __repr__(self) -> str:
        return "ComplexLinV2VAE"


class LinearVAE(VAESecondVolterraForward):
    """The reference's complex VAE with LINEAR channel model and linear equaliser (lib/complex_equalization.py:746-916) on the
    same kernels: second-order channel kernel frozen at zero, unit noise scaling.

    The reference stores both filters as Conv1d weights [1, 2, n] = (w0, w1) and applies them as cross-correlations,
    the equaliser "pre-conjugated" (:908-913); in this package's conventions (convolution channel, plain-product
    equaliser) that is   h1 = flip(w0 + i w1)   and   k1 = conj(w0 + i w1).   `encoder_weight` / `decoder_weight` give
    the parameters in the reference's layout, `load_reference_weights` takes them."""
    _freeze_h2 = True

    def __init__(self, encoder_n_taps, learning_rate, constellation: np.ndarray, samples_per_symbol,
                 batch_size, dtype=torch.float32, noise_variance=1, adaptive_noise_variance=True, torch_device=None,
                 flex_update_interval=None, **decoder_kwargs) -> None:
        super().__init__(channel_memory=encoder_n_taps, learning_rate=learning_rate, constellation=constellation,
                         samples_per_symbol=samples_per_symbol, batch_size=batch_size, dtype=dtype,
                         noise_variance=noise_variance, adaptive_noise_variance=adaptive_noise_variance, noise_scaling=False,
                         torch_device=torch_device, flex_update_interval=flex_update_interval,
                         equaliser_n_taps=decoder_kwargs['decoder_n_taps'])
        self.encoder_n_taps = encoder_n_taps
        # Dirac initialisation of the reference's correlation filter (:763) sits at tap n // 2 of w0, i.e. at the mirrored
        # tap of the convolution kernel
        with torch.no_grad():
            self.channel_h1.zero_()
            self.channel_h1[0, encoder_n_taps - 1 - encoder_n_taps // 2] = 1.0

    def _equaliser_sizes(self, **equaliser_kwargs):
        n = equaliser_kwargs['equaliser_n_taps']
        assert (n + 1) % 2 == 0
        return n, 0

    @property
    def encoder_weight(self):
        return torch.flip(self.channel_h1.detach(), dims=(1,)).unsqueeze(0).clone()

    @property
    def decoder_weight(self):
        k = self.equaliser_k1.detach()
        return torch.stack((k[0], -k[1])).unsqueeze(0)

    def load_reference_weights(self, encoder_weight=None, decoder_weight=None):
        with torch.no_grad():
            if encoder_weight is not None:
                w = torch.as_tensor(encoder_weight, dtype=self.dtype, device=self.torch_device).reshape(2, -1)
                self.channel_h1.copy_(torch.flip(w, dims=(1,)))
            if decoder_weight is not None:
                w = torch.as_tensor(decoder_weight, dtype=self.dtype, device=self.torch_device).reshape(2, -1)
                self.equaliser_k1[0].copy_(w[0])
                self.equaliser_k1[1].copy_(-w[1])

    def print_model_parameters(self):
        print(f"encoder: {self.encoder_weight}")
        print(f"decoder: {self.decoder_weight}")
        print(f"Noise variance: {self.noise_variance}")

    def __repr__(self) -> str:
        return "LinVAE"
