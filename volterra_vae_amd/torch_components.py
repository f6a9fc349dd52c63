"""Drop-in torch modules for the equaliser side of the hot path.

Same constructor arguments, parameter names, initialisation and call conventions as the
reference's lib/torch_components.py (`SecondOrderVolterraSeries` :53-103, `FIRfilter` :31-51),
but forward and backward run in the HIP kernels.  float32 on a GPU only; anything else raises.
"""
from __future__ import annotations

import numpy as np
import torch

from .ops import EqualiserFn, Plan


def _resolve_device(torch_device):
    dev = torch.device("cuda", torch.cuda.current_device()) if torch_device is None else torch.device(torch_device)
    if dev.type != "cuda":
        raise RuntimeError("volterra_vae_amd runs the V2VAE hot path in HIP kernels only: pass a GPU torch_device "
                           f"(got {dev}); there is no CPU implementation in this package")
    if dev.index is None:
        dev = torch.device("cuda", torch.cuda.current_device())
    return dev


def _check_dtype(dtype):
    if dtype != torch.float32:
        raise NotImplementedError(f"the HIP kernels compute in float32; dtype={dtype} is not supported")


def carry_remainder(src: torch.Tensor, dst: torch.Tensor, n: int = None) -> torch.Tensor:
    """`dst` is a reshaped / truncated view of the equaliser output `src`: the float remainder of the output (real model,
    Plan.xhat_remainder) travels with it as the attribute `_v2vae_xlo`, which `_soft_demapping` / `_calculate_loss` of the
    enclosing estimator hand to the soft demapper.  A tensor derived in any other way simply has no remainder."""
    lo = getattr(src, "_v2vae_xlo", None)
    if lo is not None:
        dst._v2vae_xlo = lo if n is None else lo[:n]
    return dst


class _PlannedEqualiser(torch.nn.Module):
    """Shared machinery: a cache of kernel plans keyed by input length."""

    def __init__(self):
        super().__init__()
        self._plans = {}
        self._host = None  # (L, C, ns) of the VAE that owns this equaliser, if any

    def _plan(self, n_samples: int, ns: int = 1) -> Plan:
        key = (n_samples, ns)
        if key not in self._plans:
            L, C = (max(2, self.sps), 1) if self._host is None else self._host
            self._plans[key] = Plan(ns=ns, sps=self.sps, l1=self._l1, pad1=self._pad1, l2=self._l2, L=L, C_levels=C,
                                    n_glob=n_samples, device=self.torch_device)
        return self._plans[key]

    def _run(self, x: torch.Tensor, k1, k2):
        """x: [N] (one stream) or [ns, N]."""
        squeeze = x.dim() == 1
        y = x[None, :] if squeeze else x
        y = y.to(torch.float32).contiguous()
        n = y.shape[-1]
        pad = (-n) % self.sps
        if pad:  # a trailing partial symbol sees zero padding, exactly as the reference's padded conv / unfold
            y = torch.nn.functional.pad(y, (0, pad))
        plan = self._plan(y.shape[-1], y.shape[0])
        out = EqualiserFn.apply(plan, y, k1, k2)
        res = out[0] if squeeze else out
        lo = plan.xhat_remainder() if squeeze else None
        if lo is not None:
            res._v2vae_xlo = lo
        return res


class SecondOrderVolterraSeries(_PlannedEqualiser):
    """Second-order Volterra equaliser, interface of lib/torch_components.py:53-103:
    kernel1 [n_lags1] initialised to a centred Dirac (:69-71), kernel2 [n_lags2, n_lags2] zeros (:75-76),
    odd lag counts asserted (:65-66); forward(x[N]) -> [ceil(N / sps)]."""

    def __init__(self, n_lags1, n_lags2, samples_per_symbol, dtype=torch.float32, torch_device=None,
                 _storage=None, **kwargs) -> None:
        super().__init__(**kwargs)
        _check_dtype(dtype)
        self.sps = samples_per_symbol
        self.dtype = dtype
        self.torch_device = _resolve_device(torch_device)
        assert (n_lags1 + 1) % 2 == 0  # odd lengths
        assert (n_lags2 + 1) % 2 == 0
        self._l1, self._l2, self._pad1 = n_lags1, n_lags2, n_lags1 // 2
        if _storage is None:
            k1 = torch.zeros(n_lags1, dtype=dtype, device=self.torch_device)
            k2 = torch.zeros((n_lags2, n_lags2), dtype=dtype, device=self.torch_device)
        else:  # views into a flat parameter pack owned by the enclosing VAE
            k1, k2 = _storage
            with torch.no_grad():
                k1.zero_()
                k2.zero_()
        with torch.no_grad():
            k1[n_lags1 // 2] = 1.0
        self.kernel1 = torch.nn.Parameter(k1, requires_grad=True)
        self.lag1_padding = n_lags1 // 2
        self.kernel2 = torch.nn.Parameter(k2)
        self.lag2_padding = n_lags2 // 2

    def forward(self, x: torch.Tensor):
        return self._run(x, self.kernel1, self.kernel2)

    def print_parameters(self):
        print("Kernel 1:")
        print(self.kernel1)
        print("Kernel 2:")
        print(self.kernel2)


class StridedFIREqualiser(_PlannedEqualiser):
    """The Conv1d(1, 1, n_taps, stride=sps, padding=(n_taps-1)//2, bias=False) equaliser of `LinearV2VAE`
    (lib/real_equalization.py:971-982): `.weight` has Conv1d's shape [1, 1, n_taps] and a Dirac
    initialisation (nn.init.dirac_); forward takes [1, 1, N] like Conv1d and returns [1, 1, B]."""

    def __init__(self, n_taps, samples_per_symbol, dtype=torch.float32, torch_device=None, _storage=None):
        super().__init__()
        _check_dtype(dtype)
        self.sps = samples_per_symbol
        self.dtype = dtype
        self.torch_device = _resolve_device(torch_device)
        self._l1, self._l2, self._pad1 = n_taps, 0, (n_taps - 1) // 2
        w = torch.zeros(n_taps, dtype=dtype, device=self.torch_device) if _storage is None else _storage
        with torch.no_grad():
            w.zero_()
            w[n_taps // 2] = 1.0
        self.weight = torch.nn.Parameter(w.view(1, 1, n_taps))
        self.kernel_size = (n_taps,)
        self.stride = (samples_per_symbol,)
        self.padding = ((n_taps - 1) // 2,)

    def forward(self, x: torch.Tensor):
        n_out = (x.shape[-1] + 2 * self._pad1 - self._l1) // self.sps + 1
        out = self._run(x.reshape(-1), self.weight.view(-1), None)
        return carry_remainder(out, out[:n_out].view(1, 1, -1), n_out)

    def print_parameters(self):
        print(self.weight)


class FIRfilter(_PlannedEqualiser):
    """Interface of lib/torch_components.py:31-51: true convolution (the stored weights are applied
    flipped, :46), asymmetric zero padding ((n-1)//2, (n - n%2)//2) (:38), Dirac initialisation (:35).
    The reference never instantiates it; it is kept because the equaliser kernels cover it for free:
    it is the first-order equaliser with flipped taps."""

    def __init__(self, filter_length, stride=1, trainable=True, dtype=torch.float32, torch_device=None, *args, **kwargs) -> None:
        super().__init__(*args, **kwargs)
        _check_dtype(dtype)
        self.torch_device = _resolve_device(torch_device)
        filter_weights = np.zeros((filter_length,))
        filter_weights[filter_length // 2] = 1.0
        self.filter_length = filter_length
        self.padding = ((filter_length - 1) // 2, (filter_length - filter_length % 2) // 2)
        self.weights = torch.nn.Parameter(torch.from_numpy(filter_weights).to(dtype).to(self.torch_device), requires_grad=trainable)
        self.trainalbe = trainable
        self.stride = stride
        self.sps = stride
        self._l1, self._l2, self._pad1 = filter_length, 0, self.padding[0]

    def forward(self, x):
        n_out = (x.shape[-1] + self.padding[0] + self.padding[1] - self.filter_length) // self.stride + 1
        out = self._run(x, torch.flip(self.weights, (0,)), None)
        return carry_remainder(out, out[:n_out], n_out)

    def print_parameters(self):
        print(self.weights)
