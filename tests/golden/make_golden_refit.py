"""Golden vectors for the time-varying-channel refit sequence of the reference's change-point script
(main_time_varying_wh.py:66-80 model, :121-149 loop): one SecondVolterraV2VAE (l1=25, l2=15, L=15, lr 5e-4,
lr_schedule=None, 500-symbol batches) is fitted chunk after chunk with the optimiser state, the parameters and the
noise variance carried over, `estimate_symbol_probs` + SER on a fresh validation sequence after every chunk, and the
first FIR of the Wiener-Hammerstein channel swapped between [1, 0.3, 0.1] and [1, 0.5, 0.1525] after every chunk
(lib/data_generation.py:312-315).  Chunks are 20 batches here instead of 1000 so that the fixture stays small.

Runs only where /root/reference exists; the reference is imported unmodified (see make_golden.py for the one
stand-in, torchaudio.functional.convolve).  Output: tests/golden/v2vae_real_refit_cfg3_{f32,f64}.npz (data only).
Usage:  python tests/golden/make_golden_refit.py
"""
import os

import numpy as np
import torch

import make_golden as mg  # registers the stand-in and puts the reference on sys.path

from volterra_vae_amd.synth import FIR1, FIR1_ALT, nonlinear_isi, pam_constellation

HERE = os.path.dirname(os.path.abspath(__file__))


def refit_case(tag, dtype, n_chunks=3, n_batches=20, B=500, n_val=2000, seed=41):
    rng = np.random.default_rng(seed)
    const = pam_constellation(4)
    L, S, l1, l2, lr = 15, 2, 25, 15, 5e-4
    nv0 = 10 ** (-16.0 / 10)
    model = mg.SecondVolterraV2VAE(channel_memory=L, equaliser_n_lags1=l1, equaliser_n_lags2=l2, learning_rate=lr,
                                   constellation=const, batch_size=B, samples_per_symbol=S, noise_variance=nv0,
                                   adaptive_noise_variance=True, dtype=dtype, lr_schedule=None)
    model.initialize_optimizer()
    out = dict(L=L, S=S, l1=l1, l2=l2, B=B, n_batches=n_batches, n_chunks=n_chunks, order=4, lr=lr, lr_schedule="none",
               linear_eq=0, constellation=const, noise_variance_in=nv0, n_val=n_val)
    fir1, alt = FIR1, FIR1_ALT
    for c in range(n_chunks):
        rx, _ = nonlinear_isi(B * n_batches, const, S, 16.0, rng, fir1=fir1)
        y_eq, loss = model.fit(rx)
        rx_val, syms_val = nonlinear_isi(n_val, const, S, 16.0, rng, fir1=fir1)
        q_val = model.estimate_symbol_probs(rx_val)
        ser, delay = mg.calc_ser_from_probs(q_val, syms_val, const)
        out.update({f"rx_{c}": rx, f"y_eq_{c}": y_eq, f"loss_{c}": loss, f"rx_val_{c}": rx_val, f"syms_val_{c}": syms_val,
                    f"q_val_{c}": q_val.astype(np.float32), f"ser_{c}": ser, f"ser_delay_{c}": delay,
                    f"noise_variance_{c}": float(model.noise_variance),
                    f"h1_{c}": model.channel_h1.detach().numpy().copy(), f"h2_{c}": model.channel_h2.detach().numpy().copy(),
                    f"k1_{c}": model.equaliser.kernel1.detach().numpy().copy(),
                    f"k2_{c}": model.equaliser.kernel2.detach().numpy().copy(),
                    f"noise_scaling_{c}": model.noise_scaling.detach().numpy().copy()})
        print(f"refit {tag} chunk {c} (fir1 {fir1}): loss {loss[0]:.3f} -> {loss[-1]:.3f}, ser {ser:.4e}")
        fir1, alt = alt, fir1  # change_wh(): the channel changes AFTER the chunk's validation
    st = model.optimizer.state_dict()["state"]
    out["adam_step"] = int(next(iter(st.values()))["step"])
    np.savez_compressed(os.path.join(HERE, f"v2vae_real_refit_cfg3_{tag}.npz"), **out)


if __name__ == "__main__":
    torch.manual_seed(0)
    torch.set_num_threads(4)
    refit_case("f64", torch.float64)
    refit_case("f32", torch.float32)
