"""ctypes binding of libv2vae_hip.so (the C ABI declared in include/v2vae_hip.h).

The shared library is built in-tree by `__graft_entry__.build()` / `make -C volterra_vae_amd/csrc`.
There is no CPU fallback: if the library is missing, importing this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

# torch must be imported first: it brings its own HIP runtime (libamdhip64), and the kernels have to run
# on that same runtime instance to share torch's device context, streams and allocations.  Loading
# libv2vae_hip.so first would bind the system runtime instead and the two would not see each other.
import torch  # noqa: F401

_HERE = os.path.dirname(os.path.abspath(__file__))
# V2VAE_LIB_PATH: diagnostic builds of the same library (tools/), never a different backend
LIB_PATH = os.environ.get("V2VAE_LIB_PATH") or os.path.join(_HERE, "libv2vae_hip.so")

OK, EINVAL, EHALO, ELAUNCH, ESIZE = 0, -1, -2, -3, -4
_ERR = {EINVAL: "invalid argument / unsupported shape", EHALO: "local sample or symbol range does not cover the halo",
        ELAUNCH: "HIP launch error", ESIZE: "problem exceeds the kernels' on-chip tiling limits"}

FLAG_GENERIC, FLAG_EQ_MFMA_FWD, FLAG_ELBO_TILES, FLAG_REAL_MFMA, FLAG_F32_MATRIX = 1, 2, 4, 8, 16


def default_flags() -> int:
    """Kernel A/B switches for measurements and tooling, taken from the environment when a problem is set up
    (V2VAE_FORCE_GENERIC: generic FMA kernels everywhere; V2VAE_EQ_MFMA_FWD: matrix-core forward equaliser; V2VAE_ELBO_TILES:
    block-tile instead of run matrix-core ELBO kernel; V2VAE_REAL_MFMA: the real model on the matrix-core ELBO kernel at every batch
    size, not only from 2^12 symbols on; V2VAE_F32_MATRIX: the f32 matrix / vector-pipe routes instead of the bf16 matrix instructions
    in split arithmetic).  The
    library itself never reads the environment: the choice travels in `Problem.flags`."""
    return (FLAG_GENERIC if os.environ.get("V2VAE_FORCE_GENERIC") else 0) | \
        (FLAG_EQ_MFMA_FWD if os.environ.get("V2VAE_EQ_MFMA_FWD") else 0) | \
        (FLAG_ELBO_TILES if os.environ.get("V2VAE_ELBO_TILES") else 0) | \
        (FLAG_REAL_MFMA if os.environ.get("V2VAE_REAL_MFMA") else 0) | \
        (FLAG_F32_MATRIX if os.environ.get("V2VAE_F32_MATRIX") else 0)


SCAL_E, SCAL_KL, SCAL_LOSS, SCAL_LAMBDA, SCAL_W, SCAL_BAD_STEPS, SCAL_COUNT = 0, 1, 2, 3, 4, 5, 8


class Problem(C.Structure):
    _fields_ = [("ns", C.c_int32), ("mode", C.c_int32), ("sps", C.c_int32), ("l1", C.c_int32),
                ("pad1", C.c_int32), ("l2", C.c_int32), ("L", C.c_int32), ("C", C.c_int32),
                ("flags", C.c_int32), ("reserved", C.c_int32),
                ("n_glob", C.c_int64), ("y_lo", C.c_int64), ("y_len", C.c_int64),
                ("sym_lo", C.c_int64), ("sym_cnt", C.c_int64), ("own_lo", C.c_int64), ("own_cnt", C.c_int64)]

    def key(self):
        return tuple(getattr(self, f) for f, _ in self._fields_)


class TrainArgs(C.Structure):
    _fields_ = [("y", C.c_void_p), ("levels", C.c_void_p), ("prior", C.c_void_p), ("params", C.c_void_p),
                ("grads", C.c_void_p), ("adam_m", C.c_void_p), ("adam_v", C.c_void_p), ("adam_vmax", C.c_void_p),
                ("noise_var", C.c_void_p), ("ws", C.c_void_p), ("scal", C.c_void_p), ("xhat", C.c_void_p),
                ("gxhat", C.c_void_p), ("loss_out", C.c_void_p),
                ("lr", C.c_float), ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("step", C.c_int64), ("learn_scaling", C.c_int32), ("adaptive_noise", C.c_int32),
                ("freeze_h2", C.c_int32), ("accumulate_scaling_grad", C.c_int32)]


class FitArgs(C.Structure):
    _fields_ = [("y", C.c_void_p), ("y_model_stride", C.c_int64), ("y_stream_stride", C.c_int64),
                ("levels", C.c_void_p), ("prior", C.c_void_p), ("params", C.c_void_p), ("grads", C.c_void_p),
                ("adam_m", C.c_void_p), ("adam_v", C.c_void_p), ("adam_vmax", C.c_void_p), ("noise_var", C.c_void_p),
                ("ws", C.c_void_p), ("scal", C.c_void_p), ("xhat", C.c_void_p), ("gxhat", C.c_void_p),
                ("y_eq", C.c_void_p), ("loss_curve", C.c_void_p), ("step_size", C.c_void_p), ("bc2_sqrt", C.c_void_p),
                ("beta1", C.c_float), ("beta2", C.c_float), ("eps", C.c_float),
                ("n_models", C.c_int32), ("n_batches", C.c_int32), ("learn_scaling", C.c_int32),
                ("adaptive_noise", C.c_int32), ("freeze_h2", C.c_int32), ("accumulate_scaling_grad", C.c_int32)]


class SynthArgs(C.Structure):
    _fields_ = [("ns", C.c_int32), ("sps", C.c_int32), ("C", C.c_int32), ("n_pulse", C.c_int32), ("n_fir1", C.c_int32),
                ("n_fir2", C.c_int32), ("levels", C.c_float * 16), ("pulse", C.c_float * 65), ("fir1", C.c_float * 9),
                ("fir2", C.c_float * 9), ("poly1", C.c_float), ("poly2", C.c_float), ("mean", C.c_float * 2),
                ("inv_std", C.c_float), ("noise_std", C.c_float), ("seed", C.c_uint64), ("b_glob", C.c_int64),
                ("sym_lo", C.c_int64), ("sym_cnt", C.c_int64)]


EXPORTS = {
    # name: (restype, argtypes)
    "v2vae_version": (C.c_char_p, []),
    "v2vae_last_error": (C.c_char_p, []),
    "v2vae_workspace_bytes": (C.c_size_t, [C.POINTER(Problem)]),
    "v2vae_check_problem": (C.c_int, [C.POINTER(Problem)]),
    "v2vae_prep_tables": (C.c_int, [C.POINTER(Problem)] + [C.c_void_p] * 6),
    "v2vae_eq_fwd": (C.c_int, [C.POINTER(Problem)] + [C.c_void_p] * 4),
    "v2vae_demap_moments": (C.c_int, [C.POINTER(Problem)] + [C.c_void_p] * 8),
    "v2vae_elbo_prepass": (C.c_int, [C.POINTER(Problem)] + [C.c_void_p] * 2),
    "v2vae_elbo_fused": (C.c_int, [C.POINTER(Problem)] + [C.c_void_p] * 3),
    "v2vae_reduce_scalars": (C.c_int, [C.POINTER(Problem)] + [C.c_void_p] * 3),
    "v2vae_finalize_scalars": (C.c_int, [C.POINTER(Problem)] + [C.c_void_p] * 2),
    "v2vae_sym_bwd": (C.c_int, [C.POINTER(Problem)] + [C.c_void_p] * 10),
    "v2vae_eq_bwd": (C.c_int, [C.POINTER(Problem)] + [C.c_void_p] * 4),
    "v2vae_finish_grads": (C.c_int, [C.POINTER(Problem), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 6),
    "v2vae_adam_amsgrad": (C.c_int, [C.c_int64] + [C.c_void_p] * 5 + [C.c_float] * 4 + [C.c_int64] + [C.c_void_p] * 4),
    "v2vae_synth_channel": (C.c_int, [C.POINTER(SynthArgs), C.c_void_p, C.c_void_p, C.c_void_p]),
    "v2vae_synth_bits": (C.c_uint64, [C.c_uint64, C.c_uint64, C.c_uint64]),
    "v2vae_decide": (C.c_int, [C.POINTER(Problem)] + [C.c_void_p] * 6),
    "v2vae_pack_sizes": (None, [C.POINTER(Problem), C.POINTER(C.c_int64 * 5)]),
    "v2vae_train_step": (C.c_int, [C.POINTER(Problem), C.POINTER(TrainArgs), C.c_void_p]),
    "v2vae_train_step_sharded_pre": (C.c_int, [C.POINTER(Problem), C.POINTER(TrainArgs), C.c_void_p]),
    "v2vae_train_step_sharded_post": (C.c_int, [C.POINTER(Problem), C.POINTER(TrainArgs), C.c_void_p]),
    "v2vae_train_step_update": (C.c_int, [C.POINTER(Problem), C.POINTER(TrainArgs), C.c_void_p]),
    "v2vae_train_step_static": (C.c_int, [C.POINTER(Problem), C.POINTER(TrainArgs), C.c_void_p]),
    "v2vae_static_scratch_bytes": (C.c_size_t, [C.POINTER(Problem)]),
    "v2vae_adam_schedule": (C.c_int, [C.c_int64, C.c_int64, C.c_void_p, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "v2vae_fit_persistent": (C.c_int, [C.POINTER(Problem), C.POINTER(FitArgs), C.c_void_p]),
    "v2vae_fit_persistent_kind": (C.c_int, [C.POINTER(Problem)]),
    "v2vae_ws_offset_moments": (C.c_int64, [C.POINTER(Problem)]),
    "v2vae_ws_offset_gmoments": (C.c_int64, [C.POINTER(Problem)]),
    "v2vae_ws_offset_xhat_float": (C.c_int64, [C.POINTER(Problem)]),
    "v2vae_ws_offset_xhat_remainder": (C.c_int64, [C.POINTER(Problem)]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build the HIP library first (python -c 'import __graft_entry__ as g; g.build()' "
            "or make -C volterra_vae_amd/csrc).  There is no CPU fallback for the V2VAE hot path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


class V2VAEError(RuntimeError):
    pass


def check(rc: int, what: str = ""):
    if rc != OK:
        detail = lib.v2vae_last_error().decode() if rc == ELAUNCH or (rc == ESIZE and "static" in what) else ""
        raise V2VAEError(f"{what}: {_ERR.get(rc, 'error')} (code {rc}) {detail}")


def pack_sizes(p: Problem):
    out = (C.c_int64 * 5)()
    lib.v2vae_pack_sizes(C.byref(p), C.byref(out))
    return [int(v) for v in out]
