"""The C-ABI library loads and exports every symbol include/v2vae_hip.h declares (no GPU needed)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "v2vae_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(v2vae_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_something():
    syms = declared_symbols()
    assert "v2vae_train_step" in syms and "v2vae_elbo_fused" in syms and len(syms) >= 15


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as ge
    ge.build()
    lib = ctypes.CDLL(os.path.join(ROOT, "volterra_vae_amd", "libv2vae_hip.so"))
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in the header but not exported: {missing}"


def test_binding_table_matches_header():
    from volterra_vae_amd import _lib
    assert sorted(_lib.EXPORTS) == declared_symbols()


def test_problem_validation_without_gpu():
    """Host-side geometry checks run without a device."""
    from volterra_vae_amd import _lib
    p = _lib.Problem()
    p.ns, p.mode, p.sps, p.l1, p.pad1, p.l2, p.L, p.C = 1, 0, 2, 25, 12, 25, 15, 4
    p.n_glob, p.y_lo, p.y_len, p.sym_lo, p.sym_cnt, p.own_lo, p.own_cnt = 1000, 0, 1000, 0, 500, 0, 500
    assert _lib.lib.v2vae_check_problem(ctypes.byref(p)) == _lib.OK
    assert _lib.lib.v2vae_workspace_bytes(ctypes.byref(p)) > 0
    assert _lib.pack_sizes(p) == [15, 225, 25, 625, 4]
    p.l2 = 24  # even second-order lag count (reference asserts odd, lib/torch_components.py:66)
    assert _lib.lib.v2vae_check_problem(ctypes.byref(p)) == _lib.EINVAL
    p.l2 = 25
    p.sps, p.L = 3, 2  # samples per symbol above channel memory (lib/real_equalization.py:751)
    p.n_glob = p.y_len = 999
    p.sym_cnt = p.own_cnt = 333
    assert _lib.lib.v2vae_check_problem(ctypes.byref(p)) == _lib.EINVAL
    # a shard whose local samples do not cover its halo is refused
    p.sps, p.L, p.n_glob, p.y_len, p.sym_cnt, p.own_cnt = 2, 15, 4000, 1000, 500, 500
    p.y_lo, p.sym_lo, p.own_lo = 1000, 500, 500
    assert _lib.lib.v2vae_check_problem(ctypes.byref(p)) == _lib.EHALO
