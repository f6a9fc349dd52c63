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


def test_documented_size_limits_are_enforced_and_their_maxima_accepted():
    """Limits of the library (levels per axis <= 16, samples per symbol <= 8, l1 <= 255, l2 odd <= 129, one or two streams
    matching the model, non-negative padding inside the kernel): the largest accepted values pass, one beyond is refused with
    an error code instead of a crash; unknown flag bits and a non-zero reserved field are refused too."""
    from volterra_vae_amd import _lib

    def problem(**over):
        p = _lib.Problem()
        p.ns, p.mode, p.sps, p.l1, p.pad1, p.l2, p.L, p.C = 2, 1, 2, 25, 12, 25, 25, 4
        p.n_glob, p.y_lo, p.y_len, p.sym_lo, p.sym_cnt, p.own_lo, p.own_cnt = 8192, 0, 8192, 0, 4096, 0, 4096
        for k, v in over.items():
            setattr(p, k, v)
        return p

    def rc(**over):
        return _lib.lib.v2vae_check_problem(ctypes.byref(problem(**over)))

    assert rc() == _lib.OK
    assert rc(C=16) == _lib.OK and rc(C=17) == _lib.EINVAL and rc(C=0) == _lib.EINVAL
    assert rc(l1=255, pad1=127) == _lib.OK and rc(l1=256, pad1=127) == _lib.EINVAL
    assert rc(pad1=25) == _lib.EINVAL and rc(pad1=-1) == _lib.EINVAL
    assert rc(l2=129) in (_lib.OK, _lib.ESIZE) and rc(l2=131) == _lib.EINVAL and rc(l2=-1) == _lib.EINVAL
    assert rc(l2=0) == _lib.OK   # first-order (Conv1d) equaliser
    assert rc(sps=8, L=25, n_glob=8192, y_len=8192, sym_cnt=1024, own_cnt=1024) == _lib.OK
    assert rc(sps=9, L=25, n_glob=9216, y_len=9216, sym_cnt=1024, own_cnt=1024) == _lib.EINVAL
    assert rc(ns=1) == _lib.EINVAL and rc(ns=3) == _lib.EINVAL       # the complex model has two streams
    assert rc(ns=1, mode=0) == _lib.OK and rc(ns=2, mode=0) == _lib.EINVAL
    assert rc(mode=4) == _lib.EINVAL and rc(mode=-1) == _lib.EINVAL
    assert rc(flags=1 << 10) == _lib.EINVAL and rc(reserved=1) == _lib.EINVAL
    assert rc(n_glob=8191, y_len=8191) == _lib.EINVAL                 # not a whole number of symbols
    assert rc(sym_cnt=0, own_cnt=0) == _lib.EINVAL and rc(own_lo=1) == _lib.EINVAL
    assert _lib.lib.v2vae_check_problem(None) == _lib.EINVAL
    assert _lib.lib.v2vae_workspace_bytes(ctypes.byref(problem(C=17))) == 0
