"""Size-independent properties at the size of the headline benchmark (complex 16-QAM, L = l1 = l2 = 25, 2^22 symbols; far
beyond what the CPU oracle finishes) and, for the first, of the long-channel 64-QAM shape (L = l1 = l2 = 65): (1) E, the KL term and every gradient are sums over windows / symbols, so the two halo
shards of the batch, each evaluated by the kernels with non-zero range offsets, must add up to the whole batch; (2) the
matrix-core kernels against the generic FMA kernels (the ones the reference's goldens pin) on the same 2^22 symbols."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HEADLINE = (25, 2, 25, 25, 4, 22)      # L, S, l1, l2, levels per axis, log2(symbols)
LONG_CHANNEL = (65, 2, 65, 65, 8, 22)  # the 64-QAM shape of the 8-GPU configuration (blocked matrix-core kernels)


def rel(a, b):
    a = a.double().flatten()
    b = b.double().flatten()
    return float((a - b).norm() / b.norm().clamp_min(1e-300))


def _setup(shape, ns=2):
    from volterra_vae_amd.synth_device import nonlinear_isi_device
    L, S, l1, l2, M, LOG2B = shape
    B = LOG2B if LOG2B > 64 else 1 << LOG2B   # the last entry of a shape: log2 of the symbol count, or the count itself
    lv1 = np.arange(-(M - 1), M, 2)
    y = nonlinear_isi_device(4242, lv1, ns, S, 16.0, B, 0, B, device="cuda").contiguous()
    g = torch.Generator().manual_seed(11)
    n = [ns * L, ns * L * L, ns * l1, ns * l2 * l2, ns * M]
    o = np.cumsum([0] + n)
    params = (0.01 * torch.randn(sum(n), generator=g)).cuda()
    params[L // 2] += 1.0
    params[o[2] + l1 // 2] += 1.0
    params[o[4]:] = 1.0 + 0.1 * torch.rand(n[4], generator=g).cuda()
    lv = torch.tensor(lv1, dtype=torch.float32, device="cuda")
    levels = torch.stack([lv] * ns).contiguous()
    prior = torch.full_like(levels, 1.0 / M)
    nv = torch.tensor([0.04], device="cuda")
    return B, y, params, o, levels, prior, nv


def _forward(plan, y, params, o, levels, prior, nv):
    h1, h2, k1, k2, sc = (params[o[i]:o[i + 1]] for i in range(5))
    xhat = torch.empty((plan.p.ns, plan.p.sym_cnt), dtype=torch.float32, device="cuda")
    plan.prep_tables(h1, h2, k1, k2)
    plan.eq_fwd(y, xhat)
    plan.demap_moments(xhat, levels, prior, sc, nv)
    plan.elbo_fused(y)
    plan.reduce_scalars()
    return xhat


def _backward(plan, y, xhat, params, o, levels, prior, nv):
    sc = params[o[4]:o[5]]
    plan.finalize_scalars()
    gx = plan.sym_bwd(xhat, levels, prior, sc, nv)
    plan.eq_bwd(y, gx)
    g = torch.zeros_like(params)
    plan.finish_grads(7, g[o[0]:o[1]], g[o[1]:o[2]], g[o[2]:o[3]], g[o[3]:o[4]], g[o[4]:o[5]])
    torch.cuda.synchronize()
    return g


@pytest.mark.parametrize("shape,world", [(HEADLINE, 2), (LONG_CHANNEL, 2), ((25, 2, 25, 25, 4, 100003), 3),
                                         ((47, 1, 31, 31, 8, 50021), 5), ((25, 2, 25, 25, 4, 1 << 20), 8)])
def test_halo_shards_of_a_full_size_batch_add_up_to_the_whole_batch(shape, world):
    from volterra_vae_amd import _lib
    from volterra_vae_amd.distributed import shard_ranges
    from volterra_vae_amd.ops import Plan
    L, S, l1, l2, M, LOG2B = shape
    B, y, params, o, levels, prior, nv = _setup(shape)
    kw = dict(ns=2, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, n_glob=B * S, device="cuda")
    whole = Plan(**kw)
    xw = _forward(whole, y, params, o, levels, prior, nv)
    torch.cuda.synchronize()
    e_whole, kl_whole = float(whole.scal[_lib.SCAL_E]), float(whole.scal[_lib.SCAL_KL])
    gw = _backward(whole, y, xw, params, o, levels, prior, nv)
    loss_whole = float(whole.scal[_lib.SCAL_LOSS])

    shards = []
    for rank in range(world):   # uneven shares when the symbol count is not a multiple of the number of ranks
        sr = shard_ranges(rank, world, B * S, S, L, l1, l1 // 2, l2)
        plan = Plan(y_lo=sr.y_lo, y_len=sr.y_len, sym_lo=sr.sym_lo, sym_cnt=sr.sym_cnt, own_lo=sr.own_lo, own_cnt=sr.own_cnt, **kw)
        ys = y[:, sr.y_lo:sr.y_lo + sr.y_len].contiguous()
        xs = _forward(plan, ys, params, o, levels, prior, nv)
        shards.append((sr, plan, ys, xs))
    torch.cuda.synchronize()
    e_sum = sum(float(p.scal[_lib.SCAL_E]) for _, p, _, _ in shards)
    kl_sum = sum(float(p.scal[_lib.SCAL_KL]) for _, p, _, _ in shards)
    assert abs(e_sum - e_whole) < 1e-9 * abs(e_whole), (e_sum, e_whole)
    assert abs(kl_sum - kl_whole) < 1e-9 * abs(kl_whole), (kl_sum, kl_whole)
    # what the all-reduce of the two scalars does, then the backward of each shard with the global sums
    g_sum = torch.zeros_like(params)
    for sr, plan, ys, xs in shards:
        plan.scal[_lib.SCAL_E] = e_sum
        plan.scal[_lib.SCAL_KL] = kl_sum
        g_sum += _backward(plan, ys, xs, params, o, levels, prior, nv)
        assert abs(float(plan.scal[_lib.SCAL_LOSS]) - loss_whole) < 1e-9 * abs(loss_whole)
        a = sr.own_lo - sr.sym_lo
        assert torch.equal(xs[:, a:a + sr.own_cnt], xw[:, sr.own_lo:sr.own_lo + sr.own_cnt])   # same samples, same arithmetic
    for name, i in (("h1", 0), ("h2", 1), ("k1", 2), ("k2", 3), ("noise scaling", 4)):
        e = rel(g_sum[o[i]:o[i + 1]], gw[o[i]:o[i + 1]])
        print(f"{B} symbols, gradient of {name}: {world} shards against the whole batch {e:.2e}")
        assert e < 2e-6, (name, e)   # float32 partial sums in another grouping


@pytest.mark.parametrize("ns", [2, 1])
def test_matrix_core_kernels_agree_with_the_generic_kernels_at_the_headline_size(ns):
    """ns = 1: the real PAM-4 model on the same kernels (zero quadrature stream), against the generic real kernels."""
    from volterra_vae_amd import _lib
    from volterra_vae_amd.ops import Plan
    L, S, l1, l2, M, LOG2B = HEADLINE
    B, y, params, o, levels, prior, nv = _setup(HEADLINE, ns)
    res = []
    for flags in (0, _lib.FLAG_GENERIC):
        plan = Plan(ns=ns, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, n_glob=B * S, device="cuda", flags=flags)
        xh = _forward(plan, y, params, o, levels, prior, nv)
        g = _backward(plan, y, xh, params, o, levels, prior, nv)
        res.append((float(plan.scal[_lib.SCAL_E]), float(plan.scal[_lib.SCAL_KL]), float(plan.scal[_lib.SCAL_LOSS]), xh.clone(), g))
        del plan
    (e0, k0, l0, x0, g0), (e1, k1_, l1_, x1, g1) = res
    assert abs(e0 - e1) < 2e-6 * abs(e1) and abs(k0 - k1_) < 2e-6 * abs(k1_) and abs(l0 - l1_) < 2e-6 * abs(l1_)
    assert rel(x0, x1) < 2e-6
    for name, i in (("h1", 0), ("h2", 1), ("k1", 2), ("k2", 3), ("noise scaling", 4)):
        e = rel(g0[o[i]:o[i + 1]], g1[o[i]:o[i + 1]])
        print(f"2^{LOG2B} symbols, gradient of {name}: matrix-core against generic kernels {e:.2e}")
        assert e < 1e-5, (name, e)


# further kernel instances at 2^18 symbols: 17 taps per phase (blocked ELBO kernel, one block and a tap on the side, two waves per
# SIMD), one sample per symbol with 49 lags (three lag blocks and a lag on the side), four samples per symbol, 33 / 81 lags
# (two blocks and a lag on the side / six blocks), 47-tap channel with three tap blocks
OTHER_INSTANCES = [(33, 2, 33, 33, 4, 18), (25, 1, 49, 49, 4, 18), (9, 4, 25, 25, 4, 18), (65, 2, 81, 81, 4, 17), (95, 2, 17, 17, 4, 17),
                   (41, 1, 65, 65, 8, 17)]


@pytest.mark.parametrize("shape", [HEADLINE, LONG_CHANNEL] + OTHER_INSTANCES)
def test_full_size_step_is_bitwise_reproducible(shape):
    """Every sum of the path has a fixed order (per-workgroup records, two-stage record sums, no floating-point atomics): the
    same inputs give the same bits, on a fresh workspace and on one that has been used."""
    from volterra_vae_amd import _lib
    from volterra_vae_amd.ops import Plan
    L, S, l1, l2, M, LOG2B = shape
    B, y, params, o, levels, prior, nv = _setup(shape)
    kw = dict(ns=2, sps=S, l1=l1, pad1=l1 // 2, l2=l2, L=L, C_levels=M, n_glob=B * S, device="cuda")
    runs = []
    plan = Plan(**kw)
    for rep in range(3):
        if rep == 2:
            plan = Plan(**kw)   # fresh workspace
        xh = _forward(plan, y, params, o, levels, prior, nv)
        g = _backward(plan, y, xh, params, o, levels, prior, nv)
        runs.append((plan.scal.clone(), xh.clone(), g.clone()))
    for sc, xh, g in runs[1:]:
        assert torch.equal(sc, runs[0][0]) and torch.equal(xh, runs[0][1]) and torch.equal(g, runs[0][2])
