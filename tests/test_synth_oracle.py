"""CPU tests of the synthetic-channel oracle (oracle/synth_oracle.py) and of the host-side hash export of the library."""
import numpy as np


def test_hash_is_stable_and_matches_library_export():
    from oracle import synth_oracle as so
    from volterra_vae_amd._lib import lib
    assert int(so.bits(1, 0, np.array([5], dtype=np.uint64))[0]) == 0xB1441D254BD86CD7  # known answer, fixed by the definition
    idx = np.array([0, 1, 2, 77, 2 ** 33 + 11, 2 ** 63], dtype=np.uint64)
    for seed, kind in ((0, 0), (42, 1), (2 ** 60 + 7, 0)):
        ours = so.bits(seed, kind, idx)
        for i, v in zip(idx, ours):
            assert lib.v2vae_synth_bits(seed, kind, int(i)) == int(v)


def test_oracle_sequence_properties():
    from oracle import synth_oracle as so
    from volterra_vae_amd import synth
    levels = [-3.0, -1.0, 1.0, 3.0]
    y, idx = so.channel(9, levels, 2, 2, 40000, synth._lowpass_pulse(2), synth.FIR1, synth.FIR2, synth.POLY)
    assert y.shape == (2, 80000) and idx.shape == (2, 40000)
    counts = np.stack([np.bincount(idx[c], minlength=4) for c in range(2)]) / 40000
    assert np.abs(counts - 0.25).max() < 0.01
    assert abs(np.corrcoef(idx[0], idx[1])[0, 1]) < 0.02          # I and Q draws are independent
    g0, g1 = so.gaussian_pair(9, np.arange(200000, dtype=np.uint64))
    for g in (g0, g1):
        assert abs(g.mean()) < 0.01 and abs(g.std() - 1.0) < 0.01
    assert abs(np.mean(g0 * g1)) < 0.01
    # the chain is the host generator's: same symbols through synth._wh_chain give the same noiseless signal
    sym = np.asarray(levels)[idx[0]] + 1j * np.asarray(levels)[idx[1]]
    x = np.zeros(80000, dtype=complex)
    x[::2] = sym
    ref = synth._wh_chain(x, 2, synth.FIR1, synth.FIR2, synth.POLY)
    assert np.abs((y[0] + 1j * y[1]) - ref).max() < 1e-12
