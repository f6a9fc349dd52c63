"""Host logic of the batched sweep: the model dimension is partitioned over ranks without gaps or overlaps."""
import pytest


@pytest.mark.parametrize("n_models,world", [(1024, 8), (10, 3), (3, 8), (1, 1), (17, 4)])
def test_shard_models_partitions_the_models(n_models, world):
    from volterra_vae_amd.sweep import shard_models
    covered = []
    sizes = []
    for r in range(world):
        lo, hi = shard_models(r, world, n_models)
        assert 0 <= lo <= hi <= n_models
        covered += list(range(lo, hi))
        sizes.append(hi - lo)
    assert covered == list(range(n_models))
    assert max(sizes) - min(sizes) <= 1
