import pytest


def test_partition_covers_the_set_in_equal_slots(nb):
    for n in (1, 2, 7, 8, 9, 1000, 131072, 1048576 + 3):
        for world in (1, 2, 3, 4, 8):
            parts = nb.partition(n, world)
            assert len(parts) == world
            slot = -(-n // world)
            covered = 0
            for r, (first, count) in enumerate(parts):
                assert count <= slot and (count == 0 or first == r * slot)
                assert first == covered or count == 0
                covered += count
            assert covered == n


def test_partition_baseline_shapes(nb):
    # BASELINE config 4: N=131 072 on 8 GPUs -> 16 384 bodies, 256 KiB of position records per rank
    assert nb.partition(131072, 8) == [(i * 16384, 16384) for i in range(8)]
    assert nb.partition(5, 8)[5:] == [(5, 0)] * 3


def test_partition_rejects_nonsense(nb):
    with pytest.raises(ValueError):
        nb.partition(0, 2)
    with pytest.raises(ValueError):
        nb.partition(4, 0)
