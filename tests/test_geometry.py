"""Host integer geometry of the product (cpecan_band_construct / cpecan_split_points, C-ABI) against
the reference's golden vectors and, on random anchors, against the oracle.  Bit-exact."""
import numpy as np
import pytest

import pyoracle as o
from cpecan_load import binding

cp = binding()


def test_band_golden_test_bands():
    # tests/pairwiseAlignerTest.c:74-99
    L, R = cp.band_construct([(1, 0), (2, 1), (3, 3)], 6, 5, 2)
    assert list(zip(L.tolist(), R.tolist())) == [
        (0, 0), (-1, 1), (-2, 2), (-1, 3), (-2, 4), (-1, 3), (-2, 4), (-3, 3), (-2, 2), (-1, 3),
        (0, 2), (1, 1)]


def _random_anchors(rng, lX, lY, step):
    out, x, y = [], -1, -1
    while True:
        x += int(rng.integers(1, step))
        y += int(rng.integers(1, step))
        if x >= lX or y >= lY:
            return out
        out.append((x, y))


@pytest.mark.parametrize("seed", range(40))
def test_band_matches_oracle_random(seed):
    rng = np.random.default_rng(seed)
    lX, lY = int(rng.integers(0, 400)), int(rng.integers(0, 400))
    e = int(rng.integers(0, 30)) * 2
    anchors = _random_anchors(rng, lX, lY, int(rng.integers(2, 60))) if seed % 5 else []
    L, R = cp.band_construct(anchors, lX, lY, e)
    L2, R2 = o.band(anchors, lX, lY, e)
    assert np.array_equal(L, L2) and np.array_equal(R, R2)


def test_band_edge_cases():
    L, R = cp.band_construct([], 0, 0, 20)
    assert L.tolist() == [0] and R.tolist() == [0]
    L, R = cp.band_construct([], 0, 7, 2)   # empty X: a single column
    L2, R2 = o.band([], 0, 7, 2)
    assert np.array_equal(L, L2) and np.array_equal(R, R2)
    L, R = cp.band_construct([(4, 4)], 5, 5, 0)  # anchor on the last cell, zero expansion
    L2, R2 = o.band([(4, 4)], 5, 5, 0)
    assert np.array_equal(L, L2) and np.array_equal(R, R2)
    with pytest.raises(cp.CpecanError) as ei:
        cp.band_construct([(3, 3), (2, 5)], 10, 10, 4)  # anchors must increase (asserts :164-169)
    assert ei.value.code == cp.EBAND
    with pytest.raises(cp.CpecanError):
        cp.band_construct([], 5, 5, 3)  # odd expansion (assert :136)


def test_split_points_golden():
    # tests/pairwiseAlignerTest.c:596-665
    ms = 2000 * 2000
    assert cp.split_points([], 3000, 1000, ms, 0, 0).tolist() == [[0, 0, 3000, 1000]]
    lX, lY = 20000, 25000
    assert cp.split_points([], lX, lY, ms, 1, 1).tolist() == []
    assert cp.split_points([], lX, lY, ms, 1, 0).tolist() == [[18000, 23000, lX, lY]]
    assert cp.split_points([], lX, lY, ms, 0, 1).tolist() == [[0, 0, 2000, 2000]]
    assert cp.split_points([], lX, lY, ms, 0, 0).tolist() == [[0, 0, 2000, 2000], [18000, 23000, lX, lY]]
    anchors = [(2000, 2000), (4002, 4001), (5000, 5000), (8000, 6000), (9000, 9000),
               (10000, 14000), (15000, 15000), (16000, 16000)]
    assert cp.split_points(anchors, lX, lY, ms, 0, 0).tolist() == [
        [0, 0, 3001, 3001], [3002, 3001, 9500, 11001], [9501, 12000, 12001, 14500],
        [13000, 14501, 18000, 18001], [18001, 23000, 20000, 25000]]


@pytest.mark.parametrize("seed", range(20))
def test_split_points_match_oracle_random(seed):
    rng = np.random.default_rng(100 + seed)
    lX, lY = int(rng.integers(100, 30000)), int(rng.integers(100, 30000))
    anchors = _random_anchors(rng, lX, lY, int(rng.integers(50, 6000)))
    for rl in (0, 1):
        for rr in (0, 1):
            a = cp.split_points(anchors, lX, lY, 3000 * 3000, rl, rr)
            b = o.split_points(anchors, lX, lY, 3000 * 3000, rl, rr)
            assert a.tolist() == b.tolist()
