"""GPU parity of the general kernel (any band width) against the oracle, through the C-ABI.

Bar: forward cells, backward cells, totalProbability values and the posterior exponents are
bit-identical to the oracle's doubles; integer posteriors may differ by 1 unit of 1e-7 (device exp
vs host exp).  Known answers of the reference (8 toy pairs, 986 pairs on the Zymo read) are checked
on the GPU result directly."""
import numpy as np
import pytest

import pyoracle as o
import synth
from harness import (assert_same_pairs, band_params, cp, make_items, orc_params, run_gpu,
                     run_oracle_item)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = cp.Context(0)
    yield c
    c.close()


def _single(x, events, match, gapy, anchors=()):
    ev = np.asarray(events, dtype=np.float64).reshape(-1, 3)
    lX = len(x) - 5
    return dict(x_chars=x.encode(), events=ev, anchors=np.array(anchors, np.int64).reshape(-1, 2),
                items=[dict(x_offset=0, lX=lX, y_offset=0, lY=len(ev), anchor_offset=0,
                            n_anchors=len(anchors), model=0)],
                models=[(match, np.full(4096, -2.3025850929940455), gapy)])


def test_toy_strawman_known_answer(ctx, template_model):
    # tests/signalPairwiseTest.c:580-685
    match, _, gapy = template_model
    sY = [58.743435, 0.887833, 0.0571, 53.604965, 0.816836, 0.0571, 58.432015, 0.735143, 0.0571,
          63.684352, 0.795437, 0.0571, 58.921430, 0.812959, 0.0571, 59.895882, 0.740952, 0.0571,
          61.684303, 0.722332, 0.0571]
    batch = _single("ACGATACGGACAT", sY, match, gapy)
    bp = band_params(threshold=0.2)
    res, b = run_gpu(ctx, batch, bp, kernel=cp.KERNEL_GENERAL, flags=cp.FLAG_UNBANDED)
    pairs = sorted((int(x), int(y)) for _, x, y in res[0]["triples"])
    assert pairs == [(0, 0), (1, 1), (2, 2), (3, 3), (4, 3), (5, 4), (6, 5), (7, 6)]
    ref = o.aligned_pairs_without_banding(o.Sm3Model(match, gapy), "ACGATACGGACAT", 8, sY,
                                          o.default_params(threshold=0.2))
    assert res[0]["totals"][0] == ref["totals"][0]
    got = {(int(x), int(y)): lp for (_, x, y), lp in zip(res[0]["triples"], res[0]["logp"])}
    exp = {(int(x), int(y)): lp for (_, x, y), lp in zip(ref["triples"], ref["logp"])}
    assert got == exp


def test_real_read_unbanded_986(ctx, template_model, zymo_read):
    # tests/signalPairwiseTest.c:1166-1173: exactly 986 aligned pairs
    match, _, gapy = template_model
    m = synth.scale_model(match, *zymo_read["template_params"])
    ms = o.Sm3Model(match, gapy).scaled(*zymo_read["template_params"])
    m = ms.match  # scaled with libm pow exactly as emissions_signal_scaleModel
    ref = zymo_read["reference"]
    batch = _single(ref, zymo_read["template_events"], m, gapy)
    res, b = run_gpu(ctx, batch, band_params(), kernel=cp.KERNEL_GENERAL, flags=cp.FLAG_UNBANDED)
    tri = res[0]["triples"]
    assert len(tri) == 986
    orc = o.aligned_pairs_without_banding(ms, ref, len(ref) - 5, zymo_read["template_events"],
                                          o.default_params())
    assert res[0]["totals"][0] == orc["totals"][0]
    got = {(int(x), int(y)): (int(p), lp) for (p, x, y), lp in zip(tri, res[0]["logp"])}
    for (p, x, y), lp in zip(orc["triples"], orc["logp"]):
        gp, glp = got[(int(x), int(y))]
        assert glp == lp and abs(gp - int(p)) <= 1
    assert res[0]["cells"] == orc["cells"]


@pytest.mark.parametrize("case", [
    dict(lX=120, lY=250, e=20, md=60, tb=10, every=25, ragged=(0, 0)),
    dict(lX=300, lY=610, e=40, md=100, tb=40, every=50, ragged=(1, 1)),
    dict(lX=257, lY=400, e=100, md=150, tb=40, every=50, ragged=(1, 0)),
    dict(lX=90, lY=200, e=0, md=30, tb=5, every=10, ragged=(0, 1)),
])
def test_banded_cells_bit_exact(ctx, case):
    batch = synth.make_batch(11, 3, case["lX"], case["lY"], anchor_every=case["every"])
    bp = band_params(0.01, case["md"], case["tb"], case["e"])
    res, b = run_gpu(ctx, batch, bp, kernel=cp.KERNEL_GENERAL, flags=cp.FLAG_DEBUG_DUMP,
                     ragged=case["ragged"])
    for i in range(3):
        ref = run_oracle_item(batch, i, bp, case["ragged"], dump=True)
        n = ref["F"].shape[0]
        assert res[i]["cells"] == n
        F, B = b.debug_cells(i, n)
        assert np.array_equal(F, ref["F"]), "forward cells differ"
        ok = ~np.isnan(ref["B"][:, 0])  # diagonal 0 gets no posterior pass
        assert np.array_equal(B[ok], ref["B"][ok]), "backward cells differ"
        assert np.array_equal(res[i]["totals_xay"], ref["totals_xay"])
        assert np.array_equal(res[i]["totals"], ref["totals"])
        ref["triples"] = ref["triples"]
        assert_same_pairs(res[i], ref)


def test_empty_and_tiny_items(ctx):
    batch = synth.make_batch(12, 2, 40, 80, anchor_every=10)
    # append degenerate items: empty X, empty Y, both empty
    base = batch["items"][0]
    batch["items"] += [dict(base, lX=0, n_anchors=0), dict(base, lY=0, n_anchors=0),
                       dict(base, lX=0, lY=0, n_anchors=0)]
    bp = band_params(0.01, 20, 5, 10)
    res, b = run_gpu(ctx, batch, bp, kernel=cp.KERNEL_GENERAL)
    for i in range(len(batch["items"])):
        ref = run_oracle_item(batch, i, bp)
        assert_same_pairs(res[i], ref)
        assert np.array_equal(res[i]["totals"], ref["totals"])


def test_expectations_match_oracle(ctx):
    batch = synth.make_batch(13, 4, 150, 310, anchor_every=25, distinct_models=False)
    bp = band_params(0.01, 80, 20, 40)
    res, b = run_gpu(ctx, batch, bp, mode=cp.MODE_EXPECTATIONS, kernel=cp.KERNEL_GENERAL, ragged=(1, 1))
    got = b.expectations(0)
    hmm = o.OrcExpectations()
    for i in range(4):
        run_oracle_item(batch, i, bp, (1, 1), expectations=hmm)
    exp_t = np.array(hmm.transitions[:])
    exp_k = np.array(hmm.kmerGap[:])
    # tolerance: the device sums in a different order and uses its own exp(): 1e-9 relative
    assert np.allclose(got[:9], exp_t, rtol=1e-9, atol=1e-12)
    assert np.allclose(got[9:9 + 4096], exp_k, rtol=1e-9, atol=1e-12)
    assert np.isclose(got[-1], hmm.likelihood, rtol=1e-12)
