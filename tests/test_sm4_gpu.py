"""GPU parity of the 4-state signal machine (getStateMachine4, impl/stateMachine.c:867-897, :960-1037, :1750-1759)
against the oracle, through the C-ABI (cpecan_hip_models4_create / cpecan_hip_batch_create_sm4 -> cpecan_k_general4).

The oracle's 4-state machine is pinned by the reference's own known answers in test_oracle_golden.py: the 8 toy pairs
(tests/signalPairwiseTest.c:687-787) and 988 aligned pairs on the shipped read (:1237).  Bar: every totalProbability
refresh and posterior exponent bit-identical, pairs in the reference's emission order, integer posteriors identical
after the host's libm pass."""
import numpy as np
import pytest

import pyoracle as o
import synth
from harness import assert_same_pairs, band_params, cp, make_items, orc_params
from test_oracle_golden import TOY4_PAIRS, TOY4_X, TOY_EVENTS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = cp.Context(0)
    yield c
    c.close()


def run(ctx, batch, models, bp, ragged, unbanded=False):
    """batch as made by synth.make_batch; models: one pyoracle.Sm4Model per batch model"""
    ctx.models_clear()
    ctx.models4_create([(m.transitions, m.match, m.gap_x, m.gap_y) for m in models])
    b = cp.Batch(ctx, make_items(batch, ragged), batch["x_chars"], batch["events"], batch["anchors"], bp,
                 flags=cp.FLAG_UNBANDED if unbanded else 0, sm4=True)
    assert b.info()["kernel"] == "general"
    b.run()
    b.sync()
    npairs, ntot, ncells = b.counts()
    p = orc_params(bp, split=1 << 60)
    out = []
    for i, it in enumerate(batch["items"]):
        x = batch["x_chars"][it["x_offset"]: it["x_offset"] + it["lX"] + 5]
        ev = batch["events"][it["y_offset"]: it["y_offset"] + it["lY"]]
        an = batch["anchors"][it["anchor_offset"]: it["anchor_offset"] + it["n_anchors"]]
        tri, lp = b.pairs(i, npairs[i])
        xay, tot = b.totals(i, ntot[i])
        if unbanded:
            ref = o.aligned_pairs_without_banding(models[it["model"]], x, it["lX"], ev, p, ragged[0], ragged[1])
            order = np.lexsort((ref["triples"][:, 1], -(ref["triples"][:, 1] + ref["triples"][:, 2])))
            ref["triples"], ref["logp"] = ref["triples"][order], ref["logp"][order]
        else:
            ref = o.aligned_pairs_using_anchors(models[it["model"]], x, it["lX"], ev, an, p, ragged[0], ragged[1])
            ref["triples"], ref["logp"] = ref["triples"][::-1], ref["logp"][::-1]
        assert np.array_equal(xay, ref["totals_xay"])
        assert np.array_equal(tot, ref["totals"])
        assert_same_pairs(dict(triples=tri, logp=lp), ref)
        out.append(tri)
    b.close()
    return out


def models_of(batch):
    return [o.Sm4Model(m, gy) for (m, _, gy) in batch["models"]]


@pytest.mark.parametrize("case", [
    dict(n=4, lX=120, lY=250, e=20, md=60, tb=10, every=25, ragged=(0, 0)),
    dict(n=3, lX=300, lY=610, e=40, md=100, tb=40, every=50, ragged=(1, 1)),
    dict(n=2, lX=257, lY=400, e=100, md=150, tb=40, every=50, ragged=(1, 0)),
    dict(n=2, lX=90, lY=200, e=0, md=30, tb=5, every=10, ragged=(0, 1)),
    dict(n=2, lX=200, lY=410, e=30, md=12, tb=10, every=40, ragged=(1, 1)),  # windows shorter than the margin
])
def test_banded_matches_oracle(ctx, case):
    batch = synth.make_batch(31, case["n"], case["lX"], case["lY"], anchor_every=case["every"])
    tris = run(ctx, batch, models_of(batch), band_params(0.01, case["md"], case["tb"], case["e"]), case["ragged"])
    assert all(len(t) > 0 for t in tris)


def test_unbanded_and_degenerate_items(ctx):
    batch = synth.make_batch(32, 3, 80, 170, anchor_every=10 ** 6)
    base = batch["items"][0]
    batch["items"] += [dict(base, lX=0, n_anchors=0), dict(base, lY=0, n_anchors=0), dict(base, lX=1, lY=1, n_anchors=0)]
    run(ctx, batch, models_of(batch), band_params(0.01, 100, 20, 40), (1, 1), unbanded=True)
    run(ctx, batch, models_of(batch), band_params(0.01, 100, 20, 40), (0, 0), unbanded=False)


def test_toy_pairs_of_the_reference_on_the_gpu(ctx, template_model):
    # tests/signalPairwiseTest.c:687-787: the seven toy events inside a 67-nucleotide sequence, exactly 8 pairs >= 0.2
    match, _, gapy = template_model
    m = o.Sm4Model(match, gapy)
    ev = np.array(TOY_EVENTS).reshape(-1, 3)
    batch = dict(items=[dict(x_offset=0, lX=len(TOY4_X) - 5, y_offset=0, lY=7, anchor_offset=0, n_anchors=0, model=0)],
                 x_chars=TOY4_X, events=ev, anchors=np.zeros((0, 2), np.int64), models=[(match, None, gapy)])
    (tri,) = run(ctx, batch, [m], band_params(0.2, 100, 20, 40), (0, 0), unbanded=True)
    assert sorted((int(x), int(y)) for _, x, y in tri) == TOY4_PAIRS


def test_the_shipped_read_gives_988_pairs_on_the_gpu(ctx, template_model, zymo_read):
    # tests/signalPairwiseTest.c:1230-1237: the 4-state machine scaled for the read, un-banded, default threshold
    match, _, gapy = template_model
    m = o.Sm4Model(match, gapy).scaled(*zymo_read["template_params"])
    ref = zymo_read["reference"]
    ev = np.asarray(zymo_read["template_events"], np.float64).reshape(-1, 3)
    batch = dict(items=[dict(x_offset=0, lX=len(ref) - 5, y_offset=0, lY=len(ev), anchor_offset=0, n_anchors=0, model=0)],
                 x_chars=ref, events=ev, anchors=np.zeros((0, 2), np.int64), models=[(m.match, None, gapy)])
    (tri,) = run(ctx, batch, [m], band_params(0.01, 1000, 40, 20), (1, 1), unbanded=True)
    assert len(tri) == 988
