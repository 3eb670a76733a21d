"""GPU parity of the 3-state vanilla signal machine against the oracle, through the C-ABI
(cpecan_hip_modelsv_create / cpecan_hip_batch_create_vanilla -> cpecan_k_generalv).

The oracle's vanilla machine is pinned by the reference's toy known answer
(tests/signalPairwiseTest.c:795-892: exactly 5 pairs >= 0.5) in test_oracle_golden.py.  Bar: every
totalProbability refresh and posterior exponent bit-identical, pairs in the reference's emission order,
integer posteriors within 1 of 1e7 (device exp vs host libm exp)."""
import numpy as np
import pytest

import pyoracle as o
import synth
from harness import assert_same_pairs, band_params, cp, make_items, orc_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = cp.Context(0)
    yield c
    c.close()


def skip_bins(seed):
    rng = np.random.default_rng(seed)
    return np.sort(rng.uniform(0.05, 0.4, 30))[::-1].copy()  # larger level difference, fewer skips


def run(ctx, batch, models, bp, ragged, unbanded=False, general=False):
    """batch as made by synth.make_batch; models: one pyoracle.VanillaModel per batch model"""
    ctx.models_clear()
    ctx.modelsv_create([(m.scalars, m.match, m.skip, m.gap_y) for m in models])
    b = cp.Batch(ctx, make_items(batch, ragged), batch["x_chars"], batch["events"], batch["anchors"], bp,
                 flags=(cp.FLAG_UNBANDED if unbanded else 0) | (cp.FLAG_GENERAL_KERNEL if general else 0), vanilla=True)
    if general or unbanded:
        assert b.info()["kernel"] == "general"
    elif b.info()["max_band_width"] <= 184:
        assert b.info()["kernel"] == "systolic" and b.info()["family"] == "wave"
    b.run()
    b.sync()
    npairs, ntot, ncells = b.counts()
    p = orc_params(bp, split=1 << 60)
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
        assert len(tri) > 0
    b.close()


def test_toy_known_answer(ctx, template_model):
    # the reference's test_vanilla_diagonalDPCalculations inputs: exactly these 5 pairs >= 0.5
    match, skip, gapy = template_model
    m = o.VanillaModel(match, skip, gapy)
    sY = np.array([58.743435, 0.887833, 0.0571, 53.604965, 0.816836, 0.0571, 58.432015, 0.735143, 0.0571,
                   63.684352, 0.795437, 0.0571, 58.921430, 0.812959, 0.0571, 59.895882, 0.740952, 0.0571,
                   61.684303, 0.722332, 0.0571]).reshape(-1, 3)
    batch = dict(x_chars="ACGATACGGACAT", events=sY, anchors=np.zeros((0, 2), np.int64),
                 items=[dict(x_offset=0, lX=8, y_offset=0, lY=7, anchor_offset=0, n_anchors=0, model=0)])
    ctx.models_clear()
    ctx.modelsv_create([(m.scalars, m.match, m.skip, m.gap_y)])
    b = cp.Batch(ctx, make_items(batch), batch["x_chars"], sY, batch["anchors"], band_params(0.5),
                 flags=cp.FLAG_UNBANDED, vanilla=True)
    b.run()
    b.sync()
    tri, _ = b.pairs(0, b.counts()[0][0])
    assert sorted((int(x), int(y)) for _, x, y in tri) == [(2, 0), (3, 3), (5, 4), (6, 5), (7, 6)]
    b.close()
    run(ctx, batch, [m], band_params(0.5), (0, 0), unbanded=True)


@pytest.mark.parametrize("general", [False, True], ids=["wave", "general"])
@pytest.mark.parametrize("case", [
    dict(n=3, lX=120, lY=250, e=20, md=60, tb=10, every=25, ragged=(0, 0)),
    dict(n=3, lX=300, lY=610, e=40, md=100, tb=40, every=50, ragged=(1, 1)),
    dict(n=2, lX=400, lY=800, e=100, md=300, tb=40, every=50, ragged=(1, 0)),
])
def test_vanilla_matches_oracle(ctx, case, general):
    batch = synth.make_batch(51, case["n"], case["lX"], case["lY"], anchor_every=case["every"])
    models = []
    for i, (match, _, gapy) in enumerate(batch["models"]):
        # strand-specific fudge factors as stateMachine3Vanilla_setStrandTransitionsToDefaults sets them
        strand = (np.float32(0.17), np.float32(0.55)) if i % 2 == 0 else (np.float32(0.14), np.float32(0.49))
        models.append(o.VanillaModel(match, skip_bins(i), gapy, float(strand[0]), float(strand[1])))
    run(ctx, batch, models, band_params(0.01, case["md"], case["tb"], case["e"]), case["ragged"], general=general)


@pytest.mark.parametrize("general", [False, True], ids=["wave", "general"])
@pytest.mark.parametrize("shape", [
    dict(n=4, lX=150, lY=310, every=30, md=60, tb=10, e=20),      # two cells per lane, short windows
    dict(n=3, lX=700, lY=1500, every=50, md=300, tb=40, e=100),   # three cells per lane (band 101-156), long windows
    dict(n=2, lX=90, lY=200, every=10 ** 6, md=40, tb=5, e=20),    # no anchors
])
def test_vanilla_expectations_match_oracle(ctx, shape, general):
    """Baum-Welch sums of the vanilla machine (diagonalCalculation_Expectations with
    cell_signal_updateBetaAndAlphaProb, impl/pairwiseAligner.c:478-498): 30 beta + 30 alpha skip bins and the
    likelihood, per model -- on the wave-per-alignment kernels (sweeps + cpecan_k_wv_expect) and on the general
    kernel.  Per-cell terms are added in another order than the host loop: 1e-9 relative."""
    batch = synth.make_batch(57, shape["n"], shape["lX"], shape["lY"], anchor_every=shape["every"])
    models = [o.VanillaModel(match, skip_bins(i), gapy) for i, (match, _, gapy) in enumerate(batch["models"])]
    bp = band_params(0.01, shape["md"], shape["tb"], shape["e"])
    ctx.models_clear()
    ids = ctx.modelsv_create([(m.scalars, m.match, m.skip, m.gap_y) for m in models])
    b = cp.Batch(ctx, make_items(batch, (1, 1)), batch["x_chars"], batch["events"], batch["anchors"], bp,
                 flags=cp.FLAG_EXPECTATIONS | (cp.FLAG_GENERAL_KERNEL if general else 0), vanilla=True)
    info = b.info()
    assert info["kernel"] == ("general" if general else "systolic"), info
    b.run()
    b.sync()
    p = orc_params(bp, split=1 << 60)
    hmms = [o.OrcExpectationsV() for _ in models]
    for it in batch["items"]:
        x = batch["x_chars"][it["x_offset"]: it["x_offset"] + it["lX"] + 5]
        ev = batch["events"][it["y_offset"]: it["y_offset"] + it["lY"]]
        an = batch["anchors"][it["anchor_offset"]: it["anchor_offset"] + it["n_anchors"]]
        o.expectations_v_using_anchors(models[it["model"]], x, it["lX"], ev, an, p, hmms[it["model"]], True, True)
    seen = 0
    for mid, hmm in zip(ids, hmms):
        ref = hmm.as_array()
        got = b.expectations(mid)
        assert np.allclose(got, ref, rtol=1e-9, atol=1e-12)
        seen += np.count_nonzero(ref[:60])
        assert ref[-1] < 0
    assert seen > 20  # several beta and alpha bins were hit
    b.close()
