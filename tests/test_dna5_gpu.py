"""GPU parity of the 5-state symbol machine (DNA against DNA, BASELINE configs[0]) against
the oracle, through the C-ABI (cpecan_hip_batch_create_dna -> cpecan_k_general5).

The oracle's 5-state machine is pinned by the reference's toy known answer
(tests/pairwiseAlignerTest.c:278-373: "AGCG" / "AGTTCG", 4 aligned pairs) in test_oracle_golden.py.
Bar: totalProbability refreshes and posterior exponents bit-identical, pairs in the reference's
emission order, integer posteriors within 1 of 1e7 (device exp vs host libm exp)."""
import numpy as np
import pytest

import pyoracle as o
from harness import assert_same_pairs, band_params, cp, orc_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = cp.Context(0)
    yield c
    c.close()


def evolve(rng, n, sub=0.2, indel=0.05):
    """random ACGT string and a mutated copy (as the reference's tests do with evolveSequence,
    tests/randomSequences.c); returns x, y and the (x, y) index pairs of the unmutated bases"""
    x = rng.choice(list("ACGT"), n)
    y, pairs = [], []
    for i, ch in enumerate(x):
        r = rng.random()
        if r < indel:
            continue                                   # deletion
        if r < 2 * indel:
            y.append(rng.choice(list("ACGT")))          # insertion, then the base itself
        if rng.random() < sub:
            y.append(rng.choice([b for b in "ACGT" if b != ch]))
        else:
            pairs.append((i, len(y)))
            y.append(ch)
    return "".join(x), "".join(y), np.array(pairs, np.int64).reshape(-1, 2)


def run_case(ctx, seqs, bp, ragged, flags=0, unbanded=False):
    model = o.Sm5Model()
    ctx.models_clear()
    ids = ctx.models5_create([(list(model.c.t), model.match, model.gx, model.gy)])
    xs, ys, an = "", "", []
    items = np.zeros(len(seqs), cp.ITEM_DTYPE)
    for i, (x, y, a) in enumerate(seqs):
        items[i] = (len(xs), len(x), len(ys), len(y), sum(len(q) for q in an), len(a), ids[0],
                    ragged[0], ragged[1], 0)
        xs += x
        ys += y
        an.append(a)
    anchors = np.concatenate(an) if an else np.zeros((0, 2), np.int64)
    b = cp.Batch(ctx, items, xs, None, anchors, bp, flags=flags, y_chars=ys)
    b.run()
    b.sync()
    npairs, ntot, ncells = b.counts()
    info = b.info()
    assert info["kernel"] == "general"
    if not unbanded and info["max_band_width"] <= 192:  # posterior decode of bands a wave covers: the 5-state wave kernel
        assert (info.get("family") == "wave (5-state)") == (not flags & cp.FLAG_GENERAL_KERNEL)
    p = orc_params(bp, split=1 << 60)
    for i, (x, y, a) in enumerate(seqs):
        tri, lp = b.pairs(i, npairs[i])
        xay, tot = b.totals(i, ntot[i])
        if unbanded:  # the oracle returns this list in the caller's (ascending) order
            ref = o.aligned_pairs_without_banding(model, x, len(x), y, p, ragged[0], ragged[1])
            order = np.lexsort((ref["triples"][:, 1], -(ref["triples"][:, 1] + ref["triples"][:, 2])))
            ref["triples"], ref["logp"] = ref["triples"][order], ref["logp"][order]
        else:
            ref = o.aligned_pairs_using_anchors(model, x, len(x), y, a, p, ragged[0], ragged[1])
            ref["triples"] = ref["triples"][::-1]  # undo the stList_pop reversal: emission order
            ref["logp"] = ref["logp"][::-1]
        assert int(ncells[i]) == ref["cells"]
        assert np.array_equal(xay, ref["totals_xay"])
        assert np.array_equal(tot, ref["totals"])
        assert_same_pairs(dict(triples=tri, logp=lp), ref)
        assert len(tri) > 0
    b.close()


def test_toy_known_answer(ctx):
    # the reference's test_diagonalDPCalculations inputs (tests/pairwiseAlignerTest.c:278-373)
    bp = band_params(0.2, 4, 1, 2)
    run_case(ctx, [("AGCG", "AGTTCG", np.zeros((0, 2), np.int64))], bp, (0, 0), flags=cp.FLAG_UNBANDED,
             unbanded=True)


KERNEL_FORMS = [("0", False), ("1", False), (None, True)]  # (CPECAN_WAVE5_PAIRED, general kernel)
KERNEL_IDS = ["wave", "pair", "general"]


def pick_form(monkeypatch, form):
    """one wave per alignment, a forward + a backward wave per alignment (what the library picks for batches that
    leave SIMDs idle), or the general kernel"""
    paired, general = form
    if paired is None:
        monkeypatch.delenv("CPECAN_WAVE5_PAIRED", raising=False)
    else:
        monkeypatch.setenv("CPECAN_WAVE5_PAIRED", paired)
    return cp.FLAG_GENERAL_KERNEL if general else 0


@pytest.mark.parametrize("form", KERNEL_FORMS, ids=KERNEL_IDS)
@pytest.mark.parametrize("case", [
    dict(n=3, length=60, e=20, md=30, tb=5, ragged=(0, 0), anchored=False),
    dict(n=3, length=150, e=10, md=40, tb=8, ragged=(1, 1), anchored=True),
    dict(n=2, length=300, e=20, md=100, tb=40, ragged=(1, 0), anchored=True),
    dict(n=2, length=700, e=50, md=150, tb=20, ragged=(0, 1), anchored=True),   # two cells per lane
    dict(n=2, length=900, e=80, md=200, tb=40, ragged=(0, 0), anchored=True),   # three cells per lane
])
def test_dna5_matches_oracle(ctx, case, form, monkeypatch):
    """the wave-per-alignment kernels (cpecan_kernel_wave5.hip, one to three cells per lane by band width; one wave
    for both sweeps or a pair of waves) and the general one (CPECAN_FLAG_GENERAL_KERNEL)"""
    flags = pick_form(monkeypatch, form)
    rng = np.random.default_rng(31 + case["length"])
    seqs = []
    for _ in range(case["n"]):
        x, y, pairs = evolve(rng, case["length"])
        a = pairs[5::12] if case["anchored"] else np.zeros((0, 2), np.int64)
        seqs.append((x, y, a))
    bp = band_params(0.01, case["md"], case["tb"], case["e"])
    run_case(ctx, seqs, bp, case["ragged"], flags=flags)


def test_config1_two_1kb_sequences(ctx):
    # BASELINE configs[0]: two ~1 kb sequences, no anchors (full matrix), the reference's default
    # banding parameters
    rng = np.random.default_rng(41)
    x, y, _ = evolve(rng, 1000)
    bp = band_params(0.01, 1000, 40, 20)
    run_case(ctx, [(x, y, np.zeros((0, 2), np.int64))], bp, (0, 0))


@pytest.mark.parametrize("form", KERNEL_FORMS, ids=KERNEL_IDS)
@pytest.mark.parametrize("shape", [dict(base=120, step=40, e=12, md=60, tb=10, every=9),
                                   dict(base=500, step=100, e=50, md=150, tb=20, every=40),   # two cells per lane
                                   dict(base=700, step=100, e=84, md=200, tb=40, every=60)])  # three
def test_discrete_expectations_match_oracle(ctx, shape, form, monkeypatch):
    """Baum-Welch sums of the 5-state machine (getExpectationsUsingAnchors with
    diagonalCalculation_Expectations and cell_updateExpectations, impl/pairwiseAligner.c:407-424,:841-863):
    25 transitions, 5 x 16 emissions and the likelihood of a batch, summed per model.  The device adds
    the per-cell terms in another order than the host loop: 1e-9 relative (north_star asks 1e-6)."""
    general = pick_form(monkeypatch, form) != 0
    rng = np.random.default_rng(77)
    model = o.Sm5Model()
    ctx.models_clear()
    ids = ctx.models5_create([(list(model.c.t), model.match, model.gx, model.gy)])
    bp = band_params(0.01, shape["md"], shape["tb"], shape["e"])
    p = orc_params(bp, split=1 << 60)
    xs, ys, an, seqs = "", "", [], []
    items = np.zeros(4, cp.ITEM_DTYPE)
    for i in range(4):
        x, y, pairs = evolve(rng, shape["base"] + shape["step"] * i)
        a = pairs[4::shape["every"]]
        items[i] = (len(xs), len(x), len(ys), len(y), sum(len(q) for q in an), len(a), ids[0], 0, 0, 0)
        xs += x
        ys += y
        an.append(a)
        seqs.append((x, y, a))
    b = cp.Batch(ctx, items, xs, None, np.concatenate(an), bp,
                 flags=cp.FLAG_EXPECTATIONS | (cp.FLAG_GENERAL_KERNEL if general else 0), y_chars=ys)
    assert (b.info().get("family") == "wave (5-state)") == (not general)
    b.run()
    b.sync()
    got = b.expectations(ids[0])
    hmm = o.OrcExpectations5()
    for x, y, a in seqs:
        o.expectations5_using_anchors(model, x, len(x), y, a, p, hmm)
    ref = hmm.as_array()
    assert ref[-1] < 0 and ref[0] > 10  # a real likelihood, many match->match transitions
    assert np.count_nonzero(ref[:25]) == 13  # the transitions stateMachine5_cellCalculate takes
    assert np.allclose(got, ref, rtol=1e-9, atol=1e-12)
    b.close()


def test_degenerate_shapes_agree_between_the_kernels(ctx, monkeypatch):
    """one empty sequence, single bases, a 1 x n strip: the wave kernels (both forms) and the general kernel give the same pairs,
    totals and cell counts (the oracle is not asked: the reference's own entry points return before the DP there)"""
    model = o.Sm5Model()
    ctx.models_clear()
    ids = ctx.models5_create([(list(model.c.t), model.match, model.gx, model.gy)])
    seqs = [("", "ACGT"), ("ACGTAC", ""), ("A", "A"), ("A", "C"), ("G", "ACGTACGTAC"), ("ACGTACGTACGT", "T"),
            ("ACGTACGTTGCA", "ACGTCGTTGCA")]
    xs, ys = "", ""
    items = np.zeros(len(seqs), cp.ITEM_DTYPE)
    for i, (x, y) in enumerate(seqs):
        items[i] = (len(xs), len(x), len(ys), len(y), 0, 0, ids[0], i % 2, (i // 2) % 2, 0)
        xs += x
        ys += y
    out = []
    for form in KERNEL_FORMS:
        flags = pick_form(monkeypatch, form)
        b = cp.Batch(ctx, items, xs + "A", None, np.zeros((0, 2), np.int64), band_params(0.01, 4, 1, 2), flags=flags,
                     y_chars=ys + "A")
        b.run()
        b.sync()
        npairs, ntot, ncells = b.counts()
        res = []
        for i in range(len(seqs)):
            tri, lp = b.pairs(i, npairs[i])
            xay, tot = b.totals(i, ntot[i])
            res.append((tri.copy(), lp.copy(), xay.copy(), tot.copy(), int(ncells[i])))
        out.append(res)
        b.close()
    for res in out[:2]:
        for w, g in zip(res, out[2]):
            assert np.array_equal(w[0], g[0]) and np.array_equal(w[1].view(np.uint64), g[1].view(np.uint64))
            assert np.array_equal(w[2], g[2]) and np.array_equal(w[3].view(np.uint64), g[3].view(np.uint64)) and w[4] == g[4]
    assert len(out[0][-1][0]) > 5
