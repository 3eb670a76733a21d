"""GPU parity of the 3-state HDP signal machine (BASELINE configs[4]) against the oracle,
through the C-ABI (cpecan_hip_modelsh_create / cpecan_hip_batch_create_hdp -> cpecan_k_generalh), on the
reference's own serialized HDP (tests/test_hdp/testTemplate.nhdp, copied as data to tests/golden/).

Parity status of the ORACLE for this machine: partly pinned.  The k-mer id function reproduces the
reference's known answers (tests/nanoporeHdpTests.c:104-108, test_oracle_golden.py); the recurrence is the
strawMan machine's (pinned) with another emission; the density function (dir_proc_density /
grid_spline_interp) has no known answer in the reference's tests that can be reproduced here (its
alignment tests need lastz anchors), so the density values themselves are parity-unpinned.  This file
pins the GPU against the oracle: totals and exponents bit-identical."""
import os

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


@pytest.fixture(scope="module")
def nhdp(golden_dir):
    return o.load_nhdp(os.path.join(golden_dir, "testTemplate.nhdp"))


def hdp_batch(seed, n, lX, every, nhdp):
    """reads whose event means are drawn around the mode of each k-mer's HDP density"""
    rng = np.random.default_rng(seed)
    model = o.HdpModel(nhdp)
    xs, evs, ans, items = "", [], [], []
    for _ in range(n):
        x = "".join(rng.choice(list("ACGT"), lX + 5))
        ev, anchors = [], []
        for k in range(lX):
            row = nhdp["kmer_row"][model.kmer_id(x[k:k + 6])]
            mode = nhdp["grid"][int(np.argmax(nhdp["y"][row]))]
            if rng.random() < 0.1:
                continue                                   # skipped k-mer
            if k % every == 0:
                anchors.append((k, len(ev)))
            for _ in range(1 + rng.geometric(0.6) - 1 if rng.random() < 0.5 else 1):
                ev.append((mode + rng.normal(0, 1.0), abs(rng.normal(1.0, 0.2)) + 1e-3, 0.01))
        items.append(dict(x_offset=len(xs), lX=lX, y_offset=sum(len(e) for e in evs), lY=len(ev),
                          anchor_offset=sum(len(a) for a in ans), n_anchors=len(anchors), model=0))
        xs += x
        evs.append(np.array(ev))
        ans.append(np.array(anchors, np.int64).reshape(-1, 2))
    return dict(x_chars=xs, events=np.concatenate(evs), anchors=np.concatenate(ans), items=items), model


@pytest.mark.parametrize("general", [False, True], ids=["wave", "general"])
@pytest.mark.parametrize("case", [
    dict(n=3, lX=100, e=20, md=60, tb=10, every=25, ragged=(0, 0)),
    dict(n=2, lX=300, e=40, md=100, tb=40, every=40, ragged=(1, 1)),
    dict(n=2, lX=700, e=100, md=200, tb=40, every=50, ragged=(1, 1)),   # three cells per lane, several windows
    dict(n=1, lX=400, e=160, md=150, tb=40, every=80, ragged=(1, 0)),   # four cells per lane
])
def test_hdp_matches_oracle(ctx, nhdp, case, general):
    """the wave-per-alignment HDP kernels (the default for the posterior decode) and the general kernel
    (CPECAN_FLAG_GENERAL_KERNEL) against the oracle: totals, exponents, pairs and their order identical"""
    batch, model = hdp_batch(61, case["n"], case["lX"], case["every"], nhdp)
    ctx.models_clear()
    ctx.modelsh_create([(cp.NANOPORE_TRANSITIONS, nhdp["alphabet"], nhdp["grid"], nhdp["y"], nhdp["slope"],
                         nhdp["kmer_row"])])
    bp = band_params(0.01, case["md"], case["tb"], case["e"])
    b = cp.Batch(ctx, make_items(batch, case["ragged"]), batch["x_chars"], batch["events"], batch["anchors"], bp,
                 flags=cp.FLAG_GENERAL_KERNEL if general else 0, hdp=True)
    assert b.info()["kernel"] == ("general" if general else "systolic")
    if not general:
        assert b.info()["family"] == "wave"
    b.run()
    b.sync()
    npairs, ntot, _ = b.counts()
    p = orc_params(bp, split=1 << 60)
    for i, it in enumerate(batch["items"]):
        x = batch["x_chars"][it["x_offset"]: it["x_offset"] + it["lX"] + 5]
        ev = batch["events"][it["y_offset"]: it["y_offset"] + it["lY"]]
        an = batch["anchors"][it["anchor_offset"]: it["anchor_offset"] + it["n_anchors"]]
        tri, lp = b.pairs(i, npairs[i])
        xay, tot = b.totals(i, ntot[i])
        ref = o.aligned_pairs_using_anchors(model, x, it["lX"], ev, an, p, case["ragged"][0], case["ragged"][1])
        ref["triples"], ref["logp"] = ref["triples"][::-1], ref["logp"][::-1]
        assert np.array_equal(xay, ref["totals_xay"])
        assert np.array_equal(tot, ref["totals"])
        assert_same_pairs(dict(triples=tri, logp=lp), ref)
        assert len(tri) > it["lX"] // 4
    b.close()


@pytest.mark.parametrize("general", [False, True], ids=["wave", "general"])
@pytest.mark.parametrize("shape", [dict(seed=67, n=3, lX=120, every=25, md=60, tb=10, e=20),
                                   dict(seed=68, n=2, lX=500, every=50, md=200, tb=40, e=100),
                                   dict(seed=69, n=2, lX=90, every=10 ** 6, md=40, tb=5, e=20)])
def test_hdp_expectations_and_assignments_match_oracle(ctx, nhdp, shape, general):
    """Baum-Welch sums of the HDP machine (diagonalCalculation_Expectations with
    cell_signal_updateTransAndKmerSkipExpectations2, impl/pairwiseAligner.c:445-476): 9 transitions and the
    likelihood per model (another summation order than the host loop: 1e-9 relative), and per alignment the
    event-to-k-mer assignments -- one per transition INTO match whose posterior reaches the HdpHmm's
    threshold -- bit-identical and in the reference's order.  On the wave-per-alignment kernels (the assignments
    leave the device unordered, tagged with their window, and the library orders them) and on the general kernel."""
    batch, model = hdp_batch(shape["seed"], shape["n"], shape["lX"], shape["every"], nhdp)
    ctx.models_clear()
    ids = ctx.modelsh_create([(cp.NANOPORE_TRANSITIONS, nhdp["alphabet"], nhdp["grid"], nhdp["y"],
                               nhdp["slope"], nhdp["kmer_row"])])
    threshold = 0.05  # the machine's posteriors are flat (quirk Q6): a low bar gives a few hundred assignments
    bp = band_params(threshold, shape["md"], shape["tb"], shape["e"])
    b = cp.Batch(ctx, make_items(batch, (1, 1)), batch["x_chars"], batch["events"], batch["anchors"], bp,
                 flags=cp.FLAG_EXPECTATIONS | (cp.FLAG_GENERAL_KERNEL if general else 0), hdp=True)
    assert b.info()["kernel"] == ("general" if general else "systolic")
    b.run()
    b.sync()
    npairs, _, _ = b.counts()
    p = orc_params(bp, split=1 << 60)
    reads = []
    for it in batch["items"]:
        x = batch["x_chars"][it["x_offset"]: it["x_offset"] + it["lX"] + 5]
        ev = batch["events"][it["y_offset"]: it["y_offset"] + it["lY"]]
        an = batch["anchors"][it["anchor_offset"]: it["anchor_offset"] + it["n_anchors"]]
        reads.append((x, it["lX"], ev, an))
    total = o.expectations_h_using_anchors(model, reads, p, threshold, True, True)
    got = b.expectations(ids[0])
    assert np.allclose(got[:9], total["transitions"], rtol=1e-9, atol=1e-12)
    assert np.isclose(got[9], total["likelihood"], rtol=1e-12) and total["likelihood"] != 0.0
    n_assign = 0
    for i, rd in enumerate(reads):
        ref = o.expectations_h_using_anchors(model, [rd], p, threshold, True, True)
        tri, lp = b.pairs(i, npairs[i])
        assert np.array_equal(tri, ref["assign"])
        assert np.array_equal(lp, ref["logp"])
        n_assign += len(tri)
    assert n_assign > 50 and n_assign == len(total["assign"])
    b.close()
