"""GPU parity of the systolic (register-wavefront) kernel against the oracle, through the C-ABI.

Same bar as the general kernel: posterior exponents and totalProbability values bit-identical to the
oracle's doubles, aligned pairs in the reference's emission order, integer posteriors within 1."""
import numpy as np
import pytest

import synth
from harness import assert_same_pairs, band_params, cp, run_gpu, run_oracle_item

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = cp.Context(0)
    yield c
    c.close()


def test_division_by_reciprocal_is_exact(ctx):
    # the kernel's (x-mu)/sigma: Markstein-corrected multiply == IEEE division on 1e8 operands
    assert ctx.selftest_division(100_000_000, seed=7) == 0


CASES = [
    dict(n=4, lX=120, lY=250, e=20, md=60, tb=10, every=25, ragged=(0, 0)),
    dict(n=4, lX=300, lY=610, e=40, md=100, tb=40, every=50, ragged=(1, 1)),
    dict(n=3, lX=700, lY=1400, e=100, md=300, tb=40, every=50, ragged=(1, 1)),
    dict(n=3, lX=257, lY=400, e=100, md=150, tb=40, every=50, ragged=(1, 0)),
    dict(n=3, lX=90, lY=200, e=0, md=30, tb=5, every=10, ragged=(0, 1)),
    dict(n=2, lX=520, lY=500, e=60, md=120, tb=20, every=40, ragged=(0, 0)),   # more k-mers than events
    dict(n=2, lX=64, lY=64, e=200, md=1000, tb=40, every=1000, ragged=(1, 1)),  # unanchored, full matrix
    # windows shorter than the traceback margin: a launch's diagonals are decoded by windows two and
    # more ahead, so the forward kernel keeps every state everywhere
    dict(n=2, lX=200, lY=410, e=30, md=12, tb=10, every=40, ragged=(1, 1)),
    dict(n=2, lX=200, lY=410, e=30, md=25, tb=10, every=40, ragged=(0, 0)),
    # bands of 181..231 k-mers: too wide for the three-wave build, the four-wave build is picked
    dict(n=2, lX=600, lY=1200, e=180, md=400, tb=40, every=50, ragged=(1, 1)),
]


@pytest.fixture(params=[0, 4], ids=["rows-auto", "rows-4"])
def rows(request, monkeypatch):
    """the builds of the kernels: the one with the fewest waves per workgroup that holds the band (1, 2, 3 or 4
    waves for bands up to 56, 120, 184, 248 k-mers; picked by itself), and the four-wave build forced"""
    if request.param:
        monkeypatch.setenv("CPECAN_SYSTOLIC_ROWS", str(request.param))
    return request.param


@pytest.fixture(params=[0, cp.FLAG_WORKGROUP_KERNELS], ids=["wave", "workgroup"])
def family(request):
    """the two families of register-resident kernels, each against the oracle: one wave per alignment (the default;
    its C3-shaped batches run the assembly sweeps) and one workgroup per alignment (CPECAN_FLAG_WORKGROUP_KERNELS: what
    pipelined batches and the two-context E-step run on)"""
    return request.param


def expected_build(family, rows, width):
    """waves_per_workgroup of the build the library picks by itself: the workgroup family takes the fewest waves that
    hold the band (1, 2, 3, 4 for bands up to 56, 120, 184, 248 k-mers), the wave family the fewest cells per lane
    from 2 (bands up to 120, 184, 248)"""
    if rows == 4:
        return 4
    if family == cp.FLAG_WORKGROUP_KERNELS:
        return 1 + (width > 56) + (width > 120) + (width > 184)
    return 2 + (width > 120) + (width > 184)


@pytest.mark.parametrize("case", CASES)
def test_systolic_matches_oracle(ctx, case, rows, family):
    batch = synth.make_batch(21, case["n"], case["lX"], case["lY"], anchor_every=case["every"])
    bp = band_params(0.01, case["md"], case["tb"], case["e"])
    res, b = run_gpu(ctx, batch, bp, kernel=cp.KERNEL_SYSTOLIC, flags=family, ragged=case["ragged"])
    info = b.info()
    assert info["family"] == ("workgroup" if family else "wave")
    assert info["waves_per_workgroup"] == expected_build(family, rows, info["max_band_width"])
    for i in range(case["n"]):
        ref = run_oracle_item(batch, i, bp, case["ragged"])
        assert res[i]["cells"] == ref["cells"]
        assert np.array_equal(res[i]["totals_xay"], ref["totals_xay"])
        assert np.array_equal(res[i]["totals"], ref["totals"])
        assert_same_pairs(res[i], ref)


@pytest.mark.parametrize("flags", [0, cp.FLAG_SMALL_FOOTPRINT], ids=["three-window-ring", "small-footprint"])
def test_assembly_sweeps_in_both_layouts_match_oracle(ctx, flags):
    # a band of 121-158 k-mers over seven traceback windows: the hand-scheduled sweeps, with the ring of three windows
    # and the post kernel on a stream of its own, and with the two-window ring of CPECAN_FLAG_SMALL_FOOTPRINT
    batch = synth.make_batch(33, 3, 700, 1400, anchor_every=50)
    bp = band_params(0.01, 300, 40, 100)
    res, b = run_gpu(ctx, batch, bp, kernel=cp.KERNEL_AUTO, flags=flags, ragged=(1, 1))
    assert b.info()["assembly_sweeps"] == 2
    for i in range(3):
        ref = run_oracle_item(batch, i, bp, (1, 1))
        assert res[i]["cells"] == ref["cells"]
        assert np.array_equal(res[i]["totals_xay"], ref["totals_xay"])
        assert np.array_equal(res[i]["totals"], ref["totals"])
        assert_same_pairs(res[i], ref)
    # run again on the same batch (the windows' records, ring and scratch halves are reused)
    b.run()
    b.sync()
    npairs, _, _ = b.counts()
    for i in range(3):
        tri, lp = b.pairs(i, npairs[i])
        assert np.array_equal(tri, res[i]["triples"]) and np.array_equal(lp, res[i]["logp"])
    b.close()


@pytest.mark.parametrize("threshold", [0.0, 0.01, 1e-4, 1e-7])
def test_systolic_decode_paths_agree(ctx, threshold, rows, family):
    # the candidate-list decode (default) and the full-scan decode (CPECAN_FLAG_SCAN_DECODE, also the
    # path a window falls back to by itself; a tiny threshold makes long candidate lists, threshold 0 emits every
    # cell of the band with x, y > 0 and overflows the first pair allocation: the count-then-allocate re-run)
    batch = synth.make_batch(25, 3, 300, 620, anchor_every=50)
    bp = band_params(threshold, 100, 40, 60)
    a, _ = run_gpu(ctx, batch, bp, kernel=cp.KERNEL_SYSTOLIC, flags=family, ragged=(1, 1))
    s, _ = run_gpu(ctx, batch, bp, kernel=cp.KERNEL_SYSTOLIC, flags=family | cp.FLAG_SCAN_DECODE, ragged=(1, 1))
    for i, (x, y) in enumerate(zip(a, s)):
        assert np.array_equal(x["triples"], y["triples"])
        assert np.array_equal(x["logp"], y["logp"])
        ref = run_oracle_item(batch, i, bp, (1, 1))
        assert_same_pairs(x, ref)


def test_systolic_ragged_batch_and_degenerate_items(ctx, family):
    batch = synth.make_batch(22, 12, 200, 400, anchor_every=40, length_sigma=0.6)
    base = batch["items"][0]
    batch["items"] += [dict(base, lX=0, n_anchors=0), dict(base, lY=0, n_anchors=0),
                       dict(base, lX=0, lY=0, n_anchors=0), dict(base, lX=1, lY=1, n_anchors=0)]
    bp = band_params(0.01, 100, 20, 60)
    res, b = run_gpu(ctx, batch, bp, kernel=cp.KERNEL_SYSTOLIC, flags=family, ragged=(1, 1))
    for i in range(len(batch["items"])):
        ref = run_oracle_item(batch, i, bp, (1, 1))
        assert np.array_equal(res[i]["totals"], ref["totals"]), i
        assert_same_pairs(res[i], ref)


def test_systolic_equals_general_kernel(ctx):
    batch = synth.make_batch(23, 6, 400, 800, anchor_every=50)
    bp = band_params(0.01, 200, 40, 100)
    a, _ = run_gpu(ctx, batch, bp, kernel=cp.KERNEL_SYSTOLIC, ragged=(1, 1))
    g, _ = run_gpu(ctx, batch, bp, kernel=cp.KERNEL_GENERAL, ragged=(1, 1))
    for x, y in zip(a, g):
        assert np.array_equal(x["triples"], y["triples"])
        assert np.array_equal(x["logp"], y["logp"])
        assert np.array_equal(x["totals"], y["totals"])


def test_too_wide_band_is_refused_by_systolic_and_routed_by_auto(ctx):
    batch = synth.make_batch(24, 1, 400, 800, anchor_every=400)  # sparse anchors: band > 256
    bp = band_params(0.01, 200, 40, 300)
    with pytest.raises(cp.CpecanError) as ei:
        run_gpu(ctx, batch, bp, kernel=cp.KERNEL_SYSTOLIC)
    assert ei.value.code == cp.EINVAL
    res, b = run_gpu(ctx, batch, bp, kernel=cp.KERNEL_AUTO)
    ref = run_oracle_item(batch, 0, bp)
    assert_same_pairs(res[0], ref)


@pytest.mark.parametrize("case", [
    dict(n=4, lX=150, lY=310, e=40, md=80, tb=20, every=25, ragged=(1, 1)),
    dict(n=3, lX=400, lY=800, e=100, md=200, tb=40, every=50, ragged=(0, 0)),
    dict(n=2, lX=200, lY=410, e=30, md=12, tb=10, every=40, ragged=(1, 0)),
])
def test_systolic_expectations_match_oracle(ctx, case, rows, family):
    # Baum-Welch sufficient statistics from the systolic kernels (forward, backward with the B ring,
    # element-wise expectation kernel) against the oracle and against the general kernel
    import pyoracle as o
    batch = synth.make_batch(27, case["n"], case["lX"], case["lY"], anchor_every=case["every"],
                             distinct_models=False)
    bp = band_params(0.01, case["md"], case["tb"], case["e"])
    res, b = run_gpu(ctx, batch, bp, mode=cp.MODE_EXPECTATIONS, kernel=cp.KERNEL_SYSTOLIC, flags=family,
                     ragged=case["ragged"])
    assert b.info()["kernel"] == "systolic" and b.info()["family"] == ("workgroup" if family else "wave")
    got = b.expectations(0)
    hmm = o.OrcExpectations()
    for i in range(case["n"]):
        ref = run_oracle_item(batch, i, bp, case["ragged"], expectations=hmm)
        assert np.array_equal(res[i]["totals"], ref["totals"])
    # tolerance: the device sums in a different order and uses its own exp(): 1e-9 relative
    assert np.allclose(got[:9], np.array(hmm.transitions[:]), rtol=1e-9, atol=1e-12)
    assert np.allclose(got[9:9 + 4096], np.array(hmm.kmerGap[:]), rtol=1e-9, atol=1e-12)
    assert np.isclose(got[-1], hmm.likelihood, rtol=1e-12)
    _, g = run_gpu(ctx, batch, bp, mode=cp.MODE_EXPECTATIONS, kernel=cp.KERNEL_GENERAL, ragged=case["ragged"])
    assert np.allclose(got, g.expectations(0), rtol=1e-9, atol=1e-12)
