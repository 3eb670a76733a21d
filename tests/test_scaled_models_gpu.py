"""cpecan_hip_models_create_scaled: per-read scaled pore models assembled on the device from the unscaled model, the
read's five scaling parameters (emissions_signal_scaleModel, impl/stateMachine.c:631-651) and the few values per k-mer
that need the host libm.  The device tables must be bit-identical to cpecan_hip_models_create on tables scaled on the
host (by the oracle's C restatement of scaleModel), and so must every alignment made with them."""
import numpy as np
import pytest

import pyoracle as o
import synth
from harness import assert_same_pairs, band_params, cp, make_items

pytestmark = pytest.mark.gpu

SCALINGS = np.array([
    [1.0, 0.0, 1.0, 1.0, 1.0],
    [0.97, 3.25, 1.08, 1.13, 0.94],
    [1.05, -4.5, 0.9, 0.9, 1.2],
    [1.0, 0.0, 0.0, 1.0, 1.0],   # level sd 0: K1 = -inf, 1/sd = 0
    [1.0, 0.0, 1.0, 0.0, 1.0],   # noise mean 0: noise sd 0, K2 = -inf
    [0.7311, 11.0, 2.5, 3.0, 0.125],
])


def _host_scaled(match, sc):
    return o.Sm3Model(match, match).scaled(*[float(v) for v in sc]).match


def _same_bits(a, b):
    return np.array_equal(a.view(np.uint64), b.view(np.uint64))


@pytest.mark.parametrize("source", ["synthetic", "template_median68pA"])
def test_device_tables_are_bit_identical(source, template_model):
    if source == "synthetic":
        match, gx, gy = synth.synthetic_pore_model()
    else:
        match, gy = template_model[0], template_model[2]
        gx = np.full(cp.NUM_KMERS, np.log(0.1))
    t = cp.NANOPORE_TRANSITIONS
    ctx = cp.Context(0)
    ids = ctx.models_create_scaled((t, match, gx, gy), SCALINGS)
    assert list(ids) == list(range(len(SCALINGS)))
    got = [ctx.models_download(i) for i in ids]
    ctx.models_clear()
    ids = ctx.models_create([(t, _host_scaled(match, sc), gx, gy) for sc in SCALINGS])
    for i, sc in zip(ids, SCALINGS):
        want = ctx.models_download(i)
        assert got[i].size == want.size
        bad = np.flatnonzero(got[i].view(np.uint64) != want.view(np.uint64))
        assert bad.size == 0, (sc, bad[:5], got[i][bad[:5]], want[bad[:5]])
    ctx.close()


def test_tables_survive_appends_and_the_m_step():
    match, gx, gy = synth.synthetic_pore_model()
    t = np.array(cp.NANOPORE_TRANSITIONS, float)
    ctx = cp.Context(0)
    a = ctx.models_create([(t, _host_scaled(match, SCALINGS[1]), gx, gy), (t, match, gx, gy)])
    first = [ctx.models_download(i) for i in a]
    b = ctx.models_create_scaled((t, match, gx, gy), SCALINGS[1:4])
    c = ctx.models_create([(t, _host_scaled(match, SCALINGS[2]), gx, gy)])
    assert list(a) + list(b) + list(c) == list(range(6))
    # the tables created first are where they were (the table grew on the device), and equal rows came out equal
    assert _same_bits(ctx.models_download(0), first[0]) and _same_bits(ctx.models_download(1), first[1])
    assert _same_bits(ctx.models_download(2), first[0])
    assert _same_bits(ctx.models_download(5), ctx.models_download(3))
    # the M-step's in-place update reaches every model, whichever way it was created
    t2 = t + np.linspace(-0.3, -0.1, 9)
    g2 = np.log(np.random.default_rng(5).dirichlet(np.ones(cp.NUM_KMERS)))
    before = [ctx.models_download(i) for i in range(6)]
    ctx.models_set_transitions(t2, g2)
    for i in range(6):
        now = ctx.models_download(i)
        assert np.array_equal(now[:9], t2)
        rows = now[-4097 * 18:].reshape(4097, 18)
        assert np.array_equal(rows[:4096, 16], g2)
        keep = np.ones_like(rows, bool)
        keep[:4096, 16] = False
        assert _same_bits(rows[keep], before[i][-4097 * 18:].reshape(4097, 18)[keep])
    ctx.close()


@pytest.mark.parametrize("flags", [0, cp.FLAG_WORKGROUP_KERNELS, cp.FLAG_GENERAL_KERNEL])
def test_alignments_with_device_scaled_models_are_identical(flags):
    bt = synth.make_batch(41, 24, 300, 420, anchor_every=50)
    bp = band_params()
    t = cp.NANOPORE_TRANSITIONS
    res = []
    for scaled in (False, True):
        ctx = cp.Context(0)
        if scaled:
            ids = ctx.models_create_scaled((t,) + bt["base_model"], bt["scalings"])
        else:
            match, gx, gy = bt["base_model"]  # (scaled here with the C libm's pow, as the reference would)
            ids = ctx.models_create([(t, _host_scaled(match, sc), gx, gy) for sc in bt["scalings"]])
        assert list(ids) == list(range(24))
        b = cp.Batch(ctx, make_items(bt), bt["x_chars"], bt["events"], bt["anchors"], bp, cp.MODE_POSTERIOR,
                     cp.KERNEL_AUTO, flags)
        b.run()
        b.sync()
        npairs, ntot, _ = b.counts()
        out = []
        for i in range(b.n):
            tri, lp = b.pairs(i, npairs[i])
            xay, tot = b.totals(i, ntot[i])
            out.append(dict(triples=tri, logp=lp, totals=tot))
        res.append(out)
        b.close()
        ctx.close()
    for g, r in zip(res[1], res[0]):
        assert len(g["triples"]) > 100
        assert_same_pairs(g, r)
        assert _same_bits(np.asarray(g["totals"], float), np.asarray(r["totals"], float))
