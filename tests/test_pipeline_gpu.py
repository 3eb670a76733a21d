"""cpecan_hip_batch_run_after: a batch of one context queued behind a batch of another context on the device (the
stream-of-batches pipeline of bench.py --mode service).  Ordering must not change any result."""
import numpy as np
import pytest

import synth
from harness import assert_same_pairs, band_params, cp, make_items

pytestmark = pytest.mark.gpu


def _results(b):
    npairs, ntot, _ = b.counts()
    out = []
    for i in range(b.n):
        tri, lp = b.pairs(i, npairs[i])
        xay, tot = b.totals(i, ntot[i])
        out.append(dict(triples=tri, logp=lp, totals=tot))
    return out


@pytest.mark.parametrize("flags", [0, cp.FLAG_WORKGROUP_KERNELS])
def test_batches_chained_on_the_device_give_the_same_pairs(flags):
    bp = band_params()
    data = [synth.make_batch(60 + k, 16, 260 + 40 * k, 400 + 50 * k, anchor_every=50) for k in range(3)]
    ctxs = [cp.Context(0) for _ in data]
    alone, batches = [], []
    for cx, bt in zip(ctxs, data):
        cx.models_create_scaled((cp.NANOPORE_TRANSITIONS,) + bt["base_model"], bt["scalings"])
        b = cp.Batch(cx, make_items(bt), bt["x_chars"], bt["events"], bt["anchors"], bp, cp.MODE_POSTERIOR,
                     cp.KERNEL_AUTO, flags)
        b.run()
        b.sync()
        alone.append(_results(b))
        batches.append(b)
    # all three queued at once, each behind the one before; then a second round behind the first
    for rnd in range(2):
        for k, b in enumerate(batches):
            b.run(after=batches[k - 1] if (k > 0 or rnd > 0) else None)
    for b in reversed(batches):
        b.sync()
    for b, want in zip(batches, alone):
        got = _results(b)
        for g, r in zip(got, want):
            assert len(g["triples"]) > 100
            assert_same_pairs(g, r)
            assert np.array_equal(np.asarray(g["totals"]).view(np.uint64), np.asarray(r["totals"]).view(np.uint64))
    for b in batches:
        b.close()
    for cx in ctxs:
        cx.close()


def test_run_after_refuses_nothing_sensible():
    bt = synth.make_batch(71, 4, 200, 300, anchor_every=50)
    cx = cp.Context(0)
    cx.models_create_scaled((cp.NANOPORE_TRANSITIONS,) + bt["base_model"], bt["scalings"])
    b = cp.Batch(cx, make_items(bt), bt["x_chars"], bt["events"], bt["anchors"], band_params())
    b.run(after=b)  # itself: no condition
    b.sync()
    first = _results(b)
    other = cp.Batch(cx, make_items(bt), bt["x_chars"], bt["events"], bt["anchors"], band_params())
    b.run(after=other)  # a batch that has never run: no condition
    b.sync()
    for g, r in zip(_results(b), first):
        assert_same_pairs(g, r)
    other.close()
    b.close()
    cx.close()
