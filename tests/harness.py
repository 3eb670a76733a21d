"""Shared helpers of the GPU parity tests: run work items through the C-ABI, run the same items on
the oracle, and compare."""
import numpy as np

import pyoracle as o
from cpecan_load import binding

cp = binding()


def band_params(threshold=0.01, min_diags=1000, tb_diags=40, expansion=20):
    return cp.BandParams(threshold, min_diags, tb_diags, expansion)


def orc_params(bp, split=3000 * 3000):
    return o.default_params(threshold=bp.threshold, minDiagsBetweenTraceBack=bp.minDiagsBetweenTraceBack,
                            traceBackDiagonals=bp.traceBackDiagonals,
                            diagonalExpansion=bp.diagonalExpansion, splitMatrixBiggerThanThis=split)


def make_items(batch, ragged=(0, 0)):
    items = np.zeros(len(batch["items"]), cp.ITEM_DTYPE)
    for i, it in enumerate(batch["items"]):
        items[i] = (it["x_offset"], it["lX"], it["y_offset"], it["lY"], it["anchor_offset"],
                    it["n_anchors"], it["model"], ragged[0], ragged[1], 0)
    return items


def run_gpu(ctx, batch, bp, mode=0, kernel=0, flags=0, ragged=(0, 0), transitions=None):
    """Returns (list of per-item dicts, Batch).  Uploads the batch's models (ids = list index)."""
    t = transitions if transitions is not None else cp.NANOPORE_TRANSITIONS
    ctx.models_clear()
    ctx.models_create([(t, m, gx, gy) for (m, gx, gy) in batch["models"]])
    b = cp.Batch(ctx, make_items(batch, ragged), batch["x_chars"], batch["events"], batch["anchors"],
                 bp, mode, kernel, flags)
    b.run()
    b.sync()
    npairs, ntot, ncells = b.counts()
    out = []
    for i in range(b.n):
        tri, lp = b.pairs(i, npairs[i])
        xay, tot = b.totals(i, ntot[i])
        out.append(dict(triples=tri, logp=lp, totals_xay=xay, totals=tot, cells=int(ncells[i])))
    return out, b


def run_oracle_item(batch, i, bp, ragged=(0, 0), transitions=None, dump=False, expectations=None):
    it = batch["items"][i]
    m, gx, gy = batch["models"][it["model"]]
    model = o.Sm3Model(m, gy, gx, transitions)
    x = batch["x_chars"][it["x_offset"]: it["x_offset"] + it["lX"] + 5]
    ev = batch["events"][it["y_offset"]: it["y_offset"] + it["lY"]]
    an = batch["anchors"][it["anchor_offset"]: it["anchor_offset"] + it["n_anchors"]]
    p = orc_params(bp, split=1 << 60)  # one item == one getPosteriorProbsWithBanding call
    if dump:
        return o.banded_dump(model, x, it["lX"], ev, an, p, ragged[0], ragged[1])
    r = o.aligned_pairs_using_anchors(model, x, it["lX"], ev, an, p, ragged[0], ragged[1],
                                      expectations=expectations)
    r["triples"] = r["triples"][::-1]  # undo the stList_pop reversal: emission order
    r["logp"] = r["logp"][::-1]
    return r


def assert_same_pairs(g, r, exact_logp=True):
    """GPU result g vs oracle result r for one item: same cells, same order; the exponent
    (F+B)-total is bit-identical, and so is the integer posterior floor(p * 1e7): the device selects pairs by the
    exponent, exp() and the threshold test are finished by the C-ABI layer with the host libm, the one the
    reference (and the oracle) calls (impl/pairwiseAligner.c:776-786)."""
    gt, rt = g["triples"], r["triples"]
    if exact_logp:
        assert len(gt) == len(rt), (len(gt), len(rt))
        assert np.array_equal(g["logp"], r["logp"])
        assert np.array_equal(gt, rt)
    else:
        gd = {(int(x), int(y)): int(p) for p, x, y in gt}
        rd = {(int(x), int(y)): int(p) for p, x, y in rt}
        for k in set(gd) ^ set(rd):  # membership may differ only at the threshold itself
            assert abs((gd.get(k) or rd.get(k)) - 100000) <= 2, k
        for k in set(gd) & set(rd):
            assert abs(gd[k] - rd[k]) <= max(1, 1e-6 * rd[k]), (k, gd[k], rd[k])
