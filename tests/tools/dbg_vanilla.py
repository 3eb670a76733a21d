"""wave-per-alignment vanilla kernels against the general kernel and the oracle on one case: where do totals / pairs part"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import pyoracle as o, synth
import test_vanilla_gpu as tv
from harness import band_params, cp, make_items, orc_params

case = dict(n=3, lX=int(sys.argv[1]) if len(sys.argv) > 1 else 300, lY=int(sys.argv[2]) if len(sys.argv) > 2 else 610,
            e=int(sys.argv[3]) if len(sys.argv) > 3 else 40, md=int(sys.argv[4]) if len(sys.argv) > 4 else 100, tb=40, every=50,
            ragged=(int(sys.argv[5]) if len(sys.argv) > 5 else 1, int(sys.argv[6]) if len(sys.argv) > 6 else 1))
batch = synth.make_batch(51, case["n"], case["lX"], case["lY"], anchor_every=case["every"])
models = []
for i, (match, _, gapy) in enumerate(batch["models"]):
    strand = (np.float32(0.17), np.float32(0.55)) if i % 2 == 0 else (np.float32(0.14), np.float32(0.49))
    models.append(o.VanillaModel(match, tv.skip_bins(i), gapy, float(strand[0]), float(strand[1])))
ctx = cp.Context(0)
ctx.modelsv_create([(m.scalars, m.match, m.skip, m.gap_y) for m in models])
bp = band_params(0.01, case["md"], case["tb"], case["e"])
res = {}
for name, fl in (("wave", 0), ("general", cp.FLAG_GENERAL_KERNEL)):
    b = cp.Batch(ctx, make_items(batch, case["ragged"]), batch["x_chars"], batch["events"], batch["anchors"], bp, flags=fl, vanilla=True)
    print(name, b.info())
    b.run(); b.sync()
    npairs, ntot, ncells = b.counts()
    res[name] = [(b.pairs(i, npairs[i]), b.totals(i, ntot[i])) for i in range(b.n)]
    b.close()
for i in range(case["n"]):
    (tw, lw), (xw, totw) = res["wave"][i]
    (tg, lg), (xg, totg) = res["general"][i]
    bad = np.flatnonzero(totw != totg) if len(totw) == len(totg) else None
    print("item", i, "totals", len(totw), len(totg), "first differing", None if bad is None or bad.size == 0 else (int(bad[0]), int(xw[bad[0]]), totw[bad[0]], totg[bad[0]], totw[bad[0]] - totg[bad[0]]), "n bad", None if bad is None else bad.size)
    if bad is not None and bad.size:
        print("   diagonals of differing totals:", xw[bad][:20])
    same = len(tw) == len(tg) and np.array_equal(tw, tg)
    print("   pairs", len(tw), len(tg), "identical" if same else "DIFFER")
    if not same and len(tw) == len(tg):
        k = np.flatnonzero((tw != tg).any(1))
        print("   first differing pair", k[0], tw[k[0]], tg[k[0]], lw[k[0]], lg[k[0]])
