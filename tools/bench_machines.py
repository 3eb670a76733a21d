"""Throughput of the general kernels' machines (vanilla, HDP, 5-state DNA; posterior decode and expectations) on
mid-size batches, for the record in DESIGN.md.  Run on the GPU box: python tools/bench_machines.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import pyoracle as o  # noqa: E402  (model construction only)
import synth  # noqa: E402
import test_dna5_gpu as td  # noqa: E402
import test_hdp_gpu as th  # noqa: E402
import test_vanilla_gpu as tv  # noqa: E402
from harness import band_params, cp, make_items  # noqa: E402

ctx = cp.Context(0)
bp = band_params(0.01, 1000, 40, 100)


def timed(make, label):
    for flags, what in ((0, "posterior"), (cp.FLAG_EXPECTATIONS, "expectations")):
        b = make(flags)
        b.run(); b.sync()
        t0 = time.perf_counter(); b.run(); b.sync(); dt = time.perf_counter() - t0
        cells = int(b.counts()[2].sum())
        print("%s %s: %d items, %.1f ms, %.2f Gcells/s" % (label, what, b.n, dt * 1e3, cells / dt / 1e9), flush=True)
        b.close()


n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
batch = synth.make_batch(5, n, 2000, 4000, anchor_every=50)
models = [o.VanillaModel(m, tv.skip_bins(i), gy) for i, (m, _, gy) in enumerate(batch["models"])]
ctx.models_clear()
ctx.modelsv_create([(m.scalars, m.match, m.skip, m.gap_y) for m in models])
timed(lambda f: cp.Batch(ctx, make_items(batch, (1, 1)), batch["x_chars"], batch["events"], batch["anchors"], bp,
                         flags=f, vanilla=True), "vanilla")

nhdp = o.load_nhdp(os.path.join(ROOT, "tests", "golden", "testTemplate.nhdp"))
hb, _ = th.hdp_batch(7, n, 2000, 50, nhdp)
ctx.models_clear()
ctx.modelsh_create([(cp.NANOPORE_TRANSITIONS, nhdp["alphabet"], nhdp["grid"], nhdp["y"], nhdp["slope"],
                     nhdp["kmer_row"])])
timed(lambda f: cp.Batch(ctx, make_items(hb, (1, 1)), hb["x_chars"], hb["events"], hb["anchors"],
                         band_params(0.05 if f else 0.01, 1000, 40, 100), flags=f, hdp=True), "hdp")

rng = np.random.default_rng(3)
m5 = o.Sm5Model()
ctx.models_clear()
ids = ctx.models5_create([(list(m5.c.t), m5.match, m5.gx, m5.gy)])
xs, ys, an = "", "", []
items = np.zeros(n, cp.ITEM_DTYPE)
for i in range(n):
    x, y, pairs = td.evolve(rng, 3000)
    a = pairs[5::50]
    items[i] = (len(xs), len(x), len(ys), len(y), sum(len(q) for q in an), len(a), ids[0], 0, 0, 0)
    xs += x; ys += y; an.append(a)
anchors = np.concatenate(an)
timed(lambda f: cp.Batch(ctx, items, xs, None, anchors, bp, flags=f, y_chars=ys), "dna5")
