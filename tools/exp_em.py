"""Timing of the Baum-Welch expectation pass (general kernel) on a C3-shaped sub-batch."""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from cpecan_load import binding
cp = binding()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
batch = synth.make_batch(3, n, 5000, 10000, anchor_every=50, distinct_models=1)
ctx = cp.Context(0)
ctx.models_create([(cp.NANOPORE_TRANSITIONS, m, gx, gy) for (m, gx, gy) in batch["models"]])
bp = cp.BandParams(0.01, 1000, 40, 100)
items = np.zeros(len(batch["items"]), cp.ITEM_DTYPE)
for i, it in enumerate(batch["items"]):
    items[i] = (it["x_offset"], it["lX"], it["y_offset"], it["lY"], it["anchor_offset"], it["n_anchors"], it["model"], 1, 1, 0)
for mode, kern, name in ((cp.MODE_EXPECTATIONS, cp.KERNEL_GENERAL, "expectations/general"),
                         (cp.MODE_POSTERIOR, cp.KERNEL_SYSTOLIC, "posterior/systolic"),
                         (cp.MODE_EXPECTATIONS, cp.KERNEL_SYSTOLIC, "expectations/systolic")):
    if kern == cp.KERNEL_GENERAL and n > 256:
        continue
    b = cp.Batch(ctx, items, batch["x_chars"], batch["events"], batch["anchors"], bp, mode, kern, 0)
    b.run(); b.sync()
    t0 = time.perf_counter(); b.run(); b.sync(); dt = time.perf_counter() - t0
    cells = int(b.counts()[2].sum())
    print("%s: %d reads, %.1f ms, %.2f Gcells/s" % (name, n, dt * 1e3, cells / dt / 1e9))
    b.close()
