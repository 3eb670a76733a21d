"""E-step of the EM driver on a C3-shaped batch: one context against two (two concurrent batches)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import torch  # noqa
import numpy as np
import synth
from cpecan_load import binding, em
cp, E = binding(), em()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
batch = synth.make_batch(3, n, 5000, 10000, anchor_every=50, distinct_models=False)
bp = cp.BandParams(0.01, 1000, 40, 100)
c1, c2 = cp.Context(0), cp.Context(0)
gap0 = batch["models"][0][1]
for label, kw in (("one context", {}), ("two contexts", {"ctx2": c2})):
    E.gpu_e_step(cp, c1, batch, bp, range(n), cp.NANOPORE_TRANSITIONS, gap0, **kw)
    t0 = time.perf_counter()
    v = E.gpu_e_step(cp, c1, batch, bp, range(n), cp.NANOPORE_TRANSITIONS, gap0, **kw)
    print("%s: %.1f ms per E-step of %d reads (likelihood %.3f)" % (label, 1e3 * (time.perf_counter() - t0), n, v[-1]))
keep = E.PersistentEStep(cp, [c1, c2], batch, bp, range(n), cp.NANOPORE_TRANSITIONS, gap0)
keep(cp.NANOPORE_TRANSITIONS, gap0)
t0 = time.perf_counter()
for _ in range(3):
    v = keep(cp.NANOPORE_TRANSITIONS, gap0)
print("persistent, two contexts: %.1f ms per E-step of %d reads (likelihood %.3f)" % (1e3 * (time.perf_counter() - t0) / 3, n, v[-1]))
