"""Throughput of the vanilla, HDP and 5-state DNA machines (posterior decode and expectations; wave-per-alignment or
general kernels as the library picks them) on mid-size batches, for the record in DESIGN.md.  Run on the GPU box: python tools/bench_machines.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "oracle")]
import pyoracle as o  # noqa: E402  (model construction only)
import synth  # noqa: E402
import test_dna5_gpu as td  # noqa: E402
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
        info = b.info()
        print("%s %s: %d items, %.1f ms, %.2f Gcells/s (%s kernel, widest band %d)" % (
            label, what, b.n, dt * 1e3, cells / dt / 1e9, info.get("family", info["kernel"]), info["max_band_width"]),
            flush=True)
        b.close()


n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
batch = synth.make_batch(5, n, 2000, 4000, anchor_every=50)
models = [o.VanillaModel(m, tv.skip_bins(i), gy) for i, (m, _, gy) in enumerate(batch["models"])]
ctx.models_clear()
ctx.modelsv_create([(m.scalars, m.match, m.skip, m.gap_y) for m in models])
timed(lambda f: cp.Batch(ctx, make_items(batch, (1, 1)), batch["x_chars"], batch["events"], batch["anchors"], bp,
                         flags=f, vanilla=True), "vanilla")

# the HDP machine: reads drawn as bench.py --config 5 draws them (anchors on k-mers that emitted), through the host
# library's own reader of the reference's serialized HDP
import ctypes as C  # noqa: E402
sys.path.insert(0, ROOT)
import bench  # noqa: E402
host = C.CDLL(os.path.join(ROOT, "cpecan-signal_amd", "libcpecan_host.so"))
host.deserialize_nhdp.restype = C.c_void_p
host.deserialize_nhdp.argtypes = [C.c_char_p]
host.getHdpStateMachine3.restype = C.c_void_p
host.getHdpStateMachine3.argtypes = [C.c_void_p]
host.cpecan_hdp_machine_as_model.argtypes = [C.c_void_p, C.c_void_p]
sm = host.getHdpStateMachine3(host.deserialize_nhdp(os.path.join(ROOT, "tests", "golden", "testTemplate.nhdp").encode()))
desc = cp.HdpModelDesc()
host.cpecan_hdp_machine_as_model(sm, C.byref(desc))
hb = bench.hdp_reads(n, 2000, 4000, 7, desc)
ctx.models_clear()
ids = np.zeros(1, np.int32)
assert cp.lib().cpecan_hip_modelsh_create(ctx.h, C.byref(desc), 1, ids.ctypes.data_as(C.c_void_p)) == 0
timed(lambda f: cp.Batch(ctx, bench.make_items(cp, hb), hb["x_chars"], hb["events"], hb["anchors"],
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
# ... and with the reference's default band for DNA (diagonalExpansion 20, inc/pairwiseAligner.h defaults)
timed(lambda f: cp.Batch(ctx, items, xs, None, anchors, band_params(0.01, 1000, 40, 20), flags=f, y_chars=ys),
      "dna5 (expansion 20)")
if os.environ.get("BENCH_MACHINES_NARROW"):  # per-diagonal overhead: bands a handful of cells wide
    timed(lambda f: cp.Batch(ctx, items, xs, None, anchors, band_params(0.01, 1000, 40, 2), flags=f, y_chars=ys),
          "dna5 (expansion 2)")
