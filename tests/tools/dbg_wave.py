"""development aid: run parity cases of tests/test_systolic_gpu.py and print where totals / pairs differ"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import numpy as np
import synth
from harness import band_params, cp, run_gpu, run_oracle_item
from test_systolic_gpu import CASES
ctx = cp.Context(0)
for ci in [int(a) for a in sys.argv[1:]] or range(len(CASES)):
    case = CASES[ci]
    batch = synth.make_batch(21, case["n"], case["lX"], case["lY"], anchor_every=case["every"])
    bp = band_params(0.01, case["md"], case["tb"], case["e"])
    res, b = run_gpu(ctx, batch, bp, kernel=cp.KERNEL_SYSTOLIC, ragged=case["ragged"])
    print("case", ci, case, b.info())
    for i in range(case["n"]):
        ref = run_oracle_item(batch, i, bp, case["ragged"])
        g = res[i]
        ok_xay = np.array_equal(g["totals_xay"], ref["totals_xay"])
        n = min(len(g["totals"]), len(ref["totals"]))
        bad = np.nonzero(g["totals"][:n] != ref["totals"][:n])[0]
        print(" item", i, "cells", g["cells"] == ref["cells"], "xay", ok_xay, "ntot", len(g["totals"]), len(ref["totals"]),
              "bad totals:", len(bad), "at xay", [int(x) for x in ref["totals_xay"][bad][:12]],
              "diff", [float(d) for d in (g["totals"][:n] - ref["totals"][:n])[bad][:6]])
        print("   pairs", len(g["triples"]), len(ref["triples"]),
              "same" if len(g["triples"]) == len(ref["triples"]) and np.array_equal(g["triples"][:, 1:], ref["triples"][:, 1:]) else "DIFF")
    b.close()
