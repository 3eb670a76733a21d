"""development aid: one parity case of tests/test_systolic_gpu.py through the assembly sweeps, step by step
usage: dbg_asm.py CASE [create|run]   (CPECAN_ASM=0/1/unset picks compiled / forward only / both)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "oracle"))
import numpy as np
import synth
from harness import band_params, cp, make_items, run_oracle_item
from test_systolic_gpu import CASES
ci = int(sys.argv[1])
mode = sys.argv[2] if len(sys.argv) > 2 else "run"
case = CASES[ci]
ctx = cp.Context(0)
batch = synth.make_batch(21, case["n"], case["lX"], case["lY"], anchor_every=case["every"])
bp = band_params(0.01, case["md"], case["tb"], case["e"])
ctx.models_clear()
ctx.models_create([(cp.NANOPORE_TRANSITIONS, m, gx, gy) for (m, gx, gy) in batch["models"]])
b = cp.Batch(ctx, make_items(batch, case["ragged"]), batch["x_chars"], batch["events"], batch["anchors"], bp, 0, cp.KERNEL_SYSTOLIC, 0)
print("created", b.info(), flush=True)
if mode == "create":
    sys.exit(0)
b.run()
b.sync()
print("ran", flush=True)
npairs, ntot, ncells = b.counts()
for i in range(b.n):
    ref = run_oracle_item(batch, i, bp, case["ragged"])
    tri, lp = b.pairs(i, npairs[i])
    xay, tot = b.totals(i, ntot[i])
    n = min(len(tot), len(ref["totals"]))
    bad = np.nonzero(tot[:n] != ref["totals"][:n])[0]
    print(" item", i, "ntot", len(tot), len(ref["totals"]), "bad totals", len(bad), [int(x) for x in ref["totals_xay"][bad][:8]],
          "pairs", len(tri), len(ref["triples"]),
          "same" if len(tri) == len(ref["triples"]) and np.array_equal(tri[:, 1:], ref["triples"][:, 1:]) else "DIFF", flush=True)
