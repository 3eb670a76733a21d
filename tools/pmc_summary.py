#!/usr/bin/env python3
"""Folds the counter passes of tools/pmc_sweeps.sh into one table: per kernel, the sum of every counter over its
dispatches and the number of dispatches.  usage: tools/pmc_summary.py gpurun_out/r03a_pmc_ > profiles/r03_pmc_sweeps.txt"""
import collections
import csv
import glob
import sys

prefix = sys.argv[1]
tab = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.defaultdict(lambda: collections.defaultdict(set))
for path in sorted(glob.glob(prefix + "*/run_counter_collection.csv")):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        tab[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(tab, key=lambda k: -tab[k].get("SQ_WAVE_CYCLES", 0)):
    print(k)
    for c in sorted(tab[k]):
        n = len(calls[k][c])
        print("    %-34s %18.0f  over %4d dispatches  (%.4g per dispatch)" % (c, tab[k][c], n, tab[k][c] / n))
