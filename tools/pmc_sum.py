"""Sum rocprofv3 counter_collection.csv files per (kernel, counter); print per-kernel totals."""
import csv, glob, sys, collections
tot = collections.defaultdict(float); n = collections.defaultdict(int)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
            tot[k] += float(r["Counter_Value"]); n[k] += 1
for (k, c) in sorted(tot):
    if k.startswith("cpecan_k_sy_f") or k.startswith("cpecan_k_sy_b") or k.startswith("cpecan_k_wv_"):
        print("%-24s %-34s %18.0f  launches %d" % (k, c, tot[(k, c)], n[(k, c)]))
