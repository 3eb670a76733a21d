"""Summarise gpurun_out/<tag>_* (written by tools/profile_round.sh) into profiles/<tag>_*.

  profiles/<tag>_rocprofv3_kernel_stats.csv   rocprofv3 --kernel-trace --stats table, as is
  profiles/<tag>_rocprofv3_pmc_summary.json   per-kernel counter sums, bytes per pass, calibration
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
go = os.path.join(root, "gpurun_out")
PASSES = 3  # bench.py --steps 2 --warmup 1


def counters(d):
    tot, n = collections.defaultdict(float), collections.defaultdict(int)
    for f in glob.glob(os.path.join(go, d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
            tot[k] += float(r["Counter_Value"])
            n[k] += 1
    return tot, n


out = {"command": "python3 bench.py --steps 2 --warmup 1 --check 0 --cpu-reads 0 (3 passes over the C3 batch)",
       "passes": PASSES}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    tot, n = counters("%s_pmc_%s" % (tag, c))
    out[c] = {k[0]: {"sum_KiB": v, "launches": n[k], "GB_per_pass": v * 1024 / PASSES / 1e9}
              for k, v in sorted(tot.items()) if k[0].startswith("cpecan")}
    out[c + "_GB_per_pass"] = sum(x["GB_per_pass"] for x in out[c].values())
    cal, _ = counters("%s_cal_%s" % (tag, c))
    out[c + "_calibration"] = {k[0]: {"reported_bytes": v * 1024, "true_bytes": 4 * 2 ** 30,
                                       "reported_over_true": v * 1024 / (4 * 2 ** 30)}
                               for k, v in sorted(cal.items()) if k[0].startswith("k_")}
tot, n = counters("%s_pmc_SQ_INSTS_VALU" % tag)
sq = collections.defaultdict(dict)
for (k, c), v in tot.items():
    if k.startswith("cpecan_k_sy"):
        sq[k][c] = v
        sq[k]["launches"] = n[(k, c)]
for k, d in sq.items():
    if "SQ_WAVE_CYCLES" in d:
        # the SQ pass runs one batch in one stream group: 1024 workgroups on 256 CUs, 4 workgroups per CU for the
        # whole launch, i.e. as many waves per SIMD as a workgroup has waves (3 in the _r3 build, else 4)
        per_simd = 3.0 if k.endswith("_r3") else 4.0
        d["waves_per_simd"] = per_simd
        d["valu_busy_fraction_of_simd_time"] = d["SQ_ACTIVE_INST_VALU"] / (d["SQ_WAVE_CYCLES"] / per_simd)
out["SQ"] = sq
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(root, "profiles", "%s_rocprofv3_pmc_summary.json" % tag), "w"), indent=1)
for f in glob.glob(os.path.join(go, "%s_stats" % tag, "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(root, "profiles", "%s_rocprofv3_kernel_stats.csv" % tag))
print(json.dumps({k: out[k] for k in out if k.endswith("per_pass") or k.endswith("calibration")}, indent=1))
