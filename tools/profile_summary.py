"""Summarise gpurun_out/<tag>_* (written by tools/profile_round.sh) into profiles/<tag>_*.

  profiles/<tag>_rocprofv3_kernel_stats.csv   rocprofv3 --kernel-trace --stats table, as is
  profiles/<tag>_rocprofv3_pmc_summary.json   per-kernel counter sums, bytes per pass, calibration; one section per
                                              kernel family (wave-per-alignment, workgroup-per-alignment)
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
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


def family(fam):
    out = {"command": "%spython3 bench.py --steps 2 --warmup 1 --check 0 --cpu-reads 0 --inflight 1 --single-steps 0 "
                      "--no-finalise --family %s (3 passes over the C3 batch)"
                      % ("CPECAN_ASM=0 " if fam == "wave" else "", "wave" if fam == "assembly" else fam), "passes": PASSES}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        tot, n = counters("%s_pmc_%s_%s" % (tag, fam, c))
        out[c] = {k[0]: {"sum_KiB": v, "launches": n[k], "GB_per_pass": v * 1024 / PASSES / 1e9}
                  for k, v in sorted(tot.items()) if k[0].startswith("cpecan")}
        out[c + "_GB_per_pass"] = sum(x["GB_per_pass"] for x in out[c].values())
    tot, n = counters("%s_pmc_%s_SQ" % (tag, fam))
    sq = collections.defaultdict(dict)
    for (k, c), v in tot.items():
        if k.startswith(("cpecan_k_sy", "cpecan_k_wv", "cpecan_k_asm")):
            sq[k][c] = v
            sq[k]["launches"] = n[(k, c)]
    for k, d in sq.items():
        if "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"] > 0:
            # workgroup family: 1024 workgroups on 256 CUs = as many waves per SIMD as a workgroup has waves for the
            # whole launch; wave family: 1024 one-wave workgroups = one wave of this kernel per SIMD (the forward and
            # the backward kernel of neighbouring windows run together: their busy fractions add up per SIMD)
            per_simd = 1.0 if k.startswith(("cpecan_k_wv", "cpecan_k_asm")) else 3.0 if k.endswith("_r3") else 2.0 if k.endswith("_r2") \
                else 1.0 if k.endswith("_r1") else 4.0
            d["waves_per_simd"] = per_simd
            d["valu_busy_fraction_of_simd_time"] = d["SQ_ACTIVE_INST_VALU"] / (d["SQ_WAVE_CYCLES"] / per_simd)
            d["valu_instructions_per_wave_cycle"] = d["SQ_INSTS_VALU"] / d["SQ_WAVE_CYCLES"]
    out["SQ"] = sq
    return out


out = {"families": {f: family(f) for f in ("assembly", "wave", "workgroup")}}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    cal, _ = counters("%s_cal_%s" % (tag, c))
    out[c + "_calibration"] = {k[0]: {"reported_bytes": v * 1024, "true_bytes": 4 * 2 ** 30,
                                       "reported_over_true": v * 1024 / (4 * 2 ** 30)}
                               for k, v in sorted(cal.items()) if k[0].startswith("k_")}
os.makedirs(os.path.join(root, "profiles"), exist_ok=True)
json.dump(out, open(os.path.join(root, "profiles", "%s_rocprofv3_pmc_summary.json" % tag), "w"), indent=1)
for f in glob.glob(os.path.join(go, "%s_stats" % tag, "**", "*kernel_stats.csv"), recursive=True):
    shutil.copy(f, os.path.join(root, "profiles", "%s_rocprofv3_kernel_stats.csv" % tag))
print(json.dumps({f: {k: v for k, v in out["families"][f].items() if k.endswith("per_pass")} for f in out["families"]},
                 indent=1))
for f in out["families"]:
    for k, d in out["families"][f]["SQ"].items():
        print(f, k, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in d.items()
                     if a in ("launches", "valu_busy_fraction_of_simd_time", "valu_instructions_per_wave_cycle")})
