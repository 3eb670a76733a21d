#!/bin/bash
# Same-box comparison of library builds (boxes differ by +-25 %): ab/lib<NAME>_hip.so for every NAME given, alternating,
# `rounds` times.  usage (on the GPU box): tools/ab_bench.sh OUTDIR ROUNDS NAME... [-- bench.py arguments]
out=$1; rounds=$2; shift 2
names=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do names+=("$1"); shift; done
[ "$1" == "--" ] && shift
args=("$@")
[ ${#args[@]} -eq 0 ] && args=(--family wave --inflight 1 --steps 6 --warmup 2 --single-steps 0 --check 0 --cpu-reads 0)
mkdir -p "$out"
for r in $(seq 1 "$rounds"); do
    for n in "${names[@]}"; do
        CPECAN_HIP_LIB=$PWD/ab/lib${n}_hip.so timeout -k 10 200 python bench.py "${args[@]}" > "$out/$n.$r.log" 2>&1 || exit 1
        python - "$out/$n.$r.log" "$n" "$r" <<'PY' | tee -a "$out/summary.txt"
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")][-1]
j = json.loads(l)
print(sys.argv[2], sys.argv[3], "ms_per_step", j["ms_per_step"], "latency", j["config"].get("step_latency_ms"), "clock",
      j["config"].get("shader_clock_mhz_in_timed_region"), "B", j["roofline"]["backward_kernel"]["avg_launch_ms"], "F",
      j["roofline"]["forward_kernel"]["avg_launch_ms"])
PY
    done
done
