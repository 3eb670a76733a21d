#!/bin/bash
cd "$(dirname "$0")/.."
run() { timeout -k 10 150 python bench.py --steps 4 --warmup 1 --check 0 --cpu-reads 0 "$@" 2>/dev/null | python -c "
import json,sys;j=json.loads(sys.stdin.read());r=j['roofline'];print(j['ms_per_step'],r['dominant_kernel']['avg_launch_ms'],r['forward_kernel']['avg_launch_ms'])"; }
for q in 4 8; do for g in 2 4; do echo "hwq=$q groups=$g  $(GPU_MAX_HW_QUEUES=$q CPECAN_SYSTOLIC_GROUPS=$g run)"; done; done
echo "default groups=2 $(run)"
