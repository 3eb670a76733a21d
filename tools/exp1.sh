#!/bin/bash
cd "$(dirname "$0")/.."
run() { timeout -k 10 150 python bench.py --steps 3 --warmup 1 --check 0 --cpu-reads 0 "$@" | python -c "
import json,sys;j=json.loads(sys.stdin.read());r=j['roofline'];print(j['ms_per_step'],r['dominant_kernel']['avg_launch_ms'],r['forward_kernel']['avg_launch_ms'])"; }
echo "base        $(run)"
echo "reads256    $(run --reads 256)"
echo "reads512    $(run --reads 512)"
echo "reads2048   $(run --reads 2048)"
echo "pad1312     $(CPECAN_RING_PAD=1312 run)"
echo "pad33*1280+32 $(CPECAN_RING_PAD=42272 run)"
