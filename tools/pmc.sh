#!/bin/bash
# Collect PMC counters for the bench in separate passes (rocprofv3 --pmc with --kernel-trace only).
# usage: bash tools/pmc.sh TAG "CTR1 CTR2 ..." "CTR..." ...   (run on the GPU box, from the repo root)
tag=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --kernel-trace --pmc $set -d $root/gpurun_out/${tag}_$i -o run --output-format csv -- \
      python3 $root/bench.py --steps 1 --warmup 1 --check 0 --cpu-reads 0 > $root/gpurun_out/${tag}_$i.log 2>&1 || exit 1
done
python3 $root/tools/pmc_sum.py $root/gpurun_out/${tag}_*/ > $root/gpurun_out/${tag}_summary.txt
cat $root/gpurun_out/${tag}_summary.txt
