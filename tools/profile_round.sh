#!/bin/bash
# Round profile, run on the GPU box from the repo root: kernel stats, HBM traffic counters (separate
# --pmc passes, never combined with other trace domains), SQ utilisation counters, and the
# calibration of FETCH_SIZE/WRITE_SIZE for 8-byte-per-lane accesses.  Outputs under gpurun_out/$1_*.
tag=${1:-r03}
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
# kernel durations: the bench's own default command line shape (pipelined timed region on the wave kernels' assembly
# sweeps, then one batch alone on each family), fewer steps
timeout -k 5 300 rocprofv3 --kernel-trace --stats -d $root/gpurun_out/${tag}_stats -o run --output-format csv -- python3 $root/bench.py --steps 6 --warmup 2 --single-steps 6 --check 0 --cpu-reads 0 > $root/gpurun_out/${tag}_stats.log 2>&1 || exit 1
# counters: one batch at a time, one family at a time, so that a pass (--steps 2 --warmup 1 = 3 passes) is the
# unit the byte counts are divided by
# (assembly: the wave family as it runs by default, the hand-scheduled sweeps; wave: its compiled kernels, CPECAN_ASM=0)
for fam in assembly wave workgroup; do
  kf=$fam; unset CPECAN_ASM
  [ $fam = assembly ] && kf=wave
  [ $fam = wave ] && export CPECAN_ASM=0
  B="python3 $root/bench.py --steps 2 --warmup 1 --check 0 --cpu-reads 0 --inflight 1 --single-steps 0 --no-finalise --family $kf"
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c -d $root/gpurun_out/${tag}_pmc_${fam}_$c -o run --output-format csv -- $B > $root/gpurun_out/${tag}_pmc_${fam}_$c.log 2>&1 || exit 1
  done
  # SQ utilisation with the whole batch in one stream group
  export CPECAN_SYSTOLIC_GROUPS=1
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $root/gpurun_out/${tag}_pmc_${fam}_SQ -o run --output-format csv -- $B > $root/gpurun_out/${tag}_pmc_${fam}_SQ.log 2>&1 || exit 1
  unset CPECAN_SYSTOLIC_GROUPS
done
unset CPECAN_ASM
hipcc --offload-arch=gfx950 -O2 $root/tools/ubench_fetch.hip -o /tmp/ubf 2>/dev/null || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $c -d $root/gpurun_out/${tag}_cal_$c -o run --output-format csv -- /tmp/ubf > $root/gpurun_out/${tag}_cal_$c.log 2>&1 || exit 1
done
echo done
