#!/bin/bash
# Timing ablations of the systolic kernels (results are WRONG by construction; timing only).
# usage (on the GPU box): bash tools/ablate.sh BARRIER LADD COEF STORE INSTALL EMIT BAND XCH PMPY ...
# (an argument may carry further -D flags: "LADD -DSY_ABLATE_EMIT"); columns: ms per pass, backward and
# forward launch averages with the whole batch in one stream group
set -e
export CPECAN_SYSTOLIC_ROWS=4 # the timing switches are built into the four-wave objects only
cd "$(dirname "$0")/../cpecan-signal_amd"
out=../gpurun_out/ablate.txt
: > $out
for v in NONE "$@"; do
  flag=""; [ "$v" != NONE ] && flag="-DSY_ABLATE_$v"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
      -Wno-unused-function -I../include -Icsrc $flag -c csrc/cpecan_kernel_systolic.hip -o csrc/cpecan_kernel_systolic.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libcpecan_hip.so csrc/cpecan_hip.o \
      csrc/cpecan_kernel_general.o csrc/cpecan_kernel_general5.o csrc/cpecan_kernel_generalv.o csrc/cpecan_kernel_generalh.o csrc/cpecan_kernel_systolic.o csrc/cpecan_kernel_systolic_r1.o csrc/cpecan_kernel_systolic_r2.o csrc/cpecan_kernel_systolic_r3.o csrc/cpecan_geometry.o -lpthread
  r=$(cd .. && CPECAN_SYSTOLIC_GROUPS=1 timeout -k 10 120 python bench.py --steps 3 --warmup 1 --check 0 --cpu-reads 0 | python -c "
import json,sys;j=json.loads(sys.stdin.read());r=j['roofline'];print(j['ms_per_step'],r['dominant_kernel']['avg_launch_ms'],r['forward_kernel']['avg_launch_ms'])")
  echo "$v $r" | tee -a $out
done
