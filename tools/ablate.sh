#!/bin/bash
# Timing ablations of the workgroup-per-alignment (systolic) kernels (results are WRONG by construction; timing only).
# Builds every variant of the four-wave object into gpurun_out/abl/ (never over the product library) and runs the
# bench on it through CPECAN_HIP_LIB.
# usage (on the GPU box, from the repo root): bash tools/ablate.sh BARRIER LADD COEF STORE INSTALL EMIT BAND XCH PMPY ...
# (an argument may carry further -D flags: "LADD -DSY_ABLATE_EMIT"); columns: ms per pass, backward and
# forward launch averages with the whole batch in one stream group
root=$(cd "$(dirname "$0")/.." && pwd)
export CPECAN_SYSTOLIC_ROWS=4 # the timing switches are built into the four-wave objects only
cd $root/cpecan-signal_amd
out=$root/gpurun_out/abl; mkdir -p $out; : > $out/result_systolic.txt
for v in NONE "$@"; do
  flag=""; [ "$v" != NONE ] && flag="-DSY_ABLATE_$v"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
      -Wno-unused-function -I../include -Icsrc $flag -c csrc/cpecan_kernel_systolic.hip -o $out/systolic.o || { echo "$v build failed" | tee -a $out/result_systolic.txt; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libcpecan_hip_abl.so csrc/cpecan_hip.o \
      csrc/cpecan_kernel_general.o csrc/cpecan_kernel_general5.o csrc/cpecan_kernel_generalv.o csrc/cpecan_kernel_generalh.o $out/systolic.o \
      csrc/cpecan_kernel_systolic_r1.o csrc/cpecan_kernel_systolic_r2.o csrc/cpecan_kernel_systolic_r3.o \
      csrc/cpecan_kernel_wave_l2.o csrc/cpecan_kernel_wave_l3.o csrc/cpecan_kernel_wave_l4.o csrc/cpecan_kernel_wave_h2.o csrc/cpecan_kernel_wave_h3.o csrc/cpecan_kernel_wave_h4.o csrc/cpecan_kernel_wave_v2.o csrc/cpecan_kernel_wave_v3.o csrc/cpecan_kernel_wave_v4.o csrc/cpecan_geometry.o -lpthread
  r=$(cd $root && CPECAN_HIP_LIB=$out/libcpecan_hip_abl.so CPECAN_SYSTOLIC_GROUPS=1 timeout -k 10 120 python bench.py --steps 12 --warmup 3 --check 0 --cpu-reads 0 --inflight 1 --family workgroup --single-steps 0 2>/dev/null | python -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=j['roofline'];print(j['ms_per_step'],r['backward_kernel']['avg_launch_ms'],r['forward_kernel']['avg_launch_ms'])")
  echo "$v $r" | tee -a $out/result_systolic.txt
done
