#!/bin/bash
# Timing ablations of the wave kernels (results are WRONG by construction; timing only).  Builds every variant of the
# three-cells-per-lane object into gpurun_out/abl/ (never over the product library) and runs the bench on it.
# usage (on the GPU box, from the repo root): bash tools/ablate_wave.sh NONE FSTORE FFEED BLOAD BFEED LADD EMIT "LADD -DWV_ABL_EMIT" ...
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/cpecan-signal_amd
out=$root/gpurun_out/abl; mkdir -p $out; : > $out/result.txt
for v in "$@"; do
  flag=""; [ "$v" != NONE ] && flag="-DWV_ABL_$v"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function \
      -I../include -Icsrc -DWV_L=3 $flag -c csrc/cpecan_kernel_wave.hip -o $out/wave_l3.o || { echo "$v build failed" | tee -a $out/result.txt; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libcpecan_hip_abl.so csrc/cpecan_hip.o csrc/cpecan_kernel_general.o \
      csrc/cpecan_kernel_general5.o csrc/cpecan_kernel_generalv.o csrc/cpecan_kernel_generalh.o csrc/cpecan_kernel_systolic.o csrc/cpecan_kernel_wave5.o \
      csrc/cpecan_kernel_systolic_r1.o csrc/cpecan_kernel_systolic_r2.o csrc/cpecan_kernel_systolic_r3.o \
      csrc/cpecan_kernel_wave_l2.o $out/wave_l3.o csrc/cpecan_kernel_wave_l4.o csrc/cpecan_kernel_wave_h2.o csrc/cpecan_kernel_wave_h3.o csrc/cpecan_kernel_wave_h4.o csrc/cpecan_kernel_wave_v2.o csrc/cpecan_kernel_wave_v3.o csrc/cpecan_kernel_wave_v4.o csrc/cpecan_geometry.o -lpthread
  r=$(cd $root && CPECAN_HIP_LIB=$out/libcpecan_hip_abl.so timeout -k 10 120 python bench.py --steps 12 --warmup 3 --check 0 --cpu-reads 0 --inflight 1 --family wave --single-steps 0 --no-finalise $BENCH_ARGS 2>/dev/null | python -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=j['roofline'];print(j['ms_per_step'],r['backward_kernel']['avg_launch_ms'],r['forward_kernel']['avg_launch_ms'])")
  echo "$v $r" | tee -a $out/result.txt
done
