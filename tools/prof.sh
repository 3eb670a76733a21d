#!/bin/bash
# In-kernel cycle profile of the forward step (timing build; run on the GPU box).
set -e
export CPECAN_SYSTOLIC_ROWS=4 # the timing switches are built into the four-wave objects only
cd "$(dirname "$0")/../cpecan-signal_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math \
    -Wno-unused-function -I../include -Icsrc -DSY_PROFILE "$@" -c csrc/cpecan_kernel_systolic.hip -o csrc/cpecan_kernel_systolic.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libcpecan_hip.so csrc/cpecan_hip.o \
    csrc/cpecan_kernel_general.o csrc/cpecan_kernel_general5.o csrc/cpecan_kernel_generalv.o csrc/cpecan_kernel_generalh.o csrc/cpecan_kernel_systolic.o csrc/cpecan_kernel_systolic_r1.o csrc/cpecan_kernel_systolic_r2.o csrc/cpecan_kernel_systolic_r3.o csrc/cpecan_geometry.o -lpthread
cd .. && CPECAN_PROF=1 timeout -k 10 120 python bench.py --steps 1 --warmup 0 --check 0 --cpu-reads 0 $BENCH_ARGS 2>&1 | grep -v amdgpu.ids 
