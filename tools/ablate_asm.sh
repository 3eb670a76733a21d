#!/bin/bash
# Timing ablations of the assembly sweeps (results are WRONG by construction; timing only).  Generates every variant of
# the code object into gpurun_out/abl_asm/ and links a side library there (never over the product library); the bench
# loads it through CPECAN_HIP_LIB.
# usage (on the GPU box, from the repo root): bash tools/ablate_asm.sh NONE COEF0 NOLDS HALFLDS NOEVENTS NOSTORE "NOLDS NOEVENTS" ...
root=$(cd "$(dirname "$0")/.." && pwd)
cd $root/cpecan-signal_amd
out=$root/gpurun_out/abl_asm; mkdir -p $out; : > $out/result.txt
LLVM=/opt/rocm/lib/llvm/bin
for v in "$@"; do
  a="$v"; [ "$v" = NONE ] && a=""
  CPECAN_ASM_ABLATE="$a" python3 csrc/asm/gen_sweeps.py $out/sweeps.s $out/cpecan_asm_gen.h 2> /dev/null || { echo "$v generation failed" | tee -a $out/result.txt; continue; }
  $LLVM/clang -x assembler -target amdgcn-amd-amdhsa -mcpu=gfx950 -c $out/sweeps.s -o $out/sweeps.o && $LLVM/ld.lld -shared $out/sweeps.o -o $out/sweeps.hsaco || { echo "$v assembly failed" | tee -a $out/result.txt; continue; }
  python3 -c "import sys; d = open(sys.argv[1], 'rb').read(); open(sys.argv[2], 'w').write(''.join('%d,%s' % (b, chr(10) if i % 32 == 31 else '') for i, b in enumerate(d)) + chr(10))" $out/sweeps.hsaco $out/cpecan_sweeps_hsaco.inc
  cp csrc/cpecan_asm.hip $out/cpecan_asm.hip   # (so that its #include of the code object finds the variant's)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function \
      -I../include -Icsrc -c $out/cpecan_asm.hip -o $out/cpecan_asm.o || { echo "$v build failed" | tee -a $out/result.txt; continue; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function \
      -DCPECAN_TIMING_BUILD -I../include -Icsrc -c csrc/cpecan_hip.hip -o $out/cpecan_hip.o || { echo "$v build failed" | tee -a $out/result.txt; continue; }
  objs="$(ls csrc/*.o | grep -v "cpecan_asm.o\|cpecan_hip.o") $out/cpecan_hip.o"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libcpecan_hip_abl.so $objs $out/cpecan_asm.o -lpthread
  r=$(cd $root && CPECAN_TIMELINE=1 CPECAN_HIP_LIB=$out/libcpecan_hip_abl.so timeout -k 10 120 python bench.py --steps 8 --warmup 3 --check 0 --cpu-reads 0 --inflight 1 --family wave --single-steps 0 --no-finalise $BENCH_ARGS 2> $out/err.txt | python -c "
import json,sys;j=json.loads(sys.stdin.read().strip().splitlines()[-1]);r=j['roofline'];print(j['ms_per_step'],r['backward_kernel']['avg_launch_ms'],r['forward_kernel']['avg_launch_ms'])")
  tl=$(grep "window  7:" $out/err.txt | tail -1)
  echo "$v $r | $tl" | tee -a $out/result.txt
done
