#!/bin/bash
# Where does a box stand: the wave kernels' pass time, the timeline of a pass, the sweeps' clock, and what the
# management interface says about the card (clocks, power cap, temperature, partition mode) while it is idle and loaded.
out=gpurun_out/box_diag; mkdir -p $out
rocm-smi --showclocks --showpower --showmaxpower --showtemp --showperflevel --showmemuse > $out/smi_idle.txt 2>&1
( sleep 6; rocm-smi --showclocks --showpower --showtemp > $out/smi_load.txt 2>&1 ) &
CPECAN_TIMELINE=1 timeout -k 10 200 python bench.py --steps 30 --warmup 3 --check 0 --cpu-reads 0 --inflight 1 --family wave --single-steps 0 --no-finalise 2> $out/timeline.err > $out/bench.json
wait
python - <<'PY'
import json
j = json.loads(open("gpurun_out/box_diag/bench.json").read().strip().splitlines()[-1])
r = j["roofline"]
print("ms_per_step", j["ms_per_step"], "clock", j["config"]["shader_clock_mhz_in_timed_region"], "bwd", r["backward_kernel"]["avg_launch_ms"], "fwd", r["forward_kernel"]["avg_launch_ms"])
PY
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 tools/ubench_hbm.hip -o /tmp/ubench_hbm 2>/dev/null && /tmp/ubench_hbm
grep timeline $out/timeline.err | tail -16 | head -6
grep -i "sclk\|mclk\|power\|temp" $out/smi_load.txt | head -12
