#!/bin/bash
# Where the assembly sweeps' time goes, from the hardware counters: one rocprofv3 --pmc pass per counter group (never with
# other trace domains), one batch at a time on the wave family.  Run on the GPU box from the repo root; outputs under
# gpurun_out/$1_pmc_*; tools/pmc_summary.py folds them into one table per kernel.
tag=${1:-r03}
root=$(pwd)
cd /tmp && export TMPDIR=/tmp
B="python3 $root/bench.py --steps 2 --warmup 1 --check 0 --cpu-reads 0 --inflight 1 --single-steps 0 --no-finalise --family wave"
i=0
while read -r group; do
  [ -z "$group" ] && continue
  i=$((i + 1))
  timeout -k 5 200 rocprofv3 --kernel-trace --pmc $group -d $root/gpurun_out/${tag}_pmc_$i -o run --output-format csv -- $B > $root/gpurun_out/${tag}_pmc_$i.log 2>&1 || { echo "group $i failed: $group"; tail -3 $root/gpurun_out/${tag}_pmc_$i.log; }
  echo "group $i done: $group"
done <<'GROUPS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_VMEM_RD SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA
SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_VMEM SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL
TA_TA_BUSY TA_ADDR_STALLED_BY_TC_CYCLES TA_DATA_STALLED_BY_TC_CYCLES TA_ADDR_STALLED_BY_TD_CYCLES GRBM_GUI_ACTIVE
TCC_EA0_WRREQ_STALL TCC_EA0_WRREQ TCC_EA0_RDREQ TCC_EA0_WRREQ_DRAM_CREDIT_STALL TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_BUSY TCC_CYCLE
TCC_HIT TCC_MISS TCC_REQ TCC_READ TCC_WRITE
FETCH_SIZE
WRITE_SIZE
GROUPS
echo all done
