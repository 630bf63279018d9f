#!/bin/bash
# Issue / LDS / wait counters per kernel of a bench workload: bash tools/pmc_kernels.sh <name-filter> <bench args...>
# (GPU box; one rocprofv3 pass per counter group; instances are shader engines: n = dispatches x 32)
R=$(pwd); FLT=$1; shift
cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM"; do
  rm -rf /tmp/rp; timeout 300 rocprofv3 --pmc $grp --kernel-trace -d /tmp/rp -- python $R/bench.py --no-cpu-baseline --no-also --steps 2 --warmup 1 "$@" > /tmp/rp.log 2>&1
  db=$(find /tmp/rp -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py $db $FLT | grep -E "avg=" 
done
