#!/bin/bash
# LDS / issue counters of the scan kernel: bash tools/pmc_hot.sh <G> <funcs>   (GPU box; separate passes per counter group)
R=$(pwd); G=$1; F=$2
cd /tmp; export TMPDIR=/tmp PYTHONPATH=$R
for grp in "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_BUSY_CYCLES" "SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_LDS_UNALIGNED_STALL" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
  rm -rf /tmp/rp; timeout 200 rocprofv3 --pmc $grp --kernel-trace -d /tmp/rp -- python $R/tools/hotfuncs.py 1e9 $G $F > /tmp/rp.log 2>&1
  db=$(find /tmp/rp -name "*.db" | head -1)
  [ -n "$db" ] && python $R/tools/rocpd_summary.py $db agg_hot | grep -E "avg=|avg_us|agg_hot" | grep -v "^kernel"
done
