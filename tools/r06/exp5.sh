#!/bin/bash
# round 6, GPU session 5: the whole GPU suite, then the round's profiles (kernel stats + PMC traffic)
mkdir -p gpurun_out/r06
O=gpurun_out/r06
timeout 1800 python -m pytest tests -x -q -m gpu > $O/pytest_all.txt 2>&1
tail -12 $O/pytest_all.txt
timeout 2400 bash tools/profile.sh r06 > $O/profile.log 2>&1
tail -5 $O/profile.log
cat gpurun_out/prof_r06/r06_rocprofv3_kernel_stats_groupbygroups1e8.txt | head -12
cat gpurun_out/prof_r06/r06_traffic.json | head -40
