#!/bin/bash
mkdir -p gpurun_out/r06
bash tools/pmc_kernels.sh xsort --workload topk --limit 0 > gpurun_out/r06/r06_pmc_sq_sort.txt 2>&1
cat gpurun_out/r06/r06_pmc_sq_sort.txt | grep -E "xsort_scatter|xsort_local" | cut -c1-160
