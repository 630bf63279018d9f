#!/bin/bash
# Regenerates the rocprofv3 summaries kept under profiles/ (run on the GPU box: gpurun -- 'bash tools/profile.sh r01').
# Kernel trace and each PMC counter are collected in SEPARATE runs (MI355X_MICROARCH.md, HBM section).
R=${1:-r05}
ONLY=${2:-}     # optional: only the workloads whose argument string contains this (e.g. 'count_star')
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() {  # name, rocprof args..., -- bench args
    local name=$1; shift
    local pargs=(); while [ "$1" != "--" ]; do pargs+=("$1"); shift; done; shift
    rm -rf /tmp/rp_$name
    timeout 600 rocprofv3 "${pargs[@]}" -d /tmp/rp_$name -- python $ROOT/bench.py --no-cpu-baseline --no-also --no-check "$@" > /tmp/rp_$name.log 2>&1
    local db=$(find /tmp/rp_$name -name '*.db' | head -1)
    echo "# rocprofv3 ${pargs[*]} -- python bench.py --no-cpu-baseline --no-also --no-check $*"
    python $ROOT/tools/rocpd_summary.py "$db" vnm
    echo
}
for wl in "groupby --groups 1e8" "groupby --groups 7" "groupby --groups 1e3" "groupby --groups 1e5" "groupby --groups 1e6" \
          "groupby --groups 1e8 --shape count_star" "groupby --groups 1e8 --shape minmax" "stream --groups 1e6" "stream --groups 7" "filter" "topk" "topk --limit 0" "project"; do
    [[ -n "$ONLY" && "$wl" != *"$ONLY"* ]] && continue
    tag=$(echo $wl | tr -d ' -' ); 
    prof ks_$tag --kernel-trace -- --workload $wl --steps 5 --warmup 2 > $OUT/${R}_rocprofv3_kernel_stats_$tag.txt
done
for wl in "groupby --groups 1e8" "filter" "groupby --groups 1e6" "stream --groups 1e6" "topk --limit 0" "groupby --groups 1e8 --shape count_star"; do
    [[ -n "$ONLY" && "$wl" != *"$ONLY"* ]] && continue
    tag=$(echo $wl | tr -d ' -' )
    { prof pf_$tag --pmc FETCH_SIZE --kernel-trace -- --workload $wl --steps 2 --warmup 1
      prof pw_$tag --pmc WRITE_SIZE --kernel-trace -- --workload $wl --steps 2 --warmup 1; } > $OUT/${R}_rocprofv3_pmc_$tag.txt
done
# several input columns (tools/manycol.py N G C): the few-groups scan (agg_hotn_kernel) and the per-column / per-pair dense split
for mc in "5e8 7 4" "5e8 7 6" "5e8 1e6 3" "5e8 1e8 3"; do
    [[ -n "$ONLY" && "manycol $mc" != *"$ONLY"* ]] && continue
    tag=manycol_$(echo $mc | tr ' ' '_')
    rm -rf /tmp/rp_$tag
    timeout 600 rocprofv3 --kernel-trace -d /tmp/rp_$tag -- python $ROOT/tools/manycol.py $mc > /tmp/rp_$tag.log 2>&1
    { echo "# rocprofv3 --kernel-trace -- python tools/manycol.py $mc   (3 repetitions; SELECT k, sum(c1..cC), count(*) GROUP BY k)"
      python $ROOT/tools/rocpd_summary.py "$(find /tmp/rp_$tag -name '*.db' | head -1)" vnm; tail -1 /tmp/rp_$tag.log; } > $OUT/${R}_rocprofv3_kernel_stats_$tag.txt
done
cd $ROOT && cp $OUT/${R}_*.txt profiles/ 2>/dev/null; python tools/traffic_from_pmc.py $R; cp profiles/${R}_traffic.json $OUT/; ls -la $OUT
