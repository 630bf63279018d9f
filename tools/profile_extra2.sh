#!/bin/bash
# rocprofv3 kernel traces of the last session's paths (nullable value column, skewed keys, nullable key, streams of small-G batches);
# same method as tools/profile_extra.sh.  bash tools/profile_extra2.sh r03 -> gpurun_out/prof_r03/ (copy into profiles/)
R=${1:-r03}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() {  # name, script + args
    local name=$1; shift
    rm -rf /tmp/rp_$name
    PYTHONPATH=$ROOT timeout 200 rocprofv3 --kernel-trace -d /tmp/rp_$name -- python $ROOT/tools/"$@" > /tmp/rp_$name.log 2>&1
    local db=$(find /tmp/rp_$name -name '*.db' | head -1)
    echo "# rocprofv3 --kernel-trace -- python tools/$*"
    grep -v amdgpu.ids /tmp/rp_$name.log | grep -E "ms,|ms " | tail -3 | sed 's/^/# /'
    python $ROOT/tools/rocpd_summary.py "$db" vnm
    echo
}
prof nv nullcol.py 1e9 1e8 > $OUT/${R}_rocprofv3_kernel_stats_nullable_value.txt
prof sk skew.py 1e9 1e8 4 nohint > $OUT/${R}_rocprofv3_kernel_stats_skewed_keys.txt
prof nk nullkey.py 5e8 1e8 > $OUT/${R}_rocprofv3_kernel_stats_nullable_key.txt
{ prof s7 stream2.py 7 2; prof s3 stream2.py 1000 2; } > $OUT/${R}_rocprofv3_kernel_stats_streams_small_g.txt
ls -la $OUT | tail -5
