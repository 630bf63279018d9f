#!/bin/bash
# rocprofv3 kernel traces of the round's later paths (dictionary-coded keys, wide entries with many columns, heavy-key spill),
# same method as tools/profile.sh: kernel trace only, summary from the rocpd database.
R=${1:-r03}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$R
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() {  # name, script + args
    local name=$1; shift
    rm -rf /tmp/rp_$name
    PYTHONPATH=$ROOT timeout 150 rocprofv3 --kernel-trace -d /tmp/rp_$name -- python $ROOT/tools/"$@" > /tmp/rp_$name.log 2>&1
    local db=$(find /tmp/rp_$name -name '*.db' | head -1)
    echo "# rocprofv3 --kernel-trace -- python tools/$*"
    tail -1 /tmp/rp_$name.log | sed 's/^/# /'
    python $ROOT/tools/rocpd_summary.py "$db" vnm
    echo
}
prof mk multikey.py 1e9 7 3 wide > $OUT/${R}_rocprofv3_kernel_stats_multikey_wide.txt
prof mc manycol.py 5e8 1e6 4 > $OUT/${R}_rocprofv3_kernel_stats_manycol4.txt
prof hk heavykey.py 2e8 1e6 0.5 sum2 value > $OUT/${R}_rocprofv3_kernel_stats_heavykey.txt
if [ "$R" != "r02" ]; then   # round 3: split programs, the tuple dictionary, the string dictionary, the sample sort with heavy values
prof mc8 manycol.py 5e8 1e6 8 > $OUT/${R}_rocprofv3_kernel_stats_manycol8_split.txt
prof wk widekey.py 2e8 2e7 > $OUT/${R}_rocprofv3_kernel_stats_widekey_tuple_dictionary.txt
prof sd strdict_bench.py 2e7 > $OUT/${R}_rocprofv3_kernel_stats_string_dictionary.txt
prof so sortbench.py 1e9 > $OUT/${R}_rocprofv3_kernel_stats_sort_heavy_values.txt
fi
cd $ROOT && cp $OUT/${R}_rocprofv3_kernel_stats_{manycol8_split,widekey_tuple_dictionary,string_dictionary,sort_heavy_values}.txt profiles/ 2>/dev/null
ls -la $OUT | tail -6
