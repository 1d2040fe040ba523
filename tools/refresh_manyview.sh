#!/bin/bash
# tools/refresh_manyview.sh [round]: the many-view part of tools/refresh_profiles.sh alone (48 x 960x540 on the 32- and 64-view builds:
# bench line + SQ counters each), into gpurun_out/profiles_new/
set -e
R=${1:-r04}
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_new
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-config5 --no-config4 --cpu-seconds 0"
M="--views 48 --width 960 --height 540"
timeout -k 10 300 $B $M --list-cap 32 > $out/${R}_bench_48x540p_cap32.json 2> $out/${R}_bench_48x540p_cap32.log
timeout -k 10 300 $B $M > $out/${R}_bench_48x540p.json 2> $out/${R}_bench_48x540p.log
rm -rf $out/sq48 $out/sq48c64
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/sq48 -o sq48 -- $B $M --list-cap 32 > $out/sq48.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/sq48c64 -o sq48c64 -- $B $M > $out/sq48c64.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_sq.py $(find $out/sq48 -name "*counter_collection.csv" | head -1) > $out/${R}_pmc_sq_k_sweep_48x540p_cap32.json
python3 tools/pmc_sq.py $(find $out/sq48c64 -name "*counter_collection.csv" | head -1) > $out/${R}_pmc_sq_k_sweep_48x540p_cap64.json
python3 -c "
import json
for f in ('${R}_bench_48x540p_cap32.json','${R}_bench_48x540p.json'):
    d=json.load(open('$out/'+f)); print(f, d['value'], d['roofline']['frac'], d['ms_by_iteration'])
"
