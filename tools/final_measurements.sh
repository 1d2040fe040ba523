#!/bin/bash
# tools/final_measurements.sh (on a GPU box, after tools/refresh_profiles.sh): the driver's --steps 20 form of the bench line, the 48-view
# run with Filter::run, two ranks on one GPU over the loopback transport, and the stage shares of the diagnostic builds
# (mvskit_amd/lib/variant_st{,32,64}.so: tools/build_here.sh st -DMVS_STAGE_TIMING [-DMVS_LISTCAP=32 | -DMVS_LISTCAP=64 -DMVS_MAX_IMAGES=64]),
# into gpurun_out/r04final/
set -e
O=gpurun_out/r04final; mkdir -p $O; L=$PWD/mvskit_amd/lib
timeout -k 10 700 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $O/r04_bench_1gpu_steps20.json 2> $O/steps20.log
timeout -k 10 300 python bench.py --filter --views 48 --width 960 --height 540 --steps 3 --warmup 1 --cpu-seconds 0 --no-config5 --no-config4 > $O/r04_bench_filter_48x540p.json 2> $O/f48.log
MVS_CCL_LIBRARY=$PWD/tests/loopback_ccl/_build/libloopback_ccl.so timeout -k 10 400 python bench.py --gpus 2 --steps 3 --warmup 1 --cpu-seconds 0 --no-config5 --no-config4 > $O/r04_bench_2ranks_one_gpu_loopback.json 2> $O/loop.log
(echo "## 16-view build (12 views 1920x1080)"; MVS_ENGINE_LIB=$L/variant_st.so timeout -k 10 300 python bench.py --filter --steps 3 --warmup 1 --cpu-seconds 0 --no-config5 --no-config4 2>&1 | grep -a "cycles\]"
 echo "## 32-view build (48 views 960x540)"; MVS_ENGINE_LIB32=$L/variant_st32.so timeout -k 10 300 python bench.py --views 48 --width 960 --height 540 --list-cap 32 --steps 3 --warmup 1 --cpu-seconds 0 --no-config5 --no-config4 2>&1 | grep -a "cycles\]"
 echo "## 64-view build (48 views 960x540)"; MVS_ENGINE_LIB64=$L/variant_st64.so timeout -k 10 300 python bench.py --views 48 --width 960 --height 540 --steps 3 --warmup 1 --cpu-seconds 0 --no-config5 --no-config4 2>&1 | grep -a "cycles\]") > $O/stage_shares_body.txt
python3 -c "
import json
for f in ('r04_bench_1gpu_steps20.json','r04_bench_filter_48x540p.json','r04_bench_2ranks_one_gpu_loopback.json'):
    d=json.loads(open('$O/'+f).read()); print(f, d['value'], d.get('validation'), d['roofline']['frac'])
"
