#!/bin/bash
# tools/refresh_profiles.sh [round]: on a GPU box, regenerates what profiles/ holds for the current build
# (run through gpurun; copies land in gpurun_out/profiles_new/, to be moved into profiles/ after review)
set -e
R=${1:-r04}
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_new
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B0="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1"
B="$B0 --no-config5 --no-config4"
echo "== bench (with CPU baseline and the config5 block)"; timeout -k 10 500 $B0 > $out/${R}_bench_1gpu.json 2> $out/${R}_bench_1gpu.log
echo "== kernel trace"; timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- $B --cpu-seconds 0 > $out/kt.log 2>&1
echo "== pmc fetch"; timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o fetch -- $B --cpu-seconds 0 > $out/fetch.log 2>&1
echo "== pmc write"; timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o write -- $B --cpu-seconds 0 > $out/write.log 2>&1
echo "== pmc sq"; timeout -k 10 500 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/sq -o sq -- $B --cpu-seconds 0 > $out/sq.log 2>&1
echo "== bench --filter"; timeout -k 10 500 $B --filter --cpu-seconds 0 > $out/${R}_bench_filter.json 2> $out/${R}_bench_filter.log
echo "== kernel trace --filter"; timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ktf -o ktf -- $B --filter --cpu-seconds 0 > $out/ktf.log 2>&1
echo "== pmc fetch --filter"; timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetchf -o fetchf -- $B --filter --cpu-seconds 0 > $out/fetchf.log 2>&1
echo "== pmc write --filter"; timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/writef -o writef -- $B --filter --cpu-seconds 0 > $out/writef.log 2>&1
echo "== instruction mix (four passes)"
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 --output-format csv -d $out/mixa -o a -- $B --cpu-seconds 0 > $out/mixa.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $out/mixb -o b -- $B --cpu-seconds 0 > $out/mixb.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_SALU --output-format csv -d $out/mixc -o c -- $B --cpu-seconds 0 > $out/mixc.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_LDS_BANK_CONFLICT --output-format csv -d $out/mixd -o d -- $B --cpu-seconds 0 > $out/mixd.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_mix.py $(find $out/mixa $out/mixb $out/mixc $out/mixd -name "*counter_collection.csv") > $out/${R}_pmc_mix_k_sweep.json
python3 tools/pmc_sq.py $(find $out/sq -name "*counter_collection.csv" | head -1) > $out/${R}_pmc_sq_k_sweep.json
# (the many-view part -- 48 x 960x540 on the 32- and the 64-view build -- is tools/refresh_manyview.sh: its own gpurun call)
find $out -name "*.csv" | head -40
