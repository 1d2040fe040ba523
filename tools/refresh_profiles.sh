#!/bin/bash
# tools/refresh_profiles.sh: on a GPU box, regenerates what profiles/ holds for the current build
# (run through gpurun; copies land in gpurun_out/profiles_new/, to be moved into profiles/ after review)
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/profiles_new
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1"
echo "== bench (with CPU baseline)"; timeout -k 10 500 $B > $out/bench_1gpu.json 2> $out/bench_1gpu.log
echo "== kernel trace"; timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o kt -- $B --cpu-seconds 0 > $out/kt.log 2>&1
echo "== pmc fetch"; timeout -k 10 500 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o fetch -- $B --cpu-seconds 0 > $out/fetch.log 2>&1
echo "== pmc write"; timeout -k 10 500 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o write -- $B --cpu-seconds 0 > $out/write.log 2>&1
echo "== bench --filter"; timeout -k 10 500 $B --filter --cpu-seconds 0 > $out/bench_filter.json 2> $out/bench_filter.log
echo "== kernel trace --filter"; timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/ktf -o ktf -- $B --filter --cpu-seconds 0 > $out/ktf.log 2>&1
find $out -name "*.csv" | head -20
