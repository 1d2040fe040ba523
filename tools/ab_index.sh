#!/bin/bash
# tools/ab_index.sh <lib>...: the headline bench three times per build; per timed iteration its wall time and the engine's own split
for round in 1 2 3; do for lib in "$@"; do
MVS_ENGINE_LIB=$lib timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-config5 --no-config4 2>gpurun_out/x.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $lib)', round(d['value']/1e6,2), [round(x,1) for x in d['ms_by_iteration']], round(d['roofline']['index_ms'],1))"
grep "iter " gpurun_out/x.err | tail -3 | sed -e "s/.*): //" -e "s/, patches.*timing//" -e "s/'sweep_launches.*//"
done; done
