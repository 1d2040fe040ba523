#!/bin/bash
# tools/ab.sh <libA> <libB> ...: bench.py with each engine build in turn (twice, alternating) on the same box
for round in 1 2; do
  for lib in "$@"; do
    v=$(MVS_ENGINE_LIB=$lib timeout -k 10 300 python bench.py --steps 3 --warmup 1 --cpu-seconds 0 --no-config5 --no-config4 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('%.0f patches/s  %.1f ms/step  sweep %.1f ms' % (d['value'], d['ms_per_step'], d['roofline']['sweep_ms']/d['steps']))")
    echo "$(basename $lib): $v"
  done
done
