#!/bin/bash
# tools/ab_filter.sh <libA> <libB> ...: `bench.py --filter` with each engine build in turn on the same box; prints Filter::run's
# stage times per call (mvs_engine_filter_stats) -- the A/B harness of the filter kernels, as tools/ab.sh is of k_sweep
for lib in "$@"; do
  MVS_ENGINE_LIB=$lib timeout -k 10 300 python bench.py --filter --steps 3 --warmup 1 --cpu-seconds 0 --no-config5 --no-config4 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); f = d['roofline_filter']; s = f['stage_ms']; n = 3.0
print('$(basename $lib): %.2f M patches/s | per call: outside %.1f exact %.1f neighbor %.1f groups %.1f rebuild %.1f total %.1f ms' % (d['value'] / 1e6, s['outside_ms'] / n, s['exact_ms'] / n, s['neighbor_ms'] / n, s['groups_ms'] / n, s['rebuild_ms'] / n, s['total_ms'] / n))"
done
