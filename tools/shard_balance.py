#!/usr/bin/env python3
"""tools/shard_balance.py [--shards 8]: how evenly the contiguous job ranges of a view-sharded run divide the sweep.

No second GPU is needed: `--shards` engines with shard_index = r, shard_count = N run the SAME colour pass from the SAME pool one
after another on one card; each reports the HIP-event time of its share of the sweep.  Two passes are measured: iteration 0,
colour 1 from the seeds (m_depth 1, no Optim::check) and iteration 1, colour 0 from the pool a full run holds after iteration 0
(m_depth 2, check on).  Done for the three cut rules of mvs_engine_pass (MVS_SPLIT_PROXY): 0 = equal job counts (round 2),
1 = equal source entries, 2 = equal expected trials by kind (the default).  Prints one JSON object."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--shards", type=int, default=8)
ap.add_argument("--views", type=int, default=12)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--modes", default="0,1,2")
args = ap.parse_args()

import bench  # noqa: E402  (scene cache + constants)
from mvskit_amd import engine  # noqa: E402

bargs = argparse.Namespace(views=args.views, width=args.width, height=args.height, seed_stride=2, scene_cache=os.path.join("/tmp", "mvskit_scene_cache"))
sc, seeds = bench.load_scene(bargs, 0)
KW = dict(level=0, csize=2, wsize=7, minImageNum=3, enable_check=1, seed=1)

e = engine.Engine(args.views, **KW)
e.set_scene(sc)
e.upload_patches(seeds)
e.set_thresholds(0.7, 0.4, 1)
c0 = e.propagate(0)
e.update_threshold()
pool1 = e.patches()
e.close()
cases = [("iteration 0, colour 1, from the seeds (no check)", seeds, 0, 1, (0.7, 0.4, 1)),
         ("iteration 1, colour 0, from the pool after iteration 0 (check on)", pool1, 1, 0, (0.65, 0.35, 2))]
out = {"shards": args.shards, "views": args.views, "width": args.width, "height": args.height, "cases": []}
for mode in [int(m) for m in args.modes.split(",")]:
    os.environ["MVS_SPLIT_PROXY"] = str(mode)
    per_case = [[] for _ in cases]
    patches = [[] for _ in cases]
    for r in range(args.shards):
        e = engine.Engine(args.views, shard_index=r, shard_count=args.shards, **KW)
        e.set_scene(sc)
        for k, (_, pool, it, colour, thr) in enumerate(cases):
            e.clear_patches()
            e.upload_patches(pool)
            e.set_thresholds(*thr)
            c = e.engine_pass(it, colour)
            per_case[k].append(e.timing()["sweep_ms"])
            patches[k].append(c["patches"])
        e.close()
    for k, (name, *_rest) in enumerate(cases):
        ms = per_case[k]
        out["cases"].append({"split": {0: "equal job counts", 1: "equal source entries", 2: "equal expected trials by kind"}[mode], "MVS_SPLIT_PROXY": mode,
                             "pass": name, "sweep_ms_per_shard": ms, "patches_per_shard": patches[k], "max_over_mean": max(ms) / (sum(ms) / len(ms))})
        print(f"[shard_balance] mode {mode} {name}: max/mean {out['cases'][-1]['max_over_mean']:.3f}  ms {['%.1f' % m for m in ms]}", file=sys.stderr, flush=True)
print(json.dumps(out))
