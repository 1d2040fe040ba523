"""Run by tests/test_oracle_kat.py::test_list_cap_48_views in a process of its own, on the wide build of the oracle (64 views per
list, 192-byte records): the same 48-view scene once with the engine's 16-view list cap and once with
lists as long as the reference makes them (optim.cpp:165-205 pushes every qualifying view), two iterations of
PmMvps::run's loop with Optim::check.  Prints one JSON line: how often a list wanted to be longer than 16, the patch
counts, and how far the depth / normal maps of the two runs are apart."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
import oracle_binding as ob  # noqa: E402
from mvskit_amd import synth  # noqa: E402


def _run(sc, seeds, cap):
    nviews = sc.nviews
    o = ob.Oracle(nviews, wide=True, level=0, csize=2, wsize=7, minImageNum=3, schedule=ob.SCHEDULE_ENGINE, sum_mode=ob.SUM_TREE64, enable_check=1,
                  seed=6, nthreads=8, list_cap=cap)
    assert o.L.orc_list_storage() >= cap, "needs the wide oracle build (make -C oracle wide)"
    o.set_scene(sc)
    o.add_patches(seeds)
    patches = 0
    for it in range(2):
        patches += o.propagate(it)["patches"]
        o.update_threshold()
    maps = [o.depth_normal_map(v, kind) for v in range(0, nviews, 4) for kind in (0, 1)]
    p = o.patches()
    r = dict(patches=patches, alive=int(p.shape[0]), trunc=int(o.L.orc_list_truncations(o.h)), maps=maps,
             mean_nimages=float(p["nimages"].mean()), max_nimages=int(p["nimages"].max()))
    o.close()
    return r


def _compare(a, b):
    tot = both = close = 0
    rel_all = []
    for (da, na, _), (db, nb, _) in zip(a["maps"], b["maps"]):
        ma, mb = ~np.isnan(da), ~np.isnan(db)
        tot += int((ma | mb).sum())
        m = ma & mb
        both += int(m.sum())
        rel = np.abs(da[m] - db[m]) / np.abs(db[m])
        ang = np.arccos(np.clip((na[m] * nb[m]).sum(-1), -1, 1))
        close += int(((rel <= 1e-3) & (ang <= 1e-3)).sum())
        rel_all.append(rel)
    rel_all = np.concatenate(rel_all) if rel_all else np.zeros(0)
    return {"cells_either": tot, "cells_both": both, "cells_within_1e-3": close,
            "median_rel_depth_diff": float(np.median(rel_all)) if rel_all.size else 0.0}


def main():
    nviews = 48
    sc = synth.make_scene(nviews=nviews, W=192, H=144, arc_deg=141.0, radius=4.0, kind="multi")
    seeds = synth.make_seeds(sc, stride=4, seed=31, views=range(0, nviews, 3))
    runs = {cap: _run(sc, seeds, cap) for cap in (16, 32, 64)}
    out = {}
    for cap in (16, 32, 64):
        r = runs[cap]
        out[f"cap{cap}"] = {"truncations": r["trunc"], "patches": r["patches"], "alive": r["alive"], "mean_nimages": r["mean_nimages"],
                            "max_nimages": r["max_nimages"]}
    out["cap16_vs_untruncated"] = _compare(runs[16], runs[64])
    out["cap32_vs_untruncated"] = _compare(runs[32], runs[64])
    print(json.dumps(out))


if __name__ == "__main__":
    main()
