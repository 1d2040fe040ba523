"""Regenerates the oracle-derived fixtures in this directory (run from the repo root:
`python tests/golden/make_golden.py`).  rng_minstd_rand0.json is NOT produced here: it holds the
libstdc++ draws recorded in SURVEY.md section 0.4 and pins the oracle, not the other way round."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle_binding as ob  # noqa: E402


def rng_counter():
    L = ob.lib()
    keys = [[1, 0, 0, 0, 0, 0], [1, 0, 0, 0, 0, 1], [7, 3, 11, 259199, 17, 40], [0xFFFFFFFF, 2, 5, 100, 31, 16],
            [12345, 1, 2, 3, 4, 5]]
    vals = [float(np.float32(L.orc_rng_uniform(*k))) for k in keys]
    json.dump({"source": "oracle engine counter RNG (no reference counterpart)", "keys": keys, "values": vals},
              open(os.path.join(HERE, "rng_counter.json"), "w"), indent=1)


if __name__ == "__main__":
    rng_counter()
    print("golden fixtures written to", HERE)


def tiny_scene_fixture():
    """Self-contained parity fixture: a 3-view 112x80 scene (inputs stored byte for byte), seed patches, and what the
    oracle (engine schedule, TREE64 sums) makes of them: per-seed NCC, preProcess flags/records, refinePatch results, and
    the counters + patch pool after two Propagate::run iterations (m_depth 1 -> 2, Optim::check active in the second)."""
    from mvskit_amd import synth

    sc = synth.make_scene(nviews=3, W=112, H=80, arc_deg=30.0, radius=4.0, kind="plane")
    seeds = synth.make_seeds(sc, stride=3, seed=41)
    kw = dict(level=0, csize=2, wsize=7, minImageNum=2, enable_check=1, seed=77, schedule=ob.SCHEDULE_ENGINE, sum_mode=ob.SUM_TREE64)
    o = ob.Oracle(3, **kw)
    o.set_scene(sc)
    ncc = np.array([o.compute_ncc(s) for s in seeds], dtype=np.float32)
    pre_flag, pre_rec = [], []
    for s in seeds:
        f, r = o.preprocess(s)
        pre_flag.append(f)
        pre_rec.append(r)
    pre_flag = np.array(pre_flag, dtype=np.int32)
    pre_rec = np.array(pre_rec, dtype=ob.PATCH_DTYPE)
    keep = np.nonzero(pre_flag == 0)[0]
    ref_rec = np.array([o.refine(pre_rec[i], (0, 0, j, 0))[1] for j, i in enumerate(keep)], dtype=ob.PATCH_DTYPE)
    o.add_patches(seeds)
    counters = []
    for it in range(2):
        counters.append(o.propagate(it))
        o.update_threshold()
    pool = o.patches()
    np.savez_compressed(os.path.join(HERE, "tiny_scene.npz"), W=sc.W, H=sc.H, P=sc.P, images=sc.images, seeds=seeds.view(np.uint8),
                        ncc=ncc, pre_flag=pre_flag, pre_rec=pre_rec.view(np.uint8), keep=keep, ref_rec=ref_rec.view(np.uint8),
                        pool=pool.view(np.uint8), counters=json.dumps(counters), config=json.dumps({k: v for k, v in kw.items()}))
    print("tiny_scene.npz:", seeds.shape[0], "seeds,", pool.shape[0], "patches after 2 iterations,", counters[-1]["patches"], "patches in iteration 1")


if __name__ == "__main__":
    tiny_scene_fixture()
