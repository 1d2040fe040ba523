"""Ground truth: the synthetic scenes are analytic (planes + sphere), so every patch can be compared with the surface at
the pixel of its reference view it projects to.  The reference has no such test (and no fixtures at all);
this pins the behaviour of the path -- propagation must cover the scene with patches that are closer to the surface than
the noisy seeds it started from -- independently of the oracle/engine parity."""
import os
import sys

import numpy as np
import pytest

import oracle_binding as ob
from mvskit_amd import synth

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
from quality_probe import patch_errors  # noqa: E402


def test_oracle_reconstructs_the_plane():
    sc = synth.make_scene(nviews=3, W=160, H=120, arc_deg=30.0, radius=4.0, kind="plane")
    seeds = synth.make_seeds(sc, stride=4, seed=5)
    rel0, ang0 = patch_errors(sc, seeds)
    o = ob.Oracle(sc.nviews, level=0, csize=2, wsize=7, minImageNum=2, enable_check=0, seed=2, schedule=ob.SCHEDULE_ENGINE,
                  sum_mode=ob.SUM_TREE64, nthreads=8)
    o.set_scene(sc)
    o.add_patches(seeds)
    for it in range(2):
        o.propagate(it)
        o.update_threshold()
    p = o.patches()
    made = p[p["dscale"] > 0]
    assert made.shape[0] > 10 * seeds.shape[0]
    rel, ang = patch_errors(sc, made)
    # one pixel of the 160-pixel-wide views is 5e-3 of the depth: the seeds sit 0.3 px off the surface, the patches 0.2 px
    assert np.median(rel) < 0.7 * np.median(rel0)
    assert np.median(rel) < 1.5e-3 and np.percentile(rel, 90) < 4e-3
    assert np.median(ang) < 10.0


@pytest.mark.gpu
def test_engine_reconstructs_the_scene(small_multi_scene):
    from mvskit_amd import engine

    sc = small_multi_scene
    seeds = synth.make_seeds(sc, stride=3, seed=19)
    rel0, _ = patch_errors(sc, seeds)
    e = engine.Engine(sc.nviews, level=0, csize=2, wsize=7, minImageNum=3, enable_check=1, seed=3)
    e.set_scene(sc)
    e.upload_patches(seeds)
    for it in range(3):
        e.propagate(it)
        e.filter()
        e.update_threshold()
    p = e.patches()
    made = p[p["dscale"] > 0]
    assert made.shape[0] > 10 * seeds.shape[0]
    rel, ang = patch_errors(sc, made)
    assert np.median(rel) < 0.7 * np.median(rel0)
    assert np.median(rel) < 6e-4 and np.percentile(rel, 90) < 2e-3 and np.percentile(rel, 99) < 8e-3
    assert np.median(ang) < 8.0 and np.median(made["ncc"]) > 0.95


@pytest.mark.gpu
def test_many_view_engines_reconstruct_the_scene():
    """The same ground-truth check for the many-view builds, whose setRefImage goes through the matrix cores: 40 views (the 64-view
    library: no list cut, lists of up to 40 views) and the first 24 of them (the 32-view library), three iterations of PmMvps::run's
    loop with Optim::check and Filter::run.  Independent of the oracle: the patches must lie on the analytic surface, closer than
    the seeds."""
    from mvskit_amd import engine

    full = synth.make_scene(nviews=40, W=256, H=160, arc_deg=120.0, radius=4.0, kind="multi")
    for nv, cap in ((40, 64), (24, 32)):
        sc = synth.Scene(W=full.W, H=full.H, P=full.P[:nv], images=full.images[:nv], centers=full.centers[:nv], points=full.points[:nv],
                         normals=full.normals[:nv], meta=full.meta)
        seeds = synth.make_seeds(sc, stride=4, seed=19)
        rel0, _ = patch_errors(sc, seeds)
        e = engine.Engine(nv, level=0, csize=2, wsize=7, minImageNum=3, enable_check=1, seed=3)
        assert e.list_cap == cap
        e.set_scene(sc)
        e.upload_patches(seeds)
        for it in range(3):
            e.propagate(it)
            e.filter()
            e.update_threshold()
        p = e.patches()
        made = p[p["dscale"] > 0]
        assert made.shape[0] > 3 * seeds.shape[0] and int(made["nimages"].max()) > cap // 2, (made.shape[0], int(made["nimages"].max()))
        rel, ang = patch_errors(sc, made)
        # one pixel of the 256-pixel-wide views is ~3e-3 of the depth
        assert np.median(rel) < 0.8 * np.median(rel0), (nv, np.median(rel), np.median(rel0))
        assert np.median(rel) < 1e-3 and np.percentile(rel, 90) < 3e-3, (nv, np.median(rel), np.percentile(rel, 90))
        assert np.median(ang) < 10.0 and np.median(made["ncc"]) > 0.9, (nv, np.median(ang), np.median(made["ncc"]))
        e.close()


def test_engine_schedule_is_as_good_as_the_reference_order():
    """The red-black / all-views-at-once ENGINE schedule against the FAITHFUL one (the reference's sequential raster sweep,
    live lists, minstd_rand0 draws): different patches, the same quality."""
    sc = synth.make_scene(nviews=3, W=160, H=120, arc_deg=30.0, radius=4.0, kind="plane")
    seeds = synth.make_seeds(sc, stride=4, seed=5)
    stats = {}
    for name, sched, mode in (("faithful", ob.SCHEDULE_FAITHFUL, ob.SUM_SEQ), ("engine", ob.SCHEDULE_ENGINE, ob.SUM_TREE64)):
        o = ob.Oracle(sc.nviews, level=0, csize=2, wsize=7, minImageNum=2, enable_check=0, seed=2, schedule=sched, sum_mode=mode, nthreads=8)
        o.set_scene(sc)
        o.add_patches(seeds)
        for it in range(2):
            o.propagate(it)
            o.update_threshold()
        p = o.patches()
        made = p[p["dscale"] > 0]
        rel, ang = patch_errors(sc, made)
        stats[name] = (made.shape[0], float(np.median(rel)), float(np.median(ang)))
        o.close()
    (nf, rf, af), (ne, re_, ae) = stats["faithful"], stats["engine"]
    assert ne > 0.5 * nf and re_ < 1.25 * rf and ae < 1.25 * af, stats
