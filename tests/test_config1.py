"""BASELINE.json configs[0] at its stated size: 2 views 640x480 of a textured plane, cameras 20 degrees
apart (beyond Option::m_maxAngleThreshold's 10 degrees), f = 765.7 px (cf. test/test.cpp:96-100), level 0, csize 2,
wsize 7, minImageNum 2, threshold 0.7, seeds 1 per 8x8 cells, ONE PatchMatch iteration.

CPU part ("CPU reference path, plumbing, no GPU"): the oracle in its FAITHFUL schedule (the reference's sequential sweep) and in
the ENGINE schedule both grow the seed set and stay on the true plane.  GPU part: the HIP engine reproduces the ENGINE schedule."""
import numpy as np
import pytest

import oracle_binding as ob
from mvskit_amd import synth

CFG1 = dict(level=0, csize=2, wsize=7, minImageNum=2, nccThreshold=0.7, enable_check=0, seed=11)


@pytest.fixture(scope="module")
def cfg1():
    sc = synth.make_scene(nviews=2, W=640, H=480, arc_deg=20.0, radius=4.0, kind="plane")
    seeds = synth.make_seeds(sc, level=0, csize=2, stride=8, seed=777)
    return sc, seeds


def _plane_error(sc, p):
    """distance from the plane z = 0 relative to the depth in the reference view"""
    depth = np.linalg.norm(p["coord"][:, :3].astype(np.float64) - sc.centers[p["images"][:, 0]].astype(np.float64)[:, :3], axis=1)
    return np.abs(p["coord"][:, 2].astype(np.float64)) / depth


def test_config1_cpu_reference_path(cfg1):
    sc, seeds = cfg1
    assert sc.P.shape == (2, 3, 4) and seeds.shape[0] > 2000
    runs = {}
    for name, sched, summ in (("faithful", ob.SCHEDULE_FAITHFUL, ob.SUM_SEQ), ("engine", ob.SCHEDULE_ENGINE, ob.SUM_TREE64)):
        o = ob.Oracle(2, schedule=sched, sum_mode=summ, nthreads=4, **CFG1)
        o.set_scene(sc)
        assert o.grid_dims(0) == (320, 240)
        o.add_patches(seeds)
        if sched == ob.SCHEDULE_FAITHFUL:
            o.set_cell_budget(12000)  # the raster sweep carries patches across the whole grid in ONE iteration (880 k candidates,
                                      # 40 s on one core); the first 12 000 source cells (37 rows of view 0) are plumbing enough here
        c = o.propagate(0)
        p = o.patches()
        o.close()
        assert c["candidates"] == c["prefiltered"] + c["patches"]
        assert c["patches"] == c["fail0"] + c["fail1"] + c["inserted"] + c["replaced"]
        assert c["inserted"] > 1000, c                               # the seed set grows: minImageNum 2 lets two-view patches through (D14)
        made = p[p["dscale"] > 0]
        assert made.shape[0] > 1000 and np.all(made["nimages"] == 2)
        err = _plane_error(sc, made)
        assert np.median(err) < 1e-3 and np.percentile(err, 90) < 5e-3, (name, np.median(err), np.percentile(err, 90))
        assert np.median(made["ncc"]) > 0.9
        runs[name] = (c, made)
    # the two schedules differ by design: the sequential raster sweep hands a patch on from cell to cell
    # within one iteration, the red-black passes move information one cell per pass
    assert runs["faithful"][0]["patches"] > runs["engine"][0]["patches"] > 5000


@pytest.mark.gpu
def test_config1_gpu_matches_oracle(cfg1):
    from mvskit_amd import engine

    sc, seeds = cfg1
    o = ob.Oracle(2, schedule=ob.SCHEDULE_ENGINE, sum_mode=ob.SUM_TREE64, nthreads=8, **CFG1)
    e = engine.Engine(2, **CFG1)
    o.set_scene(sc)
    e.set_scene(sc)
    assert e.grid_dims(1) == (320, 240)
    o.add_patches(seeds)
    e.upload_patches(seeds)
    co, ce = o.propagate(0), e.propagate(0)
    assert co == ce, (co, ce)
    po, pe = o.patches(), e.patches()
    assert po.shape == pe.shape and pe.shape[0] > seeds.shape[0] + 1000
    np.testing.assert_array_equal(po["nimages"], pe["nimages"])
    np.testing.assert_array_equal(po["images"], pe["images"])
    np.testing.assert_allclose(pe["coord"], po["coord"], rtol=1e-3, atol=1e-6)
    np.testing.assert_allclose(pe["normal"], po["normal"], rtol=0, atol=1e-3)
    assert (pe["coord"] == po["coord"]).all(axis=1).mean() > 0.99
    for v in range(2):
        for kind in (0, 1):
            do, no, io = o.depth_normal_map(v, kind)
            de, ne, ie = e.depth_normal_map(v, kind)
            np.testing.assert_array_equal(np.isnan(do), np.isnan(de))
            m = ~np.isnan(do)
            assert m.sum() > 1000
            np.testing.assert_allclose(de[m], do[m], rtol=1e-3)
            assert np.all(np.arccos(np.clip((ne[m] * no[m]).sum(-1), -1, 1)) <= 1e-3)
    o.close()
    e.close()
