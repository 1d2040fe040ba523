"""BASELINE.json configs[1] at full size (12 views 1920x1080, level 0, csize 2, wsize 7, minImageNum 3) -- too large for
the CPU oracle, so the HIP engine is checked through properties that do not depend on the size:
  * the counters of a pass add up (every candidate is pre-filtered or reaches preProcess; every patch that reaches
    preProcess fails, is inserted or replaces),
  * every patch it leaves behind is well formed,
  * the run is reproducible bit for bit,
  * two engines that each sweep one half of the (view, cell) sequence and exchange their records (the multi-GPU protocol,
    here inside one process) end with exactly the single engine's pool,
  * the depth map handed out is the nearest patch of each cell."""
import numpy as np
import pytest

from mvskit_amd import engine, synth

pytestmark = pytest.mark.gpu
CFG = dict(level=0, csize=2, wsize=7, minImageNum=3, enable_check=1, seed=1)
ITERS = 2  # m_depth 1 -> 2: the second iteration runs Optim::check


@pytest.fixture(scope="module")
def full_scene():
    sc = synth.make_scene(nviews=12, W=1920, H=1080, arc_deg=110.0, radius=4.0, kind="multi")
    seeds = synth.make_seeds(sc, level=0, csize=2, stride=4, seed=777)
    return sc, seeds


def _run_single(sc, seeds):
    e = engine.Engine(sc.nviews, **CFG)
    e.set_scene(sc)
    e.upload_patches(seeds)
    counters = []
    for it in range(ITERS):
        counters.append(e.propagate(it))
        e.update_threshold()
    return e, counters


def test_full_size_properties(full_scene):
    sc, seeds = full_scene
    e, counters = _run_single(sc, seeds)
    for c in counters:
        assert c["candidates"] == c["prefiltered"] + c["patches"]
        assert c["patches"] == c["fail0"] + c["fail1"] + c["inserted"] + c["replaced"]
        assert c["patches"] > 500000 and c["view_evals"] > 50 * c["patches"]
    p = e.patches()
    assert p.shape[0] > seeds.shape[0]
    assert p["nimages"].min() >= 1 and p["nimages"].max() <= 16  # seeds may list fewer than minImageNum views (12 views: the 16-view library)
    np.testing.assert_array_equal(p["coord"][:, 3], 1.0)
    made = p[p["dscale"] > 0]                                     # patches the engine created (seeds carry dscale 0)
    assert made.shape[0] > 500000
    assert made["nimages"].min() >= CFG["minImageNum"]
    np.testing.assert_allclose(np.linalg.norm(made["normal"][:, :3].astype(np.float64), axis=1), 1.0, atol=1e-4)
    assert np.all(made["ncc"] <= 1.0 + 1e-6) and np.all(np.isfinite(made["coord"]))
    k = np.arange(16)[None, :] < made["nimages"][:, None]
    imgs = np.where(k, made["images"][:, :16], 255)
    assert np.all(imgs[k] < sc.nviews)
    srt = np.sort(imgs, axis=1)
    assert not np.any((srt[:, 1:] == srt[:, :-1]) & (srt[:, 1:] != 255))  # no view listed twice

    # the depth map is the nearest patch of the cell along the view's optical axis
    depth, normal, ids = e.depth_normal_map(3, 0)
    have = ids >= 0
    assert have.mean() > 0.5
    oaxis = sc.P[3][2].astype(np.float64) / np.linalg.norm(sc.P[3][2, :3].astype(np.float64))
    row = np.full(int(p["id"].max()) + 1, -1, np.int64)  # pool index -> row of the download
    row[p["id"]] = np.arange(p.shape[0])
    assert np.all(row[ids[have]] >= 0)
    d = p["coord"][row[ids[have]]].astype(np.float64) @ oaxis
    np.testing.assert_allclose(depth[have], d, rtol=1e-5)

    # reproducible bit for bit
    e2, counters2 = _run_single(sc, seeds)
    assert counters2 == counters
    p2 = e2.patches()
    assert p.tobytes() == p2.tobytes()
    e2.close()

    # two half-sweeps + exchange == one sweep (device buffers, one process, no collective needed)
    import torch

    dev = torch.device("cuda", 0)
    halves = []
    for r in range(2):
        h = engine.Engine(sc.nviews, shard_index=r, shard_count=2, **CFG)
        h.set_scene(sc)
        h.upload_patches(seeds)
        halves.append(h)
    total = 0
    for it in range(ITERS):
        for pss in range(2):
            recs, kills, counts = [], [], []
            for h in halves:
                total += h.engine_pass(it, pss)["patches"]
                n_new, n_kill, per_view = h.export_counts()
                rec = torch.zeros(max(n_new, 1), 128, dtype=torch.uint8, device=dev)
                kil = torch.full((max(n_kill, 1),), -1, dtype=torch.int32, device=dev)
                torch.cuda.synchronize(dev)
                h.export_device(rec.data_ptr(), rec.shape[0], kil.data_ptr(), kil.shape[0])
                recs.append(rec[:n_new]); kills.append(kil[:n_kill]); counts.append(per_view)
            from mvskit_amd.dist import merge_in_view_order

            parts = merge_in_view_order(recs, counts, sc.nviews, 2)
            allrec = torch.cat(parts).contiguous() if parts else torch.zeros(0, 128, dtype=torch.uint8, device=dev)
            allkill = torch.cat(kills).contiguous()
            torch.cuda.synchronize(dev)
            for h in halves:
                h.commit_device(allrec.data_ptr(), allrec.shape[0], allkill.data_ptr(), allkill.shape[0])
        for h in halves:
            h.update_threshold()
    assert total == sum(c["patches"] for c in counters)
    for h in halves:
        assert h.patches().tobytes() == p.tobytes()
        h.close()
    e.close()


def _cells_in_ref_view(sc, seeds, csize=2):
    """Cell (cx, cy) of every seed in its reference view: PatchManager::setGrids, patch_manager.cpp:241-250."""
    ref = seeds["images"][:, 0].astype(np.int64)
    P = sc.P.astype(np.float64)[ref]
    X = seeds["coord"].astype(np.float64)
    x = np.einsum("nij,nj->ni", P, X)
    px, py = x[:, 0] / x[:, 2], x[:, 1] / x[:, 2]
    return np.floor(px + 0.5).astype(np.int64) // csize, np.floor(py + 0.5).astype(np.int64) // csize


def test_full_size_windowed_parity_vs_oracle(full_scene):
    """BASELINE configs[1] geometry (12 x 1920x1080: 960x540 cells per view, 6.2 M cells, cell_base up to 5.7 M, rows of
    1920 texels) against the CPU oracle -- affordable because the seeds are confined to one 64x64-cell window per view,
    each at a different place of its grid, so only those neighbourhoods have work.  Two iterations of PmMvps::run's loop
    (pmmvps.cpp:90-105; the second with Optim::check): every counter equal, lists equal, coordinates and both depth /
    normal maps within the north-star tolerance.  This is the test that would catch a W = 1920 addressing, cell_base or
    job-index error that the 384x216 parity scenes cannot see."""
    import oracle_binding as ob

    from test_gpu_parity import REL_TOL, _maps_close

    sc, seeds = full_scene
    cx, cy = _cells_in_ref_view(sc, seeds)
    ref = seeds["images"][:, 0].astype(np.int64)
    gw, gh = 960, 540
    keep = np.zeros(seeds.shape[0], bool)
    for v in range(sc.nviews):  # windows spread over the grid: columns 140 .. 760, rows 100 .. 380
        wx = 140 + (v * 389) % 620
        wy = 100 + (v * 97) % 280
        keep |= (ref == v) & (cx >= wx) & (cx < wx + 64) & (cy >= wy) & (cy < wy + 64)
    win = np.ascontiguousarray(seeds[keep])
    assert win.shape[0] > 1500, win.shape
    kw = dict(level=0, csize=2, wsize=7, minImageNum=3, enable_check=1, seed=31)
    o = ob.Oracle(sc.nviews, schedule=ob.SCHEDULE_ENGINE, sum_mode=ob.SUM_TREE64, nthreads=16, **kw)
    e = engine.Engine(sc.nviews, **kw)
    o.set_scene(sc)
    e.set_scene(sc)
    assert e.grid_dims(11) == (gw, gh) == o.grid_dims(11)
    o.add_patches(win)
    e.upload_patches(win)
    total = 0
    for it in range(2):
        co, ce = o.propagate(it), e.propagate(it)
        for k in ("candidates", "prefiltered", "patches", "fail0", "fail1", "inserted", "replaced", "evals", "view_evals", "trimmed"):
            assert co[k] == ce[k], (it, k, co, ce)
        total += ce["patches"]
        o.update_threshold()
        e.update_threshold()
    assert total > 10000, total
    po, pe = o.patches(), e.patches()
    assert po.shape == pe.shape and pe.shape[0] > win.shape[0]
    np.testing.assert_array_equal(po["nimages"], pe["nimages"])
    np.testing.assert_array_equal(po["images"], pe["images"])
    np.testing.assert_array_equal(po["vimages"], pe["vimages"])
    np.testing.assert_allclose(pe["coord"], po["coord"], rtol=REL_TOL, atol=1e-6)
    np.testing.assert_allclose(pe["normal"], po["normal"], rtol=0, atol=REL_TOL)
    np.testing.assert_allclose(pe["ncc"], po["ncc"], rtol=REL_TOL, atol=1e-5)
    assert (pe["coord"] == po["coord"]).all(axis=1).mean() > 0.99
    tot, bad = _maps_close(o, e, sc.nviews)
    assert tot > 5000 and bad == 0, (tot, bad)
    o.close()
    e.close()


def test_full_size_filter_run_properties(full_scene):
    """BASELINE configs[4] at its size: Filter::run (filter.cpp:25-49) on the pool two full-size iterations leave behind
    (too large for the oracle; the stage-by-stage parity is tests/test_gpu_parity.py::test_filter_run_matches_oracle)."""
    sc, seeds = full_scene
    e, _ = _run_single(sc, seeds)
    before = e.patches()
    removed = e.filter()
    after = e.patches()
    nrem = sum(removed.values())
    assert all(v >= 0 for v in removed.values()) and nrem > 0
    assert before.shape[0] - nrem == after.shape[0] == e.num_patches()
    assert after.shape[0] > 0.5 * before.shape[0]
    # every survivor is a patch that existed before (same position and normal), in the same relative order
    key = lambda p: np.ascontiguousarray(np.concatenate([p["coord"], p["normal"]], axis=1)).view(np.dtype((np.void, 32))).ravel()
    kb, ka = key(before), key(after)
    pos = {k.tobytes(): i for i, k in enumerate(kb)}
    idx = np.array([pos.get(k.tobytes(), -1) for k in ka])
    assert (idx >= 0).all() and np.all(np.diff(idx) > 0)
    b = before[idx]
    # filterExact only removes views from m_images (filter.cpp:165-203) and every survivor keeps minImageNum of them
    sel = np.arange(16)[None, :]
    ia = np.where(sel < after["nimages"][:, None], after["images"][:, :16], 255)
    ib = np.where(sel < b["nimages"][:, None], b["images"][:, :16], 254)
    assert np.all((ia[:, :, None] == ib[:, None, :]).any(axis=2) | (ia == 255))
    assert after["nimages"].min() >= CFG["minImageNum"]
    assert np.all(after["nvimages"] <= 16) and np.isfinite(after["coord"]).all()
    # bit-reproducible
    e2, _ = _run_single(sc, seeds)
    removed2 = e2.filter()
    assert removed2 == removed and e2.patches().tobytes() == after.tobytes()
    # and the propagation goes on from the filtered pool (PmMvps::run's next iteration)
    e.update_threshold()
    c = e.propagate(2)
    assert c["patches"] > 100000 and c["candidates"] == c["prefiltered"] + c["patches"]
    e.close()
    e2.close()


@pytest.fixture(scope="module")
def scene_48x4k():
    """BASELINE configs[3] geometry: 48 views of 3840x2160 on the 110 degree arc (the ground truth behind the seeds kept for all of them)."""
    return synth.make_scene(nviews=48, W=3840, H=2160, arc_deg=110.0, radius=4.0, kind="multi")


def test_config4_48_views_4k_five_iterations(scene_48x4k):
    """BASELINE configs[3] on ONE MI355X, for real: 48 views of 3840x2160 (99.5 M cells), the 64-view library (no list is cut: the
    reference's m_images is unbounded, optim.cpp:165-205), seeds in ALL views, and the five iterations of PmMvps::run's loop
    (pmmvps.cpp:90-110) -- Propagate::run, Filter::run, updateThreshold, ++m_depth -- so Optim::check runs from the second
    iteration on and Filter::run five times at that size.  Too large for the oracle: the engine is held to the properties that do
    not depend on the size, and what it took (HBM, time per stage) is recorded in gpurun_out/config4_run.json.
    Seeds: one per 4x4 cells over ALL cells of ALL 48 grids (6.2 M seeds).  Every patch sits in the cell lists of 20-40 views; the
    index holds a 4-byte id per membership (an 8-byte key more for m_pgrids) behind 64-bit offsets, so the billions of memberships of
    this pool fit the card (rounds 1-3 seeded the central quarter only: a 60-byte entry behind 32-bit offsets capped the pool at
    ~55 M patches)."""
    import json
    import os
    import time

    import torch

    sc = scene_48x4k
    n = sc.nviews
    seeds = synth.make_seeds(sc, level=0, csize=2, stride=4, seed=777)
    cx, cy = _cells_in_ref_view(sc, seeds)
    assert seeds.shape[0] > 5_000_000 and len(set(seeds["images"][:, 0].tolist())) == n
    assert cx.min() < 64 and cx.max() > 1855 and cy.min() < 64 and cy.max() > 1015  # no mask: the seeds reach the borders of the grids
    MAX_PATCHES = 128_000_000
    free0, total_mem = torch.cuda.mem_get_info(0)
    e = engine.Engine(n, max_patches=MAX_PATCHES, **CFG)
    assert e.list_cap == 64 and e.dtype.itemsize == 192  # more than 32 views: libmvskit_engine_cap64.so
    e.set_scene(sc)
    assert e.grid_dims(47) == (1920, 1080)
    e.upload_patches(seeds)
    log = []
    peak = 0.0
    for it in range(5):
        t0 = time.perf_counter()
        c = e.propagate(it)
        t = e.timing()
        t1 = time.perf_counter()
        f = e.filter()
        fs = e.filter_stats()
        t2 = time.perf_counter()
        e.update_threshold()
        free1, _ = torch.cuda.mem_get_info(0)
        peak = max(peak, (free0 - free1) / 2 ** 30)
        assert c["candidates"] == c["prefiltered"] + c["patches"], (it, c)
        assert c["patches"] == c["fail0"] + c["fail1"] + c["inserted"] + c["replaced"], (it, c)
        assert c["patches"] > 4_000_000 and c["inserted"] > 2_000_000, (it, c)
        assert all(v >= 0 for v in f.values())
        log.append({"iteration": it, "counters": c, "propagate_s": t1 - t0, "timing_ms": t, "filter_removed": f, "filter_s": t2 - t1,
                    "filter_total_ms": fs["total_ms"], "filter_stage_ms": {k: fs[k] for k in ("outside_ms", "exact_ms", "neighbor_ms", "groups_ms", "rebuild_ms")},
                    "pool_alive": e.num_patches(), "hbm_used_GiB": (free0 - free1) / 2 ** 30})
    assert log[1]["counters"]["fail1"] > 0 or log[2]["counters"]["fail1"] > 0  # Optim::check (m_depth >= 2) rejects something at this size too
    assert sum(sum(l["filter_removed"].values()) for l in log) > 0
    p = e.patches()
    assert p.shape[0] > 20_000_000  # the central-quarter runs of round 3 ended with 9.2 M
    made = p[p["dscale"] > 0][::5]  # the properties below on every fifth patch (host time)
    assert made.shape[0] > 3_000_000
    assert made["nimages"].min() >= CFG["minImageNum"] and made["nimages"].max() > 32  # lists longer than the 32-view build could keep
    k = np.arange(64)[None, :] < made["nimages"][:, None]
    imgs = np.where(k, made["images"], 255)
    assert np.all(imgs[k] < n)
    srt = np.sort(imgs, axis=1)
    assert not np.any((srt[:, 1:] == srt[:, :-1]) & (srt[:, 1:] != 255))  # no view listed twice
    assert np.isfinite(made["coord"]).all() and np.all(made["ncc"] <= 1.0 + 1e-6)
    np.testing.assert_allclose(np.linalg.norm(made["normal"][:, :3].astype(np.float64), axis=1), 1.0, atol=1e-4)
    # the patches stay on the scene: every object of the synthetic scene lies within 3 units of the origin
    assert np.percentile(np.linalg.norm(made["coord"][:, :3], axis=1), 99.9) < 4.0
    # depth map of the last view: the nearest patch of each cell
    depth, normal, ids = e.depth_normal_map(47, 0)
    have = ids >= 0
    assert have.sum() > 100000
    oaxis = sc.P[47][2].astype(np.float64) / np.linalg.norm(sc.P[47][2, :3].astype(np.float64))
    row = np.full(int(p["id"].max()) + 1, -1, np.int64)
    row[p["id"]] = np.arange(p.shape[0])
    d = p["coord"][row[ids[have]]].astype(np.float64) @ oaxis
    np.testing.assert_allclose(depth[have], d, rtol=1e-5)
    assert peak < 288.0
    rec = {"views": n, "width": sc.W, "height": sc.H, "list_cap": e.list_cap, "record_bytes": int(e.dtype.itemsize), "cells": 48 * 1920 * 1080, "seeds": int(seeds.shape[0]),
           "max_patches": MAX_PATCHES, "seeding": "1 seed per 4x4 cells over all cells of all 48 grids", "iterations": log, "hbm_peak_GiB": peak, "hbm_total_GiB": total_mem / 2 ** 30, "pool_alive": int(p.shape[0]),
           "mean_nimages": float(made["nimages"].mean()), "max_nimages": int(made["nimages"].max())}
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "config4_run.json"), "w") as fh:
        json.dump(rec, fh, indent=1)
    e.close()


def test_config4_windowed_parity_vs_oracle(scene_48x4k):
    """BASELINE configs[3] addressing against the CPU oracle: 48 x 3840x2160 (1920x1080 cells per view, cell_base up to 97.5 M, rows
    of 3840 texels, 64-view lists) -- affordable because the seeds are confined to one 64x64-cell window per seeded view, placed near
    the far corners of the grids of the LAST views.  Two iterations of PmMvps::run's loop (the second with Optim::check): every
    counter equal, lists equal, coordinates and both depth / normal maps within the north-star tolerance."""
    import oracle_binding as ob

    from test_gpu_parity import REL_TOL

    sc = scene_48x4k
    views = [47, 46, 44, 40, 33, 20]
    seeds = synth.make_seeds(sc, level=0, csize=2, stride=3, seed=31, views=views)
    cx, cy = _cells_in_ref_view(sc, seeds)
    ref = seeds["images"][:, 0].astype(np.int64)
    gw, gh = 1920, 1080
    corners = {47: (gw - 70 - 64, gh - 70 - 64), 46: (70, gh - 70 - 64), 44: (gw - 70 - 64, 70), 40: (70, 70), 33: (gw - 150 - 64, gh // 2), 20: (gw // 2, gh - 150 - 64)}
    keep = np.zeros(seeds.shape[0], bool)
    for v, (wx, wy) in corners.items():
        keep |= (ref == v) & (cx >= wx) & (cx < wx + 64) & (cy >= wy) & (cy < wy + 64)
    win = np.ascontiguousarray(seeds[keep])
    assert win.shape[0] > 1000, win.shape
    kw = dict(level=0, csize=2, wsize=7, minImageNum=3, enable_check=1, seed=31)
    o = ob.Oracle(sc.nviews, wide=True, list_cap=64, schedule=ob.SCHEDULE_ENGINE, sum_mode=ob.SUM_TREE64, nthreads=16, **kw)
    e = engine.Engine(sc.nviews, max_patches=4_000_000, **kw)
    assert e.list_cap == 64
    o.set_scene(sc)
    e.set_scene(sc)
    assert e.grid_dims(47) == (gw, gh) == o.grid_dims(47)
    o.add_patches(win)
    e.upload_patches(win)
    total = 0
    for it in range(2):
        co, ce = o.propagate(it), e.propagate(it)
        for k in ("candidates", "prefiltered", "patches", "fail0", "fail1", "inserted", "replaced", "evals", "view_evals", "trimmed"):
            assert co[k] == ce[k], (it, k, co, ce)
        total += ce["patches"]
        o.update_threshold()
        e.update_threshold()
    assert total > 5000, total
    po, pe = o.patches(), e.patches()
    assert po.shape == pe.shape and pe.shape[0] > win.shape[0]
    assert int(pe["nimages"].max()) > 16
    np.testing.assert_array_equal(po["nimages"], pe["nimages"])
    np.testing.assert_array_equal(po["images"], pe["images"])
    np.testing.assert_array_equal(po["vimages"], pe["vimages"])
    np.testing.assert_allclose(pe["coord"], po["coord"], rtol=REL_TOL, atol=1e-6)
    np.testing.assert_allclose(pe["normal"], po["normal"], rtol=0, atol=REL_TOL)
    np.testing.assert_allclose(pe["ncc"], po["ncc"], rtol=REL_TOL, atol=1e-5)
    assert (pe["coord"] == po["coord"]).all(axis=1).mean() > 0.99
    tot = bad = 0
    for v in (47, 44, 33, 12):  # maps of seeded views and of a view that only receives patches
        for kind in (0, 1):
            do, no, _ = o.depth_normal_map(v, kind)
            de, ne, _ = e.depth_normal_map(v, kind)
            assert np.array_equal(np.isnan(do), np.isnan(de)), (v, kind)
            m = ~np.isnan(do)
            rel = np.abs(de[m] - do[m]) / np.abs(do[m])
            ang = np.arccos(np.clip((ne[m] * no[m]).sum(-1), -1, 1))
            tot += int(m.sum())
            bad += int(((rel > REL_TOL) | (ang > REL_TOL)).sum())
    assert tot > 2000 and bad == 0, (tot, bad)
    o.close()
    e.close()
