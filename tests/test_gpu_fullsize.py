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
    assert p["nimages"].min() >= 1 and p["nimages"].max() <= 16  # seeds may list fewer than minImageNum views
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
