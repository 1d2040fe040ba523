"""World-size-2 test (gloo, CPU) of the view-sharded Propagate::run in mvskit_amd/dist.py: two ranks, each sweeping
half of the views and exchanging new records + kill ids after every colour pass, must end with exactly the pool a
single rank produces.  The CPU oracle stands in for the HIP engine (same pass/export/commit split)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene():
    from mvskit_amd import synth

    sc = synth.make_scene(nviews=4, W=192, H=128, arc_deg=45.0, radius=4.0, kind="plane")
    seeds = synth.make_seeds(sc, stride=4, seed=9)
    return sc, seeds


def _worker(rank, world, port, out_dir, ranged=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist

    import oracle_binding as ob
    from mvskit_amd import dist as mdist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc, seeds = _scene()
    shard = dict(shard_index=rank, shard_count=world) if ranged else dict(view_begin=rank, view_stride=world)
    o = ob.Oracle(sc.nviews, level=0, minImageNum=2, enable_check=0, seed=5, schedule=ob.SCHEDULE_ENGINE, sum_mode=ob.SUM_TREE64,
                  nthreads=2, **shard)
    o.set_scene(sc)
    o.add_patches(seeds)
    ex = mdist.HostExchange()
    tot = None
    for it in range(2):
        c = mdist.sharded_propagate_host(o, it, ex, sc.nviews, ob.PATCH_DTYPE)
        tot = c if tot is None else {k: tot[k] + c[k] for k in c}
    np.save(os.path.join(out_dir, f"pool_{rank}.npy"), o.patches().view(np.uint8))
    np.save(os.path.join(out_dir, f"patches_{rank}.npy"), np.array([tot["patches"]]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,ranged", [(2, False), (3, True)])
def test_ranks_equal_one_rank(tmp_path, world, ranged):
    """world 2 sharded by whole views; world 3 by contiguous ranges of the (view, cell) sequence (4 views do not divide
    by 3: a view is split between two ranks)."""
    import oracle_binding as ob

    sc, seeds = _scene()
    o = ob.Oracle(sc.nviews, level=0, minImageNum=2, enable_check=0, seed=5, schedule=ob.SCHEDULE_ENGINE, sum_mode=ob.SUM_TREE64, nthreads=4)
    o.set_scene(sc)
    o.add_patches(seeds)
    patches = 0
    for it in range(2):
        patches += o.propagate(it)["patches"]
    single = o.patches()
    assert patches > 500 and single.shape[0] > seeds.shape[0]

    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), ranged), nprocs=world, join=True)
    pools = [np.load(tmp_path / f"pool_{r}.npy").view(ob.PATCH_DTYPE).reshape(-1) for r in range(world)]
    n = int(sum(np.load(tmp_path / f"patches_{r}.npy")[0] for r in range(world)))
    assert n == patches  # the ranks together did exactly the single rank's work
    for f in ("coord", "normal", "ncc", "dscale", "nimages", "images", "nvimages", "vimages"):
        for p in pools:
            np.testing.assert_array_equal(p[f], single[f], err_msg=f)  # replicated pools stay identical and equal the 1-rank result


def test_merge_in_view_order():
    from mvskit_amd.dist import merge_in_view_order

    # 5 views, 2 ranks: rank 0 owns 0,2,4; rank 1 owns 1,3
    recs = [np.array([[0], [0], [2], [4], [4], [4]]), np.array([[1], [3], [3]])]
    counts = [np.array([2, 0, 1, 0, 3]), np.array([0, 1, 0, 2, 0])]
    out = np.concatenate(merge_in_view_order(recs, counts, 5, 2)).ravel()
    np.testing.assert_array_equal(out, [0, 0, 1, 2, 3, 3, 4, 4, 4])


def test_merge_in_view_order_split_views():
    from mvskit_amd.dist import merge_in_view_order

    # 3 views cut into 2 contiguous job ranges: rank 0 holds view 0 and the first part of view 1, rank 1 the rest
    recs = [np.array([[0], [0], [10], [11]]), np.array([[12], [20], [21]])]
    counts = [np.array([2, 2, 0]), np.array([0, 1, 2])]
    out = np.concatenate(merge_in_view_order(recs, counts, 3, 2)).ravel()
    np.testing.assert_array_equal(out, [0, 0, 10, 11, 12, 20, 21])
