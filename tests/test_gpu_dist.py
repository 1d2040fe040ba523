"""Two ranks on one MI355X (two processes, HIP engine each, views 0,2,.. and 1,3,..) exchanging over gloo through host
staging must end with exactly the single-rank pool: the multi-process version of what bench.py --gpus N does with
RCCL (mvskit_amd/dist.py: same pass / export / merge-in-view-order / commit sequence)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _scene():
    from mvskit_amd import synth

    sc = synth.make_scene(nviews=5, W=256, H=160, arc_deg=60.0, radius=4.0, kind="multi")
    seeds = synth.make_seeds(sc, stride=3, seed=9)
    return sc, seeds


ITERS = 3  # m_depth reaches 2: Optim::check runs, m_vimages / m_vpgrids are exchanged too


def _worker(rank, world, port, out_dir, ranged):
    sys.path.insert(0, ROOT)
    import torch
    import torch.distributed as dist

    from mvskit_amd import dist as mdist
    from mvskit_amd import engine

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sc, seeds = _scene()
    shard = dict(shard_index=rank, shard_count=world) if ranged else dict(view_begin=rank, view_stride=world)
    e = engine.Engine(sc.nviews, level=0, csize=2, wsize=7, minImageNum=3, enable_check=1, seed=5, device=0, **shard)
    e.set_scene(sc)
    e.upload_patches(seeds)
    hs = mdist.HostStaged(e, torch.device("cuda", 0))
    ex = mdist.HostExchange()
    tot = 0
    for it in range(ITERS):
        tot += mdist.sharded_propagate_host(hs, it, ex, sc.nviews, engine.PATCH_DTYPE)["patches"]
        e.update_threshold()
    np.save(os.path.join(out_dir, f"pool_{rank}.npy"), e.patches().view(np.uint8))
    np.save(os.path.join(out_dir, f"patches_{rank}.npy"), np.array([tot]))
    dist.destroy_process_group()


@pytest.mark.parametrize("world,ranged", [(2, False), (3, True)])
def test_gpu_ranks_equal_one_rank(tmp_path, world, ranged):
    """world 2 sharded by whole views; world 3 by contiguous job ranges (5 views: views are split between ranks) --
    what bench.py --gpus N uses."""
    from mvskit_amd import engine

    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), ranged), nprocs=world, join=True)  # children first: this process has not touched the GPU yet
    sc, seeds = _scene()
    e = engine.Engine(sc.nviews, level=0, csize=2, wsize=7, minImageNum=3, enable_check=1, seed=5)
    e.set_scene(sc)
    e.upload_patches(seeds)
    patches = 0
    for it in range(ITERS):
        patches += e.propagate(it)["patches"]
        e.update_threshold()
    single = e.patches()
    assert patches > 2000 and single.shape[0] > seeds.shape[0]
    pools = [np.load(tmp_path / f"pool_{r}.npy").view(engine.PATCH_DTYPE).reshape(-1) for r in range(world)]
    n = int(sum(np.load(tmp_path / f"patches_{r}.npy")[0] for r in range(world)))
    assert n == patches
    for f in ("coord", "normal", "ncc", "dscale", "nimages", "images", "nvimages", "vimages"):
        for p in pools:
            np.testing.assert_array_equal(p[f], single[f], err_msg=f)
