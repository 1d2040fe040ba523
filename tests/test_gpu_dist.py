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


LOOPBACK_DIR = os.path.join(ROOT, "tests", "loopback_ccl")
LOOPBACK_LIB = os.path.join(LOOPBACK_DIR, "_build", "libloopback_ccl.so")


def _worker_c_abi(rank, world, out_dir):
    """One engine rank driven through the C ABI alone: mvs_engine_comm_init + mvs_engine_propagate (pass, mvs_engine_exchange,
    commit of the union).  The collective library behind the engine is the shared-memory loopback (RCCL refuses ranks that
    share a device); everything above it -- counts, offsets, in-place block broadcasts, kill ids, commit -- is the product code."""
    import time

    sys.path.insert(0, ROOT)
    os.environ["MVS_CCL_LIBRARY"] = LOOPBACK_LIB
    from mvskit_amd import engine

    sc, seeds = _scene()
    e = engine.Engine(sc.nviews, level=0, csize=2, wsize=7, minImageNum=3, enable_check=1, seed=5, device=0, shard_index=rank, shard_count=world)
    e.set_scene(sc)
    e.upload_patches(seeds)
    uid_path = os.path.join(out_dir, "uid.bin")
    if rank == 0:
        with open(uid_path + ".tmp", "wb") as f:
            f.write(e.comm_unique_id())
        os.replace(uid_path + ".tmp", uid_path)
    t0 = time.time()
    while not os.path.exists(uid_path):
        assert time.time() - t0 < 120, "rank 0 never published the communicator id"
        time.sleep(0.05)
    with open(uid_path, "rb") as f:
        uid = f.read()
    e.comm_init(uid, rank, world)
    tot, moved, fmoved = 0, 0, 0
    removed = []
    for it in range(ITERS):
        tot += e.propagate(it)["patches"]
        moved += e.timing()["exchange_bytes"]
        removed.append(list(e.filter().values()))  # Filter::run, collective: this rank filters its share of the pool
        fmoved += e.filter_stats()["exchange_bytes"]
        e.update_threshold()
    np.save(os.path.join(out_dir, f"pool_{rank}.npy"), e.patches().view(np.uint8))
    np.save(os.path.join(out_dir, f"patches_{rank}.npy"), np.array([tot, moved, fmoved]))
    np.save(os.path.join(out_dir, f"removed_{rank}.npy"), np.array(removed))
    e.comm_release()


@pytest.mark.parametrize("world", [2, 3])
def test_c_abi_exchange_ranks_equal_one_rank(tmp_path, world):
    """mvs_engine_exchange with world > 1 on one GPU: `world` processes, contiguous job ranges of equal work, the engine's own
    exchange (what bench.py --gpus N and PmMvps::setRanks use on a node) over the loopback transport, and Filter::run after every
    iteration with each rank filtering its share of the pool; every rank must end with exactly the pool of the one-rank run."""
    import subprocess

    from mvskit_amd import engine

    subprocess.check_call(["make", "-C", LOOPBACK_DIR, "-s"])
    mp.spawn(_worker_c_abi, args=(world, str(tmp_path)), nprocs=world, join=True)
    sc, seeds = _scene()
    e = engine.Engine(sc.nviews, level=0, csize=2, wsize=7, minImageNum=3, enable_check=1, seed=5)
    e.set_scene(sc)
    e.upload_patches(seeds)
    patches = 0
    removed = []
    for it in range(ITERS):
        patches += e.propagate(it)["patches"]
        removed.append(list(e.filter().values()))
        e.update_threshold()
    single = e.patches()
    pools = [np.load(tmp_path / f"pool_{r}.npy").view(engine.PATCH_DTYPE).reshape(-1) for r in range(world)]
    stats = [np.load(tmp_path / f"patches_{r}.npy") for r in range(world)]
    assert int(sum(s[0] for s in stats)) == patches and patches > 2000
    assert all(int(s[1]) > 0 and int(s[2]) > 0 for s in stats)  # every rank received blocks from the others, in the sweep and in Filter::run
    assert sum(sum(r) for r in removed) > 0
    for r in range(world):  # the four removal counts of every Filter::run, the same on every rank as on one
        np.testing.assert_array_equal(np.load(tmp_path / f"removed_{r}.npy"), np.array(removed))
    for f in ("coord", "normal", "ncc", "dscale", "nimages", "images", "nvimages", "vimages"):
        for p in pools:
            np.testing.assert_array_equal(p[f], single[f], err_msg=f)


def _worker_failing_rank(rank, world, out_dir, mode):
    """mode "fault": the pass (iteration 1, colour 0) of rank 1 reports a capacity failure (MVS_FAULT_PASS);
    mode "pool": rank 1's pool has room for the seeds only (mvs_config.max_patches), rank 0's the default capacity."""
    import time

    sys.path.insert(0, ROOT)
    os.environ["MVS_CCL_LIBRARY"] = LOOPBACK_LIB
    from mvskit_amd import build
    if mode == "fault":
        os.environ["MVS_FAULT_PASS"] = "1:1:0"
    if mode.startswith("filter"):
        os.environ["MVS_FAULT_FILTER"] = "1:" + mode[len("filter"):]
    if mode != "pool":  # the hooks exist only in the test build of the engine (-DMVS_FAULT_INJECTION)
        os.environ["MVS_ENGINE_LIB"] = build.build_engine(fault_injection=True)
    from mvskit_amd import engine

    sc, seeds = _scene()
    kw = dict(max_patches=int(seeds.shape[0])) if (mode == "pool" and rank == 1) else {}
    e = engine.Engine(sc.nviews, level=0, csize=2, wsize=7, minImageNum=3, enable_check=1, seed=5, device=0, shard_index=rank, shard_count=world, **kw)
    e.set_scene(sc)
    e.upload_patches(seeds)
    uid_path = os.path.join(out_dir, "uid.bin")
    if rank == 0:
        with open(uid_path + ".tmp", "wb") as f:
            f.write(e.comm_unique_id())
        os.replace(uid_path + ".tmp", uid_path)
    t0 = time.time()
    while not os.path.exists(uid_path):
        assert time.time() - t0 < 120
        time.sleep(0.05)
    with open(uid_path, "rb") as f:
        e.comm_init(f.read(), rank, world)
    log = []
    for it in range(ITERS):
        try:
            c = e.propagate(it)
            log.append((it, 0, c["patches"]))
        except engine.EngineError as err:
            log.append((it, err.status, 0))
            if mode == "fault":  # the pass was given up on every rank alike: the same iteration runs again and goes through
                c = e.propagate(it)
                log.append((it, 0, c["patches"]))
            else:
                break
        if mode.startswith("filter"):  # Filter::run after every iteration; rank 1 fails inside the first one
            try:
                e.filter()
                log.append((it, 0, -1))
            except engine.EngineError as err:
                log.append((it, err.status, -1))
                break
        e.update_threshold()
    np.save(os.path.join(out_dir, f"fail_log_{rank}.npy"), np.array(log, dtype=np.int64))
    np.save(os.path.join(out_dir, f"fail_pool_{rank}.npy"), e.patches().view(np.uint8))
    e.comm_release()


@pytest.mark.parametrize("point", [0, 1, 2, 3])
def test_filter_rank_local_failure_is_agreed_by_all_ranks(tmp_path, point):
    """The same inside the collective Filter::run: rank 1 fails on its own (MVS_FAULT_FILTER, in the test build of the engine) before the
    exchange of filterOutside's kill bytes (0), of filterExact's rewritten records (1), inside a rebuild (2) or before filterNeighbor's
    exchange (3).  An agreement stands in front of every collective of the call, so rank 0 -- which is fine -- learns of it at its next
    appointment: both ranks return MVS_ERR_CAPACITY from the same mvs_engine_filter and neither waits in a broadcast (the loopback's
    barrier would time out and the spawn fail)."""
    import subprocess

    subprocess.check_call(["make", "-C", LOOPBACK_DIR, "-s"])
    mp.spawn(_worker_failing_rank, args=(2, str(tmp_path), f"filter{point}"), nprocs=2, join=True)
    logs = [np.load(tmp_path / f"fail_log_{r}.npy") for r in range(2)]
    np.testing.assert_array_equal(logs[0][:, :2], logs[1][:, :2])
    assert logs[0].shape[0] == 2 and logs[0][0, 1] == 0 and logs[0][1, 1] == -4, logs[0]  # iteration 0 propagated, its Filter::run gave up


@pytest.mark.parametrize("mode", ["fault", "pool"])
def test_rank_local_failure_is_agreed_by_all_ranks(tmp_path, mode):
    """A failure on ONE rank -- its pass overflowed, or its pool is smaller than the other ranks' -- reaches every rank through
    the status word / the minimum headroom of mvs_engine_exchange's all-gather: all ranks return the same MVS_ERR_CAPACITY
    from the same mvs_engine_propagate, none is left waiting in a collective (the workers would run into the loopback's
    barrier time-out and the spawn would fail), and all hold identical pools afterwards."""
    import subprocess

    from mvskit_amd import engine

    subprocess.check_call(["make", "-C", LOOPBACK_DIR, "-s"])
    mp.spawn(_worker_failing_rank, args=(2, str(tmp_path), mode), nprocs=2, join=True)
    logs = [np.load(tmp_path / f"fail_log_{r}.npy") for r in range(2)]
    pools = [np.load(tmp_path / f"fail_pool_{r}.npy").view(engine.PATCH_DTYPE).reshape(-1) for r in range(2)]
    np.testing.assert_array_equal(logs[0][:, :2], logs[1][:, :2])  # the same statuses in the same iterations on both ranks
    failed = logs[0][logs[0][:, 1] != 0]
    assert failed.shape[0] == 1 and failed[0, 1] == -4, logs[0]    # MVS_ERR_CAPACITY, once
    assert failed[0, 0] == (1 if mode == "fault" else 0)
    assert pools[0].tobytes() == pools[1].tobytes()
    if mode == "fault":
        assert logs[0][-1, 0] == ITERS - 1 and logs[0][-1, 1] == 0  # the run went on to the end
        assert pools[0].shape[0] > _scene()[1].shape[0]


def _worker_host_mirror(rank, world, out_dir):
    """One rank of the C++ host mirror: PmMvps::setRanks(rank, world, id file) -> init -> run (Propagate::run + Filter::run per
    iteration), the engine exchanging inside Propagate::run; the collective library is the loopback (ranks share the GPU)."""
    import ctypes as C

    sys.path.insert(0, ROOT)
    os.environ["MVS_CCL_LIBRARY"] = LOOPBACK_LIB
    from mvskit_amd import build, engine

    engine.load_library()
    L = C.CDLL(build.build_host())
    L.mvshost_set_ranks.argtypes = [C.c_int, C.c_int, C.c_char_p, C.c_int]
    L.mvshost_set_ranks.restype = None
    L.mvshost_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint,
                              C.c_int, C.c_longlong, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p]
    sc, seeds = _scene()
    P = np.ascontiguousarray(sc.P, dtype=np.float32)
    img = np.ascontiguousarray(sc.images)
    sd = np.ascontiguousarray(seeds)
    cap = 400000
    out = np.zeros(cap, dtype=engine.PATCH_DTYPE)
    n, ptot = C.c_longlong(), C.c_longlong()
    L.mvshost_set_ranks(rank, world, os.path.join(out_dir, "comm.id").encode(), 0)
    r = L.mvshost_run(sc.nviews, sc.W, sc.H, P.ctypes.data, img.ctypes.data, 0, 2, 7, 3, C.c_float(0.7), 5, ITERS, sd.shape[0], sd.ctypes.data,
                      cap, out.ctypes.data, C.byref(n), C.byref(ptot))
    assert r == 0, r
    np.save(os.path.join(out_dir, f"hm_pool_{rank}.npy"), out[: n.value].view(np.uint8))
    np.save(os.path.join(out_dir, f"hm_total_{rank}.npy"), np.array([ptot.value]))


def test_host_mirror_two_ranks_equal_one_rank(tmp_path):
    """PmMvps::setRanks with world = 2 (mvskit_amd/host): rank 0 writes the communicator id file, rank 1 waits for it, both run
    the reference's driver sequence; every rank must return the patches of the plain one-rank PmMvps::run."""
    import ctypes as C
    import subprocess

    from mvskit_amd import build, engine

    subprocess.check_call(["make", "-C", LOOPBACK_DIR, "-s"])
    build.build_host()
    mp.spawn(_worker_host_mirror, args=(2, str(tmp_path)), nprocs=2, join=True)
    engine.load_library()
    L = C.CDLL(build.build_host())
    L.mvshost_run.argtypes = [C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_uint,
                              C.c_int, C.c_longlong, C.c_void_p, C.c_longlong, C.c_void_p, C.c_void_p, C.c_void_p]
    sc, seeds = _scene()
    P = np.ascontiguousarray(sc.P, dtype=np.float32)
    img = np.ascontiguousarray(sc.images)
    sd = np.ascontiguousarray(seeds)
    cap = 400000
    out = np.zeros(cap, dtype=engine.PATCH_DTYPE)
    n, ptot = C.c_longlong(), C.c_longlong()
    assert L.mvshost_run(sc.nviews, sc.W, sc.H, P.ctypes.data, img.ctypes.data, 0, 2, 7, 3, C.c_float(0.7), 5, ITERS, sd.shape[0], sd.ctypes.data,
                         cap, out.ctypes.data, C.byref(n), C.byref(ptot)) == 0
    single = out[: n.value]
    assert n.value > seeds.shape[0]
    for r in range(2):
        pool = np.load(tmp_path / f"hm_pool_{r}.npy").view(engine.PATCH_DTYPE).reshape(-1)
        assert pool.shape == single.shape
        assert pool.tobytes() == single.tobytes()
    # Propagate::m_pcount is this rank's share: the two shares add up to the one-rank count
    assert int(sum(np.load(tmp_path / f"hm_total_{r}.npy")[0] for r in range(2))) == ptot.value
