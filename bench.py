#!/usr/bin/env python3
"""bench.py -- patches/s of the propagate+optim iteration on a 12-view 1920x1080 synthetic scene.

One "step" = one Propagate::run(iter) (pmmvps/propagate.cpp:28-64) over all views: two colour passes, each an
index build + the sweep kernel + the commit, followed by PmMvps::updateThreshold and ++m_depth (pmmvps.cpp:70-74,105).
The steps walk the schedule of PmMvps::run (pmmvps.cpp:90-105): iterations 0, 1, 2 with nccThreshold 0.7 / 0.65 / 0.6 and
m_depth 1 / 2 / 3, then the state is reset (pool cleared, seeds uploaded again, thresholds back) OUTSIDE the timed region
and the schedule starts over -- so any --steps measures BASELINE.json configs[1] ("3 iterations") and the thresholds never
leave [0.6, 0.7].  `value` = patches (candidates that reached Optim::preProcess, propagate.cpp:182) of all ranks /
time of the K timed steps (max over ranks), inputs resident in HBM.  See DESIGN.md "Measurement".

    python bench.py --gpus 1 --steps 3 --warmup 1
    python bench.py --gpus N ...              (spawns N rank processes itself)
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_VIEW_EVAL = 588  # 49 samples x 4 texels x 3 B (optim.cpp:835-842 x image.cpp:462-470), SURVEY.md 8d
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
SCHEDULE_ITERS = 3             # ITER of PmMvps::run, pmmvps.cpp:90
NCC0, NCC_BEFORE0, DEPTH0 = 0.7, 0.4, 1  # Option::m_nccThreshold, nccThreshold - 0.3 (pmmvps.cpp:57), m_depth after createPatches (pmmvps.cpp:85)


def log(*a):
    print("[bench]", *a, file=sys.stderr, flush=True)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--views", type=int, default=12)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--seed-stride", type=int, default=2, help="one seed patch per stride x stride cells per view")
    ap.add_argument("--refine-steps", type=int, default=6)  # four proposals per step: 1 + 4 * 6 = 25 cost evaluations per candidate
    ap.add_argument("--cpu-seconds", type=float, default=20.0, help="target duration of the CPU baseline sample (0 = skip)")
    ap.add_argument("--filter", action="store_true", help="BASELINE config 5: run Filter::run (filter.cpp:25-49) after every iteration, inside the timed region")
    ap.add_argument("--no-config5", dest="config5", action="store_false", help="skip the extra repetition with Filter::run that fills the `config5` block")
    ap.add_argument("--no-config4", dest="config4", action="store_false", help="skip the BASELINE configs[3] block (48 views of 3840x2160, five iterations with Filter::run, the 64-view library)")
    ap.add_argument("--force-exchange", action="store_true", help="rehearsal: run the N>1 code path (RCCL exchange) with a world of 1")
    ap.add_argument("--list-cap", type=int, default=0, help="views per m_images / m_vimages list = which engine library (16, 32, 64); 0 = the smallest that holds --views")
    ap.add_argument("--scene-cache", default=os.path.join("/tmp", "mvskit_scene_cache"))
    return ap.parse_args()


def spawn_ranks(args) -> int:
    """`python bench.py --gpus N` without a launcher: start N rank processes (fresh interpreters, so nothing that has
    touched the GPU is ever re-executed) and wait for them.  Rank 0 prints the JSON line on the inherited stdout."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # poll all ranks: when one exits non-zero its siblings (possibly waiting for it in a collective) are terminated
    rc = 0
    live = list(procs)
    while live:
        time.sleep(0.2)
        for p in list(live):
            r = p.poll()
            if r is None:
                continue
            live.remove(p)
            if r != 0:
                rc = max(rc, abs(r))
                log(f"rank process {p.pid} exited with {r}: terminating the other ranks")
                for q in live:
                    q.terminate()
                t_kill = time.time() + 10.0
                for q in live:
                    try:
                        q.wait(timeout=max(0.1, t_kill - time.time()))
                    except subprocess.TimeoutExpired:
                        q.kill()
                        q.wait()
                live = []
                break
    return rc


def load_scene(args, rank):
    """Synthetic cfg2 scene + seeds; cached on local disk so that N ranks do not all render it."""
    import numpy as np

    from mvskit_amd import synth

    tag = f"v{args.views}_{args.width}x{args.height}_s{args.seed_stride}"
    path = os.path.join(args.scene_cache, tag + ".npz")
    if os.path.exists(path):
        z = np.load(path)
        sc = synth.Scene(W=args.width, H=args.height, P=z["P"], images=z["images"], centers=z["centers"])
        return sc, z["seeds"].view(synth.PATCH_DTYPE).reshape(-1)
    sc = synth.make_scene(nviews=args.views, W=args.width, H=args.height, arc_deg=110.0, radius=4.0, kind="multi")
    seeds = synth.make_seeds(sc, level=0, csize=2, stride=args.seed_stride, seed=777)
    sc.points = None
    sc.normals = None
    if rank == 0:
        os.makedirs(args.scene_cache, exist_ok=True)
        tmp = path + f".tmp{os.getpid()}.npz"
        np.savez(tmp, P=sc.P, images=sc.images, centers=sc.centers, seeds=seeds.view(np.uint8))
        os.replace(tmp, path)
    return sc, seeds


def cpu_baseline(args, sc, seeds, pool_after_iter0, gpu_patches_by_iter):
    """The oracle (oracle/pmmvs_oracle.cpp) in its FAITHFUL schedule -- sequential raster sweep, one thread, the
    reference's execution model -- on two bounded samples of the same workload, so that the work mix is the GPU's:
      A. iteration 0 (m_depth 1, no Optim::check) from the seeds;
      B. iteration 1 (m_depth 2, nccThreshold 0.65, Optim::check active) from the pool the GPU held after iteration 0.
    Each runs the source cells of view 0 in sweep order until half of --cpu-seconds has passed.  The reported rate weights
    the two by the GPU's own patch counts (iteration 0 -> A; iterations 1, 2 -> B)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob

    half = max(1.0, args.cpu_seconds / 2)
    kw = dict(level=0, csize=2, wsize=7, minImageNum=3, schedule=ob.SCHEDULE_FAITHFUL, sum_mode=ob.SUM_SEQ, refine_steps=args.refine_steps, seed=1,
              list_cap=args.list_cap, wide=args.list_cap > 32)
    samples = []
    for name, it, recs, check in (("A", 0, seeds, 0), ("B", 1, pool_after_iter0, 1)):
        if recs is None:
            continue
        o = ob.Oracle(sc.nviews, enable_check=check, **kw)
        o.set_scene(sc)
        o.add_patches(recs)
        for _ in range(it):
            o.update_threshold()
        o.set_time_budget(half)
        t0 = time.perf_counter()
        c = o.propagate(it)
        spent = time.perf_counter() - t0
        o.close()
        samples.append({"name": name, "iter": it, "patches": c["patches"], "view_evals": c["view_evals"], "seconds": spent,
                        "rate": c["patches"] / spent if spent > 0 else 0.0})
        log(f"cpu sample {name}: {c['patches']} patches, {c['view_evals']} view evaluations in {spent:.1f} s")
    wa = gpu_patches_by_iter[0]
    wb = sum(gpu_patches_by_iter[1:]) if len(samples) > 1 else 0
    sec_per_patch = (wa / samples[0]["rate"] + (wb / samples[1]["rate"] if wb else 0.0)) / max(wa + wb, 1)
    res = {"value": 1.0 / sec_per_patch if sec_per_patch > 0 else 0.0, "unit": "patches/s", "cores": 1, "kind": "port",
           "sample": f"oracle faithful schedule, one thread, same scene, source cells of view 0 in sweep order for {half:.0f} s each: "
                     + "; ".join(f"{s['name']} = iteration {s['iter']} ({'seeds, m_depth 1' if s['iter'] == 0 else 'the pool of the GPU run after iteration 0, m_depth 2, Optim::check on'}): "
                                 f"{s['patches']} patches in {s['seconds']:.1f} s = {s['rate']:.0f} patches/s" for s in samples)
                     + f"; weighted by the GPU run's patches per iteration {gpu_patches_by_iter}",
           "view_evals_per_s": sum(s["view_evals"] for s in samples) / max(sum(s["seconds"] for s in samples), 1e-9)}
    # SURVEY 8(d) asks for the all-core figure as well: the engine schedule (what the GPU runs), OpenMP over the
    # destination cells of one colour pass, every host core this process may use
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, int(os.environ.get("MVS_CPU_THREADS", "16"))))  # a one-GPU box's CPU share is 16 cores
    o = ob.Oracle(sc.nviews, level=0, csize=2, wsize=7, minImageNum=3, schedule=ob.SCHEDULE_ENGINE, sum_mode=ob.SUM_TREE64,
                  enable_check=0, refine_steps=args.refine_steps, seed=1, nthreads=cores, list_cap=args.list_cap, wide=args.list_cap > 32)
    o.set_scene(sc)
    o.add_patches(seeds)
    o.set_time_budget(max(1.0, args.cpu_seconds / 3))
    c = o.engine_pass(0, 1)  # the seeds sit on even cells: their destinations have colour 1
    spent = o.last_sweep_seconds()  # the parallel sweep alone; the (serial) index build before it is not counted
    o.close()
    res["all_cores"] = {"value": c["patches"] / spent if spent > 0 else 0.0, "unit": "patches/s", "cores": cores,
                        "sample": f"oracle engine schedule, OpenMP over destination cells, colour pass 1 of iteration 0 until {max(1.0, args.cpu_seconds / 3):.0f} s "
                                  f"had passed: {c['patches']} patches in {spent:.1f} s"}
    return res


def run_config4(log_fn):
    """BASELINE configs[3] on ONE MI355X: 48 views of 3840x2160 (99.5 M cells of 2x2 pixels), the 64-view library (no list is cut),
    seeds in ALL cells' neighbourhoods of ALL views (one per 4x4 cells: 6.2 M), and the five iterations of PmMvps::run's loop
    (pmmvps.cpp:90-110): Propagate::run, Filter::run, updateThreshold, ++m_depth.  Its own scene, engine and clock; the clock runs
    around the five (Propagate::run + Filter::run) pairs, inputs resident."""
    import numpy as np
    import torch

    from mvskit_amd import engine as eng
    from mvskit_amd import synth

    t0 = time.perf_counter()
    sc = synth.make_scene(nviews=48, W=3840, H=2160, arc_deg=110.0, radius=4.0, kind="multi")
    seeds = synth.make_seeds(sc, level=0, csize=2, stride=4, seed=777)
    sc.points = None
    sc.normals = None
    t_scene = time.perf_counter() - t0
    torch.cuda.synchronize()
    free0, total_mem = torch.cuda.mem_get_info()
    max_patches = 128_000_000
    e = eng.Engine(48, level=0, csize=2, wsize=7, minImageNum=3, enable_check=1, seed=1, nccThreshold=NCC0, depth=DEPTH0, max_patches=max_patches)
    e.set_scene(sc)
    e.upload_patches(seeds)
    torch.cuda.synchronize()
    log_fn(f"config4: scene + seeds in {t_scene:.0f} s, {seeds.shape[0]} seeds, list_cap {e.list_cap}")
    its = []
    patches = view_evals = 0
    sweep_ms = index_ms = commit_ms = filter_ms = 0.0
    launches = 0
    peak = 0.0
    timed = 0.0
    for it in range(5):
        torch.cuda.synchronize()
        ta = time.perf_counter()
        c = e.propagate(it)
        t = e.timing()
        tb = time.perf_counter()
        removed = e.filter()
        fs = e.filter_stats()
        e.update_threshold()
        torch.cuda.synchronize()
        tc = time.perf_counter()
        timed += tc - ta
        free1, _ = torch.cuda.mem_get_info()
        peak = max(peak, (free0 - free1) / 2 ** 30)
        patches += c["patches"]; view_evals += c["view_evals"]
        sweep_ms += t["sweep_ms"]; index_ms += t["index_ms"]; commit_ms += t["commit_ms"]; launches += t["sweep_launches"]; filter_ms += fs["total_ms"]
        alive = e.num_patches()
        its.append({"iteration": it, "patches": c["patches"], "inserted": c["inserted"], "replaced": c["replaced"], "trimmed": c["trimmed"], "check_rejected_or_failed": c["fail1"],
                    "propagate_ms": 1000.0 * (tb - ta), "sweep_ms": t["sweep_ms"], "index_ms": t["index_ms"], "filter_ms": fs["total_ms"],
                    "filter_stage_ms": {k: fs[k] for k in ("outside_ms", "exact_ms", "neighbor_ms", "groups_ms", "rebuild_ms")}, "filter_removed": removed,
                    "pool_alive": alive, "hbm_used_GiB": (free0 - free1) / 2 ** 30, "check_retried_cells": t["check_retried_cells"]})
        log_fn(f"config4 iter {it}: {its[-1]}")
    e.close()
    alg = view_evals * ALG_BYTES_PER_VIEW_EVAL
    ach = alg / (sweep_ms * 1e-3) / 1e9 if sweep_ms > 0 else 0.0
    return {"workload": "BASELINE configs[3] on one GPU: 48 views of 3840x2160 on the 110 degree arc (99.5 M cells), level 0, csize 2, wsize 7, minImageNum 3, the 64-view "
                        "library (192-byte records, no list cut), 1 seed per 4x4 cells over ALL cells of ALL 48 views; the five iterations of PmMvps::run's loop: "
                        "Propagate::run + Filter::run + updateThreshold + ++m_depth (Optim::check from the second), timed together, inputs resident",
            "metric": "patches/s (propagate+optim iteration + Filter::run), 48-view 4K", "value": patches / timed if timed > 0 else 0.0, "unit": "patches/s",
            "steps": 5, "ms_per_step": 1000.0 * timed / 5, "patches": patches, "view_evals": view_evals, "seeds": int(seeds.shape[0]), "max_patches": max_patches,
            "list_cap": 64, "record_bytes": 192, "cells": 48 * 1920 * 1080, "hbm_peak_GiB": peak, "hbm_total_GiB": total_mem / 2 ** 30,
            "propagate_only_value": patches / ((sweep_ms + index_ms + commit_ms) * 1e-3) if sweep_ms > 0 else 0.0,
            "filter_ms_per_call": filter_ms / 5, "index_ms_per_step": index_ms / 5, "scene_generation_s": t_scene,
            "roofline": {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": None, "kernel": "k_sweep (64-view build)",
                         "launches": launches, "avg_launch_ms": sweep_ms / max(launches, 1), "algorithmic_bytes_per_launch": alg / max(launches, 1)},
            "iterations": its}


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))  # nothing has touched the GPU in this process

    # The contract is ONE JSON line on stdout.  Libraries print there too (RCCL's version banner at communicator set-up, Gloo's
    # connection notes): from here on file descriptor 1 is stderr, and the JSON line goes to a private copy of the real stdout.
    real_stdout = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}")
    ndev = torch.cuda.device_count()
    if ndev < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the engine has no CPU path)")
    shared_gpu = world > ndev  # rehearsal on a box with fewer GPUs than ranks: RCCL refuses two ranks on one device
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    dist = None
    if world > 1 or args.force_exchange:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if shared_gpu:
            dist.init_process_group(backend="gloo")
        elif world > 1:
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend="nccl", device_id=device, rank=0, world_size=1)

    from mvskit_amd import engine as eng
    from mvskit_amd import dist as mdist

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if rank == 0:
        sc, seeds = load_scene(args, rank)
    if world > 1:
        dist.barrier()
    if rank != 0:
        sc, seeds = load_scene(args, rank)

    e = eng.Engine(args.views, list_cap=args.list_cap or None, level=0, csize=2, wsize=7, minImageNum=3, enable_check=1, seed=1, refine_steps=args.refine_steps,
                   shard_index=rank, shard_count=world, device=dev_index, nccThreshold=NCC0, depth=DEPTH0)
    args.list_cap = e.list_cap
    if rank == 0:
        log(f"scene ready: {sc.nviews} views {sc.W}x{sc.H}, {seeds.shape[0]} seeds; world {world}" + (" (ranks share a GPU: host-staged exchange over gloo)" if shared_gpu else ""))
    torch.cuda.synchronize()
    t_h0 = time.perf_counter()
    e.set_scene(sc)  # host images over PCIe + pyramids built on the device (synchronous)
    torch.cuda.synchronize()
    host_ms = {"set_views_ms": 1000.0 * (time.perf_counter() - t_h0)}
    # the cell indexes sized once, for MAX_NUM_OF_PATCHES entries per cell, instead of growing (free + allocate, gigabytes at a
    # time) inside the first iterations of the timed schedule -- set-up, like the pyramids; skipped where that bound is out of
    # proportion (a 4K many-view scene: the indexes then grow as they are needed)
    cells = args.views * ((args.width + 1) // 2) * ((args.height + 1) // 2)
    if cells * 8 * 120 <= (16 << 30):
        e.reserve(0)
    torch.cuda.synchronize()
    exchange = "none"
    ex = None
    if world > 1 or args.force_exchange:
        if shared_gpu and os.environ.get("MVS_CCL_LIBRARY"):
            # rehearsal of the N-GPU code path on fewer GPUs: the engine's own exchange, its collective library replaced (tests/loopback_ccl)
            ex = mdist.EngineExchange(e, device)
            exchange = "in-engine exchange through MVS_CCL_LIBRARY=" + os.environ["MVS_CCL_LIBRARY"] + " (rehearsal: ranks share one GPU)"
        elif shared_gpu:
            ex = mdist.HostStagedExchange(e, device, sc.nviews)
            exchange = "host-staged all-gather over gloo (rehearsal: ranks share one GPU)"
        else:
            ex = mdist.EngineExchange(e, device)
            exchange = ex.description

    def reset_state():
        """PmMvps::init thresholds + DepthNormInit::createPatches (pmmvps.cpp:54-67,83-85): outside the timed region."""
        e.clear_patches()
        t_u0 = time.perf_counter()
        e.upload_patches(seeds)
        torch.cuda.synchronize()
        host_ms["upload_patches_ms"] = 1000.0 * (time.perf_counter() - t_u0)
        e.set_thresholds(NCC0, NCC_BEFORE0, DEPTH0)

    def step(it, with_filter=args.filter):
        ts = time.perf_counter()
        thr = e.thresholds()
        assert 0.6 - 1e-4 <= thr[0] <= 0.7 + 1e-4 and 1 <= thr[2] <= SCHEDULE_ITERS, thr
        c = ex.propagate(it) if ex else e.propagate(it)
        t = dict(ex.last_timing) if ex else e.timing()
        if with_filter:
            tf = time.perf_counter()
            c["filter_removed"] = e.filter()
            t["filter_ms"] = 1000.0 * (time.perf_counter() - tf)
            c["filter_stats"] = e.filter_stats()
        e.update_threshold()
        if rank == 0:
            log(f"iter {it} (nccThreshold {thr[0]:.2f}, m_depth {thr[2]}): {time.perf_counter() - ts:.3f} s, patches {c['patches']}, candidates {c['candidates']}, "
                f"view_evals {c['view_evals']}, inserted {c['inserted']}, replaced {c['replaced']}, timing {t}" + (f", filter removed {c['filter_removed']}" if with_filter else ""))
        return c, t

    # warm-up: W steps of the same schedule
    for s in range(args.warmup):
        if s % SCHEDULE_ITERS == 0:
            reset_state()
        step(s % SCHEDULE_ITERS)

    def pool_digest(recs):
        import hashlib

        b = recs[["coord", "normal", "ncc", "dscale", "nimages", "nvimages", "images", "vimages"]].tobytes() if recs.shape[0] else b""
        return int.from_bytes(hashlib.blake2b(b, digest_size=7).digest(), "little") ^ (int(recs.shape[0]) << 1)

    pool_hashes = []
    patches = view_evals = evals = 0
    sweep_ms = index_ms = commit_ms = exchange_ms = 0.0
    launches = exchange_bytes = local_view_evals = 0
    patches_by_iter = [0] * SCHEDULE_ITERS
    sec_by_iter = [0.0] * SCHEDULE_ITERS
    steps_by_iter = [0] * SCHEDULE_ITERS
    fstats = {}
    pool_after_iter0 = None
    timed = 0.0
    for s in range(args.steps):
        it = s % SCHEDULE_ITERS
        if it == 0:
            reset_state()  # untimed: the clock runs only around the steps
        barrier()
        t0 = time.perf_counter()
        c, t = step(it)
        barrier()
        timed += time.perf_counter() - t0
        sec_by_iter[it] += time.perf_counter() - t0
        steps_by_iter[it] += 1
        patches += c["patches"]; view_evals += c["view_evals"]; evals += c["evals"]
        patches_by_iter[it] += c["patches"]
        sweep_ms += t["sweep_ms"]; index_ms += t["index_ms"]; commit_ms += t["commit_ms"]; launches += t["sweep_launches"]
        exchange_ms += t.get("exchange_ms", 0.0); exchange_bytes += t.get("exchange_bytes", 0)
        local_view_evals += c["view_evals"]
        for k, v in c.get("filter_stats", {}).items():
            fstats[k] = fstats.get(k, 0) + v
        if it == 0 and pool_after_iter0 is None and world == 1 and args.cpu_seconds > 0 and not args.filter:
            pool_after_iter0 = e.patches()  # for the CPU baseline's sample B (untimed: after the closing barrier)
        if world > 1 and it == SCHEDULE_ITERS - 1:
            # self-validation of a multi-rank run (untimed): every rank hashes the pool it holds at the end of a repetition; the hashes
            # are all-gathered below.  The result does not depend on the sharding, so all ranks must hold the SAME pool -- and `patches`
            # of the whole job must equal the one-GPU run's for the same --steps.
            pool_hashes.append(pool_digest(e.patches()))
    dt = timed
    if world > 1:
        tt = torch.tensor([dt, float(patches), float(view_evals)], dtype=torch.float64, device="cpu" if shared_gpu else device)
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tt.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt = float(mx[0]); patches = int(sm[1]); view_evals = int(sm[2])
    validation = None
    if world > 1:
        info = e.comm_info() if ex is not None else {"world": 0, "comm_count": -1}
        mine = torch.tensor((pool_hashes + [0] * 64)[:64] + [len(pool_hashes), info["world"], info["comm_count"]], dtype=torch.int64, device="cpu" if shared_gpu else device)
        allh = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allh, mine)
        allh = [t.cpu().tolist() for t in allh]
        nrep = allh[0][64]
        validation = {"ranks_identical": bool(nrep > 0 and all(a[:65] == allh[0][:65] for a in allh)), "repetitions_hashed": int(nrep),
                      "rccl_world": [int(a[66]) for a in allh], "engine_world": [int(a[65]) for a in allh],
                      "note": "every rank hashes its pool (coordinates, normals, scores, lists) after each repetition; identical hashes on all ranks = identical pools; "
                              "rccl_world = ncclCommCount of each rank's communicator (-1: the transport cannot be asked, 0 where the host-staged rehearsal uses none)"}
    # BASELINE configs[4] ("12-view 1080p with filter.cpp geometric-consistency pass fused on GPU") beside the headline: one more
    # repetition of the 3-iteration schedule with Filter::run (filter.cpp:25-49) after every iteration, as PmMvps::run has it
    # (pmmvps.cpp:95-105), on its own clock -- the headline `value` above is not touched by it.
    cfg5 = None
    if not args.filter and args.config5:
        f5 = {}
        p5 = 0
        t5 = 0.0
        f5_ms = 0.0
        for it in range(SCHEDULE_ITERS):
            if it == 0:
                reset_state()
            barrier()
            t0 = time.perf_counter()
            c, t = step(it, True)
            barrier()
            t5 += time.perf_counter() - t0
            p5 += c["patches"]
            f5_ms += t["filter_ms"]
            for k, v in c["filter_stats"].items():
                f5[k] = f5.get(k, 0) + v
        if world > 1:
            tt = torch.tensor([t5, float(p5)], dtype=torch.float64, device="cpu" if shared_gpu else device)
            mx = tt.clone()
            dist.all_reduce(mx, op=dist.ReduceOp.MAX)
            sm = tt.clone()
            dist.all_reduce(sm, op=dist.ReduceOp.SUM)
            t5 = float(mx[0]); p5 = int(sm[1])
        cfg5 = (p5, t5, f5, f5_ms)

    n_alive = e.num_patches()
    t_d0 = time.perf_counter()
    final_pool = e.patches()  # what a caller takes back over PCIe after the last iteration
    host_ms["download_patches_ms"] = 1000.0 * (time.perf_counter() - t_d0)
    host_ms["downloaded_patches"] = int(final_pool.shape[0])
    del final_pool

    if rank == 0:
        reps, extra = divmod(args.steps, SCHEDULE_ITERS)
        out = {
            "metric": "patches/s (propagate+optim iteration), 12-view 1080p",
            "value": patches / dt,
            "unit": "patches/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1000.0 * dt / max(args.steps, 1),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.views}-view {args.width}x{args.height} synthetic scene (3 planes + sphere), level 0, csize 2, wsize 7, minImageNum 3, "
                                   f"1 seed per {args.seed_stride}x{args.seed_stride} cells per view; the 3-iteration schedule of PmMvps::run (nccThreshold 0.70/0.65/0.60, "
                                   f"m_depth 1/2/3, Optim::check from m_depth 2) run {reps} time(s)" + (f" + its first {extra} iteration(s)" if extra else "")
                                   + f" = {args.steps} timed steps after {args.warmup} warm-up step(s); pool and thresholds reset between repetitions outside the timed region",
                       "views": args.views, "width": args.width, "height": args.height, "csize": 2, "wsize": 7, "list_cap": args.list_cap, "iterations_per_repetition": SCHEDULE_ITERS,
                       "repetitions": reps, "extra_iterations": extra, "refine_evals": 1 + 4 * args.refine_steps, "check_depth2": True, "filter_run": bool(args.filter),
                       "parallelism": "single GPU" if world == 1 else f"the (view, cell) sequence sharded in {world} contiguous ranges over {world} ranks; per colour pass: {exchange}"},
            "patches": patches,
            "patches_by_iteration": patches_by_iter if world == 1 else None,
            # per iteration of the schedule (mean over the timed steps that ran it): iteration 0 has no Optim::check and is the fast one;
            # `value_per_repetition` weights the three equally whatever --steps is (with --steps 20 the headline holds 7 / 7 / 6 of them)
            "ms_by_iteration": [1000.0 * sec_by_iter[i] / max(steps_by_iter[i], 1) for i in range(SCHEDULE_ITERS)],
            "steps_by_iteration": steps_by_iter,
            "view_evals": view_evals,
            "pool_alive": n_alive,
        }
        if world == 1 and all(steps_by_iter):
            out["patches_per_s_by_iteration"] = [patches_by_iter[i] / sec_by_iter[i] for i in range(SCHEDULE_ITERS)]
            out["value_per_repetition"] = sum(patches_by_iter[i] / steps_by_iter[i] for i in range(SCHEDULE_ITERS)) / sum(sec_by_iter[i] / steps_by_iter[i] for i in range(SCHEDULE_ITERS))
        if sweep_ms > 0:
            alg = local_view_evals * ALG_BYTES_PER_VIEW_EVAL  # rank 0's own launches (= the whole job at N = 1)
            ach = alg / (sweep_ms * 1e-3) / 1e9
            traffic = traffic_from = None
            prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(prof):
                try:
                    pj = json.load(open(prof))
                    traffic = pj.get("hbm_bytes_per_launch")
                    traffic_from = f"profiles/pmc_traffic.json ({pj.get('command', 'separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes')}); not measured in this run"
                except Exception:
                    traffic = None
            avg_ms = sweep_ms / max(launches, 1)
            out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                               "traffic_from": traffic_from,
                               # north star: "achieved HBM GB/s" -- the counters' bytes per launch over THIS run's average launch time
                               "hbm_gbs_counter": (traffic / (avg_ms * 1e-3) / 1e9) if (traffic and avg_ms > 0) else None,
                               "kernel": "k_sweep", "launches": launches, "avg_launch_ms": sweep_ms / max(launches, 1),
                               "algorithmic_bytes_per_launch": alg / max(launches, 1),
                               "index_ms": index_ms, "commit_ms": commit_ms, "sweep_ms": sweep_ms,
                               "index_ms_per_step": index_ms / max(args.steps, 1), "commit_ms_per_step": commit_ms / max(args.steps, 1),
                               "note": "achieved = 588 B x view evaluations counted on the device / HIP-event time of k_sweep; the label is the contract's choice of "
                                       "two: the kernel is bound by VALU issue (DESIGN.md section 5) and the 131 MB of pyramids sit in the Infinity Cache"}
            if validation is not None:
                out["validation"] = validation
                out["ranks_identical"] = validation["ranks_identical"]
                out["rccl_world"] = validation["rccl_world"][0]
            if ex is not None:
                out["roofline"]["rank"] = 0
                out["exchange"] = {"ms": exchange_ms, "bytes_gathered_per_rank": exchange_bytes, "collective": exchange}
        def filter_roofline(f, calls):
            # Filter::run (filter.cpp:25-49) -- algorithmic bytes per DESIGN.md "Filter::run": what the two heavy stages have to
            # read and write if every datum moves once
            ex_bytes = f["exact_patches"] * (128 + 72) + f["exact_view_evals"] * (5 * (8 + 16) + ALG_BYTES_PER_VIEW_EVAL)
            nb_bytes = f["neighbor_patches"] * 128 + f["neighbor_tasks"] * 8 + f["neighbor_entries"] * 4 + f["neighbor_visited"] * 32 + f["neighbor_accepted"] * 16

            def blk(ms, nbytes, units, unit_name):
                gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
                return {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "ms": ms,
                        "algorithmic_bytes": nbytes, unit_name: units, "calls": calls}

            return {
                "filterExact": blk(f["exact_ms"], ex_bytes, f["exact_patches"], "patches"),
                "filterNeighbor": blk(f["neighbor_ms"], nb_bytes, f["neighbor_patches"], "patches"),
                "stage_ms": {k: f[k] for k in ("outside_ms", "exact_ms", "neighbor_ms", "groups_ms", "rebuild_ms", "total_ms")},
                "ms_per_call": f["total_ms"] / max(calls, 1),
                "counts": {k: f[k] for k in ("patches_in", "exact_view_evals", "neighbor_tasks", "neighbor_entries", "neighbor_visited", "neighbor_accepted", "neighbor_retried")},
                "note": "per stage: HIP-event time of its kernel(s) summed over the calls; bytes = record (128 B) + per surviving view 5 depth-map cells "
                        "(8 B) with the patch each names (16 B) + 588 B of setRefImage samples + 72 B of lists written (filterExact); record + 8 B per "
                        "list opened + 4 B per id walked + 32 B per distinct patch met (its packed geometry) + 16 B per neighbour fitted (filterNeighbor)"}

        if fstats:
            out["roofline_filter"] = filter_roofline(fstats, args.steps)
        if cfg5 is not None:
            p5, t5, f5, f5_ms = cfg5
            out["config5"] = {"workload": "BASELINE configs[4]: the same scene and schedule, one repetition (3 iterations) with Filter::run (filter.cpp:25-49) after every "
                                          "iteration inside the timed region, run after the headline loop on its own clock",
                              "metric": "patches/s (propagate+optim iteration + Filter::run), 12-view 1080p", "value": p5 / t5 if t5 > 0 else 0.0, "unit": "patches/s",
                              "steps": SCHEDULE_ITERS, "ms_per_step": 1000.0 * t5 / SCHEDULE_ITERS, "patches": p5,
                              "filter_ms_per_call_host_clock": f5_ms / SCHEDULE_ITERS, "roofline_filter": filter_roofline(f5, SCHEDULE_ITERS)}
        if world == 1 and reps >= 1:
            # what a caller that hands over host buffers sees for ONE 3-iteration job: images and seeds in over PCIe (pyramids built on
            # the device), the three iterations, the patches back out.  Never `value` (inputs resident), reported beside it.
            per_job = dt / reps if extra == 0 else dt * SCHEDULE_ITERS / max(args.steps, 1)
            host_s = 1e-3 * (host_ms["set_views_ms"] + host_ms.get("upload_patches_ms", 0.0) + host_ms["download_patches_ms"])
            out["host_transfers"] = dict(host_ms, patches_per_job=patches / max(args.steps, 1) * SCHEDULE_ITERS,
                                         pcie_inclusive_value=(patches / max(args.steps, 1) * SCHEDULE_ITERS) / (per_job + host_s), unit="patches/s",
                                         note="one job = set_views (host RGB in, pyramids on the device) + upload_patches + 3 iterations + download of the pool")
        if world == 1 and args.config4 and not args.filter and args.views == 12 and args.width == 1920:
            # BASELINE configs[3] in the driver's own line: the 12-view engine goes first (its memory is the card's), then the 48 x 4K run
            e.close()
            e = None
            try:
                out["config4"] = run_config4(log)
            except Exception as err:  # the headline must not be lost to the extra block
                out["config4"] = {"error": f"{type(err).__name__}: {err}"}
        if world == 1 and args.cpu_seconds > 0:
            log("cpu baseline ...")
            out["cpu_baseline"] = cpu_baseline(args, sc, seeds, pool_after_iter0, patches_by_iter if patches_by_iter[0] else [1, 0, 0])
            out["cpu_baseline_all_cores"] = out["cpu_baseline"]["all_cores"]
        print(json.dumps(out), file=real_stdout, flush=True)
    if e is not None:
        e.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
