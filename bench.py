#!/usr/bin/env python3
"""bench.py -- patches/s of the propagate+optim iteration on a 12-view 1920x1080 synthetic scene.

One "step" = one Propagate::run(iter) (pmmvps/propagate.cpp:28-64) over all views: two colour passes, each an
index build + the sweep kernel + the commit, followed by PmMvps::updateThreshold (pmmvps.cpp:70-74).
`value` = patches (candidates that reached Optim::preProcess, propagate.cpp:182) of all ranks / wall time of
the K timed steps (max over ranks), inputs resident in HBM.  See DESIGN.md "Measurement".

    python bench.py --gpus 1 --steps 3 --warmup 1
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N --steps K --warmup W
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

ALG_BYTES_PER_VIEW_EVAL = 588  # 49 samples x 4 texels x 3 B (optim.cpp:835-842 x image.cpp:462-470), SURVEY.md 8d
HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s


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
    ap.add_argument("--refine-steps", type=int, default=8)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target duration of the CPU baseline sample (0 = skip)")
    ap.add_argument("--filter", action="store_true", help="BASELINE config 5: run Filter::run (filter.cpp:25-49) after every iteration, inside the timed region")
    ap.add_argument("--force-exchange", action="store_true", help="rehearsal: run the N>1 code path (RCCL exchange) with a world of 1")
    ap.add_argument("--scene-cache", default=os.path.join("/tmp", "mvskit_scene_cache"))
    return ap.parse_args()


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


def cpu_baseline(args, sc, seeds):
    """The oracle (oracle/pmmvs_oracle.cpp) in its FAITHFUL schedule -- sequential raster sweep, one thread, the
    reference's execution model -- on a bounded sample of the same workload: the first source cells of view 0
    in raster order until about --cpu-seconds have passed."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_binding as ob

    o = ob.Oracle(sc.nviews, level=0, csize=2, wsize=7, minImageNum=3, schedule=ob.SCHEDULE_FAITHFUL, sum_mode=ob.SUM_SEQ,
                  enable_check=0, refine_steps=args.refine_steps, seed=1)
    o.set_scene(sc)
    o.add_patches(seeds)
    o.set_time_budget(args.cpu_seconds)
    t0 = time.perf_counter()
    c = o.propagate(0)
    spent = time.perf_counter() - t0
    patches, evals = c["patches"], c["view_evals"]
    o.close()
    res = {"value": patches / spent if spent > 0 else 0.0, "unit": "patches/s", "cores": 1, "kind": "port",
           "sample": f"oracle faithful schedule, single thread, the source cells of view 0 in raster order (iteration 0) of the same scene until {args.cpu_seconds:.0f} s had passed: "
                     f"{patches} patches, {evals} view evaluations in {spent:.1f} s",
           "view_evals_per_s": evals / spent if spent > 0 else 0.0}
    # SURVEY 8(d) asks for the all-core figure as well: the engine schedule (what the GPU runs), OpenMP over the
    # destination cells of one colour pass, every host core this process may use
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(cores, int(os.environ.get("MVS_CPU_THREADS", "16"))))  # a one-GPU box's CPU share is 16 cores
    o = ob.Oracle(sc.nviews, level=0, csize=2, wsize=7, minImageNum=3, schedule=ob.SCHEDULE_ENGINE, sum_mode=ob.SUM_TREE64,
                  enable_check=0, refine_steps=args.refine_steps, seed=1, nthreads=cores)
    o.set_scene(sc)
    o.add_patches(seeds)
    o.set_time_budget(max(1.0, args.cpu_seconds / 2))
    t0 = time.perf_counter()
    c = o.engine_pass(0, 1)  # the seeds sit on even cells: their destinations have colour 1
    spent = o.last_sweep_seconds()  # the parallel sweep alone; the (serial) index build before it is not counted
    o.close()
    res["all_cores"] = {"value": c["patches"] / spent if spent > 0 else 0.0, "unit": "patches/s", "cores": cores,
                        "sample": f"oracle engine schedule, OpenMP over destination cells, colour pass 1 of iteration 0 until {max(1.0, args.cpu_seconds / 2):.0f} s "
                                  f"had passed: {c['patches']} patches in {spent:.1f} s"}
    return res


def main():
    args = parse()
    import numpy as np
    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus N > 1 must be launched with torch.distributed.run --nproc-per-node N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X (the engine has no CPU path)")
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1 or args.force_exchange:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if world > 1:
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend="nccl", device_id=device, rank=0, world_size=1)

    from mvskit_amd import engine as eng
    from mvskit_amd.dist import DeviceExchange

    if rank == 0:
        sc, seeds = load_scene(args, rank)
    if world > 1:
        dist.barrier()
    if rank != 0:
        sc, seeds = load_scene(args, rank)

    e = eng.Engine(args.views, level=0, csize=2, wsize=7, minImageNum=3, enable_check=1, seed=1, refine_steps=args.refine_steps,
                   shard_index=rank, shard_count=world, device=local_rank)
    if rank == 0:
        log(f"scene ready: {sc.nviews} views {sc.W}x{sc.H}, {seeds.shape[0]} seeds")
    e.set_scene(sc)
    e.upload_patches(seeds)
    ex = DeviceExchange(device) if (world > 1 or args.force_exchange) else None

    def step(it):
        ts = time.perf_counter()
        c = ex.propagate(e, it) if ex else e.propagate(it)
        t = dict(ex.last_timing) if ex else e.timing()
        if args.filter:
            tf = time.perf_counter()
            c["filter_removed"] = e.filter()
            t["filter_ms"] = 1000.0 * (time.perf_counter() - tf)
        e.update_threshold()
        if rank == 0:
            log(f"iter {it}: {time.perf_counter() - ts:.3f} s, patches {c['patches']}, candidates {c['candidates']}, view_evals {c['view_evals']}, "
                f"inserted {c['inserted']}, replaced {c['replaced']}, timing {t}" + (f", filter removed {c['filter_removed']}" if args.filter else ""))
        return c, t

    it = 0
    for _ in range(args.warmup):
        step(it)
        it += 1
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    patches = view_evals = evals = 0
    sweep_ms = index_ms = commit_ms = exchange_ms = 0.0
    launches = exchange_bytes = local_view_evals = 0
    for _ in range(args.steps):
        c, t = step(it)
        it += 1
        patches += c["patches"]; view_evals += c["view_evals"]; evals += c["evals"]
        sweep_ms += t["sweep_ms"]; index_ms += t["index_ms"]; commit_ms += t["commit_ms"]; launches += t["sweep_launches"]
        exchange_ms += t.get("exchange_ms", 0.0); exchange_bytes += t.get("exchange_bytes", 0)
        local_view_evals += c["view_evals"]
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt, float(patches), float(view_evals)], dtype=torch.float64, device=device)
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = tt.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        dt = float(mx[0]); patches = int(sm[1]); view_evals = int(sm[2])
    n_alive = e.num_patches()

    if rank == 0:
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
            "config": {"workload": f"{args.views}-view {args.width}x{args.height} synthetic scene (3 planes + sphere), level 0, csize 2, wsize 7, "
                                   f"minImageNum 3, 1 seed per {args.seed_stride}x{args.seed_stride} cells per view, {args.steps} iterations after {args.warmup} warm-up",
                       "views": args.views, "width": args.width, "height": args.height, "csize": 2, "wsize": 7,
                       "refine_evals": 1 + 3 * args.refine_steps, "check_depth2": True, "filter_run": bool(args.filter),
                       "parallelism": "single GPU" if world == 1 else f"the (view, cell) sequence sharded in {world} contiguous ranges over {world} GPUs, RCCL all-gather of patch records per colour pass"},
            "patches": patches,
            "view_evals": view_evals,
            "pool_alive": n_alive,
        }
        if sweep_ms > 0:
            alg = local_view_evals * ALG_BYTES_PER_VIEW_EVAL  # rank 0's own launches (= the whole job at N = 1)
            ach = alg / (sweep_ms * 1e-3) / 1e9
            traffic = None
            prof = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(prof):
                try:
                    traffic = json.load(open(prof)).get("hbm_bytes_per_launch")
                except Exception:
                    traffic = None
            out["roofline"] = {"bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                               "kernel": "k_sweep", "launches": launches, "avg_launch_ms": sweep_ms / max(launches, 1),
                               "algorithmic_bytes_per_launch": alg / max(launches, 1),
                               "index_ms": index_ms, "commit_ms": commit_ms, "sweep_ms": sweep_ms,
                               "note": "achieved = 588 B x view evaluations counted on the device / HIP-event time of k_sweep; the kernel is bound by VALU "
                                       "issue (over 90 % of SIMD time, DESIGN.md section 5), and the 131 MB of pyramids sit in the Infinity Cache"}
            if ex is not None:
                out["roofline"]["rank"] = 0
                out["exchange"] = {"ms": exchange_ms, "bytes_gathered_per_rank": exchange_bytes, "collective": "all_gather_into_tensor (RCCL)",
                                   "note": "counts + padded 128-byte patch records + kill ids, after each colour pass"}
        if world == 1 and args.cpu_seconds > 0:
            log("cpu baseline ...")
            out["cpu_baseline"] = cpu_baseline(args, sc, seeds)
        print(json.dumps(out))
    e.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
