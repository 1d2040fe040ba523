"""debug: the 48 x 4K windowed parity case, one iteration, patch-by-patch differences between the wide oracle and the cap64 engine"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_binding as ob  # noqa: E402
from mvskit_amd import engine, synth  # noqa: E402
from test_gpu_fullsize import _cells_in_ref_view  # noqa: E402

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (3840, 2160)
sc = synth.make_scene(nviews=48, W=W, H=H, arc_deg=110.0, radius=4.0, kind="multi")
views = [47, 46, 44, 40, 33, 20]
seeds = synth.make_seeds(sc, level=0, csize=2, stride=3, seed=31, views=views)
cx, cy = _cells_in_ref_view(sc, seeds)
ref = seeds["images"][:, 0].astype(np.int64)
gw, gh = W // 2, H // 2
corners = {47: (gw - 70 - 64, gh - 70 - 64), 46: (70, gh - 70 - 64), 44: (gw - 70 - 64, 70), 40: (70, 70), 33: (gw - 150 - 64, gh // 2), 20: (gw // 2, gh - 150 - 64)}
keep = np.zeros(seeds.shape[0], bool)
for v, (wx, wy) in corners.items():
    keep |= (ref == v) & (cx >= wx) & (cx < wx + 64) & (cy >= wy) & (cy < wy + 64)
win = np.ascontiguousarray(seeds[keep])
print("seeds", win.shape)
kw = dict(level=0, csize=2, wsize=7, minImageNum=3, enable_check=1, seed=31)
o = ob.Oracle(48, wide=True, list_cap=64, schedule=ob.SCHEDULE_ENGINE, sum_mode=ob.SUM_TREE64, nthreads=16, **kw)
e = engine.Engine(48, max_patches=4_000_000, **kw)
o.set_scene(sc)
e.set_scene(sc)
o.add_patches(win)
e.upload_patches(win)
for p in range(2):
    co, ce = o.engine_pass(0, p), e.engine_pass(0, p)
    print("pass", p, {k: (co[k], ce[k]) for k in co if co[k] != ce[k]} or "counters equal")
    no, pv = o.export_new()
    ko = o.export_kills()
    o.commit(no, ko)
    e.commit_local()
    po, pe = o.patches(), e.patches()
    print(" pool", po.shape, pe.shape)
    if po.shape == pe.shape:
        for f in ("nimages", "images", "nvimages", "vimages", "coord", "normal", "ncc", "dscale", "tmp"):
            a, b = po[f], pe[f]
            d = (a != b)
            if d.ndim > 1:
                d = d.any(axis=1)
            print("  field", f, "differs in", int(d.sum()))
            if d.sum() and f in ("images", "nimages", "ncc"):
                i = int(np.nonzero(d)[0][0])
                print("   first", i, "oracle", po[i], "\n   engine", pe[i])
