"""What would a pool ordered by (reference view, cell) buy?  Times one Propagate::run and one Filter::run on the pool as the
engine leaves it (creation order) and on the same patches re-uploaded in cell order."""
import sys, time
import numpy as np
sys.path.insert(0, '/root/repo')
from mvskit_amd import engine, synth

sc = synth.make_scene(nviews=12, W=1920, H=1080, arc_deg=110.0, radius=4.0, kind="multi")
seeds = synth.make_seeds(sc, level=0, csize=2, stride=2, seed=777)
e = engine.Engine(12, level=0, csize=2, wsize=7, minImageNum=3, enable_check=1, seed=1)
e.set_scene(sc); e.upload_patches(seeds)
for it in range(2):
    e.propagate(it); e.update_threshold()
p = e.patches()
ncc_t, nb, d = e.thresholds()
def timed(tag):
    t = time.perf_counter(); c = e.propagate(2); tp = time.perf_counter() - t
    tm = e.timing()
    t = time.perf_counter(); r = e.filter(); tf = time.perf_counter() - t
    print(f"{tag}: propagate {tp*1e3:.0f} ms (sweep {tm['sweep_ms']:.0f}, index {tm['index_ms']:.0f}), filter {tf*1e3:.0f} ms, patches {c['patches']}", flush=True)
timed("creation order")
# same patches, cell order
ref = p["images"][:, 0].astype(int)
X = p["coord"].astype(np.float64)
x = np.einsum("nij,nj->ni", sc.P.astype(np.float64)[ref], X)
cx = np.floor(x[:, 0] / x[:, 2] + 0.5).astype(np.int64) // 2
cy = np.floor(x[:, 1] / x[:, 2] + 0.5).astype(np.int64) // 2
order = np.lexsort((cx, cy, ref))
q = p[order].copy()
e.clear_patches()
e.set_thresholds(ncc_t, nb, d)
e.upload_patches(q)
# upload resets vimages / tmp; run one filter-free iteration state as comparable as possible
timed("cell order    ")
