import sys, time, numpy as np
sys.path.insert(0,'/root/repo')
from mvskit_amd import engine, synth
sc = synth.make_scene(nviews=12, W=1920, H=1080, arc_deg=110.0, radius=4.0, kind="multi")
seeds = synth.make_seeds(sc, stride=4, seed=777)
print("seeds", seeds.shape[0], flush=True)
e = engine.Engine(12, level=0, csize=2, wsize=7, minImageNum=3, enable_check=0, seed=1)
e.set_scene(sc)
pre, _, flag = e.probe(engine.PROBE_PREPROCESS, seeds)
ok = pre[flag == 0]
print("pre ok", ok.shape[0], "mean nimages", ok['nimages'].mean(), flush=True)
ok = np.concatenate([ok]*3)[:1000000]
for op,name in ((engine.PROBE_NCC,"ncc(1 eval)"),(engine.PROBE_COST,"cost(1 eval)"),(engine.PROBE_REFINE,"refine(26 evals)"),(engine.PROBE_PREPROCESS,"pre"),(engine.PROBE_POSTPROCESS,"post")):
    e.probe(op, ok[:1000])
    t=time.perf_counter(); e.probe(op, ok); dt=time.perf_counter()-t
    print(f"{name}: {ok.shape[0]/dt/1e6:.2f} M/s  ({dt*1e3:.1f} ms for {ok.shape[0]})", flush=True)
