import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from mvskit_amd import engine, synth
n = 48
sc = synth.make_scene(nviews=n, W=480, H=270, arc_deg=141.0, radius=4.0, kind="multi")
seeds = synth.make_seeds(sc, stride=2, seed=31)
print('seeds', seeds.shape, flush=True)
e = engine.Engine(n, level=0, csize=2, wsize=7, minImageNum=3, enable_check=1, seed=6)
e.set_scene(sc); e.upload_patches(seeds)
for it in range(4):
    try:
        c = e.propagate(it)
        print(it, c['patches'], c['inserted'], e.timing()['sweep_ms'], flush=True)
    except Exception as ex:
        print('ERR', it, ex, flush=True); break
    e.update_threshold()
p = e.patches(); print('pool', p.shape, 'mean nimg', p['nimages'].mean(), 'max', p['nimages'].max())
