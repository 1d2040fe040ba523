"""Reconstruction quality on a synthetic scene: every patch against the analytic surface at the pixel of its reference
view it projects to (depth along the optical axis, normal)."""
import sys
import numpy as np
sys.path.insert(0, '/root/repo')
from mvskit_amd import engine, synth


def patch_errors(sc, p):
    ref = p["images"][:, 0].astype(int)
    X = p["coord"].astype(np.float64)
    P = sc.P.astype(np.float64)[ref]
    x = np.einsum("nij,nj->ni", P, X)
    u, v = x[:, 0] / x[:, 2], x[:, 1] / x[:, 2]
    iu, iv = np.clip(np.rint(u).astype(int), 0, sc.W - 1), np.clip(np.rint(v).astype(int), 0, sc.H - 1)
    true_pt = sc.points[ref, iv, iu].astype(np.float64)
    true_n = sc.normals[ref, iv, iu].astype(np.float64)
    oax = P[:, 2, :] / np.linalg.norm(P[:, 2, :3], axis=1, keepdims=True)
    d_true = np.einsum("ni,ni->n", true_pt, oax[:, :3]) + oax[:, 3]
    d = np.einsum("ni,ni->n", X, oax)
    ok = np.isfinite(d_true)
    rel = np.abs(d[ok] - d_true[ok]) / d_true[ok]
    cosang = np.abs(np.einsum("ni,ni->n", p["normal"][ok][:, :3].astype(np.float64), true_n[ok]))
    return rel, np.degrees(np.arccos(np.clip(cosang, -1, 1)))


if __name__ == "__main__":
    sc = synth.make_scene(nviews=5, W=384, H=216, arc_deg=60.0, radius=4.0, kind="multi")
    seeds = synth.make_seeds(sc, stride=3, seed=19)
    rel, ang = patch_errors(sc, seeds)
    print(f"seeds: {seeds.shape[0]}, depth rel err median {np.median(rel):.2e} p90 {np.percentile(rel, 90):.2e}, normal angle median {np.median(ang):.2f} p90 {np.percentile(ang, 90):.2f} deg")
    e = engine.Engine(sc.nviews, level=0, csize=2, wsize=7, minImageNum=3, enable_check=1, seed=3)
    e.set_scene(sc)
    e.upload_patches(seeds)
    for it in range(4):
        e.propagate(it)
        e.filter()
        e.update_threshold()
        p = e.patches()
        p = p[p["dscale"] > 0]
        rel, ang = patch_errors(sc, p)
        print(f"iter {it}: {p.shape[0]} patches, depth rel err median {np.median(rel):.2e} p90 {np.percentile(rel, 90):.2e} p99 {np.percentile(rel, 99):.2e}, "
              f"normal angle median {np.median(ang):.2f} p90 {np.percentile(ang, 90):.2f} deg, ncc median {np.median(p['ncc']):.3f}")
