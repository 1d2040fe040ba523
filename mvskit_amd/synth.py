"""Synthetic multi-view scenes for the PatchMatch-MVS hot path.

The reference ships no data (no image, camera, option or .patch file is checked in), so the
parity tests and bench.py generate their inputs here: analytic pinhole cameras written as the
3x4 `CONTOUR` projection matrices Camera::setProjection reads (image/camera.cpp:110-116), a solid
(3D) band-limited texture sampled on planes and a sphere so that every view sees the same surface
colour, and seed patches in the record layout of pmmvps/patch.hpp:33-66.

Pure numpy; when torch is importable the per-pixel texture synthesis runs through torch (on the
GPU if there is one).  The bytes produced are handed unchanged to the oracle and to the engine, so
the backend that made them does not matter for parity.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import numpy as np

MAX_IMAGES = 32

def patch_dtype(max_images: int = MAX_IMAGES) -> np.dtype:
    """numpy mirror of `mvs_patch` (include/mvskit_engine.h) with MVS_MAX_IMAGES = max_images: 128 bytes at 32, 192 bytes at 64
    (libmvskit_engine_cap64.so)."""
    return np.dtype(
        [
            ("coord", "<f4", (4,)),
            ("normal", "<f4", (4,)),
            ("ncc", "<f4"),
            ("dscale", "<f4"),
            ("ascale", "<f4"),
            ("tmp", "<f4"),
            ("nimages", "<i4"),
            ("nvimages", "<i4"),
            ("flags", "<i4"),
            ("id", "<i4"),
            ("images", "u1", (max_images,)),
            ("vimages", "u1", (max_images,)),
        ],
        align=False,
    )


def convert_records(recs, dtype: np.dtype) -> np.ndarray:
    """Patch records in another record width (32 <-> 64 list slots): scalar fields copied, lists copied as far as they fit."""
    recs = np.asarray(recs)
    if recs.dtype == dtype:
        return np.ascontiguousarray(recs)
    out = np.zeros(recs.shape[0], dtype=dtype)
    for name in ("coord", "normal", "ncc", "dscale", "ascale", "tmp", "nimages", "nvimages", "flags", "id"):
        out[name] = recs[name]
    for name in ("images", "vimages"):
        k = min(out[name].shape[1], recs[name].shape[1])
        out[name][:, :k] = recs[name][:, :k]
    cap = out["images"].shape[1]
    out["nimages"] = np.minimum(out["nimages"], cap)
    out["nvimages"] = np.minimum(out["nvimages"], cap)
    return out


#: the default record: 32 list slots, 128 bytes
PATCH_DTYPE = patch_dtype(MAX_IMAGES)
assert PATCH_DTYPE.itemsize == 128


@dataclass
class Scene:
    W: int
    H: int
    P: np.ndarray  # [n,3,4] float32 level-0 projections
    images: np.ndarray  # [n,H,W,3] uint8
    centers: np.ndarray  # [n,3] float64
    points: np.ndarray | None = None  # [n,H,W,3] float32 ground-truth surface point per pixel (NaN = background)
    normals: np.ndarray | None = None  # [n,H,W,3] float32
    meta: dict = field(default_factory=dict)

    @property
    def nviews(self) -> int:
        return int(self.P.shape[0])


def look_at(center, target, up=(0.0, 1.0, 0.0)):
    """World->camera rotation R and translation t (x right, y down, z forward)."""
    c = np.asarray(center, dtype=np.float64)
    z = np.asarray(target, dtype=np.float64) - c
    z /= np.linalg.norm(z)
    x = np.cross(z, np.asarray(up, dtype=np.float64))
    x /= np.linalg.norm(x)
    y = np.cross(z, x)
    R = np.stack([x, y, z])
    t = -R @ c
    return R, t


def make_cameras(n, W, H, arc_deg, radius, focal_px=None, elev=0.15, target=(0.0, 0.0, 0.0)):
    """n cameras on an arc in the x-z plane, all looking at `target`.  Returns (P[n,3,4] f32, C[n,3])."""
    if focal_px is None:
        focal_px = 765.702941895 * W / 640.0  # focal of the sample camera in test/test.cpp:96-100
    K = np.array([[focal_px, 0.0, W / 2.0], [0.0, focal_px, H / 2.0], [0.0, 0.0, 1.0]])
    P = np.zeros((n, 3, 4), dtype=np.float64)
    C = np.zeros((n, 3), dtype=np.float64)
    for i in range(n):
        a = 0.0 if n == 1 else math.radians(-arc_deg / 2.0 + arc_deg * i / (n - 1))
        c = np.array([radius * math.sin(a), radius * elev, radius * math.cos(a)]) + np.asarray(target)
        R, t = look_at(c, target)
        P[i] = K @ np.concatenate([R, t[:, None]], axis=1)
        C[i] = c
    return P.astype(np.float32), C


# --------------------------------------------------------------------------- geometry
@dataclass
class Plane:
    n: tuple  # unit normal
    d: float  # n.X = d
    bounds: tuple = (-1e9, 1e9, -1e9, 1e9)  # xmin, xmax, ymin, ymax of the hit point


@dataclass
class Sphere:
    c: tuple
    r: float


def default_objects(kind="multi"):
    if kind == "plane":
        return [Plane((0.0, 0.0, 1.0), 0.0)]
    s30, c30 = math.sin(math.radians(25.0)), math.cos(math.radians(25.0))
    return [
        Plane((0.0, 0.0, 1.0), 0.0),  # back wall z = 0
        Plane((s30, 0.0, c30), -1.1 * s30 + 0.25 * c30, bounds=(-3.0, -0.2, -1e9, 1e9)),  # left ramp
        Plane((-s30, 0.0, c30), -1.1 * s30 + 0.25 * c30, bounds=(0.2, 3.0, -1e9, 1e9)),  # right ramp
        Sphere((0.0, 0.0, 0.55), 0.5),
    ]


def _texture_basis(px_size, nfreq=64, seed=12345):
    """64 random 3D sinusoids, wavelengths 3..64 pixels (at `px_size` world units per pixel),
    amplitude ~ 1/f, per-channel mixing weights.  mt19937(12345) as in SURVEY.md section 8d."""
    rng = np.random.RandomState(seed)
    lam = np.exp(rng.uniform(math.log(3.0), math.log(64.0), nfreq)) * px_size
    dirs = rng.normal(size=(nfreq, 3))
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    k = (2.0 * math.pi / lam)[:, None] * dirs  # [nfreq,3]
    phase = rng.uniform(0.0, 2.0 * math.pi, nfreq)
    amp = lam / lam.max()
    mix = rng.uniform(0.3, 1.0, size=(nfreq, 3)) * amp[:, None]
    mix *= 95.0 / np.sqrt((mix ** 2).sum(axis=0) / 2.0) / 2.2  # std ~ 43 grey levels per channel
    return k, phase, mix


def _shade(points, basis, chunk=1 << 18):
    """uint8 RGB colour of surface points [N,3] (NaN rows -> mid grey)."""
    k, phase, mix = basis
    N = points.shape[0]
    out = np.empty((N, 3), dtype=np.uint8)
    try:
        import torch

        dev = "cuda" if torch.cuda.is_available() else "cpu"
        kt = torch.as_tensor(k, dtype=torch.float64, device=dev)
        pt = torch.as_tensor(phase, dtype=torch.float64, device=dev)
        mt = torch.as_tensor(mix, dtype=torch.float32, device=dev)
        if dev == "cuda":  # the points go up once (float32, as the ray caster returned them), the colours come back once
            chunk = 1 << 21
            pall = torch.nan_to_num(torch.as_tensor(points, dtype=torch.float32, device=dev))
            oall = torch.empty((N, 3), dtype=torch.uint8, device=dev)
            for b in range(0, N, chunk):
                x = pall[b : b + chunk].to(torch.float64)
                ph = torch.remainder(x @ kt.T + pt, 2.0 * math.pi).to(torch.float32)
                oall[b : b + chunk] = (128.0 + torch.sin(ph) @ mt).round().clamp(0, 255).to(torch.uint8)
            out[:] = oall.cpu().numpy()
            bad = np.isnan(points).any(axis=1)
            out[bad] = 128
            return out
        for b in range(0, N, chunk):
            x = torch.as_tensor(np.nan_to_num(points[b : b + chunk]), dtype=torch.float64, device=dev)
            ph = torch.remainder(x @ kt.T + pt, 2.0 * math.pi).to(torch.float32)  # phase in f64, sin in f32
            v = 128.0 + torch.sin(ph) @ mt
            out[b : b + chunk] = v.round().clamp(0, 255).to(torch.uint8).cpu().numpy()
    except ImportError:  # pragma: no cover
        for b in range(0, N, chunk):
            x = np.nan_to_num(points[b : b + chunk]).astype(np.float64)
            v = 128.0 + np.sin(x @ k.T + phase) @ mix
            out[b : b + chunk] = np.clip(np.rint(v), 0, 255).astype(np.uint8)
    bad = np.isnan(points).any(axis=1)
    out[bad] = 128
    return out


def _raycast_torch(P, C, W, H, objects, dev):
    """_raycast on the GPU (float64, the same formulas): a 4K view takes milliseconds instead of seconds."""
    import torch

    f64 = dict(dtype=torch.float64, device=dev)
    Minv = torch.as_tensor(np.linalg.inv(P[:, :3].astype(np.float64)), **f64)
    v, u = torch.meshgrid(torch.arange(H, **f64), torch.arange(W, **f64), indexing="ij")
    d = torch.stack([u, v, torch.ones_like(u)], dim=-1) @ Minv.T
    o = torch.as_tensor(np.asarray(C, dtype=np.float64), **f64)
    best = torch.full((H, W), float("inf"), **f64)
    pts = torch.full((H, W, 3), float("nan"), **f64)
    nrm = torch.full((H, W, 3), float("nan"), **f64)
    for ob in objects:
        if isinstance(ob, Plane):
            n = torch.as_tensor(np.asarray(ob.n, dtype=np.float64), **f64)
            denom = d @ n
            t = (ob.d - o @ n) / denom
            X = o + t[..., None] * d
            ok = (t > 1e-6) & (t < best) & (denom < 0)
            xmin, xmax, ymin, ymax = ob.bounds
            ok &= (X[..., 0] >= xmin) & (X[..., 0] <= xmax) & (X[..., 1] >= ymin) & (X[..., 1] <= ymax)
            best = torch.where(ok, t, best)
            pts = torch.where(ok[..., None], X, pts)
            nrm = torch.where(ok[..., None], n.expand_as(nrm), nrm)
        else:
            c = torch.as_tensor(np.asarray(ob.c, dtype=np.float64), **f64)
            oc = o - c
            a = (d * d).sum(-1)
            b = 2.0 * (d @ oc)
            cc = oc @ oc - ob.r ** 2
            disc = b * b - 4 * a * cc
            t = (-b - torch.sqrt(disc)) / (2 * a)
            ok = (disc > 0) & (t > 1e-6) & (t < best)
            X = o + t[..., None] * d
            best = torch.where(ok, t, best)
            pts = torch.where(ok[..., None], X, pts)
            nrm = torch.where(ok[..., None], (X - c) / ob.r, nrm)
    return pts.to(torch.float32).cpu().numpy(), nrm.to(torch.float32).cpu().numpy()


def _raycast(P, C, W, H, objects):
    """Nearest hit of every pixel-centre ray.  Returns points [H,W,3], normals [H,W,3] (NaN = miss)."""
    try:
        import torch

        if torch.cuda.is_available():
            return _raycast_torch(P, C, W, H, objects, "cuda")
    except ImportError:  # pragma: no cover
        pass
    M = P[:, :3].astype(np.float64)
    Minv = np.linalg.inv(M)
    u, v = np.meshgrid(np.arange(W, dtype=np.float64), np.arange(H, dtype=np.float64))
    d = np.stack([u, v, np.ones_like(u)], axis=-1) @ Minv.T  # ray directions (z_cam = 1)
    o = np.asarray(C, dtype=np.float64)
    best = np.full((H, W), np.inf)
    pts = np.full((H, W, 3), np.nan)
    nrm = np.full((H, W, 3), np.nan)
    for ob in objects:
        if isinstance(ob, Plane):
            n = np.asarray(ob.n, dtype=np.float64)
            denom = d @ n
            with np.errstate(divide="ignore", invalid="ignore"):
                t = (ob.d - o @ n) / denom
            X = o + t[..., None] * d
            ok = (t > 1e-6) & (t < best) & (denom < 0)
            xmin, xmax, ymin, ymax = ob.bounds
            ok &= (X[..., 0] >= xmin) & (X[..., 0] <= xmax) & (X[..., 1] >= ymin) & (X[..., 1] <= ymax)
            best = np.where(ok, t, best)
            pts[ok] = X[ok]
            nrm[ok] = n
        else:
            c = np.asarray(ob.c, dtype=np.float64)
            oc = o - c
            a = (d * d).sum(-1)
            b = 2.0 * (d @ oc)
            cc = oc @ oc - ob.r ** 2
            disc = b * b - 4 * a * cc
            with np.errstate(invalid="ignore"):
                t = (-b - np.sqrt(disc)) / (2 * a)
            ok = (disc > 0) & (t > 1e-6) & (t < best)
            X = o + t[..., None] * d
            best = np.where(ok, t, best)
            pts[ok] = X[ok]
            nn = (X - c) / ob.r
            nrm[ok] = nn[ok]
    return pts.astype(np.float32), nrm.astype(np.float32)


def make_scene(nviews=12, W=1920, H=1080, arc_deg=110.0, radius=4.0, kind="multi", keep_geometry=True, tex_seed=12345,
               noise_sigma=2.0, geometry_views=None):
    """cfg2-style scene: `nviews` cameras on an arc looking at 3 textured planes + a sphere
    (kind="multi") or one textured plane z = 0 (kind="plane").  geometry_views: keep the ground-truth points / normals
    (needed by make_seeds for reference views and occlusion tests) only for these views -- the others hold NaN, which
    make_seeds reads as "not visible there" (48 views of 4K would need 19 GB otherwise)."""
    P, C = make_cameras(nviews, W, H, arc_deg, radius)
    focal = 765.702941895 * W / 640.0
    basis = _texture_basis(radius / focal, seed=tex_seed)
    objects = default_objects(kind)
    imgs = np.empty((nviews, H, W, 3), dtype=np.uint8)
    pts_all = np.empty((nviews, H, W, 3), dtype=np.float32) if keep_geometry else None
    nrm_all = np.empty((nviews, H, W, 3), dtype=np.float32) if keep_geometry else None
    # independent sensor noise per view, mt19937(tex_seed + 1000 + view): drawn on host threads (numpy releases the GIL) while the next
    # views are ray-cast and shaded -- the same images as drawing it in line, a 48 x 4K scene in a third of the time
    import concurrent.futures
    import os

    def add_noise(i):
        nz = np.random.RandomState(tex_seed + 1000 + i).normal(0.0, noise_sigma, size=imgs[i].shape)
        imgs[i] = np.clip(np.rint(imgs[i].astype(np.float64) + nz), 0, 255).astype(np.uint8)

    pool = concurrent.futures.ThreadPoolExecutor(max_workers=max(1, min(8, (os.cpu_count() or 2) - 1))) if noise_sigma > 0 else None
    jobs = []
    for i in range(nviews):
        pts, nrm = _raycast(P[i].astype(np.float64), C[i], W, H, objects)
        imgs[i] = _shade(pts.reshape(-1, 3), basis).reshape(H, W, 3)
        if pool is not None:
            jobs.append(pool.submit(add_noise, i))
        if keep_geometry:
            if geometry_views is None or i in geometry_views:
                pts_all[i] = pts
                nrm_all[i] = nrm
            else:
                pts_all[i] = np.nan
                nrm_all[i] = np.nan
    for j in jobs:
        j.result()
    if pool is not None:
        pool.shutdown()
    return Scene(W=W, H=H, P=P, images=imgs, centers=C, points=pts_all, normals=nrm_all,
                 meta={"kind": kind, "arc_deg": arc_deg, "radius": radius, "focal": focal})


def _project(P, X):
    x = X @ P[:, :3].T + P[:, 3]
    return x[..., :2] / x[..., 2:3], x[..., 2]


def make_seeds(scene: Scene, level=0, csize=2, stride=16, depth_noise=0.5, normal_noise_deg=5.0, seed=777,
               views=None, max_cone_deg=60.0):
    """Seed patches: for every view v (the reference view) one patch per `stride` x `stride` cells at
    the cell-centre pixel, on the true surface, with depth noise N(0,(depth_noise*unit)^2) along the
    viewing ray and `normal_noise_deg` of normal noise (RNG mt19937(seed + v)).  m_images = the
    reference view followed by every other view that sees the point inside its image within
    `max_cone_deg` of the (true) normal and unoccluded.  m_ncc = -1 (PatchManager::sortPatches
    computes it on first use, patch_manager.cpp:411-415)."""
    assert scene.points is not None, "make_scene(keep_geometry=True) needed"
    n = scene.nviews
    scale = 1 << level
    Wl, Hl = scene.W // scale, scene.H // scale
    gw, gh = (Wl + csize - 1) // csize, (Hl + csize - 1) // csize
    out = []
    cosmax = math.cos(math.radians(max_cone_deg))
    Pd = scene.P.astype(np.float64)
    for v in (range(n) if views is None else views):
        rng = np.random.RandomState(seed + v)
        cx, cy = np.meshgrid(np.arange(stride // 2, gw, stride), np.arange(stride // 2, gh, stride))
        cx, cy = cx.ravel(), cy.ravel()
        # cell-centre pixel at `level` (propagate.cpp:147-148), mapped to level-0 pixel for the GT lookup
        px = (csize * (2 * cx + 1) - 1) / 2.0
        py = (csize * (2 * cy + 1) - 1) / 2.0
        u0 = np.clip(np.rint(px * scale).astype(int), 0, scene.W - 1)
        v0 = np.clip(np.rint(py * scale).astype(int), 0, scene.H - 1)
        X = scene.points[v, v0, u0].astype(np.float64)
        N = scene.normals[v, v0, u0].astype(np.float64)
        ok = ~np.isnan(X).any(axis=1)
        X, N = X[ok], N[ok]
        if X.shape[0] == 0:
            continue
        C = scene.centers[v]
        ray = X - C
        dist = np.linalg.norm(ray, axis=1, keepdims=True)
        ray /= dist
        unit = 2.0 * dist * scale / (2.0 * scene.meta["focal"])  # Optim::getUnit, optim.cpp:34-41
        X = X + ray * unit * depth_noise * rng.normal(size=(X.shape[0], 1))
        # normal noise: rotate towards a random tangent direction
        tang = rng.normal(size=N.shape)
        tang -= (tang * N).sum(1, keepdims=True) * N
        tang /= np.linalg.norm(tang, axis=1, keepdims=True)
        ang = np.radians(normal_noise_deg) * rng.normal(size=(N.shape[0], 1))
        Nn = N * np.cos(ang) + tang * np.sin(ang)
        rec = np.zeros(X.shape[0], dtype=PATCH_DTYPE)
        rec["coord"][:, :3] = X
        rec["coord"][:, 3] = 1.0
        rec["normal"][:, :3] = Nn
        rec["ncc"] = -1.0
        rec["flags"] = 1
        rec["images"][:, 0] = v
        cnt = np.ones(X.shape[0], dtype=np.int32)
        order = sorted((u for u in range(n) if u != v), key=lambda u: np.linalg.norm(scene.centers[u] - C))
        for u in order:
            uv, z = _project(Pd[u], X)
            inside = (z > 0) & (uv[:, 0] >= 8) & (uv[:, 0] < scene.W - 8) & (uv[:, 1] >= 8) & (uv[:, 1] < scene.H - 8)
            r = scene.centers[u] - X
            r /= np.linalg.norm(r, axis=1, keepdims=True)
            inside &= (r * N).sum(1) >= cosmax
            ui = np.clip(np.rint(uv[:, 0]).astype(int), 0, scene.W - 1)
            vi = np.clip(np.rint(uv[:, 1]).astype(int), 0, scene.H - 1)
            G = scene.points[u, vi, ui].astype(np.float64)
            with np.errstate(invalid="ignore"):
                occl = np.linalg.norm(G - X, axis=1) > 4.0 * unit[:, 0]
            inside &= ~(occl | np.isnan(G).any(axis=1))
            sel = inside & (cnt < MAX_IMAGES)
            rec["images"][sel, cnt[sel]] = u
            cnt[sel] += 1
        rec["nimages"] = cnt
        out.append(rec)
    if not out:
        return np.zeros(0, dtype=PATCH_DTYPE)
    res = np.concatenate(out)
    res["id"] = np.arange(res.shape[0], dtype=np.int32)
    return res
