"""ctypes binding of the CPU oracle (oracle/pmmvs_oracle.h).  TEST INFRASTRUCTURE ONLY: imported by
tests/, by __graft_entry__.smoke() and by the cpu_baseline leg of bench.py, never by mvskit_amd."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "_build", "liboracle.so")
WIDE_LIB_PATH = os.path.join(ORACLE_DIR, "_build", "liboracle_wide.so")  # 64 views per list, 192-byte records (make -C oracle wide)

MAX_IMAGES = 32
PATCH_DTYPE = np.dtype(
    [("coord", "<f4", (4,)), ("normal", "<f4", (4,)), ("ncc", "<f4"), ("dscale", "<f4"), ("ascale", "<f4"),
     ("tmp", "<f4"), ("nimages", "<i4"), ("nvimages", "<i4"), ("flags", "<i4"), ("id", "<i4"),
     ("images", "u1", (MAX_IMAGES,)), ("vimages", "u1", (MAX_IMAGES,))])
assert PATCH_DTYPE.itemsize == 128
WIDE_PATCH_DTYPE = np.dtype(
    [("coord", "<f4", (4,)), ("normal", "<f4", (4,)), ("ncc", "<f4"), ("dscale", "<f4"), ("ascale", "<f4"),
     ("tmp", "<f4"), ("nimages", "<i4"), ("nvimages", "<i4"), ("flags", "<i4"), ("id", "<i4"),
     ("images", "u1", (64,)), ("vimages", "u1", (64,))])
assert WIDE_PATCH_DTYPE.itemsize == 192


def convert_records(recs, dtype):
    """records in the other list width: scalar fields copied, lists as far as they fit"""
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

SCHEDULE_FAITHFUL, SCHEDULE_ENGINE = 0, 1
SUM_SEQ, SUM_TREE64 = 0, 1


class Config(C.Structure):
    _fields_ = [("nviews", C.c_int32), ("level", C.c_int32), ("csize", C.c_int32), ("wsize", C.c_int32),
                ("minImageNum", C.c_int32), ("max_propag", C.c_int32), ("nccThreshold", C.c_float),
                ("maxAngleThreshold", C.c_float), ("quadThreshold", C.c_float), ("depth", C.c_int32),
                ("seed", C.c_uint32), ("schedule", C.c_int32), ("sum_mode", C.c_int32), ("refine_steps", C.c_int32),
                ("refine_rd0", C.c_float), ("refine_ra0", C.c_float), ("enable_check", C.c_int32),
                ("view_begin", C.c_int32), ("view_stride", C.c_int32), ("nthreads", C.c_int32),
                ("view_propagation", C.c_int32), ("shard_index", C.c_int32), ("shard_count", C.c_int32),
                ("list_cap", C.c_int32), ("literal_evals", C.c_int32), ("literal_groups", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("candidates", "prefiltered", "patches", "fail0", "fail1", "inserted",
                                         "replaced", "evals", "view_evals", "trimmed")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


def build(force=False):
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < os.path.getmtime(
            os.path.join(ORACLE_DIR, "pmmvs_oracle.cpp")):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB_PATH


_libs = {}


def lib(wide=False):
    """wide: the build with 64 views per list and 192-byte records (-DORC_WIDE_LISTS) -- what the 64-view engine is compared with"""
    path = WIDE_LIB_PATH if wide else os.environ.get("MVS_ORACLE_LIB", LIB_PATH)  # `make -C oracle asan-test` points this at the sanitizer build
    if path in _libs:
        return _libs[path]
    if path in (LIB_PATH, WIDE_LIB_PATH) and not os.path.exists(path):
        build(force=True)
    L = C.CDLL(path)
    vp, f32p, u8p, i32p = C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint8), C.POINTER(C.c_int32)
    L.orc_default_config.argtypes = [C.POINTER(Config)]
    L.orc_create.argtypes = [C.POINTER(Config)]
    L.orc_create.restype = vp
    L.orc_destroy.argtypes = [vp]
    L.orc_last_error.restype = C.c_char_p
    L.orc_set_view.argtypes = [vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]
    L.orc_finalize_views.argtypes = [vp]
    L.orc_get_pyramid.argtypes = [vp, C.c_int, C.c_int, vp, i32p, i32p]
    L.orc_get_camera.argtypes = [vp, C.c_int] + [vp] * 6
    L.orc_grid_dims.argtypes = [vp, C.c_int, i32p, i32p]
    L.orc_set_thresholds.argtypes = [vp, C.c_float, C.c_float, C.c_int]
    L.orc_get_thresholds.argtypes = [vp, f32p, f32p, i32p]
    L.orc_update_threshold.argtypes = [vp]
    L.orc_add_patches.argtypes = [vp, C.c_int, vp]
    L.orc_num_patches.argtypes = [vp]
    L.orc_get_patches.argtypes = [vp, C.c_int, vp]
    L.orc_clear_patches.argtypes = [vp]
    L.orc_propagate.argtypes = [vp, C.c_int, C.POINTER(Counters)]
    L.orc_filter.argtypes = [vp, vp]
    L.orc_list_truncations.argtypes = [vp]
    L.orc_list_truncations.restype = C.c_int64
    L.orc_list_storage.restype = C.c_int
    L.orc_set_cell_budget.argtypes = [vp, C.c_int64]
    L.orc_set_time_budget.argtypes = [vp, C.c_double]
    L.orc_engine_pass.argtypes = [vp, C.c_int, C.c_int, C.POINTER(Counters)]
    L.orc_export_new.argtypes = [vp, C.c_int, vp, vp]
    L.orc_export_kills.argtypes = [vp, C.c_int, vp]
    L.orc_commit.argtypes = [vp, C.c_int, vp, C.c_int, vp]
    L.orc_depth_normal_map.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    L.orc_project.argtypes = [vp, C.c_int, vp, C.c_int, vp]
    L.orc_unproject.argtypes = [vp, C.c_int, vp, C.c_int, vp]
    L.orc_get_unit.argtypes = [vp, C.c_int, vp]
    L.orc_get_unit.restype = C.c_float
    L.orc_get_paxes.argtypes = [vp, C.c_int, vp, vp, vp, vp]
    L.orc_get_color.argtypes = [vp, C.c_int, C.c_float, C.c_float, C.c_int, vp]
    L.orc_get_tex.argtypes = [vp, vp, vp, vp, vp, C.c_int, vp, C.c_int]
    L.orc_compute_incc.argtypes = [vp, vp, C.c_int]
    L.orc_compute_incc.restype = C.c_float
    L.orc_compute_ncc.argtypes = [vp, vp]
    L.orc_compute_ncc.restype = C.c_float
    L.orc_set_inccs.argtypes = [vp, vp, C.c_int, vp]
    L.orc_set_inccs_matrix.argtypes = [vp, vp, C.c_int, vp]
    L.orc_preprocess.argtypes = [vp, vp]
    L.orc_refine.argtypes = [vp, vp, vp]
    L.orc_postprocess.argtypes = [vp, vp]
    L.orc_cost.argtypes = [vp, vp, vp]
    L.orc_cost.restype = C.c_double
    L.orc_encode.argtypes = [vp, vp, vp]
    L.orc_decode.argtypes = [vp, vp, vp, vp, vp]
    L.orc_generate_patch.argtypes = [vp, vp, vp, vp]
    L.orc_quad_residual.argtypes = [vp, vp, vp, C.c_int]
    L.orc_quad_residual.restype = C.c_float
    L.orc_robustincc.argtypes = [C.c_float]
    L.orc_robustincc.restype = C.c_float
    L.orc_unrobustincc.argtypes = [C.c_float]
    L.orc_unrobustincc.restype = C.c_float
    L.orc_minstd_draws.argtypes = [C.c_int, vp]
    L.orc_rng_uniform.argtypes = [C.c_uint32] * 6
    L.orc_rng_uniform.restype = C.c_float
    for fn in ("orc_sinf", "orc_cosf", "orc_asinf", "orc_acosf", "orc_atanf"):
        getattr(L, fn).argtypes = [C.c_float]
        getattr(L, fn).restype = C.c_float
    L.orc_is_neighbor.argtypes = [vp, vp, vp, C.c_float]
    L.orc_compute_gain.argtypes = [vp, vp]
    L.orc_compute_gain.restype = C.c_float
    L.orc_check.argtypes = [vp, vp]
    L.orc_patch_bytes.restype = C.c_int
    _libs[path] = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def f4(*v):
    return np.asarray(v, dtype=np.float32)


class Oracle:
    """One PmMvps-like scene held by the oracle."""

    def __init__(self, nviews, wide=False, **kw):
        L = lib(wide)
        self.dtype = WIDE_PATCH_DTYPE if L.orc_patch_bytes() == 192 else PATCH_DTYPE
        self.cfg = Config()
        L.orc_default_config(C.byref(self.cfg))
        self.cfg.nviews = nviews
        for k, v in kw.items():
            if not hasattr(self.cfg, k):
                raise AttributeError(k)
            setattr(self.cfg, k, v)
        self.h = L.orc_create(C.byref(self.cfg))
        if not self.h:
            raise RuntimeError(L.orc_last_error().decode())
        self.L = L
        self._keep = []

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- scene
    def set_scene(self, scene, masks=None):
        for v in range(scene.nviews):
            img = np.ascontiguousarray(scene.images[v])
            P = np.ascontiguousarray(scene.P[v], dtype=np.float32)
            m = None if masks is None else np.ascontiguousarray(masks[v], dtype=np.uint8)
            if self.L.orc_set_view(self.h, v, scene.W, scene.H, _ptr(P), _ptr(img), _ptr(m)) != 0:
                raise RuntimeError(self.L.orc_last_error().decode())
        if self.L.orc_finalize_views(self.h) != 0:
            raise RuntimeError(self.L.orc_last_error().decode())

    def pyramid(self, v, level):
        w, h = C.c_int32(), C.c_int32()
        self.L.orc_get_pyramid(self.h, v, level, None, C.byref(w), C.byref(h))
        out = np.empty((h.value, w.value, 3), dtype=np.uint8)
        self.L.orc_get_pyramid(self.h, v, level, _ptr(out), C.byref(w), C.byref(h))
        return out

    def camera(self, v):
        c, o = np.zeros(4, np.float32), np.zeros(4, np.float32)
        x, y, z = np.zeros(3, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32)
        ip = np.zeros(1, np.float32)
        self.L.orc_get_camera(self.h, v, _ptr(c), _ptr(o), _ptr(x), _ptr(y), _ptr(z), _ptr(ip))
        return dict(center=c, oaxis=o, xaxis=x, yaxis=y, zaxis=z, ipscale=float(ip[0]))

    def grid_dims(self, v):
        gw, gh = C.c_int32(), C.c_int32()
        self.L.orc_grid_dims(self.h, v, C.byref(gw), C.byref(gh))
        return gw.value, gh.value

    def set_thresholds(self, ncc, before, depth):
        self.L.orc_set_thresholds(self.h, ncc, before, depth)

    def thresholds(self):
        a, b, d = C.c_float(), C.c_float(), C.c_int32()
        self.L.orc_get_thresholds(self.h, C.byref(a), C.byref(b), C.byref(d))
        return a.value, b.value, d.value

    def update_threshold(self):
        self.L.orc_update_threshold(self.h)

    # ---- patches
    def add_patches(self, recs):
        recs = convert_records(recs, self.dtype)
        if self.L.orc_add_patches(self.h, recs.shape[0], _ptr(recs)) != 0:
            raise RuntimeError(self.L.orc_last_error().decode())

    def num_patches(self):
        return self.L.orc_num_patches(self.h)

    def patches(self):
        n = self.num_patches()
        out = np.zeros(n, dtype=self.dtype)
        got = self.L.orc_get_patches(self.h, n, _ptr(out))
        return out[:got]

    def clear_patches(self):
        self.L.orc_clear_patches(self.h)

    def propagate(self, it):
        c = Counters()
        if self.L.orc_propagate(self.h, it, C.byref(c)) != 0:
            raise RuntimeError(self.L.orc_last_error().decode())
        return c.as_dict()

    def filter(self):
        r = np.zeros(4, dtype=np.int64)
        self.L.orc_filter(self.h, _ptr(r))
        return {"outside": int(r[0]), "exact": int(r[1]), "neighbor": int(r[2]), "groups": int(r[3])}

    def set_cell_budget(self, n):
        self.L.orc_set_cell_budget(self.h, n)

    def set_time_budget(self, seconds):
        self.L.orc_set_time_budget(self.h, float(seconds))

    def last_sweep_seconds(self):
        self.L.orc_last_sweep_seconds.restype = C.c_double
        self.L.orc_last_sweep_seconds.argtypes = [C.c_void_p]
        return float(self.L.orc_last_sweep_seconds(self.h))

    def engine_pass(self, it, p):
        c = Counters()
        self.L.orc_engine_pass(self.h, it, p, C.byref(c))
        return c.as_dict()

    def export_new(self):
        per_view = np.zeros(self.cfg.nviews, dtype=np.int32)
        n = self.L.orc_export_new(self.h, 0, None, _ptr(per_view))
        out = np.zeros(n, dtype=self.dtype)
        self.L.orc_export_new(self.h, n, _ptr(out), _ptr(per_view))
        return out, per_view

    def export_kills(self):
        n = self.L.orc_export_kills(self.h, 0, None)
        ids = np.zeros(n, dtype=np.int32)
        self.L.orc_export_kills(self.h, n, _ptr(ids))
        return ids

    def commit(self, recs, kills):
        recs = convert_records(recs, self.dtype)
        kills = np.ascontiguousarray(kills, dtype=np.int32)
        self.L.orc_commit(self.h, recs.shape[0], _ptr(recs), kills.shape[0], _ptr(kills))

    def depth_normal_map(self, view, kind):
        gw, gh = self.grid_dims(view)
        d = np.zeros((gh, gw), np.float32)
        n = np.zeros((gh, gw, 3), np.float32)
        ids = np.zeros((gh, gw), np.int32)
        self.L.orc_depth_normal_map(self.h, view, kind, _ptr(d), _ptr(n), _ptr(ids))
        return d, n, ids

    # ---- probes
    def project(self, v, coord, level):
        out = np.zeros(3, np.float32)
        self.L.orc_project(self.h, v, _ptr(f4(*coord)), level, _ptr(out))
        return out

    def unproject(self, v, icoord, level):
        out = np.zeros(4, np.float32)
        self.L.orc_unproject(self.h, v, _ptr(f4(*icoord)), level, _ptr(out))
        return out

    def get_unit(self, v, coord):
        return self.L.orc_get_unit(self.h, v, _ptr(f4(*coord)))

    def get_paxes(self, v, coord, normal):
        px, py = np.zeros(4, np.float32), np.zeros(4, np.float32)
        self.L.orc_get_paxes(self.h, v, _ptr(f4(*coord)), _ptr(f4(*normal)), _ptr(px), _ptr(py))
        return px, py

    def get_color(self, v, x, y, level):
        out = np.zeros(3, np.float32)
        self.L.orc_get_color(self.h, v, x, y, level, _ptr(out))
        return out

    def get_tex(self, coord, px, py, normal, v, normalize=False):
        w = self.cfg.wsize
        out = np.zeros((w * w, 3), np.float32)
        flag = self.L.orc_get_tex(self.h, _ptr(f4(*coord)), _ptr(f4(*px)), _ptr(f4(*py)), _ptr(f4(*normal)), v,
                                  _ptr(out), int(normalize))
        return flag, out

    def _one(self, rec):
        return convert_records(np.array(rec).reshape(1), self.dtype).copy()  # a private copy: the probes write into it

    def compute_incc(self, rec, robust=1):
        a = self._one(rec)
        return self.L.orc_compute_incc(self.h, _ptr(a), robust)

    def compute_ncc(self, rec):
        a = self._one(rec)
        return self.L.orc_compute_ncc(self.h, _ptr(a))

    def set_inccs(self, rec, robust=0):
        a = self._one(rec)
        out = np.zeros(64, np.float32)
        n = self.L.orc_set_inccs(self.h, _ptr(a), robust, _ptr(out))
        return out[:n]

    def set_inccs_matrix(self, rec, robust=1):
        a = self._one(rec)
        n = int(a["nimages"][0])
        out = np.zeros((n, n), np.float32)
        self.L.orc_set_inccs_matrix(self.h, _ptr(a), robust, _ptr(out))
        return out

    def preprocess(self, rec):
        a = self._one(rec)
        f = self.L.orc_preprocess(self.h, _ptr(a))
        return f, a[0].copy()

    def refine(self, rec, key=(0, 0, 0, 0)):
        a = self._one(rec)
        k = np.asarray(key, dtype=np.uint32)
        f = self.L.orc_refine(self.h, _ptr(a), _ptr(k))
        return f, a[0].copy()

    def postprocess(self, rec):
        a = self._one(rec)
        f = self.L.orc_postprocess(self.h, _ptr(a))
        return f, a[0].copy()

    def cost(self, rec, x):
        a = self._one(rec)
        return self.L.orc_cost(self.h, _ptr(a), _ptr(f4(*x)))

    def encode(self, rec):
        a = self._one(rec)
        x = np.zeros(3, np.float32)
        self.L.orc_encode(self.h, _ptr(a), _ptr(x))
        return x

    def decode(self, rec, x):
        a = self._one(rec)
        c, n = np.zeros(4, np.float32), np.zeros(4, np.float32)
        self.L.orc_decode(self.h, _ptr(a), _ptr(f4(*x)), _ptr(c), _ptr(n))
        return c, n

    def generate_patch(self, src, icoord):
        a = self._one(src)
        out = np.zeros(1, dtype=self.dtype)
        f = self.L.orc_generate_patch(self.h, _ptr(a), _ptr(f4(*icoord)), _ptr(out))
        return f, out[0].copy()

    def is_neighbor(self, a, b, thr):
        return self.L.orc_is_neighbor(self.h, _ptr(self._one(a)), _ptr(self._one(b)), thr)

    def quad_residual(self, rec, coords):
        c = np.ascontiguousarray(coords, dtype=np.float32)
        return self.L.orc_quad_residual(self.h, _ptr(self._one(rec)), _ptr(c), c.shape[0])

    def compute_gain(self, rec):
        return self.L.orc_compute_gain(self.h, _ptr(self._one(rec)))

    def check(self, rec):
        a = self._one(rec)
        f = self.L.orc_check(self.h, _ptr(a))
        return f, a[0].copy()
