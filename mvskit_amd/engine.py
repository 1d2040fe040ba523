"""ctypes binding of the C ABI in include/mvskit_engine.h.

There is no CPU path: if the HIP library is missing or no GPU is visible, creating an Engine raises."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import build
from . import synth
from .synth import MAX_IMAGES, PATCH_DTYPE  # noqa: F401  (PATCH_DTYPE mirrors mvs_patch at 32 list slots)

PROBE_NCC, PROBE_PREPROCESS, PROBE_REFINE, PROBE_POSTPROCESS, PROBE_COST, PROBE_MATH = range(6)

#: every symbol include/mvskit_engine.h declares
EXPORTS = [
    "mvs_last_error", "mvs_device_count", "mvs_default_config", "mvs_engine_create", "mvs_engine_destroy",
    "mvs_engine_set_views", "mvs_engine_grid_dims", "mvs_engine_get_pyramid", "mvs_engine_set_thresholds",
    "mvs_engine_get_thresholds", "mvs_engine_update_threshold", "mvs_engine_upload_patches",
    "mvs_engine_clear_patches", "mvs_engine_num_patches", "mvs_engine_download_patches", "mvs_engine_propagate",
    "mvs_engine_pass", "mvs_engine_export_counts", "mvs_engine_export_device", "mvs_engine_commit_device",
    "mvs_engine_commit_local", "mvs_engine_depth_normal_map", "mvs_engine_probe", "mvs_engine_last_timing",
    "mvs_engine_filter", "mvs_comm_unique_id", "mvs_engine_comm_init", "mvs_engine_comm_attach", "mvs_engine_comm_release",
    "mvs_engine_exchange", "mvs_list_cap", "mvs_engine_filter_stats", "mvs_patch_bytes", "mvs_engine_reserve", "mvs_engine_comm_info",
]


class Config(C.Structure):
    _fields_ = [("nviews", C.c_int32), ("level", C.c_int32), ("csize", C.c_int32), ("wsize", C.c_int32),
                ("minImageNum", C.c_int32), ("max_propag", C.c_int32), ("nccThreshold", C.c_float),
                ("maxAngleThreshold", C.c_float), ("quadThreshold", C.c_float), ("depth", C.c_int32),
                ("seed", C.c_uint32), ("refine_steps", C.c_int32), ("refine_rd0", C.c_float), ("refine_ra0", C.c_float),
                ("enable_check", C.c_int32), ("view_begin", C.c_int32), ("view_stride", C.c_int32),
                ("device", C.c_int32), ("view_propagation", C.c_int32), ("shard_index", C.c_int32), ("shard_count", C.c_int32),
                ("literal_groups", C.c_int32), ("max_patches", C.c_int64)]


class ViewDesc(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("P", C.c_float * 12), ("rgb", C.c_void_p),
                ("mask", C.c_void_p)]


class Counters(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("candidates", "prefiltered", "patches", "fail0", "fail1", "inserted",
                                         "replaced", "evals", "view_evals", "trimmed")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


class Timing(C.Structure):
    _fields_ = [("index_ms", C.c_float), ("sweep_ms", C.c_float), ("commit_ms", C.c_float), ("sweep_launches", C.c_int32),
                ("exchange_ms", C.c_float), ("exchange_bytes", C.c_int64), ("check_retried_cells", C.c_int64)]


class FilterStats(C.Structure):
    _fields_ = [(n, C.c_float) for n in ("outside_ms", "exact_ms", "neighbor_ms", "groups_ms", "rebuild_ms", "total_ms")] + \
               [(n, C.c_int64) for n in ("patches_in", "exact_patches", "exact_view_evals", "neighbor_patches", "neighbor_tasks", "neighbor_entries",
                                         "neighbor_visited", "neighbor_accepted", "neighbor_retried", "exchange_bytes")]


class EngineError(RuntimeError):
    """A call of the C ABI returned a negative mvs_status (`status`; include/mvskit_engine.h)."""

    def __init__(self, msg, status=None):
        super().__init__(msg)
        self.status = status


_libs = {}


def load_library(cap32: bool = False, cap: int = 0):
    """Loads libmvskit_engine.so (view lists of 16) -- or libmvskit_engine_cap32.so / _cap64.so, the same sources built with
    -DMVS_LISTCAP=32 / 64 (the latter with 192-byte records) -- built in-tree by mvskit_amd.build / __graft_entry__.build."""
    cap = cap or (32 if cap32 else 16)
    default = build.ENGINE_LIBS[cap]
    LIB_PATH = os.environ.get({16: "MVS_ENGINE_LIB", 32: "MVS_ENGINE_LIB32", 64: "MVS_ENGINE_LIB64"}[cap], default)  # development: A/B timing of two builds on one box
    if LIB_PATH in _libs:
        return _libs[LIB_PATH]
    if not os.path.exists(LIB_PATH):
        raise EngineError(f"{LIB_PATH} is missing: run `python -m mvskit_amd.build` (the engine has no CPU fallback)")
    try:
        # torch bundles its own libamdhip64.so.7; two HIP runtimes in one process cannot both open the GPU.
        # Loaded first, torch's copy satisfies this library's NEEDED entry (same SONAME), so both share one.
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp = C.c_void_p
    L.mvs_last_error.restype = C.c_char_p
    L.mvs_device_count.restype = C.c_int
    L.mvs_default_config.argtypes = [C.POINTER(Config)]
    L.mvs_engine_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    L.mvs_engine_destroy.argtypes = [vp]
    L.mvs_engine_set_views.argtypes = [vp, C.c_int, C.POINTER(ViewDesc)]
    L.mvs_engine_grid_dims.argtypes = [vp, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mvs_engine_get_pyramid.argtypes = [vp, C.c_int, C.c_int, vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.mvs_engine_set_thresholds.argtypes = [vp, C.c_float, C.c_float, C.c_int]
    L.mvs_engine_get_thresholds.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_int)]
    L.mvs_engine_update_threshold.argtypes = [vp]
    L.mvs_engine_upload_patches.argtypes = [vp, C.c_int64, vp]
    L.mvs_engine_clear_patches.argtypes = [vp]
    L.mvs_engine_num_patches.argtypes = [vp, C.POINTER(C.c_int64)]
    L.mvs_engine_download_patches.argtypes = [vp, C.c_int64, vp, C.POINTER(C.c_int64)]
    L.mvs_engine_propagate.argtypes = [vp, C.c_int, C.POINTER(Counters)]
    L.mvs_engine_pass.argtypes = [vp, C.c_int, C.c_int, C.POINTER(Counters)]
    L.mvs_engine_export_counts.argtypes = [vp, C.POINTER(C.c_int64), C.POINTER(C.c_int64), vp]
    L.mvs_engine_export_device.argtypes = [vp, vp, C.c_int64, vp, C.c_int64]
    L.mvs_engine_commit_device.argtypes = [vp, vp, C.c_int64, vp, C.c_int64]
    L.mvs_engine_commit_local.argtypes = [vp]
    L.mvs_engine_depth_normal_map.argtypes = [vp, C.c_int, C.c_int, vp, vp, vp]
    L.mvs_engine_probe.argtypes = [vp, C.c_int, C.c_int64, vp, vp, vp, vp, vp]
    L.mvs_engine_last_timing.argtypes = [vp, C.POINTER(Timing)]
    L.mvs_engine_filter.argtypes = [vp, vp]
    L.mvs_comm_unique_id.argtypes = [vp]
    L.mvs_engine_comm_init.argtypes = [vp, vp, C.c_int, C.c_int]
    L.mvs_engine_comm_attach.argtypes = [vp, vp, C.c_int, C.c_int]
    L.mvs_engine_comm_release.argtypes = [vp]
    L.mvs_engine_exchange.argtypes = [vp]
    L.mvs_list_cap.restype = C.c_int
    L.mvs_patch_bytes.restype = C.c_int
    L.mvs_engine_reserve.argtypes = [vp, C.c_int64]
    L.mvs_engine_filter_stats.argtypes = [vp, C.POINTER(FilterStats)]
    _libs[LIB_PATH] = L
    return L


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Engine:
    """One PmMvps instance whose Propagate::run lives on an MI355X (pmmvps/pmmvps.cpp:76-114)."""

    def __init__(self, nviews, list_cap=None, **kw):
        """list_cap: 16, 32 or 64 views per m_images / m_vimages list (which library); default: the smallest that holds `nviews`,
        so that no list is ever cut short.  `dtype` is the record this library takes and returns (192 bytes at 64)."""
        if list_cap is None:
            list_cap = 16 if nviews <= 16 else (32 if nviews <= 32 else 64)
        self.L = load_library(cap=list_cap)
        self.list_cap = self.L.mvs_list_cap()
        self.dtype = synth.patch_dtype((self.L.mvs_patch_bytes() - 64) // 2)
        self.cfg = Config()
        self.L.mvs_default_config(C.byref(self.cfg))
        self.cfg.nviews = nviews
        for k, v in kw.items():
            if not hasattr(self.cfg, k):
                raise AttributeError(k)
            setattr(self.cfg, k, v)
        self.h = C.c_void_p()
        self._check(self.L.mvs_engine_create(C.byref(self.cfg), C.byref(self.h)))
        self._keep = None

    def _check(self, status):
        if status != 0:
            raise EngineError(f"mvskit engine error {status}: {self.L.mvs_last_error().decode()}", status)

    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            self.L.mvs_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- views
    def set_scene(self, scene, masks=None):
        n = scene.nviews
        descs = (ViewDesc * n)()
        keep = []
        for v in range(n):
            img = np.ascontiguousarray(scene.images[v], dtype=np.uint8)
            keep.append(img)
            descs[v].width, descs[v].height = scene.W, scene.H
            P = np.ascontiguousarray(scene.P[v], dtype=np.float32).ravel()
            for k in range(12):
                descs[v].P[k] = float(P[k])
            descs[v].rgb = img.ctypes.data
            descs[v].mask = None
            if masks is not None:
                m = np.ascontiguousarray(masks[v], dtype=np.uint8)
                keep.append(m)
                descs[v].mask = m.ctypes.data
        self._check(self.L.mvs_engine_set_views(self.h, n, descs))

    def grid_dims(self, v):
        gw, gh = C.c_int(), C.c_int()
        self._check(self.L.mvs_engine_grid_dims(self.h, v, C.byref(gw), C.byref(gh)))
        return gw.value, gh.value

    def pyramid(self, v, level):
        w, h = C.c_int(), C.c_int()
        self._check(self.L.mvs_engine_get_pyramid(self.h, v, level, None, C.byref(w), C.byref(h)))
        out = np.empty((h.value, w.value, 3), dtype=np.uint8)
        self._check(self.L.mvs_engine_get_pyramid(self.h, v, level, _ptr(out), C.byref(w), C.byref(h)))
        return out

    def set_thresholds(self, ncc, before, depth):
        self._check(self.L.mvs_engine_set_thresholds(self.h, ncc, before, depth))

    def thresholds(self):
        a, b, d = C.c_float(), C.c_float(), C.c_int()
        self._check(self.L.mvs_engine_get_thresholds(self.h, C.byref(a), C.byref(b), C.byref(d)))
        return a.value, b.value, d.value

    def update_threshold(self):
        self._check(self.L.mvs_engine_update_threshold(self.h))

    # ---- patches
    def upload_patches(self, recs):
        recs = synth.convert_records(recs, self.dtype)
        self._check(self.L.mvs_engine_upload_patches(self.h, recs.shape[0], _ptr(recs)))

    add_patches = upload_patches

    def reserve(self, list_entries=0):
        """Sizes the cell indexes up front (0: MAX_NUM_OF_PATCHES per cell of every view): no allocation inside the iterations."""
        self._check(self.L.mvs_engine_reserve(self.h, int(list_entries)))

    def clear_patches(self):
        self._check(self.L.mvs_engine_clear_patches(self.h))

    def num_patches(self):
        n = C.c_int64()
        self._check(self.L.mvs_engine_num_patches(self.h, C.byref(n)))
        return n.value

    def patches(self):
        n = C.c_int64()
        self._check(self.L.mvs_engine_download_patches(self.h, 0, None, C.byref(n)))
        out = np.zeros(n.value, dtype=self.dtype)
        if n.value:
            self._check(self.L.mvs_engine_download_patches(self.h, n.value, _ptr(out), C.byref(n)))
        return out

    # ---- the hot path
    def propagate(self, it):
        c = Counters()
        self._check(self.L.mvs_engine_propagate(self.h, it, C.byref(c)))
        return c.as_dict()

    def filter(self):
        r = np.zeros(4, dtype=np.int64)
        self._check(self.L.mvs_engine_filter(self.h, _ptr(r)))
        return {"outside": int(r[0]), "exact": int(r[1]), "neighbor": int(r[2]), "groups": int(r[3])}

    def filter_stats(self):
        f = FilterStats()
        self._check(self.L.mvs_engine_filter_stats(self.h, C.byref(f)))
        return {n: getattr(f, n) for n, _ in f._fields_}

    def engine_pass(self, it, p):
        c = Counters()
        self._check(self.L.mvs_engine_pass(self.h, it, p, C.byref(c)))
        return c.as_dict()

    def export_counts(self):
        a, b = C.c_int64(), C.c_int64()
        pv = np.zeros(self.cfg.nviews, dtype=np.int32)
        self._check(self.L.mvs_engine_export_counts(self.h, C.byref(a), C.byref(b), _ptr(pv)))
        return a.value, b.value, pv

    def export_device(self, new_ptr, cap_new, kill_ptr, cap_kill):
        self._check(self.L.mvs_engine_export_device(self.h, new_ptr, cap_new, kill_ptr, cap_kill))

    def commit_device(self, new_ptr, n_new, kill_ptr, n_kill):
        self._check(self.L.mvs_engine_commit_device(self.h, new_ptr, n_new, kill_ptr, n_kill))

    def commit_local(self):
        self._check(self.L.mvs_engine_commit_local(self.h))

    # ---- multi-GPU (RCCL communicator inside the engine)
    COMM_ID_BYTES = 128

    def comm_unique_id(self):
        """ncclGetUniqueId as bytes; rank 0 calls it and hands the bytes to the other ranks."""
        buf = C.create_string_buffer(self.COMM_ID_BYTES)
        self._check(self.L.mvs_comm_unique_id(buf))
        return buf.raw

    def comm_init(self, uid: bytes, rank: int, world: int):
        assert len(uid) == self.COMM_ID_BYTES
        self._check(self.L.mvs_engine_comm_init(self.h, C.c_char_p(uid), rank, world))

    def comm_info(self):
        """rank / world the engine holds and what its communicator reports (ncclCommCount / ncclCommUserRank; -1 if it cannot be asked)"""
        r, w, cc, cr = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        self.L.mvs_engine_comm_info.argtypes = [C.c_void_p] + [C.POINTER(C.c_int)] * 4
        self._check(self.L.mvs_engine_comm_info(self.h, C.byref(r), C.byref(w), C.byref(cc), C.byref(cr)))
        return {"rank": r.value, "world": w.value, "comm_count": cc.value, "comm_rank": cr.value}

    def comm_release(self):
        self._check(self.L.mvs_engine_comm_release(self.h))

    def exchange(self):
        self._check(self.L.mvs_engine_exchange(self.h))

    def timing(self):
        t = Timing()
        self._check(self.L.mvs_engine_last_timing(self.h, C.byref(t)))
        return {"index_ms": t.index_ms, "sweep_ms": t.sweep_ms, "commit_ms": t.commit_ms, "sweep_launches": t.sweep_launches,
                "exchange_ms": t.exchange_ms, "exchange_bytes": int(t.exchange_bytes), "check_retried_cells": int(t.check_retried_cells)}

    def depth_normal_map(self, view, kind):
        gw, gh = self.grid_dims(view)
        d = np.zeros((gh, gw), np.float32)
        n = np.zeros((gh, gw, 3), np.float32)
        ids = np.zeros((gh, gw), np.int32)
        self._check(self.L.mvs_engine_depth_normal_map(self.h, view, kind, _ptr(d), _ptr(n), _ptr(ids)))
        return d, n, ids

    # ---- batched single functions
    def probe(self, op, recs=None, values=None):
        if op == PROBE_MATH:
            x = np.ascontiguousarray(values, dtype=np.float32)
            out = np.zeros((x.shape[0], 5), np.float32)
            self._check(self.L.mvs_engine_probe(self.h, op, x.shape[0], None, _ptr(x), None, _ptr(out), None))
            return out
        recs = synth.convert_records(recs, self.dtype)
        n = recs.shape[0]
        out_rec = np.zeros(n, dtype=self.dtype)
        out_f = np.zeros(n, np.float32)
        out_i = np.zeros(n, np.int32)
        self._check(self.L.mvs_engine_probe(self.h, op, n, _ptr(recs), None, _ptr(out_rec), _ptr(out_f), _ptr(out_i)))
        return out_rec, out_f, out_i
