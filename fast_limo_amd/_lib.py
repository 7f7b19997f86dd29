"""ctypes loader for the in-tree native libraries.

``libflimo_hip.so``  -- HIP kernels + C ABI declared in ``include/flimo_c.h`` (the drop-in boundary)
``libfast_limo.so``  -- host C++ mirror of the reference's Localizer / Mapper / esekf on top of it

There is no Python or CPU fallback: if a library is missing or no gfx950 device is present the
calls raise (``FlimoError``).
"""
from __future__ import annotations

import ctypes as C
import os
import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(_PKG)

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


class FlimoError(RuntimeError):
    pass


class MapCfg(C.Structure):
    _fields_ = [("min_extent", C.c_float), ("bucket_size", C.c_int), ("downsample", C.c_int),
                ("cell_size", C.c_float)]


class MatchCfg(C.Structure):
    _fields_ = [("NUM_MATCH_POINTS", C.c_int), ("MAX_NUM_MATCHES", C.c_int), ("MAX_NUM_PC2MATCH", C.c_int),
                ("MAX_DIST_PLANE", C.c_double), ("PLANE_THRESHOLD", C.c_double), ("estimate_extrinsics", C.c_int)]


class Frame(C.Structure):
    _fields_ = [("p", C.c_float * 3), ("q", C.c_float * 4), ("v", C.c_float * 3), ("g", C.c_float * 3),
                ("w", C.c_float * 3), ("a", C.c_float * 3), ("bg", C.c_float * 3), ("ba", C.c_float * 3),
                ("time", C.c_double)]


CHAIN_MAX_PASSES = 12


class FilterCfg(C.Structure):
    """flimo_filter_cfg (include/flimo_c.h)."""
    _fields_ = [("crop_active", C.c_int), ("crop_min", C.c_float * 3), ("crop_max", C.c_float * 3), ("dist_active", C.c_int),
                ("min_dist", C.c_float), ("rate_active", C.c_int), ("rate_value", C.c_int), ("time_kind", C.c_int),
                ("end_of_sweep", C.c_int), ("sweep_ref_time", C.c_double), ("fov_active", C.c_int), ("fov_angle", C.c_float)]


class ChainPass(C.Structure):
    _fields_ = [("M", C.c_int), ("stragglers", C.c_int), ("ties", C.c_int), ("HTH", C.c_double * 144), ("HTh", C.c_double * 12),
                ("dx", C.c_double * 23), ("x_after", C.c_double * 26)]


class ChainIO(C.Structure):
    """flimo_chain_io (include/flimo_c.h): arguments and results of flimo_update_chain."""
    _fields_ = [("x26", C.c_double * 26), ("P", C.c_double * 529), ("limits", C.c_double * 23), ("R", C.c_double), ("D", C.c_double),
                ("max_iter", C.c_int), ("want_log", C.c_int),
                ("status", C.c_int), ("reason", C.c_int), ("passes", C.c_int), ("it_next", C.c_int), ("t", C.c_int),
                ("x26_out", C.c_double * 26), ("meas_valid", C.c_int), ("meas_M", C.c_int), ("meas_HTH", C.c_double * 144),
                ("meas_HTh", C.c_double * 12), ("log", ChainPass * CHAIN_MAX_PASSES)]


MATCH_REC_DTYPE = np.dtype([
    ("H", np.float32, 12), ("h", np.float32), ("valid", np.float32), ("n", np.float32, 4),
    ("p_global", np.float32, 3), ("sqd", np.float32, 5), ("nbr", np.int32, 5), ("n_nbr", np.int32)])

FRAME_DTYPE = np.dtype([
    ("p", np.float32, 3), ("q", np.float32, 4), ("v", np.float32, 3), ("g", np.float32, 3), ("w", np.float32, 3),
    ("a", np.float32, 3), ("bg", np.float32, 3), ("ba", np.float32, 3), ("_pad", np.float32), ("time", np.float64)])

# every symbol include/flimo_c.h declares (tests check the .so exports each one)
HIP_SYMBOLS = [
    "flimo_ctx_create", "flimo_ctx_destroy", "flimo_last_error", "flimo_version",
    "flimo_map_config", "flimo_map_add", "flimo_map_clear", "flimo_map_size", "flimo_map_last_time",
    "flimo_map_points", "flimo_knn", "flimo_scan_set", "flimo_scan_size", "flimo_scan_get",
    "flimo_scan_voxel_filter", "flimo_raw_scan_set", "flimo_raw_scan_filter_set", "flimo_raw_scan_filter_order_set", "flimo_raw_scan_order", "flimo_deskew_resident", "flimo_deskew_resident_offset", "flimo_deskew",
    "flimo_match_reduce", "flimo_match_fetch", "flimo_match_fetch_H",
    "flimo_scan_to_world", "flimo_scan_clouds", "flimo_upload_stage", "flimo_match_reduce_overlap", "flimo_map_add_scan",
    "flimo_set_timing", "flimo_set_timing_stride", "flimo_set_timing_deferred", "flimo_pass_count", "flimo_fused_pass_count", "flimo_tie_stats", "flimo_map_index_bytes", "flimo_fine_stats", "flimo_map_grid_selfcheck", "flimo_set_debug_records", "flimo_last_kernel_ms",
    "flimo_last_candidates_per_query", "flimo_last_widen_count", "flimo_last_stragglers", "flimo_stragglers_by_pass", "flimo_timing_totals", "flimo_timing_split", "flimo_set_path_switches", "flimo_set_wait_timeout_ms", "flimo_insert_rule_replay", "flimo_plane_fit5_host", "flimo_plane_eval5_host", "flimo_calculate_H_host",
    "flimo_update_chain", "flimo_chain_stats", "flimo_set_update_mode", "flimo_update_mode", "flimo_scan_adopt", "flimo_set_pass_pipeline", "flimo_pass_pipeline_end", "flimo_pass_pipeline_last", "flimo_pass_pipeline_stats",
]

_hip = None


def hip_lib_path() -> str:
    # FLIMO_HIP_LIB: developer override (e.g. the phase-stamp build libflimo_hip_trace.so)
    return os.environ.get("FLIMO_HIP_LIB") or os.path.join(_PKG, "libflimo_hip.so")


def load_hip():
    """Load libflimo_hip.so and declare the C ABI.  Raises FlimoError when the library is absent."""
    global _hip
    if _hip is not None:
        return _hip
    path = hip_lib_path()
    if not os.path.exists(path):
        raise FlimoError(f"{path} not found: run `python -c 'import __graft_entry__ as g; g.build()'` first")
    L = C.CDLL(path)
    vp = C.c_void_p
    L.flimo_ctx_create.restype = C.c_int
    L.flimo_ctx_create.argtypes = [C.c_int, C.POINTER(vp)]
    L.flimo_ctx_destroy.restype = None
    L.flimo_ctx_destroy.argtypes = [vp]
    L.flimo_last_error.restype = C.c_char_p
    L.flimo_last_error.argtypes = [vp]
    L.flimo_version.restype = C.c_char_p
    L.flimo_map_config.argtypes = [vp, C.POINTER(MapCfg)]
    L.flimo_map_add.argtypes = [vp, f32p, C.c_size_t, C.c_size_t, C.c_double]
    L.flimo_map_clear.argtypes = [vp]
    L.flimo_map_size.restype = C.c_size_t
    L.flimo_map_size.argtypes = [vp]
    L.flimo_map_last_time.restype = C.c_double
    L.flimo_map_last_time.argtypes = [vp]
    L.flimo_map_points.argtypes = [vp, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.flimo_knn.argtypes = [vp, f32p, C.c_size_t, C.c_int, i32p, f32p, i32p]
    L.flimo_scan_set.argtypes = [vp, f32p, C.c_size_t, C.c_size_t]
    L.flimo_scan_size.restype = C.c_size_t
    L.flimo_scan_size.argtypes = [vp]
    L.flimo_scan_get.argtypes = [vp, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.flimo_scan_voxel_filter.argtypes = [vp, C.c_float, C.POINTER(C.c_size_t)]
    L.flimo_raw_scan_set.argtypes = [vp, f32p, C.c_size_t, C.c_size_t, f64p]
    L.flimo_deskew_resident.argtypes = [vp, C.c_void_p, C.c_size_t, f32p, f64p]
    L.flimo_deskew.argtypes = [vp, f32p, C.c_size_t, C.c_size_t, f64p, C.c_void_p, C.c_size_t, f32p, f64p]
    L.flimo_match_reduce.argtypes = [vp, f64p, C.POINTER(MatchCfg), f64p, f64p, C.POINTER(C.c_int)]
    L.flimo_match_fetch.argtypes = [vp, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.flimo_match_fetch_H.argtypes = [vp, C.c_void_p, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.flimo_scan_to_world.argtypes = [vp, f64p, C.c_void_p, C.c_size_t]
    L.flimo_scan_clouds.argtypes = [vp, f64p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
    L.flimo_upload_stage.argtypes = [vp, C.c_size_t, C.POINTER(C.c_void_p)]
    L.flimo_map_add_scan.argtypes = [vp, f64p, C.c_double]
    L.flimo_set_timing.argtypes = [vp, C.c_int]
    L.flimo_set_wait_timeout_ms.argtypes = [vp, C.c_int]
    L.flimo_set_timing_stride.argtypes = [vp, C.c_int]
    L.flimo_set_timing_deferred.argtypes = [vp, C.c_int]
    L.flimo_pass_count.restype = C.c_ulonglong
    L.flimo_pass_count.argtypes = [vp]
    L.flimo_tie_stats.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    L.flimo_fine_stats.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    L.flimo_map_index_bytes.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.flimo_fused_pass_count.restype = C.c_ulonglong
    L.flimo_fused_pass_count.argtypes = [vp]
    L.flimo_map_grid_selfcheck.restype = C.c_int
    L.flimo_map_grid_selfcheck.argtypes = [vp, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.flimo_set_debug_records.argtypes = [vp, C.c_int]
    L.flimo_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    L.flimo_timing_totals.argtypes = [vp, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                      C.POINTER(C.c_longlong), C.POINTER(C.c_longlong), C.c_int]
    L.flimo_timing_split.argtypes = [vp, f64p, C.c_int]
    L.flimo_set_path_switches.argtypes = [vp, C.c_int, C.c_int]
    L.flimo_insert_rule_replay.argtypes = [C.c_float, C.c_int, f32p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t)]
    L.flimo_calculate_H_host.argtypes = [f64p, f32p, f32p, f32p, C.c_size_t, C.c_int, f64p, f64p]
    L.flimo_update_chain.argtypes = [vp, C.POINTER(MatchCfg), C.POINTER(ChainIO)]
    L.flimo_chain_stats.argtypes = [vp, f64p, C.c_int]
    L.flimo_set_update_mode.argtypes = [vp, C.c_int]
    L.flimo_scan_adopt.argtypes = [vp, vp]
    L.flimo_raw_scan_filter_order_set.argtypes = [vp, C.c_void_p, C.c_size_t, C.POINTER(FilterCfg), C.c_int, C.POINTER(C.c_size_t),
                                                  C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.flimo_raw_scan_order.argtypes = [vp, C.c_void_p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.flimo_deskew_resident_offset.argtypes = [vp, C.c_void_p, C.c_size_t, np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS"),
                                               np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS"), C.c_double]
    L.flimo_set_pass_pipeline.argtypes = [vp, C.c_int]
    L.flimo_pass_pipeline_end.argtypes = [vp]
    L.flimo_pass_pipeline_last.argtypes = [vp]
    L.flimo_pass_pipeline_stats.argtypes = [vp, C.POINTER(C.c_ulonglong)]
    L.flimo_update_mode.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.flimo_last_widen_count.restype = C.c_int
    L.flimo_last_widen_count.argtypes = [vp]
    L.flimo_last_stragglers.restype = C.c_int
    L.flimo_last_stragglers.argtypes = [vp]
    L.flimo_stragglers_by_pass.argtypes = [vp, C.POINTER(C.c_int)]
    L.flimo_last_candidates_per_query.restype = C.c_double
    L.flimo_last_candidates_per_query.argtypes = [vp]
    for name in HIP_SYMBOLS:
        fn = getattr(L, name)
        if fn.restype is C.c_int and name not in ("flimo_ctx_create",):
            pass
    _hip = L
    return L


_ERRS = {-1: "no gfx950 device", -2: "invalid argument", -3: "HIP error", -4: "no map", -5: "too large",
         -6: "unsupported", -7: "TIMEOUT"}


class HipCtx:
    """Thin OO wrapper over one ``flimo_ctx`` (one GPU, one map, one stream)."""

    def __init__(self, device: int = 0):
        L = load_hip()
        h = C.c_void_p()
        rc = L.flimo_ctx_create(device, C.byref(h))
        if rc != 0:
            raise FlimoError(f"flimo_ctx_create({device}) failed: {_ERRS.get(rc, rc)}")
        self._h = h
        self._L = L

    def close(self):
        if getattr(self, "_h", None):
            self._L.flimo_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise FlimoError(f"{_ERRS.get(rc, rc)}: {self._L.flimo_last_error(self._h).decode()}")

    # ---- map ----
    def map_config(self, min_extent=0.2, bucket_size=2, downsample=True, cell_size=0.0):
        cfg = MapCfg(min_extent, bucket_size, int(downsample), cell_size)
        self._chk(self._L.flimo_map_config(self._h, C.byref(cfg)))

    def map_add(self, xyz, stamp=0.0):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        n = xyz.shape[0]
        stride = xyz.strides[0] if xyz.ndim == 2 else 12
        self._chk(self._L.flimo_map_add(self._h, xyz.reshape(-1), n, stride, float(stamp)))

    def map_clear(self):
        self._chk(self._L.flimo_map_clear(self._h))

    def map_size(self) -> int:
        return int(self._L.flimo_map_size(self._h))

    def map_points(self) -> np.ndarray:
        n = C.c_size_t(0)
        self._chk(self._L.flimo_map_points(self._h, None, 0, C.byref(n)))
        out = np.empty((max(n.value, 1), 3), dtype=np.float32)
        self._chk(self._L.flimo_map_points(self._h, out.ctypes.data, n.value, C.byref(n)))
        return out[:n.value]

    def knn(self, q, k=5):
        q = np.ascontiguousarray(q, dtype=np.float32).reshape(-1, 3)
        nq = q.shape[0]
        idx = np.empty((nq, k), np.int32)
        sqd = np.empty((nq, k), np.float32)
        cnt = np.empty((nq,), np.int32)
        self._chk(self._L.flimo_knn(self._h, q.reshape(-1), nq, k, idx.reshape(-1), sqd.reshape(-1), cnt))
        return idx, sqd, cnt

    # ---- scan ----
    def scan_set(self, xyz):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        n = xyz.shape[0]
        stride = xyz.strides[0] if xyz.ndim == 2 else 12
        self._chk(self._L.flimo_scan_set(self._h, xyz.reshape(-1), n, stride))

    def scan_size(self) -> int:
        return int(self._L.flimo_scan_size(self._h))

    def scan_get(self) -> np.ndarray:
        n = self.scan_size()
        out = np.empty((max(n, 1), 3), dtype=np.float32)
        m = C.c_size_t(0)
        self._chk(self._L.flimo_scan_get(self._h, out.ctypes.data, n, C.byref(m)))
        return out[:n]

    def scan_voxel_filter(self, leaf: float) -> int:
        n = C.c_size_t(0)
        self._chk(self._L.flimo_scan_voxel_filter(self._h, float(leaf), C.byref(n)))
        return int(n.value)

    def raw_scan_filter_order_set(self, records, time_order=0, **cfg):
        """flimo_raw_scan_filter_order_set: ``records`` = the sweep as 32-byte PointType records (itemsize 32) or, with bit 2 of
        ``time_order`` set, as 16-byte {x, y, z, time word} records.  Returns (kept, last_stamp, nan_stamp, tied)."""
        fc = FilterCfg()
        for k, v in cfg.items():
            if k in ("crop_min", "crop_max"):
                setattr(fc, k, (C.c_float * 3)(*v))
            else:
                setattr(fc, k, v)
        rec = np.ascontiguousarray(records)
        n = rec.shape[0]
        assert rec.dtype.itemsize * (rec.size // max(n, 1)) == (16 if (time_order & 4) else 32)
        kept, last, nan, tied = C.c_size_t(0), C.c_double(0), C.c_int(0), C.c_int(0)
        self._chk(self._L.flimo_raw_scan_filter_order_set(self._h, rec.ctypes.data, n, C.byref(fc), int(time_order), C.byref(kept),
                                                          C.byref(last), C.byref(nan), C.byref(tied)))
        return int(kept.value), float(last.value), int(nan.value), int(tied.value)

    def raw_scan_order(self):
        n = C.c_size_t(0)
        self._chk(self._L.flimo_raw_scan_order(self._h, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), np.uint32)
        self._chk(self._L.flimo_raw_scan_order(self._h, out.ctypes.data, n.value, C.byref(n)))
        return out[:n.value]

    def deskew_resident_offset(self, frames, L2B, last_x26, t_offset):
        assert frames.dtype == FRAME_DTYPE
        frames = np.ascontiguousarray(frames)
        self._chk(self._L.flimo_deskew_resident_offset(self._h, frames.ctypes.data, frames.shape[0],
                                                       np.ascontiguousarray(L2B, dtype=np.float32).reshape(-1),
                                                       np.ascontiguousarray(last_x26, dtype=np.float64), float(t_offset)))

    def scan_adopt(self, src: "HipCtx"):
        """The resident raw sweep of ``src`` becomes this context's (flimo_scan_adopt)."""
        self._chk(self._L.flimo_scan_adopt(self._h, src._h))

    def raw_scan_set(self, xyz, t):
        xyz = np.ascontiguousarray(xyz, dtype=np.float32)
        t = np.ascontiguousarray(t, dtype=np.float64)
        n = xyz.shape[0]
        stride = xyz.strides[0] if xyz.ndim == 2 else 12
        self._chk(self._L.flimo_raw_scan_set(self._h, xyz.reshape(-1), n, stride, t))

    def deskew_resident(self, frames: np.ndarray, L2B, last_x26):
        assert frames.dtype == FRAME_DTYPE
        frames = np.ascontiguousarray(frames)
        self._chk(self._L.flimo_deskew_resident(self._h, frames.ctypes.data, frames.shape[0],
                                                np.ascontiguousarray(L2B, dtype=np.float32).reshape(-1),
                                                np.ascontiguousarray(last_x26, dtype=np.float64)))

    # ---- measurement pass ----
    def match_reduce(self, x26, cfg: MatchCfg):
        HTH = np.zeros(144, np.float64)
        HTh = np.zeros(12, np.float64)
        M = C.c_int(0)
        self._chk(self._L.flimo_match_reduce(self._h, np.ascontiguousarray(x26, dtype=np.float64), C.byref(cfg), HTH,
                                             HTh, C.byref(M)))
        return HTH.reshape(12, 12), HTh, M.value

    def match_fetch(self) -> np.ndarray:
        n = C.c_size_t(0)
        self._chk(self._L.flimo_match_fetch(self._h, None, 0, C.byref(n)))
        out = np.zeros(max(n.value, 1), dtype=MATCH_REC_DTYPE)
        self._chk(self._L.flimo_match_fetch(self._h, out.ctypes.data, n.value, C.byref(n)))
        return out[:n.value]

    def match_fetch_H(self):
        n = C.c_size_t(0)
        self._chk(self._L.flimo_match_fetch_H(self._h, None, None, 0, C.byref(n)))
        H = np.zeros((max(n.value, 1), 12), np.float64)
        h = np.zeros(max(n.value, 1), np.float64)
        self._chk(self._L.flimo_match_fetch_H(self._h, H.ctypes.data, h.ctypes.data, n.value, C.byref(n)))
        return H[:n.value], h[:n.value]

    def scan_to_world(self, x26) -> np.ndarray:
        n = self.scan_size()
        out = np.empty((max(n, 1), 3), dtype=np.float32)
        self._chk(self._L.flimo_scan_to_world(self._h, np.ascontiguousarray(x26, dtype=np.float64), out.ctypes.data, n))
        return out[:n]

    def scan_clouds(self, x26):
        """(body, world) xyz of the resident scan in one round trip (flimo_scan_clouds); copies of the context's pinned records."""
        b, w, n = C.c_void_p(), C.c_void_p(), C.c_size_t(0)
        self._chk(self._L.flimo_scan_clouds(self._h, np.ascontiguousarray(x26, dtype=np.float64), C.byref(b), C.byref(w), C.byref(n)))
        if n.value == 0:
            return np.zeros((0, 3), np.float32), np.zeros((0, 3), np.float32)
        rec = lambda p: np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_float)), shape=(n.value, 4))[:, :3].copy()
        return rec(b), rec(w)

    def map_add_scan(self, x26, stamp=0.0):
        self._chk(self._L.flimo_map_add_scan(self._h, np.ascontiguousarray(x26, dtype=np.float64), float(stamp)))

    # ---- instrumentation ----
    def set_timing(self, level=2):
        """0 off, 1 k-NN kernel only, 2 every stage (True == 2)."""
        self._chk(self._L.flimo_set_timing(self._h, 2 if level is True else int(level)))

    def set_timing_stride(self, every=1):
        """Level 1 only: time every ``every``-th pass (sampling)."""
        self._chk(self._L.flimo_set_timing_stride(self._h, int(every)))

    def set_timing_deferred(self, on=True):
        """Read the timed passes' events when the totals are asked for, not right behind each pass (include/flimo_dev.h)."""
        self._chk(self._L.flimo_set_timing_deferred(self._h, 1 if on else 0))

    def pass_count(self) -> int:
        return int(self._L.flimo_pass_count(self._h))

    def fine_stats(self):
        o = (C.c_ulonglong * 4)()
        self._chk(self._L.flimo_fine_stats(self._h, o))
        return dict(active=bool(o[0]), points=int(o[1]), builds=int(o[2]), passes=int(o[3]))

    def tie_stats(self):
        o = (C.c_ulonglong * 2)()
        self._chk(self._L.flimo_tie_stats(self._h, o))
        return dict(passes_redone=int(o[0]), queries_settled=int(o[1]))

    def map_index_bytes(self):
        o = (C.c_uint64 * 6)()
        self._chk(self._L.flimo_map_index_bytes(self._h, o))
        return dict(points=int(o[0]), index=int(o[1]), second_level=int(o[2]), tiles=int(o[3]), tile_pool_relayouts=int(o[4]),
                    sorted_array_allocated=int(o[5]))

    def fused_pass_count(self) -> int:
        return int(self._L.flimo_fused_pass_count(self._h))

    def grid_selfcheck(self):
        """(mismatching words of the incrementally maintained index vs a from-scratch sort, merges, full builds)."""
        mm = C.c_uint64(0)
        st = (C.c_uint64 * 2)()
        self._chk(self._L.flimo_map_grid_selfcheck(self._h, C.byref(mm), st))
        return int(mm.value), int(st[0]), int(st[1])

    def set_debug_records(self, on=True):
        self._chk(self._L.flimo_set_debug_records(self._h, int(on)))

    def last_kernel_ms(self):
        a = C.c_float(0)
        b = C.c_float(0)
        d = C.c_float(0)
        self._chk(self._L.flimo_last_kernel_ms(self._h, C.byref(a), C.byref(b), C.byref(d)))
        return a.value, b.value, d.value

    def timing_totals(self, reset=False):
        a = C.c_double(0); b = C.c_double(0); d = C.c_double(0); n = C.c_longlong(0); q = C.c_longlong(0)
        self._chk(self._L.flimo_timing_totals(self._h, C.byref(a), C.byref(b), C.byref(d), C.byref(n), C.byref(q), int(reset)))
        return dict(knn_ms=a.value, widen_ms=b.value, fit_ms=d.value, passes=n.value, queries=q.value)

    def set_path_switches(self, tail=-1, fuse=-1):
        self._chk(self._L.flimo_set_path_switches(self._h, int(tail), int(fuse)))

    def timing_split(self, reset=False):
        o = np.zeros(6)
        self._chk(self._L.flimo_timing_split(self._h, o, int(reset)))
        return dict(fused_ms=o[0], fused_n=int(o[1]), knn_ms=o[2], widen_ms=o[3], fit_ms=o[4], separate_n=int(o[5]))

    def update_chain(self, cfg: "MatchCfg", x26, P, limits, R=0.001, D=5.0, max_iter=3, want_log=True):
        """The iterations of the update of the resident scan enqueued at once (flimo_update_chain).  Returns a dict: status (0 declined,
        2 handed back), reason (1 M < 23, 2 ties, 3 degenerate, 5 the iteration that ends the loop), passes, it_next, t, x (26), the
        handed-back iteration's sums (meas: M, HTH, HTh or None) and the per-pass log."""
        io = ChainIO()
        io.x26[:] = list(np.asarray(x26, dtype=np.float64))
        io.P[:] = list(np.asarray(P, dtype=np.float64).reshape(-1))
        io.limits[:] = list(np.asarray(limits, dtype=np.float64))
        io.R, io.D, io.max_iter, io.want_log = float(R), float(D), int(max_iter), int(bool(want_log))
        self._chk(self._L.flimo_update_chain(self._h, C.byref(cfg), C.byref(io)))
        n_log = min(CHAIN_MAX_PASSES, io.passes + (1 if io.status == 2 else 0))   # + the handed-back iteration's counts
        log = [dict(M=io.log[i].M, stragglers=io.log[i].stragglers, ties=io.log[i].ties,
                    HTH=np.array(io.log[i].HTH).reshape(12, 12), HTh=np.array(io.log[i].HTh), dx=np.array(io.log[i].dx),
                    x_after=np.array(io.log[i].x_after)) for i in range(n_log)]
        meas = dict(M=io.meas_M, HTH=np.array(io.meas_HTH).reshape(12, 12), HTh=np.array(io.meas_HTh)) if io.meas_valid else None
        return dict(status=io.status, reason=io.reason, passes=io.passes, it_next=io.it_next, t=io.t, x=np.array(io.x26_out),
                    meas=meas, log=log)

    def set_update_mode(self, mode: int):
        """0: chain or host loop by this host's launch -> result round trip; 1: host loop; 2: chain."""
        self._chk(self._L.flimo_set_update_mode(self._h, int(mode)))

    def update_mode(self):
        ch = C.c_int(0); rtt = C.c_double(0)
        self._chk(self._L.flimo_update_mode(self._h, C.byref(ch), C.byref(rtt)))
        return dict(chained=bool(ch.value), launch_rtt_us=rtt.value)

    def set_pass_pipeline(self, on: bool):
        self._chk(self._L.flimo_set_pass_pipeline(self._h, int(on)))

    def pass_pipeline_end(self):
        self._chk(self._L.flimo_pass_pipeline_end(self._h))

    def pass_pipeline_last(self):
        """The next match_reduce is the last pass its update can run: nothing is queued behind it."""
        self._chk(self._L.flimo_pass_pipeline_last(self._h))

    def pass_pipeline_stats(self):
        o = (C.c_ulonglong * 4)()
        self._chk(self._L.flimo_pass_pipeline_stats(self._h, o))
        return dict(published=int(o[0]), cancelled=int(o[1]), aged=int(o[2]), left=int(o[3]))

    def chain_stats(self, reset=False):
        o = np.zeros(5)
        self._chk(self._L.flimo_chain_stats(self._h, o, int(reset)))
        return dict(algebra_ms=o[0], algebra_n=int(o[1]), chains=int(o[2]), handed_back=int(o[3]), declined=int(o[4]))

    def last_widen_count(self) -> int:
        return int(self._L.flimo_last_widen_count(self._h))

    def set_wait_timeout_ms(self, ms: int):
        self._chk(self._L.flimo_set_wait_timeout_ms(self._h, int(ms)))

    def last_stragglers(self) -> int:
        return int(self._L.flimo_last_stragglers(self._h))

    def stragglers_by_pass(self):
        o = (C.c_int * 4)()
        self._chk(self._L.flimo_stragglers_by_pass(self._h, o))
        return [int(v) for v in o]

    def last_candidates_per_query(self) -> float:
        return float(self._L.flimo_last_candidates_per_query(self._h))


def default_match_cfg(**kw) -> MatchCfg:
    c = MatchCfg(5, 2000, 10000, 2.0, 5.0e-2, 1)
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def insert_rule_replay(batches, min_extent=0.2, downsample=True):
    """Host-only replay of the reference's octree insert rule; returns (keep flags per batch, stored count)."""
    L = load_hip()
    sizes = np.array([b.shape[0] for b in batches], dtype=np.uint64)
    xyz = np.ascontiguousarray(np.concatenate([np.asarray(b, dtype=np.float32).reshape(-1, 3) for b in batches]))
    keep = np.zeros(xyz.shape[0], dtype=np.uint8)
    stored = C.c_size_t(0)
    rc = L.flimo_insert_rule_replay(float(min_extent), int(downsample), xyz.reshape(-1), sizes.ctypes.data, len(batches),
                                    keep.ctypes.data, C.byref(stored))
    if rc != 0:
        raise FlimoError(f"flimo_insert_rule_replay failed ({rc})")
    out, off = [], 0
    for b in batches:
        out.append(keep[off:off + b.shape[0]].astype(bool))
        off += b.shape[0]
    return out, int(stored.value)
