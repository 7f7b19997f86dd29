"""ctypes binding of liboracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this module.  The oracle is a CPU restatement of the reference algorithm (see the headers of
``oracle/rl_*.h`` for the reference file:line each function follows).  PARITY UNPINNED: the
reference has no tests / golden vectors and cannot be compiled here.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

f32p = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")
f64p = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
i32p = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")


class OracleCfg(C.Structure):
    _fields_ = [
        ("NUM_MATCH_POINTS", C.c_int), ("MAX_NUM_MATCHES", C.c_int), ("MAX_NUM_PC2MATCH", C.c_int),
        ("bucket_size", C.c_int),
        ("MAX_DIST_PLANE", C.c_double), ("PLANE_THRESHOLD", C.c_double),
        ("min_extent", C.c_float), ("downsampling", C.c_int),
        ("MAX_NUM_ITERS", C.c_int), ("estimate_extrinsics", C.c_int),
        ("LIMITS", C.c_double * 23),
        ("cov_gyro", C.c_double), ("cov_acc", C.c_double), ("cov_bias_gyro", C.c_double), ("cov_bias_acc", C.c_double),
        ("time_offset", C.c_int), ("end_of_sweep", C.c_int), ("num_threads", C.c_int),
        ("imu2baselink_t", C.c_float * 3), ("imu2baselink_R", C.c_float * 9),
        ("lidar2baselink_t", C.c_float * 3), ("lidar2baselink_R", C.c_float * 9),
        ("accel_bias", C.c_float * 3), ("gyro_bias", C.c_float * 3), ("imu_sm", C.c_float * 9),
        ("gravity_align", C.c_int), ("calibrate_accel", C.c_int), ("calibrate_gyro", C.c_int),
        ("imu_calib_time", C.c_double),
        ("voxel_active", C.c_int), ("leaf_size", C.c_float),
        ("sensor_type", C.c_int),
        ("crop_active", C.c_int), ("crop_min", C.c_float * 3), ("crop_max", C.c_float * 3),
        ("dist_active", C.c_int), ("min_dist", C.c_double),
        ("rate_active", C.c_int), ("rate_value", C.c_int),
        ("fov_active", C.c_int), ("fov_angle", C.c_float),
    ]


# the reference's PointType (Common.hpp:100-113): xyz1, intensity, 4 bytes of padding, 8-byte time union
POINT_DTYPE = np.dtype({"names": ["x", "y", "z", "w", "intensity", "tu"],
                        "formats": [np.float32, np.float32, np.float32, np.float32, np.float32, np.uint64],
                        "offsets": [0, 4, 8, 12, 16, 24], "itemsize": 32})


def make_points(xyz, intensity=None, t_ns=None, time_s=None, timestamp=None):
    """Build PointType records; exactly one of t_ns (uint32, OUSTER), time_s (float32, VELODYNE) and
    timestamp (float64: HESAI seconds / LIVOX nanoseconds) selects the view of the time union."""
    xyz = np.asarray(xyz, np.float32).reshape(-1, 3)
    p = np.zeros(xyz.shape[0], POINT_DTYPE)
    p["x"], p["y"], p["z"], p["w"] = xyz[:, 0], xyz[:, 1], xyz[:, 2], 1.0
    if intensity is not None:
        p["intensity"] = intensity
    raw = p.view(np.uint8).reshape(-1, 32)
    if t_ns is not None:
        raw[:, 24:28] = np.asarray(t_ns, np.uint32).reshape(-1, 1).view(np.uint8)
    elif time_s is not None:
        raw[:, 24:28] = np.asarray(time_s, np.float32).reshape(-1, 1).view(np.uint8)
    elif timestamp is not None:
        raw[:, 24:32] = np.asarray(timestamp, np.float64).reshape(-1, 1).view(np.uint8)
    return p


MATCH_REC_DTYPE = np.dtype([
    ("p_global", np.float32, 3), ("n", np.float32, 4), ("dist", np.float32), ("is_plane", np.int32),
    ("n_nbr", np.int32), ("nbr", np.float32, (5, 3)), ("sqd", np.float32, 5)])


def default_cfg(**kw) -> OracleCfg:
    """Defaults of reference src/main.cpp:101-168 with the synthetic-benchmark deltas of SURVEY.md
    section 8 d (identity extrinsics / sm, calibration and filters off)."""
    c = OracleCfg()
    c.NUM_MATCH_POINTS = 5
    c.MAX_NUM_MATCHES = 2000
    c.MAX_NUM_PC2MATCH = 10000
    c.bucket_size = 2
    c.MAX_DIST_PLANE = 2.0
    c.PLANE_THRESHOLD = 5.0e-2
    c.min_extent = 0.2
    c.downsampling = 1
    c.MAX_NUM_ITERS = 3
    c.estimate_extrinsics = 1
    for i in range(23):
        c.LIMITS[i] = 1e-3
    c.cov_gyro, c.cov_acc, c.cov_bias_gyro, c.cov_bias_acc = 6e-4, 1e-2, 1e-5, 3e-4
    c.time_offset, c.end_of_sweep, c.num_threads = 1, 0, 10
    c.gravity_align = c.calibrate_accel = c.calibrate_gyro = 0
    c.imu_calib_time = 3.0
    c.voxel_active, c.leaf_size = 0, 0.25
    c.sensor_type = 1                                   # VELODYNE
    c.crop_active, c.dist_active, c.rate_active, c.fov_active = 0, 0, 0, 0
    for i in range(3):
        c.crop_min[i], c.crop_max[i] = -1.0, 1.0
    c.min_dist, c.rate_value, c.fov_angle = 4.0, 4, 3.14159265
    eye = [1, 0, 0, 0, 1, 0, 0, 0, 1]
    for i in range(9):
        c.imu2baselink_R[i] = eye[i]
        c.lidar2baselink_R[i] = eye[i]
        c.imu_sm[i] = eye[i]
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    return c


def build(force: bool = False) -> str:
    """Compile liboracle.so with oracle/Makefile (g++, -ffp-contract=off)."""
    so = os.path.join(_HERE, "liboracle.so")
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".h", ".cpp"))]
    stale = (not os.path.exists(so)) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return so


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    so = build()
    L = C.CDLL(so)
    vp = C.c_void_p
    L.oracle_octree_create.restype = vp
    L.oracle_octree_create.argtypes = [C.c_float, C.c_int]
    L.oracle_octree_destroy.argtypes = [vp]
    L.oracle_octree_update.argtypes = [vp, f32p, C.c_size_t]
    L.oracle_octree_size.restype = C.c_size_t
    L.oracle_octree_size.argtypes = [vp]
    L.oracle_octree_points.restype = C.c_size_t
    L.oracle_octree_points.argtypes = [vp, f32p, C.c_size_t]
    L.oracle_octree_knn.restype = C.c_longlong
    L.oracle_octree_knn.argtypes = [vp, f32p, C.c_size_t, C.c_int, f32p, f32p, i32p, C.c_int]
    L.oracle_voxel_grid.restype = C.c_size_t
    L.oracle_voxel_grid.argtypes = [f32p, C.c_size_t, C.c_float, f32p, C.c_size_t]
    L.oracle_plane_fit.argtypes = [f32p, f32p, C.c_int, C.c_int, C.c_double, C.c_double, f32p, C.POINTER(C.c_int)]
    L.oracle_pose_mats.argtypes = [f64p, f32p, f32p, f32p, f32p, f32p]
    L.oracle_state_boxplus.argtypes = [f64p, f64p]
    L.oracle_state_boxminus.argtypes = [f64p, f64p, f64p]
    L.oracle_match_H.restype = C.c_longlong
    L.oracle_match_H.argtypes = [vp, C.POINTER(OracleCfg), f64p, f32p, C.c_size_t, C.c_void_p, f64p, f64p,
                                 C.POINTER(C.c_int)]
    L.oracle_loc_create.restype = vp
    L.oracle_loc_create.argtypes = [C.POINTER(OracleCfg)]
    L.oracle_loc_destroy.argtypes = [vp]
    L.oracle_loc_update_imu.argtypes = [vp, C.c_double, f32p, f32p]
    L.oracle_loc_update_pointcloud.restype = C.c_int
    L.oracle_loc_update_pointcloud.argtypes = [vp, f32p, C.c_size_t, C.c_double, C.c_int]
    L.oracle_loc_update_pointcloud_points.restype = C.c_int
    L.oracle_loc_update_pointcloud_points.argtypes = [vp, C.c_void_p, C.c_size_t, C.c_double, C.c_int]
    L.oracle_loc_map_add.argtypes = [vp, f32p, C.c_size_t, C.c_double]
    L.oracle_loc_map_size.restype = C.c_size_t
    L.oracle_loc_map_size.argtypes = [vp]
    L.oracle_loc_get_x.argtypes = [vp, f64p]
    L.oracle_loc_set_x.argtypes = [vp, f64p]
    L.oracle_loc_get_P.argtypes = [vp, f64p]
    L.oracle_loc_set_P.argtypes = [vp, f64p]
    L.oracle_loc_num_iters.restype = C.c_int
    L.oracle_loc_num_iters.argtypes = [vp]
    L.oracle_loc_get_iter.argtypes = [vp, C.c_int, C.POINTER(C.c_int), f64p, f64p, f64p, f64p]
    L.oracle_loc_get_pc2match.restype = C.c_size_t
    L.oracle_loc_get_pc2match.argtypes = [vp, f32p, C.c_size_t]
    L.oracle_loc_get_final_scan.restype = C.c_size_t
    L.oracle_loc_get_final_scan.argtypes = [vp, f32p, C.c_size_t]
    L.oracle_loc_get_stats.argtypes = [vp, f64p, C.POINTER(C.c_longlong), C.POINTER(C.c_longlong)]
    L.oracle_loc_deskew.restype = C.c_longlong
    L.oracle_loc_deskew.argtypes = [vp, f32p, C.c_size_t, C.c_double, f32p]
    L.oracle_loc_update_only.restype = C.c_int
    L.oracle_loc_update_only.argtypes = [vp, f32p, C.c_size_t]
    L.oracle_eskf_update_fixed.argtypes = [f64p, f64p, f64p, f64p, C.c_int, C.c_int, f64p, C.c_double, C.c_double,
                                           C.POINTER(C.c_int)]
    L.oracle_eskf_predict.argtypes = [f64p, f64p, C.c_double, f64p, f64p, f64p]
    _LIB = L
    return L


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


class Octree:
    """reference Objects/Octree.hpp restated (initialize/update/knn)."""

    def __init__(self, min_extent: float = 0.2, downsample: bool = True):
        self._h = lib().oracle_octree_create(min_extent, int(downsample))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_octree_destroy(self._h)
            self._h = None

    def update(self, xyz):
        xyz = _f32(xyz).reshape(-1, 3)
        lib().oracle_octree_update(self._h, xyz, xyz.shape[0])

    def size(self) -> int:
        return int(lib().oracle_octree_size(self._h))

    def points(self) -> np.ndarray:
        n = self.size()
        out = np.empty((max(n, 1), 3), dtype=np.float32)
        lib().oracle_octree_points(self._h, out, n)
        return out[:n]

    def knn(self, q, k: int = 5, num_threads: int = 1):
        q = _f32(q).reshape(-1, 3)
        nq = q.shape[0]
        nbr = np.empty((nq, k, 3), dtype=np.float32)
        sqd = np.empty((nq, k), dtype=np.float32)
        cnt = np.empty((nq,), dtype=np.int32)
        evals = lib().oracle_octree_knn(self._h, q, nq, k, nbr, sqd, cnt, num_threads)
        return nbr, sqd, cnt, int(evals)


def eigen_solver6(A):
    """Eigen::EigenSolver<Matrix6d> restated: (eigenvalues real, imag, real parts of the normalised eigenvectors as columns)."""
    A = np.ascontiguousarray(A, dtype=np.float64).reshape(36)
    wr = np.zeros(6); wi = np.zeros(6); V = np.zeros(36)
    L = lib()
    L.oracle_eigen_solver6.argtypes = [f64p, f64p, f64p, f64p]
    L.oracle_eigen_solver6.restype = None
    L.oracle_eigen_solver6(A, wr, wi, V)
    return wr, wi, V.reshape(6, 6)


def voxel_grid(xyz, leaf):
    xyz = _f32(xyz).reshape(-1, 3)
    out = np.empty((max(xyz.shape[0], 1), 3), np.float32)
    n = lib().oracle_voxel_grid(xyz, xyz.shape[0], float(leaf), out, xyz.shape[0])
    return out[:n]


def state_update(s25, time, t):
    """State::update on a flat state (p3 q4(xyzw) v3 g3 w3 a3 bg3 ba3); returns the updated copy."""
    s = np.ascontiguousarray(s25, dtype=np.float32).copy()
    L = lib()
    L.oracle_state_update.argtypes = [np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS"), C.c_double, C.c_double]
    L.oracle_state_update.restype = None
    L.oracle_state_update(s, float(time), float(t))
    return s


def plane_fit(nbr, sqd, k=5, max_dist_plane=2.0, plane_threshold=0.05):
    nbr = _f32(nbr).reshape(-1, 3)
    sqd = _f32(sqd).reshape(-1)
    n = np.zeros(4, dtype=np.float32)
    ok = C.c_int(0)
    lib().oracle_plane_fit(nbr, sqd, nbr.shape[0], k, max_dist_plane, plane_threshold, n, C.byref(ok))
    return n, bool(ok.value)


def pose_mats(x26):
    x26 = np.ascontiguousarray(x26, dtype=np.float64)
    RT = np.empty(16, np.float32); RTi = np.empty(16, np.float32); TLIi = np.empty(16, np.float32)
    Ri = np.empty(9, np.float32); RLIi = np.empty(9, np.float32)
    lib().oracle_pose_mats(x26, RT, RTi, TLIi, Ri, RLIi)
    return RT.reshape(4, 4), RTi.reshape(4, 4), TLIi.reshape(4, 4), Ri.reshape(3, 3), RLIi.reshape(3, 3)


def identity_x26(pos=(0, 0, 0), grav=(0, 0, -9.809)):
    x = np.zeros(26, dtype=np.float64)
    x[0:3] = pos
    x[6] = 1.0      # rot w
    x[10] = 1.0     # offR w
    x[23:26] = grav
    return x


def match_H(octree: Octree, cfg: OracleCfg, x26, scan_xyz, want_recs=True):
    scan = _f32(scan_xyz).reshape(-1, 3)
    n = scan.shape[0]
    recs = np.zeros(n, dtype=MATCH_REC_DTYPE) if want_recs else None
    H = np.zeros((n, 12), dtype=np.float64)
    h = np.zeros(n, dtype=np.float64)
    M = C.c_int(0)
    ev = lib().oracle_match_H(octree._h, C.byref(cfg), np.ascontiguousarray(x26, dtype=np.float64), scan, n,
                              recs.ctypes.data if recs is not None else None, H, h, C.byref(M))
    return recs, H[:M.value], h[:M.value], int(ev)


class Localizer:
    """reference Modules/Localizer.cpp restated (updateIMU / updatePointCloud, filters off)."""

    def __init__(self, cfg: OracleCfg):
        self.cfg = cfg
        self._h = lib().oracle_loc_create(C.byref(cfg))

    def __del__(self):
        if getattr(self, "_h", None):
            lib().oracle_loc_destroy(self._h)
            self._h = None

    def update_imu(self, stamp, ang_vel, lin_accel):
        lib().oracle_loc_update_imu(self._h, float(stamp), _f32(ang_vel), _f32(lin_accel))

    def update_pointcloud(self, pts5, stamp, add_to_map=True) -> int:
        p = _f32(pts5).reshape(-1, 5)
        return int(lib().oracle_loc_update_pointcloud(self._h, p, p.shape[0], float(stamp), int(add_to_map)))

    def update_pointcloud_points(self, pts32, stamp, add_to_map=True) -> int:
        """pts32: structured array in the reference's 32-byte PointType layout (see POINT_DTYPE)."""
        p = np.ascontiguousarray(pts32)
        assert p.dtype.itemsize == 32
        return int(lib().oracle_loc_update_pointcloud_points(self._h, p.ctypes.data, p.shape[0], float(stamp), int(add_to_map)))

    def map_add(self, xyz, stamp=0.0):
        xyz = _f32(xyz).reshape(-1, 3)
        lib().oracle_loc_map_add(self._h, xyz, xyz.shape[0], float(stamp))

    def map_size(self) -> int:
        return int(lib().oracle_loc_map_size(self._h))

    def get_x(self):
        x = np.empty(26, np.float64)
        lib().oracle_loc_get_x(self._h, x)
        return x

    def set_x(self, x):
        lib().oracle_loc_set_x(self._h, np.ascontiguousarray(x, dtype=np.float64))

    def get_P(self):
        P = np.empty(529, np.float64)
        lib().oracle_loc_get_P(self._h, P)
        return P.reshape(23, 23)

    def set_P(self, P):
        lib().oracle_loc_set_P(self._h, np.ascontiguousarray(P, dtype=np.float64).reshape(-1))

    def iters(self):
        out = []
        for i in range(lib().oracle_loc_num_iters(self._h)):
            M = C.c_int(0)
            HTH = np.empty(144, np.float64); HTh = np.empty(12, np.float64)
            dx = np.empty(23, np.float64); xa = np.empty(26, np.float64)
            lib().oracle_loc_get_iter(self._h, i, C.byref(M), HTH, HTh, dx, xa)
            out.append(dict(M=M.value, HTH=HTH.reshape(12, 12), HTh=HTh, dx=dx, x_after=xa))
        return out

    def pc2match(self):
        n = int(lib().oracle_loc_get_pc2match(self._h, np.empty((1, 3), np.float32), 0))
        out = np.empty((max(n, 1), 3), np.float32)
        lib().oracle_loc_get_pc2match(self._h, out, n)
        return out[:n]

    def final_scan(self):
        n = int(lib().oracle_loc_get_final_scan(self._h, np.empty((1, 3), np.float32), 0))
        out = np.empty((max(n, 1), 3), np.float32)
        lib().oracle_loc_get_final_scan(self._h, out, n)
        return out[:n]

    def stats(self):
        t = np.zeros(6, np.float64)
        ev = C.c_longlong(0); q = C.c_longlong(0)
        lib().oracle_loc_get_stats(self._h, t, C.byref(ev), C.byref(q))
        return dict(t_deskew=t[0], t_update=t[1], t_mapadd=t[2], t_sort=t[3], t_match=t[4], t_hrows=t[5],
                    evals=ev.value, queries=q.value)

    def deskew(self, pts5, stamp):
        p = _f32(pts5).reshape(-1, 5)
        out = np.empty((p.shape[0], 3), np.float32)
        n = lib().oracle_loc_deskew(self._h, p, p.shape[0], float(stamp), out)
        return None if n < 0 else out[:n]

    def update_only(self, xyz) -> int:
        xyz = _f32(xyz).reshape(-1, 3)
        return int(lib().oracle_loc_update_only(self._h, xyz, xyz.shape[0]))


def eskf_update_fixed(x26, P, H, h, max_iters=3, limits=None, R=0.001, D=5.0):
    x = np.ascontiguousarray(x26, dtype=np.float64).copy()
    Pm = np.ascontiguousarray(P, dtype=np.float64).reshape(-1).copy()
    H = np.ascontiguousarray(H, dtype=np.float64).reshape(-1, 12)
    h = np.ascontiguousarray(h, dtype=np.float64).reshape(-1)
    lim = np.full(23, 1e-3) if limits is None else np.ascontiguousarray(limits, dtype=np.float64)
    n = C.c_int(0)
    lib().oracle_eskf_update_fixed(x, Pm, H.reshape(-1) if H.size else np.zeros(1), h if h.size else np.zeros(1),
                                   H.shape[0], max_iters, lim, R, D, C.byref(n))
    return x, Pm.reshape(23, 23), n.value


def eskf_predict(x26, P, dt, Qdiag, acc, gyro):
    x = np.ascontiguousarray(x26, dtype=np.float64).copy()
    Pm = np.ascontiguousarray(P, dtype=np.float64).reshape(-1).copy()
    lib().oracle_eskf_predict(x, Pm, float(dt), np.ascontiguousarray(Qdiag, dtype=np.float64),
                              np.ascontiguousarray(acc, dtype=np.float64), np.ascontiguousarray(gyro, dtype=np.float64))
    return x, Pm.reshape(23, 23)
