"""Synthetic inputs for the registration hot path (SURVEY.md section 8 d).

All generators are deterministic in their seed (numpy ``RandomState`` = MT19937) and return
float32 arrays.  The scene is "box-world": a ground plane plus four walls, Gaussian noise of
``sigma`` metres along the surface normal.  The ground sits at ``z = -sensor_height`` so that a
body frame at the origin is ``sensor_height`` above it (the filter starts at identity,
reference ``Localizer.cpp:672-677``).
"""
from __future__ import annotations

import math
import numpy as np

__all__ = ["rpy_to_R", "box_world_map", "box_world_scan_random", "velodyne_scan", "stationary_imu",
           "T_STAR_T", "T_STAR_RPY_DEG", "corridor_map"]

# true pose offset T* used by every config (SURVEY.md section 8 d)
T_STAR_T = (0.30, -0.20, 0.05)
T_STAR_RPY_DEG = (0.5, -0.3, 1.0)


def rpy_to_R(roll: float, pitch: float, yaw: float) -> np.ndarray:
    """Rz(yaw) @ Ry(pitch) @ Rx(roll), float64, angles in radians."""
    cr, sr = math.cos(roll), math.sin(roll)
    cp, sp = math.cos(pitch), math.sin(pitch)
    cy, sy = math.cos(yaw), math.sin(yaw)
    Rx = np.array([[1, 0, 0], [0, cr, -sr], [0, sr, cr]])
    Ry = np.array([[cp, 0, sp], [0, 1, 0], [-sp, 0, cp]])
    Rz = np.array([[cy, -sy, 0], [sy, cy, 0], [0, 0, 1]])
    return Rz @ Ry @ Rx


def _surface_points(rs: np.random.RandomState, n: int, L: float, wall_h: float, ground_z: float,
                    sigma: float) -> np.ndarray:
    """60 % ground, 10 % per wall; noise along the normal."""
    n_ground = int(round(0.6 * n))
    n_wall = (n - n_ground) // 4
    counts = [n_ground, n_wall, n_wall, n_wall, n - n_ground - 3 * n_wall]
    out = np.empty((n, 3), dtype=np.float64)
    k = 0
    # ground
    m = counts[0]
    out[k:k + m, 0] = rs.uniform(-L, L, m)
    out[k:k + m, 1] = rs.uniform(-L, L, m)
    out[k:k + m, 2] = ground_z + rs.normal(0.0, sigma, m)
    k += m
    walls = [(0, +L), (0, -L), (1, +L), (1, -L)]
    for (axis, val), m in zip(walls, counts[1:]):
        other = 1 - axis
        out[k:k + m, axis] = val + rs.normal(0.0, sigma, m)
        out[k:k + m, other] = rs.uniform(-L, L, m)
        out[k:k + m, 2] = ground_z + rs.uniform(0.0, wall_h, m)
        k += m
    return out


def box_world_map(n: int, L: float, seed: int, sensor_height: float = 1.8, wall_h: float = 20.0,
                  sigma: float = 0.01) -> np.ndarray:
    """Map points (n, 3) float32 in the world frame."""
    rs = np.random.RandomState(seed)
    return _surface_points(rs, n, L, wall_h, -sensor_height, sigma).astype(np.float32)


def _to_body(p_world: np.ndarray, t=T_STAR_T, rpy_deg=T_STAR_RPY_DEG) -> np.ndarray:
    R = rpy_to_R(*[math.radians(a) for a in rpy_deg])
    return (p_world - np.asarray(t, dtype=np.float64)) @ R  # R^T (p - t), row-vector form


def box_world_scan_random(n: int, L: float, seed: int, sensor_height: float = 1.8, wall_h: float = 20.0,
                          sigma: float = 0.01, t=T_STAR_T, rpy_deg=T_STAR_RPY_DEG) -> np.ndarray:
    """cfg-1 style scan: n points sampled from the same surfaces, expressed in the body frame of the
    true pose T*.  Returns (n, 5) float32: x y z intensity time (time = i/n * 0.1 s)."""
    rs = np.random.RandomState(seed)
    pw = _surface_points(rs, n, L, wall_h, -sensor_height, sigma)
    pb = _to_body(pw, t, rpy_deg)
    out = np.zeros((n, 5), dtype=np.float32)
    out[:, :3] = pb.astype(np.float32)
    out[:, 3] = 1.0
    out[:, 4] = (np.arange(n, dtype=np.float64) / n * 0.1).astype(np.float32)
    return out


def velodyne_scan(rings: int, azimuths: int, L: float, seed: int, sensor_height: float = 1.8,
                  wall_h: float = 20.0, sigma: float = 0.01, t=T_STAR_T, rpy_deg=T_STAR_RPY_DEG,
                  vfov_deg=(-24.8, 2.0), sweep_s: float = 0.1) -> np.ndarray:
    """cfg-2 style scan: rings x azimuths rays cast from the true sensor pose T* into box-world.
    Returns (rings*azimuths, 5) float32 in time order (azimuth-major), body frame of T*;
    per-point time = azimuth/azimuths * sweep_s."""
    rs = np.random.RandomState(seed)
    Rw = rpy_to_R(*[math.radians(a) for a in rpy_deg])
    origin = np.asarray(t, dtype=np.float64)
    el = np.radians(np.linspace(vfov_deg[0], vfov_deg[1], rings))
    az = np.arange(azimuths, dtype=np.float64) / azimuths * 2.0 * math.pi
    AZ, EL = np.meshgrid(az, el, indexing="ij")          # azimuth-major
    d_body = np.stack([np.cos(EL) * np.cos(AZ), np.cos(EL) * np.sin(AZ), np.sin(EL)], axis=-1).reshape(-1, 3)
    d = d_body @ Rw.T                                       # world directions
    gz = -sensor_height
    big = 1e30
    with np.errstate(divide="ignore", invalid="ignore"):
        tg = np.where(d[:, 2] < 0, (gz - origin[2]) / d[:, 2], big)
        tx = np.where(d[:, 0] > 0, (L - origin[0]) / d[:, 0], np.where(d[:, 0] < 0, (-L - origin[0]) / d[:, 0], big))
        ty = np.where(d[:, 1] > 0, (L - origin[1]) / d[:, 1], np.where(d[:, 1] < 0, (-L - origin[1]) / d[:, 1], big))
    tt = np.minimum(tg, np.minimum(tx, ty))
    hit = origin + d * tt[:, None]
    which = np.where(tt == tg, 2, np.where(tt == tx, 0, 1))
    noise = rs.normal(0.0, sigma, hit.shape[0])
    hit[np.arange(hit.shape[0]), which] += noise           # noise along the surface normal
    pb = (hit - origin) @ Rw
    n = pb.shape[0]
    out = np.zeros((n, 5), dtype=np.float32)
    out[:, :3] = pb.astype(np.float32)
    out[:, 3] = 1.0
    tcol = np.repeat(np.arange(azimuths, dtype=np.float64) / azimuths * sweep_s, rings)
    out[:, 4] = tcol.astype(np.float32)
    return out


def stationary_imu(t0: float, t1: float, rate_hz: float = 200.0):
    """Stationary IMU stream: omega = 0, specific force (0, 0, +9.809) (SURVEY.md section 8 d).
    Returns (stamps float64 [k], ang_vel float32 [k,3], lin_accel float32 [k,3])."""
    k = int(math.floor((t1 - t0) * rate_hz)) + 1
    stamps = t0 + np.arange(k, dtype=np.float64) / rate_hz
    w = np.zeros((k, 3), dtype=np.float32)
    a = np.zeros((k, 3), dtype=np.float32)
    a[:, 2] = 9.809
    return stamps, w, a


def corridor_scan(k: int, n: int, seed: int, speed: float = 10.0, sweep_s: float = 0.1, half_width: float = 6.0,
                  sensor_height: float = 1.8, view: float = 30.0, sigma: float = 0.01) -> np.ndarray:
    """Scan k of the config-3 stand-in (SURVEY.md section 8 d): a sensor driving at `speed` m/s along +x through a
    corridor (ground, two side walls, transverse fins every 5 m that constrain x).  Every point is sampled
    from the surfaces within `view` metres of the sensor position AT ITS OWN TIMESTAMP, so the sweep carries
    real motion distortion.  Returns (n, 5) float32: x y z intensity time (time = i/n * sweep_s, sorted).
    The sweep reference time of scan k is k * sweep_s."""
    rs = np.random.RandomState(seed + 1000 * k)
    t = np.arange(n, dtype=np.float64) / n * sweep_s
    sx = speed * (k * sweep_s + t)                       # sensor x at each point's time
    kind = rs.uniform(size=n)
    gz = -sensor_height
    pw = np.empty((n, 3), dtype=np.float64)
    # ground
    g = kind < 0.55
    pw[g, 0] = sx[g] + rs.uniform(-view, view, g.sum())
    pw[g, 1] = rs.uniform(-half_width, half_width, g.sum())
    pw[g, 2] = gz + rs.normal(0, sigma, g.sum())
    # side walls
    w = (kind >= 0.55) & (kind < 0.85)
    side = np.where(rs.uniform(size=w.sum()) < 0.5, -1.0, 1.0)
    pw[w, 0] = sx[w] + rs.uniform(-view, view, w.sum())
    pw[w, 1] = side * half_width + rs.normal(0, sigma, w.sum())
    pw[w, 2] = gz + rs.uniform(0, 6.0, w.sum())
    # fins: planes x = 5 m * j, 3 <= |y| <= half_width, facing x
    f = kind >= 0.85
    j = np.round((sx[f] + rs.uniform(-view, view, f.sum())) / 5.0)
    pw[f, 0] = 5.0 * j + rs.normal(0, sigma, f.sum())
    pw[f, 1] = np.where(rs.uniform(size=f.sum()) < 0.5, -1.0, 1.0) * rs.uniform(3.0, half_width, f.sum())
    pw[f, 2] = gz + rs.uniform(0, 4.0, f.sum())
    out = np.zeros((n, 5), dtype=np.float32)
    out[:, 0] = (pw[:, 0] - sx).astype(np.float32)
    out[:, 1] = pw[:, 1].astype(np.float32)
    out[:, 2] = pw[:, 2].astype(np.float32)
    out[:, 3] = 1.0
    out[:, 4] = t.astype(np.float32)
    return out


def spinning_stamps(scan5: np.ndarray, columns: int = 1800, sweep_s: float = 0.1) -> np.ndarray:
    """The stamps of a spinning sensor: the sweep is `columns` firings, and every point of a firing (all rings of a column) carries
    that firing's stamp -- column j at j / columns * sweep_s, like a Velodyne driver writes them.  Returns a copy of the (n, 5)
    scan with its time column quantised down to its column's stamp (the order of the points is kept: arrival order)."""
    out = np.array(scan5, dtype=np.float32, copy=True)
    col = np.floor(out[:, 4].astype(np.float64) / sweep_s * columns)
    out[:, 4] = (col * (sweep_s / columns)).astype(np.float32)
    return out


def corridor_map(n: int, x0: float, x1: float, seed: int, half_width: float = 6.0, sensor_height: float = 1.8,
                 sigma: float = 0.01) -> np.ndarray:
    """World-frame map (n, 3) float32 of the corridor `corridor_scan` drives through, over x in [x0, x1]: the same ground,
    side walls and fins (same proportions), used to prime a multi-million-point rolling map for the config-3 stand-in."""
    rs = np.random.RandomState(seed)
    kind = rs.uniform(size=n)
    gz = -sensor_height
    pw = np.empty((n, 3), dtype=np.float64)
    g = kind < 0.55
    pw[g, 0] = rs.uniform(x0, x1, g.sum())
    pw[g, 1] = rs.uniform(-half_width, half_width, g.sum())
    pw[g, 2] = gz + rs.normal(0, sigma, g.sum())
    w = (kind >= 0.55) & (kind < 0.85)
    side = np.where(rs.uniform(size=w.sum()) < 0.5, -1.0, 1.0)
    pw[w, 0] = rs.uniform(x0, x1, w.sum())
    pw[w, 1] = side * half_width + rs.normal(0, sigma, w.sum())
    pw[w, 2] = gz + rs.uniform(0, 6.0, w.sum())
    f = kind >= 0.85
    j = np.round(rs.uniform(x0, x1, f.sum()) / 5.0)
    pw[f, 0] = 5.0 * j + rs.normal(0, sigma, f.sum())
    pw[f, 1] = np.where(rs.uniform(size=f.sum()) < 0.5, -1.0, 1.0) * rs.uniform(3.0, half_width, f.sum())
    pw[f, 2] = gz + rs.uniform(0, 4.0, f.sum())
    return pw.astype(np.float32)
