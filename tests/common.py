"""Shared scene builders for the tests (cfg 1 of BASELINE.json and smaller)."""
import numpy as np
from fast_limo_amd import synth

CAPS = dict(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)


def cfg1_scene(n_map=50000, n_scan=4096, L=25.0, sigma=0.01):
    mp = synth.box_world_map(n_map, L, 1, sigma=sigma)
    scan = synth.box_world_scan_random(n_scan, L, 2, sigma=sigma)
    imu = synth.stationary_imu(0.0, 0.35)
    return mp, scan, imu


def drive_two_scans(loc, mp, scan, imu, second_scan=None):
    """map prime -> IMU -> scan 1 (null iteration, reference a-note 8) -> IMU -> scan 2 (registered)."""
    st, w, a = imu
    loc.map_add(mp)
    i = 0
    rcs = []
    for until, stamp, pts in ((0.105, 0.0, scan), (0.205, 0.1, scan if second_scan is None else second_scan)):
        while i < len(st) and st[i] <= until:
            loc.update_imu(st[i], w[i], a[i])
            i += 1
        rcs.append(loc.update_pointcloud(pts, stamp))
    return rcs


def sort_rows(a):
    a = np.asarray(a)
    return a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]


def pose_delta(xa, xb):
    """(max |dpos| [m], rotation angle between the two attitudes [rad])."""
    dpos = float(np.abs(xa[0:3] - xb[0:3]).max())
    qa, qb = xa[3:7], xb[3:7]
    dot = abs(float(np.dot(qa, qb)) / (np.linalg.norm(qa) * np.linalg.norm(qb)))
    ang = 2.0 * np.arccos(min(1.0, dot))
    return dpos, float(ang)
