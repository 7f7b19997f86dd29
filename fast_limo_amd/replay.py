"""Sequence replay harness: reads a directory of PCD scans plus an IMU CSV, feeds them to
the Localizer the way the reference's ROS callbacks do (reference src/main.cpp:16-25,69-75: every IMU sample up to the
scan's last point, then the scan), and returns the trajectory.  No ROS, no datasets are shipped: `write_pcd` /
`write_imu_csv` exist so that tests and users can produce inputs in the same formats.

PCD support: DATA ascii and DATA binary (not binary_compressed), fields x y z [intensity] and one per-point time field:
`t` (uint32 ns, OUSTER), `time` (float32 s, VELODYNE), `timestamp` (float64: HESAI s / LIVOX ns).
IMU CSV: `stamp,gx,gy,gz,ax,ay,az` per line (header line optional).
KITTI raw recordings (the layout of BASELINE.json's config 3, `<drive>_extract/velodyne_points/data/*.bin` + `timestamps*.txt` and
`oxts/data/*.txt` + `timestamps.txt`): `read_kitti_bin`, `read_kitti_timestamps`, `read_kitti_oxts`; `replay` takes a directory of
`.bin` sweeps and the IMU as arrays just as well.  A raw KITTI sweep has no per-point time: it is synthesised from the azimuth (the
sensor spins at a constant rate), as the VELODYNE `time` field [s since the first point of the sweep]."""
from __future__ import annotations

import glob
import os

import numpy as np

POINT_DTYPE = np.dtype({"names": ["x", "y", "z", "w", "intensity", "tu"],
                        "formats": [np.float32, np.float32, np.float32, np.float32, np.float32, np.uint64],
                        "offsets": [0, 4, 8, 12, 16, 24], "itemsize": 32})
_TIME_FIELDS = {"t": (np.uint32, 4), "time": (np.float32, 4), "timestamp": (np.float64, 8)}
_NP = {("F", 4): np.float32, ("F", 8): np.float64, ("U", 1): np.uint8, ("U", 2): np.uint16, ("U", 4): np.uint32,
       ("U", 8): np.uint64, ("I", 1): np.int8, ("I", 2): np.int16, ("I", 4): np.int32, ("I", 8): np.int64}


def read_pcd(path: str) -> np.ndarray:
    """-> structured array in the reference's 32-byte PointType layout (time union filled from t / time / timestamp)."""
    with open(path, "rb") as f:
        raw = f.read()
    hdr, pos = {}, 0
    while True:
        end = raw.index(b"\n", pos)
        line = raw[pos:end].decode("ascii", "replace").strip()
        pos = end + 1
        if not line or line.startswith("#"):
            continue
        key, _, val = line.partition(" ")
        hdr[key.upper()] = val.split()
        if key.upper() == "DATA":
            break
    fields, sizes, types = hdr["FIELDS"], [int(v) for v in hdr["SIZE"]], hdr["TYPE"]
    counts = [int(v) for v in hdr.get("COUNT", ["1"] * len(fields))]
    n = int(hdr["POINTS"][0]) if "POINTS" in hdr else int(hdr["WIDTH"][0]) * int(hdr["HEIGHT"][0])
    names, formats = [], []
    for f_, s_, t_, c_ in zip(fields, sizes, types, counts):
        names.append(f_)
        formats.append((_NP[(t_, s_)], c_) if c_ > 1 else _NP[(t_, s_)])
    dt = np.dtype({"names": names, "formats": formats})
    mode = hdr["DATA"][0].lower()
    if mode == "binary":
        rec = np.frombuffer(raw, dtype=dt, count=n, offset=pos)
    elif mode == "ascii":
        txt = np.loadtxt(raw[pos:].decode("ascii").splitlines(), dtype=np.float64, ndmin=2)[:n]
        rec = np.zeros(n, dtype=dt)
        col = 0
        for f_, c_ in zip(fields, counts):
            rec[f_] = txt[:, col] if c_ == 1 else txt[:, col:col + c_]
            col += c_
    else:
        raise ValueError(f"{path}: DATA {mode} is not supported")
    out = np.zeros(n, POINT_DTYPE)
    out["x"], out["y"], out["z"], out["w"] = rec["x"], rec["y"], rec["z"], 1.0
    if "intensity" in names:
        out["intensity"] = rec["intensity"]
    bytes_ = out.view(np.uint8).reshape(-1, 32)
    for name, (typ, width) in _TIME_FIELDS.items():
        if name in names:
            bytes_[:, 24:24 + width] = np.ascontiguousarray(rec[name].astype(typ)).reshape(-1, 1).view(np.uint8)
            break
    return out


def write_pcd(path: str, xyz, intensity=None, time_field: str = "time", time_values=None, binary: bool = True) -> None:
    xyz = np.asarray(xyz, np.float32).reshape(-1, 3)
    n = xyz.shape[0]
    typ, width = _TIME_FIELDS[time_field]
    dt = np.dtype([("x", np.float32), ("y", np.float32), ("z", np.float32), ("intensity", np.float32), (time_field, typ)])
    rec = np.zeros(n, dt)
    rec["x"], rec["y"], rec["z"] = xyz[:, 0], xyz[:, 1], xyz[:, 2]
    rec["intensity"] = 0.0 if intensity is None else intensity
    rec[time_field] = 0 if time_values is None else np.asarray(time_values).astype(typ)
    tcode = {"t": "U", "time": "F", "timestamp": "F"}[time_field]
    head = (f"# .PCD v0.7 - Point Cloud Data file format\nVERSION 0.7\nFIELDS x y z intensity {time_field}\n"
            f"SIZE 4 4 4 4 {width}\nTYPE F F F F {tcode}\nCOUNT 1 1 1 1 1\nWIDTH {n}\nHEIGHT 1\nVIEWPOINT 0 0 0 1 0 0 0\n"
            f"POINTS {n}\nDATA {'binary' if binary else 'ascii'}\n")
    with open(path, "wb") as f:
        f.write(head.encode("ascii"))
        if binary:
            f.write(rec.tobytes())
        else:
            for r in rec:
                f.write((" ".join(repr(v.item()) for v in r) + "\n").encode("ascii"))


def read_kitti_bin(path: str, sweep_s: float = 0.1, clockwise: bool = True) -> np.ndarray:
    """KITTI Velodyne sweep (`float32 x, y, z, reflectance` per point) -> the 32-byte PointType layout.  `time` = the fraction of
    the revolution at the point's azimuth, counted from the first point's azimuth in the sensor's spin direction (KITTI's HDL-64E
    spins clockwise seen from above), times `sweep_s`."""
    a = np.fromfile(path, dtype=np.float32)
    a = a[: (a.size // 4) * 4].reshape(-1, 4)
    out = np.zeros(a.shape[0], POINT_DTYPE)
    out["x"], out["y"], out["z"], out["w"], out["intensity"] = a[:, 0], a[:, 1], a[:, 2], 1.0, a[:, 3]
    if a.shape[0]:
        az = np.arctan2(a[:, 1].astype(np.float64), a[:, 0].astype(np.float64))
        d = (az[0] - az) if clockwise else (az - az[0])
        frac = np.mod(d, 2.0 * np.pi) / (2.0 * np.pi)
        out["tu"] = (frac * sweep_s).astype(np.float32).view(np.uint32).astype(np.uint64)      # float32 seconds in the union's low word
    return out


def read_kitti_timestamps(path: str) -> np.ndarray:
    """`YYYY-MM-DD hh:mm:ss.nnnnnnnnn` per line -> seconds since the first line (float64)."""
    import calendar, time as _time
    out = []
    with open(path) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            date, _, clock = line.partition(" ")
            hms, _, frac = clock.partition(".")
            secs = calendar.timegm(_time.strptime(date + " " + hms, "%Y-%m-%d %H:%M:%S"))
            out.append((secs, float("0." + (frac or "0"))))
    if not out:
        return np.zeros(0)
    s0 = out[0][0]
    t = np.array([(s - s0) + fr for s, fr in out], np.float64)
    return t - t[0]


def read_kitti_oxts(oxts_dir: str):
    """KITTI OXTS packets (`oxts/data/%010d.txt`, 30 values per file; `oxts/timestamps.txt`) -> (stamps [s since the first
    packet], gyro wx wy wz = values 17..19, accel ax ay az = values 11..13), the arrays `replay` takes as its IMU."""
    files = sorted(glob.glob(os.path.join(oxts_dir, "data", "*.txt")))
    st = read_kitti_timestamps(os.path.join(oxts_dir, "timestamps.txt"))
    rows = [np.loadtxt(f, dtype=np.float64).reshape(-1)[:30] for f in files]
    a = np.asarray(rows, np.float64).reshape(len(rows), -1)
    n = min(len(st), a.shape[0])
    return st[:n], a[:n, 17:20].astype(np.float32), a[:n, 11:14].astype(np.float32)


def read_imu_csv(path: str):
    rows = []
    with open(path) as f:
        for line in f:
            p = line.replace(",", " ").split()
            try:
                v = [float(x) for x in p[:7]]
            except ValueError:
                continue                       # header
            if len(v) == 7:
                rows.append(v)
    a = np.asarray(rows, np.float64).reshape(-1, 7)
    return a[:, 0], a[:, 1:4].astype(np.float32), a[:, 4:7].astype(np.float32)


def write_imu_csv(path: str, stamps, gyro, accel) -> None:
    with open(path, "w") as f:
        f.write("stamp,gx,gy,gz,ax,ay,az\n")
        for s, w, a in zip(stamps, gyro, accel):
            f.write("%.9f,%s,%s\n" % (s, ",".join(repr(float(v)) for v in w), ",".join(repr(float(v)) for v in a)))


def replay(loc, scan_dir: str, imu_csv, scan_stamps=None, sweep_s: float = 0.1, imu_lead_s: float = 0.005):
    """Feed every sweep of `scan_dir` (`*.pcd`, or KITTI `*.bin`; sorted by name) to `loc` (anything with update_imu /
    update_pointcloud_points / get_x: the product's api.Localizer or the oracle's Localizer).  `imu_csv`: path of an IMU CSV, or
    the (stamps, gyro, accel) arrays themselves (read_kitti_oxts).  `scan_stamps`: sweep reference time per scan
    (default k * sweep_s).  IMU samples are delivered up to the end of each sweep (+ imu_lead_s) before the scan, the
    order the reference's two callbacks produce on a live system.  Returns (status codes, poses [n, 26])."""
    files = sorted(glob.glob(os.path.join(scan_dir, "*.pcd")))
    reader = read_pcd
    if not files:
        files = sorted(glob.glob(os.path.join(scan_dir, "*.bin")))
        reader = lambda f: read_kitti_bin(f, sweep_s)
    st, w, a = read_imu_csv(imu_csv) if isinstance(imu_csv, (str, os.PathLike)) else imu_csv
    if scan_stamps is None:
        scan_stamps = [k * sweep_s for k in range(len(files))]
    rcs, poses, i = [], [], 0
    for f, stamp in zip(files, scan_stamps):
        until = stamp + sweep_s + imu_lead_s
        while i < len(st) and st[i] <= until:
            loc.update_imu(st[i], w[i], a[i])
            i += 1
        rcs.append(loc.update_pointcloud_points(reader(f), stamp))
        poses.append(np.array(loc.get_x()))
    return rcs, np.asarray(poses)


def ate(poses_a, poses_b) -> float:
    """Absolute trajectory error (RMS of the position differences) of two pose sequences of equal length."""
    d = np.asarray(poses_a)[:, 0:3] - np.asarray(poses_b)[:, 0:3]
    return float(np.sqrt((d * d).sum(1).mean()))
