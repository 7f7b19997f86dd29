"""Developer script: wall time per scan in the reference's shipped configuration (config/kitti.yaml: crop, min distance,
every 4th point, 1 m voxel grid, caps 1e4 / 5000, LiDAR off the IMU) on KITTI-sized raw sweeps, GPU product vs CPU oracle."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from fast_limo_amd import synth, api
import oracle_py as O
n_scans, n_pts, speed = int(os.environ.get("NSCANS", 20)), int(os.environ.get("NPTS", 120000)), 10.0
st, w, a = synth.stationary_imu(0.0, 0.1 * n_scans + 0.06)
lid_t = (8.086759e-01, -3.195559e-01, 7.997231e-01)
lid_R = (9.999976e-01, -7.854027e-04, 2.024406e-03, 7.553071e-04, 9.998898e-01, 1.482454e-02, -2.035826e-03, -1.482298e-02, 9.998881e-01)
common = dict(MAX_NUM_PC2MATCH=10000, MAX_NUM_MATCHES=5000, voxel_active=1, leaf_size=1.0, crop_active=1, dist_active=1, min_dist=4.0,
              rate_active=1, rate_value=4, time_offset=1, lidar2baselink_t=lid_t, lidar2baselink_R=lid_R,
              accel_bias=(0.01, 0.01, 0.01), gyro_bias=(0.01, 0.01, 0.01))
G = api.Localizer(api.default_cfg(cropBoxMin=(-1.0, -1.0, -1.0), cropBoxMax=(1.0, 1.0, 1.0), **common))
WITH_ORACLE = int(os.environ.get("ORACLE", 1)) != 0      # ORACLE=0: product only (the oracle's OpenMP team otherwise keeps spinning next to the host thread)
Lo = O.Localizer(O.default_cfg(crop_min=(-1.0, -1.0, -1.0), crop_max=(1.0, 1.0, 1.0), num_threads=int(os.environ.get("THREADS", 32)), **common))
x0 = G.get_x(); x0[14] = speed
G.set_x(x0); Lo.set_x(x0)
scans = [synth.corridor_scan(k, n_pts, 321, speed=speed) for k in range(n_scans)]
i = 0
tg, to = [], []
hist = []
for k in range(n_scans):
    until = 0.1 * (k + 1) + 0.005
    while i < len(st) and st[i] <= until:
        G.update_imu(st[i], w[i], a[i])
        if WITH_ORACLE: Lo.update_imu(st[i], w[i], a[i])
        i += 1
    t0 = time.perf_counter(); rg = G.update_pointcloud(scans[k], 0.1 * k); G.sync(); t1 = time.perf_counter()
    ro = Lo.update_pointcloud(scans[k], 0.1 * k) if WITH_ORACLE else rg; t2 = time.perf_counter()
    tg.append(t1 - t0); to.append(t2 - t1)
    sg = G.stage_times()
    xg, xo = G.get_x(), Lo.get_x()
    true_x = speed * (0.1 * k + 0.1) if k >= 1 else 0.0      # x(t) = 10 t at the scan-end stamp
    hist.append((np.abs(xg[:3] - xo[:3]).max(), G.map_size() - Lo.map_size(), abs(xg[0] - true_x), abs(xo[0] - true_x)))
    if k >= n_scans - 3:
        print("scan %d rc %d/%d raw %d -> pc2match %d  map %d/%d | GPU %.2f ms (prep %.2f deskew+sort %.2f update %.2f exit %.2f) CPU %.2f ms  dpos %.1e"
              % (k, rg, ro, n_pts, G.pc2match().shape[0], G.map_size(), Lo.map_size(), tg[-1] * 1e3, sg['host_prep'] * 1e3, sg['deskew'] * 1e3,
                 sg['update'] * 1e3, sg['map_insert'] * 1e3, to[-1] * 1e3, np.abs(G.get_x()[:3] - Lo.get_x()[:3]).max()))
print("GPU-vs-CPU deviation per scan [m] (map size difference):", " ".join("%.0e(%+d)" % (h[0], h[1]) for h in hist))
print("error along the drive vs the true position, scans 3..: GPU mean %.2e m, CPU mean %.2e m" % (np.mean([h[2] for h in hist[3:]]), np.mean([h[3] for h in hist[3:]])))
print("median per scan (scans 3..): GPU %.2f ms, CPU oracle %.2f ms" % (np.median(tg[3:]) * 1e3, np.median(to[3:]) * 1e3))
G.close()
