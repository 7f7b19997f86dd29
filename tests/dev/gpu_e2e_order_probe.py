"""Developer probe (GPU box): is the end-to-end leg's "tied stamps slower than unique stamps" the stamps or the ORDER of the legs?
Three groups of 12 sweeps through one Localizer (map inserts on): tied, unique, tied again; median sweep (call + wait for its insert)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from fast_limo_amd import api, synth
caps = dict(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
mp, scan, imu = bench.workload(0, 64, 1024, 1000000, 100.0)
st, w, a = imu
loc = api.Localizer(api.default_cfg(num_threads=os.cpu_count() or 1, **caps))
loc.set_flags(add_to_map=False, download_clouds=False, keep_log=False)
bench.drive_to_prior(loc, mp, scan, imu)
loc.update_pointcloud(scan, 0.1)
i = loc._imu_cursor
loc.set_flags(add_to_map=True, download_clouds=False, keep_log=False)
k = 2
seed = 100
for label in ("tied", "unique", "tied", "unique"):
    tot = []
    for j in range(12):
        sc = synth.velodyne_scan(64, 1024, 100.0, seed); seed += 1
        if label == "unique":
            sc[:, 4] += (np.arange(sc.shape[0]) % 64).astype(np.float32) * np.float32(1.5e-6)
        sw = api.make_points_velodyne(sc)
        until = 0.1 * (k + 1) + 0.005
        while i < len(st) and st[i] <= until:
            loc.update_imu(st[i], w[i], a[i]); i += 1
        t1 = time.perf_counter()
        rc = loc.update_pointcloud_points(sw, 0.1 * k)
        loc.sync()
        tot.append(time.perf_counter() - t1)
        assert rc == 0, rc
        k += 1
    print("%-6s sweeps: %s  median of all %.3f ms, of the last 8 %.3f ms" % (label, " ".join("%.2f" % (1e3 * t) for t in tot), 1e3 * np.median(tot), 1e3 * np.median(tot[4:])), flush=True)
loc.close()
