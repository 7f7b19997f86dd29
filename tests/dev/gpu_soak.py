"""Developer script: a long drive through new territory (map inserts on the worker thread, index merges, grid re-layouts,
growing buffers) -- per-scan time over the run, final index self-check, device memory in use."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from fast_limo_amd import synth, api
CAPS = dict(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
n_scans, n_pts, speed = int(os.environ.get("NSCANS", 600)), int(os.environ.get("NPTS", 20000)), 10.0
st, w, a = synth.stationary_imu(0.0, 0.1 * n_scans + 0.06)
G = api.Localizer(api.default_cfg(**CAPS))
G.set_flags(add_to_map=True, download_clouds=False, keep_log=False)
x0 = G.get_x(); x0[14] = speed; G.set_x(x0)
i = 0
ts = []
bad = 0
for k in range(n_scans):
    until = 0.1 * (k + 1) + 0.005
    while i < len(st) and st[i] <= until:
        G.update_imu(st[i], w[i], a[i]); i += 1
    scan = synth.corridor_scan(k, n_pts, 777, speed=speed)
    t0 = time.perf_counter(); rc = G.update_pointcloud(scan, 0.1 * k); ts.append(time.perf_counter() - t0)
    bad += int(rc not in (0, 1))
    if k % 100 == 99:
        import ctypes as C
        hip = C.CDLL("libamdhip64.so")
        fr, tot = C.c_size_t(0), C.c_size_t(0)
        hip.hipMemGetInfo(C.byref(fr), C.byref(tot))
        free, total = fr.value, tot.value
        print("scan %4d  map %8d  x %.2f  last 100 scans: median %.2f ms  max %.2f ms  | device memory in use %.2f GB"
              % (k, G.map_size(), G.get_x()[0], np.median(ts[-100:]) * 1e3, np.max(ts[-100:]) * 1e3, (total - free) / 2**30), flush=True)
mm, merges, builds = G.hip.grid_selfcheck()
print("status errors %d | index: %d merges, %d full builds, self-check mismatches %d | final x %.2f (true %.2f)"
      % (bad, merges, builds, mm, G.get_x()[0], speed * (0.1 * (n_scans - 1) + 0.1)))
G.close()
