"""Developer probe (GPU box): where a back-to-back sweep's wall time goes (bench workload, arrival-order path)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from fast_limo_amd import synth, api
caps = dict(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
mp = synth.box_world_map(1000000, 100.0, 1)
st, w, a = synth.stationary_imu(0.0, 4.0)
G = api.Localizer(api.default_cfg(num_threads=32, **caps))
G.set_flags(add_to_map=True, download_clouds=False)
G.map_add(mp)
i = 0
sweeps = [api.make_points_velodyne(synth.velodyne_scan(64, 1024, 100.0, 100 + j)) for j in range(12)]
for sync in (False, True):
    G.set_async_insert(not sync)
    rows = []
    k0 = 0 if not sync else 14
    G.sync(); T0 = time.perf_counter()
    for j in range(12):
        k = k0 + j
        until = 0.1 * (k + 1) + 0.005
        while st[i] <= until:
            G.update_imu(st[i], w[i], a[i]); i += 1
        t0 = time.perf_counter(); rc = G.update_pointcloud_points(sweeps[j], 0.1 * k); dt = time.perf_counter() - t0
        s = G.stage_times()
        rows.append((dt, s["host_prep"], s["deskew"], s["update"], s["map_insert"], G.last_insert_seconds()))
    G.sync(); T1 = time.perf_counter()
    r = np.array(rows[3:]) * 1e3
    print("%s insert: %.3f ms per sweep back to back | call %.3f = prep %.3f + deskew %.3f + update %.3f + insert-call %.3f ; insert itself %.3f ms"
          % ("sync" if sync else "async", (T1 - T0) / 12 * 1e3, *np.median(r, axis=0)))
G.close()
