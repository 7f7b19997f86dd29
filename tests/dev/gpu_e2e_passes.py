"""Developer probe (GPU box): per sweep of the end-to-end loop (fresh sweeps, map insert on): passes, time inside the update,
time inside flimo_match_reduce, how many passes ran as one launch, stragglers of the last pass."""
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
for j in range(14):
    until = 0.1 * (j + 1) + 0.005
    while st[i] <= until:
        G.update_imu(st[i], w[i], a[i]); i += 1
    sw = api.make_points_velodyne(synth.velodyne_scan(64, 1024, 100.0, 100 + j))
    G.sync()
    p0 = G.host_profile(); f0 = G.hip.fused_pass_count(); n0 = G.hip.pass_count()
    t0 = time.perf_counter(); rc = G.update_pointcloud_points(sw, 0.1 * j); t1 = time.perf_counter()
    G.sync(); t2 = time.perf_counter()
    p1 = G.host_profile(); s = G.stage_times()
    print("sweep %2d rc %d: call %.3f ms (+insert %.3f)  deskew %.3f update %.3f | passes %d (one launch: %d)  in match_reduce %.1f us  update-host %.1f us  stragglers(last) %d  map %d" % (
        j, rc, (t1 - t0) * 1e3, (t2 - t1) * 1e3, s["deskew"] * 1e3, s["update"] * 1e3, p1["passes"] - p0["passes"], G.hip.fused_pass_count() - f0,
        (p1["match_reduce_s"] - p0["match_reduce_s"]) * 1e6, (p1["update_s"] - p0["update_s"]) * 1e6, G.hip.last_stragglers(), G.map_size()))
G.close()
