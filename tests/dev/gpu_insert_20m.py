import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
import bench
from fast_limo_amd import api, synth
caps = dict(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
R = bench.HBM_REGIME
mp = synth.box_world_map(R["map_points"], R["box"], 1)
scan = synth.velodyne_scan(R["rings"], R["azimuths"], R["box"], 2)
imu = synth.stationary_imu(0.0, 0.35)
loc = api.Localizer(api.default_cfg(gpu_device=0, num_threads=8, **caps))
loc.set_flags(add_to_map=False, download_clouds=False, keep_log=False)
rc1 = bench.drive_to_prior(loc, mp, scan, imu)
rc2 = loc.update_pointcloud(scan, 0.1)
for k in range(4):
    t1 = time.perf_counter()
    loc.hip.map_add_scan(loc.get_x(), 0.2 + 0.1 * k)
    print("insert %d: %.3f ms, map %d" % (k, 1e3 * (time.perf_counter() - t1), loc.map_size()), flush=True)
loc.close()
