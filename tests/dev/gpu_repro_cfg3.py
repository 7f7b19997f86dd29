"""Developer probe: are repeated registrations of the resident 256k scan against the 20M map identical?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from fast_limo_amd import synth, api
import bench
NMAP = int(os.environ.get("NMAP", 20000000)); LBOX = float(os.environ.get("LBOX", 447.0))
RINGS = int(os.environ.get("RINGS", 128)); AZ = int(os.environ.get("AZ", 2048))
caps = dict(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
mp = synth.box_world_map(NMAP, LBOX, 1)
scan = synth.velodyne_scan(RINGS, AZ, LBOX, 2)
imu = synth.stationary_imu(0.0, 0.5)
L = api.Localizer(api.default_cfg(num_threads=8, **caps)); L.set_flags(add_to_map=False, download_clouds=False, keep_log=True)
assert bench.drive_to_prior(L, mp, scan, imu) == 1
x0, P0 = L.get_x(), L.get_P()
assert L.update_pointcloud(scan, 0.1) == 0
xr = L.get_x()
print("first registration: M per pass", [p["M"] for p in L.passes()], "fused so far", L.hip.fused_pass_count(), "of", L.hip.pass_count(), "ties", L.hip.tie_stats())
prev = xr
for k in range(6):
    assert L.register_resident(x0, P0) == 0
    x = L.get_x()
    print("step %d: |x - x_ref| %.3e  |x - x_prev| %.3e  M %s  fused %d of %d  stragglers(last) %d ties %s" % (
        k, np.abs(x - xr).max(), np.abs(x - prev).max(), [p["M"] for p in L.passes()], L.hip.fused_pass_count(), L.hip.pass_count(), L.hip.last_stragglers(), L.hip.tie_stats()))
    prev = x
L.close()
