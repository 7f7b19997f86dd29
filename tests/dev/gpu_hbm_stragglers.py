"""Developer probe (GPU box): 256k x 20M (bench.py's hbm_regime).  Stragglers by pass position, the passes' dispatch times, and -- with the
records' counters on -- candidates per query and worklist length of the last pass."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from fast_limo_amd import api, synth
caps = dict(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
R = bench.HBM_REGIME
mp = synth.box_world_map(R["map_points"], R["box"], 1)
scan = synth.velodyne_scan(R["rings"], R["azimuths"], R["box"], 2)
imu = synth.stationary_imu(0.0, 0.35)
loc = api.Localizer(api.default_cfg(num_threads=os.cpu_count() or 1, **caps))
loc.set_flags(add_to_map=False, download_clouds=False, keep_log=False)
bench.drive_to_prior(loc, mp, scan, imu)
x_prior, P_prior = loc.get_x(), loc.get_P()
loc.update_pointcloud(scan, 0.1)
reg = loc.register_resident_call(x_prior, P_prior)
for _ in range(3):
    reg()
x_ref = loc.get_x()
t0 = time.perf_counter()
for _ in range(20):
    reg()
ms = 1e3 * (time.perf_counter() - t0) / 20
loc.hip.set_timing(1); loc.hip.set_timing_stride(1); loc.hip.set_timing_deferred(True)
loc.hip.timing_split(reset=True)
for _ in range(8):
    reg()
d = loc.hip.timing_split(reset=True)
loc.hip.set_timing_deferred(False); loc.hip.set_timing(0)
print("step %.3f ms; stragglers by pass position %s; one-launch passes %d of %d: %.1f us; separate: k-NN %.1f + widening %.1f + fit %.1f us; state as the reference run: %s"
      % (ms, loc.hip.stragglers_by_pass(), d["fused_n"], d["fused_n"] + d["separate_n"], 1e3 * d["fused_ms"] / max(1, d["fused_n"]),
         1e3 * d["knn_ms"] / max(1, d["separate_n"]), 1e3 * d["widen_ms"] / max(1, d["separate_n"]), 1e3 * d["fit_ms"] / max(1, d["separate_n"]),
         np.array_equal(loc.get_x(), x_ref)), flush=True)
loc.close()
