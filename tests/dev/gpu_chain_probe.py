"""Developer probe: the chained update on configs[1] -- wall time per step, the algebra kernel's duration (HIP events on its
dispatch) and, under rocprofv3 --kernel-trace, the gaps between consecutive kernels of a chain (tools/chain_timeline.py).
usage: python tests/dev/gpu_chain_probe.py [steps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from fast_limo_amd import api, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
caps = dict(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
mp = synth.box_world_map(1000000, 100.0, 1)
scan = synth.velodyne_scan(64, 1024, 100.0, 2)
st, w, a = synth.stationary_imu(0.0, 0.35)
loc = api.Localizer(api.default_cfg(num_threads=8, **caps))
loc.set_flags(add_to_map=False, download_clouds=False, keep_log=False)
loc.map_add(mp)
i = 0
while st[i] <= 0.105:
    loc.update_imu(st[i], w[i], a[i]); i += 1
loc.update_pointcloud(scan, 0.0)
while st[i] <= 0.205:
    loc.update_imu(st[i], w[i], a[i]); i += 1
x_true = loc.get_x().copy()
x_prior = x_true.copy(); x_prior[0:3] += [0.3, -0.2, 0.1]
ang = np.deg2rad(1.0); x_prior[3:7] = [0, 0, np.sin(ang / 2), np.cos(ang / 2)]
loc.set_x(x_prior)
P_prior = loc.get_P()
rc = loc.update_pointcloud(scan, 0.1)
reg = loc.register_resident_call(x_prior, P_prior)
for _ in range(300):
    reg()
t0 = time.perf_counter()
for _ in range(steps):
    reg()
dt = (time.perf_counter() - t0) / steps
print("step %.1f us  (%.0f scans/s)  passes %d" % (1e6 * dt, 1.0 / dt, loc.hip.pass_count()))
loc.hip.set_timing(1); loc.hip.set_timing_stride(1)
loc.hip.timing_split(reset=True); loc.hip.chain_stats(reset=True)
for _ in range(20):
    reg()
d, cs = loc.hip.timing_split(reset=True), loc.hip.chain_stats(reset=True)
print("timed: one-launch pass %.2f us x %d, separate knn %.2f + second %.2f us x %d, algebra %.2f us x %d; chains %d handed back %d declined %d" % (
    1e3 * d["fused_ms"] / max(d["fused_n"], 1), d["fused_n"], 1e3 * d["knn_ms"] / max(d["separate_n"], 1),
    1e3 * d["fit_ms"] / max(d["separate_n"], 1), d["separate_n"], 1e3 * cs["algebra_ms"] / max(cs["algebra_n"], 1), cs["algebra_n"],
    cs["chains"], cs["handed_back"], cs["declined"]))
loc.hip.set_timing(0)
loc.close()
