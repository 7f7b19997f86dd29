"""Developer probe (GPU box): cost of a measurement pass on the primed map vs after N raw sweeps have been inserted into it
(crowded cells under the sensor).  Prints the one-launch pass time (HIP events) before / after and the ratio."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from fast_limo_amd import synth, api
caps = dict(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
NMAP = int(os.environ.get("NMAP", 1000000)); LBOX = float(os.environ.get("LBOX", 100.0))
RINGS = int(os.environ.get("RINGS", 64)); AZ = int(os.environ.get("AZ", 1024)); NINS = int(os.environ.get("NINS", 50))
mp = synth.box_world_map(NMAP, LBOX, 1)
st, w, a = synth.stationary_imu(0.0, 0.1 * (NINS + 20) + 0.4)
G = api.Localizer(api.default_cfg(num_threads=8, **caps))
G.set_flags(add_to_map=False, download_clouds=False, keep_log=False)
G.map_add(mp)
i = 0
k = 0
def sweep(seed, insert):
    global i, k
    until = 0.1 * (k + 1) + 0.005
    while st[i] <= until:
        G.update_imu(st[i], w[i], a[i]); i += 1
    G.set_flags(add_to_map=insert, download_clouds=False, keep_log=False)
    rc = G.update_pointcloud(synth.velodyne_scan(RINGS, AZ, LBOX, seed), 0.1 * k)
    G.sync()
    k += 1
    return rc
def measure(tag):
    G.hip.set_timing(1); G.hip.set_timing_stride(1)
    sweep(900, False)                       # warm
    G.hip.timing_split(reset=True)
    for s in range(4):
        sweep(901 + s, False)
    d = G.hip.timing_split(reset=True)
    G.hip.set_timing(0)
    one = 1e3 * d["fused_ms"] / max(d["fused_n"], 1)
    sep = 1e3 * (d["knn_ms"] + d["widen_ms"] + d["fit_ms"]) / max(d["separate_n"], 1)
    print("%s: map %d points; one-launch pass %.2f us (%d timed), separate-dispatch pass %.2f us (%d)" % (tag, G.map_size(), one, d["fused_n"], sep, d["separate_n"]), flush=True)
    return one
sweep(2, False); sweep(3, False)
t0 = measure("primed")
t_ins = time.perf_counter()
for j in range(NINS):
    sweep(100 + j, True)
print("inserted %d sweeps in %.1f ms each" % (NINS, (time.perf_counter() - t_ins) / NINS * 1e3))
t1 = measure("after %d raw sweeps" % NINS)
print("ratio %.2f" % (t1 / t0))
G.close()
