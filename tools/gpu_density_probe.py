"""Developer script: how the k-NN kernel's work changes once scans have been inserted into the map (the steady state of the
real pipeline): mean candidates per query and k-NN kernel time of each pass, on the primed map and after every insert."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fast_limo_amd import synth, api
caps = dict(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
NMAP = int(os.environ.get("NMAP", 1000000)); LBOX = float(os.environ.get("LBOX", 100.0))
RINGS = int(os.environ.get("RINGS", 64)); AZ = int(os.environ.get("AZ", 1024)); NSCANS = int(os.environ.get("NSCANS", 6))
mp = synth.box_world_map(NMAP, LBOX, 1)
st, w, a = synth.stationary_imu(0.0, 0.1 * NSCANS + 0.4)
G = api.Localizer(api.default_cfg(num_threads=8, gpu_cell_size=float(os.environ.get("CELL", 0)), **caps))
G.set_flags(add_to_map=True, download_clouds=False, keep_log=False)
G.map_add(mp)
G.hip.set_timing(2)
i = 0
for k in range(NSCANS):
    until = 0.1 * (k + 1) + 0.005
    while st[i] <= until:
        G.update_imu(st[i], w[i], a[i]); i += 1
    scan = synth.velodyne_scan(RINGS, AZ, LBOX, 2 + k)
    G.hip.timing_totals(reset=True)
    rc = G.update_pointcloud(scan, 0.1 * k)
    G.sync()
    tt = G.hip.timing_totals()
    print("scan %d rc %d  map %d  passes %d: knn %.1f us/pass  widen %.1f  fit %.1f | candidates/query (last pass) %.1f  widened %d"
          % (k, rc, G.map_size(), tt["passes"], 1e3 * tt["knn_ms"] / max(tt["passes"], 1), 1e3 * tt["widen_ms"] / max(tt["passes"], 1),
             1e3 * tt["fit_ms"] / max(tt["passes"], 1), G.hip.last_candidates_per_query(), G.hip.last_widen_count()), flush=True)
G.close()
