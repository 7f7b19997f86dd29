"""Developer script: wall time of the FULL Localizer::updatePointCloud path (host filters + time sort + upload +
GPU deskew + update + transform + map insert) on config 2, scan after scan, GPU vs oracle."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from fast_limo_amd import synth, api
import oracle_py as O
caps = dict(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
NMAP = int(os.environ.get("NMAP", 1000000)); LBOX = float(os.environ.get("LBOX", 100.0))
RINGS = int(os.environ.get("RINGS", 64)); AZ = int(os.environ.get("AZ", 1024)); NSCANS = int(os.environ.get("NSCANS", 6))
WITH_ORACLE = int(os.environ.get("ORACLE", 1)) != 0          # ORACLE=0: GPU only (large maps)
mp = synth.box_world_map(NMAP, LBOX, 1)
B2B = int(os.environ.get("B2B", 8))                          # scans per back-to-back run (two runs: async / sync insert)
st, w, a = synth.stationary_imu(0.0, 0.1 * (NSCANS + 2 * B2B) + 0.4)
G = api.Localizer(api.default_cfg(num_threads=32, gpu_cell_size=float(os.environ.get("CELL", 0)), **caps))
class _NoOracle:
    def map_add(self, *a): pass
    def update_imu(self, *a): pass
    def update_pointcloud(self, *a): return -9
    def stats(self): return dict(t_deskew=0.0, t_update=0.0, t_mapadd=0.0)
    def map_size(self): return 0
    def get_x(self): return np.zeros(26)
Lo = O.Localizer(O.default_cfg(num_threads=32, **caps)) if WITH_ORACLE else _NoOracle()
t0 = time.perf_counter(); G.map_add(mp); print("GPU map prime %.1f ms (%d points)" % ((time.perf_counter() - t0) * 1e3, NMAP), flush=True)
Lo.map_add(mp)
i = 0
for k in range(NSCANS):
    until = 0.1 * (k + 1) + 0.005
    while st[i] <= until:
        G.update_imu(st[i], w[i], a[i]); Lo.update_imu(st[i], w[i], a[i]); i += 1
    scan = synth.velodyne_scan(RINGS, AZ, LBOX, 2 + k)
    if int(os.environ.get("UNIQUE_TIMES", 0)):             # every point its own stamp (no ties: the time sort takes its fast path)
        scan[:, 4] += (np.arange(scan.shape[0]) % RINGS).astype(np.float32) * np.float32(1.5e-6)
    t0 = time.perf_counter(); rg = G.update_pointcloud(scan, 0.1 * k); tg = time.perf_counter() - t0
    t0 = time.perf_counter(); ro = Lo.update_pointcloud(scan, 0.1 * k); to = time.perf_counter() - t0
    sg = G.stage_times(); so = Lo.stats()
    print("scan %d rc %d/%d  GPU total %.2f ms (prep %.2f deskew+upload %.2f update %.2f insert %.2f)  |  CPU total %.1f ms (deskew %.1f update %.1f insert %.1f)  map %d/%d  dpos %.2e"
          % (k, rg, ro, tg * 1e3, sg['host_prep'] * 1e3, sg['deskew'] * 1e3, sg['update'] * 1e3, sg['map_insert'] * 1e3,
             to * 1e3, so['t_deskew'] * 1e3, so['t_update'] * 1e3, so['t_mapadd'] * 1e3, G.map_size(), Lo.map_size(),
             np.abs(G.get_x()[:3] - Lo.get_x()[:3]).max()), flush=True)
# Back to back, GPU only: the wall time per scan a caller streaming sweeps sees.  The map insert that ends scan k runs on the
# Mapper's worker thread while the host prepares scan k+1 (filters, time sort), so it is hidden up to the host share.
for label, on in (("async insert", True), ("sync insert", False)):
    G.set_async_insert(on)
    scans = []
    for j in range(B2B):
        sc = synth.velodyne_scan(RINGS, AZ, LBOX, 100 + j)
        if int(os.environ.get("UNIQUE_TIMES", 0)):
            sc[:, 4] += (np.arange(sc.shape[0]) % RINGS).astype(np.float32) * np.float32(1.5e-6)
        scans.append(sc)
    lat = []
    k0 = NSCANS if on else NSCANS + B2B
    G.sync(); T0 = time.perf_counter()
    for j in range(B2B):
        k = k0 + j
        until = 0.1 * (k + 1) + 0.005
        while st[i] <= until:
            G.update_imu(st[i], w[i], a[i]); i += 1
        t0 = time.perf_counter(); rg = G.update_pointcloud(scans[j], 0.1 * k); lat.append(time.perf_counter() - t0)
        assert rg == 0, rg
    G.sync(); T1 = time.perf_counter()
    print("back to back, %s: %.2f ms per scan (updatePointCloud returns after %.2f ms median), last insert %.2f ms, map %d"
          % (label, (T1 - T0) / B2B * 1e3, np.median(lat) * 1e3, G.last_insert_seconds() * 1e3, G.map_size()), flush=True)
G.close()
