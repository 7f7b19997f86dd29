"""Developer probe (GPU box): time of one measurement pass with the reference's default caps
(MAX_NUM_PC2MATCH = 10000, MAX_NUM_MATCHES = 2000) against the uncapped pass of the benchmark."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fast_limo_amd import synth, _lib
mp = synth.box_world_map(1000000, 100.0, 1)
scan = np.ascontiguousarray(synth.velodyne_scan(64, 1024, 100.0, 2)[:, :3])
ctx = _lib.HipCtx(0)
ctx.map_config(); ctx.map_add(mp); ctx.scan_set(scan)
x0 = np.zeros(26); x0[6] = 1.0; x0[10] = 1.0; x0[25] = -9.809
for name, caps in (("uncapped", dict(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)),
                   ("pc2match 10000", dict(MAX_NUM_PC2MATCH=10000, MAX_NUM_MATCHES=10**7)),
                   ("reference defaults 10000 / 2000", dict(MAX_NUM_PC2MATCH=10000, MAX_NUM_MATCHES=2000))):
    cfg = _lib.default_match_cfg(**caps)
    for _ in range(5):
        HTH, HTh, M = ctx.match_reduce(x0, cfg)
    t0 = time.perf_counter()
    for _ in range(200):
        HTH, HTh, M = ctx.match_reduce(x0, cfg)
    dt = (time.perf_counter() - t0) / 200
    print(f"{name:34s} M = {M:6d}  {dt * 1e6:8.1f} us per pass")
ctx.close()
