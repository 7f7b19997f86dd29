"""Developer probe (GPU box): device insert rule on batches of different character (overlapping / fresh territory)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fast_limo_amd import synth, _lib
os.environ["FLIMO_PROF_INSERT"] = "1"
ctx = _lib.HipCtx(0)
ctx.map_config()
ctx.map_add(synth.box_world_map(1000000, 100.0, 1))
rs = np.random.RandomState(0)
cases = {
    "overlapping scan (64k)": np.ascontiguousarray(synth.velodyne_scan(64, 1024, 100.0, 2)[:, :3]),
    "fresh territory, 64k points in a 40 m cube 500 m away": (rs.uniform(-20, 20, (65536, 3)) + [500, 0, 0]).astype(np.float32),
    "fresh dense blob, 64k points within 1 m": (rs.normal(0, 0.3, (65536, 3)) + [-300, 50, 0]).astype(np.float32),
    "same dense blob again": (rs.normal(0, 0.3, (65536, 3)) + [-300, 50, 0]).astype(np.float32),
}
for name, pts in cases.items():
    n0 = ctx.map_size()
    t0 = time.perf_counter()
    ctx.map_add(pts)
    dt = time.perf_counter() - t0
    print(f"{name:58s} {dt * 1e3:8.2f} ms  stored {ctx.map_size() - n0}", flush=True)
ctx.close()
