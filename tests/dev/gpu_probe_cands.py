"""Developer probe: candidates per query of the first pass on a crowded 1M map for the current FLIMO_PROBE."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from fast_limo_amd import synth, _lib
mp = synth.box_world_map(1000000, 100.0, 1)
x = np.zeros(26); x[6] = 1; x[10] = 1; x[25] = -9.809
x[0:3] = synth.T_STAR_T
r, p_, y = [np.deg2rad(v) for v in synth.T_STAR_RPY_DEG]
cr, sr, cp, sp, cy, sy = np.cos(r / 2), np.sin(r / 2), np.cos(p_ / 2), np.sin(p_ / 2), np.cos(y / 2), np.sin(y / 2)
x[3:7] = [sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy]
ctx = _lib.HipCtx(0); ctx.map_config(); ctx.map_add(mp)
for j in range(20):
    ctx.scan_set(np.ascontiguousarray(synth.velodyne_scan(64, 1024, 100.0, 100 + j)[:, :3])); ctx.map_add_scan(x, 0.0)
q = np.ascontiguousarray(synth.velodyne_scan(64, 1024, 100.0, 999)[:, :3])
cfg = _lib.default_match_cfg(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
ctx.set_debug_records(True)
ctx.scan_set(q); r1 = ctx.match_reduce(x, cfg); c1 = ctx.last_candidates_per_query()
r2 = ctx.match_reduce(x, cfg); c2 = ctx.last_candidates_per_query()
print("PROBE %s: candidates per query first pass %.1f, second pass %.1f  M %d" % (os.environ.get("FLIMO_PROBE", "default"), c1, c2, r2[2]))
ctx.close()
