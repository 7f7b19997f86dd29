"""Developer script: the map inserts of one fuzz trial, with the index self-check after each (FLIMO_* as the trial set them)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from fast_limo_amd import synth, _lib
import oracle_py as oracle
SEED, trial, L, n_map, n_ins = int(os.environ.get("SEED", 9)), int(os.environ.get("TRIAL", 171)), float(os.environ.get("L", 10.0)), int(os.environ.get("NMAP", 30000)), int(os.environ.get("NINS", 3))
mp = synth.box_world_map(n_map, L, 1000 * SEED + trial)
xt = oracle.identity_x26(); xt[0:3] = synth.T_STAR_T
r, p_, y = [np.deg2rad(v) for v in synth.T_STAR_RPY_DEG]
cr, sr, cp, sp, cy, sy = np.cos(r / 2), np.sin(r / 2), np.cos(p_ / 2), np.sin(p_ / 2), np.cos(y / 2), np.sin(y / 2)
xt[3:7] = [sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy]
ctx = _lib.HipCtx(0)
ctx.map_config(); ctx.map_add(mp)
print("after the map:", ctx.grid_selfcheck(), ctx.map_index_bytes())
for j in range(n_ins):
    sw = np.ascontiguousarray(synth.velodyne_scan(32, 512, L, 9000 + 10 * trial + j)[:, :3])
    ctx.scan_set(sw); ctx.map_add_scan(xt, 0.1 * (j + 1))
    print("after sweep %d: map %d" % (j, ctx.map_size()), ctx.grid_selfcheck(), ctx.map_index_bytes())
scan = np.ascontiguousarray(synth.velodyne_scan(32, 512, L, 7000 + trial)[:, :3])
gcfg = _lib.default_match_cfg(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7, MAX_DIST_PLANE=2.0, PLANE_THRESHOLD=0.1, estimate_extrinsics=1)
ctx.scan_set(scan)
for k in range(3):
    x = xt.copy(); x[0] += 0.01 * k
    HTH, HTh, M = ctx.match_reduce(x, gcfg)
    print("pass %d: M %d" % (k, M), ctx.grid_selfcheck()[0])
    if os.environ.get("FETCH", "1") == "1":
        Hd, hd = ctx.match_fetch_H()
        print("   after fetch:", ctx.grid_selfcheck()[0])
print("fine", ctx.fine_stats(), "ties", ctx.tie_stats(), "selfcheck", ctx.grid_selfcheck())
ctx.close()
