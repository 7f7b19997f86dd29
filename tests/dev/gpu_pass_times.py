"""Developer probe (GPU box): per-pass kernel times of the bench workload, first pass (poor prior, unpruned) vs later passes
(converged pose, pruned), for the current env switches.  Level-2 timing (events around every stage) and the level-1 pair."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from fast_limo_amd import synth, _lib

mp = synth.box_world_map(1000000, 100.0, 1)
scan = np.ascontiguousarray(synth.velodyne_scan(64, 1024, 100.0, 2)[:, :3])
ctx = _lib.HipCtx(0)
ctx.map_config(); ctx.map_add(mp)
x0 = np.zeros(26); x0[6] = 1.0; x0[10] = 1.0; x0[25] = -9.809
xs = x0.copy(); xs[0:3] = synth.T_STAR_T
r, p_, y = [np.deg2rad(v) for v in synth.T_STAR_RPY_DEG]
cr, sr, cp, sp, cy, sy = np.cos(r / 2), np.sin(r / 2), np.cos(p_ / 2), np.sin(p_ / 2), np.cos(y / 2), np.sin(y / 2)
xs[3:7] = [sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy]
cfg = _lib.default_match_cfg(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
for level in (0, 1, 2):
    ctx.set_timing(level)
    acc = {}
    for rep in range(12):
        ctx.scan_set(scan)                       # new scan: no bound from a previous pass
        for k, x in enumerate((x0, xs, xs, xs)):
            t0 = time.perf_counter()
            _, _, M = ctx.match_reduce(x, cfg)
            wall = (time.perf_counter() - t0) * 1e6
            a, w, f = ctx.last_kernel_ms()
            if rep >= 2:
                acc.setdefault(k, []).append((a * 1e3, w * 1e3, f * 1e3, wall, ctx.last_stragglers(), M))
    for k in sorted(acc):
        v = np.array(acc[k])
        if k == 0: print("fused passes so far", ctx.fused_pass_count(), "of", ctx.pass_count())
        print("level %d pass %d: knn %6.2f  widen %6.2f  fit %6.2f us  wall %6.1f us  stragglers %d  M %d" % (
            level, k + 1, np.median(v[:, 0]), np.median(v[:, 1]), np.median(v[:, 2]), np.median(v[:, 3]), int(v[-1, 4]), int(v[-1, 5])))
ctx.close()
