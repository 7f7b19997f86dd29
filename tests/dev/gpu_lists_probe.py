"""Developer probe (GPU box): neighbour lists handed from pass to pass (ListCtl) on the bench workload.  Four passes per scan
(poor prior, then the converged pose perturbed by about a millimetre and a third of one), lists off / on / on with the first pass
leaving lists too: per-pass kernel time (level-1 events), stragglers, list misses; sums of every pass bit for bit between the
settings."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from fast_limo_amd import synth, _lib

NMAP = int(os.environ.get("NMAP", 1000000)); LBOX = float(os.environ.get("LBOX", 100.0))
RINGS = int(os.environ.get("RINGS", 64)); AZ = int(os.environ.get("AZ", 1024))
mp = synth.box_world_map(NMAP, LBOX, 1)
scan = np.ascontiguousarray(synth.velodyne_scan(RINGS, AZ, LBOX, 2)[:, :3])
x0 = np.zeros(26); x0[6] = 1.0; x0[10] = 1.0; x0[25] = -9.809
xs = x0.copy(); xs[0:3] = synth.T_STAR_T
r, p_, y = [np.deg2rad(v) for v in synth.T_STAR_RPY_DEG]
cr, sr, cp, sp, cy, sy = np.cos(r / 2), np.sin(r / 2), np.cos(p_ / 2), np.sin(p_ / 2), np.cos(y / 2), np.sin(y / 2)
xs[3:7] = [sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy]
def nudge(x, d):
    o = x.copy(); o[0:3] += d
    return o
D1 = float(os.environ.get("D1", 1.0e-3)); D2 = float(os.environ.get("D2", 3.0e-4))
poses = (x0, xs, nudge(xs, [D1, -0.5 * D1, 0.3 * D1]), nudge(xs, [D1 + D2, -0.5 * D1, 0.3 * D1 - D2]))
cfg = _lib.default_match_cfg(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
REPS = int(os.environ.get("REPS", 14))
ref = None
for name, mode, margin in (("off", 0, -1.0), ("on", 1, -1.0), ("on+first", 2, -1.0), ("on m=0.05", 1, 0.05), ("on m=0.2", 1, 0.2)):
    ctx = _lib.HipCtx(0)
    ctx.map_config(); ctx.map_add(mp)
    ctx.set_lists(mode, margin)
    ctx.set_timing(1)
    acc = {}
    sums = []
    ctx.list_stats(reset=True)
    for rep in range(REPS):
        ctx.scan_set(scan)
        for k, x in enumerate(poses):
            t0 = time.perf_counter()
            HTH, HTh, M = ctx.match_reduce(x, cfg)
            wall = (time.perf_counter() - t0) * 1e6
            a, w, f = ctx.last_kernel_ms()
            ls = ctx.list_stats()
            if rep == 0: sums.append((HTH.copy(), HTh.copy(), M))
            if rep >= 2: acc.setdefault(k, []).append((a * 1e3, w * 1e3, f * 1e3, wall, ctx.last_stragglers(), M, ls["last_misses"]))
    if ref is None: ref = sums
    same = all(np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2] for a, b in zip(ref, sums))
    print("== lists %-10s  sums equal to lists-off bit for bit: %s   stats %s" % (name, same, ctx.list_stats()))
    for k in sorted(acc):
        v = np.array(acc[k])
        print("   pass %d: knn %6.2f  widen %6.2f  fit %6.2f us  wall %6.1f us  stragglers %5d  M %d  list misses %d" % (
            k + 1, np.median(v[:, 0]), np.median(v[:, 1]), np.median(v[:, 2]), np.median(v[:, 3]), int(v[-1, 4]), int(v[-1, 5]), int(v[-1, 6])))
    ctx.close()
print("LISTS PROBE DONE")
