"""Developer benchmark (GPU box): pass time on a map into which N raw sweeps were inserted at the TRUE pose (deterministic map),
second level on / off.  Wall time of flimo_match_reduce per pass: first pass (no bound) and passes with the previous bound."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from fast_limo_amd import synth, _lib
NMAP = int(os.environ.get("NMAP", 1000000)); LBOX = float(os.environ.get("LBOX", 100.0))
RINGS = int(os.environ.get("RINGS", 64)); AZ = int(os.environ.get("AZ", 1024)); NINS = int(os.environ.get("NINS", 50))
mp = synth.box_world_map(NMAP, LBOX, 1)
x = np.zeros(26); x[6] = 1; x[10] = 1; x[25] = -9.809
x[0:3] = synth.T_STAR_T
r, p_, y = [np.deg2rad(v) for v in synth.T_STAR_RPY_DEG]
cr, sr, cp, sp, cy, sy = np.cos(r / 2), np.sin(r / 2), np.cos(p_ / 2), np.sin(p_ / 2), np.cos(y / 2), np.sin(y / 2)
x[3:7] = [sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy]
sweeps = [np.ascontiguousarray(synth.velodyne_scan(RINGS, AZ, LBOX, 100 + j)[:, :3]) for j in range(NINS)]
query = np.ascontiguousarray(synth.velodyne_scan(RINGS, AZ, LBOX, 999)[:, :3])
cfg = _lib.default_match_cfg(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
ref = None
for fine in os.environ.get("FINES", "1,0").split(","):
    os.environ["FLIMO_FINE"] = fine
    ctx = _lib.HipCtx(0)
    ctx.map_config(); ctx.map_add(mp)
    def cands(ctx):
        ctx.set_debug_records(True); ctx.scan_set(query); out = []
        for k in range(2):
            ctx.match_reduce(x, cfg); out.append("%.1f" % ctx.last_candidates_per_query())
        ctx.set_debug_records(False)
        return " / ".join(out)
    def timed(tag):
        ctx.scan_set(query)
        t = []
        for rep in range(6):
            ctx.scan_set(query)
            t0 = time.perf_counter(); r1 = ctx.match_reduce(x, cfg); t1 = time.perf_counter()
            r2 = ctx.match_reduce(x, cfg); t2 = time.perf_counter()
            r3 = ctx.match_reduce(x, cfg); t3 = time.perf_counter()
            t.append((t1 - t0, t2 - t1, t3 - t2))
        t = np.median(np.array(t[1:]), axis=0) * 1e6
        # kernel times of the same three passes (events on the dispatches)
        ctx.set_timing(1); kt = []
        for rep in range(4):
            ctx.scan_set(query); row = []
            for k in range(3):
                ctx.match_reduce(x, cfg); a, w, f = ctx.last_kernel_ms(); row.append((a + w + f) * 1e3)
                if k == 0: parts = (a * 1e3, w * 1e3, f * 1e3)
            kt.append(row)
        ctx.set_timing(0)
        kt = np.median(np.array(kt[1:]), axis=0)
        print("   kernels: first pass %.1f us (k-NN %.1f + widening %.1f + fit %.1f in the last repetition), second %.1f us, third %.1f us;  candidates per query: %s" % (
            kt[0], parts[0], parts[1], parts[2], kt[1], kt[2], cands(ctx)), flush=True)
        print("FINE=%s %s: map %d  first pass %.1f us, second %.1f us, third %.1f us  M %d  fine %s stragglers %d" % (
            fine, tag, ctx.map_size(), t[0], t[1], t[2], r3[2], ctx.fine_stats(), ctx.last_stragglers()), flush=True)
        return r3
    timed("primed")
    t0 = time.perf_counter()
    for sw in sweeps:
        ctx.scan_set(sw); ctx.map_add_scan(x, 0.0)
    print("  %d sweeps inserted, %.2f ms each" % (NINS, (time.perf_counter() - t0) / NINS * 1e3))
    r = timed("after %d raw sweeps" % NINS)
    if ref is None: ref = r
    else: print("  same sums as the first configuration:", np.array_equal(ref[0], r[0]), ref[2] == r[2])
    ctx.close()
