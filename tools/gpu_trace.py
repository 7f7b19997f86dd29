"""Developer tool (GPU box): in-kernel phase stamps of knn5_kernel / fit_kernel from the -DFLIMO_TRACE build.
    make -C fast_limo_amd/csrc trace;  FLIMO_HIP_LIB=fast_limo_amd/libflimo_hip_trace.so python tools/gpu_trace.py
Prints, per phase boundary, when (relative to the first block's start, in us) the median / first / last block passed it."""
import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("FLIMO_HIP_LIB", os.path.join(ROOT, "fast_limo_amd", "libflimo_hip_trace.so"))
sys.path.insert(0, ROOT)
import numpy as np
from fast_limo_amd import synth, _lib

NMAP = int(os.environ.get("NMAP", 1000000)); L = float(os.environ.get("LBOX", 100.0))
mp = synth.box_world_map(NMAP, L, 1)
scan = np.ascontiguousarray(synth.velodyne_scan(64, 1024, L, 2)[:, :3])
ctx = _lib.HipCtx(0)
ctx.map_config(cell_size=0.5)
ctx.map_add(mp)
ctx.scan_set(scan)
x0 = np.zeros(26); x0[6] = 1.0; x0[10] = 1.0; x0[25] = -9.809
if os.environ.get("X0") == "tstar":        # the converged pose: few stragglers, like passes 2..4 of a registration
    x0[0:3] = synth.T_STAR_T
    r, p_, y = [np.deg2rad(v) for v in synth.T_STAR_RPY_DEG]
    cr, sr, cp, sp, cy, sy = np.cos(r / 2), np.sin(r / 2), np.cos(p_ / 2), np.sin(p_ / 2), np.cos(y / 2), np.sin(y / 2)
    x0[3:7] = [sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy]
mcfg = _lib.default_match_cfg(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
lib = _lib.load_hip()
lib.flimo_trace_read.argtypes = [C.c_int, C.c_void_p, C.c_size_t]
# NINS=<n>: n raw sweeps inserted at the pose first (a map crowded under the sensor); FIRSTPASS=1: every traced pass is the
# first pass of a scan (no bound from a previous pass)
for j in range(int(os.environ.get("NINS", 0))):
    ctx.scan_set(np.ascontiguousarray(synth.velodyne_scan(64, 1024, L, 100 + j)[:, :3]))
    ctx.map_add_scan(x0, 0.0)
if int(os.environ.get("NINS", 0)):
    scan = np.ascontiguousarray(synth.velodyne_scan(64, 1024, L, 999)[:, :3])
    ctx.scan_set(scan)
for it in range(6):
    if os.environ.get("FIRSTPASS") == "1":
        ctx.scan_set(scan)
    ctx.match_reduce(x0, mcfg)
names = {0: ["start", "query loaded", "row bounds loaded", "candidates done", "merged", "stored (tail done)"],
         1: ["start", "scan+nbr loaded / fused: fit starts", "5 points gathered", "row computed", "partial stored", "ticket taken",
             "LAST: partials summed", "LAST: published"]}
nblk = {0: (scan.shape[0] * 2 + 255) // 256,
        1: (scan.shape[0] + 255) // 256}
for k in (0, 1):
    nb = nblk[k]
    buf = np.zeros(nb * 8, np.uint64)
    rc = lib.flimo_trace_read(k, buf.ctypes.data, buf.size)
    assert rc == 0, rc
    t = buf.reshape(nb, 8).astype(np.int64)
    if k == 0:
        t0_knn = t[:, 0].min()
    fused = k == 1 and os.environ.get("FLIMO_FUSE", "1") != "0"
    if fused:       # the fit / reduction stamps were written by the blocks of the fused k-NN launch: same block count, same time base
        nb = nblk[0]
        buf = np.zeros(nb * 8, np.uint64)
        assert lib.flimo_trace_read(1, buf.ctypes.data, buf.size) == 0
        t = buf.reshape(nb, 8).astype(np.int64)
        t[:, 0] = 0; t[:, 2] = 0
    t0 = t0_knn if fused else t[:, 0].min()
    print("kernel", "knn5" if k == 0 else "fit", "blocks", nb, " (100 MHz clock: 0.01 us resolution)")
    for s, nm in enumerate(names[k]):
        col = t[:, s]
        ok = col > t0 - 10**9 if fused else col > 0
        ok = ok & (col > 0)
        if not ok.any():
            continue
        v = (col[ok] - t0) / 100.0
        print(f"  {nm:24s} n={int(ok.sum()):5d}  first {v.min():7.2f}  median {np.median(v):7.2f}  p90 {np.percentile(v, 90):7.2f}  last {v.max():7.2f} us")
    if k == 0:
        dur = (t[:, 5] - t[:, 0]) / 100.0
        cph = (t[:, 3] - t[:, 2]) / 100.0
        tot = t[:, 6] // 6          # accumulated over the 6 identical passes
        order = np.argsort(-dur)
        print("  slowest blocks: (block, xcd=b%8, dur us, cand-phase us, block candidates, smid)")
        for b in order[:12]:
            print(f"    {b:5d} {b % 8} {dur[b]:6.2f} {cph[b]:6.2f} {int(tot[b]):6d} {int(t[b, 7]):#x}")
        print("  fastest nonzero:")
        for b in order[-6:]:
            print(f"    {b:5d} {b % 8} {dur[b]:6.2f} {cph[b]:6.2f} {int(tot[b]):6d} {int(t[b, 7]):#x}")
        info = t[:, 7]
        has = (info >> 40) & 1 == 1
        if has.any():
            d = (info[has] & 0xffff) / 100.0; F = (info[has] >> 16) & 0xff; it = (info[has] >> 24) & 0xff
            print("  tail: blocks with stragglers %d; duration us median %.2f p90 %.2f max %.2f; F max %d; ring iterations max %d (mean %.2f)"
                  % (int(has.sum()), np.median(d), np.percentile(d, 90), d.max(), int(F.max()), int(it.max()), it.mean()))
            for Fv in sorted(set(F.tolist()))[:6]:
                m = F == Fv
                print("    F=%d: n=%d median %.2f us, iterations mean %.2f" % (Fv, int(m.sum()), np.median(d[m]), it[m].mean()))
        ok = tot > 0
        print("  corr(dur, block candidates) =", np.corrcoef(dur[ok], tot[ok])[0, 1], " block candidates: median", np.median(tot[ok]), "max", tot.max())
        end = (t[:, 5] - t0) / 100.0
        for x in range(8):
            m = (np.arange(nb) % 8) == x
            print(f"  xcd {x}: median end {np.median(end[m]):6.2f}  last end {end[m].max():6.2f}  sum cand {int(tot[m].sum())}")
# LATE=<n>: the n workgroups of the one-launch pass that took their group ticket last -- every stamp of each (what made them late?)
if int(os.environ.get("LATE", 0)) and os.environ.get("FLIMO_FUSE", "1") != "0":
    nb = nblk[0]
    b0 = np.zeros(nb * 8, np.uint64); b1 = np.zeros(nb * 8, np.uint64)
    assert lib.flimo_trace_read(0, b0.ctypes.data, b0.size) == 0 and lib.flimo_trace_read(1, b1.ctypes.data, b1.size) == 0
    a = b0.reshape(nb, 8).astype(np.int64); f = b1.reshape(nb, 8).astype(np.int64)
    t0 = a[:, 0].min()
    rel = lambda v: (v - t0) / 100.0
    order = np.argsort(-f[:, 5])
    print("latest group tickets: block, xcd | start, query, bounds, candidates, merged, stored(tail) | fit starts, row computed, partial stored, ticket  [us]")
    for b in order[:int(os.environ["LATE"])]:
        print("  %4d %d | %5.2f %5.2f %5.2f %5.2f %5.2f %5.2f | %5.2f %5.2f %5.2f %5.2f" % (b, b % 8, rel(a[b, 0]), rel(a[b, 1]), rel(a[b, 2]), rel(a[b, 3]), rel(a[b, 4]),
              rel(a[b, 5]), rel(f[b, 1]), rel(f[b, 3]), rel(f[b, 4]), rel(f[b, 5])))
    med = lambda col: float(np.median(rel(col[col > 0])))
    print("  median      | %5.2f %5.2f %5.2f %5.2f %5.2f %5.2f | %5.2f %5.2f %5.2f %5.2f" % (med(a[:, 0]), med(a[:, 1]), med(a[:, 2]), med(a[:, 3]), med(a[:, 4]), med(a[:, 5]),
          med(f[:, 1]), med(f[:, 3]), med(f[:, 4]), med(f[:, 5])))
ctx.close()
