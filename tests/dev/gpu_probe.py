"""Developer probe (runs on the GPU box through gpurun): parity of the HIP path against the CPU
oracle on cfg-2-like inputs and a sweep of kernel variants.  Not part of the product."""
import sys, os, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import oracle_py as O
from fast_limo_amd import synth, _lib

NMAP = int(os.environ.get("NMAP", 1000000)); L = float(os.environ.get("LBOX", 100.0))
RINGS = int(os.environ.get("RINGS", 64)); AZ = int(os.environ.get("AZ", 1024))
out = {}
t0 = time.time()
mp = synth.box_world_map(NMAP, L, 1)
scan5 = synth.velodyne_scan(RINGS, AZ, L, 2)
scan = np.ascontiguousarray(scan5[:, :3])
print("gen", time.time() - t0, "s; scan", scan.shape, "map", mp.shape, flush=True)

ctx = _lib.HipCtx(0)
ctx.map_config(cell_size=float(os.environ.get("CELL", 0.5)))
t0 = time.time(); ctx.map_add(mp); print("gpu map_add", time.time() - t0, flush=True)
t0 = time.time(); ctx.map_add(mp[:0]); 
oc = O.Octree(); t0 = time.time(); oc.update(mp); print("oracle octree build", time.time() - t0, flush=True)
dev_pts = ctx.map_points()
assert dev_pts.shape[0] == NMAP
assert np.array_equal(np.sort(dev_pts.view([('x','f4'),('y','f4'),('z','f4')]).ravel(), order=('x','y','z')),
                      np.sort(mp.view([('x','f4'),('y','f4'),('z','f4')]).ravel(), order=('x','y','z')))

# ---- kNN parity (a7) ----
x0 = O.identity_x26()
RT = O.pose_mats(x0)[0]
rs = np.random.RandomState(5)
q = (scan[rs.choice(scan.shape[0], 20000, replace=False)]).copy()
q[:100] += 1000.0   # far outside the map
q[100:200] = rs.uniform(-L, L, (100, 3)).astype(np.float32)  # in the air
t0 = time.time(); idx, sqd, cnt = ctx.knn(q, 5); t_g = time.time() - t0
t0 = time.time(); onbr, osqd, ocnt, ev = oc.knn(q, 5, 1); t_o = time.time() - t0
print("knn gpu %.3fs oracle %.3fs evals/q %.1f" % (t_g, t_o, ev / q.shape[0]), flush=True)
assert np.all(cnt == 5), (cnt.min(), cnt.max())
same_d = np.array_equal(sqd, osqd)
nb = dev_pts[idx]
same_p = np.all(nb == onbr, axis=(1, 2))
ties = (~same_p).sum()
print("knn: sqd bit-exact:", same_d, " neighbour-coordinate mismatches (ties):", int(ties), flush=True)
if ties:
    bad = np.where(~same_p)[0][:5]
    for b in bad: print("  q", b, sqd[b], osqd[b])
out["knn_sqd_bitexact"] = bool(same_d); out["knn_nbr_mismatch"] = int(ties)

# ---- match parity (a5-a10) ----
ocfg = O.default_cfg(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7, num_threads=1)
t0 = time.time(); recs, H, h, ev = O.match_H(oc, ocfg, x0, scan); t_or = time.time() - t0
E = ev / scan.shape[0]
print("oracle match_H %.3fs  M=%d  E=%.2f evals/query" % (t_or, H.shape[0], E), flush=True)
mcfg = _lib.default_match_cfg(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
ctx.scan_set(scan)
ctx.set_debug_records(True); ctx.set_timing(True)
HTH, HTh, M = ctx.match_reduce(x0, mcfg)
g = ctx.match_fetch()
print("gpu M=%d cand/query=%.2f" % (M, ctx.last_candidates_per_query()), flush=True)
valid_g = g["valid"] > 0; valid_o = recs["is_plane"] > 0
print("valid mismatch:", int((valid_g != valid_o).sum()))
both = valid_g & valid_o
print("n bit-exact:", np.array_equal(g["n"][both], recs["n"][both]), " dist bit-exact:", np.array_equal(-g["h"][both], recs["dist"][both]))
print("sqd bit-exact:", np.array_equal(g["sqd"][both], recs["sqd"][both]))
Hg = g["H"][valid_g].astype(np.float64)
if valid_g.sum() == H.shape[0]:
    print("H rows bit-exact:", np.array_equal(Hg, H), " max|dH|", np.abs(Hg - H).max())
HTH_o = H.T @ H; HTh_o = H.T @ h
print("HTH rel err:", np.abs(HTH - HTH_o).max() / np.abs(HTH_o).max(), " HTh rel err:", np.abs(HTh - HTh_o).max() / np.abs(HTh_o).max(), " M:", M, H.shape[0])
out.update(valid_mismatch=int((valid_g != valid_o).sum()), M_gpu=int(M), M_oracle=int(H.shape[0]), E=E)

# ---- variant sweep ----
ctx.set_debug_records(False)
res = []
for cell in [float(c) for c in os.environ.get("CELLS", "0.5,0.6").split(",")]:
    ctx.map_clear(); ctx.map_config(cell_size=cell); ctx.map_add(mp)
    ctx.set_debug_records(True); ctx.match_reduce(x0, mcfg); cq = ctx.last_candidates_per_query(); ctx.set_debug_records(False)
    for lpq in [2]:    # two lanes per query is the only layout since round 5
        for _ in range(3): ctx.match_reduce(x0, mcfg)
        ms = []; ws_ = []; rs_ = []
        t0 = time.time()
        for _ in range(20):
            _, _, M2 = ctx.match_reduce(x0, mcfg); a, wd, b = ctx.last_kernel_ms(); ms.append(a); ws_.append(wd); rs_.append(b)
        wall = (time.time() - t0) / 20
        assert M2 == M, (M2, M)
        r = dict(cell=cell, lpq=lpq, knn_us=float(np.median(ms) * 1e3), widen_us=float(np.median(ws_) * 1e3), widen_n=ctx.last_widen_count(), fit_us=float(np.median(rs_) * 1e3), wall_us=wall * 1e6, cand_per_q=cq)
        res.append(r); print(r, flush=True)
out["sweep"] = res
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "probe.json"), "w"), indent=1)
ctx.close()
print("PROBE DONE")
