"""Developer soak (GPU box): differential fuzz of the measurement pass against the oracle.  Random scenes (box-world + tilted clutter),
Velodyne-like or random scans, maps crowded by raw sweeps inserted through the product AND the oracle's octree (same world points),
poor / good priors, gates, caps, extrinsics; second level and own-cell probe forced on by low thresholds in half of the trials.
Per trial three passes (no bound, pruned, pruned after a small move): same M, the oracle's H rows and residuals bit for bit.
usage: TRIALS=200 SEED=1 python tests/dev/gpu_fuzz.py      (START=<trial>: the trials before it only draw their random numbers)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from fast_limo_amd import synth, _lib
import oracle_py as oracle
TRIALS = int(os.environ.get("TRIALS", 100)); SEED = int(os.environ.get("SEED", 1)); START = int(os.environ.get("START", 0))
rs = np.random.RandomState(SEED)
t_start = time.time()
stats = dict(M=0, fine=0, ties=0, widened=0, crowded=0)
for trial in range(TRIALS):
    L = float(rs.choice([10.0, 25.0, 50.0]))
    n_map = int(rs.choice([30000, 100000, 300000]))
    live = trial >= START
    mp = synth.box_world_map(n_map if live else 16, L, 1000 * SEED + trial)
    if trial % 3 == 1:
        parts = [mp]
        for k in range(4):
            nrm = rs.normal(size=3); nrm /= np.linalg.norm(nrm)
            u = np.cross(nrm, [0.2, 0.9, 0.4]); u /= np.linalg.norm(u); v = np.cross(nrm, u)
            c0 = rs.uniform(-0.6 * L, 0.6 * L, 3); c0[2] = rs.uniform(0, 6)
            parts.append((c0 + rs.uniform(-4, 4, (4000, 1)) * u + rs.uniform(-4, 4, (4000, 1)) * v + rs.normal(0, 0.01, (4000, 1)) * nrm).astype(np.float32))
        mp = np.concatenate(parts)
    velo = bool(rs.randint(0, 2))
    if velo:
        rings, az = int(rs.choice([16, 32, 64])), int(rs.choice([128, 256, 512]))
        scan = np.ascontiguousarray(synth.velodyne_scan(rings, az, L, 7000 + trial)[:, :3]) if live else None
    else:
        nscan = int(rs.choice([500, 3000, 9000]))
        scan = np.ascontiguousarray(synth.box_world_scan_random(nscan, L, 7000 + trial)[:, :3]) if live else None
    force = bool(trial % 2)
    os.environ["FLIMO_FINE_THRESHOLD"] = "12" if force else "64"
    os.environ["FLIMO_FINE_MIN_POINTS"] = "0" if force else "32768"
    os.environ["FLIMO_PROBE"] = str(int(rs.choice([24, 96]))) if force else "96"
    mdp = float(rs.choice([2.0, 1.0])); pth = float(rs.choice([0.05, 0.02, 0.1])); est = int(rs.randint(0, 2))
    xt = oracle.identity_x26()
    xt[0:3] = synth.T_STAR_T
    r, p_, y = [np.deg2rad(v) for v in synth.T_STAR_RPY_DEG]
    cr, sr, cp, sp, cy, sy = np.cos(r / 2), np.sin(r / 2), np.cos(p_ / 2), np.sin(p_ / 2), np.cos(y / 2), np.sin(y / 2)
    xt[3:7] = [sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy]
    n_ins = int(rs.choice([0, 0, 3, 8]))
    if not live:                                                     # (the same draws as a live trial, nothing else)
        if rs.randint(0, 2):
            rs.normal(0, 0.2, 3); rs.normal(0, 0.01, 3)
        rs.normal(0, 0.01, 3); rs.normal(0, 0.002, 3)
        continue
    ctx = _lib.HipCtx(0)
    oc = oracle.Octree(); oc.update(mp)
    ctx.map_config(); ctx.map_add(mp)
    for j in range(n_ins):                                           # raw sweeps at the true pose crowd the cells near the sensor
        sw = np.ascontiguousarray(synth.velodyne_scan(32, 512, L, 9000 + 10 * trial + j)[:, :3])
        ctx.scan_set(sw); oc.update(ctx.scan_to_world(xt)); ctx.map_add_scan(xt, 0.1 * (j + 1))
    assert ctx.map_size() == oc.size(), (trial, ctx.map_size(), oc.size())
    stats["crowded"] += int(n_ins > 0)
    x = xt.copy()
    if rs.randint(0, 2):                                             # poor prior: many stragglers in the first pass
        x[0:3] += rs.normal(0, 0.2, 3)
        q = x[3:7] + np.concatenate([rs.normal(0, 0.01, 3), [0.0]]); x[3:7] = q / np.linalg.norm(q)
    x2 = x.copy(); x2[0:3] += rs.normal(0, 0.01, 3)
    x3 = x2.copy(); x3[0:3] += rs.normal(0, 0.002, 3)
    ocfg = oracle.default_cfg(num_threads=8, MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7, MAX_DIST_PLANE=mdp, PLANE_THRESHOLD=pth, estimate_extrinsics=est)
    gcfg = _lib.default_match_cfg(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7, MAX_DIST_PLANE=mdp, PLANE_THRESHOLD=pth, estimate_extrinsics=est)
    ctx.scan_set(scan)
    ctx.set_debug_records(bool(trial % 4 < 2))
    tag0 = f"trial {trial} seed {SEED}: L={L} map={ctx.map_size()} scan={scan.shape[0]} velo={velo} ins={n_ins} force={force} mdp={mdp} pth={pth} est={est}"
    for k, xk in enumerate((x, x2, x3)):
        _, H, h, _ = oracle.match_H(oc, ocfg, xk, scan)
        HTH, HTh, M = ctx.match_reduce(xk, gcfg)
        tag = tag0 + f" pass {k}"
        assert M == H.shape[0], (tag, M, H.shape[0])
        Hd, hd = ctx.match_fetch_H()
        assert np.array_equal(Hd, H), tag
        assert np.array_equal(hd, h), tag
        np.testing.assert_allclose(HTH, H.T @ H if M else np.zeros((12, 12)), rtol=1e-11, atol=1e-9, err_msg=tag)
        stats["M"] += M
    fs = ctx.fine_stats(); stats["fine"] += int(fs["passes"] > 0); stats["ties"] += ctx.tie_stats()["queries_settled"]
    mm, _, _ = ctx.grid_selfcheck(); assert mm == 0, tag0
    ctx.close()
    if trial % 10 == 9:
        print("trial %d ok (%.0f s)  %s" % (trial, time.time() - t_start, stats), flush=True)
print("FUZZ OK: %d trials, %s" % (TRIALS, stats))
