#!/usr/bin/env python3
"""bench.py -- headline benchmark of the MI355X-native fast_LIMO registration hot path.

A "step" is ONE scan registration of BASELINE.json's config 2: GPU deskew of a resident 64k-point
Velodyne-like scan + the iterated ESKF update (<= MAX_NUM_ITERS+1 passes of
k-NN -> plane fit -> point-to-plane residual/Jacobian -> H^T H reduction on the GPU, 23x23 solve on
the host) against a resident 1M-point map.  Inputs are resident in HBM before the timed region
(host filters / time sort / PCIe upload and the map insert are outside it: SURVEY.md section 8 rows
f-1/f-2).  Every step restarts from the same predicted prior, so all steps do the
same work and produce the same pose.

  python bench.py --gpus N --steps K --warmup W
For N > 1 the driver launches it under torch.distributed.run (one rank per GPU); a plain `python bench.py --gpus N` starts
the N ranks itself (self_launch).  The path shards
by independent scan streams (config 5): no data-path collective, weak scaling; torch.distributed is
used only for the barrier and the max-over-ranks time.

Rank 0 prints ONE JSON line (see the contract in the task description) with these extra objects:
  roofline      dominant kernel = the one-launch measurement pass (k-NN fast path + in-kernel widening + plane fit +
                residual / Jacobian + H^T H reduction, `knn5_kernel<2, 8, true, false>`): ALGORITHMIC bytes per launch / its mean
                launch duration from HIP events attached to the dispatch during the timed region, vs 8 TB/s HBM; `stage`
                gives every kind of pass (the first pass of the benchmark's poor prior runs k-NN / widening / fit as
                separate dispatches); `traffic` = HBM-side bytes from the committed PMC profile of the same sources
  end_to_end    the same scan from a host cloud: filters + time sort + upload + deskew + update + map insert
  cpu_baseline  the CPU oracle (a restatement of the reference algorithm, "port") timed on this box's
                host cores on the same scan / map, N = 1 only
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBPS = 8000.0          # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md
E_FALLBACK = 40.37              # oracle leaf-point distance evaluations per query on cfg 2, mean over the 4 passes of the
                                # benchmark registration (what cpu_baseline measures at N = 1; used when it does not run: N > 1, --no-cpu-baseline)
NBR_BYTES = 32                  # bytes the k-NN kernel writes per query (5 indices + flag, padded)


def pmc_traffic(queries_per_launch):
    """HBM-side bytes per launch of the dominant kernel from the committed PMC profile (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    in separate passes of this same command, FETCH_SIZE doubled per MI355X_MICROARCH.md).  PMC counters cannot be collected
    from inside the run; the profile carries the hash of the kernel sources it was taken with and is only reported when that is
    the hash of the sources in the tree (and the launch shape is the benchmark's)."""
    f = os.path.join(ROOT, "profiles", "r06", "pmc_fetch_write_per_kernel.json")
    try:
        from fast_limo_amd import build as b
        d = json.load(open(f))
        if d.get("sources_hash") != b.sources_hash() or abs(queries_per_launch - 65536) > 0.5:
            return None
        return d["dominant_kernel_traffic_bytes_per_launch"]["total_corrected"]
    except Exception:
        return None


def cpu_info():
    model, phys = None, None
    try:
        cores = set()
        phys_id = core_id = None
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name") and model is None:
                model = line.split(":", 1)[1].strip()
            elif line.startswith("physical id"):
                phys_id = line.split(":", 1)[1].strip()
            elif line.startswith("core id"):
                core_id = line.split(":", 1)[1].strip()
                cores.add((phys_id, core_id))
        phys = len(cores) or None
    except OSError:
        pass
    return model, phys, os.cpu_count()


def workload(rank: int, rings: int, az: int, nmap: int, L: float):
    from fast_limo_amd import synth
    mp = synth.box_world_map(nmap, L, 1)
    scan_seed = 2 if rank == 0 else 10 + rank        # cfg 2 on rank 0, cfg 5 seeds on the others
    scan = synth.velodyne_scan(rings, az, L, scan_seed)
    imu = synth.stationary_imu(0.0, 6.6)            # 0.35 s are used by the registration, the rest by the end-to-end sweeps
    return mp, scan, imu


def drive_to_prior(loc, mp, scan, imu):
    """Prime the map, run the null first scan (reference a-note 8) and stop right before scan 2."""
    st, w, a = imu
    loc.map_add(mp)
    i = 0
    while i < len(st) and st[i] <= 0.105:
        loc.update_imu(st[i], w[i], a[i]); i += 1
    rc1 = loc.update_pointcloud(scan, 0.0)
    while i < len(st) and st[i] <= 0.205:
        loc.update_imu(st[i], w[i], a[i]); i += 1
    loc._imu_cursor = i
    return rc1


def cpu_baseline(args):
    """The CPU oracle's baseline in a child process: OpenMP threads pinned (OMP_PROC_BIND=close, OMP_PLACES=cores -- in the child's
    environment only: a pinned OpenMP runtime in THIS process would bind the main thread, and with it every helper thread the
    product starts later, to one core), thread counts swept up to the CPUs the process may run on."""
    import subprocess
    env = dict(os.environ)
    env.setdefault("OMP_PROC_BIND", "close")
    env.setdefault("OMP_PLACES", "cores")
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-child", "--rings", str(args.rings), "--azimuths", str(args.azimuths),
           "--map-points", str(args.map_points), "--box", str(args.box)]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=900)
    for line in r.stdout.decode(errors="replace").splitlines():
        if line.startswith("CPU_BASELINE_JSON "):
            d = json.loads(line[len("CPU_BASELINE_JSON "):])
            return d["cb"], d["E"], np.array(d["x_o"])
    raise RuntimeError("the CPU baseline child did not deliver (rc %d)" % r.returncode)


def cpu_baseline_here(mp, scan, imu, caps, max_threads, ncpu=None):
    """Oracle Localizer timed on the host: deskew + iterated update of the same scan (no map insert).  3 warm-ups, then up to 20
    registrations per thread count within a time budget (about 20 s in total)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import oracle_py as O
    best = None
    E = None
    x_o = None
    per_threads = {}
    model, phys, logical = cpu_info()
    # 1 thread and every power of two up to the machine's hardware threads (Config.num_threads is what the wrapper hands to
    # omp_set_num_threads, Localizer.cpp:46-50): the best is shown, all are reported (`by_threads`); about 25 s in total
    # (the CPUs this process may run on, not the machine's: a container or a cpuset-limited box oversubscribes otherwise --
    #  round 5's sweep collapsed at 64 / 128 threads; OMP_PROC_BIND / OMP_PLACES come with the child's environment: cpu_baseline)
    if ncpu is None:
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    tried = sorted(set([1] + [t for t in (8, 16, 32, 64, 128) if t <= min(ncpu, max_threads)]))
    for nt in tried:
        L = O.Localizer(O.default_cfg(num_threads=nt, **caps))
        drive_to_prior(_OracleNoInsert(L), mp, scan, imu)
        x_prior, P_prior = L.get_x(), L.get_P()
        times = []
        stages = []
        budget_t0 = time.time()
        budget = 4.0 if nt == 1 else 3.5
        for rep in range(23):
            L.set_x(x_prior); L.set_P(P_prior)
            rc = L.update_pointcloud(scan, 0.1, add_to_map=False)
            st = L.stats()
            if rep == 0:
                x_o = L.get_x()
                E = st["evals"] / max(st["queries"], 1)
            if rep >= 3 or (nt == 1 and rep >= 1):      # warm-ups dropped (one is enough for the slow single-thread run)
                times.append(st["t_deskew"] - st["t_sort"] + st["t_update"])   # the time sort is outside the GPU step too
                stages.append((st["t_deskew"] - st["t_sort"], st["t_match"], st["t_hrows"], st["t_update"] - st["t_match"] - st["t_hrows"]))
            if time.time() - budget_t0 > budget and len(times) >= 3:
                break
        t = float(np.median(times))
        per_threads[str(nt)] = {"scans_per_s": 1.0 / t, "ms": 1e3 * t, "reps": len(times)}
        # the path exit once (transform + Mapper::add of the registered scan into the map): reported as a stage, not part of `value`
        L.set_x(x_prior); L.set_P(P_prior)
        L.update_pointcloud(scan, 0.1, add_to_map=True)
        t_add = float(L.stats()["t_mapadd"])
        if best is None or t < best[0]:
            best = (t, nt, len(times), [float(v) * 1e3 for v in np.median(np.array(stages), axis=0)] + [t_add * 1e3])
    t, nt, reps, stg = best
    return dict(value=1.0 / t, unit="scans/s", cores=nt, kind="port", cpu_model=model, physical_cores=phys, logical_cpus=logical,
                cpus_allowed=ncpu, omp_proc_bind=os.environ.get("OMP_PROC_BIND"), omp_places=os.environ.get("OMP_PLACES"),
                threads=nt, reps=reps, by_threads=per_threads,
                stages_ms={"deskew": stg[0], "knn_plane_fit": stg[1], "H_rows": stg[2], "HtH_and_solve": stg[3],
                           "map_add_once_not_in_value": stg[4]},
                sample=f"median of {reps} registrations after warm-up (deskew + iterated update, no map insert) of the same "
                       f"{scan.shape[0]}-pt scan vs {mp.shape[0]}-pt map by the CPU oracle (restatement of the "
                       f"reference; the reference itself cannot be built without Eigen/PCL/Boost) on {model}, "
                       f"{phys} physical cores, OpenMP threads tried {tried}, best ({nt}) shown"), E, x_o


class _OracleNoInsert:
    _imu_cursor = 0

    def __init__(self, L):
        self.L = L

    def map_add(self, mp):
        self.L.map_add(mp)

    def update_imu(self, *a):
        self.L.update_imu(*a)

    def update_pointcloud(self, pts, stamp):
        return self.L.update_pointcloud(pts, stamp, add_to_map=False)


HBM_REGIME = dict(rings=128, azimuths=2048, map_points=20000000, box=447.0)      # BASELINE.json configs[3]: 256k-pt scan, 20M-pt map


def pmc_traffic_hbm_regime():
    """HBM-side bytes per launch at 256k x 20M from the committed PMC profile (same rule as pmc_traffic): the one-launch pass and
    the k-NN kernel of the passes that run as separate dispatches."""
    f = os.path.join(ROOT, "profiles", "r06", "pmc_hbm_regime.json")
    try:
        from fast_limo_amd import build as b
        d = json.load(open(f))
        if d.get("sources_hash") != b.sources_hash():
            return None, None
        one = d.get("dominant_kernel_traffic_bytes_per_launch", {}).get("total_corrected")
        knn = d.get("knn5_separate_traffic_bytes_per_launch", {}).get("total_corrected")
        return one, knn
    except Exception:
        return None, None


def hbm_regime_leg(device, steps, with_oracle, max_threads=32):
    """"use config 4 (256k x 20M, 320 MB map) for the real HBM-roofline number" -- the one configuration
    whose map (320 MB of points + two index tables) does not fit the 256 MB Infinity Cache.  Same step as the headline (GPU deskew
    + iterated update of a resident scan against a resident map), its passes timed with HIP events on their dispatches; E from
    the CPU oracle's own traversal on the identical registration."""
    from fast_limo_amd import api, synth
    caps = dict(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
    R = HBM_REGIME
    mp = synth.box_world_map(R["map_points"], R["box"], 1)
    scan = synth.velodyne_scan(R["rings"], R["azimuths"], R["box"], 2)
    imu = synth.stationary_imu(0.0, 0.35)
    loc = api.Localizer(api.default_cfg(gpu_device=device, num_threads=os.cpu_count() or 1,
                                        gpu_cell_size=float(os.environ.get('FLIMO_BENCH_CELL', '0')), **caps))   # 0 = library default
    loc.set_flags(add_to_map=False, download_clouds=False, keep_log=False)
    rc1 = drive_to_prior(loc, mp, scan, imu)
    x_prior, P_prior = loc.get_x(), loc.get_P()
    rc2 = loc.update_pointcloud(scan, 0.1)
    assert rc1 == 1 and rc2 == 0, (rc1, rc2)
    x_ref = loc.get_x()
    reg = loc.register_resident_call(x_prior, P_prior)
    for _ in range(3):
        assert reg() == 0
    x_step = loc.get_x()
    loc.hip.set_timing(0)
    p0, f0 = loc.hip.pass_count(), loc.hip.fused_pass_count()
    t0 = time.perf_counter()
    for _ in range(steps):
        reg()
    elapsed = time.perf_counter() - t0
    n_passes, n_fused = loc.hip.pass_count() - p0, loc.hip.fused_pass_count() - f0
    assert np.allclose(loc.get_x(), x_step, rtol=0.0, atol=1e-9) and np.allclose(x_step, x_ref, rtol=0.0, atol=1e-6)
    # every pass of 8 more steps timed by HIP events on its dispatch (the context's own stream)
    loc.hip.set_timing(1); loc.hip.set_timing_stride(1)
    loc.hip.timing_totals(reset=True); loc.hip.timing_split(reset=True)
    for _ in range(8):
        reg()
    d, tot = loc.hip.timing_split(reset=True), loc.hip.timing_totals(reset=True)
    qpl = tot["queries"] / max(tot["passes"], 1)
    # the k-NN stage on its own (fast path + widening, no fit): the pass split into dispatches (A/B switch)
    loc.hip.set_path_switches(fuse=0)
    for _ in range(2):
        reg()
    loc.hip.timing_split(reset=True)
    for _ in range(8):
        reg()
    ds = loc.hip.timing_split(reset=True)
    loc.hip.set_path_switches(fuse=1)
    loc.hip.set_timing(0)
    stragglers = loc.hip.last_stragglers()
    out = {"workload": "BASELINE.json configs[3] as a resident-input step: %d-pt scan (%d rings x %d azimuths) vs %d-pt box-world map (L = %.0f m), "
                       "k=5, MAX_NUM_ITERS=3, GPU deskew + iterated ESKF update per step" % (scan.shape[0], R["rings"], R["azimuths"], mp.shape[0], R["box"]),
           "map_bytes": int(mp.shape[0]) * 16, "steps": steps, "ms_per_step": 1e3 * elapsed / steps, "scans_per_s": steps / elapsed,
           "passes_per_step": n_passes / steps, "passes_in_one_launch": n_fused, "passes_total": n_passes,
           "stragglers_last_pass": stragglers,
           "one_launch_pass_us": (1e3 * d["fused_ms"] / d["fused_n"]) if d["fused_n"] else None, "one_launch_passes_timed": d["fused_n"],
           "separate_dispatch_pass_us": ({"knn": 1e3 * d["knn_ms"] / d["separate_n"], "widen": 1e3 * d["widen_ms"] / d["separate_n"],
                                          "fit_reduce": 1e3 * d["fit_ms"] / d["separate_n"]} if d["separate_n"] else None),
           "separate_dispatch_passes_timed": d["separate_n"], "queries_per_launch": qpl}
    E = None
    if with_oracle:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_py as O
        nt = max(1, min(max_threads, os.cpu_count() or 1))
        L = O.Localizer(O.default_cfg(num_threads=nt, **caps))
        drive_to_prior(_OracleNoInsert(L), mp, scan, imu)
        rc = L.update_pointcloud(scan, 0.1, add_to_map=False)
        st = L.stats()
        E = st["evals"] / max(st["queries"], 1)
        x_o = L.get_x()
        out["pose_err_vs_cpu"] = {"pos_m": float(np.abs(x_ref[0:3] - x_o[0:3]).max()), "rot_rad": float(2.0 * np.abs(x_ref[3:6] - x_o[3:6]).max()),
                                  "tolerance": 1e-4}
        out["cpu_oracle_one_registration_s"] = float(st["t_deskew"] - st["t_sort"] + st["t_update"])
        out["cpu_threads"] = nt
        del L
    out["E_evals_per_query"] = E
    if E:
        bpq = 16.0 + 16.0 * E + NBR_BYTES
        out["bytes_per_query"] = bpq
        alg = bpq * qpl
        def rate(us):
            return (alg / (us * 1e-6) / 1e9) if us else None
        one = out["one_launch_pass_us"]
        out["achieved"] = rate(one)
        out["frac"] = (rate(one) / HBM_PEAK_GBPS) if one else None
        if ds["separate_n"]:
            knn_us, widen_us = 1e3 * ds["knn_ms"] / ds["separate_n"], 1e3 * ds["widen_ms"] / ds["separate_n"]
            out["knn_stage_separate_dispatches"] = {"knn_us": knn_us, "widen_us": widen_us, "fit_us": 1e3 * ds["fit_ms"] / ds["separate_n"],
                                                    "passes_timed": ds["separate_n"], "stage_us_per_pass": knn_us + widen_us,
                                                    "achieved": rate(knn_us + widen_us), "frac": rate(knn_us + widen_us) / HBM_PEAK_GBPS,
                                                    "knn_kernel_frac": rate(knn_us) / HBM_PEAK_GBPS}
        traffic, traffic_knn = pmc_traffic_hbm_regime()
        out["traffic"] = traffic
        out["traffic_over_algorithmic"] = (traffic / alg) if traffic else None
        out["hbm_utilisation_measured"] = (traffic / (one * 1e-6) / 1e9 / HBM_PEAK_GBPS) if (traffic and one) else None
        if ds["separate_n"] and traffic_knn:
            ks = out["knn_stage_separate_dispatches"]
            ks["knn_kernel_traffic"] = traffic_knn
            ks["knn_kernel_traffic_over_algorithmic"] = traffic_knn / alg
            ks["knn_kernel_hbm_utilisation_measured"] = traffic_knn / (ks["knn_us"] * 1e-6) / 1e9 / HBM_PEAK_GBPS
        out["unit"], out["peak"], out["bound"] = "GB/s", HBM_PEAK_GBPS, "hbm"
    # the path exit at this size (transform + Mapper::add of the registered 256k-point scan into the 20M-point map): the first insert
    # stores the scan's new points, a repeat of the same scan is mostly rejected by the reference's down-sampling rule
    ins = []
    n0 = loc.map_size()
    for k in range(3):
        t1 = time.perf_counter()
        loc.hip.map_add_scan(loc.get_x(), 0.2 + 0.1 * k)
        ins.append(time.perf_counter() - t1)
    ib = loc.hip.map_index_bytes()
    out["index_bytes"] = dict(ib, over_map_bytes=ib["index"] / max(ib["points"], 1),
                              sorted_array_allocated_over_map_bytes=ib["sorted_array_allocated"] / max(ib["points"], 1))
    out["map_insert_ms"] = {"first": 1e3 * ins[0], "repeat": 1e3 * float(np.median(ins[1:])), "points_stored": loc.map_size() - n0,
                            "note": "flimo_map_add_scan of the resident scan, waited for; first: the scan's high points grow the map's box (the grid grows in place, nothing is re-sorted); repeat: the points go into their rows in place"}
    loc.close()
    return out


def crowded_leg(device, rings, az, nmap, box, n_sweeps=50):
    """`roofline.crowded`: the same registration's passes on a map crowded by raw sweeps -- 50 raw sweeps of the workload's size
    inserted at the true pose through the product's insert rule (the first batch that lands in a leaf is kept whole: cells with
    hundreds of points under the sensor).  Kernel time (HIP events on the dispatches) of the first pass (no bound from a previous
    pass) and of the later passes, before and after the inserts; median of 5 registrations each."""
    from fast_limo_amd import synth, _lib
    mp = synth.box_world_map(nmap, box, 1)
    x = np.zeros(26); x[6] = 1; x[10] = 1; x[25] = -9.809
    x[0:3] = synth.T_STAR_T
    r, p_, y = [np.deg2rad(v) for v in synth.T_STAR_RPY_DEG]
    cr, sr, cp, sp, cy, sy = np.cos(r / 2), np.sin(r / 2), np.cos(p_ / 2), np.sin(p_ / 2), np.cos(y / 2), np.sin(y / 2)
    x[3:7] = [sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy]
    query = np.ascontiguousarray(synth.velodyne_scan(rings, az, box, 999)[:, :3])
    cfg = _lib.default_match_cfg(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
    ctx = _lib.HipCtx(device)
    ctx.map_config(); ctx.map_add(mp)

    def passes():
        ctx.set_timing(1)
        rows = []
        for rep in range(6):
            ctx.scan_set(query)
            row = []
            for k in range(3):
                ctx.match_reduce(x, cfg)
                a, w, f = ctx.last_kernel_ms()
                row.append((a + w + f) * 1e3)
            rows.append(row)
        ctx.set_timing(0)
        m = np.median(np.array(rows[1:]), axis=0)
        return float(m[0]), float(0.5 * (m[1] + m[2]))

    first0, later0 = passes()
    sweeps = [np.ascontiguousarray(synth.velodyne_scan(rings, az, box, 100 + j)[:, :3]) for j in range(n_sweeps)]
    t_ins = 0.0
    for sw in sweeps:
        ctx.scan_set(sw)
        t0 = time.perf_counter()
        ctx.map_add_scan(x, 0.0)
        t_ins += time.perf_counter() - t0
    ins_ms = t_ins / n_sweeps * 1e3
    first1, later1 = passes()
    out = {"workload": "%d raw %d-point sweeps inserted at the true pose into the %d-point map, then the same registration" % (n_sweeps, rings * az, nmap),
           "clean_map_us": {"first_pass": first0, "later_passes": later0},
           "crowded_map_us": {"first_pass": first1, "later_passes": later1},
           "first_pass_ratio": first1 / first0, "later_passes_ratio": later1 / later0,
           "map_points_after": ctx.map_size(), "insert_ms_per_sweep": ins_ms,
           "note": "kernel time of all dispatches of a pass (HIP events), median of 5 registrations; results on the crowded map are "
                   "bit-identical to the oracle's (test_crowded_cells_second_level_is_exact)"}
    ctx.close()
    return out


def shipped_config_leg(device, with_oracle, n_sweeps=10, n_pts=120000, tied=False):
    """The reference's shipped configuration (config/kitti.yaml: crop box +-1 m, min distance 4 m, every 4th point, voxel grid 1 m,
    MAX_NUM_PC2MATCH 1e4 / MAX_NUM_MATCHES 5000, the yaml's extrinsics and biases, debug on, clouds handed back) on raw 120k-point
    sweeps of a drive, map inserts on: what a drop-in user behind the ROS wrapper runs.  ms per sweep (call + map insert), the
    CPU oracle driven through the same sweeps beside it."""
    from fast_limo_amd import api, synth
    speed = 10.0
    st, w, a = synth.stationary_imu(0.0, 0.1 * n_sweeps + 0.06)
    lid_t = (8.086759e-01, -3.195559e-01, 7.997231e-01)
    lid_R = (9.999976e-01, -7.854027e-04, 2.024406e-03, 7.553071e-04, 9.998898e-01, 1.482454e-02,
             -2.035826e-03, -1.482298e-02, 9.998881e-01)
    common = dict(MAX_NUM_PC2MATCH=10000, MAX_NUM_MATCHES=5000, voxel_active=1, leaf_size=1.0, crop_active=1,
                  dist_active=1, min_dist=4.0, rate_active=1, rate_value=4, time_offset=1,
                  lidar2baselink_t=lid_t, lidar2baselink_R=lid_R, accel_bias=(0.01, 0.01, 0.01), gyro_bias=(0.01, 0.01, 0.01),
                  cov_gyro=6.01e-4, cov_acc=1.53e-2, cov_bias_gyro=1.54e-5, cov_bias_acc=3.38e-4)
    # tied: the stamps a spinning sensor's driver writes (all rings of a column share one: 1800 columns per sweep) -- what the
    # reference's kitti.yaml configuration is actually fed; otherwise every point its own stamp, in order
    sweeps5 = [synth.corridor_scan(k, n_pts, 4321, speed=speed) for k in range(n_sweeps)]
    if tied:
        sweeps5 = [synth.spinning_stamps(s5, columns=1800) for s5 in sweeps5]
    sweeps = [api.make_points_velodyne(s5) for s5 in sweeps5]

    def drive(L, feed, sync):
        x0 = L.get_x(); x0[14] = speed; L.set_x(x0)
        i = 0
        times = []
        for k in range(n_sweeps):
            until = 0.1 * (k + 1) + 0.005
            while i < len(st) and st[i] <= until:
                L.update_imu(st[i], w[i], a[i]); i += 1
            t1 = time.perf_counter()
            rc = feed(L, k)
            sync(L)
            times.append(time.perf_counter() - t1)
            assert rc == (1 if k == 0 else 0), (k, rc)
        return times

    G = api.Localizer(api.default_cfg(gpu_device=device, cropBoxMin=(-1.0, -1.0, -1.0), cropBoxMax=(1.0, 1.0, 1.0), debug=1,
                                      num_threads=os.cpu_count() or 1, **common))
    G.set_flags(add_to_map=True, download_clouds=True, keep_log=False)

    fresh = [s.copy() for s in sweeps]                                  # (the call filters its input cloud in place, like the reference)

    cloud_buf = np.zeros((n_pts, 3), np.float32)

    def feed_gpu(L, k):
        rc = L.update_pointcloud_points(fresh[k], 0.1 * k)
        L.final_scan(out=cloud_buf)
        return rc
    tg = drive(G, feed_gpu, lambda L: L.sync())
    out = {"workload": "config/kitti.yaml filters / caps / voxel grid / extrinsics, debug on, clouds downloaded: %d raw sweeps of %d points, "
                       "map inserts on; stamps: %s" % (n_sweeps, n_pts, "a spinning sensor's (1800 columns, all rings of a column share one)" if tied
                                                       else "one per point, in order"),
           "sweeps_with_equal_stamps_kept_on_the_device": bool(G.last_sweep_tied()),
           "ms_per_sweep": 1e3 * float(np.median(tg[3:])), "pc2match_points": int(G.pc2match().shape[0]), "map_points": G.map_size(),
           "last_sweep_stages_ms": {kk: 1e3 * float(v) for kk, v in G.stage_times().items()}}
    xg = G.get_x()
    G.close()
    # the same drive fed back to back (one wait for the last insert at the end): the insert that ends a sweep runs beside the
    # caller's IMU propagation and the next sweep's input stage -- the sustained rate, what the sequential CPU figure below is too
    G2 = api.Localizer(api.default_cfg(gpu_device=device, cropBoxMin=(-1.0, -1.0, -1.0), cropBoxMax=(1.0, 1.0, 1.0), debug=1,
                                       num_threads=os.cpu_count() or 1, **common))
    G2.set_flags(add_to_map=True, download_clouds=True, keep_log=False)
    fresh = [s.copy() for s in sweeps]
    x0 = G2.get_x(); x0[14] = speed; G2.set_x(x0)
    i = 0
    for k in range(n_sweeps):
        if k == 3:
            G2.sync()
            t1 = time.perf_counter()
        i1 = int(np.searchsorted(st, 0.1 * (k + 1) + 0.005, side="right"))
        G2.update_imu_n(st[i:i1], w[i:i1], a[i:i1]); i = i1
        feed_gpu(G2, k)
    G2.sync()
    # (`ms_per_sweep` keeps its round-1..3 meaning: the median sweep, each followed by a wait for its map insert; the back-to-back
    #  rate is `ms_per_sweep_sustained`)
    out["ms_per_sweep_each_waited_for"] = out["ms_per_sweep"]
    out["ms_per_sweep_sustained"] = 1e3 * (time.perf_counter() - t1) / (n_sweeps - 3)
    assert np.array_equal(G2.get_x(), xg)                               # the same drive
    G2.close()
    if with_oracle:
        sys.path.insert(0, os.path.join(ROOT, "oracle"))
        import oracle_py as O
        nt = max(1, min(32, os.cpu_count() or 1))
        Lo = O.Localizer(O.default_cfg(crop_min=(-1.0, -1.0, -1.0), crop_max=(1.0, 1.0, 1.0), num_threads=nt, **common))
        fresh_o = [s.copy() for s in sweeps]
        to = drive(Lo, lambda L, k: L.update_pointcloud_points(fresh_o[k], 0.1 * k), lambda L: None)
        out["cpu_oracle_ms_per_sweep"] = 1e3 * float(np.median(to[3:]))          # the same statistic on both sides: median sweep
        out["cpu_oracle_ms_per_sweep_mean"] = 1e3 * float(np.mean(to[3:]))
        out["cpu_threads"] = nt
        out["speedup_vs_cpu_oracle"] = out["cpu_oracle_ms_per_sweep"] / out["ms_per_sweep"]
        # (like for like is the line above: median sweep, each waited for, on both sides.  This one is the GPU's back-to-back
        #  THROUGHPUT against the CPU's sequential mean LATENCY -- two statistics, two execution modes: context, not a speed-up)
        out["gpu_sustained_throughput_vs_cpu_mean_latency"] = out["cpu_oracle_ms_per_sweep_mean"] / out["ms_per_sweep_sustained"]
        xo = Lo.get_x()
        out["free_running_pose_difference_m"] = float(np.abs(xg[0:3] - xo[0:3]).max())
    return out


def rank_barrier(dist, torch):
    """Barrier over the ranks (if any) + device synchronisation: brackets the timed region on both sides."""
    if dist is not None:
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


def aggregate(dist, torch, elapsed: float, world: int, steps: int):
    """(max over ranks of the timed region [s], whole-job scans/s): every rank registered `steps` scans of its own stream."""
    if dist is not None:
        dev = "cuda" if (torch.cuda.is_available() and dist.get_backend() == "nccl") else "cpu"
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    return elapsed, world * steps / elapsed


def visible_gpus() -> int:
    """GPUs this process would see, without loading the HIP runtime (the parent of `--gpus N` never does): the visibility
    variables when set, else the KFD topology (a node with SIMDs is a GPU)."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([x for x in v.split(",") if x.strip() != ""])
    n = 0
    base = "/sys/class/kfd/kfd/topology/nodes"
    try:
        for node in os.listdir(base):
            try:
                props = dict(line.split()[:2] for line in open(os.path.join(base, node, "properties")) if len(line.split()) >= 2)
            except OSError:
                continue
            if int(props.get("simd_count", "0")) > 0:
                n += 1
    except OSError:
        return 0
    return n


def self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes (RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_* in their environment, the protocol torch.distributed.run uses) and wait for them.  This parent never touches the GPU
    -- nothing that initialised HIP is ever re-executed -- and prints nothing itself: rank 0 inherits stdout and writes the one
    JSON line.  On a box with fewer devices than ranks (tests: two ranks on one GPU) the ranks share devices and the barrier goes
    through gloo, RCCL refusing two ranks on one device."""
    import socket
    import subprocess
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env0 = dict(os.environ, WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if "FLIMO_BENCH_BACKEND" not in env0 and visible_gpus() < n:
        env0["FLIMO_BENCH_BACKEND"] = "gloo"
    procs = []
    for r in range(n):
        env = dict(env0, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL))
    rc = 0
    for p in procs:
        try:
            p.wait(timeout=3600)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
        rc = rc or p.returncode
    return 1 if rc else 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rings", type=int, default=64)
    ap.add_argument("--azimuths", type=int, default=1024)
    ap.add_argument("--map-points", type=int, default=1000000)
    ap.add_argument("--box", type=float, default=100.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-hbm-regime", action="store_true", help="skip the 256k x 20M leg (roofline.hbm_regime)")
    ap.add_argument("--no-crowded", action="store_true", help="skip the crowded-map leg (roofline.crowded)")
    ap.add_argument("--hbm-regime-only", action="store_true", help="run only the 256k x 20M leg and print it (profiling)")
    ap.add_argument("--hbm-steps", type=int, default=20)
    ap.add_argument("--e2e-sweeps", type=int, default=8)
    ap.add_argument("--streams", type=int, default=3, help="informational leg: this many independent scan streams on ONE GPU at once (0: skip)")
    ap.add_argument("--with-insert", action="store_true",
                    help="time the step WITH the path exit (transform + map insert) over six steps instead of the default two "
                         "(first insertion + one repeat), for a steadier 'repeat' figure")
    ap.add_argument("--cpu-baseline-child", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.cpu_baseline_child:
        # the CPU oracle's sweep over thread counts in a process of its own (cpu_baseline): OpenMP pinned by the environment the
        # parent set, no GPU runtime loaded, nothing of the product in the address space
        allowed = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)      # (before libgomp binds this thread)
        caps = dict(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
        mp, scan, imu = workload(0, args.rings, args.azimuths, args.map_points, args.box)
        cb, E, x_o = cpu_baseline_here(mp, scan, imu, caps, max_threads=128, ncpu=allowed)
        print("CPU_BASELINE_JSON " + json.dumps({"cb": cb, "E": E, "x_o": [float(v) for v in x_o]}))
        return

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args.gpus))            # plain `python bench.py --gpus N`: this process only starts the N ranks

    # stdout carries exactly ONE JSON line: the native library reports status lines the way the reference does
    # (std::cout), so fd 1 is pointed at stderr for the whole run and the result goes to the saved descriptor
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    if args.hbm_regime_only:
        out = hbm_regime_leg(0, args.hbm_steps, with_oracle=not args.no_cpu_baseline)
        os.write(json_fd, (json.dumps({"roofline": {"hbm_regime": out}}) + "\n").encode())
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    n_gpus = world                      # one rank per GPU; WORLD_SIZE is what actually runs
    if args.gpus != world and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: reporting n_gpus={world}", file=sys.stderr)

    import torch
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("FLIMO_BENCH_BACKEND", "nccl")      # "gloo": barrier and max over ranks on CPU tensors
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
        dist_mod.init_process_group(backend=backend, rank=rank, world_size=world)
        dist = dist_mod

    from fast_limo_amd import api
    caps = dict(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
    mp, scan, imu = workload(rank, args.rings, args.azimuths, args.map_points, args.box)
    n_dev = max(1, torch.cuda.device_count())        # one rank per GPU on a node; ranks wrap around on a smaller box (tests)
    loc = api.Localizer(api.default_cfg(gpu_device=local_rank % n_dev, num_threads=os.cpu_count() or 1,
                                        gpu_cell_size=float(os.environ.get('FLIMO_BENCH_CELL', '0')), **caps))   # 0 = library default (0.5 m)
    loc.set_flags(add_to_map=False, download_clouds=False, keep_log=False)
    rc1 = drive_to_prior(loc, mp, scan, imu)
    x_prior, P_prior = loc.get_x(), loc.get_P()
    rc2 = loc.update_pointcloud(scan, 0.1)          # makes the raw scan + IMU frames resident
    assert rc1 == 1 and rc2 == 0, (rc1, rc2)
    x_ref = loc.get_x()

    reg = loc.register_resident_call(x_prior, P_prior)     # arguments and prototype bound once: the loop times the library

    def step():
        rc = reg()
        assert rc == 0, rc

    def barrier():
        rank_barrier(dist, torch)

    step()
    x_step = loc.get_x()               # the state every later step must reproduce
    # level 1: start/stop HIP events attached to the dispatches of a pass (on the context's own stream): the kernels' own begin /
    # end timestamps.  A timed pass costs tens of microseconds of wall time, so inside the timed region the passes are SAMPLED:
    # every (4k+1)-th -- the stride walks through the 4 pass positions of a step evenly -- about 6 samples (every launch of a very
    # short run), which keeps the perturbation of `value` near 1 %.  A dense series (every pass of 12 more steps) follows the
    # timed region for the statistics (`stage.dense_after_timed_region`).  FLIMO_BENCH_TIMING_STRIDE=1 times all.
    loc.hip.set_timing(int(os.environ.get('FLIMO_BENCH_TIMING', '1')))
    auto_stride = max(5, ((4 * args.steps // 6) // 4) * 4 + 1)
    if args.steps < 50:
        auto_stride = 1 << 30           # a short run (the driver's 20 steps last 3 ms) is not sampled at all: three timed passes would
                                        # cost 3 % of it; the dense series below provides the kernel statistics
    loc.hip.set_timing_stride(int(os.environ.get('FLIMO_BENCH_TIMING_STRIDE', str(auto_stride))))
    # (setup: the first barrier of a process initialises torch's own HIP context; it belongs here, ahead of the warm-up, not between
    #  the warm-up and the timed region)
    barrier()
    # W untimed warm-up steps, exactly (the registration above that made the scan resident and `step()` for x_step are setup)
    warm_marks = []
    for _ in range(args.warmup):
        tw0 = time.perf_counter()
        step()
        warm_marks.append(time.perf_counter() - tw0)
    if os.environ.get("FLIMO_BENCH_STEP_TIMES"):
        print("warm-up step times [us]: " + " ".join("%.0f" % (1e6 * d) for d in warm_marks), file=sys.stderr)
    loc.hip.timing_totals(reset=True)
    loc.hip.timing_split(reset=True)
    loc.hip.chain_stats(reset=True)
    pipe0 = loc.hip.pass_pipeline_stats()
    passes0 = loc.hip.pass_count()
    fused0 = loc.hip.fused_pass_count()
    loc.host_profile(reset=True)
    barrier()
    step_marks = [] if os.environ.get("FLIMO_BENCH_STEP_TIMES") else None      # developer: the region's profile, step by step
    import gc
    gc.disable()                       # (a collector pause inside a 3 ms region is a fifth of it: the interpreter's, not the path's.  No
                                       #  gc.collect() here: a full collection right before the region costs its first step 60 us of cold caches)
    t0 = time.perf_counter()
    if step_marks is None:
        for _ in range(args.steps):
            step()
    else:
        for _ in range(args.steps):
            step()
            step_marks.append(time.perf_counter())
    barrier()
    elapsed = time.perf_counter() - t0
    gc.enable()
    if step_marks:
        print("step times [us]: " + " ".join("%.0f" % (1e6 * d) for d in np.diff([t0] + step_marks)), file=sys.stderr)
    tot = loc.hip.timing_totals()
    split = loc.hip.timing_split()
    chain = loc.hip.chain_stats()
    pipe1 = loc.hip.pass_pipeline_stats()
    pipe_found, pipe_wasted = pipe1["published"] - pipe0["published"], pipe1["cancelled"] - pipe0["cancelled"]
    n_passes = loc.hip.pass_count() - passes0
    n_fused_passes = loc.hip.fused_pass_count() - fused0
    hp = loc.host_profile()
    # The spread of the timed region: the same K-step region nine more times, untimed by events (timing level 0), right after `value`'s
    # (the driver's 20 steps last 3 ms; one region is a thin sample)
    value_regions = None
    if rank == 0:
        loc.hip.set_timing(0)
        regs = []
        for _ in range(9):
            tr0 = time.perf_counter()
            for _ in range(args.steps):
                step()
            regs.append(args.steps / (time.perf_counter() - tr0))
        regs = sorted(regs + [args.steps / elapsed])
        value_regions = {"regions": len(regs), "steps_each": args.steps, "scans_per_s_min": regs[0], "scans_per_s_median": float(np.median(regs)),
                         "scans_per_s_max": regs[-1], "note": "`value` is the FIRST region (the one between the barriers); the others follow it back to back"}
        # ... and one region of at least 100 steps beside a short timed one (the driver's 20 steps last 3 ms)
        k100 = max(100, args.steps)
        tr0 = time.perf_counter()
        for _ in range(k100):
            step()
        value_regions["scans_per_s_region_of_%d_steps" % k100] = k100 / (time.perf_counter() - tr0)
        value_regions["long_region_steps"] = k100
        loc.hip.set_timing(int(os.environ.get('FLIMO_BENCH_TIMING', '1')))
    # The update's two layouts on THIS host, the same K steps each (informational; `value` above is the layout the library chose by
    # its launch round-trip measurement): the chain queued at once and the host loop over single passes
    modes = None
    if rank == 0:
        um = loc.hip.update_mode()
        modes = {"chosen": "chain" if um["chained"] else "host_loop", "launch_round_trip_us": um["launch_rtt_us"]}
        for name, mode in (("chain", 2), ("host_loop", 1)):
            loc.hip.set_update_mode(mode)
            for _ in range(3):
                step()
            rates = []
            for _ in range(3):
                tr0 = time.perf_counter()
                for _ in range(args.steps):
                    step()
                rates.append(args.steps / (time.perf_counter() - tr0))
            modes[name + "_scans_per_s"] = float(np.median(rates))
        loc.hip.set_update_mode(0)
        for _ in range(2):
            step()
    dense = None
    kernel_us_per_step = None
    if int(os.environ.get('FLIMO_BENCH_TIMING', '1')) == 1:
        loc.hip.set_timing_stride(1)
        # (the series as rounds 3-5 ran it, each pass's events read right behind it: the host's reading leaves the GPU idle half
        # of the time, and on some boxes its clocks follow -- kept as a second figure, never the one the roofline uses)
        loc.hip.timing_split(reset=True)
        for _ in range(12):
            step()
        d0 = loc.hip.timing_split(reset=True)
        read_behind_us = (1e3 * d0["fused_ms"] / d0["fused_n"]) if d0["fused_n"] else None
        loc.hip.set_timing_deferred(True)          # events read after the series: the passes run back to back as they do untimed
        loc.hip.chain_stats(reset=True)
        for _ in range(12):
            step()
        d = loc.hip.timing_split(reset=True)
        cs = loc.hip.chain_stats(reset=True)
        # GPU time of one step, every launch of it timed by events on its dispatch: passes (all their launches) + the filter's algebra
        kernel_us_per_step = {"passes": 1e3 * (d["fused_ms"] + d["knn_ms"] + d["widen_ms"] + d["fit_ms"]) / 12.0,
                              "filter_algebra": 1e3 * cs["algebra_ms"] / 12.0, "algebra_launches_per_step": cs["algebra_n"] / 12.0,
                              "algebra_launch_us": (1e3 * cs["algebra_ms"] / cs["algebra_n"]) if cs["algebra_n"] else None}
        kernel_us_per_step["total"] = kernel_us_per_step["passes"] + kernel_us_per_step["filter_algebra"]
        dense = {"one_launch_pass_us": (1e3 * d["fused_ms"] / d["fused_n"]) if d["fused_n"] else None, "one_launch_passes_timed": d["fused_n"],
                 "separate_dispatch_pass_us": ({"knn": 1e3 * d["knn_ms"] / d["separate_n"], "widen": 1e3 * d["widen_ms"] / d["separate_n"],
                                                "fit_reduce": 1e3 * d["fit_ms"] / d["separate_n"]} if d["separate_n"] else None),
                 "separate_dispatch_passes_timed": d["separate_n"],
                 "one_launch_pass_us_events_read_behind_each_pass": read_behind_us,
                 "note": "every pass of 12 steps carries its events; they are read after the series (flimo_set_timing_deferred), so the passes run back to back as in the timed region"}
    x_end = loc.get_x()
    # Every step must land on the same state, bit for bit when every step runs the same pass layout.  The full-path registration
    # that made the scan resident (x_ref) may have used another layout for some passes (256k x 20M: its later passes ran as one
    # launch, the steps' as separate dispatches: sums partitioned differently, 1e-13 relative, amplified to 1e-8 in the weakly
    # observable states); the headline workload reproduces x_ref itself bit for bit
    repro_bitwise = bool(np.array_equal(x_end, x_step))
    assert np.allclose(x_end, x_step, rtol=0.0, atol=1e-9), "registration is not reproducible across steps"
    assert np.allclose(x_end, x_ref, rtol=0.0, atol=1e-6), "the resident re-registration left the full path's result"
    headline = (args.rings, args.azimuths, args.map_points, args.box) == (64, 1024, 1000000, 100.0)
    assert (repro_bitwise and np.array_equal(x_end, x_ref)) or not headline, "registration is not bit-reproducible across steps"
    # the k-NN STAGE on its own (fast path + widening, no fit): the same registration with the pass split into separate dispatches
    # (A/B switch), every pass timed -- what the fused launch's k-NN part costs when rocprofv3 / HIP events can see it
    knn_stage = None
    if int(os.environ.get('FLIMO_BENCH_TIMING', '1')) == 1 and os.environ.get('FLIMO_FUSE', '1') != '0':
        loc.hip.set_path_switches(fuse=0)
        for _ in range(2):
            step()
        loc.hip.timing_split(reset=True)
        for _ in range(12):
            step()
        d = loc.hip.timing_split(reset=True)
        loc.hip.set_path_switches(fuse=1)
        if d["separate_n"]:
            knn_stage = {"knn_us": 1e3 * d["knn_ms"] / d["separate_n"], "widen_us": 1e3 * d["widen_ms"] / d["separate_n"],
                         "fit_us": 1e3 * d["fit_ms"] / d["separate_n"], "passes_timed": d["separate_n"]}
    loc.hip.set_timing_deferred(False)
    loc.hip.set_timing(0)
    if knn_stage:
        step()          # back on the one-launch path: same state as at the end of the timed region
        assert np.array_equal(loc.get_x(), x_ref) or not headline

    # Informational (never `value`): the same step from S independent scan streams on ONE GPU at once -- S Localizers, each with its
    # own map and resident scan, each driven by its own host thread (SURVEY 8 e / BASELINE configs[4] on a single GPU).  One stream is
    # a chain of dependent round trips that leaves most of the chip idle; streams overlap.
    concurrent = None
    if rank == 0 and world == 1 and args.streams > 1:
        import threading
        from fast_limo_amd import synth
        S, K = args.streams, max(50, min(args.steps * 4, 300))
        extra = [(None, reg)]                                # stream 0 = the Localizer of the timed region (an idle context would still
        for sidx in range(1, S):                             # hold one of the GPU's few hardware queues)
            sc = synth.velodyne_scan(args.rings, args.azimuths, args.box, 10 + sidx)     # the seeds of configs[4]
            L = api.Localizer(api.default_cfg(gpu_device=local_rank % n_dev, num_threads=4, **caps))
            L.set_flags(add_to_map=False, download_clouds=False, keep_log=False)
            r1 = drive_to_prior(L, mp, sc, imu)
            xs_, Ps_ = L.get_x(), L.get_P()
            r2 = L.update_pointcloud(sc, 0.1)
            assert r1 == 1 and r2 == 0, (r1, r2)
            extra.append((L, L.register_resident_call(xs_, Ps_)))
        gate = threading.Barrier(S + 1)
        def stream_loop(reg_):
            for _ in range(10):
                assert reg_() == 0
            gate.wait()
            for _ in range(K):
                reg_()
        th = [threading.Thread(target=stream_loop, args=(e[1],)) for e in extra]
        for t in th: t.start()
        gate.wait(); tc0 = time.perf_counter()
        for t in th: t.join()
        tc = time.perf_counter() - tc0
        concurrent = {"streams_on_one_gpu": S, "steps_per_stream": K, "scans_per_s_aggregate": S * K / tc,
                      "scans_per_s_per_stream": K / tc,
                      "note": "informational: S independent Localizer streams (own map, own resident scan, own host thread) on one GPU; "
                              "`value` above is ONE stream.  Per-kernel profiles of this command are taken with --streams 0 (the overlapped "
                              "launches of this leg would enter rocprofv3's per-kernel averages)"}
        for e in extra[1:]:
            e[0].close()

    # the same step WITH the path exit (transform + Mapper::add of the registered scan), reported
    # beside `value`, never as `value`.  The first insertion stores the scan's new points; repeating the same scan is
    # then mostly rejected by the reference's down-sampling rule, so both are shown.  Runs after the timed region.
    with_insert = None
    if rank == 0:
        # two steps by default (the scan's first insertion and one repeat: 8 more k-NN launches next to the 200+ of the timed
        # region, so the profiler's per-kernel average stays the benchmark's); --with-insert runs six for a steadier repeat figure
        t_ins = []
        sizes = [loc.map_size()]
        for k in range(6 if args.with_insert else 2):
            t1 = time.perf_counter()
            step()
            loc.hip.map_add_scan(loc.get_x(), 0.2 + 0.1 * k)
            t_ins.append(time.perf_counter() - t1)
            sizes.append(loc.map_size())
        with_insert = {"first_ms": 1e3 * t_ins[0], "points_stored_first": sizes[1] - sizes[0],
                       "repeat_ms": 1e3 * float(np.median(t_ins[1:])), "points_stored_repeat": sizes[-1] - sizes[1],
                       "scans_per_s_first": 1.0 / t_ins[0], "scans_per_s_repeat": 1.0 / float(np.median(t_ins[1:]))}

    # the whole Localizer::updatePointCloud from a HOST cloud (input filters, time sort, upload,
    # deskew, update, device-resident map insert on the Mapper's worker thread), sweep after sweep, reported beside `value`.
    # "tied": the stamps of a spinning sensor (all rings of a column share one stamp), "unique": every point its own stamp.
    end_to_end = None
    if rank == 0 and not args.no_end_to_end:
        from fast_limo_amd import synth
        st, w, a = imu
        i = loc._imu_cursor
        loc.set_flags(add_to_map=True, download_clouds=False, keep_log=False)
        end_to_end = {"points_per_sweep": int(scan.shape[0])}
        k = 2
        # The first sweeps inserted into a map that has never seen a sweep are slower than the ones behind them (raw sweeps crowd the
        # cells under the sensor; the second level over that region comes into being after about ten of them:
        # tests/dev/gpu_e2e_order_probe.py -- rounds 1-5 ran the "tied" leg FIRST and read those sweeps as the cost of tied stamps).
        # They are run and reported on their own; the two legs behind them start from the same warmed map.
        first = []
        for j in range(12):
            sc = synth.velodyne_scan(args.rings, args.azimuths, args.box, 500 + j)
            sw = api.make_points_velodyne(sc)
            until = 0.1 * (k + 1) + 0.005
            while i < len(st) and st[i] <= until:
                loc.update_imu(st[i], w[i], a[i]); i += 1
            t1 = time.perf_counter()
            rc = loc.update_pointcloud_points(sw, 0.1 * k)
            loc.sync()
            first.append(time.perf_counter() - t1)
            assert rc == 0, rc
            k += 1
        end_to_end["first_sweeps_into_a_static_map"] = {"sweeps": len(first), "ms_per_sweep": 1e3 * float(np.median(first[1:])),
                                                        "ms_first": 1e3 * first[0], "ms_each": [round(1e3 * t, 3) for t in first],
                                                        "note": "tied stamps, each sweep waited for; the legs below follow on the same map"}
        for label in ("tied", "unique"):
            sweeps = []
            for j in range(2 * args.e2e_sweeps):
                # different sweeps for every run: a sweep the map has already seen stores almost nothing
                sc = synth.velodyne_scan(args.rings, args.azimuths, args.box, 100 + j + (2 * args.e2e_sweeps if label == "unique" else 0))
                if label == "unique":
                    sc[:, 4] += (np.arange(sc.shape[0]) % args.rings).astype(np.float32) * np.float32(1.5e-6)
                sweeps.append(api.make_points_velodyne(sc))
            lat, tot_sw = [], []
            loc.sync()
            for j in range(args.e2e_sweeps):
                until = 0.1 * (k + 1) + 0.005
                while i < len(st) and st[i] <= until:          # IMU propagation between two sweeps: not part of the sweep's cost
                    loc.update_imu(st[i], w[i], a[i]); i += 1
                t1 = time.perf_counter()
                rc = loc.update_pointcloud_points(sweeps[j], 0.1 * k)
                t2 = time.perf_counter()
                loc.sync()                                      # the map insert that ends the sweep (Mapper's worker thread)
                t3 = time.perf_counter()
                lat.append(t2 - t1); tot_sw.append(t3 - t1)
                assert rc == 0, rc
                k += 1
            stg = loc.stage_times()
            # ... and back to back, as a driver feeds them: the map insert that ends sweep j (Mapper's worker thread) runs while the
            # caller propagates the IMU and the input stage of sweep j + 1 (upload, filters, stamps) runs on its own context
            loc.sync()
            t1 = time.perf_counter()
            ks = np.arange(k, k + args.e2e_sweeps)
            i1 = int(np.searchsorted(st, 0.1 * (ks[-1] + 1) + 0.005, side="right"))
            # (from native code, flimo_loc_replay: a binding's per-call cost between two sweeps -- 0.1 ms through ctypes -- would
            #  hide the insert by itself)
            status, _ = loc.replay(sweeps[args.e2e_sweeps:], 0.1 * ks, 0.1 * (ks + 1) + 0.005, st[i:i1], w[i:i1], a[i:i1])
            loc.sync()
            back_to_back = (time.perf_counter() - t1) / args.e2e_sweeps
            assert not status.any(), status
            i = i1
            k += args.e2e_sweeps
            end_to_end[label + "_stamps"] = {"ms_per_sweep": 1e3 * float(np.median(tot_sw[1:])),
                                             "ms_per_sweep_sustained": 1e3 * back_to_back,
                                             "ms_per_sweep_each_waited_for": 1e3 * float(np.mean(tot_sw[1:])),
                                             "call_returns_after_ms": 1e3 * float(np.median(lat)),
                                             "sweeps": args.e2e_sweeps,
                                             "note": "ms_per_sweep: median sweep, every call followed by a wait for its map insert (the meaning of rounds "
                                                     "1-3; each_waited_for = the mean of the same); ms_per_sweep_sustained: sweeps fed back to back from "
                                                     "native code (flimo_loc_replay; IMU propagation between them included, one wait for the last "
                                                     "insert at the end)",
                                             "last_sweep_stages_ms": {kk: 1e3 * float(v) for kk, v in stg.items()}}
        end_to_end["ms"] = end_to_end["tied_stamps"]["ms_per_sweep"]
        end_to_end["scans_per_s"] = 1e3 / end_to_end["ms"]
        end_to_end["ms_sustained"] = end_to_end["tied_stamps"]["ms_per_sweep_sustained"]
        # the same sweeps with the clouds handed back to the caller (download_clouds = true is the class's default and what the
        # reference's ROS wrapper needs: src/main.cpp:27-31 publishes get_pointcloud() after every sweep)
        loc.set_flags(add_to_map=True, download_clouds=True, keep_log=False)
        sweeps = [api.make_points_velodyne(synth.velodyne_scan(args.rings, args.azimuths, args.box, 300 + j)) for j in range(2 * args.e2e_sweeps)]
        lat, tot_sw = [], []
        cloud_buf = np.zeros((max(s_.shape[0] for s_ in sweeps), 3), np.float32)
        loc.sync()
        for j in range(args.e2e_sweeps):
            until = 0.1 * (k + 1) + 0.005
            while i < len(st) and st[i] <= until:
                loc.update_imu(st[i], w[i], a[i]); i += 1
            t1 = time.perf_counter()
            rc = loc.update_pointcloud_points(sweeps[j], 0.1 * k)
            n_final = loc.final_scan(out=cloud_buf).shape[0]     # the caller takes the cloud (get_pointcloud) into its own buffer
            t2 = time.perf_counter()
            loc.sync()
            t3 = time.perf_counter()
            lat.append(t2 - t1); tot_sw.append(t3 - t1)
            assert rc == 0 and n_final == sweeps[j].shape[0], (rc, n_final)
            k += 1
        stg = loc.stage_times()
        loc.sync()
        t1 = time.perf_counter()
        for j in range(args.e2e_sweeps, 2 * args.e2e_sweeps):                     # back to back (see above)
            i1 = int(np.searchsorted(st, 0.1 * (k + 1) + 0.005, side="right"))
            loc.update_imu_n(st[i:i1], w[i:i1], a[i:i1]); i = i1
            rc = loc.update_pointcloud_points(sweeps[j], 0.1 * k)
            n_final = loc.final_scan(out=cloud_buf).shape[0]
            assert rc == 0 and n_final == sweeps[j].shape[0], (rc, n_final)
            k += 1
        loc.sync()
        back_to_back = (time.perf_counter() - t1) / args.e2e_sweeps
        end_to_end["with_clouds"] = {"ms": 1e3 * float(np.median(tot_sw[1:])), "ms_sustained": 1e3 * back_to_back,
                                     "ms_each_waited_for": 1e3 * float(np.mean(tot_sw[1:])),
                                     "call_returns_after_ms": 1e3 * float(np.median(lat)),
                                     "sweeps": args.e2e_sweeps,
                                     "last_sweep_stages_ms": {kk: 1e3 * float(v) for kk, v in stg.items()}}
        end_to_end["map_points_after"] = loc.map_size()
        end_to_end["shipped_config"] = shipped_config_leg(local_rank % n_dev, not args.no_cpu_baseline)
        end_to_end["shipped_config_tied"] = shipped_config_leg(local_rank % n_dev, not args.no_cpu_baseline, tied=True)

    elapsed, value = aggregate(dist, torch, elapsed, world, args.steps)

    if rank == 0:
        out = {
            "metric": "scans/sec (64k-pt scan, 1M-pt map) + kNN HBM GB/s vs roofline; ATE vs CPU ref",
            "value": value,
            "unit": "scans/s",
            "n_gpus": n_gpus,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": ("BASELINE.json configs[1]: " if (args.rings, args.azimuths, args.map_points, args.box) == (64, 1024, 1000000, 100.0)
                                    else "scaled variant of BASELINE.json configs[1]: ") +
                                   "%d-pt Velodyne-like scan (%d rings x %d azimuths), "
                                   "%d-pt box-world map, k=5, MAX_NUM_ITERS=3, GPU deskew + iterated ESKF update per step"
                                   % (scan.shape[0], args.rings, args.azimuths, mp.shape[0]),
                       "parallelism": "replicas x%d (independent scan streams, no collective)" % world,
                       "passes_per_step": n_passes / max(args.steps, 1), "steps_bit_reproducible": repro_bitwise,
                       "update": ("chained: every iteration's launches queued at once, the filter's algebra on the device (flimo_update_chain)"
                                  if chain["chains"] else ("host loop, pipelined: the next pass waits on the GPU for the pose the host stores into device memory (flimo_set_pass_pipeline)"
                                                           if pipe_found else "host loop over single passes (this host's launch round trip is short, or FLIMO_HOST_UPDATE=1)")),
                       "chains_run": chain["chains"], "chains_handed_back_early": chain["handed_back"], "chains_declined": chain["declined"],
                       "host_loop_passes_found_waiting": pipe_found, "host_loop_passes_queued_for_nothing": pipe_wasted},
            "update_layouts": modes,
            "host_us_per_step": {"deskew_call": 1e6 * hp["deskew_s"] / args.steps, "update": 1e6 * hp["update_s"] / args.steps,
                                 "in_match_reduce": 1e6 * hp["match_reduce_s"] / args.steps},
            "with_map_insert": with_insert,
            "end_to_end": end_to_end,
            "concurrent_streams": concurrent,
        }
        cb, E, x_o = (None, None, None)
        if world == 1 and not args.no_cpu_baseline:
            cb, E, x_o = cpu_baseline(args)
            out["cpu_baseline"] = cb
            dpos = float(np.abs(x_ref[0:3] - x_o[0:3]).max())
            drot = float(2.0 * np.abs(x_ref[3:6] - x_o[3:6]).max())
            out["pose_err_vs_cpu"] = {"pos_m": dpos, "rot_rad": drot, "tolerance": 1e-4}
        Eq = E if E else E_FALLBACK
        if not E and (args.rings, args.azimuths, args.map_points, args.box) != (64, 1024, 1000000, 100.0):
            Eq = None                      # the constant only describes configs[1]
        bytes_per_query = (16.0 + 16.0 * Eq + NBR_BYTES) if Eq else None
        qpl = tot["queries"] / tot["passes"] if tot["passes"] else float(scan.shape[0])      # queries per launch (every point of the scan is a query here)
        us = lambda ms, n: (1e3 * ms / n) if n else None
        in_region_us = us(split["fused_ms"], split["fused_n"])      # sparse samples inside the timed region (each perturbs the wall time)
        one_us = dense["one_launch_pass_us"] if (dense and dense.get("one_launch_pass_us")) else in_region_us
        # ^ mean duration of the one-launch pass from HIP events on its dispatch: the dense series (every pass of 12 steps timed,
        #   right after the timed region, events read after the series) when it ran -- a timed dispatch issued once in 30 passes
        #   runs cold and reads ~10 % long
        sep = {"knn": us(split["knn_ms"], split["separate_n"]), "widen": us(split["widen_ms"], split["separate_n"]),
               "fit_reduce": us(split["fit_ms"], split["separate_n"])} if split["separate_n"] else None
        if one_us:
            kernel, dur_us, n_timed = "knn5_kernel<2, 8, true, false>: the whole measurement pass in one launch (k-NN fast path + in-kernel widening + plane fit + residual/Jacobian + H^T H reduction)", one_us, (dense["one_launch_passes_timed"] if (dense and dense.get("one_launch_pass_us")) else split["fused_n"])
        else:                              # developer switches (FLIMO_FUSE=0 ...): the k-NN dispatch alone
            kernel, dur_us, n_timed = "knn5_kernel<2, 8, false, false>: k-NN dispatch (separate widening / fit dispatches)", us(tot["knn_ms"], tot["passes"]), tot["passes"]
        achieved = bytes_per_query * qpl / (dur_us * 1e-6) / 1e9 if (dur_us and bytes_per_query) else None
        traffic = pmc_traffic(qpl)
        out["roofline"] = {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBPS,
                           "unit": "GB/s", "frac": (achieved / HBM_PEAK_GBPS) if achieved else None, "traffic": traffic,
                           "hbm_utilisation_measured": (traffic / (dur_us * 1e-6) / 1e9 / HBM_PEAK_GBPS) if (traffic and dur_us) else None,
                           "bytes_per_query": bytes_per_query, "E_evals_per_query": Eq,
                           "queries_per_launch": qpl, "mean_launch_us": dur_us, "timed_launches": n_timed,
                           "kernel_in_timed_region": ("knn5_chain_kernel<2, 8, true, false>: the same device function queued AHEAD of its pose (pipelined host loop): its "
                                                      "event duration includes the wait for the pose, so `mean_launch_us` is taken from the dense series right after the "
                                                      "region, where every pass is launched when its pose is known (knn5_kernel<2, 8, true, false>)"
                                                      if pipe_found else "knn5_kernel<2, 8, true, false> (passes launched when their pose is known)"),
                           "reproduce": "rocprofv3 --kernel-trace --stats of `FLIMO_PIPELINE=0 python3 bench.py --streams 0`: profiles/r06/bench_r06_prof_nopipeline_kernel_stats.csv, "
                                        "row knn5_kernel<2, 8, true, false>",
                           "step": {"value_regions": value_regions, "kernel_us_per_step": kernel_us_per_step,
                                    "step_minus_kernels_us": ((1e3 * 1e3 * elapsed / args.steps) - kernel_us_per_step["total"]) if kernel_us_per_step else None,
                                    "note": "value_regions: the timed K-step region and nine more right behind it (min / median / max of scans/s); "
                                            "kernel_us_per_step: every launch of 12 steps timed by HIP events on its dispatch"},
                           "note": "achieved = ALGORITHMIC k-NN bytes (16 B query + 16 B x E candidate points of the reference's own traversal + 32 B neighbour "
                                   "record) per launch / launch duration; the launch also does the plane fit, the residual / Jacobian rows and the "
                                   "reduction, which add no algorithmic bytes (the five neighbours were just read).  `traffic` / "
                                   "`hbm_utilisation_measured` are the HBM-side bytes from PMC counters: the 1M-point map lives in L2 / Infinity Cache",
                           "stage": {"one_launch_pass_us_sampled_inside_timed_region": in_region_us, "one_launch_passes_timed": split["fused_n"],
                                     "separate_dispatch_pass_us": sep, "separate_dispatch_passes_timed": split["separate_n"],
                                     "passes_in_one_launch": n_fused_passes, "passes_total": n_passes,
                                     "dense_after_timed_region": dense,
                                     "note": "separate-dispatch passes (the first pass of the poor prior) are three launches: k-NN, widening of the "
                                             "worklist, fit + reduction"}}
        if concurrent and bytes_per_query:
            # the chip's aggregate k-NN rate with S streams in flight: algorithmic bytes of all their passes / wall time
            ppstep = n_passes / max(args.steps, 1)
            agg = concurrent["scans_per_s_aggregate"] * ppstep * bytes_per_query * qpl / 1e9
            concurrent["knn_algorithmic_GBps_aggregate"] = agg
            concurrent["knn_algorithmic_frac_of_hbm_peak"] = agg / HBM_PEAK_GBPS
        if knn_stage and bytes_per_query:
            st_us = knn_stage["knn_us"] + knn_stage["widen_us"]
            out["roofline"]["knn_stage_separate_dispatches"] = dict(
                knn_stage, stage_us_per_pass=st_us, achieved=bytes_per_query * qpl / (st_us * 1e-6) / 1e9,
                frac=bytes_per_query * qpl / (st_us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                note="A/B series after the timed region with the pass split into dispatches: k-NN (fast path + in-kernel widening; the "
                     "first pass of the poor prior hands its clustered stragglers to the widening dispatch) + widening, mean per pass "
                     "over all 4 pass positions; same algorithmic bytes")
        loc.close()
        loc = None
        if world == 1 and not args.no_hbm_regime:
            out["roofline"]["hbm_regime"] = hbm_regime_leg(local_rank % n_dev, args.hbm_steps, with_oracle=not args.no_cpu_baseline)
        if rank == 0 and world == 1 and not args.no_crowded:
            out["roofline"]["crowded"] = crowded_leg(local_rank % n_dev, args.rings, args.azimuths, args.map_points, args.box)
        # the figures a reader of a trimmed record wants, as plain scalars of `config` / `roofline` (nested objects may be dropped)
        flat = out["config"]
        if value_regions:
            flat["scans_per_s_long_region"] = value_regions.get("scans_per_s_region_of_%d_steps" % value_regions["long_region_steps"])
            flat["scans_per_s_median_of_regions"] = value_regions["scans_per_s_median"]
        if end_to_end:
            if end_to_end.get("first_sweeps_into_a_static_map"):
                flat["pcie_inclusive_ms_per_sweep_first_sweeps_into_a_static_map"] = end_to_end["first_sweeps_into_a_static_map"]["ms_per_sweep"]
            for key, src in (("tied", end_to_end.get("tied_stamps")), ("unique", end_to_end.get("unique_stamps")),
                             ("shipped", end_to_end.get("shipped_config")), ("shipped_tied", end_to_end.get("shipped_config_tied"))):
                if not src:
                    continue
                flat["pcie_inclusive_ms_per_sweep_%s" % key] = src.get("ms_per_sweep")
                flat["pcie_inclusive_ms_per_sweep_%s_sustained" % key] = src.get("ms_per_sweep_sustained")
                if "speedup_vs_cpu_oracle" in src:
                    flat["speedup_vs_cpu_oracle_%s" % key] = src["speedup_vs_cpu_oracle"]
        rf = out["roofline"]
        ks = rf.get("knn_stage_separate_dispatches")
        if ks and bytes_per_query:
            rf["knn_kernel_alone_us"] = ks["knn_us"]
            rf["knn_kernel_alone_frac"] = bytes_per_query * qpl / (ks["knn_us"] * 1e-6) / 1e9 / HBM_PEAK_GBPS
            rf["knn_stage_frac"] = ks["frac"]
        hb = rf.get("hbm_regime")
        if hb:
            rf["hbm_regime_ms_per_step"] = hb.get("ms_per_step")
            rf["hbm_regime_frac"] = hb.get("frac")
            rf["hbm_regime_widen_us"] = (hb.get("separate_dispatch_pass_us") or {}).get("widen")
            rf["hbm_regime_insert_ms_repeat"] = (hb.get("map_insert_ms") or {}).get("repeat")
            rf["hbm_regime_insert_ms_first"] = (hb.get("map_insert_ms") or {}).get("first")
        if kernel_us_per_step:
            rf["kernel_us_per_step"] = kernel_us_per_step["total"]
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if loc is not None:
        loc.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
