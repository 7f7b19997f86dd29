"""Developer probe: the shipped-configuration leg of bench.py alone (no CPU oracle), stage times of every sweep.
FLIMO_PROF_FRONT=1 adds the input stage's own breakdown; run under rocprofv3 --kernel-trace --stats for the kernels."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench

if __name__ == "__main__":
    import __graft_entry__ as g
    g.build()
    from fast_limo_amd import api, synth
    n_sweeps, n_pts, speed = 12, 120000, 10.0
    st, w, a = synth.stationary_imu(0.0, 0.1 * n_sweeps + 0.06)
    lid_t = (8.086759e-01, -3.195559e-01, 7.997231e-01)
    lid_R = (9.999976e-01, -7.854027e-04, 2.024406e-03, 7.553071e-04, 9.998898e-01, 1.482454e-02,
             -2.035826e-03, -1.482298e-02, 9.998881e-01)
    common = dict(MAX_NUM_PC2MATCH=10000, MAX_NUM_MATCHES=5000, voxel_active=1, leaf_size=1.0, crop_active=1,
                  dist_active=1, min_dist=4.0, rate_active=1, rate_value=4, time_offset=1,
                  lidar2baselink_t=lid_t, lidar2baselink_R=lid_R, accel_bias=(0.01, 0.01, 0.01), gyro_bias=(0.01, 0.01, 0.01),
                  cov_gyro=6.01e-4, cov_acc=1.53e-2, cov_bias_gyro=1.54e-5, cov_bias_acc=3.38e-4)
    sweeps = [api.make_points_velodyne(synth.corridor_scan(k, n_pts, 4321, speed=speed)) for k in range(n_sweeps)]
    clouds = os.environ.get("PROBE_CLOUDS", "1") == "1"
    G = api.Localizer(api.default_cfg(cropBoxMin=(-1.0, -1.0, -1.0), cropBoxMax=(1.0, 1.0, 1.0), debug=1 if clouds else 0,
                                      num_threads=os.cpu_count() or 1, **common))
    G.set_flags(add_to_map=True, download_clouds=clouds, keep_log=False)
    x0 = G.get_x(); x0[14] = speed; G.set_x(x0)
    buf = np.zeros((n_pts, 3), np.float32)
    i = 0
    for k in range(n_sweeps):
        i1 = int(np.searchsorted(st, 0.1 * (k + 1) + 0.005, side="right"))
        G.update_imu_n(st[i:i1], w[i:i1], a[i:i1]); i = i1
        t1 = time.perf_counter()
        rc = G.update_pointcloud_points(sweeps[k], 0.1 * k)
        if clouds:
            G.final_scan(out=buf)
        t2 = time.perf_counter()
        G.sync()
        t3 = time.perf_counter()
        stg = {kk: round(1e3 * v, 3) for kk, v in G.stage_times().items()}
        print("sweep %2d rc %d call %.3f ms  +insert %.3f ms  stages %s  pc2match %d  passes %d" %
              (k, rc, 1e3 * (t2 - t1), 1e3 * (t3 - t2), stg, G.pc2match().shape[0] if clouds else -1, G.hip.pass_count()), flush=True)
    G.close()
