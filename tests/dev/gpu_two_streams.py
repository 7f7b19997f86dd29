"""Developer probe (GPU box): aggregate scans/s of S independent Localizer streams on ONE GPU (threads; the C calls release the GIL)."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from fast_limo_amd import synth, api
import bench
caps = dict(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
mp = synth.box_world_map(1000000, 100.0, 1)
imu = synth.stationary_imu(0.0, 0.4)
def make(seed):
    scan = synth.velodyne_scan(64, 1024, 100.0, seed)
    L = api.Localizer(api.default_cfg(num_threads=4, **caps)); L.set_flags(add_to_map=False, download_clouds=False)
    assert bench.drive_to_prior(L, mp, scan, imu) == 1
    x, P = L.get_x(), L.get_P()
    assert L.update_pointcloud(scan, 0.1) == 0          # makes the raw scan + IMU frames resident
    return L, x, P
if os.environ.get("WITH_TORCH"):
    import torch; print("torch devices", torch.cuda.device_count())
    if os.environ.get("WITH_TORCH") == "2": torch.cuda.synchronize()
idle = None
if os.environ.get("WITH_IDLE"):
    idle = make(99)
if os.environ.get("DUP"):
    sys.stdout.flush(); os.dup2(2, 1)
for S in [int(v) for v in os.environ.get("SLIST", "1,2,3,4").split(",")]:
    locs = [make(2 + s) for s in range(S)]
    K = 300
    def run(L, x, P):
        reg = L.register_resident_call(x, P)
        for _ in range(20): assert reg() == 0
        bar.wait()
        for _ in range(K): reg()
    bar = threading.Barrier(S + 1)
    th = [threading.Thread(target=run, args=l) for l in locs]
    for t in th: t.start()
    bar.wait(); t0 = time.perf_counter()
    for t in th: t.join()
    dt = time.perf_counter() - t0
    print("%d stream(s) on one GPU: %.0f scans/s aggregate (%.0f per stream)" % (S, S * K / dt, K / dt), flush=True)
    for l in locs: l[0].close()
