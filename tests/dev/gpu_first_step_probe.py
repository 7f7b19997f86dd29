"""Developer probe (GPU box): why is the FIRST step of bench.py's timed region 85 us slower than the others?  The bench's resident
step in a loop; before every tenth step one of: nothing, torch.cuda.synchronize(), a 100 us pause, a 2 ms pause."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
from fast_limo_amd import api
caps = dict(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
mp, scan, imu = bench.workload(0, 64, 1024, 1000000, 100.0)
loc = api.Localizer(api.default_cfg(num_threads=os.cpu_count() or 1, **caps))
loc.set_flags(add_to_map=False, download_clouds=False, keep_log=False)
bench.drive_to_prior(loc, mp, scan, imu)
x_prior, P_prior = loc.get_x(), loc.get_P()
loc.update_pointcloud(scan, 0.1)
reg = loc.register_resident_call(x_prior, P_prior)
for _ in range(20):
    reg()
def spin(us):
    t = time.perf_counter()
    while time.perf_counter() - t < us * 1e-6:
        pass
for label, act in (("nothing", lambda: None), ("torch.cuda.synchronize()", torch.cuda.synchronize), ("spin 100 us", lambda: spin(100)),
                   ("spin 2 ms", lambda: spin(2000)), ("sleep 2 ms", lambda: time.sleep(0.002)), ("hip pass_count getter", lambda: loc.hip.pass_count()),
                   ("host_profile(reset)", lambda: loc.host_profile(reset=True)), ("timing_totals(reset)", lambda: loc.hip.timing_totals(reset=True)),
                   ("chain_stats(reset)", lambda: loc.hip.chain_stats(reset=True)), ("pass_pipeline_stats", lambda: loc.hip.pass_pipeline_stats())):
    firsts, others = [], []
    for rep in range(6):
        act()
        for j in range(10):
            t0 = time.perf_counter(); reg(); dt = time.perf_counter() - t0
            (firsts if j == 0 else others).append(dt)
    print("%-28s first step after it %.0f us, the others %.0f us" % (label, 1e6 * np.median(firsts), 1e6 * np.median(others)), flush=True)
loc.close()

# ---- the bench's own sequence between its warm-up and its timed region, then bisected ----
loc = api.Localizer(api.default_cfg(num_threads=os.cpu_count() or 1, **caps))
loc.set_flags(add_to_map=False, download_clouds=False, keep_log=False)
bench.drive_to_prior(loc, mp, scan, imu)
x_prior, P_prior = loc.get_x(), loc.get_P()
loc.update_pointcloud(scan, 0.1)
reg = loc.register_resident_call(x_prior, P_prior)
reg()
def seq_all():
    loc.hip.timing_totals(reset=True); loc.hip.timing_split(reset=True); loc.hip.chain_stats(reset=True)
    loc.hip.pass_pipeline_stats(); loc.hip.pass_count(); loc.hip.fused_pass_count(); loc.host_profile(reset=True)
    torch.cuda.synchronize()
for label, pre, act in (("timing 0, full sequence", lambda: loc.hip.set_timing(0), seq_all),
                        ("timing 1 stride 2^30, full sequence", lambda: (loc.hip.set_timing(1), loc.hip.set_timing_stride(1 << 30)), seq_all),
                        ("timing 1 stride 2^30, nothing", lambda: (loc.hip.set_timing(1), loc.hip.set_timing_stride(1 << 30)), lambda: None),
                        ("timing 1 stride 2^30, timing_totals(reset) only", lambda: None, lambda: loc.hip.timing_totals(reset=True)),
                        ("timing 1 stride 2^30, timing_split(reset) only", lambda: None, lambda: loc.hip.timing_split(reset=True)),
                        ("timing 1 stride 2^30, chain_stats(reset) only", lambda: None, lambda: loc.hip.chain_stats(reset=True))):
    pre()
    firsts, others = [], []
    for rep in range(6):
        for _ in range(5):
            reg()
        act()
        for j in range(10):
            t0 = time.perf_counter(); reg(); dt = time.perf_counter() - t0
            (firsts if j == 0 else others).append(dt)
    print("%-52s first step after it %.0f us, the others %.0f us" % (label, 1e6 * np.median(firsts), 1e6 * np.median(others)), flush=True)
loc.close()
