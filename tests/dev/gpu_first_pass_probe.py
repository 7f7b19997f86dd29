"""Developer probe (GPU box): the FIRST pass of bench.py's step.  Which layout does it run in (one launch / three dispatches), how many
queries does it leave beyond their 3x3x3 block, and what would it cost as one launch (FLIMO_TAIL_MAX lifts the straggler limit)?
Run once per setting: the switch is read when the context is made."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
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
x_ref = loc.get_x()
rates = []
for _ in range(10):
    t0 = time.perf_counter()
    for _ in range(100):
        reg()
    rates.append(100 / (time.perf_counter() - t0))
loc.hip.set_timing(1); loc.hip.set_timing_stride(1); loc.hip.set_timing_deferred(True)
loc.hip.timing_split(reset=True)
for _ in range(12):
    reg()
d = loc.hip.timing_split(reset=True)
loc.hip.set_timing_deferred(False); loc.hip.set_timing(0)
print("FLIMO_TAIL_MAX=%s: %.0f scans/s (median of 10 x 100 steps); per step: %.1f one-launch passes of %.2f us, %.1f passes in separate dispatches "
      "(k-NN %.2f + widening %.2f + fit %.2f us); pass kernels %.1f us per step; last pass left %d stragglers; state equal to the reference run: %s"
      % (os.environ.get("FLIMO_TAIL_MAX", "default"), float(np.median(rates)), d["fused_n"] / 12.0, 1e3 * d["fused_ms"] / max(1, d["fused_n"]), d["separate_n"] / 12.0,
         1e3 * d["knn_ms"] / max(1, d["separate_n"]), 1e3 * d["widen_ms"] / max(1, d["separate_n"]), 1e3 * d["fit_ms"] / max(1, d["separate_n"]),
         1e3 * (d["fused_ms"] + d["knn_ms"] + d["widen_ms"] + d["fit_ms"]) / 12.0, loc.hip.last_stragglers(), np.array_equal(loc.get_x(), x_ref)), flush=True)
print("stragglers by pass position:", loc.hip.stragglers_by_pass())
loc.close()
