"""Developer probe: two fresh Localizers through the same registration (chained update, per-pass log on): the first quantity that
differs bit for bit.  usage: python tests/dev/gpu_chain_repro.py [n_scan] [runs]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from common import CAPS, cfg1_scene, drive_two_scans
from fast_limo_amd import api

n_scan = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
mp, scan5, imu = cfg1_scene(n_map=200000, n_scan=n_scan, L=40.0)
ref = None
for r in range(runs):
    D = api.Localizer(api.default_cfg(**CAPS))
    D.set_flags(add_to_map=False, keep_log=True)
    assert drive_two_scans(D, mp, scan5, imu) == [1, 0]
    cur = dict(passes=D.passes(), x=D.get_x().copy(), P=D.get_P().copy())
    D.close()
    if ref is None:
        ref = cur
        print("run 0: passes", len(cur["passes"]), "M", [p["M"] for p in cur["passes"]])
        continue
    for i, (a, b) in enumerate(zip(ref["passes"], cur["passes"])):
        for k in ("HTH", "HTh", "dx", "x_after"):
            if not np.array_equal(a[k], b[k]):
                d = np.abs(np.asarray(a[k]) - np.asarray(b[k]))
                print(f"run {r}: pass {i} {k} differs: {int((d > 0).sum())} elements, max abs {d.max():.3e}, max rel {(d / (np.abs(a[k]) + 1e-300)).max():.3e}")
    for k in ("x", "P"):
        if not np.array_equal(ref[k], cur[k]):
            d = np.abs(ref[k] - cur[k])
            idx = np.argwhere(d > 0)
            print(f"run {r}: final {k} differs: {len(idx)} elements, max abs {d.max():.3e}; first indices {idx[:8].tolist()}")
    print(f"run {r}: compared")
