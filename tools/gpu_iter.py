"""Developer script: N passes of flimo_match_reduce on the cfg-2 workload (for rocprofv3)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from fast_limo_amd import synth, _lib
NMAP = int(os.environ.get("NMAP", 1000000)); L = float(os.environ.get("LBOX", 100.0))
mp = synth.box_world_map(NMAP, L, 1)
scan = np.ascontiguousarray(synth.velodyne_scan(int(os.environ.get("RINGS", 64)), int(os.environ.get("AZ", 1024)), L, 2)[:, :3])
ctx = _lib.HipCtx(0)
ctx.map_config(cell_size=float(os.environ.get("CELL", 0.5)))
ctx.map_add(mp); ctx.scan_set(scan)
x0 = np.zeros(26); x0[6] = 1; x0[10] = 1; x0[25] = -9.809
cfg = _lib.default_match_cfg(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
for _ in range(int(os.environ.get("ITERS", 30))):
    HTH, HTh, M = ctx.match_reduce(x0, cfg)
print("M", M)
