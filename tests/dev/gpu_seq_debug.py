import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
from fast_limo_amd import synth, api
import oracle_py as O
caps = dict(MAX_NUM_PC2MATCH=10**7, MAX_NUM_MATCHES=10**7)
mp = synth.box_world_map(1000000, 100.0, 1)
st, w, a = synth.stationary_imu(0.0, 1.0)
G = api.Localizer(api.default_cfg(num_threads=32, **caps)); G.set_flags(keep_log=True)
Lo = O.Localizer(O.default_cfg(num_threads=int(os.environ.get("OT", 32)), **caps))
G.map_add(mp); Lo.map_add(mp)
srt = lambda a: a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]
i = 0
for k in range(4):
    until = 0.1 * (k + 1) + 0.005
    while st[i] <= until:
        G.update_imu(st[i], w[i], a[i]); Lo.update_imu(st[i], w[i], a[i]); i += 1
    scan = synth.velodyne_scan(64, 1024, 100.0, 2 + k)
    xg0, xo0 = G.get_x(), Lo.get_x()
    rg = G.update_pointcloud(scan, 0.1 * k); ro = Lo.update_pointcloud(scan, 0.1 * k)
    Pg, Po = G.get_P(), Lo.get_P()
    print("   prior P rel diff", np.abs(Pg - Po).max() / np.abs(Po).max(), "diag rel", (np.abs(np.diag(Pg) - np.diag(Po)) / np.abs(np.diag(Po))).max(), "cond", np.linalg.cond(Po))
    print("scan", k, "prior diff", np.abs(xg0 - xo0).max(), "post diff pos", np.abs(G.get_x()[:3] - Lo.get_x()[:3]).max(), "map", G.map_size(), Lo.map_size())
    pg, po = G.passes(), Lo.iters()
    for j, (a_, b_) in enumerate(zip(pg, po)):
        dd = np.abs(a_["dx"] - b_["dx"]); c = int(dd.argmax())
        print("   pass", j, "M", a_["M"], b_["M"], "max|dHTH|/max", np.abs(a_["HTH"] - b_["HTH"]).max() / max(np.abs(b_["HTH"]).max(), 1e-30), "ddx", dd.max(), "comp", c, "dx[c]", a_["dx"][c], b_["dx"][c], "pose ddx", dd[:6].max())
    if rg == 0:
        fg, fo = G.final_scan(), Lo.final_scan()
        print("   final_scan max diff", np.abs(fg - fo).max(), "n differing", int((fg != fo).any(axis=1).sum()), "pc2match equal", np.array_equal(G.pc2match(), Lo.pc2match()))
G.close()
