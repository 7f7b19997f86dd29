"""Generates tests/golden/cfg1_golden.npz from the CPU oracle (run in the build container).

The reference itself has no tests, fixtures or golden vectors and cannot be built here (Eigen3, PCL
and Boost are absent), so these vectors pin the ORACLE's behaviour (regression) and give the GPU
suite a data-only target that does not need the oracle at run time.  PARITY UNPINNED w.r.t. the
reference -- see DESIGN.md.

Contents (config 1 of BASELINE.json: 4096-pt scan, 50k-pt plane map, 3 IKFoM iterations):
  x_final[26], P_diag[23]        state after scan 2
  M[p], HTH[p,12,12], HTh[p,12], dx[p,23], x_after[p,26]   per update pass
  knn_q[256,3], knn_sqd[256,5], knn_nbr[256,5,3]           octree k-NN of 256 probe queries
  plane_in[64,5,3], plane_sqd[64,5], plane_n[64,4], plane_ok[64]   plane fits
  E                              leaf-point distance evaluations per query in the last pass
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle")); sys.path.insert(0, os.path.dirname(HERE))
import oracle_py as O  # noqa: E402
from common import CAPS, cfg1_scene, drive_two_scans  # noqa: E402


class NoInsert(O.Localizer):
    def update_pointcloud(self, pts, stamp):
        return super().update_pointcloud(pts, stamp, add_to_map=False)


def main():
    mp, scan, imu = cfg1_scene()
    L = NoInsert(O.default_cfg(num_threads=1, **CAPS))
    rcs = drive_two_scans(L, mp, scan, imu)
    assert rcs == [1, 0], rcs
    it = L.iters()
    st = L.stats()
    oc = O.Octree(); oc.update(mp)
    rs = np.random.RandomState(7)
    q = (mp[rs.choice(mp.shape[0], 256, replace=False)] + rs.normal(0, 0.05, (256, 3))).astype(np.float32)
    nbr, sqd, cnt, _ = oc.knn(q, 5)
    assert np.all(cnt == 5)
    pn, pok = [], []
    for i in range(64):
        n, ok = O.plane_fit(nbr[i], sqd[i])
        pn.append(n); pok.append(ok)
    np.savez_compressed(
        os.path.join(HERE, "cfg1_golden.npz"),
        x_final=L.get_x(), P_diag=np.diag(L.get_P()),
        M=np.array([p["M"] for p in it]), HTH=np.array([p["HTH"] for p in it]), HTh=np.array([p["HTh"] for p in it]),
        dx=np.array([p["dx"] for p in it]), x_after=np.array([p["x_after"] for p in it]),
        knn_q=q, knn_sqd=sqd, knn_nbr=nbr,
        plane_in=nbr[:64], plane_sqd=sqd[:64], plane_n=np.array(pn), plane_ok=np.array(pok),
        E=np.float64(st["evals"] / st["queries"]))
    print("wrote cfg1_golden.npz: passes", len(it), "M", [p["M"] for p in it], "pos", L.get_x()[:3])


if __name__ == "__main__":
    main()
