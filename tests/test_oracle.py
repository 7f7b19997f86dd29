"""CPU suite, part 1: the oracle (CPU restatement of the reference) against
  * the committed golden vectors (regression pin of the oracle itself),
  * independent formulations available in this image: scipy cKDTree / brute force for the exact k-NN,
    numpy lstsq for the plane fit, a dense numpy re-derivation of one IESKF pass,
  * analytic known-answer tests (noise-free planes + known offset => known pose).
The reference has no tests / golden vectors of its own: PARITY UNPINNED.
"""
import math
import os

import numpy as np
import pytest
from scipy.spatial import cKDTree

from common import CAPS, cfg1_scene, drive_two_scans, pose_delta, sort_rows
from fast_limo_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg1_golden.npz")


class _NoInsert:
    def __init__(self, O, **kw):
        self.L = O.Localizer(O.default_cfg(num_threads=1, **kw))

    def map_add(self, m):
        self.L.map_add(m)

    def update_imu(self, *a):
        self.L.update_imu(*a)

    def update_pointcloud(self, p, s):
        return self.L.update_pointcloud(p, s, add_to_map=False)


def test_oracle_reproduces_golden(oracle):
    g = np.load(GOLD)
    mp, scan, imu = cfg1_scene()
    W = _NoInsert(oracle, **CAPS)
    assert drive_two_scans(W, mp, scan, imu) == [1, 0]
    it = W.L.iters()
    assert [p["M"] for p in it] == list(g["M"])
    np.testing.assert_array_equal(W.L.get_x(), g["x_final"])
    np.testing.assert_array_equal(np.array([p["HTH"] for p in it]), g["HTH"])
    np.testing.assert_array_equal(np.array([p["dx"] for p in it]), g["dx"])
    oc = oracle.Octree(); oc.update(mp)
    nbr, sqd, cnt, _ = oc.knn(g["knn_q"], 5)
    np.testing.assert_array_equal(sqd, g["knn_sqd"])
    np.testing.assert_array_equal(nbr, g["knn_nbr"])
    for i in range(64):
        n, ok = oracle.plane_fit(g["plane_in"][i], g["plane_sqd"][i])
        np.testing.assert_array_equal(n, g["plane_n"][i])
        assert ok == bool(g["plane_ok"][i])


def test_octree_knn_is_exact(oracle):
    """octree k-NN (Octree.hpp:526-599) == brute force in float32, and == cKDTree up to rounding."""
    rs = np.random.RandomState(11)
    mp = synth.box_world_map(30000, 20.0, 3)
    oc = oracle.Octree(); oc.update(mp)
    q = np.concatenate([mp[rs.choice(30000, 300)] + rs.normal(0, 0.1, (300, 3)).astype(np.float32),
                        rs.uniform(-30, 30, (100, 3)).astype(np.float32)]).astype(np.float32)
    nbr, sqd, cnt, ev = oc.knn(q, 5)
    assert np.all(cnt == 5) and ev > 0
    # brute force with the oracle's own float32 expression  x^2 + (y^2 + z^2)
    d = q[:, None, :] - mp[None, :, :]
    d2 = (d[..., 0] * d[..., 0]).astype(np.float32) + ((d[..., 1] * d[..., 1]).astype(np.float32) + (d[..., 2] * d[..., 2]).astype(np.float32))
    ref = np.sort(d2, axis=1)[:, :5]
    np.testing.assert_array_equal(sqd, ref)
    kd, _ = cKDTree(mp.astype(np.float64)).query(q.astype(np.float64), k=5)
    np.testing.assert_allclose(sqd, kd ** 2, rtol=2e-6, atol=1e-9)
    assert np.all(np.diff(sqd, axis=1) >= 0)


def test_octree_edge_cases(oracle):
    oc = oracle.Octree()
    nbr, sqd, cnt, _ = oc.knn(np.zeros((3, 3), np.float32), 5)      # empty tree (root_ == nullptr)
    assert np.all(cnt == 0)
    pts = np.array([[0, 0, 0], [1, 0, 0], [np.nan, 0, 0], [0, 1, 0]], np.float32)
    oc.update(pts)                                                    # NaN dropped (:243-244)
    assert oc.size() == 3
    nbr, sqd, cnt, _ = oc.knn(np.array([[0.1, 0, 0]], np.float32), 5)
    assert cnt[0] == 3                                                # fewer than k points
    # duplicates are kept
    oc2 = oracle.Octree(); oc2.update(np.zeros((40, 3), np.float32))
    assert oc2.size() == 40


def test_octree_insert_semantics(oracle):
    """initial build keeps everything; re-inserting into full min-extent leaves drops whole batches
    (Octree.hpp:399-401, effective bucket 32 => threshold 4); without down-sampling nothing is dropped."""
    mp = synth.box_world_map(40000, 10.0, 5)
    oc = oracle.Octree(); oc.update(mp)
    assert oc.size() == 40000
    oc.update(mp + np.float32(0.001))
    s2 = oc.size()
    assert 40000 < s2 < 80000                    # part of the second batch is dropped
    oc.update(mp + np.float32(0.002))
    assert oc.size() - s2 < s2 - 40000           # the fuller the leaves, the more is dropped
    oc_nd = oracle.Octree(downsample=False); oc_nd.update(mp); oc_nd.update(mp + np.float32(0.001))
    assert oc_nd.size() == 80000
    # root growth: points far outside the first bounding box are stored
    far = np.array([[500, 0, 0], [-500, 30, 2], [0, 0, 300]], np.float32)
    oc.update(far)
    got = sort_rows(oc.points())
    for p in far:
        assert (got == p).all(axis=1).any()


def test_plane_fit_matches_lstsq(oracle):
    rs = np.random.RandomState(2)
    for _ in range(200):
        n = rs.normal(size=3); n /= np.linalg.norm(n)
        d = rs.uniform(1.0, 30.0)
        basis = np.linalg.svd(n[None, :])[2][1:]
        pts = (basis.T @ rs.uniform(-0.3, 0.3, (2, 5))).T - d * n + rs.normal(0, 0.002, (5, 3))
        pts = pts.astype(np.float32)
        sq = np.sort(rs.uniform(0.01, 0.5, 5)).astype(np.float32)
        n4, ok = oracle.plane_fit(pts, sq)
        x = np.linalg.lstsq(pts.astype(np.float64), -np.ones(5), rcond=None)[0]
        ref = np.append(x / np.linalg.norm(x), 1.0 / np.linalg.norm(x))
        np.testing.assert_allclose(n4, ref, rtol=0, atol=5e-3 * max(1.0, abs(ref[3]) * 1e-2))
        assert ok
    # gates: 5th squared distance >= MAX_DIST_PLANE (compared un-squared, Plane.cpp:47), too few points, bent set
    n4, ok = oracle.plane_fit(pts, np.array([0.1, 0.2, 0.3, 0.4, 2.0], np.float32)); assert not ok
    n4, ok = oracle.plane_fit(pts[:4], sq[:4]); assert not ok
    bent = pts.copy(); bent[2] += (n * 0.2).astype(np.float32)
    n4, ok = oracle.plane_fit(bent, sq); assert not ok


def _np_pass(x26, P, H, h, R=0.001):
    """Dense numpy re-derivation of ONE first pass (x == x_prop, so every re-projection is the identity)."""
    HTH = H.T @ H
    Pt = np.linalg.inv(P / R)
    Pt[:12, :12] += HTH
    Pinv = np.linalg.inv(Pt)
    return Pinv[:, :12] @ (H.T @ h)


def test_ieskf_first_pass_matches_numpy(oracle):
    rs = np.random.RandomState(5)
    x0 = oracle.identity_x26(pos=(1.0, -2.0, 0.5))
    P = np.eye(23)
    P[6:12, 6:12] *= 1e-6
    M = 400
    n = rs.normal(size=(M, 3)); n /= np.linalg.norm(n, axis=1, keepdims=True)
    H = np.hstack([n, rs.normal(size=(M, 3)) * 3, rs.normal(size=(M, 3)), n])
    h = rs.normal(size=M) * 0.02
    # max_iters = 0  =>  exactly one pass
    x1, P1, npass = oracle.eskf_update_fixed(x0, P, H, h, max_iters=0)
    assert npass == 1
    dx_ref = _np_pass(x0, P, H, h)
    dx = np.zeros(23)
    import oracle_py as O
    O.lib().oracle_state_boxminus(np.ascontiguousarray(x1), np.ascontiguousarray(x0), dx)
    np.testing.assert_allclose(dx, dx_ref, rtol=1e-8, atol=1e-12)


def test_ieskf_degenerate_scene_is_frozen(oracle):
    """single-plane scene: 3 of the 6 pose eigenvalues are < D; the reference's row-zeroing projector
    (esekfom.hpp:1741-1744) then alters the pose step; with HTH == 0 (M < 23) the pose does not move."""
    x0 = oracle.identity_x26()
    P = np.eye(23)
    H = np.zeros((10, 12)); H[:, 2] = 1.0; H[:, 11] = 1.0
    h = np.full(10, 0.05)
    x1, _, _ = oracle.eskf_update_fixed(x0, P, H, h)      # M < 23 branch, HTH defined as 0
    np.testing.assert_array_equal(x1[0:7], x0[0:7])


def test_eigen_solver_restatement(oracle):
    """Eigen::EigenSolver<Matrix6d> restated from its published algorithm (rl_linalg.h: Householder Hessenberg, Francis
    double-shift QR, back substitution).  What can be checked without Eigen: eigenpairs against LAPACK (numpy), unit columns, and
    the ORDER properties that follow from the algorithm -- a diagonal matrix keeps its diagonal order (no reflector acts), a
    matrix that is already upper Hessenberg / tridiagonal is deflated from the bottom right, and the order is not sorted in general."""
    rs = np.random.RandomState(0)
    n_unsorted = 0
    for trial in range(1500):
        B = rs.randn(rs.randint(3, 40), 6) * rs.uniform(0.1, 30)
        if trial % 5 == 0:
            B[:, rs.choice(6, rs.randint(1, 6), replace=False)] *= 1e-4         # nearly degenerate directions
        if trial % 7 == 0:
            B = B[:rs.randint(1, 6)]                                           # rank deficient (corridor / open field)
        A = B.T @ B
        wr, wi, V = oracle.eigen_solver6(A)
        ref = np.linalg.eigvalsh(A)
        scale = max(np.abs(ref).max(), 1e-300)
        assert np.abs(np.sort(wr) - ref).max() <= 1e-7 * scale
        assert np.abs(wi).max() <= 1e-7 * scale                                # a symmetric matrix: pairs only by rounding
        if not np.any(wi):
            assert np.abs(np.sort(wr) - ref).max() <= 1e-12 * scale
            assert np.abs(A @ V - V * wr[None, :]).max() <= 1e-10 * scale
            np.testing.assert_allclose(np.linalg.norm(V, axis=0), 1.0, atol=1e-12)
        n_unsorted += int(np.any(np.diff(wr) < 0) and np.any(np.diff(wr) > 0))
    assert n_unsorted > 500                                                    # neither ascending nor descending: an order of its own
    wr, wi, V = oracle.eigen_solver6(np.diag([3.0, 1.0, 2.0, 6.0, 5.0, 4.0]))
    np.testing.assert_array_equal(wr, [3.0, 1.0, 2.0, 6.0, 5.0, 4.0])
    np.testing.assert_array_equal(V, np.eye(6))
    wr, wi, V = oracle.eigen_solver6(np.zeros((6, 6)))
    np.testing.assert_array_equal(wr, 0.0); np.testing.assert_array_equal(V, np.eye(6))


def _plane_rows(rs, normals, M, lever=5.0):
    """H rows of points on planes with the given normals: [n, p x n, 0.., n] like calculate_H (Localizer.cpp:564-569)."""
    H = np.zeros((M, 12)); h = rs.normal(size=M) * 0.02
    for m in range(M):
        n = np.asarray(normals[m % len(normals)], float)
        p = rs.uniform(-lever, lever, 3)
        H[m, 0:3] = n; H[m, 3:6] = np.cross(p, n); H[m, 9:12] = n
    return H, h


def test_ieskf_degenerate_projector_follows_the_solver_order(oracle):
    """Corridor (two wall normals + floor, nothing along the axis) and single-plane scenes: eigenvalues of HTH[0:6,0:6] below D = 5.
    The reference zeroes ROW i of the eigenvector matrix for eigenVALUE i (esekfom.hpp:1741) and applies VEPs^-1 * selVEPs: the
    step that results is a function of the solver's eigenpair order.  Checked: (a) the oracle's first-pass step equals the dense
    numpy evaluation of the same formula on the oracle's eigenpairs, (b) with a sorted order (what a Jacobi or LAPACK solver
    gives) the step is DIFFERENT -- the order is observable, which is why the solver is restated."""
    rs = np.random.RandomState(5)
    scenes = {"corridor": [(0, 1, 0), (0, -1, 0), (0, 0, 1)], "single plane": [(0, 0, 1)], "two planes": [(0, 0, 1), (0, 1, 0)]}
    differs = 0
    for name, normals in scenes.items():
        H, h = _plane_rows(rs, normals, 400)
        x0 = oracle.identity_x26(pos=(0.5, -0.2, 0.1))
        P = np.eye(23) * 1e-2
        x1, P1, n_pass = oracle.eskf_update_fixed(x0, P, H, h, max_iters=0)          # ONE pass (it = -1)
        HTH = H.T @ H
        wr, wi, V = oracle.eigen_solver6(HTH[:6, :6])
        assert (wr < 5.0).sum() >= 1, name                                            # degenerate indeed
        # dense evaluation of the pass (Appendix A of SURVEY.md) with the projector built from (wr, V)
        R = 0.001
        Pinv = np.linalg.inv(np.linalg.inv(P / R) + np.pad(HTH, ((0, 11), (0, 11))))
        dx = Pinv[:, :12] @ (H.T @ h)                                                 # dx_new = 0 at the first pass
        def projected(w, Vm):
            if np.prod(w) < 1e-20:
                Vm = np.eye(6)
            sel = Vm.copy(); sel[w < 5.0, :] = 0.0
            return np.linalg.inv(Vm) @ sel @ dx[:6]
        got = x1[0:3] - x0[0:3]
        want = projected(wr, V)[0:3]
        np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-9, err_msg=name)
        ws, Vs = np.linalg.eigh(HTH[:6, :6])                                          # ascending order, LAPACK's signs
        if np.abs(projected(ws, Vs)[0:3] - want).max() > 1e-6:
            differs += 1
    assert differs >= 1


def test_known_answer_noise_free(oracle):
    """noise-free box world + known offset T*  =>  the filter converges to T* (SURVEY.md 8 c (3))."""
    mp, scan, imu = cfg1_scene(sigma=0.0)
    W = _NoInsert(oracle, **CAPS)
    assert drive_two_scans(W, mp, scan, imu) == [1, 0]
    x = W.L.get_x()
    R = synth.rpy_to_R(*[math.radians(v) for v in synth.T_STAR_RPY_DEG])
    from scipy.spatial.transform import Rotation as Rot
    q = Rot.from_matrix(R).as_quat()
    x_true = x.copy(); x_true[0:3] = synth.T_STAR_T; x_true[3:7] = q
    dpos, ang = pose_delta(x, x_true)
    assert dpos < 1e-3 and ang < 2e-4, (dpos, ang)      # 4 passes from a 0.36 m / 1.2 deg offset
    assert all(p["M"] > 3500 for p in W.L.iters())


def test_first_scan_is_null_and_unmapped_scan_does_not_move(oracle):
    mp, scan, imu = cfg1_scene(n_map=2000, n_scan=512)
    L = oracle.Localizer(oracle.default_cfg(num_threads=1, **CAPS))
    st, w, a = imu
    i = 0
    while st[i] <= 0.105:
        L.update_imu(st[i], w[i], a[i]); i += 1
    assert L.update_pointcloud(scan, 0.0) == 1                        # a-note 8: first scan never registers
    while st[i] <= 0.205:
        L.update_imu(st[i], w[i], a[i]); i += 1
    x_before = L.get_x()
    assert L.update_pointcloud(scan, 0.1) == 0                        # no map yet: M = 0, pose untouched, map seeded
    np.testing.assert_allclose(L.get_x()[0:7], x_before[0:7], atol=1e-12)
    assert L.map_size() == 512


@pytest.mark.parametrize("sensor", ["OUSTER", "VELODYNE", "HESAI", "LIVOX"])
def test_input_filters_and_time_formats_against_numpy(oracle, sensor):
    """Oracle restatement of Localizer.cpp:262-302 (NaN removal, negative crop box, min distance, every-n-th point of the
    cropped cloud, FoV) and :745-781 (per-sensor time decoding) against a plain numpy statement of the same rules.  With
    a motionless sensor and distinct time stamps, pc2match is the filtered cloud in time order."""
    code = {"OUSTER": 0, "VELODYNE": 1, "HESAI": 2, "LIVOX": 3}[sensor]
    mp, scan5, imu = cfg1_scene(n_scan=3000)
    st, w, a = imu
    rs = np.random.RandomState(4)
    xyz = scan5[:, :3].copy()
    xyz[::53] = np.nan
    xyz[1::101] *= np.float32(0.02)
    rel = (rs.permutation(xyz.shape[0]).astype(np.float64) + 0.25) / xyz.shape[0] * 0.1     # distinct, shuffled
    filt = dict(crop_active=1, dist_active=1, min_dist=2.0, rate_active=1, rate_value=2, fov_active=1, fov_angle=2.5)
    L = oracle.Localizer(oracle.default_cfg(sensor_type=code, crop_min=(-1, -1, -1), crop_max=(1, 1, 1), num_threads=1,
                                            **filt, **CAPS))
    L.map_add(mp)
    i = 0
    for until, start in ((0.105, 0.0), (0.205, 0.1)):
        while i < len(st) and st[i] <= until:
            L.update_imu(st[i], w[i], a[i]); i += 1
        if sensor == "OUSTER":
            pts = oracle.make_points(xyz, 1.0, t_ns=np.round(rel * 1e9).astype(np.uint32))
        elif sensor == "VELODYNE":
            pts = oracle.make_points(xyz, 1.0, time_s=rel.astype(np.float32))
        elif sensor == "HESAI":
            pts = oracle.make_points(xyz, 1.0, timestamp=start + rel)
        else:
            pts = oracle.make_points(xyz, 1.0, timestamp=(start + rel) * 1e9)
        rc = L.update_pointcloud_points(pts, start, add_to_map=False)
    assert rc == 0
    # numpy statement of the filters
    fin = np.isfinite(xyz).all(1)
    p = xyz[fin]; t = rel[fin]
    inside = np.all((p >= -1) & (p <= 1), axis=1)
    p, t = p[~inside], t[~inside]
    keep = (np.sqrt((p.astype(np.float32) ** 2).sum(1)) > np.float32(2.0)) & (np.arange(p.shape[0]) % 2 == 0) & \
           (np.abs(np.arctan2(p[:, 1], p[:, 0])) < np.float32(2.5))
    p, t = p[keep], t[keep]
    order = np.argsort(t, kind="stable")
    got = L.pc2match()
    assert got.shape[0] == p.shape[0] and 300 < p.shape[0] < 1500
    np.testing.assert_allclose(got, p[order], rtol=0, atol=2e-5)      # deskew of a motionless sensor: identity up to float32 rounding
