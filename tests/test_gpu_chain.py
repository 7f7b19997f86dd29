"""GPU parity: the iterations of the update enqueued at once (flimo_update_chain: the filter's algebra inside each pass's reducing
launch, flimo_chain.h / flimo_ieskf.h) against the host loop over single passes (flimo_match_reduce + flimo_host::Esekf, the layout of
rounds 1-3, kept behind FLIMO_HOST_UPDATE=1) and against the golden per-pass vectors.  Reference: esekf::
update_iterated_dyn_share_modified, IKFoM_toolkit/esekfom/esekfom.hpp:1620-1823."""
import os

import numpy as np
import pytest

from common import CAPS, cfg1_scene, drive_two_scans, pose_delta
from fast_limo_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "cfg1_golden.npz")


def _localizer(host_update, **kw):
    """A Localizer whose context reads FLIMO_HOST_UPDATE at creation."""
    from fast_limo_amd import api
    old = os.environ.get("FLIMO_HOST_UPDATE")
    os.environ["FLIMO_HOST_UPDATE"] = "1" if host_update else "0"
    try:
        L = api.Localizer(api.default_cfg(**CAPS, **kw))
    finally:
        if old is None:
            del os.environ["FLIMO_HOST_UPDATE"]
        else:
            os.environ["FLIMO_HOST_UPDATE"] = old
    return L


def test_chain_equals_the_host_loop_pass_by_pass(built):
    """Same scene through the chained update and through the host loop: every pass's M, sums, step and state, the final state and
    covariance.  The two run the same arithmetic in the same order; what differs is libm (sin / cos / atan of doubles)."""
    mp, scan5, imu = cfg1_scene()
    D = _localizer(False); H = _localizer(True)
    for L in (D, H):
        L.set_flags(add_to_map=False, keep_log=True)
        assert drive_two_scans(L, mp, scan5, imu) == [1, 0]
    cs = D.hip.chain_stats()
    assert cs["chains"] >= 1 and cs["handed_back"] == 0, cs
    assert H.hip.chain_stats()["chains"] == 0
    pd, ph = D.passes(), H.passes()
    assert len(pd) == len(ph) >= 2
    assert [p["M"] for p in pd] == [p["M"] for p in ph]
    for a, b in zip(pd, ph):
        np.testing.assert_allclose(a["HTH"], b["HTH"], rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(a["HTh"], b["HTh"], rtol=1e-12, atol=1e-9)
        np.testing.assert_allclose(a["dx"], b["dx"], rtol=0, atol=1e-13)
        np.testing.assert_allclose(a["x_after"], b["x_after"], rtol=0, atol=1e-13)
    np.testing.assert_allclose(D.get_x(), H.get_x(), rtol=0, atol=1e-13)
    # (P = L - K_x P cancels from 1 to 1e-6: the last bits of the libm calls show up at 1e-11 absolute in its small entries)
    np.testing.assert_allclose(D.get_P(), H.get_P(), rtol=1e-9, atol=1e-10)
    # and the golden vectors of the same scene
    g = np.load(GOLD)
    assert [p["M"] for p in pd] == list(g["M"])
    for p, HTH, dx in zip(pd, g["HTH"], g["dx"]):
        np.testing.assert_allclose(p["HTH"], HTH, rtol=1e-11, atol=1e-8)
        np.testing.assert_allclose(p["dx"], dx, rtol=0, atol=1e-9)
    D.close(); H.close()


def test_chain_is_bit_reproducible_and_counts_its_passes(built):
    mp, scan5, imu = cfg1_scene(n_map=200000, n_scan=16384, L=40.0)
    xs = []
    for _ in range(2):
        D = _localizer(False)
        D.set_flags(add_to_map=False, keep_log=True)
        assert drive_two_scans(D, mp, scan5, imu) == [1, 0]
        n0 = D.hip.pass_count()
        xs.append((D.get_x().copy(), D.get_P().copy(), len(D.passes())))
        assert n0 == len(D.passes())        # the context's pass count = passes whose measurement ran
        D.close()
    assert xs[0][2] == xs[1][2]
    np.testing.assert_array_equal(xs[0][0], xs[1][0])
    np.testing.assert_array_equal(xs[0][1], xs[1][1])


def test_chain_hands_back_what_it_does_not_run(built, oracle):
    """M < 23 (a scan with a handful of usable points) and a degenerate corridor: the device stops at that iteration and the host
    filter finishes the update; the result equals the host loop's."""
    mp = synth.box_world_map(30000, 15.0, 3)
    imu = synth.stationary_imu(0.0, 0.35)
    scan = synth.box_world_scan_random(2048, 15.0, 4)
    few = scan.copy()
    few[12:, :3] += 500.0                       # all but 12 points far off the map: M < 23
    res = []
    for host in (False, True):
        L = _localizer(host)
        L.set_flags(add_to_map=False, keep_log=True)
        assert drive_two_scans(L, mp, scan, imu, second_scan=few) == [1, 0]
        res.append((L.get_x().copy(), L.get_P().copy(), [p["M"] for p in L.passes()], L.hip.chain_stats()))
        L.close()
    (xd, Pd, Md, sd), (xh, Ph, Mh, sh) = res
    assert sd["chains"] >= 1 and sd["handed_back"] >= 1, sd
    assert Md == Mh and max(Md) < 23
    np.testing.assert_allclose(xd, xh, rtol=0, atol=1e-13)
    np.testing.assert_allclose(Pd, Ph, rtol=1e-9, atol=1e-10)


def test_raw_chain_entry_point(built):
    """C ABI directly: caps that bind need the per-point records -> FLIMO_CHAIN_DECLINED and nothing is changed; otherwise the loop comes
    back at the iteration that ends it, with that iteration's sums."""
    from fast_limo_amd import _lib
    mp, scan5, _ = cfg1_scene()
    h = _lib.HipCtx()
    h.set_update_mode(2)                          # the chain, whatever this host's launch round trip
    m = h.update_mode()
    assert m["chained"] and 0.5 < m["launch_rtt_us"] < 1000.0, m
    h.map_add(np.ascontiguousarray(mp[:, :3]))
    h.scan_set(np.ascontiguousarray(scan5[:, :3]))
    x = np.zeros(26); x[6] = 1.0; x[10] = 1.0; x[25] = -9.809
    P = np.eye(23) * 1e-2
    lim = np.full(23, 1e-3)
    n0 = h.pass_count()
    r = h.update_chain(_lib.default_match_cfg(MAX_NUM_MATCHES=100, MAX_NUM_PC2MATCH=10**7), x, P, lim)
    assert r["status"] == 0 and h.pass_count() == n0
    cfg = _lib.default_match_cfg(MAX_NUM_MATCHES=10**7, MAX_NUM_PC2MATCH=10**7)
    r = h.update_chain(cfg, x, P, lim, max_iter=3)
    assert r["status"] == 2 and r["reason"] == 5 and 0 <= r["passes"] <= 3, r
    assert h.pass_count() == n0 + r["passes"] + 1 and r["it_next"] == r["passes"] - 1
    assert r["meas"] is not None and r["meas"]["M"] == r["log"][r["passes"]]["M"] > 1000
    assert all(p["M"] > 1000 for p in r["log"])
    # the handed-back iteration's pass once more, through flimo_match_reduce at the handed-back state: the same matches; the sums
    # agree to rounding (the two may split the scan into different partial sums)
    HTH, HTh, M = h.match_reduce(r["x"], cfg)
    assert M == r["meas"]["M"]
    np.testing.assert_allclose(HTH, r["meas"]["HTH"], rtol=1e-12, atol=1e-9)
    np.testing.assert_allclose(HTh, r["meas"]["HTh"], rtol=1e-11, atol=1e-9)
    h.close()


def test_pipelined_host_loop_is_bit_equal_and_lets_its_last_pass_go(built, monkeypatch):
    """Host loop, pipelined (flimo_set_pass_pipeline; fast_limo::Localizer switches it on): the next pass of an update waits on the GPU
    for its pose, which the next flimo_match_reduce stores into device memory instead of launching.  Same kernels, same inputs: state,
    covariance and every pass's dx bit-equal to the loop that launches each pass when its pose is known (FLIMO_PIPELINE=0), over a
    drive with map inserts; passes are found waiting; the pass queued behind an update's last iteration is told to leave (the drive
    would otherwise take 50 ms per scan).  C ABI: the same two passes with and without the switch, then other work on the context."""
    import time
    from fast_limo_amd import _lib, api
    n_scans, n_pts, speed = 6, 30000, 10.0
    st, w, a = synth.stationary_imu(0.0, 0.1 * n_scans + 0.06)
    out = {}
    for label, env in (("pipelined", None), ("plain", "0")):
        if env is None:
            monkeypatch.delenv("FLIMO_PIPELINE", raising=False)
        else:
            monkeypatch.setenv("FLIMO_PIPELINE", env)
        G = _localizer(True)
        monkeypatch.delenv("FLIMO_PIPELINE", raising=False)
        G.set_flags(add_to_map=True, keep_log=True)
        x0 = G.get_x(); x0[14] = speed; G.set_x(x0)
        i, res = 0, []
        t0 = time.perf_counter()
        for k in range(n_scans):
            until = 0.1 * (k + 1) + 0.005
            while i < len(st) and st[i] <= until:
                G.update_imu(st[i], w[i], a[i]); i += 1
            rc = G.update_pointcloud(synth.corridor_scan(k, n_pts, 77, speed=speed), 0.1 * k)
            res.append((rc, G.get_x().copy(), G.get_P().copy(), [p["dx"].copy() for p in G.passes()], G.map_size()))
        out[label] = (res, time.perf_counter() - t0, G.hip.pass_pipeline_stats())
        G.close()
    for ra, rb in zip(out["pipelined"][0], out["plain"][0]):
        assert ra[0] == rb[0] and ra[4] == rb[4]
        np.testing.assert_array_equal(ra[1], rb[1])
        np.testing.assert_array_equal(ra[2], rb[2])
        assert len(ra[3]) == len(rb[3])
        for da, db in zip(ra[3], rb[3]):
            np.testing.assert_array_equal(da, db)
    sp = out["pipelined"][2]
    assert sp["published"] >= n_scans - 2 and out["plain"][2]["published"] == 0, (sp, out["plain"][2])
    assert sp["cancelled"] >= 1                                      # (a pass was queued behind an update's last iteration, and let go)
    assert out["pipelined"][1] < out["plain"][1] + 0.04, (out["pipelined"][1], out["plain"][1])   # nothing waited for a pass to give up (50 ms each)
    # ---- C ABI ----
    mp, scan5, imu = cfg1_scene()
    cfg = _lib.default_match_cfg(MAX_NUM_MATCHES=10**7, MAX_NUM_PC2MATCH=10**7)
    x = np.zeros(26); x[6] = 1.0; x[10] = 1.0; x[25] = -9.809
    xs = [x.copy() for _ in range(3)]
    xs[1][0] += 0.01; xs[2][0] += 0.015; xs[2][1] -= 0.004
    got = {}
    for on in (False, True):
        h = _lib.HipCtx()
        h.set_update_mode(1)
        h.map_add(np.ascontiguousarray(mp[:, :3]))
        h.scan_set(np.ascontiguousarray(scan5[:, :3]))
        h.set_pass_pipeline(on)
        got[on] = []
        for j, xk in enumerate(xs):
            if j == len(xs) - 1:
                h.pass_pipeline_last()                                # (the filter's hint: no pass can follow this one)
            got[on].append(h.match_reduce(xk, cfg))
        h.pass_pipeline_end()
        s_ = h.pass_pipeline_stats()
        assert (s_["published"] >= 1) == on, s_
        assert s_["cancelled"] == 0, s_                               # nothing was queued behind the last pass: nothing to let go
        t0 = time.perf_counter()
        h.map_add(np.ascontiguousarray(mp[:1000, :3] + 0.01))         # other work on the context: nothing in its way
        assert time.perf_counter() - t0 < 0.15
        h.close()
    for ra, rb in zip(got[False], got[True]):
        assert ra[2] == rb[2]
        np.testing.assert_array_equal(ra[0], rb[0])
        np.testing.assert_array_equal(ra[1], rb[1])


@pytest.mark.parametrize("host_loop", [False, True])
def test_a_pass_that_fails_is_surfaced(built, host_loop):
    """The reference's Mapper::match cannot fail (Modules/Mapper.cpp:59-86); a GPU pass can (timeout, HIP error).  C ABI: the wait
    bound 0 ("do not wait") makes flimo_match_reduce and flimo_update_chain return FLIMO_ERR_TIMEOUT while their launches are
    still queued; nothing new is queued on top of them until they are gone.  Localizer: the update is abandoned -- status -4, state
    and covariance back at the propagated values, no map insert -- and the next sweep registers normally."""
    import time
    from fast_limo_amd import _lib, api
    mp, scan5, imu = cfg1_scene()
    h = _lib.HipCtx()
    h.set_update_mode(2)
    h.set_pass_pipeline(host_loop)                # (host loop: its passes are queued ahead of their poses, as the Localizer queues them)
    h.map_add(np.ascontiguousarray(mp[:, :3]))
    h.scan_set(np.ascontiguousarray(scan5[:, :3]))
    x = np.zeros(26); x[6] = 1.0; x[10] = 1.0; x[25] = -9.809
    cfg = _lib.default_match_cfg(MAX_NUM_MATCHES=10**7, MAX_NUM_PC2MATCH=10**7)
    ref = h.match_reduce(x, cfg)
    h.set_wait_timeout_ms(0)
    with pytest.raises(_lib.FlimoError, match="TIMEOUT"):
        h.match_reduce(x, cfg)
    with pytest.raises(_lib.FlimoError, match="TIMEOUT"):               # still running (refused), or gone: then this one times out
        h.update_chain(cfg, x, np.eye(23) * 1e-2, np.full(23, 1e-3))
    h.set_wait_timeout_ms(2000)
    for _ in range(200):                                               # the abandoned launches drain within microseconds
        try:
            got = h.match_reduce(x, cfg)
            break
        except _lib.FlimoError:
            time.sleep(0.001)
    assert got[2] == ref[2]
    np.testing.assert_allclose(got[0], ref[0], rtol=1e-12, atol=1e-9)
    h.close()
    # ---- through the Localizer ----
    # (both layouts of the update: the chain, and the pipelined host loop every measured host runs by default -- its abort path is
    #  fast_limo.cpp's h_share_model failure -> Esekf::update leaves with the propagated state)
    st, w, a = imu
    L = _localizer(host_loop)
    assert L.hip.update_mode()["chained"] == (0 if host_loop else 1)
    L.set_flags(add_to_map=True, keep_log=False)
    L.map_add(mp)
    i = 0
    def feed(until):
        nonlocal i
        while i < len(st) and st[i] <= until:
            L.update_imu(st[i], w[i], a[i]); i += 1
    feed(0.105); assert L.update_pointcloud(scan5, 0.0) == 1           # null iteration (reference a-note 8)
    feed(0.205); assert L.update_pointcloud(scan5, 0.1) == 0
    L.sync()
    n_map, x_ok = L.map_size(), L.get_x().copy()
    feed(0.305)
    L.hip.set_wait_timeout_ms(0)
    rc = L.update_pointcloud(scan5, 0.2)
    assert rc == -4, rc                                                # the update was abandoned
    L.sync()
    assert L.map_size() == n_map                                       # no insert at an unmeasured pose
    assert np.isfinite(L.get_x()).all() and np.abs(L.get_x()[0:3] - x_ok[0:3]).max() < 0.05      # the propagated state
    L.hip.set_wait_timeout_ms(2000)
    time.sleep(0.05)
    feed(0.405)
    rc = L.update_pointcloud(scan5, 0.3)
    assert rc == 0, rc                                                 # the next sweep registers normally
    L.sync()
    assert L.map_size() >= n_map
    dpos, ang = pose_delta(L.get_x(), x_ok)
    assert dpos < 1e-2 and ang < 1e-2
    L.close()


def test_a_waiting_pass_that_ages_or_leaves_is_launched_again(built, monkeypatch):
    """Pipelined host loop: the pass queued ahead of its pose gives up after CH_POLL_MS = 50 ms (flimo_chain.h).  (1) A caller that comes
    back late (here: 30 ms between two passes, beyond the 12.5 ms age bound) finds the pass too old, tells it to leave and launches the
    usual way.  (2) A host thread descheduled between its age check and its publish (FLIMO_TEST_PUBLISH_DELAY_MS = 70 > 50) publishes to
    a launch that has left -- as a whole: the workgroups share one verdict (chain_enter's decision word), no ticket is touched -- and
    the pass is launched again.  Either way the sums are the plain loop's, bit for bit, and nothing waits for seconds."""
    import time
    from fast_limo_amd import _lib
    mp, scan5, imu = cfg1_scene()
    cfg = _lib.default_match_cfg(MAX_NUM_MATCHES=10**7, MAX_NUM_PC2MATCH=10**7)
    x = np.zeros(26); x[6] = 1.0; x[10] = 1.0; x[25] = -9.809
    xs = [x.copy() for _ in range(4)]
    xs[1][0] += 0.01; xs[2][0] += 0.015; xs[2][1] -= 0.004; xs[3][0] += 0.016
    def run(pipeline, pause_s, delay_ms):
        if delay_ms:
            monkeypatch.setenv("FLIMO_TEST_PUBLISH_DELAY_MS", str(delay_ms))
        h = _lib.HipCtx()
        monkeypatch.delenv("FLIMO_TEST_PUBLISH_DELAY_MS", raising=False)
        h.set_update_mode(1)
        h.map_add(np.ascontiguousarray(mp[:, :3]))
        h.scan_set(np.ascontiguousarray(scan5[:, :3]))
        h.set_pass_pipeline(pipeline)
        out, t0 = [], time.perf_counter()
        for xk in xs:
            out.append(h.match_reduce(xk, cfg))
            if pause_s:
                time.sleep(pause_s)
        dt = time.perf_counter() - t0
        h.pass_pipeline_end()
        st = h.pass_pipeline_stats()
        h.close()
        return out, st, dt
    plain, st0, _ = run(False, 0.0, 0)
    aged, st1, dt1 = run(True, 0.030, 0)
    if st1["published"] + st1["aged"] + st1["cancelled"] == 0:
        pytest.skip("this device does not map fine-grained memory for the host: the plain loop runs")
    assert st1["aged"] >= 2 and st1["left"] == 0, st1
    left, st2, dt2 = run(True, 0.0, 70)
    assert st2["left"] >= 2, st2                          # the launches had left as a whole; their passes were launched again
    for got in (aged, left):
        for ra, rb in zip(plain, got):
            assert ra[2] == rb[2]
            np.testing.assert_array_equal(ra[0], rb[0])
            np.testing.assert_array_equal(ra[1], rb[1])
    assert dt1 < 1.0 and dt2 < 1.5, (dt1, dt2)            # nobody sat out the 2 s wait bound


def test_a_device_without_host_writable_memory_runs_the_plain_loop(built, monkeypatch):
    """The pipelined host loop needs device memory the HOST stores into (fine-grained allocation behind a large BAR, found coherent by a
    probe at context creation).  A context without it (FLIMO_NO_BAR=1 stands in for such a system) runs the plain loop: a pass is
    launched when its pose is known -- same sums, bit for bit, nothing published."""
    from fast_limo_amd import _lib
    mp, scan5, imu = cfg1_scene()
    cfg = _lib.default_match_cfg(MAX_NUM_MATCHES=10**7, MAX_NUM_PC2MATCH=10**7)
    x = np.zeros(26); x[6] = 1.0; x[10] = 1.0; x[25] = -9.809
    xs = [x.copy() for _ in range(3)]
    xs[1][0] += 0.01; xs[2][0] += 0.015
    got = {}
    for nobar in (False, True):
        if nobar:
            monkeypatch.setenv("FLIMO_NO_BAR", "1")
        h = _lib.HipCtx()
        monkeypatch.delenv("FLIMO_NO_BAR", raising=False)
        h.set_update_mode(1)
        h.map_add(np.ascontiguousarray(mp[:, :3]))
        h.scan_set(np.ascontiguousarray(scan5[:, :3]))
        h.set_pass_pipeline(True)
        got[nobar] = ([h.match_reduce(xk, cfg) for xk in xs], h.pass_pipeline_stats())
        h.pass_pipeline_end()
        h.close()
    assert got[True][1]["published"] == 0 and got[True][1]["cancelled"] == 0, got[True][1]
    for ra, rb in zip(got[False][0], got[True][0]):
        assert ra[2] == rb[2]
        np.testing.assert_array_equal(ra[0], rb[0])
        np.testing.assert_array_equal(ra[1], rb[1])
