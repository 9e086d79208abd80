"""Round-4 GPU tests: the partition of an ICP level that is larger than one pass over the blocks.

A level of more than 256 x 1024 pixels is walked in full passes over the blocks plus a remainder that is dealt in 64-pixel chunks to
the blocks' WAVES (kf_icp.hip icp_track_block); at 640x480 the remainder is a whole number of chunks.  Here: sizes whose
remainder ends in a PARTIAL chunk and reaches the blocks' second, third and fourth waves."""
import numpy as np
import pytest

from tests import util
from tests.util import same_bits

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def kfmod(hip):
    from slambench_amd import kfusion
    return kfusion


@pytest.mark.parametrize("W,H", [(600, 452), (644, 484), (1000, 300)])
def test_icp_level_with_a_partial_remainder(hip, kfmod, orc, W, H):
    """(600, 452): 271 200 pixels = one pass + 141.5 chunks; (644, 484): 311 696 = one pass + 774.25 chunks (waves 0-3 of the blocks);
    (1000, 300): 300 000 = one pass + 591.5 chunks, rows that are not a multiple of anything.  trackKernel + reduceKernel of the
    reference (cpp:497-560, 251-495) through ONE level-0 association per frame of a single-level context, fed the oracle's poses:
    result map and the four counters exactly equal, the 28 float sums within the re-association bound; then the queued path
    against the phase calls of a three-level context, bit for bit (both use the same partition)."""
    from tests.test_hip_kernels import _reduce_magnitudes
    n = 6
    s = util.synth.make_stream(n, W, H)
    k = list(s.k)
    res = (96, 96, 96)
    o = orc.OracleKfusion((W, H), res, util.VOL_DIM, s.init_pos, (4, 3, 3), omp=True)
    g1 = kfmod.Kfusion((W, H), res, util.VOL_DIM, s.init_pos, (1,))
    o1 = orc.OracleKfusion((W, H), res, util.VOL_DIM, s.init_pos, (1,), omp=True)
    f_run = kfmod.Kfusion((W, H), res, util.VOL_DIM, s.init_pos, (4, 3, 3))
    q = kfmod.Kfusion((W, H), res, util.VOL_DIM, s.init_pos, (4, 3, 3))
    associated = 0
    for f in range(n):
        pose_in = o.get_pose()
        for x in (g1, o1):
            x.set_pose(pose_in)
        for x in (o, g1, o1):
            x.preprocessing(s.depth_mm[f])
        o.tracking(k, util.ICP_THRESHOLD, 1, f)
        g1.tracking(k, util.ICP_THRESHOLD, 1, f); o1.tracking(k, util.ICP_THRESHOLD, 1, f)
        codes = o1.buffer("trackingResult")["result"]
        assert np.array_equal(g1.buffer("trackResult"), codes), f
        rg, ro = g1.buffer("reduction"), o1.buffer("reduction")[0]
        assert np.array_equal(rg[28:32], ro[28:32]), (f, rg[28:32], ro[28:32])
        if f >= 4:
            assert ro[28] > 0.5 * W * H, f                     # a real case: most pixels associate
            mag = _reduce_magnitudes(o1.buffer("trackingResult"), W, H)
            assert np.all(np.abs(rg[:28].astype(np.float64) - ro[:28]) <= 2e-4 * mag[:28] + 1e-9), f
            associated += 1
        po = o.get_pose()
        for x in (g1, o1):
            x.set_pose(po)
        o.integration(k, 1, util.MU, f)
        g1.integration(k, 1, util.MU, f); o1.integration(k, 1, util.MU, f)
        assert np.array_equal(g1.buffer("volume"), o1.buffer("volume")), "TSDF differs at frame %d" % f
        for x in (o, g1, o1):
            x.raycasting(k, util.MU, f)
        assert same_bits(g1.buffer("vertex")[..., :3], o1.buffer("vertex")), f
        f_run.preprocessing(s.depth_mm[f]); f_run.tracking(k, util.ICP_THRESHOLD, 1, f); f_run.integration(k, 1, util.MU, f); f_run.raycasting(k, util.MU, f)
        q.compute_frame(s.depth_mm[f], (W, H), k, 1, 1, util.ICP_THRESHOLD, util.MU, f)
    assert associated == 2
    q.synchronise()
    for name in ("volume", "vertex", "normal", "trackResult", "pose"):
        assert np.array_equal(f_run.buffer(name).view(np.uint8), q.buffer(name).view(np.uint8)), name
    assert f_run.get_tracked() and np.abs(f_run.get_pose() - o.get_pose()).max() <= 1e-4
    for x in (o, g1, o1, f_run, q):
        x.close()


def test_nine_contexts_share_the_co_residency_units(hip, kfmod):
    """k_icp_tail's grid barrier needs its workgroups resident together, so a process hands out the device's 256 CUs in eight units of
    32 (kf_pipeline.hip create_common): the first context of a device takes two (its shared launches are 64 workgroups wide and carry
    level 0's fourth iteration), the next six one each, and the eighth and ninth find none left and get a launch per iteration.  The
    schedule is never an input of the computation: with every queued iteration forced to run (threshold 0: the late ones inside 64- and
    32-workgroup launches, or in launches of their own) all nine contexts, taking turns on the device, must produce the same bits - and
    the units must come back when the contexts go."""
    n = 8
    s = util.small_stream(n)
    k = list(s.k)
    res = (48, 48, 48)
    mk = lambda: kfmod.Kfusion((s.width, s.height), res, util.VOL_DIM, s.init_pos, (10, 5, 4))
    ctxs = [mk() for _ in range(9)]
    for f in range(n):
        for c in ctxs:
            c.compute_frame(s.depth_mm[f], (s.width, s.height), k, 1, 1, 0.0, util.MU, f)
    for c in ctxs:
        c.synchronise()
    assert ctxs[0].icp_level_iterations() == [10 * n, 5 * n, 4 * n]
    ref = {name: ctxs[0].buffer(name).copy() for name in ("pose", "volume", "vertex", "normal", "trackResult")}
    for i, c in enumerate(ctxs[1:], 1):
        assert c.icp_level_iterations() == ctxs[0].icp_level_iterations(), i
        assert c.barrier_timeouts() == 0, i
        for name, want in ref.items():
            assert np.array_equal(c.buffer(name).view(np.uint8), want.view(np.uint8)), (i, name)
    for c in ctxs:
        c.close()
    again = mk()                        # the units are back: this one is a first context again and runs like one
    for f in range(n):
        again.compute_frame(s.depth_mm[f], (s.width, s.height), k, 1, 1, 0.0, util.MU, f)
    again.synchronise()
    for name, want in ref.items():
        assert np.array_equal(again.buffer(name).view(np.uint8), want.view(np.uint8)), name
    again.close()
