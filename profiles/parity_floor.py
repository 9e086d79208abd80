"""Free-running parity against the oracle next to the reference's own noise floor (its CPP and OpenMP builds
differ because the OpenMP reduction is re-associated).  For each case: HIP vs the deterministic oracle, and the oracle with the
reference-style OpenMP reduction (orc.set_omp_reduce_mode) vs the deterministic oracle, on the same stream.
    python profiles/parity_floor.py [frames_at_640x480]
"""
import os
import sys

import numpy as np

sys.path.insert(0, ".")
from oracle import orc                      # noqa: E402
from slambench_amd import kfusion           # noqa: E402
from tests import util                      # noqa: E402

orc.build()


def tick(what, f, n):
    if n >= 200 and f % 100 == 99:
        print("  ... %s frame %d of %d" % (what, f + 1, n), file=sys.stderr, flush=True)      # (a silent GPU box is taken to be hung)


def frac(a, b):
    d = np.abs(a[..., 0].astype(np.float64) - b[..., 0].astype(np.float64)) / 32766.0
    return float(np.mean(d > 1e-4)), float(np.mean(a[..., 0] != b[..., 0]))


def case(name, n, W, H, res, pyramid=(10, 5, 4), ratio=1, irate=2):
    s = util.synth.make_stream(n, W, H)
    cw, ch = W // ratio, H // ratio
    k = [v / ratio for v in s.k]
    g = kfusion.Kfusion((cw, ch), res, util.VOL_DIM, s.init_pos, pyramid)
    orc.set_omp_reduce_mode(False)
    o = orc.OracleKfusion((cw, ch), res, util.VOL_DIM, s.init_pos, pyramid, omp=True)
    worst = 0.0
    for f in range(n):
        g.preprocessing(s.depth_mm[f]); g.tracking(k, util.ICP_THRESHOLD, 1, f); g.integration(k, irate, util.MU, f); g.raycasting(k, util.MU, f)
        o.preprocessing(s.depth_mm[f]); o.tracking(k, util.ICP_THRESHOLD, 1, f); o.integration(k, irate, util.MU, f); o.raycasting(k, util.MU, f)
        worst = max(worst, float(np.abs(g.get_pose() - o.get_pose()).max()))
        tick("HIP + serial oracle", f, n)
    vo = o.buffer("volume")
    fh = frac(g.buffer("volume"), vo)
    g.close(); o.close()
    orc.set_omp_reduce_mode(True)
    r = orc.OracleKfusion((cw, ch), res, util.VOL_DIM, s.init_pos, pyramid, omp=True)
    o2 = None
    worst_r = 0.0
    orc.set_omp_reduce_mode(True)
    for f in range(n):
        r.preprocessing(s.depth_mm[f]); r.tracking(k, util.ICP_THRESHOLD, 1, f); r.integration(k, irate, util.MU, f); r.raycasting(k, util.MU, f)
        tick("OpenMP-style reduction", f, n)
    fr = frac(r.buffer("volume"), vo)
    r.close()
    orc.set_omp_reduce_mode(False)
    # the same floor with a FIXED schedule and combine order (oracle "reduce emulation": reproducible, whatever libgomp does)
    fixed = {}
    for T in (2, 16):
        orc.set_reduce_emulation(T)
        try:
            e = orc.OracleKfusion((cw, ch), res, util.VOL_DIM, s.init_pos, pyramid, omp=True)
            for f in range(n):
                e.preprocessing(s.depth_mm[f]); e.tracking(k, util.ICP_THRESHOLD, 1, f); e.integration(k, irate, util.MU, f); e.raycasting(k, util.MU, f)
                tick("fixed schedule, %d threads" % T, f, n)
            fixed[T] = frac(e.buffer("volume"), vo)[0]
            e.close()
        finally:
            orc.set_reduce_emulation(0)
    print("%-28s frames %3d  HIP vs oracle: pose %.3g, voxels off by >1e-4: %.3g (any: %.3g)   reference-style OpenMP reduction vs serial order: >1e-4: %.3g (any: %.3g); "
          "fixed schedule, 2 / 16 threads: %.3g / %.3g" % (name, n, worst, fh[0], fh[1], fr[0], fr[1], fixed[2], fixed[16]), flush=True)


if __name__ == "__main__":
    n_big = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    if len(sys.argv) > 2 and sys.argv[2] == "only":
        case("640x480 256^3", n_big, 640, 480, (256,) * 3)
        sys.exit(0)
    case("160x120 64^3", 10, 160, 120, (64,) * 3)
    case("160x120 -c 2 48^3 (4,3)", 10, 160, 120, (48,) * 3, (4, 3), 2)
    case("160x120 32^3 (6,)", 10, 160, 120, (32,) * 3, (6,))
    case("640x480 256^3 7 frames", 7, 640, 480, (256,) * 3)
    case("640x480 256^3", n_big, 640, 480, (256,) * 3)
