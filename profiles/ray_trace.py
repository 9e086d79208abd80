"""Occupancy profile of one k_raycast launch from the KFH_RAY_TRACE dump (100 MHz wall-clock stamps per wave):
python profiles/ray_trace.py <file>"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
idx = np.arange(len(a))
ok = a[:, 1] > 0
a, idx = a[ok], idx[ok]
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0      # microseconds
dur = en - st
span = en.max()
print("waves %d  span %.1f us  lifetime mean %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f us" % (len(a), span, dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max()))
print("start: p10 %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f us" % tuple(np.percentile(st, [10, 50, 90, 99, 100])))
for f in np.linspace(0.1, 1.0, 10):
    t = span * f
    print("t=%5.1f us  running %5d  finished %5d  not started %5d" % (t, int(((st <= t) & (en > t)).sum()), int((en <= t).sum()), int((st > t).sum())))
late = np.argsort(en)[-8:]
print("last finishers: " + "; ".join("wave %d id %d work %d start %.1f dur %.1f" % (idx[i], a[i, 3], a[i, 2], st[i], dur[i]) for i in late))
w = a[:, 2].astype(np.float64)
print("work (rounds*Q of the wave's longest ray): mean %.1f p90 %.0f p99 %.0f max %.0f; corr(work, lifetime) %.3f" % (w.mean(), *np.percentile(w, [90, 99]), w.max(), np.corrcoef(w, dur)[0, 1]))
for lo, hi in ((0, 24), (24, 40), (40, 60), (60, 90), (90, 400)):
    m = (w >= lo) & (w < hi)
    if m.any():
        print("  work [%3d,%3d): %5d waves, lifetime mean %.1f max %.1f us, start mean %.1f" % (lo, hi, m.sum(), dur[m].mean(), dur[m].max(), st[m].mean()))

