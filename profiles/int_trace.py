"""Lifetimes of the waves of one k_integrate launch from the KFH_INTEGRATE_TRACE dump (100 MHz wall-clock stamps per wave,
{start, end, work, block}): python profiles/int_trace.py <file>.  work = batches examined + 4 x batches updated (busiest lane)."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 4)
ok = a[:, 1] > 0
a = a[ok]
t0 = a[:, 0].min()
st, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - t0) / 100.0
dur = en - st
span = en.max()
w = a[:, 2].astype(np.int64)
print("waves %d  span %.1f us  lifetime mean %.2f p50 %.2f p90 %.2f p99 %.2f max %.2f us   sum of lifetimes %.0f us (= %.1f waves busy on average)"
      % (len(a), span, dur.mean(), *np.percentile(dur, [50, 90, 99]), dur.max(), dur.sum(), dur.sum() / span))
print("start: p50 %.1f p90 %.1f p99 %.1f max %.1f us" % tuple(np.percentile(st, [50, 90, 99, 100])))
for f in np.linspace(0.1, 1.0, 10):
    t = span * f
    print("t=%6.1f us  running %5d  finished %6d  not started %6d" % (t, int(((st <= t) & (en > t)).sum()), int((en <= t).sum()), int((st > t).sum())))
edges = [0, 1, 5, 9, 17, 33, 49, 65, 81, 1000]
for lo, hi in zip(edges[:-1], edges[1:]):
    m = (w >= lo) & (w < hi)
    if m.any():
        print("  work [%3d,%4d): %6d waves  lifetime mean %7.2f p90 %7.2f max %7.2f us  start mean %6.1f  share of wave-time %4.1f %%"
              % (lo, hi, m.sum(), dur[m].mean(), np.percentile(dur[m], 90), dur[m].max(), st[m].mean(), 100 * dur[m].sum() / dur.sum()))
