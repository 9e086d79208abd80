"""Where the longest waves of one k_integrate launch spend their time: per-batch wall-clock stamps of a build with -DKFH_INT_STAMPS=1
(KFH_INTEGRATE_TRACE dump, 32 words per wave): python profiles/int_stamps.py <file> [batches per wave]"""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 32)
a = a[a[:, 1] > 0]
MASK = (1 << 60) - 1
t0 = int(a[:, 0].min())
us = lambda v: (int(v & MASK) - t0) / 100.0
start = np.array([us(v) for v in a[:, 0]]); end = np.array([us(v) for v in a[:, 1]])
dur = end - start
print("waves %d, span %.1f us" % (len(a), end.max()))
per_stage = {0: [], 1: [], 2: [], 3: []}
heads, replays = [], []
rows = []
for i in range(len(a)):
    r = a[i]
    if int(r[5]) == 0:
        continue
    head = us(r[4]) - start[i]; replay = us(r[5]) - us(r[4])
    stamps = [(k, int(r[k])) for k in range(6, 31) if int(r[k]) != 0]
    if not stamps:
        continue
    heads.append(head); replays.append(replay)
    times = [us(v) for _, v in stamps] + [us(r[31]) if int(r[31]) else end[i]]
    stages = [int(v >> 60) for _, v in stamps[1:]] + [int(int(r[31]) >> 60)]
    b = [(times[j + 1] - times[j], stages[j]) for j in range(len(stamps))]
    for d, st in b:
        per_stage[st].append(d)
    rows.append((dur[i], start[i], head, replay, b, end[i] - times[-1], int(r[2])))
print("waves that entered the batch loop: %d; head (entry -> before the replay) mean %.2f p90 %.2f us; replay mean %.2f p90 %.2f max %.2f us"
      % (len(rows), np.mean(heads), np.percentile(heads, 90), np.mean(replays), np.percentile(replays, 90), np.max(replays)))
names = {0: "rejected on its end voxels", 1: "detailed pass, no pixel", 2: "depths read, nothing to update", 3: "updated"}
for st in range(4):
    v = np.array(per_stage[st])
    if len(v):
        print("  batches %-32s %7d  mean %5.2f p50 %5.2f p90 %5.2f max %6.2f us  (sum %.0f us)" % (names[st], len(v), v.mean(), np.median(v), np.percentile(v, 90), v.max(), v.sum()))
rows.sort(key=lambda x: -x[0])
print("the 12 longest waves: lifetime | start | head | replay | batches (us, stage) | after the loop | work")
for d, st, head, replay, b, tail, work in rows[:12]:
    print("  %6.2f | %5.2f | %4.2f | %5.2f | %s | %4.2f | %d" % (d, st, head, replay, " ".join("%.1f/%d" % x for x in b), tail, work))
heavy = [r for r in rows if r[0] >= np.percentile([x[0] for x in rows], 95)]
print("the longest 5 %% of the waves (%d): lifetime %.2f = head %.2f + replay %.2f + updated batches %.2f (%.1f of them) + other batches %.2f + after the loop %.2f us"
      % (len(heavy), np.mean([r[0] for r in heavy]), np.mean([r[2] for r in heavy]), np.mean([r[3] for r in heavy]),
         np.mean([sum(d for d, s in r[4] if s == 3) for r in heavy]), np.mean([sum(1 for d, s in r[4] if s == 3) for r in heavy]),
         np.mean([sum(d for d, s in r[4] if s != 3) for r in heavy]), np.mean([r[5] for r in heavy])))
