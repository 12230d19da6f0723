"""Condenses a rocprofv3 --kernel-trace CSV of bench.py (two HIP streams) into: wall time of the last traced step, time with
0 / 1 / 2 queues busy, per-queue busy time and gaps, and the kernels that run alone on the critical queue.
usage: python tools/timeline.py <dir with *_kernel_trace.csv>"""
import collections
import csv
import glob
import re
import sys

rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"], r["Kernel_Name"]))
rows.sort()
# steps are delimited by adamw_kernel pairs; take the window between the last two "ce_kernel" launches (one per step)
ce = [i for i, r in enumerate(rows) if re.search(r"(^|[ :])ce_kernel", r[3])]
if len(ce) < 2:
    print("not enough steps in the trace"); sys.exit(0)
a, b = ce[-2], ce[-1]
win = rows[a:b]
t0, t1 = win[0][0], rows[b][0]
print("step window %.3f ms, %d launches, queues %s" % ((t1 - t0) * 1e-6, len(win), sorted(set(r[2] for r in win))))
ev = []
for s, e, q, k in win:
    ev.append((s, 1, q)); ev.append((min(e, t1), -1, q))
ev.sort()
busy = collections.Counter(); active = collections.Counter(); last = t0; hist = collections.Counter()
for t, d, q in ev:
    n = sum(1 for v in active.values() if v > 0)
    hist[n] += t - last
    last = t
    active[q] += d
for n in sorted(hist):
    print("  %d queue(s) busy: %8.3f ms" % (n, hist[n] * 1e-6))
perq = collections.defaultdict(float); cnt = collections.Counter()
for s, e, q, k in win:
    perq[q] += (min(e, t1) - s) * 1e-6; cnt[q] += 1
for q in perq:
    print("  queue %s: %d launches, %.3f ms of kernel time" % (q, cnt[q], perq[q]))
# kernel time by name and queue
byk = collections.defaultdict(lambda: [0, 0.0])
for s, e, q, k in win:
    key = (q, k.split("(")[0].replace("void ", "").replace("swn::", "")[:60]); byk[key][0] += 1; byk[key][1] += (e - s) * 1e-6
print("top kernels (queue, name, launches, ms):")
for (q, k), (n, ms) in sorted(byk.items(), key=lambda x: -x[1][1])[:40]:
    print("  %-4s %-62s %4d %8.3f" % (q, k, n, ms))
# gaps on the busiest queue
mainq = max(perq, key=perq.get)
prev = None; gaps = []
for s, e, q, k in win:
    if q != mainq: continue
    if prev is not None and s > prev: gaps.append((s - prev) * 1e-3)
    prev = max(prev or 0, e)
print("queue %s: %d gaps, total %.3f ms, median %.2f us, > 10 us: %d (%.3f ms)" % (mainq, len(gaps), sum(gaps) * 1e-3, sorted(gaps)[len(gaps) // 2] if gaps else 0,
      sum(1 for g in gaps if g > 10), sum(g for g in gaps if g > 10) * 1e-3))

def short(k):
    return k.split("(")[0].replace("void ", "").replace("swn::", "")[:48] or k[:48]
prev = None; pk = None
for s_, e, q, k in win:
    if q != mainq: continue
    if prev is not None and s_ - prev > 10000:
        other = collections.defaultdict(float)
        for s2, e2, q2, k2 in win:
            if q2 != mainq and e2 > prev and s2 < s_:
                other[short(k2)] += (min(e2, s_) - max(s2, prev)) * 1e-6
        print("gap %.3f ms at +%.3f ms after [%s] before [%s]; other queue meanwhile: %s" % ((s_ - prev) * 1e-6, (prev - t0) * 1e-6, short(pk), short(k),
              ", ".join("%s %.3f" % kv for kv in sorted(other.items(), key=lambda x: -x[1])[:6])))
    if prev is None or e > prev: prev, pk = e, k
# what runs ALONE (the other queue idle), by kernel
alone = collections.defaultdict(float)
iv = sorted((s_, e, q, k) for s_, e, q, k in win)
for s_, e, q, k in iv:
    cov = 0
    for s2, e2, q2, k2 in iv:
        if q2 != q and e2 > s_ and s2 < e: cov += min(e, e2) - max(s_, s2)
    alone[(q, short(k))] += max(0, (e - s_) - cov) * 1e-6
print("time a kernel runs with the other queue idle (queue, name, ms):")
for (q, k), ms in sorted(alone.items(), key=lambda x: -x[1])[:30]:
    print("  %-3s %-50s %8.3f" % (q, k, ms))
# --dump-gap: every launch of both queues around the longest gap of the critical queue (what the critical queue waited for)
if "--dump-gap" in sys.argv:
    prev = None; best = (0, 0, 0)
    for s_, e, q, k in win:
        if q != mainq: continue
        if prev is not None and s_ - prev > best[0]: best = (s_ - prev, prev, s_)
        if prev is None or e > prev: prev = e
    g, ga, gb = best
    print("launches around the longest gap of queue %s (%.3f ms, from +%.3f ms):" % (mainq, g * 1e-6, (ga - t0) * 1e-6))
    for s_, e, q, k in rows:
        if e > ga - 300000 and s_ < gb + 700000:
            print("  q%-2s +%9.3f .. +%9.3f ms  %7.1f us  %s" % (q, (s_ - t0) * 1e-6, (e - t0) * 1e-6, (e - s_) * 1e-3, short(k)))
