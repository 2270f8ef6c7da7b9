#!/usr/bin/env python3
"""trace_overlap.py <kernel_trace.csv> <kernel substring> — the launch intervals of one kernel from a rocprofv3 kernel trace:
per full-size launch its start, end and duration relative to the first one, how many other launches of the same kernel were
running when it started, and the summary (mean duration, mean start-to-start interval, fraction of the span with >= 2 launches
resident).  What shows that `ms_per_step` < `kernel_ms` with frames in flight is overlap, not a shorter kernel."""
import csv
import sys

path, key = sys.argv[1], sys.argv[2]
rows = [r for r in csv.DictReader(open(path)) if key in r.get("Kernel_Name", "")]
iv = []
for r in rows:
    try:
        iv.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), float(r.get("Grid_Size", 0) or r.get("Grid_Size_X", 0) or 0),
                   r.get("Stream_Id", r.get("Queue_Id", "?"))))
    except (KeyError, ValueError):
        pass
if not iv:
    print("(no launches of %s in %s)" % (key, path))
    sys.exit(0)
big = max(g for _, _, g, _ in iv)
iv = sorted(x for x in iv if x[2] == big)
t0 = iv[0][0]
print("launches of %s (largest grid only): %d" % (key, len(iv)))
print("  #   start_ms    end_ms   dur_ms  resident_at_start  queue")
for i, (a, b, _, q) in enumerate(iv):
    res = sum(1 for (c, d, _, _) in iv if c < a < d)
    if i < 12 or i >= len(iv) - 6:
        print("%3d  %9.4f %9.4f %8.4f  %d  %s" % (i, (a - t0) * 1e-6, (b - t0) * 1e-6, (b - a) * 1e-6, res, q))
    elif i == 12:
        print("  ...")
steady = iv[len(iv) // 4: -max(1, len(iv) // 8)] if len(iv) >= 12 else iv
dur = sum(b - a for a, b, _, _ in steady) / len(steady) * 1e-6
gaps = [(steady[i + 1][0] - steady[i][0]) * 1e-6 for i in range(len(steady) - 1)]
gaps = [g for g in gaps if g < 20 * dur]                # (the warm-up / timed-region boundary is a long gap)
ends = [(steady[i + 1][1] - steady[i][1]) * 1e-6 for i in range(len(steady) - 1)]
ends = [g for g in ends if g < 20 * dur]
# time with >= 2 launches resident within the steady part
ev = sorted([(a, 1) for a, _, _, _ in steady] + [(b, -1) for _, b, _, _ in steady])
lvl, last, two, span = 0, ev[0][0], 0, ev[-1][0] - ev[0][0]
for tt, d in ev:
    if lvl >= 2:
        two += tt - last
    lvl += d
    last = tt
print("steady part (%d launches): mean duration %.4f ms, mean start-to-start %.4f ms, mean end-to-end %.4f ms, "
      ">= 2 launches resident %.1f %% of the span" % (len(steady), dur, sum(gaps) / max(len(gaps), 1), sum(ends) / max(len(ends), 1),
                                                     100.0 * two / max(span, 1)))
