#!/usr/bin/env python3
"""Summarise a rocprofv3 counter_collection.csv: per kernel name, per counter: dispatches, mean value.
--largest-grid: per kernel, only the dispatches with the largest Grid_Size (the full-frame launches)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
largest = "--largest-grid" in sys.argv
grid = collections.defaultdict(float)
for r in rows:
    k = r.get("Kernel_Name", "?")
    grid[k] = max(grid[k], float(r.get("Grid_Size", 0) or 0))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "?")
    if largest and float(r.get("Grid_Size", 0) or 0) != grid[k]:
        continue
    acc[k][r.get("Counter_Name", "?")].append(float(r.get("Counter_Value", "nan")))
for k, cs in acc.items():
    if not (k.startswith("sbx::") or "k_" in k):
        continue
    if largest and grid[k] < 1e5:
        continue
    print("kernel %s" % k[:80])
    for c, v in sorted(cs.items()):
        print("   %-28s dispatches=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
