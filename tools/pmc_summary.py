#!/usr/bin/env python3
"""Summarise a rocprofv3 counter_collection.csv: per kernel name, per counter: dispatches, mean value."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "?")
    acc[k][r.get("Counter_Name", "?")].append(float(r.get("Counter_Value", "nan")))
for k, cs in acc.items():
    if not (k.startswith("sbx::") or "k_" in k):
        continue
    print("kernel %s" % k[:80])
    for c, v in sorted(cs.items()):
        print("   %-28s dispatches=%d mean=%.6g" % (c, len(v), sum(v) / len(v)))
