#!/usr/bin/env python3
"""Quantify the distance between the sbx math spec and a glibc-libm environment.

Renders the same frames with oracle/libsbx_oracle.so (sbx math spec) and
oracle/libsbx_oracle_libm.so (same restatement, transcendental functions from glibc — what
the reference's "VML + libm" C++ path would use) and prints max |diff| per channel and the
number of pixels that differ by more than 1e-4.  Measurement only; nothing depends on it.
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from oracle.oracle import APP_IDS, Oracle  # noqa: E402

a, b = Oracle(), Oracle("_libm")
cases = [("clouds", 256, 144), ("clouds", 480, 270), ("egg", 256, 256), ("raytracer", 256, 256),
         ("atmosphere", 256, 144), ("sdf_ao", 256, 144), ("planet", 256, 144)]
if len(sys.argv) > 1:
    cases = [(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))]
for name, w, h in cases:
    for t in (0.0, 0.37, 2.5):
        t0 = time.time()
        x = a.render(APP_IDS[name], w, h, t)
        y = b.render(APP_IDS[name], w, h, t)
        both_nan = np.isnan(x) & np.isnan(y)
        d = np.where(both_nan, 0.0, np.abs(x.astype(np.float64) - y))
        d = np.nan_to_num(d, nan=np.inf)
        bad = int((d.max(axis=2) > 1e-4).sum())
        print("%-10s %4dx%-4d t=%.2f  max|diff|=%.3g  pixels>1e-4: %d / %d (%.3f%%)  nan(sbx)=%d nan(libm)=%d  [%.1fs]"
              % (name, w, h, t, d.max(), bad, w * h, 100.0 * bad / (w * h),
                 int(np.isnan(x).any(axis=2).sum()), int(np.isnan(y).any(axis=2).sum()), time.time() - t0))
