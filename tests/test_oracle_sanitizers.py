"""The CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer: a small frame of every app, the noise
library, the texture filter and the multi-threaded row renderer, in a subprocess with libasan preloaded (an ASan library
cannot be loaded into an un-instrumented python otherwise).  Any report aborts the subprocess (-fno-sanitize-recover,
halt_on_error)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
from oracle.oracle import Oracle, APP_IDS
o = Oracle(variant="_san", subdir="_san")
for app, aid in sorted(APP_IDS.items()):
    if app == "clouds_tex":
        v1, v2 = o.worley_volume(8), o.worley_volume(4)
        o.set_noise_volumes(v1, v2)
    for t, mouse in ((0.0, (0.0, 0.0)), (2.5, (11.0, 7.0))):
        img = o.render(aid, 24, 16, t, mouse=mouse)
        assert img.shape == (16, 24, 4)
    rows = o.render_rows(aid, 33, 9, 0.37, [0, 4, 8], threads=3)          # the threaded tile scheduler, ragged width
    assert rows.shape == (3, 33, 4)
xyz = np.random.default_rng(1).uniform(-5, 5, (64, 3)).astype(np.float32)
for fn in ("noise_iq", "hash_w"):
    o.noise(fn, xyz)
print("sanitized oracle: all apps rendered")
""" % ROOT


def _libasan():
    r = subprocess.run(["g++", "-print-file-name=libasan.so"], capture_output=True, text=True)
    p = r.stdout.strip()
    return p if r.returncode == 0 and os.path.isabs(p) and os.path.exists(p) else None


def test_oracle_is_clean_under_asan_and_ubsan():
    asan = _libasan()
    if asan is None:
        pytest.skip("libasan.so not found next to g++")
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "_san/libsbx_oracle_san.so"], check=True)
    env = dict(os.environ)
    env["LD_PRELOAD"] = asan
    env["ASAN_OPTIONS"] = "detect_leaks=0:halt_on_error=1"          # python itself leaks by ASan's definition
    env["UBSAN_OPTIONS"] = "halt_on_error=1:print_stacktrace=1"
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    assert "sanitized oracle: all apps rendered" in r.stdout
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]
