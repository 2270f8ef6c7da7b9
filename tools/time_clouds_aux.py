#!/usr/bin/env python3
"""APP_CLOUDS 3840x2160 away from the default aux block: kernel time (HIP events, median of 7) of the frames that select the
other instantiations of k_clouds (sun off the z axis -> the general light march; sigma*dt beyond the REG bound; more steps than
the y table holds) and a few mouse/time points, each compared with the per-lane kernel (variant 1) for identical bits."""
import math
import sys
import torch
sys.path.insert(0, ".")
import shaderbox_amd

if len(sys.argv) > 1:                      # an A/B library of tools/ab_build.py instead of the shipped one
    shaderbox_amd.LIB_PATH = sys.argv[1]
R = shaderbox_amd.Renderer(0)
R.set_timing(True)
W, H = 3840, 2160


def aux(**kw):
    a = shaderbox_amd.clouds_defaults()
    for k, v in kw.items():
        if isinstance(v, (tuple, list)):
            for i, x in enumerate(v):
                getattr(a, k)[i] = x
        else:
            setattr(a, k, v)
    return a


def norm(v):
    n = math.sqrt(sum(x * x for x in v))
    return tuple(x / n for x in v)


CASES = [("default", None, .37, (0, 0)),
         ("time 12.5", None, 12.5, (0, 0)),
         ("mouse (.3,.6)", None, .37, (.3 * W, .6 * H)),
         ("sun (.3,.5,.8)", aux(sun_dir=norm((.3, .5, .8))), .37, (0, 0)),
         ("sun (0,1,.2)", aux(sun_dir=norm((0, 1, .2))), .37, (0, 0)),
         ("sun (0,.3,-1)", aux(sun_dir=norm((0, .3, -1))), .37, (0, 0)),          # raised in the y-z plane: light_march_yz
         ("sun (0,-.6,-.8)", aux(sun_dir=(0., -.6, -.8)), .37, (0, 0)),
         ("coverage .7", aux(cld_coverage=.7), .37, (0, 0)),
         ("coverage .3", aux(cld_coverage=.3), .37, (0, 0)),
         ("thick 150", aux(cld_thick=150.), .37, (0, 0)),
         ("steps 200/12", aux(cld_march_steps=200, illum_march_steps=12), .37, (0, 0)),
         ("steps 50/3", aux(cld_march_steps=50, illum_march_steps=3), .37, (0, 0)),
         ("steps 1100/6", aux(cld_march_steps=1100), .37, (0, 0)),
         ("steps 5000/6 thick 6250", aux(cld_march_steps=5000, cld_thick=6250.), .37, (0, 0)),   # beyond the ring's 4096 rows: the on-demand table
         ("sigma 3", aux(sigma_scattering=3.), .37, (0, 0)),
         ("wind (1,0,.5)", aux(wind_dir=(1., 0., .5)), 7.0, (0, 0))]
buf = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
ref = torch.empty((H, W, 4), dtype=torch.float32, device="cuda")
for _ in range(40):                       # clocks up before the first case
    R.render("clouds", W, H, .37, out=buf)
torch.cuda.synchronize()
import os
if os.environ.get("SBX_AUX_ONLY"):
    CASES = [c for c in CASES if any(k in c[0] for k in os.environ["SBX_AUX_ONLY"].split(","))]
for name, a, t, m in CASES:
    R.set_variant(0)
    R.render("clouds", W, H, t, mouse=m, aux=a, out=buf); torch.cuda.synchronize()
    ms = []
    for _ in range(7):
        R.render("clouds", W, H, t, mouse=m, aux=a, out=buf); ms.append(R.last_kernel_ms())
    ms.sort()
    R.set_variant(1)
    R.render("clouds", W, H, t, mouse=m, aux=a, out=ref); torch.cuda.synchronize()
    ms1 = R.last_kernel_ms()
    same = bool(((buf == ref) | (torch.isnan(buf) & torch.isnan(ref))).all())
    print("%-16s %8.3f ms   per-lane kernel %8.3f ms   %s" % (name, ms[3], ms1, "same bits" if same else "DIFFERENT"))
