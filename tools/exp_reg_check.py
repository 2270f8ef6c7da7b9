"""Exhaustive check on the GPU box: the kernel-internal exp forms against exp_ on every binary32 argument with |x| < 80."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
import shaderbox_amd
R = shaderbox_amd.Renderer(0)
lim = int(np.array([80.0], dtype=np.float32).view(np.uint32)[0])
chunk = 1 << 26
bad_total = {"exp_reg64": [], "exp_reg64_plain": []}
for sign in (0, 0x80000000):
    for start in range(0, lim + 1, chunk):
        stop = min(start + chunk, lim + 1)
        bits = (torch.arange(start, stop, dtype=torch.int64, device="cuda") | sign).to(torch.int32)
        x = bits.view(torch.float32)
        b = R.math("exp", x)
        for form in bad_total:
            a = R.math(form, x)
            bad = a.view(torch.int32) != b.view(torch.int32)
            if bool(bad.any()):
                bb = bits[bad][:8].cpu().numpy() & 0xffffffff
                bad_total[form] += [hex(int(v)) for v in bb]
for f, v in bad_total.items():
    print(f, "mismatches:", len(v), v[:8])
