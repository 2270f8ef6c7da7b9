import os, sys, time, glob, subprocess, threading, ctypes
sys.path.insert(0, "/root/repo")
import torch, shaderbox_amd
hip = ctypes.CDLL("libamdhip64.so")
b = ctypes.create_string_buffer(64); hip.hipDeviceGetPCIBusId(b, 64, 0); pci = b.value.decode().lower()
card = [c for c in glob.glob("/sys/class/drm/card*/device") if os.path.realpath(c).lower().endswith(pci)][0]
print("pci", pci, "card", card)
hw = glob.glob(card + "/hwmon/hwmon*")[0]
print(sorted(os.listdir(hw)))
print(open(card + "/pp_dpm_sclk").read())
R = shaderbox_amd.Renderer(0)
def rd(p):
    try: return open(p).read().strip()
    except Exception as e: return "ERR"
for app, W, H in (("clouds", 3840, 2160), ("planet", 7680, 4320)):
    out = torch.empty((H, W, 4), device="cuda")
    stop = threading.Event(); acc = []
    def samp():
        while not stop.is_set():
            t = time.perf_counter()
            acc.append((t, rd(hw + "/freq1_input"), rd(hw + "/power1_input"), [l for l in rd(card + "/pp_dpm_sclk").splitlines() if "*" in l]))
            time.sleep(.1)
    th = threading.Thread(target=samp); th.start()
    t0 = time.perf_counter(); n = 0
    while time.perf_counter() - t0 < 3:
        for _ in range(10): R.render(app, W, H, .37, out=out)
        torch.cuda.synchronize(); n += 10
    smi = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showuse"], capture_output=True, text=True).stdout
    stop.set(); th.join()
    print(app, "%.3f ms/frame" % ((time.perf_counter() - t0) * 1e3 / n))
    for a in acc[::3]: print("   t=%.2f freq1=%s power1=%s dpm=%s" % (a[0] - t0, a[1], a[2], a[3]))
    print(smi[-1500:])
