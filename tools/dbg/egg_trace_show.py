import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "k_egg" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
t0 = None
for r in rows[-8:]:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = "finish" if "finish" in r["Kernel_Name"] else "k_egg "
    if name == "k_egg ": t0 = a
    print("%s start %+8.1f us  dur %7.1f us  end %+8.1f" % (name, (a - (t0 or a)) * 1e-3, (b - a) * 1e-3, (b - (t0 or a)) * 1e-3))
