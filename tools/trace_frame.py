"""scratch: per-frame kernel timeline (durations and gaps) from a rocprofv3 kernel trace csv"""
import csv, sys, statistics
rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if "mgs::" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
frames, cur = [], []
for r in rows:
    if "k_frame_init" in r["Kernel_Name"] and cur:
        frames.append(cur); cur = []
    cur.append(r)
frames.append(cur)
fr = [f for f in frames if len(f) > 25]; fr = fr[len(fr)//2 : len(fr)//2 + 12]
f = fr[0]
t0 = int(f[0]["Start_Timestamp"]); prev = t0
for r in f:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][:44]
    print("%-44s start %8.1f dur %7.1f gap %6.1f" % (name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3))
    prev = e
tot = [(int(f[-1]["End_Timestamp"]) - int(f[0]["Start_Timestamp"])) / 1e3 for f in fr]
busy = [sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in f) / 1e3 for f in fr]
print("frame span us", statistics.mean(tot), "sum kernel us", statistics.mean(busy), "kernels/frame", len(fr[0]))
