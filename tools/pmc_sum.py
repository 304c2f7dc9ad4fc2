"""sum rocprofv3 --pmc counter_collection CSVs per kernel: python tools/pmc_sum.py <dir> [out.json]"""
import csv, glob, json, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        a = acc[k][r["Counter_Name"]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
out = {k: {c: {"launches": v[0], "mean": v[1] / max(v[0], 1)} for c, v in d.items()} for k, d in acc.items()}
for k, d in sorted(out.items()):
    print(k[-48:], {c: round(v["mean"]) for c, v in d.items()})
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
