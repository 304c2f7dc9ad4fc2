"""counted KB / bytes moved per access width from the two rocprofv3 --pmc passes of tools/micro/pmc_calib:
python tools/micro/pmc_calib.py <fetch_dir> <write_dir> [out.json]"""
import csv, glob, json, sys, collections
MOVED = float(1 << 30)


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"bytes_moved_per_launch": MOVED, "note": "ratio = counter (KB * 1024) / bytes moved; a ratio of 0.5 is the halving MI355X_MICROARCH.md corrects for", "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0) * 1024 / MOVED, write.get(k, 0.0) * 1024 / MOVED
    out["kernels"][k] = {"FETCH_SIZE_ratio": f, "WRITE_SIZE_ratio": w}
    print(f"{k[:48]:48s} FETCH_SIZE x{f:.3f}  WRITE_SIZE x{w:.3f}")
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
