"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py -> profiles/<name>.json (per-stage HBM bytes per launch).
usage: python tools/pmc_traffic.py <fetch_dir> <write_dir> <out.json> "<command line that was profiled>"
Correction (MI355X_MICROARCH.md, HBM section): on gfx950 FETCH_SIZE tallies wide (16 B/lane) coalesced streaming reads at
half their bytes, so reads are doubled.  Round 4 calibrated the other widths (tools/micro/pmc_calib.hip,
profiles/r4_z_pmc_calibration.json): streaming reads of 4-, 8-, 16-byte lanes and of 3 x 4 bytes at a 12-byte pitch are ALL
tallied at exactly half, writes of 4-, 8-, 16-byte lanes and 32-byte records exactly: 2*FETCH_SIZE + WRITE_SIZE holds."""
import csv, glob, json, sys, collections

STAGE = {"project": ["k_project"], "sort": ["k_sort_", "k_os_"], "bin": ["k_dbin_", "k_bin_"], "composite": ["k_composite"]}


def per_kernel(d, counter):
    acc = collections.defaultdict(lambda: [0, 0.0])
    disp = collections.defaultdict(set)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"].split("(")[0]
            acc[k][0] += 1
            acc[k][1] += float(r["Counter_Value"])
    return {k: {"launches": v[0], "mean_KB": v[1] / max(v[0], 1), "total_KB": v[1]} for k, v in acc.items()}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
frames = max(v["launches"] for k, v in fetch.items() if "k_composite" in k)
out = {"command": sys.argv[4], "frames_profiled": frames,
       "correction": "per frame: 2*FETCH_SIZE (gfx950 tallies 16 B/lane streaming reads at half) + WRITE_SIZE, KB*1024",
       "kernels": {}, "all": {"FETCH_SIZE": fetch, "WRITE_SIZE": write}}
for stage, pats in STAGE.items():
    # k_project<false> is the sort-only hook (bench.py's parity leg, mgs_sort_download): not part of a frame
    f = sum(v["total_KB"] for k, v in fetch.items() if any(p in k for p in pats) and "<false>" not in k) / frames
    w = sum(v["total_KB"] for k, v in write.items() if any(p in k for p in pats) and "<false>" not in k) / frames
    out["kernels"][stage] = {"FETCH_SIZE_KB_per_frame": f, "WRITE_SIZE_KB_per_frame": w,
                             "traffic_bytes_per_launch_corrected": (2 * f + w) * 1024, "traffic_bytes_uncorrected": (f + w) * 1024}
    print(stage, "fetch %.1f MB write %.1f MB -> corrected %.1f MB" % (f / 1024, w / 1024, (2 * f + w) / 1024))
json.dump(out, open(sys.argv[3], "w"), indent=1)
