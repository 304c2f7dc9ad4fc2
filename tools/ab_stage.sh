#!/bin/bash
# timing-only A/B of experimental builds (frames may be wrong): per-stage ms of bench.py --inflight 1 for csrc/libmgs_<tag>.so
C=vk_gaussian_splatting_amd/csrc
cp $C/libmgs.so /tmp/libmgs_base.so
for tag in base "$@"; do
  if [ "$tag" = base ]; then cp /tmp/libmgs_base.so $C/libmgs.so; else cp $C/libmgs_$tag.so $C/libmgs.so; fi
  python bench.py --no-cpu-baseline --inflight 1 --sh-format 1 --steps 48 2>&1 | grep '^{' | tail -1 > gpurun_out/abs_${tag}.json
done
cp /tmp/libmgs_base.so $C/libmgs.so
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/abs_*.json")):
    try:
        d = json.loads(open(f).read())
        print(f.split("abs_")[1][:-5].ljust(10), "fps %8.1f" % d["value"], {k: round(v, 4) for k, v in d["stage_ms_single_stream"].items()}, "sorted", d["visible_splats"]["sorted"])
    except Exception as e:
        print(f, "FAILED", e)
PY
