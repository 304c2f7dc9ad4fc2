#!/bin/bash
# A/B of alternative builds of libmgs.so on the GPU box: for each csrc/libmgs_<tag>.so given, swap it in (box copy only),
# run a parity subset and the bench (1 and 3 frames in flight); results in gpurun_out/ab_<tag>_*.json
C=vk_gaussian_splatting_amd/csrc
cp $C/libmgs.so /tmp/libmgs_base.so
for tag in base "$@"; do
  if [ "$tag" = base ]; then cp /tmp/libmgs_base.so $C/libmgs.so; else cp $C/libmgs_$tag.so $C/libmgs.so; fi
  python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "keys or full_size or strips or matches_oracle or knobs or binning" 2>&1 | tail -2 > gpurun_out/ab_${tag}_tests.log
  python bench.py --no-cpu-baseline --inflight 1 2>&1 | grep '^{' | tail -1 > gpurun_out/ab_${tag}_if1.json
  python bench.py --no-cpu-baseline 2>&1 | grep '^{' | tail -1 > gpurun_out/ab_${tag}_if3.json
done
cp /tmp/libmgs_base.so $C/libmgs.so
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/ab_*_if*.json")):
    try:
        d = json.loads(open(f).read())
        print(f.split("ab_")[1][:-5].ljust(16), "fps %8.1f" % d["value"], {k: round(v, 4) for k, v in d["stage_ms_single_stream"].items()})
    except Exception as e:
        print(f, "FAILED", e)
for f in sorted(glob.glob("gpurun_out/ab_*_tests.log")):
    print(f, open(f).read().strip().splitlines()[-1:])
PY
