#!/bin/bash
# per-kernel rocprofv3 stats of ONE strip (tile rows B..E of a WxH frame), one frame in flight:
#   tools/strip_kernels.sh 1920 1080 34 38  -> gpurun_out/strip_kernels_<W>x<H>_<B>_<E>.csv
R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; export TMPDIR=/tmp; rm -rf /tmp/psk
cat > /tmp/strip_one.py <<PY
import sys, os
sys.path.insert(0, "$R")
import numpy as np, torch
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth
W, H, B, E = $1, $2, $3, $4
sc = synth.make_scene(5_830_000, seed=0xC0FFEE + 2)
s = mgs.Scene(0); s.add_instance(mgs.SplatSet.from_arrays(**sc)); s.commit()
for i in range(40):
    eye = synth.orbit_pose(i % 64)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
    p = capi.default_params(W, H); capi.set_camera(p, V, P, eye); p.strip_row_begin, p.strip_row_end = B, E
    s.render(p)
torch.cuda.synchronize()
print(s.last_stats() if hasattr(s, "last_stats") else "")
PY
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/psk -- python /tmp/strip_one.py > /tmp/lsk 2>&1
OUT=$R/gpurun_out/strip_kernels_$1x$2_$3_$4.csv
cp $(find /tmp/psk -name "*kernel_stats.csv" | head -1) $OUT || tail -20 /tmp/lsk
python - <<PY
import csv
tot = 0
for r in csv.DictReader(open('$OUT')):
    if int(r['Calls']) < 30: continue
    tot += float(r['AverageNs']) * int(r['Calls']) / 40
    print(r['Name'][:70].ljust(70), r['Calls'].rjust(4), '%9.1f'%(float(r['AverageNs'])/1000))
print('sum per frame us', tot / 1000)
PY
