#!/bin/bash
# bench.py over the configurations DESIGN.md quotes; one JSON line each into gpurun_out/matrix_*.json
mkdir -p gpurun_out
run() { name=$1; shift; python bench.py --no-cpu-baseline "$@" 2>&1 | grep -v amdgpu | tail -1 > gpurun_out/matrix_$name.json; }
run garden
run garden_if1 --inflight 1
run train --splats 1030000
run train_if1 --splats 1030000 --inflight 1
run garden_4k --width 3840 --height 2160
run garden_u8 --sh-format 2 --rgba-format 2
run garden_x8 --instances 8 --steps 48 --warmup 8
run fog --scene fog --steps 48 --warmup 8
run fog_if1 --scene fog --steps 48 --warmup 8 --inflight 1
run sparse --scene sparse
run garden_alphasum --alpha-sum --steps 16 --warmup 4 --inflight 1
run garden_3dgut --pipeline 1 --steps 32 --warmup 4 --inflight 1
run garden_stoch --stochastic
run garden_stoch_if1 --stochastic --inflight 1
run garden_3dgut_stoch --pipeline 1 --stochastic --steps 32 --warmup 4 --inflight 1
run garden_3dgut_dof --pipeline 1 --dof 0.02 --steps 32 --warmup 4 --inflight 1
python - <<'PY'
import json, glob
for f in sorted(glob.glob("gpurun_out/matrix_*.json")):
    try:
        d = json.loads(open(f).read())
        print(f.split("matrix_")[1][:-5].ljust(12), "fps %8.1f" % d["value"], "single-stream ms", {k: round(v, 3) for k, v in d["stage_ms_single_stream"].items()},
              "sortG %.1f" % d["sorted_gsplats_per_s"], "V %.2fM D %.2fM" % (d["visible_splats"]["sorted"] / 1e6, d["visible_splats"]["tile_pairs"] / 1e6), "err", d["error_flags"])
    except Exception as e:
        print(f, "FAILED", e, open(f).read()[-300:])
PY
