#!/bin/bash
# rocprofv3 passes behind profiles/${TAG}_* (run on the GPU box via gpurun; results land in gpurun_out/).
export TAG=${TAG:-r3_z}; export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; rm -rf /tmp/pf /tmp/pw /tmp/psq /tmp/ps1 /tmp/ps3
CMD="python $R/bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extras --inflight 1"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/pf -- $CMD > /tmp/l1 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/pw -- $CMD > /tmp/l2 2>&1
python $R/tools/pmc_traffic.py /tmp/pf /tmp/pw $R/gpurun_out/${TAG}_pmc_hbm_traffic.json "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace -- $CMD (two separate passes)"
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d /tmp/psq -- $CMD > /tmp/l5 2>&1
python $R/tools/pmc_sum.py /tmp/psq $R/gpurun_out/pmc_sq_all.json > /dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps1 -- $CMD > /tmp/l3 2>&1
cp $(find /tmp/ps1 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_inflight1_kernel_stats.csv
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ps3 -- python $R/bench.py --no-cpu-baseline --no-extras > /tmp/l4 2>&1
cp $(find /tmp/ps3 -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${TAG}_default_inflight3_kernel_stats.csv
cd $R
python - <<'PY'
import json, os
d = json.load(open('gpurun_out/pmc_sq_all.json'))
out = {}
for k, v in d.items():
    name = 'k_composite' if 'k_composite' in k else ('k_project' if 'k_project<true>' in k else None)
    if name:
        out[name] = {c: x['mean'] for c, x in v.items()}
        out[name]['kernel'] = k
out['command'] = "rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -- python bench.py --steps 24 --warmup 4 --no-cpu-baseline --no-extras --inflight 1 (per-launch means, chip-wide sums; SQ_*_CYCLES in quad-cycles)"
json.dump(out, open('gpurun_out/' + os.environ.get("TAG", "r2_d") + '_pmc_sq_composite.json', 'w'), indent=1)
for k in ('k_project', 'k_composite'):
    print(k, {c: round(x / 1e6, 1) for c, x in out[k].items() if c != 'kernel'})
PY
