#!/bin/bash
# SQ counters of the additive-alpha compositor (MGS_ALPHA_SUM) on the garden-sized frame: two --pmc passes, per-launch means.
export TMPDIR=/tmp; R=${GRAFT_REPO_ROOT:-/root/repo}; cd /tmp; rm -rf /tmp/pa1 /tmp/pa2
CMD="python $R/bench.py --alpha-sum --steps 6 --warmup 2 --no-cpu-baseline --no-extras --inflight 1"
rocprofv3 --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d /tmp/pa1 -- $CMD > /tmp/la1 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d /tmp/pa2 -- $CMD > /tmp/la2 2>&1
tail -n 3 /tmp/la1; tail -n 3 /tmp/la2
python $R/tools/pmc_sum.py /tmp/pa1 $R/gpurun_out/pmc_alpha_sum_1.json | grep composite
python $R/tools/pmc_sum.py /tmp/pa2 $R/gpurun_out/pmc_alpha_sum_2.json | grep composite
