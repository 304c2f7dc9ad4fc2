#!/bin/bash
# round 6, GPU call 5: the adaptive bin size (BinPolicy): its test + same-box A/B of MGS_BIN_ADAPT = 0 / 1 on the scenes of round 5's
# bin-size sweep (frames after the policy has settled)
T=${TAG:-r6_e}; mkdir -p gpurun_out; rm -f gpurun_out/${T}_ab.log
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "adaptive_bin or binning or row_costs or strips or bench_prints" 2>&1 | tail -15 ) > gpurun_out/${T}_gpu_tests.log
for rep in 1 2; do
  for v in 0 1; do
    export MGS_BIN_ADAPT=$v
    python tools/stage_times.py --splats 1030000 --skip 40 --frames 48 --graph --tag train_adapt$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
    python tools/stage_times.py --scene fog --skip 40 --frames 48 --graph --tag fog_adapt$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
    python tools/stage_times.py --scene sparse --skip 40 --frames 48 --graph --tag sparse_adapt$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
    python tools/stage_times.py --skip 40 --frames 48 --graph --tag garden_adapt$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  done
done
unset MGS_BIN_ADAPT
cat gpurun_out/${T}_ab.log; tail -5 gpurun_out/${T}_gpu_tests.log
