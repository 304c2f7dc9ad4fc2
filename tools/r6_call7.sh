#!/bin/bash
# round 6, GPU call 7: the one-kernel binning (MGS_DB_SWEEP=1, v1: every bin's segment holds n entries) against count / scan / emit
T=${TAG:-r6_h}; mkdir -p gpurun_out; rm -f gpurun_out/${T}_ab.log
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "binning_paths or key_sort_variants" 2>&1 | tail -15 ) > gpurun_out/${T}_tests.log
cat gpurun_out/${T}_tests.log | tail -5
for rep in 1 2 3; do
  for v in 0 1; do
    export MGS_DB_SWEEP=$v
    python tools/stage_times.py --graph --tag garden_sweep$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
    if [ $rep != 3 ]; then
      python tools/stage_times.py --splats 1030000 --graph --tag train_sweep$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
      python tools/stage_times.py --scene fog --graph --tag fog_sweep$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
      python tools/stage_times.py --width 3840 --height 2160 --graph --tag 4k_sweep$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
      python tools/stage_times.py --strip 34 38 --graph --tag strip_sweep$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
    fi
  done
done
for v in 0 1; do
  MGS_DB_SWEEP=$v python bench.py --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench_if3_sweep$v', round(d['value'],1), 'single', round(d['value_single_frame'],1), {k: round(v*1000,1) for k,v in d['stage_ms_single_stream'].items()}, d.get('parity',{}).get('psnr_db_min'), 'err', d['error_flags'])" >> gpurun_out/${T}_ab.log
done
cat gpurun_out/${T}_ab.log
