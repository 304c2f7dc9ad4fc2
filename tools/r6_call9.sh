#!/bin/bash
# round 6: quick look at the sweep: trace + garden / 4K stage times, same box
T=${TAG:-r6_i}; mkdir -p gpurun_out; C=$PWD/vk_gaussian_splatting_amd/csrc; rm -f gpurun_out/${T}_ab.log
( timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "binning_paths" 2>&1 | tail -3 )
true
for rep in 1 2; do for v in 0 1; do
  MGS_DB_SWEEP=$v python tools/stage_times.py --graph --tag garden_sweep$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  MGS_DB_SWEEP=$v python tools/stage_times.py --width 3840 --height 2160 --graph --tag 4k_sweep$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  MGS_DB_SWEEP=$v python tools/stage_times.py --splats 1030000 --graph --tag train_sweep$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
done; done
for v in 0 1; do
  MGS_DB_SWEEP=$v python bench.py --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench_if3_sweep$v', round(d['value'],1), 'single', round(d['value_single_frame'],1), {k: round(v*1000,1) for k,v in d['stage_ms_single_stream'].items()}, d.get('parity',{}).get('psnr_db_min'), 'err', d['error_flags'])" >> gpurun_out/${T}_ab.log
done
cat gpurun_out/${T}_ab.log
