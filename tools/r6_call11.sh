#!/bin/bash
# round 6: the sort passes' counts published before the ranking (MGS_OS_FLAT=1) against after it (0): correctness + stage times + trace
T=${TAG:-r6_l}; mkdir -p gpurun_out; C=$PWD/vk_gaussian_splatting_amd/csrc; rm -f gpurun_out/${T}_ab.log
( MGS_OS_FLAT=1 timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sort or keys" 2>&1 | tail -3 ) | tee gpurun_out/${T}_tests.log
for rep in 1 2 3; do for v in 0 1; do
  export MGS_OS_FLAT=$v
  python tools/stage_times.py --graph --tag garden_flat$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  if [ $rep != 3 ]; then
    python tools/stage_times.py --splats 1030000 --graph --tag train_flat$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
    python tools/stage_times.py --strip 34 38 --graph --tag strip_flat$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
    python tools/stage_times.py --instances 8 --frames 16 --graph --tag x8_flat$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  fi
done; done
for v in 0 1; do
  MGS_OS_FLAT=$v python bench.py --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench_if3_flat$v', round(d['value'],1), 'single', round(d['value_single_frame'],1), {k: round(v*1000,1) for k,v in d['stage_ms_single_stream'].items()}, d.get('parity',{}).get('psnr_db_min'), 'err', d['error_flags'])" >> gpurun_out/${T}_ab.log
  MGS_OS_FLAT=$v MGS_LIB=$C/libmgs_trace.so MGS_GRAPH=0 MGS_OS_TRACE_FILE=/tmp/o.bin timeout 300 python tools/os_trace.py 0 2>&1 | grep -v amdgpu > gpurun_out/${T}_os_trace_flat$v.log
done
cat gpurun_out/${T}_ab.log; grep -A8 "pass 1: \|pass 2: " gpurun_out/${T}_os_trace_flat0.log gpurun_out/${T}_os_trace_flat1.log | grep -v "^--$"
