#!/bin/bash
# round 6, GPU call 6: sub-box test in k_project (csrc/libmgs_base.so = the committed tree: per-splat strip bound only;
# libmgs_nolb.so = the sub-box build without the 6-waves register cap: 96 VGPRs, no spills) + the refined bin policy's test
T=${TAG:-r6_f}; mkdir -p gpurun_out; rm -f gpurun_out/${T}_ab.log; R=$PWD; C=$R/vk_gaussian_splatting_amd/csrc
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/${T}_gpu_tests.log
export MGS_BIN_ADAPT=0
for rep in 1 2 3; do
  for v in base new nolb; do
    unset MGS_LIB
    if [ $v = base ]; then export MGS_LIB=$C/libmgs_base.so; fi
    if [ $v = nolb ]; then export MGS_LIB=$C/libmgs_nolb.so; fi
    python tools/stage_times.py --strip 34 38 --graph --tag strip34_38_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
    python tools/stage_times.py --strip 0 12 --graph --tag strip0_12_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
    python tools/stage_times.py --graph --tag garden_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
    if [ $rep != 3 ]; then python tools/stage_times.py --splats 1030000 --graph --tag train_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log; fi
  done
done
unset MGS_LIB
for v in base new nolb; do
  unset MGS_LIB
  if [ $v = base ]; then export MGS_LIB=$C/libmgs_base.so; fi
  if [ $v = nolb ]; then export MGS_LIB=$C/libmgs_nolb.so; fi
  python bench.py --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench_if3_$v', round(d['value'],1), 'single', round(d['value_single_frame'],1), {k: round(v*1000,1) for k,v in d['stage_ms_single_stream'].items()}, d.get('parity',{}).get('psnr_db_min'))" >> gpurun_out/${T}_ab.log
done
unset MGS_LIB
unset MGS_BIN_ADAPT
MGS_LIB=$C/libmgs_trace.so MGS_GRAPH=0 MGS_PRJ_TRACE_FILE=/tmp/p.bin STRIP="34 38" timeout 300 python tools/prj_trace.py 0 > gpurun_out/${T}_prj_trace_strip.log 2>&1
cat gpurun_out/${T}_ab.log; grep -n "passed\|failed" gpurun_out/${T}_gpu_tests.log; grep -v amdgpu gpurun_out/${T}_prj_trace_strip.log | head -20
