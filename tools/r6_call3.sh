#!/bin/bash
# round 6, GPU call 3: the suite on the build with the fused launch of sort passes 2 + 3 and k_project's early-outs for empty
# partitions; same-box A/B of MGS_OS_FUSE23 = 0 / 1 and of the early-outs (csrc/libmgs_base.so = the tree before both)
T=${TAG:-r6_c}; mkdir -p gpurun_out; rm -f gpurun_out/${T}_ab.log; R=$PWD; C=$R/vk_gaussian_splatting_amd/csrc
( timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/${T}_gpu_tests.log
for rep in 1 2 3; do
  for v in base unfused new; do
    unset MGS_LIB MGS_OS_FUSE23
    if [ $v = base ]; then export MGS_LIB=$C/libmgs_base.so; fi
    if [ $v = unfused ]; then export MGS_OS_FUSE23=0; fi
    python tools/stage_times.py --graph --tag garden_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
    python tools/stage_times.py --strip 34 38 --graph --tag strip34_38_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
    python tools/stage_times.py --splats 1030000 --graph --tag train_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
  done
done
unset MGS_LIB MGS_OS_FUSE23
for v in base new; do
  if [ $v = base ]; then export MGS_LIB=$C/libmgs_base.so; else unset MGS_LIB; fi
  python bench.py --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench_if3_$v', round(d['value'],1), 'single', round(d['value_single_frame'],1), {k: round(v*1000,1) for k,v in d['stage_ms_single_stream'].items()}, d.get('parity',{}).get('psnr_db_min'))" >> gpurun_out/${T}_ab.log
done
unset MGS_LIB
cat gpurun_out/${T}_ab.log; tail -3 gpurun_out/${T}_gpu_tests.log
