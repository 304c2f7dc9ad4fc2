#!/bin/bash
# round 6, GPU call 2: the suite on the build whose records carry the 2D covariance (k_project diet, item 6a); same-box A/B against
# csrc/libmgs_base.so (the committed tree before it); SQ_INSTS_VALU of both; k_project's per-workgroup trace on a middle strip
T=${TAG:-r6_b}; mkdir -p gpurun_out; rm -f gpurun_out/${T}_ab.log; R=$PWD; C=$R/vk_gaussian_splatting_amd/csrc
( timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/${T}_gpu_tests.log
for rep in 1 2; do
  for v in base new; do
    if [ $v = base ]; then export MGS_LIB=$C/libmgs_base.so; else unset MGS_LIB; fi
    python tools/stage_times.py --graph --tag garden_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
    python tools/stage_times.py --splats 1030000 --tag train_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
    python tools/stage_times.py --scene fog --tag fog_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
    python tools/stage_times.py --width 3840 --height 2160 --tag 4k_$v 2>&1 | grep -v amdgpu >> gpurun_out/${T}_ab.log
    python bench.py --no-cpu-baseline --no-extras 2>/dev/null | grep '^{' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench_if3_$v', round(d['value'],1), 'single', round(d['value_single_frame'],1), d.get('parity',{}).get('psnr_db_min'))" >> gpurun_out/${T}_ab.log
  done
done
unset MGS_LIB
cd /tmp; export TMPDIR=/tmp
for v in base new; do
  if [ $v = base ]; then export MGS_LIB=$C/libmgs_base.so; else unset MGS_LIB; fi
  rm -rf /tmp/psq_$v
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU --kernel-trace --output-format csv -d /tmp/psq_$v -- python $R/tools/stage_times.py --frames 16 --tag pmc_$v > /tmp/lpmc_$v 2>&1
  echo "== $v" >> $R/gpurun_out/${T}_pmc.log; python $R/tools/pmc_sum.py /tmp/psq_$v >> $R/gpurun_out/${T}_pmc.log 2>&1
done
unset MGS_LIB; cd $R
MGS_LIB=$C/libmgs_trace.so MGS_GRAPH=0 MGS_PRJ_TRACE_FILE=/tmp/p.bin STRIP="34 38" timeout 300 python tools/prj_trace.py 0 20 > gpurun_out/${T}_prj_trace_strip.log 2>&1
MGS_LIB=$C/libmgs_trace.so MGS_GRAPH=0 MGS_PRJ_TRACE_FILE=/tmp/p.bin timeout 300 python tools/prj_trace.py 0 > gpurun_out/${T}_prj_trace_full.log 2>&1
cat gpurun_out/${T}_ab.log; tail -3 gpurun_out/${T}_gpu_tests.log; grep -v amdgpu gpurun_out/${T}_pmc.log | grep "k_project\|k_composite\|==" 
