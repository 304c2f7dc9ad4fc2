#!/bin/bash
# MGS_ALPHA_SUM frame time against the bin size (MGS_BIN_SHIFT=x,y: bins of (16<<x) x (16<<y) px): results in gpurun_out/alpha_sum_sweep.log
mkdir -p gpurun_out
for bs in "4,3" "3,2" "2,1" "1,0" "2,2" "3,3"; do
  echo "=== MGS_BIN_SHIFT=$bs" >> gpurun_out/alpha_sum_sweep.log
  MGS_BIN_SHIFT=$bs timeout 300 python bench.py --no-cpu-baseline --alpha-sum --steps 16 --warmup 4 --inflight 1 2>&1 | grep -v amdgpu | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read())
print('fps %.1f' % d['value'], {k: round(v, 3) for k, v in d['stage_ms_single_stream'].items()}, 'D %.2fM' % (d['visible_splats']['tile_pairs'] / 1e6), 'err', d['error_flags'], d.get('parity'))
" >> gpurun_out/alpha_sum_sweep.log 2>&1
done
cat gpurun_out/alpha_sum_sweep.log
