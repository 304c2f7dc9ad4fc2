#!/bin/bash
# everything behind profiles/${TAG}_*: run on the GPU box (gpurun), results land in gpurun_out/ (copy to profiles/ afterwards)
export TAG=${TAG:-r6_z}; R=${GRAFT_REPO_ROOT:-/root/repo}; cd $R; mkdir -p gpurun_out
python bench.py 2>/dev/null | grep '^{' | tail -1 > gpurun_out/${TAG}_bench_if3.json
python bench.py --inflight 1 2>/dev/null | grep '^{' | tail -1 > gpurun_out/${TAG}_bench_if1.json
bash tools/collect_profiles.sh > gpurun_out/${TAG}_collect.log 2>&1
cp gpurun_out/pmc_sq_all.json gpurun_out/${TAG}_pmc_sq_all_kernels.json
cd $R; bash tools/bench_matrix.sh > gpurun_out/${TAG}_bench_matrix.log 2>&1
cd $R; bash tools/pmc_alpha_sum.sh > gpurun_out/${TAG}_pmc_alpha_sum.log 2>&1
cd $R; for g in 2 4 8; do python tools/strip_throughput.py 1920 1080 $g 3 > gpurun_out/${TAG}_strip_throughput_1920x1080_g${g}_k3.log 2>&1; done
python tools/strip_throughput.py 3840 2160 8 3 > gpurun_out/${TAG}_strip_throughput_3840x2160_g8_k3.log 2>&1
TAG=$TAG bash tools/strip_kernels.sh 1920 1080 34 38 > gpurun_out/${TAG}_strip_kernels.log 2>&1
ls -la gpurun_out | tail -30
