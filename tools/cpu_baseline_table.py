"""SURVEY.md §8d "CPU baseline beside it": the reference's CPU sorter (SplatSorterAsync::innerSort,
src/splat_sorter_async.cpp:92-141, restated in oracle/ as orc_cpu_sort) timed on this host for the BASELINE scene sizes:
distance loop + std::sort, N = 0.5 / 1.03 / 5.83 / 46.64 M (the last = 8 instances of the garden-sized set).
Writes gpurun_out/cpu_baseline_table.json.  Test infrastructure only (uses oracle/)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import binding as ob
from vk_gaussian_splatting_amd import synth

ob.build()
eye = synth.orbit_pose(0)
fwd = -np.asarray(eye, np.float32) / np.linalg.norm(eye)
rows = []
garden = None
for name, n, inst in (("syn_flowers", 500_000, 1), ("syn_train", 1_030_000, 1), ("syn_garden", 5_830_000, 1), ("syn_garden x8", 5_830_000, 8)):
    pos = synth.make_scene(n, seed=0xC0FFEE + len(rows))["positions"] if not (inst == 8 and garden is not None) else garden
    if n == 5_830_000:
        garden = pos
    sets = []
    for k in range(inst):
        M = np.eye(4, dtype=np.float32)
        M[0, 3], M[2, 3] = 12.0 * (k % 4), 12.0 * (k // 4)
        sets.append((pos, None if inst == 1 else M))
    best = None
    for _ in range(2):
        _, _, dms, sms = ob.cpu_sort(fwd, eye, sets, threads=0)
        if best is None or dms + sms < best[0] + best[1]:
            best = (dms, sms)
    rows.append(dict(scene=name, splats=n * inst, dist_ms=best[0], sort_ms=best[1], msplats_per_s=n * inst / (best[0] + best[1]) / 1e3))
    print(rows[-1], flush=True)
out = dict(host_threads=os.cpu_count(), what="orc_cpu_sort: distance loop on all threads (8192-element batches), "
           "std::sort(std::execution::par_unseq) with the reference's comparator — serial unless libstdc++ finds TBB (it does not here)",
           rows=rows)
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "cpu_baseline_table.json"), "w"), indent=1)
