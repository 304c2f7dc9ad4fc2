"""how much of the key pass a strip still runs: frustum survivors / sorted splats per strip against the full frame, and the number
of 2048-splat partitions the partition test (partition_cull.h) lets through (derived from the survivor counts of the sort-only
hook)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth
W, H = 1920, 1080
sc = synth.make_scene(5_830_000, seed=1)
scene = mgs.Scene(0)
scene.add_instance(mgs.SplatSet.from_arrays(**sc))
scene.commit()
eye = synth.orbit_pose(0)
V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
p = capi.default_params(W, H)
capi.set_camera(p, V, P, eye)
p.collect_timings = 1
for rows in [(0, 68), (0, 16), (16, 24), (24, 29), (29, 34), (34, 38), (38, 43), (43, 48), (48, 68)]:
    p.strip_row_begin, p.strip_row_end = rows
    for _ in range(3):
        out = scene.render(p, want_stats=True)
    print(rows, "frustum", out.frustum_count, "sorted", out.sorted_count, "stage ms", [round(float(x), 4) for x in out.stage_ms[:7]])
