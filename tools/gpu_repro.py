import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth
def cam(i, W, H):
    eye = synth.orbit_pose(i)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
    p = capi.default_params(W, H); capi.set_camera(p, V, P, eye); return p
scA = synth.make_scene(60000, seed=21); sA = mgs.SplatSet.from_arrays(**scA); A = mgs.Scene(0); A.add_instance(sA); A.commit()
for (i, W, H) in [(3, 640, 480), (30, 333, 217), (50, 1280, 720)]:
    o = A.render(cam(i, W, H), want_stats=True); print("A", W, H, o.frustum_count, o.sorted_count, o.tile_pairs)
scB = synth.make_scene(250000, seed=0xC0FFEE + 2); sB = mgs.SplatSet.from_arrays(**scB); B = mgs.Scene(0); B.add_instance(sB); B.commit()
for rep in range(3):
    o = B.render(cam(0, 1920, 1080), want_stats=True); print("B", o.frustum_count, o.sorted_count, o.tile_pairs, o.error_flags)
o = A.render(cam(3, 640, 480), want_stats=True); print("A again", o.frustum_count, o.sorted_count, o.tile_pairs)
