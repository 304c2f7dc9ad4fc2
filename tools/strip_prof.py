import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth
W, H = 1920, 1080
sc = synth.make_scene(5_830_000, seed=1)
scene = mgs.Scene(0)
scene.add_instance(mgs.SplatSet.from_arrays(**sc))
scene.commit()
b, e = int(sys.argv[1]), int(sys.argv[2])
for i in range(264):
    eye = synth.orbit_pose(i % 64)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
    p = capi.default_params(W, H)
    capi.set_camera(p, V, P, eye)
    p.strip_row_begin, p.strip_row_end = b, e
    scene.render(p)
scene.sync()
