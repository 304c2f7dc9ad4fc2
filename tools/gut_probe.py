import sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth
sc = synth.make_scene(5_830_000, seed=0xC0FFEE + 2)
ss = mgs.SplatSet.from_arrays(**sc); scene = mgs.Scene(0); scene.add_instance(ss); scene.commit()
for deg in (3, 0):
    ts = []
    for i in range(12):
        eye = synth.orbit_pose(i)
        V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, 1920, 1080)
        p = capi.default_params(1920, 1080); capi.set_camera(p, V, P, eye); p.collect_timings = 1; p.sh_degree = deg; p.pipeline = 1
        o = scene.render(p); ts.append(list(o.stage_ms)[:6])
    print("3DGUT sh_degree", deg, "stages", np.array(ts[4:]).mean(axis=0).round(3), "shaded", o.shaded_count, "scanned", o.scanned_entries, "sorted", o.sorted_count)
