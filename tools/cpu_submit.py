"""how long does the host take to SUBMIT a frame (mgs_render is asynchronous)?  Small scene so the GPU is never the limiter."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth
sc = synth.make_scene(20000, seed=3)
scene = mgs.Scene(0); scene.add_instance(mgs.SplatSet.from_arrays(**sc)); scene.commit()
W, H = 256, 256
eye = synth.orbit_pose(3)
V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
for timings in (0, 2):
    p = capi.default_params(W, H); capi.set_camera(p, V, P, eye); p.collect_timings = timings
    for _ in range(50): scene.render(p)
    scene.sync()
    t0 = time.perf_counter()
    for _ in range(400): scene.render(p)
    t1 = time.perf_counter()
    scene.sync()
    t2 = time.perf_counter()
    print(f"collect_timings={timings}: submit {1e6*(t1-t0)/400:.1f} us/frame on the host, drained after {1e3*(t2-t1):.2f} ms more")
