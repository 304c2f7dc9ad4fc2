"""scratch: compositor sensitivity probes"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_030_000
sc = synth.make_scene(N, seed=0xC0FFEE + 2)
ss = mgs.SplatSet.from_arrays(**sc); scene = mgs.Scene(0); scene.add_instance(ss); scene.commit()
def run(tag, W=1920, H=1080, **kw):
    ts = []
    for i in range(0, 20):
        eye = synth.orbit_pose(i)
        V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
        p = capi.default_params(W, H); capi.set_camera(p, V, P, eye); p.collect_timings = 1
        for k, v in kw.items(): setattr(p, k, v)
        o = scene.render(p); ts.append(list(o.stage_ms)[:6] + [o.sorted_count, o.tile_pairs])
    r = np.array(ts[4:]).mean(axis=0)
    print(f"{tag:28s} composite {r[4]:.3f} total {r[5]:.3f}  V {r[6]/1e6:.2f}M D {r[7]/1e6:.2f}M")
run("default")
run("alpha_sum (no early out)", alpha_mode=1)
run("splat_scale 0.5", splat_scale=0.5)
run("splat_scale 0.25", splat_scale=0.25)
run("alpha_cull 0.5", alpha_cull_threshold=0.5)
run("960x540", W=960, H=540)
run("3840x2160", W=3840, H=2160)
