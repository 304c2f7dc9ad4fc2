"""scratch: isolated stage timings on the GPU box"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5_830_000
sc = synth.make_scene(N, seed=0xC0FFEE + 2)
ss = mgs.SplatSet.from_arrays(**sc); scene = mgs.Scene(0); scene.add_instance(ss); scene.commit()
W, H = 1920, 1080
res = []
for i in range(0, 24):
    eye = synth.orbit_pose(i)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
    p = capi.default_params(W, H); capi.set_camera(p, V, P, eye)
    so = scene.sort_keys(p)
    res.append((so.key_ms, so.sort_ms, so.count, so.passes))
print("sample sort stats (last frame): slices %d buckets %d streamed buckets %d" % tuple(so.reserved))
r = np.array(res[4:])
print("sort-only hook: key(phase1 only) %.3f ms  sort %.3f ms  count %.0f passes %.1f -> %.2f Gkeys/s" % (r[:,0].mean(), r[:,1].mean(), r[:,2].mean(), r[:,3].mean(), r[:,2].mean()/r[:,1].mean()/1e6))
if os.environ.get("PROBE_SORT_ONLY"):
    sys.exit(0)
for deg in (3, 0):
    ts = []
    for i in range(0, 24):
        eye = synth.orbit_pose(i)
        V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
        p = capi.default_params(W, H); capi.set_camera(p, V, P, eye); p.collect_timings = 1; p.sh_degree = deg
        o = scene.render(p); ts.append(list(o.stage_ms)[:6])
    print("sh_degree", deg, "stages", np.array(ts[4:]).mean(axis=0).round(3))
