"""Per-pose sorted count and sort-stage time over the 64-pose orbit (HIP events, serial frames): does a pose whose sorted pairs
exceed one residency wave of sort partitions (1 024 x 4 096) pay a second round?   python tools/pose_sort_times.py [--splats N]"""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth
ap = argparse.ArgumentParser()
ap.add_argument("--splats", type=int, default=5_830_000)
a = ap.parse_args()
sc = synth.make_scene(a.splats, seed=0xC0FFEE + 2)
scene = mgs.Scene(0); scene.add_instance(mgs.SplatSet.from_arrays(**sc)); scene.commit()
W, H = 1920, 1080
rows = []
for lap in range(3):
    for i in range(64):
        eye = synth.orbit_pose(i)
        V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
        p = capi.default_params(W, H); capi.set_camera(p, V, P, eye); p.collect_timings = 2
        o = scene.render(p, want_stats=True); scene.sync()
        t = scene.timings_all(0)
        if lap: rows.append((i, o.sorted_count, t[1] * 1000, t[0] * 1000, t[2] * 1000, t[4] * 1000))
r = np.array(rows)
for i in range(64):
    q = r[r[:, 0] == i]
    print(f"pose {i:2d} V {int(q[0,1]):8d} parts {(int(q[0,1]) + 4095) // 4096:5d} sort {q[:,2].mean():6.1f} project {q[:,3].mean():6.1f} bin {q[:,4].mean():6.1f} composite {q[:,5].mean():6.1f}")
small = r[(r[:, 1] + 4095) // 4096 <= 1024]; big = r[(r[:, 1] + 4095) // 4096 > 1024]
print(f"poses with <= 1024 partitions: {len(small)//2} sort mean {small[:,2].mean() if len(small) else 0:.1f} us;  > 1024: {len(big)//2} sort mean {big[:,2].mean() if len(big) else 0:.1f} us")
