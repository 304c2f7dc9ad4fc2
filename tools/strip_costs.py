"""per-strip counts and stage times of the tile-row strip partition, rendered one strip at a time on one GPU:
python tools/strip_costs.py [W H [G]]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth, multigpu
import sys
N = 5_830_000
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
sc = synth.make_scene(N, seed=0xC0FFEE + 2)
ss = mgs.SplatSet.from_arrays(**sc); scene = mgs.Scene(0); scene.add_instance(ss); scene.commit()
G = int(sys.argv[3]) if len(sys.argv) > 3 else 8
for r in range(G):
    b, e = multigpu.strip_rows(H, G, r)
    eye = synth.orbit_pose(3)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
    p = capi.default_params(W, H); capi.set_camera(p, V, P, eye); p.collect_timings = 1
    p.strip_row_begin, p.strip_row_end = b, e
    for _ in range(3): o = scene.render(p)
    print(f"strip {r} rows [{b},{e}) frustum {o.frustum_count} sorted {o.sorted_count} pairs {o.tile_pairs} stages {np.array(list(o.stage_ms)[:6]).round(3)}")
