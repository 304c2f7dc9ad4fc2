"""scratch: per-rank frame time of the strip partition, measured strip by strip on one GPU"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth, multigpu
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5_830_000
W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (1920, 1080)
sc = synth.make_scene(N, seed=0xC0FFEE + 2)
ss = mgs.SplatSet.from_arrays(**sc); scene = mgs.Scene(0); scene.add_instance(ss); scene.commit()
for G in (1, 2, 4, 8):
    worst = []
    for r in range(G):
        b, e = multigpu.strip_rows(H, G, r)
        ts = []
        for i in range(0, 16):
            eye = synth.orbit_pose(i)
            V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
            p = capi.default_params(W, H); capi.set_camera(p, V, P, eye); p.collect_timings = 1
            if G > 1: p.strip_row_begin, p.strip_row_end = b, e
            o = scene.render(p); ts.append(list(o.stage_ms)[:6])
        worst.append(np.array(ts[4:]).mean(axis=0))
    w = np.array(worst)
    print(f"G={G}: slowest strip total {w[:,5].max():.3f} ms  (stages of slowest: {w[w[:,5].argmax()].round(3)})  mean {w[:,5].mean():.3f} -> {1e3/w[:,5].max():.0f} fps, speedup {worst0/w[:,5].max():.2f}x" if G>1 else f"G=1: total {w[0,5]:.3f} ms stages {w[0].round(3)}")
    if G == 1: worst0 = w[0,5]
