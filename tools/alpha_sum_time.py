"""frame time of the additive-alpha mode (MGS_ALPHA_SUM: no early termination) on the garden-sized scene, and its parity with the
default mode's colour: python tools/alpha_sum_time.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth
W, H = 1920, 1080
sc = synth.make_scene(5_830_000, seed=0xC0FFEE + 2)
s = mgs.Scene(0); s.add_instance(mgs.SplatSet.from_arrays(**sc)); s.commit()
def params(i, mode):
    eye = synth.orbit_pose(i)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
    p = capi.default_params(W, H); capi.set_camera(p, V, P, eye); p.alpha_mode = mode; return p
for mode, name in ((capi.ALPHA_COVERAGE, "coverage (default)"), (capi.ALPHA_SUM, "sum")):
    for i in range(4): s.render(params(i, mode))
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(32): s.render(params(i, mode))
    torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 32
    print(f"alpha {name}: {dt * 1e3:.3f} ms per frame")
a = s.render(params(3, capi.ALPHA_COVERAGE)); fa = s.download_frame(params(3, capi.ALPHA_COVERAGE)).astype(np.float32)
b = s.render(params(3, capi.ALPHA_SUM)); fb = s.download_frame(params(3, capi.ALPHA_SUM)).astype(np.float32)
print("rgb max abs difference between the modes:", np.abs(fa[..., :3] - fb[..., :3]).max(), " alpha sum max:", fb[..., 3].max())
for mode in (capi.ALPHA_COVERAGE, capi.ALPHA_SUM):
    o = s.render(params(3, mode), want_stats=True)
    print(f"mode {mode}: sorted {o.sorted_count}, list entries {o.tile_pairs}, scanned {o.scanned_entries}, staged {o.shaded_count}")
