"""MGS_ALPHA_SUM: the alpha sums of two builds of the library on the garden-sized frame (fp32 target), e.g. the polynomial walk
against the (s, u) walk: python tools/alpha_sum_diff.py libA.so libB.so"""
import os, sys, subprocess
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) == 2:
    sys.path.insert(0, ROOT)
    import vk_gaussian_splatting_amd as mgs
    from vk_gaussian_splatting_amd import capi, synth
    W, H = 1920, 1080
    sc = synth.make_scene(5_830_000, seed=0xC0FFEE + 2)
    s = mgs.Scene(0); s.add_instance(mgs.SplatSet.from_arrays(**sc)); s.commit()
    eye = synth.orbit_pose(3)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
    p = capi.default_params(W, H); capi.set_camera(p, V, P, eye); p.alpha_mode = capi.ALPHA_SUM; p.target_format = capi.TARGET_RGBA32F
    s.render(p); a = s.download_frame(p)[..., 3].copy()
    s.render(p); b = s.download_frame(p)[..., 3]
    print("run-to-run identical:", bool(np.array_equal(a, b)))
    np.save(sys.argv[1], a)
    sys.exit(0)
out = []
for i, lib in enumerate(sys.argv[1:3]):
    f = f"/tmp/alpha_sum_{i}.npy"
    subprocess.run([sys.executable, __file__, f], env=dict(os.environ, MGS_LIB=lib), check=True)
    out.append(np.load(f).astype(np.float64))
a, b = out
rel = (b - a) / np.maximum(a, 1.0)
print("mean A / B: %.6f %.6f" % (a.mean(), b.mean()), " rel diff B - A: mean %.3e  p1 %.3e  p50 %.3e  p99 %.3e  min %.3e max %.3e"
      % (rel.mean(), *np.percentile(rel, [1, 50, 99]), rel.min(), rel.max()))
