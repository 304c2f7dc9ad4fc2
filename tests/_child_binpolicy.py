"""child process of test_gpu_parity.test_adaptive_bin_size_*: renders a 72-frame sequence on one context and prints, per frame, the
number of list entries the binning built (it changes when the bin size does) and the SHA-1 of all frames.  The parent runs it with
the adaptive bin size on (default) and off (MGS_BIN_ADAPT=0): libmgs reads the knob once per process.   usage: _child_binpolicy.py KIND"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import vk_gaussian_splatting_amd as mgs  # noqa: E402
from vk_gaussian_splatting_amd import capi, synth  # noqa: E402

kind = sys.argv[1]
n, W, H = 250_000, 1280, 720
sc = synth.make_scene(n, seed=31)
if kind == "thin":      # translucent: regions never saturate and walk their bins' whole lists
    sc["opacity"] = (sc["opacity"] - 3.0).astype(np.float32)
elif kind == "opaque":  # large opaque splats: every region saturates after a few entries
    sc["opacity"] = (np.abs(sc["opacity"]) + 4.0).astype(np.float32)
    sc["scale"] = (sc["scale"] + 1.2).astype(np.float32)
scene = mgs.Scene(0)
scene.add_instance(mgs.SplatSet.from_arrays(**sc))
scene.commit()
hh = hashlib.sha1()
ds = []
for f in range(72):
    eye = synth.orbit_pose((f % 4) * 5)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
    p = capi.default_params(W, H)
    capi.set_camera(p, V, P, eye)
    o = scene.render(p, want_stats=True)
    assert o.error_flags == 0
    hh.update(np.ascontiguousarray(scene.download_frame(p)).tobytes())
    if f % 4 == 0:
        ds.append(int(o.tile_pairs))
print("PAIRS", " ".join(str(d) for d in ds))
print("FRAMES_SHA1", hh.hexdigest())
scene.close()
