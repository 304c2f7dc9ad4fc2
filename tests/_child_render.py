"""child process of test_gpu_parity.test_binning_paths_bit_identical: libmgs reads its MGS_* knobs once per process,
so every configuration renders in its own interpreter and prints the SHA-1 of the frames"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import vk_gaussian_splatting_amd as mgs  # noqa: E402
from vk_gaussian_splatting_amd import capi, synth  # noqa: E402

n, w, h = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
sc = synth.make_scene(n, seed=77)
sc["scale"][: n // 50] += 2.5  # some splats that span many bins
scene = mgs.Scene(0)
scene.add_instance(mgs.SplatSet.from_arrays(**sc))
scene.commit()
hh = hashlib.sha1()
for pose in (1, 17, 40):
    eye = synth.orbit_pose(pose)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, w, h)
    p = capi.default_params(w, h)
    capi.set_camera(p, V, P, eye)
    o = scene.render(p)
    hh.update(np.ascontiguousarray(scene.download_frame(p)).tobytes())
    hh.update(np.array([o.sorted_count, o.tile_pairs if os.environ.get("MGS_BIN_SHIFT") is None else 0], np.uint64).tobytes())
print("FRAMES_SHA1", hh.hexdigest())
