"""Generates tests/golden/full_size_garden.npz: what the CPU oracle says about the FULL benchmark configuration
(syn_garden, 5.83 M splats, SH degree 3, fp32 storage, 1920x1080 and 3840x2160) — compact enough to commit:

  * per pose (the four poses of DESIGN.md §5: 0, 17, 42, 53): the visible count, SHA-1 of the sorted key stream
    and SHA-1 of the sorted id stream (ties between equal keys in ascending caller id — the GPU test
    canonicalises its stream the same way, so the hash does not depend on the library's storage order);
  * per pose two 256x256 crops of the oracle's frame (RGBA16F target semantics, stored as fp16): one through the
    densest part of the object, one off-centre across background splats.  Pixels are independent given the draw
    order, so a crop rendered alone equals the crop of the full frame bit for bit (orc_render_window).
  * one 3840x2160 pose (configs[3]) with two 256x256 crops.

Needs only the CPU oracle (no GPU, no reference): python tests/golden/make_full_size_fixture.py   (~10 min, one core)
"""
import hashlib
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
import vk_gaussian_splatting_amd as mgs  # noqa: E402  (host-side camera helper only)
from vk_gaussian_splatting_amd import synth  # noqa: E402

N = 5_830_000
SEED = 0xC0FFEE + 2  # syn_garden
# (pose, width, height, [crop windows x0,y0,x1,y1 inclusive])
VIEWS = [
    (0, 1920, 1080, [(832, 412, 1087, 667), (200, 100, 455, 355)]),
    (17, 1920, 1080, [(832, 412, 1087, 667), (1500, 700, 1755, 955)]),
    (42, 1920, 1080, [(700, 380, 955, 635), (1300, 150, 1555, 405)]),
    (53, 1920, 1080, [(960, 440, 1215, 695), (60, 760, 315, 1015)]),
    (5, 3840, 2160, [(1792, 952, 2047, 1207), (400, 300, 655, 555)]),
]


def sha(a):
    return hashlib.sha1(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    t0 = time.time()
    sc = synth.make_scene(N, seed=SEED)
    ps = ob.PreparedSet(sc)
    inst = ob.make_instances([(ps, None)])
    out = {"n": np.int64(N), "seed": np.int64(SEED)}
    for vi, (pose, W, H, wins) in enumerate(VIEWS):
        eye = synth.orbit_pose(pose)
        V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
        fr = ob.make_frame(V, P, eye, W, H, target_fp16=1)
        k, i = ob.key_cull(fr, inst)  # ascending caller id
        ks, vs = ob.sort_stable(k, i)  # ties stay in ascending caller id
        out[f"v{vi}_pose"] = np.int64(pose)
        out[f"v{vi}_size"] = np.array([W, H], np.int64)
        out[f"v{vi}_visible"] = np.int64(ks.size)
        out[f"v{vi}_sha_keys"] = np.array(sha(ks))
        out[f"v{vi}_sha_ids"] = np.array(sha(vs))
        out[f"v{vi}_windows"] = np.array(wins, np.int64)
        for wi, w in enumerate(wins):
            t = time.time()
            img, frags = ob.render_window(fr, inst, vs, w)
            out[f"v{vi}_crop{wi}"] = img.astype(np.float16)
            print(f"view {vi} pose {pose} {W}x{H} window {w}: {frags/1e6:.1f} M fragments, {time.time()-t:.1f} s, "
                  f"mean alpha {img[..., 3].mean():.3f}", flush=True)
    np.savez_compressed(os.path.join(HERE, "full_size_garden.npz"), **out)
    print("done in %.0f s" % (time.time() - t0))


if __name__ == "__main__":
    main()
