#!/usr/bin/env python3
"""mgs_render — one frame through the C ABI to .npy / .png (the "tools/mgs_render" caller of SURVEY.md §8b).

  python tools/mgs_render.py scene.ply|scene.spz|scene.splat|syn:<n> out.png [--size W H] [--eye x y z]
                             [--center x y z] [--fov deg] [--flip-y] [--sh-format 0|1|2] [--rgba-format 0|1|2]

PNG = linear RGB clamped to [0,1] over a black background, 8 bit, no tonemap — like the reference's
screenshot path (gaussian_splatting_ui.cpp:508-540).  Needs an MI355X.
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import vk_gaussian_splatting_amd as mgs  # noqa: E402
from vk_gaussian_splatting_amd import capi, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("scene")
    ap.add_argument("out")
    ap.add_argument("--size", type=int, nargs=2, default=[1920, 1080])
    ap.add_argument("--eye", type=float, nargs=3, default=[1.7, 1.5, 1.7])   # camera_set.h:48-53
    ap.add_argument("--center", type=float, nargs=3, default=[0, 0, 0])
    ap.add_argument("--fov", type=float, default=60.0)
    ap.add_argument("--flip-y", action="store_true")
    ap.add_argument("--sh-format", type=int, default=0)
    ap.add_argument("--rgba-format", type=int, default=0)
    a = ap.parse_args()
    if a.scene.startswith("syn:"):
        ss = mgs.SplatSet.from_arrays(**synth.make_scene(int(a.scene[4:])))
    else:
        ss = mgs.SplatSet.load(a.scene)
    scene = mgs.Scene(0)
    scene.add_instance(ss)
    scene.commit(a.sh_format, a.rgba_format)
    W, H = a.size
    V, P = mgs.camera_lookat_perspective(a.eye, a.center, [0, 1, 0], a.fov, 0.1, 2000.0, W, H, flip_y=a.flip_y)
    p = capi.default_params(W, H)
    capi.set_camera(p, V, P, a.eye)
    p.collect_timings = 1
    o = scene.render(p)
    img = scene.download_frame(p).astype(np.float32)
    print(f"{scene.splat_count} splats, {o.frustum_count} in frustum, {o.sorted_count} sorted, {o.tile_pairs} bin records, "
          f"{o.stage_ms[5]:.3f} ms on the GPU")
    if a.out.endswith(".npy"):
        np.save(a.out, img)
    else:
        from PIL import Image
        rgb = np.clip(img[..., :3], 0, 1)
        Image.fromarray((rgb * 255 + 0.5).astype(np.uint8)).save(a.out)


if __name__ == "__main__":
    main()
