"""Generates tests/golden/vkrepro/: a fixture a maintainer with a Vulkan box can feed to the REFERENCE itself
(SURVEY.md §8c last row: ".ply + JSON camera + expected .npy"), so that the oracle — and with it every parity claim below
the ingest — can one day be pinned against real VK3DGSR output.

  scene.ply        2 500 synthetic splats, INRIA layout (RDF, as trained models are), SH degree 3
  repro.vkgs       the reference's own project format (VkgsProjectWriter, version 5): loads scene.ply, fp32 storage,
                   the camera below, GPU sorting, default raster settings -> `vk_gaussian_splatting repro.vkgs`
  camera.json      the same camera as eye/centre/up/fov/clip, and the exact view / projection matrices (glm column-major)
                   the expected frame was rendered with, the resolution and the frame knobs
  expected_rgba16f.npy   [270][480][4] float16: the CPU oracle's frame (default pipeline = 3DGS mesh, back-to-front 'over' into
                   an RGBA16F target), row 0 = NDC y -1
  expected_3dgut_rgba16f.npy   the same for PIPELINE_MESH_3DGUT (pinhole, conic extents)
To compare on the Vulkan box: render 480x270, dump COLOR_MAIN as float (the .hdr screenshot path,
gaussian_splatting_ui.cpp:520-535), PSNR as image_compare_metric.comp.slang:116-122.  Expect >= 40 dB if the oracle
reads the shaders correctly (hardware rasterisation snaps vertices to 1/256 px; ties between equal depth keys are
drawn in nondeterministic order by the reference).
Needs only the CPU oracle:  python tests/golden/make_vkrepro.py"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
import vk_gaussian_splatting_amd as mgs  # noqa: E402
from vk_gaussian_splatting_amd import synth, project, cameras  # noqa: E402

OUT = os.path.join(HERE, "vkrepro")
W, H, N = 480, 270, 2500


def main():
    os.makedirs(OUT, exist_ok=True)
    sc = synth.make_scene(N, seed=4242)
    synth.write_ply(os.path.join(OUT, "scene.ply"), sc)
    cam = cameras.Camera()
    cam.eye, cam.ctr, cam.up = np.array([3.2, 1.1, 2.4], np.float32), np.zeros(3, np.float32), np.array([0, 1, 0], np.float32)
    cam.fov, cam.clip = 60.0, (0.1, 2000.0)
    pr = project.Project(camera=cam, cameras=[cam], sh_format=0, rgba_format=0,
                         renderer={"maxShDegree": 3, "frustumCulling": 1, "sortingMethod": 0, "sizeCulling": 0},
                         splat_sets={0: os.path.join(OUT, "scene.ply")}, instances=[project.SplatInstance(0, "scene")])
    project.save_project(pr, os.path.join(OUT, "repro.vkgs"))
    V, P = cam.matrices(W, H, False)
    json.dump({"width": W, "height": H, "eye": cam.eye.tolist(), "ctr": cam.ctr.tolist(), "up": cam.up.tolist(), "fov_deg": cam.fov,
               "clip": list(cam.clip), "view_glm_column_major": np.asarray(V, np.float32).T.reshape(-1).tolist(),
               "proj_glm_column_major": np.asarray(P, np.float32).T.reshape(-1).tolist(),
               "frame": {"splatScale": 1.0, "frustumDilation": 0.2, "alphaCullThreshold": 1.0 / 255.0, "shDegree": 3,
                         "sorting": "GPU radix, back to front", "colourTarget": "RGBA16F", "storage": "fp32 buffers"}},
              open(os.path.join(OUT, "camera.json"), "w"), indent=1)
    # the loader's view of the file (RDF -> RUB) == the arrays it was written from
    back = mgs.SplatSet.load(os.path.join(OUT, "scene.ply")).arrays()
    arrays = {k: back[k] for k in ("positions", "f_dc", "f_rest", "opacity", "scale", "rotation")}
    ps = ob.PreparedSet(arrays)
    inst = ob.make_instances([(ps, None)])
    fr = ob.make_frame(V, P, cam.eye, W, H, target_fp16=1)
    ks, vs = ob.sort_stable(*ob.key_cull(fr, inst))
    img, st = ob.render(fr, inst, order=vs)
    np.save(os.path.join(OUT, "expected_rgba16f.npy"), img.astype(np.float16))
    g, sg = ob.render_gut(fr, inst, vs)
    np.save(os.path.join(OUT, "expected_3dgut_rgba16f.npy"), g.astype(np.float16))
    print("3DGS", st, "3DGUT", sg)


if __name__ == "__main__":
    main()
