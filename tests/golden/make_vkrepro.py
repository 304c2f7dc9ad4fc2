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
Round 4: four more cases, each in its own directory vkrepro/<case>/ (repro.vkgs + camera.json + expected_rgba16f.npy at 320x180,
all on ../scene.ply), chosen where a misreading shared by kernels, oracle and tests would most likely hide — see CASES below and
the command lines in camera.json["reference_command"]; tools/compare_vkrepro.py prints the PSNR of a reference screenshot.
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


# case -> what differs from the base fixture.  "ui": what has to be set by hand on the Vulkan box because the project file does not
# carry it (vkgs_project_writer.cpp:113-146 lists what it does carry).
CASES = {
    # the reference's DEFAULT storage (src/parameters.h:88-89): SH and colour as uint8 — every published number uses it
    "u8_storage": dict(sh_format=2, rgba_format=2, eye=[3.2, 1.1, 2.4], fov=60.0, model=0, pipeline=1, ms=0, instances=None,
                       ui="none (splatsGlobals carries shFormat / rgbaFormat = 2)"),
    # CAMERA_FISHEYE through the 3DGUT mesh pipeline at a wide field of view: dist-stage validity cull, fisheye rays, unscented projection
    "fisheye150_3dgut": dict(sh_format=0, rgba_format=0, eye=[1.4, 0.5, 1.1], fov=150.0, model=1, pipeline=4, ms=0, instances=None,
                             ui="none (camera.model = 1, renderer.pipeline = 4)"),
    # MS_ANTIALIASING (threedgs.h.slang:63-76): opacity compensation by sqrt(det / det_blurred)
    "msaa_3dgs": dict(sh_format=0, rgba_format=0, eye=[3.2, 1.1, 2.4], fov=60.0, model=0, pipeline=1, ms=1, instances=None,
                      ui="tick Rasterization > 'Mip splatting antialiasing' (prmRaster.msAntialiasing is not saved in .vkgs)"),
    # two instances of one splat set with translation / rotation / non-uniform scale: global ids, per-instance transforms, unified sort
    "two_instances_trs": dict(sh_format=0, rgba_format=0, eye=[4.4, 1.6, 3.3], fov=60.0, model=0, pipeline=1, ms=0,
                              instances=[dict(position=(-1.2, 0.0, 0.3), rotation=(0.0, 35.0, 0.0), scale=(1.0, 1.0, 1.0)),
                                         dict(position=(1.5, 0.2, -0.4), rotation=(20.0, -50.0, 10.0), scale=(0.7, 1.1, 0.9))],
                              ui="none (splats[] carries position / rotation / scale)"),
}
CW, CH = 320, 180


def case_project(name, ply):
    """the Project of a case (what repro.vkgs holds)"""
    c = CASES[name]
    cam = cameras.Camera()
    cam.eye, cam.ctr, cam.up = np.array(c["eye"], np.float32), np.zeros(3, np.float32), np.array([0, 1, 0], np.float32)
    cam.fov, cam.clip, cam.model = c["fov"], (0.1, 2000.0), c["model"]
    inst = ([project.SplatInstance(0, "scene")] if c["instances"] is None else
            [project.SplatInstance(0, f"scene{i}", tuple(t["position"]), tuple(t["rotation"]), tuple(t["scale"])) for i, t in enumerate(c["instances"])])
    return project.Project(camera=cam, cameras=[cam], sh_format=c["sh_format"], rgba_format=c["rgba_format"],
                           renderer={"maxShDegree": 3, "frustumCulling": 1, "sortingMethod": 0, "sizeCulling": 0, "pipeline": c["pipeline"]},
                           splat_sets={0: ply}, instances=inst)


def render_case(name, pr):
    """the oracle's frame of a case from its loaded Project (tests re-render through this and compare with the committed .npy)"""
    from vk_gaussian_splatting_amd import capi
    c = CASES[name]
    V, P = pr.camera.matrices(CW, CH, False)
    back = mgs.SplatSet.load(pr.splat_sets[0]).arrays()
    arrays = {k: back[k] for k in ("positions", "f_dc", "f_rest", "opacity", "scale", "rotation")}
    ps = ob.PreparedSet(arrays, sh_format=pr.sh_format, rgba_format=pr.rgba_format)
    mats = [capi.compute_transform(i.scale, i.rotation, i.position)[0] for i in pr.instances]
    inst = ob.make_instances([(ps, m) for m in mats])
    gut = c["pipeline"] in (4, 5)
    ks, vs = ob.sort_stable(*ob.key_cull(ob.make_frame(V, P, pr.camera.eye, CW, CH, camera_model=c["model"], pipeline_3dgut=int(gut)), inst))
    fr = ob.make_frame(V, P, pr.camera.eye, CW, CH, target_fp16=1, ms_antialiasing=c["ms"], camera_model=c["model"], pipeline_3dgut=int(gut))
    img, st = (ob.render_gut(fr, inst, vs) if gut else ob.render(fr, inst, order=vs))
    return img.astype(np.float16), st, (V, P)


def make_cases():
    for name, c in CASES.items():
        d = os.path.join(OUT, name)
        os.makedirs(d, exist_ok=True)
        pr = case_project(name, os.path.join(OUT, "scene.ply"))
        project.save_project(pr, os.path.join(d, "repro.vkgs"))
        pr = project.load_project(os.path.join(d, "repro.vkgs"))  # what a reader of the file sees
        img, st, (V, P) = render_case(name, pr)
        np.save(os.path.join(d, "expected_rgba16f.npy"), img)
        json.dump({"width": CW, "height": CH, "eye": [float(x) for x in pr.camera.eye], "ctr": [0.0, 0.0, 0.0], "up": [0.0, 1.0, 0.0],
                   "fov_deg": pr.camera.fov, "camera_model": pr.camera.model, "clip": [0.1, 2000.0],
                   "view_glm_column_major": np.asarray(V, np.float32).T.reshape(-1).tolist(),
                   "proj_glm_column_major": np.asarray(P, np.float32).T.reshape(-1).tolist(),
                   "frame": {"splatScale": 1.0, "frustumDilation": 0.2, "alphaCullThreshold": 1.0 / 255.0, "shDegree": 3,
                             "sorting": "GPU radix, back to front", "colourTarget": "RGBA16F", "msAntialiasing": bool(c["ms"]),
                             "pipeline": c["pipeline"], "shFormat": c["sh_format"], "rgbaFormat": c["rgba_format"]},
                   "ui_settings_not_in_the_project_file": c["ui"],
                   "reference_command": f"vk_gaussian_splatting tests/golden/vkrepro/{name}/repro.vkgs   # viewport {CW}x{CH}; {c['ui']}; then "
                                        f"File > Save image as <name>.hdr (RGBA32F read-back, gaussian_splatting_ui.cpp:508-540); "
                                        f"python tools/compare_vkrepro.py {name} <name>.hdr"},
                  open(os.path.join(d, "camera.json"), "w"), indent=1)
        print(name, st)


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
    make_cases()


if __name__ == "__main__":
    main()
