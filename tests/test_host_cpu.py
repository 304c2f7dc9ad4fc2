"""CPU tests of the product's host side: C-ABI surface, ingest (own PLY/SPZ/.splat readers) against
the golden vectors made by the reference-compiled loaders, camera helpers, error behaviour.
No compute call is made here (there is no GPU in this container)."""
import os
import re
import struct

import numpy as np
import pytest

from conftest import GOLDEN, ROOT, lookat, persp
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth

FIELDS = ["positions", "f_dc", "f_rest", "opacity", "scale", "rotation"]


def test_library_exports_every_symbol_declared_in_header():
    lib = capi.load_library()
    hdr = open(os.path.join(ROOT, "include", "mgs.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = sorted(set(re.findall(r"\b(mgs_[a-z0-9_]+)\s*\(", hdr)))
    assert declared, "no declarations parsed"
    for name in declared:
        assert hasattr(lib, name), f"libmgs.so does not export {name}"
    assert sorted(capi.EXPORTED_SYMBOLS) == declared
    assert b"gfx950" in lib.mgs_version()


def test_struct_layouts_match_header_sizes():
    # MgsFrameParams: 16+16+3 floats, 2 ints, 3 floats, 12 ints + 6 reserved

    import ctypes
    assert ctypes.sizeof(capi.FrameParams) == 288  # ABI 3: + dof_mode, focus_dist, aperture, frame_sample_id, temporal_sampling, kernel_degree, 2 reserved
    assert ctypes.sizeof(capi.FrameOut) == 8 + 8 + 4 + 4 + 8 + 4 + 4 + 8 + 32 + 8
    assert ctypes.sizeof(capi.SortOut) == 32
    assert ctypes.sizeof(capi.SplatSetView) == 6 * 8 + 8 + 4 + 4


@pytest.mark.parametrize("name", ["ply_sh3", "ply_sh0", "ply_ascii", "ply_be", "spz_sh3", "spz_sh1", "spz_sh0"])
def test_ingest_matches_reference_loader_bit_for_bit(name, golden_meta):
    ext = ".spz" if name.startswith("spz") else ".ply"
    s = mgs.SplatSet.load(os.path.join(GOLDEN, f"ingest_{name}{ext}"))
    a = s.arrays()
    g = np.load(os.path.join(GOLDEN, f"ingest_{name}.npz"))
    assert a["count"] == golden_meta[name]["n"]
    assert a["sh_degree"] == int(g["sh_degree"])
    for k in FIELDS:
        assert a[k].shape == g[k].shape, k
        assert np.array_equal(a[k].view(np.uint32), g[k].view(np.uint32)), k  # bit-exact, NaN/inf safe


def test_uppercase_extension_and_ply_roundtrip(tmp_path):
    sc = synth.make_scene(257, seed=3)
    p = tmp_path / "Scene.PLY"
    synth.write_ply(str(p), sc)
    a = mgs.SplatSet.load(str(p)).arrays()
    for k in FIELDS:
        assert np.array_equal(a[k], sc[k].reshape(-1)), k  # write (RUB->RDF) then load (RDF->RUB) is the identity


def test_ply_partial_f_rest_means_degree_zero(tmp_path):
    """the reference keeps f_rest only if all 45 properties exist (ply_loader_async.cpp:383-396)"""
    sc = synth.make_scene(50, seed=4, sh_coeffs_per_channel=8)  # 24 f_rest_* properties
    p = tmp_path / "deg2.ply"
    synth.write_ply(str(p), sc)
    a = mgs.SplatSet.load(str(p)).arrays()
    assert a["f_rest"].size == 0 and a["sh_degree"] == 0 and a["count"] == 50


def test_splat_format_loader(tmp_path):
    """32-byte records (ply_loader_async.cpp:43-50,113-166): log scale, logit alpha, f_dc from colour, RDF->RUB"""
    rng = np.random.default_rng(5)
    n = 100
    pos = rng.normal(0, 1, (n, 3)).astype(np.float32)
    scl = np.exp(rng.normal(-3, 1, (n, 3))).astype(np.float32)
    col = rng.integers(0, 256, (n, 4), dtype=np.uint8)
    rot = rng.integers(0, 256, (n, 4), dtype=np.uint8)
    p = tmp_path / "a.splat"
    with open(p, "wb") as f:
        for i in range(n):
            f.write(pos[i].tobytes() + scl[i].tobytes() + col[i].tobytes() + rot[i].tobytes())
    a = mgs.SplatSet.load(str(p)).arrays()
    assert a["count"] == n and a["f_rest"].size == 0 and a["sh_degree"] == 0
    assert np.array_equal(a["positions"].reshape(n, 3), pos * np.array([1, -1, -1], np.float32))
    assert np.allclose(a["scale"].reshape(n, 3), np.log(scl), rtol=1e-6, atol=1e-6)
    q = (rot.astype(np.float32) - 128) / 128
    assert np.array_equal(a["rotation"].reshape(n, 4), q * np.array([1, 1, -1, -1], np.float32))
    assert np.allclose(a["f_dc"].reshape(n, 3), (col[:, :3] / np.float32(255) - 0.5) / 0.28209479177387814, atol=1e-5)
    al = np.clip(col[:, 3] / np.float32(255), 1e-6, 1 - 1e-6)
    assert np.allclose(a["opacity"], -np.log(1 / al - 1), rtol=1e-4, atol=1e-4)


def test_ingest_errors(tmp_path):
    with pytest.raises(mgs.MgsError) as e:
        mgs.SplatSet.load(str(tmp_path / "missing.ply"))
    assert e.value.code == -2  # MGS_ERR_IO
    bad = tmp_path / "bad.ply"
    bad.write_bytes(b"not a ply\n")
    with pytest.raises(mgs.MgsError) as e:
        mgs.SplatSet.load(str(bad))
    assert e.value.code == -3  # MGS_ERR_FORMAT
    nov = tmp_path / "mesh.ply"  # a valid PLY without the 3DGS properties
    nov.write_bytes(b"ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nproperty float y\nproperty float z\nend_header\n0 0 0\n")
    with pytest.raises(mgs.MgsError) as e:
        mgs.SplatSet.load(str(nov))
    assert e.value.code == -3
    sp = tmp_path / "odd.splat"
    sp.write_bytes(b"\0" * 33)
    with pytest.raises(mgs.MgsError) as e:
        mgs.SplatSet.load(str(sp))
    assert e.value.code == -3
    z = tmp_path / "x.spz"
    z.write_bytes(b"garbage")
    with pytest.raises(mgs.MgsError) as e:
        mgs.SplatSet.load(str(z))
    assert e.value.code == -3
    trunc = tmp_path / "trunc.ply"
    full = open(os.path.join(GOLDEN, "ingest_ply_sh0.ply"), "rb").read()
    trunc.write_bytes(full[: len(full) - 100])
    with pytest.raises(mgs.MgsError) as e:
        mgs.SplatSet.load(str(trunc))
    assert e.value.code == -3


def test_from_arrays_validation_and_roundtrip():
    sc = synth.make_scene(33, seed=6)
    a = mgs.SplatSet.from_arrays(**sc).arrays()
    for k in FIELDS:
        assert np.array_equal(a[k], sc[k].reshape(-1))
    assert a["sh_degree"] == 3 and a["f_rest_per_splat"] == 45
    sc0 = dict(sc, f_rest=None)
    assert mgs.SplatSet.from_arrays(**sc0).arrays()["sh_degree"] == 0
    with pytest.raises(mgs.MgsError):
        mgs.SplatSet.from_arrays(**dict(sc, f_rest=np.zeros((33, 7), np.float32)))  # not a multiple of 3
    with pytest.raises(mgs.MgsError):
        mgs.SplatSet.from_arrays(np.zeros((0, 3)), np.zeros((0, 3)), None, np.zeros(0), np.zeros((0, 3)), np.zeros((0, 4)))


def test_scene_creation_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(mgs.MgsError) as e:
        mgs.Scene(0)
    assert e.value.code == -4 and "no CPU fallback" in str(e.value)


def test_camera_helper_matches_numpy():
    eye = np.array([1.7, 1.5, 1.7], np.float32)  # default camera, camera_set.h:48-53
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, 1920, 1080)
    assert np.allclose(V, lookat(eye, [0, 0, 0], [0, 1, 0]), atol=1e-6)
    assert np.allclose(P, persp(60, 1920 / 1080, 0.1, 2000), rtol=1e-6, atol=1e-7)
    _, Pf = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, 1920, 1080, flip_y=True)
    assert Pf[1, 1] == -P[1, 1]
    # a point on the near plane maps to z=0, far plane to z=1 (clip z in [0,1])
    for zv, want in ((-0.1, 0.0), (-2000.0, 1.0)):
        c = P @ np.array([0, 0, zv, 1], np.float32)
        assert abs(c[2] / c[3] - want) < 1e-4


def test_compute_transform_is_trs():
    M, Mi = mgs.compute_transform([2, 1, 0.5], [0, 90, 0], [1, 2, 3])
    # rotation of +90 deg about y maps +x to -z
    assert np.allclose(M @ np.array([1, 0, 0, 1], np.float32), [1, 2, 3 - 2, 1], atol=1e-5)
    assert np.allclose(M @ Mi, np.eye(4), atol=1e-5)


def test_strip_partition_covers_all_rows():
    from vk_gaussian_splatting_amd import multigpu
    for h in (480, 1080, 2160, 17):
        for ws in (1, 2, 3, 4, 8):
            rows = multigpu.tile_rows(h)
            cover = []
            for r in range(ws):
                b, e = multigpu.strip_rows(h, ws, r)
                assert 0 <= b <= e <= rows
                cover += list(range(b, e))
            assert cover == list(range(rows))
            assert multigpu.strip_pixel_rows(h, ws) * ws >= h


def test_import_cameras_inria(tmp_path):
    """importCamerasINRIA (src/camera_set.h:219-270): RDF->RUB re-signing, up/at from the rotation columns"""
    import json
    from vk_gaussian_splatting_amd import cameras
    th = 0.3
    R = [[np.cos(th), 0.0, np.sin(th)], [0.0, 1.0, 0.0], [-np.sin(th), 0.0, np.cos(th)]]
    items = [dict(id=0, img_name="a.jpg", width=1959, height=1090, position=[1.0, 2.0, 3.0], rotation=R, fy=1160.0, fx=1159.0),
             dict(id=1, img_name="b.jpg", width=1959, height=1090, position=[0.0, -1.0, 4.0],
                  rotation=[[1, 0, 0], [0, 1, 0], [0, 0, 1]], fy=1160.0, fx=1159.0)]
    p = tmp_path / "cameras.json"
    p.write_text(json.dumps(items))
    cams = cameras.import_cameras_inria(str(p))
    assert len(cams) == 2 and cams[0].name == "a.jpg" and cams[0].fov == 60.0
    assert np.allclose(cams[0].eye, [1.0, -2.0, -3.0])
    assert np.allclose(cams[0].up, [0.0, 1.0, 0.0], atol=1e-6)                       # column 1: (-R01, R11, R21)
    assert np.allclose(cams[0].ctr - cams[0].eye, [np.sin(th), 0.0, -np.cos(th)], atol=1e-6)  # column 2: (R02, -R12, -R22)
    assert np.allclose(cams[1].ctr - cams[1].eye, [0, 0, -1]) and np.allclose(cams[1].eye, [0, 1, -4])
    V, P = cams[1].matrices(640, 480)
    assert np.allclose(V @ np.array([0, 1, -5, 1], np.float32), [0, 0, -1, 1], atol=1e-6)   # a point 1 unit ahead
    d = cameras.Camera()
    assert np.allclose(d.eye, [1.7, 1.5, 1.7]) and d.fov == 60.0 and d.clip == (0.1, 2000.0)
    (tmp_path / "bad.json").write_text("[{\"position\": [1,2], \"rotation\": [[1,0,0],[0,1,0],[0,0,1]]}]")
    with pytest.raises(ValueError):
        cameras.import_cameras_inria(str(tmp_path / "bad.json"))


def test_vkgs_project_roundtrip(tmp_path):
    """.vkgs (vkgs_project_writer.cpp:75-330 / vkgs_project_reader.cpp): sections the raster path consumes"""
    import json
    from vk_gaussian_splatting_amd import project, cameras
    sc = synth.make_scene(64, seed=8)
    synth.write_ply(str(tmp_path / "a.ply"), sc)
    doc = {"version": 5,
           "renderer": {"maxShDegree": 2, "frustumCulling": 2, "sizeCulling": 1, "sizeCullingMinPixels": 3.5,
                        "sortingMethod": 0, "pointCloudModeEnabled": True, "showShOnly": False, "pipeline": 1},
           "camera": {"model": 0, "eye": [2, 1, 2], "ctr": [0, 0.5, 0], "up": [0, 1, 0], "fov": 45.0, "clip": [0.05, 500.0]},
           "cameras": [{"eye": [1, 1, 1], "ctr": [0, 0, 0], "up": [0, 1, 0], "fov": 60.0, "clip": [0.1, 2000.0]}],
           "splatsGlobals": {"shFormat": 2, "rgbaFormat": 1},
           "splatSets": [{"id": 7, "path": "a.ply", "storage": 0, "shFormat": 2, "rgbaFormat": 1}],
           "splats": [{"splatSetId": 7, "name": "one", "position": [1, 2, 3], "rotation": [0, 90, 0], "scale": [2, 2, 2]},
                      {"splatSetId": 7, "name": "two"},
                      {"splatSetId": 99, "name": "dangling"}],
           "lights": {"assets": [], "instances": []}}
    (tmp_path / "scene.vkgs").write_text(json.dumps(doc))
    pr = project.load_project(str(tmp_path / "scene.vkgs"))
    assert pr.version == 5 and pr.sh_format == 2 and pr.rgba_format == 1
    assert pr.splat_sets == {7: str(tmp_path / "a.ply")}
    assert [i.name for i in pr.instances] == ["one", "two", "dangling"]
    assert pr.instances[0].rotation == (0, 90, 0) and pr.instances[1].scale == (1.0, 1.0, 1.0)
    assert pr.camera.fov == 45.0 and pr.camera.clip == (0.05, 500.0) and len(pr.cameras) == 1
    p = pr.frame_params(640, 480)
    assert (p.sh_degree, p.frustum_culling, p.size_culling, p.debug_flags) == (2, 2, 1, 1)
    assert abs(p.size_culling_min_pixels - 3.5) < 1e-6 and tuple(p.camera_pos) == (2.0, 1.0, 2.0)
    project.save_project(pr, str(tmp_path / "out" if False else tmp_path / "copy.vkgs"))
    pr2 = project.load_project(str(tmp_path / "copy.vkgs"))
    assert pr2.splat_sets == pr.splat_sets and pr2.renderer == pr.renderer and "lights" in pr2.extra
    assert [(i.splat_set_id, i.position, i.rotation, i.scale) for i in pr2.instances] == \
           [(i.splat_set_id, tuple(i.position), tuple(i.rotation), tuple(i.scale)) for i in pr.instances]
    (tmp_path / "new.vkgs").write_text(json.dumps({"version": 99}))
    with pytest.raises(ValueError):
        project.load_project(str(tmp_path / "new.vkgs"))
    legacy = {"splats": [{"path": "a.ply", "name": "old", "position": [0, 0, 0], "rotation": [0, 0, 0], "scale": [1, 1, 1]}]}
    (tmp_path / "legacy.vkgs").write_text(json.dumps(legacy))
    pl = project.load_project(str(tmp_path / "legacy.vkgs"))
    assert pl.version == 0 and pl.splat_sets == {0: str(tmp_path / "a.ply")} and pl.instances[0].name == "old"


def test_cpp_caller_builds_against_the_header_and_prints_usage():
    """examples/mgs_render.cpp: a plain C++ caller of include/mgs.h (what a maintainer of the C++ reference would write)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run(["make", "-C", os.path.join(root, "examples")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run([os.path.join(root, "examples", "mgs_render")], capture_output=True, text=True)
    assert r.returncode == 2 and "usage:" in r.stderr and "gfx950" in r.stderr


def test_div255_refinement_is_exact():
    """sh_eval.h::div255 — q = b*r; q + fma(-q,255,b)*r with r = fl(1/255) — equals the correctly rounded b/255 for every
    byte value (the uint8 SH / colour dequantisation, threedgs_particle_buffers.h.slang:82-86,128-131); exact rational
    arithmetic stands in for the device's fused multiply-adds"""
    from fractions import Fraction

    def r32(fr):
        x = np.float32(float(fr))
        cands = [x, np.nextafter(x, np.float32(np.inf)), np.nextafter(x, np.float32(-np.inf))]
        return np.float32(min(cands, key=lambda v: (abs(Fraction(float(v)) - fr), int(np.float32(v).view(np.uint32)) & 1)))

    r = np.float32(1.0) / np.float32(255.0)
    for b in range(256):
        fb = np.float32(b)
        q = np.float32(fb * r)
        rem = r32(Fraction(float(-q)) * 255 + Fraction(float(fb)))
        q2 = r32(Fraction(float(rem)) * Fraction(float(r)) + Fraction(float(q)))
        assert q2 == np.float32(fb / np.float32(255.0)), b


def test_balanced_strip_bounds_properties():
    """multigpu.balanced_bounds: ascending, covers all tile rows, deterministic, and actually balances a skewed cost profile"""
    from vk_gaussian_splatting_amd import multigpu
    rng = np.random.default_rng(5)
    for rows, world in ((68, 8), (135, 8), (68, 3), (5, 8), (1, 2)):
        cost = rng.random(rows) ** 4 * 1e5
        cost[rows // 2] += 3e5
        b = multigpu.balanced_bounds(cost, world)
        assert len(b) == world + 1 and b[0] == 0 and b[-1] == rows and all(b[i] <= b[i + 1] for i in range(world))
        assert b == multigpu.balanced_bounds(cost.copy(), world)
        if rows >= 4 * world:
            c = cost + 0.25 * cost.mean()
            per = [c[b[i]:b[i + 1]].sum() for i in range(world)]
            eq = [c[multigpu.strip_rows(rows * 16, world, r)[0]:multigpu.strip_rows(rows * 16, world, r)[1]].sum() for r in range(world)]
            assert max(per) <= max(eq) + 1e-6   # never worse than equal strips on the charged cost
            assert multigpu.padded_strip_rows(b) == max(b[i + 1] - b[i] for i in range(world)) * 16


def test_frame_params_defaults_cover_the_new_pipeline_fields():
    p = capi.default_params(640, 480)
    assert p.pipeline == capi.PIPELINE_3DGS and p.camera_model == capi.CAMERA_PINHOLE and p.extent_method == capi.EXTENT_CONIC
    assert abs(p.alpha_clamp - 0.99) < 1e-7 and abs(p.kernel_min_response - 0.0113) < 1e-7 and p.fov_rad == 0.0
    assert p.kernel_degree == 2 and p.dof_mode == 0 and abs(p.focus_dist - 1.3) < 1e-7 and abs(p.aperture - 0.001) < 1e-9
    assert p.frame_sample_id == 0 and p.temporal_sampling == 0
    assert p.frustum_culling == capi.CULL_AT_DIST and abs(p.frustum_dilation - 0.2) < 1e-7 and p.sh_degree == 3


def test_async_loader_queue_follows_the_reference_protocol(tmp_path):
    """mgs_loader_*: PlyLoaderAsync's READY -> LOADING -> LOADED / FAILURE states with the UI's scene-load queue on top:
    several files pushed at once load one at a time in order, a bad file fails without stopping the queue"""
    import time
    files = []
    for i, n in enumerate((300, 77, 1500)):
        sc = synth.make_scene(n, seed=40 + i)
        p = tmp_path / f"s{i}.ply"
        synth.write_ply(str(p), sc)
        files.append((str(p), sc))
    bad = tmp_path / "broken.ply"
    bad.write_bytes(b"ply\nformat ascii 1.0\nelement vertex 1\nproperty float x\nend_header\n0\n")
    L = mgs.Loader()
    assert L.status()[0] == capi.LOADER_READY
    with pytest.raises(mgs.MgsError) as e:
        L.take()
    assert e.value.code == -6  # MGS_ERR_STATE
    order = [files[0][0], str(bad), files[1][0], files[2][0]]
    for p in order:
        L.push(p)
    got = []
    for k, path in enumerate(order):
        t0 = time.time()
        while True:
            st, queued, head = L.status()
            assert head == path and queued == len(order) - 1 - k
            if st in (capi.LOADER_LOADED, capi.LOADER_FAILURE):
                break
            assert st == capi.LOADER_LOADING and time.time() - t0 < 60
            time.sleep(0.002)
        if path == str(bad):
            assert st == capi.LOADER_FAILURE
            with pytest.raises(mgs.MgsError) as e:
                L.take()
            assert e.value.code == -3 and "3DGS" in str(e.value)
        else:
            assert st == capi.LOADER_LOADED
            got.append(L.take().arrays())
    assert L.status()[:2] == (capi.LOADER_READY, 0)
    for a, (_, sc) in zip(got, files):
        for k in FIELDS:
            assert np.array_equal(a[k], sc[k].reshape(-1)), k
    L.push(files[0][0])  # destroy with a request in flight / a result not taken: no leak, no hang
    L.close()


def test_file_readers_survive_mutated_files_under_sanitizers():
    """tools/fuzz_loaders.sh, short run: csrc/host_model.cpp under ASan + UBSan over mutated .ply / .spz / .splat files —
    a reader returns an error or a consistent set, it never crashes, over-reads or allocates from an untrusted count"""
    import subprocess, shutil
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "fuzz_loaders.sh"), "120"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout + r.stderr)[-2000:]
    lines = [l for l in r.stdout.splitlines() if "mutated files" in l]
    assert len(lines) >= 9 and all("no crash / sanitizer report" in l for l in lines)


def test_camera_set_presets_and_project_camera_fields(tmp_path):
    """class CameraSet (src/camera_set.h:65-190): home preset, duplicate presets collapse (operator== ignores the DoF fields),
    the last preset cannot be erased; the .vkgs camera carries model / dofMode (old files: dofEnabled) / focusDist / aperture
    (vkgs_project_reader.cpp:569-584) and "pipeline" 4 selects 3DGUT (shaderio.h:61-66)"""
    import json
    from vk_gaussian_splatting_amd import cameras, project
    cs = cameras.CameraSet()
    a = cameras.Camera(eye=np.array([1, 2, 3], np.float32))
    b = cameras.Camera(eye=np.array([1, 2, 3], np.float32), dof_mode=1, aperture=0.05)   # same view, other lens
    c = cameras.Camera(eye=np.array([0, 2, 3], np.float32), model=1)
    cs.set_home_preset(a)
    assert len(cs) == 1 and cs.create_preset(b) == 0 and len(cs) == 1      # equal to the home preset: no duplicate
    assert cs.create_preset(c) == 1 and len(cs) == 2
    assert cs.load_preset(1) and cs.camera is c and not cs.load_preset(5)
    assert cs.store_current_camera() == 1
    assert cs.set_preset(0, c) and not cs.set_preset(9, c)
    assert cs.erase_preset(0) and len(cs) == 1 and not cs.erase_preset(0)   # the last one stays
    cs.set_home_preset(a)
    assert cs.get_preset(0) is a
    cs.reset()
    assert len(cs) == 0 and np.allclose(cs.camera.eye, [1.7, 1.5, 1.7])
    R = [[1, 0, 0], [0, 1, 0], [0, 0, 1]]
    items = [dict(id=i, img_name=f"{i}.jpg", width=8, height=8, position=[float(i % 2), 0.0, 1.0], rotation=R, fy=1.0, fx=1.0) for i in range(4)]
    (tmp_path / "cameras.json").write_text(json.dumps(items))
    assert cs.import_inria(str(tmp_path / "cameras.json")) == 4 and len(cs) == 2   # two distinct poses among the four

    doc = {"version": 5, "renderer": {"pipeline": 4, "sortingMethod": 3, "kernelDegree": 3, "kernelMinResponse": 0.02, "temporalSampling": True},
           "camera": {"model": 1, "eye": [2, 1, 2], "ctr": [0, 0, 0], "up": [0, 1, 0], "fov": 120.0, "clip": [0.1, 100.0],
                      "dofEnabled": True, "focusDist": 2.5, "aperture": 0.01},
           "cameras": [{"eye": [1, 1, 1], "dofEnabled": True, "dofMode": 0}]}
    (tmp_path / "gut.vkgs").write_text(json.dumps(doc))
    pr = project.load_project(str(tmp_path / "gut.vkgs"))
    assert pr.camera.model == 1 and pr.camera.dof_mode == 1 and pr.camera.focus_dist == 2.5 and pr.camera.aperture == 0.01
    assert pr.cameras[0].dof_mode == 0                                             # "dofMode" overrides the legacy key
    p = pr.frame_params(320, 200)
    assert (p.pipeline, p.camera_model, p.sort_mode, p.kernel_degree, p.dof_mode, p.temporal_sampling) == \
           (capi.PIPELINE_3DGUT, capi.CAMERA_FISHEYE, capi.SORT_STOCHASTIC, 3, capi.DOF_FIXED_FOCUS, 1)
    assert abs(p.kernel_min_response - 0.02) < 1e-7 and abs(p.focus_dist - 2.5) < 1e-7 and abs(p.aperture - 0.01) < 1e-7
    project.save_project(pr, str(tmp_path / "gut2.vkgs"))
    back = json.loads((tmp_path / "gut2.vkgs").read_text())["camera"]
    assert back["model"] == 1 and back["dofMode"] == 1 and back["focusDist"] == 2.5 and abs(back["aperture"] - 0.01) < 1e-9
    doc["renderer"]["pipeline"] = 1
    (tmp_path / "gs.vkgs").write_text(json.dumps(doc))
    p = project.load_project(str(tmp_path / "gs.vkgs")).frame_params(320, 200)
    assert p.pipeline == capi.PIPELINE_3DGS and p.dof_mode == capi.DOF_DISABLED


def test_compare_vkrepro_tool_reads_radiance_hdr(tmp_path):
    """tools/compare_vkrepro.py (what a maintainer runs on a Vulkan box against the reference's .hdr screenshot): its RGBE reader —
    flat and run-length encoded scanlines, as stb_image_write emits them — and its PSNR (image_compare_metric.comp.slang:116-130)"""
    import importlib.util
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("compare_vkrepro", os.path.join(root, "tools", "compare_vkrepro.py"))
    cv = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cv)
    want = np.maximum(np.load(os.path.join(root, "tests", "golden", "vkrepro", "msaa_3dgs", "expected_rgba16f.npy")).astype(np.float32), 0.0)
    flat = str(tmp_path / "flat.hdr")
    cv.write_hdr(flat, want)
    a = cv.read_hdr(flat)
    assert a.shape == want.shape[:2] + (3,) and cv.psnr_rgb(want, a)[0] > 50.0  # 8-bit mantissas
    # the same pixels, run-length encoded per channel the way stb does (runs of equal bytes >= 3, literal spans otherwise)
    raw = open(flat, "rb").read()
    head_end = raw.index(b"\n", raw.index(b"\n\n") + 2) + 1
    H, W = want.shape[:2]
    rgbe = np.frombuffer(raw, np.uint8, offset=head_end).reshape(H, W, 4)
    out = bytearray(raw[:head_end])
    for y in range(H):
        out += bytes([2, 2, W >> 8, W & 255])
        for ch in range(4):
            row, x = rgbe[y, :, ch], 0
            while x < W:
                r = 1
                while x + r < W and r < 127 and row[x + r] == row[x]:
                    r += 1
                if r >= 3:
                    out += bytes([128 + r, int(row[x])])
                    x += r
                else:
                    n = 1
                    while x + n < W and n < 128 and not (x + n + 2 < W and row[x + n] == row[x + n + 1] == row[x + n + 2]):
                        n += 1
                    out += bytes([n]) + row[x:x + n].tobytes()
                    x += n
    rle = str(tmp_path / "rle.hdr")
    open(rle, "wb").write(bytes(out))
    assert np.array_equal(cv.read_hdr(rle), a)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "compare_vkrepro.py"), "msaa_3dgs", rle], capture_output=True, text=True)
    assert r.returncode == 0 and "PASS" in r.stdout, r.stdout + r.stderr
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "compare_vkrepro.py"), "u8_storage", rle, "--flip-y"], capture_output=True, text=True)
    assert r.returncode == 1 and "FAIL" in r.stdout  # another case, upside down: far below the bar
