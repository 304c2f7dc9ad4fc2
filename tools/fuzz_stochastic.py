"""Randomised differential run of the stochastic paths (tests/test_gpu_stochastic.py over random configurations):
stochastic splats in both pipelines (fraction of pixels identical to the oracle's) and 3DGUT depth of field (PSNR).
Usage: fuzz_stochastic.py FIRST COUNT"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from oracle import binding as ob
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth
ob.build()


def one(seed):
    rng = np.random.default_rng(31000 + seed)
    n = int(rng.integers(300, 12000))
    sc = synth.make_scene(n, seed=500 + seed)
    sc["scale"] += np.float32(rng.uniform(-0.5, 0.8))
    M = None
    if rng.integers(0, 2):
        M, _ = mgs.compute_transform(rng.uniform(0.6, 1.4, 3), rng.uniform(-180, 180, 3), rng.uniform(-1, 1, 3))
    scene = mgs.Scene(0)
    scene.add_instance(mgs.SplatSet.from_arrays(**sc), M)
    scene.commit()
    perm = scene.storage_order(0, n)
    sc_p = {k: (v[perm] if v is not None else None) for k, v in sc.items()}
    W, H = int(rng.integers(64, 520)), int(rng.integers(48, 360))
    dist = float(rng.choice([1.0, 2.5, 5.0]))
    th = rng.uniform(0, 2 * np.pi)
    eye = np.array([dist * np.cos(th), rng.uniform(-1, 1), dist * np.sin(th)], np.float32)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], float(rng.uniform(35, 80)), 0.1, 2000.0, W, H)
    gut = bool(rng.integers(0, 2))
    sample = int(rng.integers(0, 200))
    mode = rng.choice(["stoch", "dof", "both", "plain"]) if gut else "stoch"   # plain: the 3DGUT frame itself (packed compositor)
    p = capi.default_params(W, H)
    capi.set_camera(p, V, P, eye)
    p.pipeline = capi.PIPELINE_3DGUT if gut else capi.PIPELINE_3DGS
    p.target_format = capi.TARGET_RGBA32F
    p.frame_sample_id = sample
    fkw = dict(frame_sample_id=sample)
    if gut and mode == "plain" and rng.integers(0, 2):
        p.ms_antialiasing = 1
        fkw["ms_antialiasing"] = 1
    if gut:
        p.camera_model = int(rng.integers(0, 2))
        p.extent_method = int(rng.integers(0, 2))
        fkw.update(camera_model=p.camera_model, extent_method=p.extent_method)
    if mode in ("stoch", "both"):
        p.sort_mode = capi.SORT_STOCHASTIC
        fkw["stochastic"] = 1
    if mode in ("dof", "both"):
        p.dof_mode, p.focus_dist, p.aperture = capi.DOF_FIXED_FOCUS, float(rng.uniform(0.5, 5.0)), float(rng.uniform(0.0, 0.05))
        fkw.update(dof_mode=1, focus_dist=p.focus_dist, aperture=p.aperture)
    out = scene.render(p, want_stats=True)
    assert out.error_flags == 0
    img = scene.download_frame(p).astype(np.float32)
    inst = ob.make_instances([(ob.PreparedSet(sc_p), M)])
    ok, oi = ob.key_cull(ob.make_frame(V, P, eye, W, H, camera_model=fkw.get("camera_model", 0), pipeline_3dgut=int(gut)), inst)
    _, ois = ob.sort_stable(ok, oi)
    oimg, st = (ob.render_gut if gut else ob.render)(ob.make_frame(V, P, eye, W, H, **fkw), inst, ois)
    if "stochastic" in fkw:
        same = float(np.all(np.abs(img[..., :3] - oimg[..., :3]) <= 2e-3, axis=-1).mean())
        assert same >= 0.99, ("identical pixels", same, gut, mode, n, W, H)
        res = f"same {same:.5f}"
    else:
        psnr = ob.psnr_rgb(img, oimg)
        assert psnr >= 50.0, ("psnr", psnr, n, W, H)
        res = f"psnr {psnr:.1f}"
    # strips == full frame
    rows = (H + 15) // 16
    if rows >= 2:
        b = int(rng.integers(0, rows - 1)); e = int(rng.integers(b + 1, rows + 1))
        p.strip_row_begin, p.strip_row_end = b, e
        scene.render(p)
        part = scene.download_frame(p).astype(np.float32)
        assert np.array_equal(part[b * 16:min(e * 16, H)], img[b * 16:min(e * 16, H)]), ("strip", b, e)
    scene.close()
    return f"{'3DGUT' if gut else '3DGS'} {mode} n={n} {W}x{H} {res}"


first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 0, int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad, t0 = [], time.time()
for seed in range(first, first + count):
    try:
        r = one(seed)
        if seed % 20 == 0:
            print("seed", seed, r, flush=True)
    except Exception as e:  # noqa: BLE001
        bad.append(seed)
        print("FAIL seed", seed, type(e).__name__, str(e)[:300], flush=True)
        traceback.print_exc(limit=1)
print(f"{count} seeds from {first}: {len(bad)} failures {bad} in {time.time() - t0:.0f} s")
