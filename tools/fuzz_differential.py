"""Extended run of tests/test_gpu_parity.py::test_randomized_differential_vs_oracle over seeds outside the suite's
0..47 (bit-exact sorted stream, PSNR vs the oracle, strips == full frame).  Usage: fuzz_differential.py FIRST COUNT"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import binding as ob
import test_gpu_parity as T
ob.build()
first, count = int(sys.argv[1]) if len(sys.argv) > 1 else 1000, int(sys.argv[2]) if len(sys.argv) > 2 else 200
bad, t0 = [], time.time()
for seed in (range(first, first + count) if not (len(sys.argv) > 3 and sys.argv[3] == "extra") else []):
    try:
        T.test_randomized_differential_vs_oracle(ob, seed)
    except Exception as e:  # noqa: BLE001
        bad.append(seed)
        print("FAIL seed", seed, type(e).__name__, str(e)[:300], flush=True)
        traceback.print_exc(limit=1)
print(f"{count} seeds from {first}: {len(bad)} failures {bad} in {time.time() - t0:.0f} s")


def extra(seed):
    """storage formats + surface outputs on a random configuration: frame vs the oracle fed the same quantisation,
    picked ids, integrated normals"""
    import numpy as np
    import vk_gaussian_splatting_amd as mgs
    from vk_gaussian_splatting_amd import capi, synth
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.integers(200, 8000))
    sc = synth.make_scene(n, seed=900 + seed)
    sc["scale"] += np.float32(rng.uniform(-0.5, 1.0))
    shf, rgf = int(rng.integers(0, 3)), int(rng.integers(0, 3))
    M = None
    if rng.integers(0, 2):
        M, _ = mgs.compute_transform(rng.uniform(0.5, 1.5, 3), rng.uniform(-180, 180, 3), rng.uniform(-1, 1, 3))
    scene = mgs.Scene(0)
    scene.add_instance(mgs.SplatSet.from_arrays(**sc), M)
    scene.commit(shf, rgf)
    W, H = int(rng.integers(64, 640)), int(rng.integers(48, 400))
    dist = float(rng.choice([0.5, 2.0, 5.0]))
    th = rng.uniform(0, 2 * np.pi)
    eye = np.array([dist * np.cos(th), rng.uniform(-1, 1), dist * np.sin(th)], np.float32)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], float(rng.uniform(30, 90)), 0.1, 2000.0, W, H)
    p = capi.default_params(W, H)
    capi.set_camera(p, V, P, eye)
    p.surface_outputs = 1
    p.quantize_normals = int(rng.integers(0, 2))
    p.depth_iso_threshold = float(rng.uniform(0.2, 0.9))
    out = scene.render(p, want_stats=True)
    assert out.error_flags == 0
    img = scene.download_frame(p).astype(np.float32)
    depth, ids, nrm = scene.download_surface(p, normals=True)
    fk = dict(view=V, proj=P, camera_pos=eye, width=W, height=H)
    oks, ois = T.oracle_sorted_stream(ob, scene, sc, fk, transforms=(M,))
    inst = ob.make_instances([(ob.PreparedSet(sc, shf, rgf), M)])
    oimg, _ = ob.render(ob.make_frame(target_fp16=1, **fk), inst, order=ois)
    psnr = ob.psnr_rgb(img, oimg)
    assert psnr >= 50.0, ("psnr", psnr, shf, rgf)
    od, oi, on = ob.render_surface(ob.make_frame(**fk), inst, ois[::-1].copy(), p.depth_iso_threshold,
                                   quantize_normals=bool(p.quantize_normals), normals=True)
    same = (ids == oi)
    assert same.mean() >= 0.99, ("ids", same.mean())
    err = np.abs(nrm - on)
    assert err.mean() < 5e-5 and np.quantile(err, 0.999) < 5e-3, ("normal", err.mean(), err.max())
    scene.close()


if len(sys.argv) > 3 and sys.argv[3] == "extra":
    bad = []
    for seed in range(first, first + count):
        try:
            extra(seed)
        except Exception as e:  # noqa: BLE001
            bad.append(seed)
            print("FAIL extra seed", seed, type(e).__name__, str(e)[:300], flush=True)
    print(f"extra: {count} seeds from {first}: {len(bad)} failures {bad}")
