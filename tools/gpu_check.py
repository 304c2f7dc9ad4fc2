"""Developer scratch check run on the GPU box: HIP path vs oracle on small inputs."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth
from oracle import binding as ob

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
    W, H = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (640, 480)
    sc = synth.make_scene(n, seed=7)
    scene = mgs.Scene(0)
    # 1) raw radix sort
    rng = np.random.default_rng(0)
    for cnt in [1, 5, 2047, 2048, 2049, 100000, 1 << 20]:
        k = rng.integers(0, 2**32, cnt, dtype=np.uint32); v = np.arange(cnt, dtype=np.uint32)
        ks, vs, ms = scene.radix_sort_host(k, v)
        order = np.argsort(k, kind="stable")
        ok = np.array_equal(ks, k[order]) and np.array_equal(vs, v[order].astype(np.uint32))
        print("radix", cnt, "ok" if ok else "MISMATCH", f"{ms:.3f} ms")
    k = (rng.integers(0, 5000, 300000, dtype=np.uint32)); v = np.arange(k.size, dtype=np.uint32)
    ks, vs, ms = scene.radix_sort_host(k, v, 0, 16)
    order = np.argsort(k, kind="stable"); print("radix16", np.array_equal(ks, k[order]) and np.array_equal(vs, v[order]))
    # 2) scene
    ss = mgs.SplatSet.from_arrays(**sc)
    scene.add_instance(ss)
    scene.commit()
    eye = synth.orbit_pose(3)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
    p = capi.default_params(W, H); capi.set_camera(p, V, P, eye)
    ps = ob.PreparedSet(sc); inst = ob.make_instances([(ps, None)])
    fr = ob.make_frame(V, P, eye, W, H)
    ok, oi = ob.key_cull(fr, inst); oks, ois = ob.sort_stable(ok, oi)
    so = scene.sort_keys(p); gk, gi = scene.sort_download(so.count)
    print("sort_keys count", so.count, "oracle", ok.size, "keys eq", np.array_equal(gk, oks), "ids eq", np.array_equal(gi, ois),
          f"key {so.key_ms:.3f} ms sort {so.sort_ms:.3f} ms passes {so.passes}")
    p.collect_timings = 1
    out = scene.render(p)
    img = scene.download_frame(p).astype(np.float32)
    t = time.time(); oimg, st = ob.render(ob.make_frame(V, P, eye, W, H, target_fp16=1), inst); t = time.time() - t
    print("frame: frustum", out.frustum_count, "sorted", out.sorted_count, "pairs", out.tile_pairs, "err", out.error_flags,
          "stages", [f"{x:.3f}" for x in out.stage_ms[:6]])
    print("oracle", st, f"{t:.2f}s")
    print("PSNR vs oracle(fp16 BTF):", ob.psnr_rgb(img, oimg), "max abs rgb", np.abs(img[..., :3] - oimg[..., :3]).max())
    np.save("gpurun_out/img_gpu.npy", img.astype(np.float16)); np.save("gpurun_out/img_orc.npy", oimg.astype(np.float16))

if __name__ == "__main__":
    os.makedirs("gpurun_out", exist_ok=True)
    main()
