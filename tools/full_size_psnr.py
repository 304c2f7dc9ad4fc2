"""One-off measurement: PSNR of the HIP frame vs the CPU oracle at the FULL benchmark configuration
(syn_garden, 5.83 M splats, SH degree 3, fp32 storage, 1920x1080).  The oracle is single-threaded: minutes."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth
from oracle import binding as ob
N = int(sys.argv[1]) if len(sys.argv) > 1 else 5_830_000
poses = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [0]
W, H = 1920, 1080
sc = synth.make_scene(N, seed=0xC0FFEE + 2)
ss = mgs.SplatSet.from_arrays(**sc); scene = mgs.Scene(0); scene.add_instance(ss); scene.commit()
perm = scene.storage_order(0, N)
ps_p = ob.PreparedSet({k: v[perm] for k, v in sc.items()})
ps = ob.PreparedSet(sc)
inst_p, inst = ob.make_instances([(ps_p, None)]), ob.make_instances([(ps, None)])
res = []
for pose in poses:
    eye = synth.orbit_pose(pose)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
    p = capi.default_params(W, H); capi.set_camera(p, V, P, eye)
    so = scene.sort_keys(p); gk, gi = scene.sort_download(so.count)
    ok, oi = ob.key_cull(ob.make_frame(V, P, eye, W, H), inst_p)
    oks, ois = ob.sort_stable(ok, oi)
    keys_equal, ids_equal = bool(np.array_equal(gk, oks)), bool(np.array_equal(gi, perm[ois]))
    out = scene.render(p, want_stats=True)
    img = scene.download_frame(p).astype(np.float32)
    t = time.time()
    oimg, st = ob.render(ob.make_frame(V, P, eye, W, H, target_fp16=1), inst, order=perm[ois])
    r = dict(pose=pose, n=N, sorted_keys_bit_exact=keys_equal, sorted_ids_bit_exact=ids_equal, visible=int(so.count),
             psnr_db=ob.psnr_rgb(img, oimg), max_abs_rgb=float(np.abs(img[..., :3] - oimg[..., :3]).max()),
             oracle_fragments=st["fragments"], oracle_seconds=time.time() - t)
    print(json.dumps(r), flush=True)
    res.append(r)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/full_size_psnr.json", "w"), indent=1)
