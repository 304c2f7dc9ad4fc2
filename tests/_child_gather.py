"""child process of test_gpu_parity.test_render_gathered_two_ranks_*: ONE rank of a multi-rank mgs_render_gathered job whose ranks
share one GPU.  RCCL is the test double tests/helpers/libfakerccl.so (MGS_RCCL_LIB, set by the test); no torch.distributed — the
128-byte unique id travels through a file.   usage: _child_gather.py RANK WORLD IDFILE MODE [BOUNDS]
MODE ok:    every rank renders its strip, the exchange reassembles the frame, each rank compares it with its own full-frame render
MODE abort: the LAST rank asks for a frame no buffer can be had for (width 0): it aborts its communicator (mgs_render_gathered's
            last resort) and the other ranks' collective must FAIL (MGS_ERR_DEVICE), not hang"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import vk_gaussian_splatting_amd as mgs  # noqa: E402
from vk_gaussian_splatting_amd import capi, synth  # noqa: E402

rank, world, idfile, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
bounds = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else None
W, H = 1280, 720
sc = synth.make_scene(120_000, seed=5)
scene = mgs.Scene(0)
scene.add_instance(mgs.SplatSet.from_arrays(**sc))
scene.commit()
eye = synth.orbit_pose(9)
V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
p = capi.default_params(W, H)
capi.set_camera(p, V, P, eye)
scene.render(p)
full = scene.download_frame(p).view(np.uint16).copy()
if rank == 0:
    uid = capi.comm_unique_id()
    with open(idfile + ".tmp", "wb") as f:
        f.write(uid)
    os.replace(idfile + ".tmp", idfile)
else:
    t0 = time.time()
    while not os.path.exists(idfile):
        if time.time() - t0 > 120:
            raise SystemExit("no unique id from rank 0")
        time.sleep(0.01)
    uid = open(idfile, "rb").read()
scene.comm_init(rank, world, uid)
print("RCCL_MAPPED", sorted({ln.split()[-1].rsplit("/", 1)[-1] for ln in open("/proc/self/maps") if "rccl" in ln.lower() and ".so" in ln}), flush=True)
if bounds:
    scene.set_strip_rows(bounds)
if mode == "ok":
    for rep in range(3):  # the same communicator over several frames (the double's segment is reused)
        # poison the frame buffer's other rows first: what the exchange must bring in
        scene.render(p) if rep == 1 else None
        scene.render_gathered(p)
        got = scene.download_frame(p).view(np.uint16)
        print(f"GATHERED_EQUALS_FULL rank {rank} rep {rep}:", bool(np.array_equal(got, full)), flush=True)
    scene.comm_destroy()
elif mode == "abort":
    if rank == world - 1:
        q = capi.default_params(W, H)
        capi.set_camera(q, V, P, eye)
        q.width = 0
        try:
            scene.render_gathered(q)
            print("ABORT_RANK no error?!", flush=True)
        except mgs.MgsError as e:
            print("ABORT_RANK raised:", str(e)[:160], flush=True)
    else:
        t0 = time.time()
        try:
            scene.render_gathered(p)
            print("PEER no error?!", flush=True)
        except mgs.MgsError as e:
            print(f"PEER raised after {time.time() - t0:.1f} s:", str(e)[:200], flush=True)
scene.close()
print("CHILD_DONE", rank, flush=True)
