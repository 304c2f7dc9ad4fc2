"""child process of test_gpu_parity.test_key_sort_oversubscribed_by_a_co_running_kernel: the key sort (frame path and stand-alone
path) while tests/helpers/libcuhog.so holds LDS on every compute unit from a second stream, so that the passes' workgroups get
their slots in instalments.  Every sort must be exactly the undisturbed one — or fail loudly with kErrSpinTimeout — and never hang
(the parent runs this under a timeout)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
import vk_gaussian_splatting_amd as mgs  # noqa: E402
from vk_gaussian_splatting_amd import capi, synth  # noqa: E402

hog = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "helpers", "libcuhog.so"))
hog.cu_hog_launch.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
hog.cu_hog_launch.restype = C.c_int
N, W, H = 3_000_000, 1280, 720
sc = synth.make_scene(N, seed=77)
scene = mgs.Scene(0)
scene.add_instance(mgs.SplatSet.from_arrays(**sc))
scene.commit()
side = torch.cuda.Stream()
eye = synth.orbit_pose(11)
V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
p = capi.default_params(W, H)
capi.set_camera(p, V, P, eye)
so = scene.sort_keys(p)
k0, i0 = scene.sort_download(so.count)
scene.render(p)
f0 = scene.download_frame(p).copy()
assert so.count > 1_000_000, so.count  # hundreds of sort partitions
rng = np.random.default_rng(5)
rk = rng.integers(0, 2**32, 2_000_003, dtype=np.uint32)
rv = np.arange(rk.size, dtype=np.uint32)
ro = np.argsort(rk, kind="stable")
timeouts = ok = 0
# (blocks, threads, LDS bytes per workgroup, base us, step us): from "two hogs per CU leave room for one sort workgroup" to
# "every CU is full until its hogs retire, 16 different retirement times"
for rep, (blocks, threads, lds, base, step) in enumerate([(512, 256, 60 * 1024, 40, 12), (256, 256, 120 * 1024, 60, 20), (1024, 64, 36 * 1024, 30, 8),
                                                          (768, 256, 48 * 1024, 20, 25), (256, 1024, 150 * 1024, 80, 15), (2048, 64, 30 * 1024, 10, 5)] * 3):
    rc = hog.cu_hog_launch(C.c_void_p(side.cuda_stream), blocks, threads, lds, base, step)
    assert rc == 0, rc
    try:
        if rep % 3 == 0:
            so = scene.sort_keys(p)
            k, i = scene.sort_download(so.count)
            assert so.count == k0.size and np.array_equal(k, k0) and np.array_equal(i, i0), "sorted stream differs under oversubscription"
        elif rep % 3 == 1:
            out = scene.render(p, want_stats=True)
            assert out.error_flags == 0
            assert np.array_equal(scene.download_frame(p), f0), "frame differs under oversubscription"
        else:
            ks, vs, _ = scene.radix_sort_host(rk, rv)
            assert np.array_equal(ks, rk[ro]) and np.array_equal(vs, rv[ro]), "stand-alone sort differs under oversubscription"
        ok += 1
    except mgs.MgsError as e:
        assert "kErrSpinTimeout" in str(e), str(e)  # the only acceptable failure: reported, not silent
        timeouts += 1
    torch.cuda.synchronize()
print(f"HOG_OK sorts/frames correct {ok}, reported spin timeouts {timeouts}")
scene.close()
