"""soak: scene create/commit/render/destroy cycles (leak check via hipMemGetInfo through torch) + a long replay"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth
W, H = 1280, 720
def cam(i):
    eye = synth.orbit_pose(i)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
    p = capi.default_params(W, H); capi.set_camera(p, V, P, eye); return p
free0 = None
for cyc in range(12):
    sc = synth.make_scene(400_000 + 50_000 * (cyc % 3), seed=cyc)
    scene = mgs.Scene(0); scene.add_instance(mgs.SplatSet.from_arrays(**sc)); scene.commit(cyc % 3, cyc % 3)
    for i in range(40):
        p = cam(i)
        if i % 7 == 0: p.strip_row_begin, p.strip_row_end = 5, 30
        if i % 11 == 0: p.surface_outputs = 1
        elif i % 13 == 0: p.pipeline = capi.PIPELINE_3DGUT          # its record buffer is allocated on first use
        if i % 17 == 0 and i % 11: p.camera_model = capi.CAMERA_FISHEYE; p.pipeline = capi.PIPELINE_3DGUT
        if i % 9 == 4: p.sort_mode = capi.SORT_STOCHASTIC; p.frame_sample_id = i; p.temporal_sampling = i % 2
        if i % 19 == 3: p.pipeline = capi.PIPELINE_3DGUT; p.dof_mode = capi.DOF_FIXED_FOCUS; p.aperture = 0.02; p.frame_sample_id = i
        if i % 23 == 5: p.pipeline = capi.PIPELINE_3DGUT; p.kernel_degree = 4; p.surface_outputs = 1
        o = scene.render(p, want_stats=(i % 5 == 0))
    img = scene.download_frame(p); assert np.isfinite(img.astype(np.float32)).all()
    scene.close()
    torch.cuda.synchronize()
    free, total = torch.cuda.mem_get_info()
    if cyc == 1: free0 = free
    print(f"cycle {cyc}: free {free/2**30:.2f} GiB")
assert free0 is not None and abs(free - free0) < 256 * 2**20, "device memory drifts across scene lifetimes"
sc = synth.make_scene(1_000_000, seed=99)
scene = mgs.Scene(0); scene.add_instance(mgs.SplatSet.from_arrays(**sc)); scene.commit()
t0 = time.time(); n = 0
ref = None
while time.time() - t0 < float(os.environ.get('SOAK_SECONDS', '20')):
    for i in range(64):
        scene.render(cam(i)); n += 1
    img = scene.download_frame(cam(63))
    if ref is None: ref = img.copy()
    assert np.array_equal(img.view(np.uint16), ref.view(np.uint16)), "frame changed across replays"
print(f"replayed {n} frames in {time.time()-t0:.1f} s, last frame identical every lap")
