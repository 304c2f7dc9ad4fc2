"""Debug-build experiment (csrc/k_raster.hip built with -DMGS_DB_TRACE): per-workgroup phase stamps of k_dbin_emit on the
garden-sized frame.  Usage: MGS_GRAPH=0 MGS_DB_TRACE_FILE=/tmp/d.bin python tools/db_trace.py [pose ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth
poses = [int(x) for x in sys.argv[1:]] or [0]
W, H = 1920, 1080
sc = synth.make_scene(5_830_000, seed=1)
scene = mgs.Scene(0); scene.add_instance(mgs.SplatSet.from_arrays(**sc)); scene.commit()
for pose in poses:
    eye = synth.orbit_pose(pose)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
    p = capi.default_params(W, H); capi.set_camera(p, V, P, eye)
    for _ in range(4):
        scene.render(p, want_stats=True)
    a = np.fromfile(os.environ["MGS_DB_TRACE_FILE"], np.uint64).reshape(-1, 8)
    b = a[a[:, 4] > 0].astype(np.int64)
    base = b[:, 0].min()
    st, en = (b[:, 0] - base) / 100.0, (b[:, 4] - base) / 100.0
    ph = np.diff(b[:, :5], axis=1) / 100.0
    print(f"--- pose {pose} k_dbin_emit: {len(b)} workgroups; span {en.max():.1f} us; sum of durations {(en - st).sum():.0f} us (= {(en - st).sum() / en.max():.0f} resident on average)")
    print("duration us 10/50/90/max:", np.percentile(en - st, [10, 50, 90, 100]).round(1), " start us 50/90/max:", np.percentile(st, [50, 90, 100]).round(1))
    for i, n in enumerate(["ids + masks -> LDS, per-bin counts", "four block scans, bases", "bit walk into the LDS stage", "copy-out (drained)"]):
        print(f"  {n:42s} median {np.median(ph[:, i]):6.2f} us  p90 {np.percentile(ph[:, i], 90):6.2f}  total {ph[:, i].sum():8.0f} workgroup-us")
    print("  list entries per workgroup median / max:", np.median(b[:, 5]), b[:, 5].max())
