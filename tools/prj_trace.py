"""Debug-build experiment (csrc/k_project.hip built with -DMGS_PRJ_TRACE): per-workgroup phase stamps of k_project on the
garden-sized frame.  Usage: MGS_GRAPH=0 MGS_PRJ_TRACE_FILE=/tmp/p.bin python tools/prj_trace.py [pose ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth

poses = [int(x) for x in sys.argv[1:]] or [0]
W, H = 1920, 1080
sc = synth.make_scene(5_830_000, seed=1)
scene = mgs.Scene(0)
scene.add_instance(mgs.SplatSet.from_arrays(**sc))
scene.commit()
for pose in poses:
    eye = synth.orbit_pose(pose)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
    p = capi.default_params(W, H)
    capi.set_camera(p, V, P, eye)
    if os.environ.get("STRIP"):  # tile rows "begin end": one device's strip of a multi-GPU frame
        p.strip_row_begin, p.strip_row_end = (int(x) for x in os.environ["STRIP"].split())
    for _ in range(4):
        scene.render(p, want_stats=True)
    a = np.fromfile(os.environ["MGS_PRJ_TRACE_FILE"], np.uint64).reshape(-1, 8)
    ran = a[:, 5] > 0
    b = a[ran].astype(np.int64)
    base = b[:, 0].min()
    st = (b[:, 0] - base) / 100.0
    en = (b[:, 5] - base) / 100.0
    ph = np.diff(b[:, :6], axis=1) / 100.0  # load, key, scan+compact, phase 2, tail
    print(f"--- pose {pose}: {int(ran.sum())} of {len(a)} workgroups did work; span {en.max():.1f} us; sum of durations {(en - st).sum():.0f} us "
          f"(= {(en - st).sum() / en.max():.0f} resident on average)")
    print("duration us 10/50/90/max:", np.percentile(en - st, [10, 50, 90, 100]).round(1), " start us 50/90/max:", np.percentile(st, [50, 90, 100]).round(1))
    names = ["centre loads", "keys + cull", "scan + compaction", "phase 2 (projection, records)", "second compaction + slot stores + marks"]
    for i, n in enumerate(names):
        print(f"  {n:42s} median {np.median(ph[:, i]):6.2f} us   total {ph[:, i].sum():8.0f} workgroup-us")
    print("  survivors per workgroup (frustum / sorted) median:", np.median(b[:, 6]), np.median(b[:, 7]))
    M, oc, dur = b[:, 6], b[:, 7], en - st
    print(f"  candidates M per workgroup p10/50/90/max {np.percentile(M, [10, 50, 90, 100])}; sorted per workgroup p10/50/90/max {np.percentile(oc, [10, 50, 90, 100])}")
    print(f"  workgroups with M == 0: {(M == 0).sum()} ({(M == 0).mean():.2%}); with no sorted pair: {(oc == 0).sum()} ({(oc == 0).mean():.2%}); "
          f"M <= 256: {(M <= 256).mean():.2%}; sum M {M.sum()} sum sorted {oc.sum()}")
    for lo, hi in ((0, 1), (1, 65), (65, 257), (257, 1025), (1025, 4096)):
        sel = (M >= lo) & (M < hi)
        if sel.any():
            print(f"    M in [{lo}, {hi}): {sel.sum():5d} workgroups, duration median {np.median(dur[sel]):6.1f} us, phases " + " ".join(f"{np.median(ph[sel, i]):5.1f}" for i in range(5)))
    ts = np.linspace(0, en.max(), 25)
    print("  resident over time:", [int(((st <= x) & (en > x)).sum()) for x in ts])
