"""Debug-build experiment (csrc/k_osort.hip built with -DMGS_OS_TRACE): per-workgroup phase stamps of the key sort's passes on
the garden-sized frame.  Usage: MGS_GRAPH=0 MGS_OS_TRACE_FILE=/tmp/o.bin python tools/os_trace.py [pose ...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth

poses = [int(x) for x in sys.argv[1:]] or [0]
W, H = 1920, 1080
# round 6: OS_SPLATS / OS_INSTANCES / OS_STRIP="b e" select the workload (train-sized, configs[4], a strip of the multi-GPU partition)
NS, NI = int(os.environ.get("OS_SPLATS", "5830000")), int(os.environ.get("OS_INSTANCES", "1"))
STRIP = [int(x) for x in os.environ.get("OS_STRIP", "").split()] or None
sc = synth.make_scene(NS, seed=1)
scene = mgs.Scene(0)
ss = mgs.SplatSet.from_arrays(**sc)
for q in range(NI):
    if NI == 1:
        scene.add_instance(ss)
    else:
        cols = (NI + 1) // 2
        M = np.eye(4, dtype=np.float32)
        M[0, 3] = ((q % cols) - (cols - 1) / 2.0) * 12.0
        M[2, 3] = ((q // cols) - 0.5) * 12.0
        scene.add_instance(ss, M)
scene.commit()
print(f"workload: {NS} splats x {NI} instance(s), strip {STRIP}")
names = ["table + zeroing", "loads + digits", "ranking", "publish + level-1 issue + scans + level-1 consume + level-2 issue", "LDS re-order + level-2 consume / poll", "scatter stores (drained)"]
for pose in poses:
    eye = synth.orbit_pose(pose)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
    p = capi.default_params(W, H)
    capi.set_camera(p, V, P, eye)
    if STRIP:
        p.strip_row_begin, p.strip_row_end = STRIP
    for _ in range(4):
        scene.render(p, want_stats=True)
    raw = np.fromfile(os.environ["MGS_OS_TRACE_FILE"], np.uint64)
    maxp = int(raw[0])
    a = raw[2:2 + 4 * maxp * 8].reshape(4, maxp, 8)
    pr = raw[2 + 4 * maxp * 8:].reshape(-1, 8).astype(np.int64)
    pr = pr[pr[:, 0] > 0]
    if len(pr):
        b0 = pr[:, 0].min()
        ph = np.diff(pr[:, :3], axis=1) / 100.0
        print(f"--- pose {pose} k_os_prepare: {len(pr)} reduce workgroups; start spread {(pr[:, 0].max() - b0) / 100.0:.1f} us; all done at {(pr[:, 2].max() - b0) / 100.0:.1f} us")
        for i, n in enumerate(["slot rows -> chunk tables + total atomics", "records -> LDS table -> count-table atomics"]):
            print(f"   {n:48s} median {np.median(ph[:, i]):6.2f} us  max {ph[:, i].max():6.2f}")
        last = pr[pr[:, 7] == 1]
        if len(last):
            print(f"   the last arriver's fold of the count table: {(last[0, 5] - last[0, 4]) / 100.0:.2f} us; kernel body ends at {(last[0, 5] - b0) / 100.0:.1f} us")
    g = a[0].astype(np.int64)
    g = g[g[:, 6] > 0]
    if len(g):  # frames: pass 0 is virtual; its area holds the sub-stamps of pass 1's source table (k_os_pass<3>)
        ph = np.diff(g[:, :7], axis=1) / 100.0
        print(f"--- pose {pose} pass 1, source table of the virtual pass 0 ({len(g)} workgroups):")
        for i, nme in enumerate(["set-up + digit totals + scan", "chunk sums (first digit value)", "chunk scan + range", "run table loads", "expansion (first digit value)", "further digit values"]):
            v = ph[:, i][g[:, i + 1] > 0] if i < 5 else ph[:, i]
            print(f"   {nme:34s} median {np.median(v):6.2f} us  p90 {np.percentile(v, 90):6.2f}")
    for ps in range(4):
        b = a[ps]
        ran = (b[:, 6] > 0) & (b[:, 7] > 0)
        if not ran.any():
            print(f"--- pose {pose} pass {ps}: did not run")
            continue
        b = b[ran].astype(np.int64)
        base = b[:, 0].min()
        st = (b[:, 0] - base) / 100.0
        en = (b[:, 6] - base) / 100.0
        ph = np.diff(b[:, :7], axis=1) / 100.0
        cnt = b[:, 7] >> 32
        spins = b[:, 7] & 0xFFFFFFFF
        print(f"--- pose {pose} pass {ps}: {int(ran.sum())} workgroups; span {en.max():.1f} us; sum of durations {(en - st).sum():.0f} us "
              f"(= {(en - st).sum() / en.max():.0f} resident on average); keys/workgroup median {np.median(cnt):.0f}; spins median {np.median(spins):.0f} max {spins.max()}")
        print("   duration us 10/50/90/max:", np.percentile(en - st, [10, 50, 90, 100]).round(1), " start us 50/90/max:", np.percentile(st, [50, 90, 100]).round(1))
        for i, n in enumerate(names):
            print(f"   {n:34s} median {np.median(ph[:, i]):6.2f} us  p90 {np.percentile(ph[:, i], 90):6.2f}  total {ph[:, i].sum():8.0f} workgroup-us")
