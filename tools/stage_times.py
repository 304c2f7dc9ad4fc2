"""Per-stage HIP-event times of serial frames (one frame in flight, plain launches with the six stage events) for one workload:
   python tools/stage_times.py [--splats N] [--instances K] [--width W --height H] [--strip B E] [--scene garden|fog] [--frames F] [--tag T]
Prints ONE line: tag, mean ms per stage over the orbit's first F poses (after 4 warm-up frames), sorted count.  Env knobs of
libmgs.so are read once per process: run it once per variant (tools/r6_ab.sh)."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth

ap = argparse.ArgumentParser()
ap.add_argument("--splats", type=int, default=5_830_000)
ap.add_argument("--instances", type=int, default=1)
ap.add_argument("--width", type=int, default=1920)
ap.add_argument("--height", type=int, default=1080)
ap.add_argument("--strip", type=int, nargs=2, default=None)
ap.add_argument("--scene", default="garden")
ap.add_argument("--frames", type=int, default=32)
ap.add_argument("--skip", type=int, default=4, help="untimed frames in front (the adaptive bin size settles after 24)")
ap.add_argument("--tag", default="")
ap.add_argument("--graph", action="store_true", help="also report frames/s of graph-replayed frames (no stage events)")
a = ap.parse_args()
sc = synth.make_scene(a.splats, seed=0xC0FFEE + 2)
if a.scene == "fog":
    sc["opacity"] = (sc["opacity"] - 2.5).astype(np.float32)
elif a.scene == "sparse":
    r = np.linalg.norm(sc["positions"], axis=1)
    keep = (r >= 4.0) | (np.random.default_rng(7).random(a.splats) < 0.10)
    sc = {k: np.ascontiguousarray(v[keep]) for k, v in sc.items()}
ss = mgs.SplatSet.from_arrays(**sc)
scene = mgs.Scene(0)
for q in range(a.instances):
    if a.instances == 1:
        scene.add_instance(ss)
    else:
        cols = (a.instances + 1) // 2
        M = np.eye(4, dtype=np.float32)
        M[0, 3] = ((q % cols) - (cols - 1) / 2.0) * 12.0
        M[2, 3] = ((q // cols) - 0.5) * 12.0
        scene.add_instance(ss, M)
scene.commit()
W, H = a.width, a.height
poses = []
for i in range(64):
    eye = synth.orbit_pose(i)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
    p = capi.default_params(W, H)
    capi.set_camera(p, V, P, eye)
    if a.strip:
        p.strip_row_begin, p.strip_row_end = a.strip
    poses.append(p)
rows = []
for i in range(a.skip + a.frames):
    p = poses[i % 64]
    p.collect_timings = 2
    scene.render(p)
    scene.sync()
    if i >= a.skip:
        rows.append(scene.timings_all(0))
ms = np.array(rows, np.float64).mean(axis=0)
o = scene.render(poses[0], want_stats=True)
names = ["project", "sort", "bin", "pairsort", "composite", "total"]
line = f"{a.tag:24s} " + " ".join(f"{n} {ms[j] * 1000:7.1f}" for j, n in enumerate(names)) + f"  V {o.sorted_count} D {o.tile_pairs} err {o.error_flags}"
if a.graph:
    import time
    for p in poses:
        p.collect_timings = 0
    for i in range(16):
        scene.render(poses[i % 64])
    scene.sync()
    t0 = time.perf_counter()
    for i in range(128):
        scene.render(poses[i % 64])
    scene.sync()
    line += f"  graph-serial {128 / (time.perf_counter() - t0):.0f} fps"
print(line, flush=True)
scene.close()
