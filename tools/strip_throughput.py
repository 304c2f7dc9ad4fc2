"""frames/s of each strip of the G-way tile-row partition rendered with K frames in flight on ONE GPU (what each of G GPUs
sustains; the job runs at the slowest strip's rate, before the exchange):  python tools/strip_throughput.py [W H [G [K]]]"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth, multigpu
N = 5_830_000
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
G = int(sys.argv[3]) if len(sys.argv) > 3 else 8
K = int(sys.argv[4]) if len(sys.argv) > 4 else 3
sc = synth.make_scene(N, seed=0xC0FFEE + 2)
ss = mgs.SplatSet.from_arrays(**sc)
scenes, streams = [], []
base = mgs.Scene(0); base.add_instance(ss); base.commit()   # one committed scene, K frame contexts over it
for c in range(K):
    s = base if c == 0 else base.frame_context()
    st = torch.cuda.Stream(); s.set_stream(st.cuda_stream)
    scenes.append(s); streams.append(st)
poses = []
for i in range(64):
    eye = synth.orbit_pose(i)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
    p = capi.default_params(W, H); capi.set_camera(p, V, P, eye); poses.append(p)
rows = multigpu.tile_rows(H)
cost = np.zeros(rows)
for i in range(0, 64, 8):
    scenes[0].render(poses[i]); cost += scenes[0].row_costs(H)
tables = {"full frame": [0, rows], "equal": [multigpu.strip_rows(H, G, r)[0] for r in range(G)] + [rows], "balanced": multigpu.balanced_bounds(cost, G)}


def rate(b, e, steps=192):
    for p in poses:
        p.strip_row_begin, p.strip_row_end = b, e
    for i in range(24):
        with torch.cuda.stream(streams[i % K]): scenes[i % K].render(poses[i % 64])
    torch.cuda.synchronize(); t = time.perf_counter()
    for i in range(steps):
        with torch.cuda.stream(streams[i % K]): scenes[i % K].render(poses[i % 64])
    torch.cuda.synchronize()
    return steps / (time.perf_counter() - t)


res = {"resolution": [W, H], "gpus": G, "frames_in_flight": K}
full = None
for name, b in tables.items():
    r = [rate(b[i], b[i + 1]) if b[i + 1] > b[i] else float("inf") for i in range(len(b) - 1)]
    if name == "full frame":
        full = r[0]
    print(f"{name:10s} bounds {b} frames/s per strip {[round(x) for x in r]} -> job rate {min(r):.0f} fps = {min(r) / full:.2f}x of one GPU")
    res[name] = dict(bounds=b, fps_per_strip=r, job_fps=min(r), speedup=min(r) / full)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open(f"gpurun_out/strip_throughput_{W}x{H}_g{G}_k{K}.json", "w"), indent=1)
