"""A/B harness: render a few poses (garden-sized by default), print stage timings and save the frames.
usage: python tools/ab_frames.py <tag> [N] [W H]    (environment knobs such as MGS_DIRECT_BIN are read by libmgs)
A second run with another tag compares its frames bit for bit with every earlier tag found in gpurun_out/ab_*.npz"""
import sys, os, glob, hashlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth
tag = sys.argv[1]
N = int(sys.argv[2]) if len(sys.argv) > 2 else 5_830_000
W, H = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
sc = synth.make_scene(N, seed=0xC0FFEE + 2)
ss = mgs.SplatSet.from_arrays(**sc); scene = mgs.Scene(0); scene.add_instance(ss); scene.commit()
frames, ts, stats = {}, [], []
for i in range(0, 28):
    eye = synth.orbit_pose(i)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
    p = capi.default_params(W, H); capi.set_camera(p, V, P, eye); p.collect_timings = 1
    o = scene.render(p); ts.append(list(o.stage_ms)[:6]); stats.append((o.sorted_count, o.tile_pairs, o.error_flags))
    if i in (0, 9, 21):
        frames[f"pose{i}"] = scene.download_frame(p).copy()
t = np.array(ts[4:]).mean(axis=0)
print(f"[{tag}] N={N} {W}x{H} stages project/sort/bin/pairsort/composite/total = {t.round(4)}  fps {1e3/t[5]:.1f}  V={np.mean([s[0] for s in stats]):.0f} D={np.mean([s[1] for s in stats]):.0f} err={max(s[2] for s in stats)}")
os.makedirs("gpurun_out", exist_ok=True)
for k, v in frames.items():
    print(f"[{tag}] {k} sha1 {hashlib.sha1(v.tobytes()).hexdigest()[:16]} mean {v.astype(np.float32).mean():.6f}")
for f in sorted(glob.glob("gpurun_out/ab_*.npz")):
    other = np.load(f)
    same = all(np.array_equal(other[k].view(np.uint16), frames[k].view(np.uint16)) for k in frames if k in other and other[k].shape == frames[k].shape)
    print(f"[{tag}] vs {os.path.basename(f)}: {'BIT-IDENTICAL' if same else 'DIFFERENT'}")
np.savez(f"gpurun_out/ab_{tag}_{N}_{W}.npz", **frames)
