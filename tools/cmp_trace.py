"""Debug-build experiment (csrc built with -DMGS_CMP_TRACE, see tools/ab_lib.sh): per-workgroup start / end times of
k_composite on the garden-sized frame -> where the kernel's time goes (tail, imbalance, per-iteration latency).
Usage: MGS_GRAPH=0 MGS_CMP_TRACE_FILE=/tmp/t.bin python tools/cmp_trace.py [pose]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth

poses = [int(x) for x in sys.argv[1:]] or [0]
W, H = 1920, 1080
sc = synth.make_scene(5_830_000, seed=1)
if os.environ.get("SCENE") == "fog":   # bench.py --scene fog: regions that never saturate
    sc["opacity"] = (sc["opacity"] - 2.5).astype(np.float32)
scene = mgs.Scene(0)
scene.add_instance(mgs.SplatSet.from_arrays(**sc))
scene.commit()
def run(pose):
  eye = synth.orbit_pose(pose)
  V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
  p = capi.default_params(W, H)
  capi.set_camera(p, V, P, eye)
  if os.environ.get("ALPHA_SUM"):   # trace the additive-alpha mode (MGS_ALPHA_SUM) instead of the default
      p.alpha_mode = capi.ALPHA_SUM
  for _ in range(4):
      scene.render(p, want_stats=True)
  path = os.environ["MGS_CMP_TRACE_FILE"]
  a = np.fromfile(path, np.uint64).reshape(-1, 10)
  a = a[a[:, 1] > 0]
  t0, t1 = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64)
  base = t0.min()
  dur = (t1 - t0) / 100.0   # us (100 MHz)
  start = (t0 - base) / 100.0
  end = (t1 - base) / 100.0
  print(f"workgroups that ran: {len(a)}, kernel span {end.max():.1f} us, sum of durations {dur.sum():.0f} us "
        f"(= {dur.sum() / end.max():.1f} workgroups resident on average; 256 CUs x 6 = 1536 slots)")
  print("duration us percentiles 10/50/90/99/max:", np.percentile(dur, [10, 50, 90, 99, 100]).round(1))
  print("start us percentiles 10/50/90/99/max:", np.percentile(start, [10, 50, 90, 99, 100]).round(1))
  it, rounds, scanned, staged = a[:, 4] & 0xFFFFFFFF, a[:, 5], a[:, 2], a[:, 3]
  needed = a[:, 4] >> 32   # staged (= shaded) records up to the one at which the region's last wave retired
  print(f"staged {int(staged.sum())}, of which needed before the region retired {int(needed.sum())} ({needed.sum() / max(staged.sum(), 1):.2f}); per region needed/staged p10/50/90:", np.percentile(needed / np.maximum(staged, 1), [10, 50, 90]).round(2))
  ok = it > 0
  print("iterations per workgroup 50/90/99/max:", np.percentile(it, [50, 90, 99, 100]), " stage-A rounds:", np.percentile(rounds, [50, 90, 99, 100]))
  print("us per iteration (median over workgroups with >= 4 iterations):", np.median(dur[it >= 4] / it[it >= 4]).round(2),
        " us per stage-A round:", np.median(dur[rounds >= 4] / rounds[rounds >= 4]).round(2))
  pa, psh, pb, ll = a[:, 6] / 100.0, a[:, 7] / 100.0, a[:, 8] / 100.0, a[:, 9]
  head = dur - pa - psh - pb
  print(f"phase totals (workgroup-us): stage A {pa.sum():.0f}, shading {psh.sum():.0f}, blend {pb.sum():.0f}, head+tail {head.sum():.0f}")
  one = it == 1
  print(f"workgroups with ONE iteration ({int(one.sum())}): median us stage A {np.median(pa[one]):.1f}, shading {np.median(psh[one]):.1f}, "
        f"blend {np.median(pb[one]):.1f}, head+tail {np.median(head[one]):.1f}; staged {np.median(staged[one]):.0f}, rounds {np.median(rounds[one]):.0f}")
  many = it >= 4
  if many.any():
      print(f"workgroups with >= 4 iterations ({int(many.sum())}): per iteration us stage A {np.median(pa[many] / it[many]):.1f}, shading {np.median(psh[many] / it[many]):.1f}, "
            f"blend {np.median(pb[many] / it[many]):.1f}; staged per iteration {np.median(staged[many] / it[many]):.0f}, rounds per iteration {np.median(rounds[many] / it[many]):.1f}")
  print("list length of the workgroup's bin 50/90/max:", np.percentile(ll, [50, 90, 100]))
  print("scanned per workgroup 50/90/99/max:", np.percentile(scanned, [50, 90, 99, 100]), " staged:", np.percentile(staged, [50, 90, 99, 100]))
  # resident workgroups over time
  ts = np.linspace(0, end.max(), 29)
  res = [(int(((start <= x) & (end > x)).sum())) for x in ts]
  print("resident workgroups over time:", res)
  # the longest workgroups
  idx = np.argsort(-dur)[:8]
  for i in idx:
      print(f"  wg: start {start[i]:.1f} dur {dur[i]:.1f} us, iterations {it[i]}, rounds {rounds[i]}, scanned {scanned[i]}, staged {staged[i]}")
  late = end > 0.8 * end.max()
  print(f"workgroups still running in the last 20 % of the span: {int(late.sum())}; their median duration {np.median(dur[late]):.1f} us, median start {np.median(start[late]):.1f}")

for q in poses:
    print('--- pose', q)
    run(q)
