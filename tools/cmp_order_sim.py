"""What a cost-aware launch order would buy k_composite (debug build -DMGS_CMP_TRACE): per-region durations of consecutive
poses -> list-scheduling simulation of the kernel span on 1536 slots for several orders.
Usage: MGS_GRAPH=0 MGS_CMP_TRACE_FILE=/tmp/t.bin python tools/cmp_order_sim.py"""
import os, sys, heapq
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth

W, H = 1920, 1080
sc = synth.make_scene(5_830_000, seed=1)
scene = mgs.Scene(0)
scene.add_instance(mgs.SplatSet.from_arrays(**sc))
scene.commit()


def trace(pose):
    eye = synth.orbit_pose(pose)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
    p = capi.default_params(W, H)
    capi.set_camera(p, V, P, eye)
    for _ in range(3):
        scene.render(p, want_stats=True)
    a = np.fromfile(os.environ["MGS_CMP_TRACE_FILE"], np.uint64).reshape(-1, 10)
    a = a[a[:, 1] > 0]
    reg = (a[:, 9] >> np.uint64(32)).astype(np.int64)
    dur = (a[:, 1].astype(np.int64) - a[:, 0].astype(np.int64)) / 100.0
    start = a[:, 0].astype(np.int64)
    order = np.argsort(start, kind="stable")          # the order the hardware started them in
    span = (a[:, 1].astype(np.int64).max() - start.min()) / 100.0
    return reg[order], dur[order], span


def simulate(durs, slots=1536):
    h = [0.0] * slots
    heapq.heapify(h)
    end = 0.0
    for d in durs:
        t = heapq.heappop(h) + d
        end = max(end, t)
        heapq.heappush(h, t)
    return end


prev = {}
for pose in range(0, 8):
    reg, dur, span = trace(pose)
    cost = dict(zip(reg.tolist(), dur.tolist()))
    rs = np.random.default_rng(pose)
    line = (f"pose {pose}: measured span {span:.1f} us; simulated: launch order {simulate(dur):.1f}, reversed {simulate(dur[::-1]):.1f}, "
            f"random {np.mean([simulate(rs.permutation(dur)) for _ in range(5)]):.1f}, longest-first (own durations) {simulate(np.sort(dur)[::-1]):.1f}")
    for lag in (1, 3):
        if pose - lag in prev:
            pc = prev[pose - lag]
            key = np.array([pc.get(r, 0.0) for r in reg.tolist()])
            o = np.argsort(-key, kind="stable")
            rho = np.corrcoef(key, dur)[0, 1]
            line += f", by pose-{lag} cost {simulate(dur[o]):.1f} (corr {rho:.2f})"
    print(line, flush=True)
    prev[pose] = cost
print("lower bound (sum / 1536):", round(float(dur.sum()) / 1536, 1), " longest workgroup:", round(float(dur.max()), 1))

# ---- same-frame predictors of a region's cost (no history): centre counts / opacity mass per region from the splat data ----
def region_stats(pose):
    eye = synth.orbit_pose(pose)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
    pos = sc["positions"].astype(np.float64)
    h = np.concatenate([pos, np.ones((pos.shape[0], 1))], axis=1)
    clip = h @ (np.asarray(P, np.float64) @ np.asarray(V, np.float64)).T
    w = clip[:, 3]
    ok = w > 1e-6
    ndc = clip[ok, :3] / w[ok, None]
    inside = (np.abs(ndc[:, 0]) < 1) & (np.abs(ndc[:, 1]) < 1) & (ndc[:, 2] > 0) & (ndc[:, 2] < 1)
    px = (ndc[inside, 0] + 1) * 0.5 * W
    py = (ndc[inside, 1] + 1) * 0.5 * H
    alpha = 1.0 / (1.0 + np.exp(-sc["opacity"].reshape(-1).astype(np.float64)))[ok][inside]
    colsX = (W // 16 + 1) // 2
    rid = (py // 16).astype(np.int64) * colsX + (px // 32).astype(np.int64)
    n = ((H + 15) // 16) * colsX
    return np.bincount(rid, minlength=n), np.bincount(rid, weights=alpha, minlength=n)


print("--- same-frame predictors")
for pose in (0, 5):
    reg, dur, span = trace(pose)
    cnt, mass = region_stats(pose)
    c, m = cnt[reg].astype(np.float64), mass[reg]
    for name, key in (("few centres first", -c), ("little opacity mass first", -m), ("1/(1+count)", 1.0 / (1.0 + c)),
                      ("1/(8+mass)", 1.0 / (8.0 + m))):
        o = np.argsort(-key, kind="stable")
        print(f"pose {pose}: {name:28s} simulated span {simulate(dur[o]):.1f} us (corr with duration {np.corrcoef(key, dur)[0, 1]:.2f}); "
              f"launch order {simulate(dur):.1f}, own durations {simulate(np.sort(dur)[::-1]):.1f}")
