"""per-strip counts and stage times of the tile-row strip partition, rendered one strip at a time on ONE GPU (what each
of G GPUs would do per frame), equal rows and cost-balanced rows:  python tools/strip_costs.py [W H [G]]"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import capi, synth, multigpu
N = 5_830_000
W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1920, 1080)
G = int(sys.argv[3]) if len(sys.argv) > 3 else 8
sc = synth.make_scene(N, seed=0xC0FFEE + 2)
ss = mgs.SplatSet.from_arrays(**sc); scene = mgs.Scene(0); scene.add_instance(ss); scene.commit()


def params(pose):
    eye = synth.orbit_pose(pose)
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, W, H)
    p = capi.default_params(W, H); capi.set_camera(p, V, P, eye)
    return p


rows = multigpu.tile_rows(H)
cost = np.zeros(rows)
for pose in range(0, 64, 8):
    scene.render(params(pose)); cost += scene.row_costs(H)
tables = {"equal": [multigpu.strip_rows(H, G, r)[0] for r in range(G)] + [rows], "balanced": multigpu.balanced_bounds(cost, G)}
p = params(3); p.collect_timings = 1
for _ in range(3): o = scene.render(p)
full = np.array(list(o.stage_ms)[:6])
print(f"{W}x{H} full frame: sorted {o.sorted_count} stages {full.round(3)}")
res = {"resolution": [W, H], "gpus": G, "full_frame_ms": full.tolist(), "tables": {}}
for name, b in tables.items():
    worst = 0.0
    rec = []
    for r in range(G):
        p.strip_row_begin, p.strip_row_end = b[r], b[r + 1]
        if b[r + 1] <= b[r]:
            continue
        for _ in range(3): o = scene.render(p)
        st = np.array(list(o.stage_ms)[:6])
        worst = max(worst, st[5])
        rec.append(dict(rank=r, rows=[b[r], b[r + 1]], sorted=int(o.sorted_count), pairs=int(o.tile_pairs), stage_ms=st.round(4).tolist()))
        print(f"{name:9s} strip {r} rows [{b[r]},{b[r+1]}) frustum {o.frustum_count:8d} sorted {o.sorted_count:8d} pairs {o.tile_pairs:8d} stages {st.round(3)}")
    print(f"{name}: heaviest strip {worst:.3f} ms vs full frame {full[5]:.3f} ms -> {full[5] / worst:.2f}x before the exchange")
    res["tables"][name] = dict(bounds=b, strips=rec, heaviest_ms=float(worst), projected_speedup=float(full[5] / worst))
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open(f"gpurun_out/strip_costs_{W}x{H}_g{G}.json", "w"), indent=1)
