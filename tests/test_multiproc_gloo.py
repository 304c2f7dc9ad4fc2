"""world_size-2 gloo test (CPU) of the multi-GPU glue: strip partition + one all-gather reassembles
exactly the single-process frame.  The strips here are cut from an oracle frame — the GPU version of
the same property (strips rendered by the HIP path are bit-identical to the full frame) is in
tests/test_gpu_parity.py."""
import os
import subprocess
import sys

import numpy as np

from conftest import ROOT

WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["MGS_ROOT"])
import numpy as np, torch, torch.distributed as dist
from vk_gaussian_splatting_amd import multigpu
dist.init_process_group("gloo")
rank, ws = dist.get_rank(), dist.get_world_size()
H, W = 120, 160
full = np.load(os.environ["MGS_FRAME"])            # [H,W,4] float16
ok = True
# equal strips and cost-balanced (unequal) strips: the table every rank derives from the same per-row costs
cost = np.abs(full.astype(np.float32)).sum(axis=(1, 2)).reshape(-1, 8).sum(axis=1)[: multigpu.tile_rows(H)] if H % 8 == 0 else None
tables = [[multigpu.strip_rows(H, ws, r)[0] for r in range(ws)] + [multigpu.tile_rows(H)],
          multigpu.balanced_bounds(np.concatenate([np.zeros(2), np.arange(multigpu.tile_rows(H) - 2) ** 2.0]), ws)]
for bounds in tables:
    b, e = bounds[rank], bounds[rank + 1]
    R = multigpu.padded_strip_rows(bounds)
    strip = torch.zeros((R, W, 4), dtype=torch.float16)
    y0, y1 = b * 16, min(e * 16, H)
    if y1 > y0:
        strip[: y1 - y0] = torch.from_numpy(full[y0:y1])
    out = multigpu.assemble(multigpu.gather_strips(strip, ws), bounds, H)
    ok = ok and out.shape[0] == H and torch.equal(out, torch.from_numpy(full))
    ok = ok and bounds[0] == 0 and bounds[-1] == multigpu.tile_rows(H) and all(bounds[i] <= bounds[i + 1] for i in range(ws))
flag = torch.tensor([1 if ok else 0]); dist.all_reduce(flag, op=dist.ReduceOp.MIN)
if rank == 0:
    print("GATHER_OK" if int(flag) == 1 else "GATHER_MISMATCH")
dist.destroy_process_group()
'''


def test_two_rank_strip_gather(tmp_path):
    g = np.load(os.path.join(ROOT, "tests", "golden", "frame_two_instances.npz"))
    np.save(tmp_path / "frame.npy", g["image"])
    (tmp_path / "worker.py").write_text(WORKER)
    env = dict(os.environ, MGS_ROOT=ROOT, MGS_FRAME=str(tmp_path / "frame.npy"), MASTER_ADDR="127.0.0.1")
    for ws in (2, 3):
        r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ws}",
                            "--master-addr", "127.0.0.1", "--master-port", str(29620 + ws), str(tmp_path / "worker.py")],
                           env=env, capture_output=True, text=True, timeout=300)
        assert "GATHER_OK" in r.stdout, r.stdout + r.stderr
