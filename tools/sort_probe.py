"""isolated sort timing on device-resident data (python tools/sort_probe.py [n ...]): the stand-alone sort of n random
32-bit keys and of depth-like keys, best of 5, through mgs_radix_sort_u32 (HIP events around the sort's launches)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import vk_gaussian_splatting_amd as mgs
sizes = [int(x) for x in sys.argv[1:]] or [40_000, 700_000, 4_200_000, 12_400_000]
scene = mgs.Scene(0)
rng = np.random.default_rng(1)
for n in sizes:
    for name, k in (("random", rng.integers(0, 2**32, n, dtype=np.uint32)),
                    ("depth-like", np.float32(-(1.0 - 0.1 / np.abs(rng.normal(4.3, 0.9, n)).clip(0.3, 40))).view(np.uint32) ^ np.uint32(0xFFFFFFFF))):
        v = np.arange(n, dtype=np.uint32)
        best = 1e9
        for _ in range(5):
            dk = torch.from_numpy(k.view(np.int32)).cuda()
            dv = torch.from_numpy(v.view(np.int32)).cuda()
            torch.cuda.synchronize()
            ms = scene.radix_sort_device(dk.data_ptr(), dv.data_ptr(), n)
            best = min(best, ms)
        ok = np.array_equal(dk.cpu().numpy().view(np.uint32), np.sort(k, kind="stable"))
        print(f"n={n:>9} {name:<10} {best*1e3:8.1f} us  {n/best/1e6:7.2f} Gkeys/s  68B/key frac of 8 TB/s {68*n/best/1e-3/8e12:.3f}  sorted={ok}")
scene.close()
