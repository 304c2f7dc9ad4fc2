"""child process of test_gpu_parity.test_key_sort_variants_are_bit_identical_to_the_stable_sort: runs under the sort knobs
(MGS_SORT_REMAP, MGS_RAW_SORT) and checks every sort against numpy's stable sort"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import vk_gaussian_splatting_amd as mgs  # noqa: E402
from vk_gaussian_splatting_amd import capi, synth  # noqa: E402

scene = mgs.Scene(0)


def check(k, v, name):
    k = np.ascontiguousarray(k, np.uint32)
    ks, vs, _ = scene.radix_sort_host(k, v)
    o = np.argsort(k, kind="stable")
    assert np.array_equal(ks, k[o]), name
    assert np.array_equal(vs, v[o]), name


# sizes around the structural thresholds: one partition (4096), one look-back group (32 partitions = 131072), one window of
# groups (16 groups = 2 M), strip-sized, multi-million
for n in [0, 1, 2, 255, 256, 257, 4095, 4096, 4097, 8191, 8193, 40_000, 131_071, 131_072, 131_073, 262_145, 2_097_152, 2_097_153,
          2_500_000, 4_190_208, 4_194_304, 4_194_305, 5_000_011]:  # (1 023 / 1 024 partitions: the largest sorts of the flat level 2; 1 025: the chain)
    rng = np.random.default_rng(n)
    check(rng.integers(0, 2**32, n, dtype=np.uint32), rng.integers(0, 2**32, n, dtype=np.uint32), f"n={n}")
rng = np.random.default_rng(99)
n = 400_003
v = np.arange(n, dtype=np.uint32)[::-1].copy()  # values deliberately NOT in index order
cases = {
    "three values": rng.choice(np.array([7, 0x3F800000, 0xFFFFFFFF], np.uint32), n),
    "one giant run + random rest": np.where(rng.random(n) < 0.6, np.uint32(0x40490FDB), rng.integers(0, 2**32, n, dtype=np.uint32)),
    "two giant runs of adjacent values": (np.uint32(1000) + (rng.random(n) < 0.5)),
    "narrow range (10 bits)": (np.uint32(0xC0000000) + rng.integers(0, 1024, n)),
    "narrow range across a byte boundary, heavy ties": (np.uint32(0x00FFFFF0) + rng.integers(0, 34, n)),
    "zeros and max": np.where(rng.random(n) < 0.5, np.uint32(0), np.uint32(0xFFFFFFFF)),
    "giant run inside a dense ramp": np.concatenate([np.arange(150_000, dtype=np.uint32) * 3, np.full(100_003, 200_000, np.uint32),
                                                     np.arange(150_000, dtype=np.uint32) * 3 + 1]),
    "sorted": np.sort(rng.integers(0, 2**32, n, dtype=np.uint32)),
    "reverse sorted": np.sort(rng.integers(0, 2**32, n, dtype=np.uint32))[::-1].copy(),
    "float bit patterns of depths": np.float32(1.0 - 0.1 / rng.uniform(0.2, 40.0, n)).view(np.uint32),
}
for name, k in cases.items():
    assert k.size == n, name
    check(k.astype(np.uint32), v, name)
# one value over thousands of partitions (every digit count sits in one column of the look-back)
n2 = 6_000_000
k2 = np.full(n2, 5, np.uint32)
k2[rng.integers(0, n2, 1000)] = rng.integers(0, 2**32, 1000, dtype=np.uint32)
check(k2, np.arange(n2, dtype=np.uint32), "6 M keys, one value")
# ... and over exactly 32 full groups: every counted sum of the flat level 2 carries its largest value (32 x 4096 in one digit)
n3 = 4_194_304
k3 = np.full(n3, 0x3F7FFFFF, np.uint32)
k3[rng.integers(0, n3, 1000)] = rng.integers(0, 2**32, 1000, dtype=np.uint32)
check(k3, np.arange(n3, dtype=np.uint32)[::-1].copy(), "4 M keys, one value, 1 024 partitions")
print("SORTS_OK")

# frames: the in-frame key sort feeds binning and compositing
sc = synth.make_scene(300_000, seed=5)
scene.add_instance(mgs.SplatSet.from_arrays(**sc))
scene.commit()
hh = hashlib.sha1()
views = [(synth.orbit_pose(1), (1280, 720)), (synth.orbit_pose(17), (640, 480)), (synth.orbit_pose(40), (1920, 1080)),
         # inside the cloud, looking along it: depth keys over many exponents (the pass elision must fall back or cope)
         (np.array([0.05, 0.02, 0.1], np.float32), (800, 600)), (np.array([0.0, 0.0, 0.35], np.float32), (640, 360))]
for pose, (eye, (w, h)) in enumerate(views):
    V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, w, h)
    p = capi.default_params(w, h)
    capi.set_camera(p, V, P, eye)
    so = scene.sort_keys(p)
    gk, gi = scene.sort_download(so.count)
    hh.update(gk.tobytes())
    hh.update(gi.tobytes())
    print("STATS view", pose, "count", so.count, "passes", so.passes, "remap on / values", list(so.reserved)[:2])
    o = scene.render(p)
    hh.update(np.ascontiguousarray(scene.download_frame(p)).tobytes())
    for strip in ((0, 8), (10, 20)):
        p.strip_row_begin, p.strip_row_end = strip
        scene.render(p)
        hh.update(np.ascontiguousarray(scene.download_frame(p)).tobytes())
# the 3DGUT front end hands over to the same sort (its own project kernel)
eye = synth.orbit_pose(9)
V, P = mgs.camera_lookat_perspective(eye, [0, 0, 0], [0, 1, 0], 60.0, 0.1, 2000.0, 960, 540)
p = capi.default_params(960, 540)
capi.set_camera(p, V, P, eye)
p.pipeline = capi.PIPELINE_3DGUT
scene.render(p)
hh.update(np.ascontiguousarray(scene.download_frame(p)).tobytes())
print("FRAMES_SHA1", hh.hexdigest())
