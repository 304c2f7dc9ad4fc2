"""Generates the golden fixtures under tests/golden/.  Run in the DEV container only (needs
/root/reference to build oracle/_ref):   python tests/golden/make_golden.py

Sources of truth:
  * ingest_*  : the REFERENCE's own miniply / spz / splat_set.h, compiled unmodified into
                oracle/_ref/libref_ingest.so (oracle/Makefile).  Inputs (.ply/.spz) are synthetic and
                committed next to the expected arrays.
  * frame_*   : the CPU oracle (oracle/mgs_oracle.cpp).  These pin the ORACLE against regressions;
                they are not reference output (the reference's shaders cannot run here).
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import binding as ob  # noqa: E402
from vk_gaussian_splatting_amd import synth  # noqa: E402


def lookat(eye, c, up):
    eye, c, up = (np.asarray(a, np.float32) for a in (eye, c, up))
    f = c - eye
    f /= np.linalg.norm(f)
    s = np.cross(f, up)
    s /= np.linalg.norm(s)
    u = np.cross(s, f)
    V = np.eye(4, dtype=np.float32)
    V[0, :3], V[1, :3], V[2, :3] = s, u, -f
    V[0, 3], V[1, 3], V[2, 3] = -s @ eye, -u @ eye, f @ eye
    return V


def persp(fov, aspect, n, f, flip=False):
    t = np.tan(np.radians(fov) / 2)
    P = np.zeros((4, 4), np.float32)
    P[0, 0] = 1 / (aspect * t)
    P[1, 1] = (-1 if flip else 1) / t
    P[2, 2] = f / (n - f)
    P[3, 2] = -1
    P[2, 3] = -(f * n) / (f - n)
    return P


def main():
    ob.build(force=True)
    R = ob.ref_lib()
    assert R is not None, "oracle/_ref could not be built (needs /root/reference)"
    meta = {}

    # ---- PLY ingest --------------------------------------------------------------------------
    for name, n, cpc, fmt in [("ply_sh3", 1000, 15, "binary_little_endian"), ("ply_sh0", 300, 0, "binary_little_endian"),
                              ("ply_ascii", 40, 15, "ascii"), ("ply_be", 64, 15, "binary_big_endian")]:
        sc = synth.make_scene(n, seed=100 + n, sh_coeffs_per_channel=cpc)
        path = os.path.join(HERE, f"ingest_{name}.ply")
        synth.write_ply(path, sc, fmt=fmt)
        ref = ob.ref_load(path)
        assert ref is not None
        np.savez_compressed(os.path.join(HERE, f"ingest_{name}.npz"), **ref)
        meta[name] = dict(n=n, sh_degree=int(ref["sh_degree"]))

    # ---- SPZ ingest (files written by the reference's own packer) -------------------------------
    for name, n, deg in [("spz_sh3", 500, 3), ("spz_sh1", 200, 1), ("spz_sh0", 100, 0)]:
        shdim = {0: 0, 1: 3, 2: 8, 3: 15}[deg]
        rng = np.random.default_rng(200 + n)
        pos = rng.normal(0, 2, (n, 3)).astype(np.float32)
        scales = rng.normal(-4, 1, (n, 3)).astype(np.float32)
        rot = rng.normal(0, 1, (n, 4)).astype(np.float32)
        rot /= np.linalg.norm(rot, axis=1, keepdims=True)
        alphas = rng.normal(0, 2, n).astype(np.float32)
        colors = rng.normal(0, 1, (n, 3)).astype(np.float32)
        sh = rng.normal(0, 0.2, (n, shdim * 3)).astype(np.float32)
        path = os.path.join(HERE, f"ingest_{name}.spz")
        P = ob._p
        rc = R.ref_spz_save(os.fsencode(path), n, deg, P(pos.reshape(-1)), P(scales.reshape(-1)), P(rot.reshape(-1)),
                            P(alphas), P(colors.reshape(-1)), P(sh.reshape(-1) if shdim else np.zeros(1, np.float32)),
                            4)  # from = RUB
        assert rc == 0
        ref = ob.ref_load(path)
        assert ref is not None
        np.savez_compressed(os.path.join(HERE, f"ingest_{name}.npz"), **ref)
        meta[name] = dict(n=n, sh_degree=int(ref["sh_degree"]))

    # ---- small facts from the reference headers -------------------------------------------------
    facts = {"max_sh_degree": {}, "flip": {}}
    for n in (1, 7, 1000):
        for per in (0, 3, 9, 12, 24, 27, 45, 48):
            facts["max_sh_degree"][f"{per * n},{n}"] = int(R.ref_max_sh_degree(per * n, n))
    facts["max_sh_degree"]["0,0"] = int(R.ref_max_sh_degree(0, 0))
    sh, fp, fq = (np.zeros(15, np.float32), np.zeros(3, np.float32), np.zeros(3, np.float32))
    for a, b in [(6, 4), (4, 6), (4, 4), (6, 8), (1, 4)]:
        R.ref_flip_sh(a, b, ob._p(sh), ob._p(fp), ob._p(fq))
        facts["flip"][f"{a},{b}"] = dict(sh=sh.tolist(), p=fp.tolist(), q=fq.tolist())
    meta["facts"] = facts

    # ---- oracle frames (regression pins of the oracle itself) -----------------------------------
    sc = synth.make_scene(3000, seed=42)
    ps = ob.PreparedSet(sc)
    eye = np.array([3.5, 1.2, 1.0], np.float32)
    V, Pm = lookat(eye, [0, 0, 0], [0, 1, 0]), persp(60, 160 / 120, 0.1, 2000)
    M2 = np.eye(4, dtype=np.float32)
    M2[:3, 3] = [0.5, -0.2, 0.3]
    M2[:3, :3] *= 0.7
    inst = ob.make_instances([(ps, None), (ps, M2)])
    fr = ob.make_frame(V, Pm, eye, 160, 120, target_fp16=1)
    img, st = ob.render(fr, inst)
    keys, ids = ob.key_cull(fr, inst)
    ks, vs = ob.sort_stable(keys, ids)
    np.savez_compressed(os.path.join(HERE, "frame_two_instances.npz"), image=img.astype(np.float16), sorted_ids=vs,
                        sorted_keys=ks, view=V, proj=Pm, eye=eye, transform1=M2)
    meta["frame_two_instances"] = dict(scene_seed=42, n=3000, width=160, height=120, **st)
    json.dump(meta, open(os.path.join(HERE, "meta.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps({k: v for k, v in meta.items() if k != "facts"}, indent=1))


if __name__ == "__main__":
    main()
