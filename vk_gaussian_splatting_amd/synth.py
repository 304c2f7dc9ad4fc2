"""Synthetic, seeded INRIA-layout scenes — no real .ply exists on this machine.

Distributions are build-defined and meant to mimic outdoor captures: 70 % "object" splats
N(0, diag(1.5,0.6,1.5)) with small scales, 30 % "background" in a shell r in [4,30] with larger
scales; SH degree 3.  Arrays are produced directly in the renderer's RUB frame, i.e. what
SplatSet holds after load (src/splat_set.h).
"""
import numpy as np

SCENES = {"syn_flowers": (500_000, 0), "syn_train": (1_030_000, 1), "syn_garden": (5_830_000, 2)}


def make_scene(n, seed=0xC0FFEE, sh_coeffs_per_channel=15, dtype=np.float32):
    rng = np.random.Generator(np.random.PCG64(seed))
    n_obj = int(round(0.7 * n))
    n_bg = n - n_obj
    pos = np.empty((n, 3), np.float32)
    pos[:n_obj] = rng.standard_normal((n_obj, 3), dtype=np.float32) * np.array([1.5, 0.6, 1.5], np.float32)
    d = rng.standard_normal((n_bg, 3), dtype=np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True) + 1e-12
    r = rng.uniform(4.0, 30.0, size=(n_bg, 1)).astype(np.float32)
    pos[n_obj:] = d * r
    scale = np.empty((n, 3), np.float32)
    scale[:n_obj] = (rng.standard_normal((n_obj, 1), dtype=np.float32) * 0.9 - 4.6)
    scale[n_obj:] = (rng.standard_normal((n_bg, 1), dtype=np.float32) * 0.8 - 2.3)
    scale += rng.standard_normal((n, 3), dtype=np.float32) * 0.5
    rot = rng.standard_normal((n, 4), dtype=np.float32)
    rot /= np.linalg.norm(rot, axis=1, keepdims=True) + 1e-12
    opacity = (rng.standard_normal(n, dtype=np.float32) * 2.0 - 0.5)
    f_dc = rng.standard_normal((n, 3), dtype=np.float32)
    f_rest = (rng.standard_normal((n, 3 * sh_coeffs_per_channel), dtype=np.float32) * 0.12
              if sh_coeffs_per_channel else np.zeros((n, 0), np.float32))
    # interleave object/background so that storage order is not spatially sorted (like trained models)
    perm = rng.permutation(n)
    return dict(positions=pos[perm], f_dc=f_dc[perm], f_rest=f_rest[perm], opacity=opacity[perm],
                scale=scale[perm], rotation=rot[perm])


def named_scene(name):
    n, sid = SCENES[name]
    return make_scene(n, seed=0xC0FFEE + sid)


def orbit_pose(i, count=64, radius=4.0, height=1.5):
    """eye of pose i on the fixed benchmark orbit: centre origin, up +y, fov 60, clip 0.1-2000."""
    th = 2.0 * np.pi * i / count
    return np.array([radius * np.cos(th), height, radius * np.sin(th)], np.float32)


def write_ply(path, arrays, fmt="binary_little_endian", to_rdf=True):
    """Write an INRIA-layout .ply.  Arrays are in RUB (renderer frame); the file convention is RDF,
    so with to_rdf the inverse of the loader's conversion is applied first (it is an involution)."""
    a = {k: np.array(v, np.float32, copy=True) for k, v in arrays.items()}
    n = a["positions"].shape[0]
    cpc = a["f_rest"].shape[1] // 3 if a["f_rest"].size else 0
    if to_rdf:
        a["positions"][:, 1:] *= -1
        a["rotation"][:, 2:] *= -1
        flip = np.array([-1, -1, 1, -1, 1, 1, -1, 1, -1, 1, -1, -1, 1, -1, 1], np.float32)[:cpc]
        if cpc:
            a["f_rest"] = (a["f_rest"].reshape(n, 3, cpc) * flip).reshape(n, 3 * cpc)
    names = ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    names += [f"f_rest_{i}" for i in range(3 * cpc)]
    names += ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    cols = [a["positions"], np.zeros((n, 3), np.float32), a["f_dc"]]
    if cpc:
        cols.append(a["f_rest"])
    cols += [a["opacity"].reshape(n, 1), a["scale"], a["rotation"]]
    table = np.concatenate(cols, axis=1).astype(np.float32)
    header = "ply\nformat %s 1.0\ncomment synthetic 3DGS scene\nelement vertex %d\n" % (fmt, n)
    header += "".join(f"property float {nm}\n" for nm in names) + "end_header\n"
    with open(path, "wb") as f:
        f.write(header.encode())
        if fmt == "ascii":
            for row in table:
                f.write((" ".join(repr(float(x)) for x in row) + "\n").encode())
        elif fmt == "binary_big_endian":
            f.write(table.astype(">f4").tobytes())
        else:
            f.write(table.astype("<f4").tobytes())
