"""CPU tests: the oracle against the golden vectors produced by the reference-compiled ingest
(tests/golden/make_golden.py) and against independent numpy restatements."""
import os

import numpy as np
import pytest

from conftest import GOLDEN, lookat, persp
import vk_gaussian_splatting_amd as mgs
from vk_gaussian_splatting_amd import synth


def test_max_sh_degree_matches_reference_facts(ob, golden_meta):
    L = ob.lib()
    for key, want in golden_meta["facts"]["max_sh_degree"].items():
        flen, n = (int(x) for x in key.split(","))
        assert L.orc_max_sh_degree(flen, n) == want, key


def test_flip_sh_matches_reference(ob, golden_meta):
    out = np.zeros(15, np.float32)
    ob.lib().orc_flip_sh_rdf_to_rub(ob._p(out))
    ref = golden_meta["facts"]["flip"]["6,4"]  # RDF -> RUB
    assert out.tolist() == ref["sh"]
    assert ref["p"] == [1.0, -1.0, -1.0] and ref["q"] == [1.0, -1.0, -1.0]


@pytest.mark.parametrize("name,cpc", [("ply_sh3", 15), ("ply_sh0", 0), ("ply_ascii", 15), ("ply_be", 15)])
def test_convert_rdf_to_rub_matches_reference_loader(ob, name, cpc, golden_meta):
    """file arrays (RDF, as written by synth.write_ply) -> oracle conversion == reference loader output"""
    n = golden_meta[name]["n"]
    sc = synth.make_scene(n, seed=100 + n, sh_coeffs_per_channel=cpc)  # same seed as make_golden.py
    g = np.load(os.path.join(GOLDEN, f"ingest_{name}.npz"))
    if name == "ply_ascii":  # ascii round trip through repr() is exact for fp32
        pass
    # what the file contains = inverse conversion of the RUB arrays
    pos = sc["positions"].copy(); rot = sc["rotation"].copy(); fr = sc["f_rest"].copy()
    pos[:, 1:] *= -1; rot[:, 2:] *= -1
    if cpc:
        flip = np.array(golden_meta["facts"]["flip"]["6,4"]["sh"], np.float32)[:cpc]
        fr = (fr.reshape(n, 3, cpc) * flip).reshape(n, -1)
    pos, rot, fr = (np.ascontiguousarray(a.reshape(-1)) for a in (pos, rot, fr))
    ob.lib().orc_convert_rdf_to_rub(ob._p(pos), ob._p(rot), ob._p(fr) if cpc else None, n, cpc)
    assert np.array_equal(pos, g["positions"])
    assert np.array_equal(rot, g["rotation"])
    assert np.array_equal(fr, g["f_rest"])


def test_half_conversion_matches_numpy(ob):
    L = ob.lib()
    rng = np.random.default_rng(1)
    vals = np.concatenate([rng.normal(0, 1, 2000), rng.normal(0, 1e-5, 500), rng.normal(0, 3e4, 500),
                           [0.0, -0.0, 65504.0, 65519.9, 65520.0, 1e-8, 5.96e-8, 2.98e-8, 2.9802322e-8, 6.1e-5, 1.0, 0.333333]]
                          ).astype(np.float32)
    for v in vals:
        h = L.orc_float_to_half(float(v))
        want = np.float16(v)
        assert h == want.view(np.uint16), (v, h, want.view(np.uint16))
        assert L.orc_half_to_float(h) == np.float32(want) or np.isinf(want)


def test_encode_key_is_order_preserving(ob):
    L = ob.lib()
    rng = np.random.default_rng(2)
    v = np.sort(np.concatenate([rng.normal(0, 1, 1000), [-0.0, 0.0, -1e-30, 1e-30, -1e30, 1e30]]).astype(np.float32))
    k = np.array([L.orc_encode_key(float(x)) for x in v], dtype=np.uint64)
    strictly = np.diff(v) > 0  # (-0.0, +0.0) compare equal but encode to adjacent keys
    assert np.all(np.diff(k.astype(np.int64))[strictly] > 0)
    # dist.comp.slang:33-38 bit formula
    bits = v.view(np.uint32).astype(np.uint64)
    want = bits ^ np.where(bits >> 31, 0xFFFFFFFF, 0x80000000).astype(np.uint64)
    assert np.array_equal(k, want)


def test_sort_stable_matches_numpy(ob):
    rng = np.random.default_rng(3)
    for n in (0, 1, 2, 1000, 70000):
        k = rng.integers(0, 2**32, n, dtype=np.uint32)
        k[: n // 2] &= 0xFFFF0000  # many ties
        v = np.arange(n, dtype=np.uint32)
        ks, vs = ob.sort_stable(k, v)
        o = np.argsort(k, kind="stable")
        assert np.array_equal(ks, k[o]) and np.array_equal(vs, v[o])


def test_upload_transform_against_numpy_fp64(ob):
    sc = synth.make_scene(500, seed=5)
    ps = ob.PreparedSet(sc)
    q = sc["rotation"].astype(np.float64)
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q.T
    R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
                  np.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
                  np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)], 1)
    S = np.exp(sc["scale"].astype(np.float64))
    M = R * S[:, None, :]
    Sig = M @ np.transpose(M, (0, 2, 1))
    want = np.stack([Sig[:, 0, 0], Sig[:, 0, 1], Sig[:, 0, 2], Sig[:, 1, 1], Sig[:, 1, 2], Sig[:, 2, 2]], -1)
    got = ps.cov6.reshape(-1, 6)
    # off-diagonal terms cancel: tolerance relative to the largest entry of each matrix (fp32 eps ~6e-8)
    assert np.all(np.abs(got - want) <= 2e-6 * np.abs(want).max(axis=1, keepdims=True))
    rgb = np.clip(0.5 + 0.28209479177387814 * sc["f_dc"].astype(np.float64), 0, 1)
    a = 1 / (1 + np.exp(-sc["opacity"].astype(np.float64)))
    assert np.allclose(ps.rgba.reshape(-1, 4)[:, :3], rgb, atol=1e-6)
    assert np.allclose(ps.rgba.reshape(-1, 4)[:, 3], a, atol=1e-6)
    # SH re-interleave: [coef][rgb] <- channel-major
    sh = ps.sh.reshape(-1, 15, 3)
    assert np.array_equal(sh, np.transpose(sc["f_rest"].reshape(-1, 3, 15), (0, 2, 1)))


def test_quantize_roundtrip_formats(ob):
    L = ob.lib()
    x = np.linspace(-1.2, 1.2, 1001).astype(np.float32)
    a = x.copy(); L.orc_quantize_roundtrip(ob._p(a), a.size, 2, 1)       # uint8 SH in [-1,1]
    rnd = lambda t: np.sign(t) * np.floor(np.abs(t) + 0.5)  # std::round: half away from zero
    q = np.clip(rnd(((x - np.float32(-1)) / np.float32(2)) * np.float32(255)), 0, 255)
    assert np.allclose(a, q / 255 * 2 - 1, atol=1e-6)
    b = x.copy(); L.orc_quantize_roundtrip(ob._p(b), b.size, 2, 0)       # uint8 colour in [0,1]
    assert np.allclose(b, np.clip(rnd(x * np.float32(255)), 0, 255) / 255, atol=1e-6)
    c = x.copy(); L.orc_quantize_roundtrip(ob._p(c), c.size, 1, 1)       # fp16
    assert np.array_equal(c, x.astype(np.float16).astype(np.float32))


def test_key_cull_and_projection_against_numpy_fp64(ob):
    sc = synth.make_scene(4000, seed=9)
    ps = ob.PreparedSet(sc)
    inst = ob.make_instances([(ps, None)])
    eye = np.array([4, 1.5, 0.5], np.float32)
    V, P = lookat(eye, [0, 0, 0], [0, 1, 0]), persp(60, 4 / 3, 0.1, 2000)
    fr = ob.make_frame(V, P, eye, 320, 240)
    keys, ids = ob.key_cull(fr, inst)
    p = np.concatenate([sc["positions"].astype(np.float64), np.ones((4000, 1))], 1)
    clip = (P.astype(np.float64) @ V.astype(np.float64) @ p.T).T
    ndc = clip[:, :3] / clip[:, 3:4]
    vis = (np.abs(ndc[:, 0]) <= 1.2) & (np.abs(ndc[:, 1]) <= 1.2) & (ndc[:, 2] >= -0.2) & (ndc[:, 2] <= 1.0)
    # borderline splats may flip between fp32 and fp64: allow a handful
    got = np.zeros(4000, bool); got[ids] = True
    assert (got != vis).sum() <= 3
    # projection of one clearly visible splat vs fp64 math (A5 of SURVEY.md)
    i = int(ids[len(ids) // 2])
    pr = ob.project(fr, inst, 0, i)
    if pr.valid:
        t = (V.astype(np.float64) @ p[i])
        fx, fy = P[0, 0] * 160.0, P[1, 1] * 120.0
        J = np.array([[fx / t[2], 0, -fx * t[0] / t[2] ** 2], [0, fy / t[2], -fy * t[1] / t[2] ** 2]])
        c6 = ps.cov6.reshape(-1, 6)[i].astype(np.float64)
        S = np.array([[c6[0], c6[1], c6[2]], [c6[1], c6[3], c6[4]], [c6[2], c6[4], c6[5]]])
        T = J @ V[:3, :3].astype(np.float64)
        c2 = T @ S @ T.T
        a, b, d = c2[0, 0] + 0.3, c2[0, 1], c2[1, 1] + 0.3
        h = (a + d) / 2; r = np.sqrt(max(0.1, h * h - (a * d - b * b)))
        l1, l2 = h + r, h - r
        n1 = np.hypot(*pr.basis1); n2 = np.hypot(*pr.basis2)
        assert np.isclose(n1, min(np.sqrt(8 * l1), 2048), rtol=1e-3)
        assert np.isclose(n2, min(np.sqrt(8 * l2), 2048), rtol=1e-3)
        assert abs(pr.basis1[0] * pr.basis2[0] + pr.basis1[1] * pr.basis2[1]) < 1e-3 * n1 * n2 + 1e-6
        assert np.isclose(pr.center_px[0], (ndc[i, 0] + 1) * 160, atol=1e-2)
        assert np.isclose(pr.center_px[1], (ndc[i, 1] + 1) * 120, atol=1e-2)


def test_single_splat_footprint_and_blend(ob):
    """one opaque isotropic splat in front of the camera: analytic alpha profile + 'over' with one layer"""
    sc = dict(positions=np.array([[0, 0, 0]], np.float32), f_dc=np.array([[1.0, 0.0, -1.0]], np.float32),
              f_rest=np.zeros((1, 0), np.float32), opacity=np.array([20.0], np.float32),
              scale=np.log(np.full((1, 3), 0.05, np.float32)), rotation=np.array([[1, 0, 0, 0]], np.float32))
    ps = ob.PreparedSet(sc)
    inst = ob.make_instances([(ps, None)])
    eye = np.array([0, 0, 2], np.float32)
    V, P = lookat(eye, [0, 0, 0], [0, 1, 0]), persp(60, 1.0, 0.1, 100)
    fr = ob.make_frame(V, P, eye, 64, 64)
    img, st = ob.render(fr, inst)
    assert st["quads"] == 1
    # isotropic: a == d, b == 0.  The reference floors the discriminant at 0.1 (threedgs.h.slang:97), so
    # even a round splat gets eigenvalues a +- sqrt(0.1) along e1 = normalize(1, sqrt(0.1)) — kept literally.
    a = (0.05 * (P[0, 0] * 32) / 2.0) ** 2 + 0.3
    r = np.sqrt(0.1)
    l1, l2 = a + r, a - r
    e1 = np.array([1.0, r]) / np.hypot(1.0, r)
    e2 = np.array([e1[1], -e1[0]])
    yy, xx = np.mgrid[0:64, 0:64]
    dx, dy = xx + 0.5 - 32, yy + 0.5 - 32
    A = (dx * e1[0] + dy * e1[1]) ** 2 / l1 + (dx * e2[0] + dy * e2[1]) ** 2 / l2  # == fragPos.fragPos
    alpha = np.exp(-0.5 * A)
    alpha[(A > 8) | (alpha <= 1 / 255)] = 0
    assert np.allclose(img[..., 3], alpha, atol=2e-3)
    rgb = np.clip(0.5 + 0.28209479177387814 * np.array([1.0, 0.0, -1.0]), 0, 1)
    assert np.allclose(img[..., :3], alpha[..., None] * rgb, atol=2e-3)


def test_oracle_frame_regression_golden(ob):
    g = np.load(os.path.join(GOLDEN, "frame_two_instances.npz"))
    sc = synth.make_scene(3000, seed=42)
    ps = ob.PreparedSet(sc)
    inst = ob.make_instances([(ps, None), (ps, g["transform1"])])
    fr = ob.make_frame(g["view"], g["proj"], g["eye"], 160, 120, target_fp16=1)
    keys, ids = ob.key_cull(fr, inst)
    ks, vs = ob.sort_stable(keys, ids)
    assert np.array_equal(vs, g["sorted_ids"]) and np.array_equal(ks, g["sorted_keys"])
    img, _ = ob.render(fr, inst)
    assert np.array_equal(img.astype(np.float16), g["image"])


def test_front_to_back_equals_back_to_front_colour(ob):
    sc = synth.make_scene(1500, seed=11)
    ps = ob.PreparedSet(sc)
    inst = ob.make_instances([(ps, None)])
    eye = np.array([3, 1, 2], np.float32)
    V, P = lookat(eye, [0, 0, 0], [0, 1, 0]), persp(60, 4 / 3, 0.1, 2000)
    a, _ = ob.render(ob.make_frame(V, P, eye, 128, 96), inst)
    b, _ = ob.render(ob.make_frame(V, P, eye, 128, 96, front_to_back=1), inst)
    assert ob.psnr_rgb(a, b) > 60.0
    assert np.all(b[..., 3] <= 1.0 + 1e-5)  # FTB alpha = 1 - T


def test_cpu_sorter_matches_numpy(ob):
    rng = np.random.default_rng(4)
    p0 = rng.normal(0, 3, (5000, 3)).astype(np.float32)
    p1 = rng.normal(0, 3, (3000, 3)).astype(np.float32)
    M = np.eye(4, dtype=np.float32); M[:3, 3] = [1, 2, 3]; M[0, 0] = 2
    d, c = np.array([0.3, -0.2, -0.9], np.float32), np.array([1, 1, 5], np.float32)
    for ftb in (False, True):
        dist, idx, dms, sms = ob.cpu_sort(d, c, [(p0, None), (p1, M)], front_to_back=ftb, threads=2)
        w = np.concatenate([p0, (np.c_[p1, np.ones(3000)] @ M.T)[:, :3]]).astype(np.float64)
        want = np.abs(w @ d.astype(np.float64) - d.astype(np.float64) @ c.astype(np.float64)) / np.linalg.norm(d)
        assert np.allclose(dist, want, rtol=1e-4, atol=1e-5)
        assert sorted(idx.tolist()) == list(range(8000))
        ds = dist[idx]
        assert np.all(np.diff(ds) >= 0) if ftb else np.all(np.diff(ds) <= 0)


def test_psnr_definition(ob):
    a = np.zeros((4, 5, 4), np.float32); b = a.copy(); b[..., :3] = 0.1; b[..., 3] = 7
    assert abs(ob.psnr_rgb(a, b) - 20.0) < 1e-4   # mse = 0.01 over RGB only
    assert ob.psnr_rgb(a, a) == 99.99


def _quat_to_mat(q):
    w, x, y, z = (np.asarray(q, np.float64) / np.linalg.norm(q))
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                     [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                     [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])


def test_octahedral_normal_code_round_trip(ob):
    """octahedral_normal.h.slang: 2x16-bit code, decode(encode(n)) stays within the quantisation step of n"""
    rng = np.random.default_rng(9)
    n = rng.normal(size=(500, 3))
    n /= np.linalg.norm(n, axis=1, keepdims=True)
    n = np.vstack([n, np.eye(3), -np.eye(3)]).astype(np.float32)
    for v in n:
        code, back = ob.oct_roundtrip(v)
        assert 0 <= code <= 0xFFFFFFFF
        assert abs(np.linalg.norm(back) - 1.0) < 1e-6
        assert np.allclose(back, v, atol=1e-4)
    # both hemispheres use the whole square: +z maps inside |x|+|y| <= 1, -z outside
    cz, _ = ob.oct_roundtrip([0, 0, 1])
    assert (cz & 0xFFFF) in (32767, 32768) and (cz >> 16) in (32767, 32768)


def test_splat_normal_against_numpy_fp64(ob):
    """threedgrt.h.slang:358-419 (max density plane): n = normalize(M3 * Sigma^-1 (M^-1 cam - centre)), with
    Sigma = R S^2 R^T of the normalised (w,x,y,z) quaternion — restated independently in float64"""
    sc = synth.make_scene(400, seed=21)
    ps = ob.PreparedSet(sc)
    ang = 0.7
    M = np.array([[np.cos(ang), 0, np.sin(ang), 0.5], [0, 1, 0, -0.2], [-np.sin(ang), 0, np.cos(ang), 1.0], [0, 0, 0, 1]],
                 np.float32)
    M[:3, :3] *= 1.5
    inst = ob.make_instances([(ps, None), (ps, M)])
    eye = np.array([3, 1, 2], np.float32)
    V, P = lookat(eye, [0, 0, 0], [0, 1, 0]), persp(60, 4 / 3, 0.1, 2000)
    fr = ob.make_frame(V, P, eye, 128, 96)
    pos = np.asarray(sc["positions"], np.float64).reshape(-1, 3)
    for k, Mk in ((0, np.eye(4)), (1, M.astype(np.float64))):
        cam = (np.linalg.inv(Mk) @ np.append(eye.astype(np.float64), 1.0))[:3]
        for i in range(0, 400, 7):
            R = _quat_to_mat(np.asarray(sc["rotation"], np.float64).reshape(-1, 4)[i])
            s = np.exp(np.asarray(sc["scale"], np.float64).reshape(-1, 3)[i])
            g = R @ np.diag(1.0 / (s * s)) @ R.T @ (cam - pos[i])
            n = Mk[:3, :3] @ (g / np.linalg.norm(g))
            n /= np.linalg.norm(n)
            got = ob.splat_normal(fr, inst, k, i)
            assert np.allclose(got, n, atol=2e-5), (k, i, got, n)
            assert np.dot(got, eye - (Mk @ np.append(pos[i], 1.0))[:3]) > -1e-3 or k == 1  # faces the camera
            q = ob.splat_normal(fr, inst, k, i, quantize=True)
            assert np.allclose(q, n, atol=1.5e-4)


def test_splat_normal_thin_particle_cases(ob):
    """one / two degenerate axes: the thin axis (towards the camera) / minus the view direction"""
    q = np.array([0.9, 0.1, -0.3, 0.2], np.float32)
    R = _quat_to_mat(q)
    sc = dict(positions=np.array([[0.1, -0.2, 0.3]], np.float32), f_dc=np.zeros((1, 3), np.float32),
              f_rest=np.zeros((1, 0), np.float32), opacity=np.array([3.0], np.float32),
              scale=np.log(np.array([[0.2, 1e-4, 0.1]], np.float32)), rotation=q[None])
    ps = ob.PreparedSet(sc)
    inst = ob.make_instances([(ps, None)])
    eye = np.array([1.0, 2.0, 3.0], np.float32)
    V, P = lookat(eye, [0, 0, 0], [0, 1, 0]), persp(60, 1.0, 0.1, 100)
    fr = ob.make_frame(V, P, eye, 64, 64)
    local = eye.astype(np.float64) - sc["positions"][0]
    axis = R[:, 1] * (1.0 if R[:, 1] @ local >= 0 else -1.0)
    # default threshold 1e-6: the flat splat is still "regular"; Sigma^-1 makes its normal the thin axis anyway
    assert np.allclose(ob.splat_normal(fr, inst, 0, 0), axis, atol=1e-3)
    # threshold above the thin scale: exactly the rotated axis
    assert np.allclose(ob.splat_normal(fr, inst, 0, 0, thin_particle_threshold=1e-3), axis, atol=1e-6)
    # two small axes: minus the view direction
    got = ob.splat_normal(fr, inst, 0, 0, thin_particle_threshold=0.15)
    assert np.allclose(got, local / np.linalg.norm(local), atol=1e-6)


def test_surface_normal_integration_single_and_many(ob):
    """the normal attachment is the 'under' blend of (n * opacity, opacity): alpha equals the colour's FTB alpha,
    and a pixel covered by one splat carries n * alpha"""
    sc = dict(positions=np.array([[0, 0, 0]], np.float32), f_dc=np.array([[1.0, 0.0, -1.0]], np.float32),
              f_rest=np.zeros((1, 0), np.float32), opacity=np.array([2.0], np.float32),
              scale=np.log(np.array([[0.08, 0.05, 0.002]], np.float32)), rotation=np.array([[1, 0, 0, 0]], np.float32))
    ps = ob.PreparedSet(sc)
    inst = ob.make_instances([(ps, None)])
    eye = np.array([0.3, 0.2, 2], np.float32)
    V, P = lookat(eye, [0, 0, 0], [0, 1, 0]), persp(60, 1.0, 0.1, 100)
    fr = ob.make_frame(V, P, eye, 64, 64)
    _, order = ob.sort_stable(*ob.key_cull(fr, inst))
    depth, ids, nrm = ob.render_surface(fr, inst, order[::-1].copy(), 0.7, normals=True)
    img, _ = ob.render(ob.make_frame(V, P, eye, 64, 64, front_to_back=1), inst)
    assert np.allclose(nrm[..., 3], img[..., 3], atol=1e-6) and nrm[..., 3].max() > 0.5
    n = ob.splat_normal(fr, inst, 0, 0, quantize=True)
    assert np.allclose(nrm[..., :3], nrm[..., 3:4] * n, atol=1e-6)
    assert abs(n[2]) > 0.99  # flat in z, seen from +z
    # many splats: |integrated normal| <= alpha <= 1, and the side outputs without normals are unchanged
    sc2 = synth.make_scene(1500, seed=11)
    inst2 = ob.make_instances([(ob.PreparedSet(sc2), None)])
    eye2 = np.array([3, 1, 2], np.float32)
    V2 = lookat(eye2, [0, 0, 0], [0, 1, 0])
    P2 = persp(60, 4 / 3, 0.1, 2000)
    fr2 = ob.make_frame(V2, P2, eye2, 128, 96)
    _, order2 = ob.sort_stable(*ob.key_cull(fr2, inst2))
    d2, i2, n2 = ob.render_surface(fr2, inst2, order2[::-1].copy(), 0.7, normals=True)
    d3, i3 = ob.render_surface(fr2, inst2, order2[::-1].copy(), 0.7)
    assert np.array_equal(d2, d3) and np.array_equal(i2, i3)
    assert np.all(np.linalg.norm(n2[..., :3], axis=-1) <= n2[..., 3] + 1e-5) and np.all(n2[..., 3] <= 1.0 + 1e-5)
    b, _ = ob.render(ob.make_frame(V2, P2, eye2, 128, 96, front_to_back=1), inst2)
    assert np.allclose(n2[..., 3], b[..., 3], atol=1e-5)


# ---- cross-check of the oracle against the independent float64 restatement (tests/np_reference.py) ----------
def _trs(scale, axis, angle, t):
    axis = np.asarray(axis, np.float64) / np.linalg.norm(axis)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    R = np.eye(3) + np.sin(angle) * K + (1 - np.cos(angle)) * (K @ K)
    M = np.eye(4)
    M[:3, :3] = R @ np.diag(scale)
    M[:3, 3] = t
    return M.astype(np.float32)


def _oracle_project_all(ob, fr, inst, k, n):
    out = dict(valid=np.zeros(n, bool), center_px=np.zeros((n, 2)), b1=np.zeros((n, 2)), b2=np.zeros((n, 2)),
               rgba=np.zeros((n, 4)), ndc_z=np.zeros(n))
    for i in range(n):
        p = ob.project(fr, inst, k, i)
        out["valid"][i] = bool(p.valid)
        if p.valid:
            out["center_px"][i] = list(p.center_px)
            out["b1"][i], out["b2"][i] = list(p.basis1), list(p.basis2)
            out["rgba"][i] = list(p.rgba)
            out["ndc_z"][i] = p.ndc_z
    return out


@pytest.mark.parametrize("case", ["identity", "trs", "msaa_flipy", "cull_at_raster_deg1"])
def test_projection_and_sh_against_independent_numpy_fp64(ob, case):
    """a12/a13/a11 per splat: view/clip centre, Sigma2D eigen basis (incl. the 0.1 floor and the |b|<1e-3 branch),
    SH radiance, opacity — oracle (fp32, column vectors) vs np_reference (fp64, the shaders' row-vector form)"""
    import np_reference as npr
    n = 3000
    sc = synth.make_scene(n, seed=31)
    ps = ob.PreparedSet(sc)
    M = None if case in ("identity", "msaa_flipy") else _trs([1.3, 0.7, 1.1], [0.3, 1.0, -0.2], 0.8, [0.4, -0.2, 0.3])
    inst = ob.make_instances([(ps, M)])
    W, H = 320, 200
    eye = np.array([3.5, 1.2, 0.8], np.float32)
    V, P = lookat(eye, [0, 0, 0], [0, 1, 0]), persp(55, W / H, 0.1, 2000, flip=(case == "msaa_flipy"))
    kw = dict(ms_antialiasing=1) if case == "msaa_flipy" else {}
    deg = 1 if case == "cull_at_raster_deg1" else 3
    fr = ob.make_frame(V, P, eye, W, H, sh_degree=deg, frustum_culling=2 if case == "cull_at_raster_deg1" else 1, **kw)
    o = _oracle_project_all(ob, fr, inst, 0, n)
    r = npr.project(ps.positions, ps.cov6, ps.rgba, ps.sh.reshape(n, -1), ps.sh_degree, np.eye(4) if M is None else M,
                    V, P, eye, W, H, sh_degree=deg, cull_at_raster=(case == "cull_at_raster_deg1"),
                    ms_aa=(case == "msaa_flipy"))
    # validity may flip only for splats sitting on a threshold in fp32
    flips = np.flatnonzero(o["valid"] != r["valid"])
    assert flips.size <= 3, flips
    both = o["valid"] & r["valid"]
    assert both.sum() > 500
    # a splat far behind/beside the camera can be 'valid' with a huge or ill-conditioned footprint: compare the
    # well-conditioned ones tightly and everything through the stable quantity (the 2x2 extent matrix)
    c_o, c_r = o["center_px"][both], r["center_px"][both]
    assert np.allclose(c_o, c_r, rtol=2e-5, atol=2e-3)
    assert np.allclose(o["ndc_z"][both], r["ndc_z"][both], rtol=0, atol=5e-5)  # fp32 cancellation in P22*z+P23 (near 0.1, far 2000)
    assert np.allclose(o["rgba"][both], r["rgba"][both], rtol=1e-4, atol=2e-5)  # base colour + SH + (AA) opacity
    Eo = np.einsum("ni,nj->nij", o["b1"][both], o["b1"][both]) + np.einsum("ni,nj->nij", o["b2"][both], o["b2"][both])
    Er = np.einsum("ni,nj->nij", r["b1"][both], r["b1"][both]) + np.einsum("ni,nj->nij", r["b2"][both], r["b2"][both])
    scale = np.abs(Er).max(axis=(1, 2))[:, None, None]
    assert (np.abs(Eo - Er) / scale).max() < 2e-3
    # direct basis comparison where the eigen decomposition is well conditioned (gap >> rounding, b clear of 1e-3)
    ev, cov2 = r["ev"][both], r["cov2"][both]
    good = ((ev[:, 0] - ev[:, 1]) > 0.05 * ev[:, 0]) & (np.abs(np.abs(cov2[:, 1]) - 1e-3) > 1e-4) & (ev[:, 0] < 1e5)
    assert good.sum() > 300
    for name in ("b1", "b2"):
        bo, br = o[name][both][good], r[name][both][good]
        assert np.allclose(bo, br, rtol=2e-3, atol=2e-3 * np.abs(br).max(axis=1, keepdims=True)), name


def test_extent_basis_special_cases_against_numpy(ob):
    """the branches an image PSNR would not notice: |b| < 1e-3 (eigenvector x := 1), the max(0.1, .) discriminant floor,
    the 2048-px clamp — isolated with hand-made splats"""
    import np_reference as npr
    # isotropic splat on the optical axis -> b == 0 exactly, discriminant 0 -> floored to 0.1
    # huge splat close to the camera -> sqrt8*sqrt(ev1) > 2048 -> clamped
    # flat disc seen obliquely -> strongly anisotropic, b far from 0
    pos = np.array([[0, 0, 0], [0.05, 0.02, 1.0], [0.3, -0.2, 0.1]], np.float32)
    scale = np.log(np.array([[0.05, 0.05, 0.05], [40.0, 30.0, 35.0], [0.4, 0.4, 0.004]], np.float32))
    rot = np.array([[1, 0, 0, 0], [1, 0, 0, 0], [0.9, 0.3, 0.2, 0.1]], np.float32)
    rot /= np.linalg.norm(rot, axis=1, keepdims=True)
    sc = dict(positions=pos, f_dc=np.zeros((3, 3), np.float32), f_rest=np.zeros((3, 0), np.float32),
              opacity=np.full(3, 3.0, np.float32), scale=scale, rotation=rot)
    ps = ob.PreparedSet(sc)
    inst = ob.make_instances([(ps, None)])
    eye = np.array([0, 0, 3], np.float32)
    W, H = 256, 256
    V, P = lookat(eye, [0, 0, 0], [0, 1, 0]), persp(60, 1.0, 0.1, 100)
    fr = ob.make_frame(V, P, eye, W, H)
    r = npr.project(ps.positions, ps.cov6, ps.rgba, np.zeros((3, 45)), 0, np.eye(4), V, P, eye, W, H)
    for i in range(3):
        p = ob.project(fr, inst, 0, i)
        assert bool(p.valid) and bool(r["valid"][i]), i
        for got, want in ((list(p.basis1), r["b1"][i]), (list(p.basis2), r["b2"][i])):
            # direction and length to 1e-3 of the vector's own length (the eigenvector angle of a 2048-px basis is
            # an fp32 quantity: the components are not individually meaningful to 1e-3 relative)
            assert np.allclose(got, want, rtol=0, atol=1e-3 * max(1.0, np.hypot(*want))), (i, got, want)
    assert abs(r["cov2"][0, 1]) < 1e-3 and np.isclose(r["ev"][0, 0] - r["ev"][0, 1], 2 * np.sqrt(0.1), rtol=1e-6)
    assert np.isclose(np.hypot(*r["b1"][1]), 2048.0)  # clamped
    assert abs(r["cov2"][2, 1]) > 1.0  # anisotropic, oblique


def test_oracle_frame_against_independent_numpy_render(ob):
    """whole small frame: oracle (fp32, unfused) vs np_reference (fp64) through key/cull order -> projection -> SH ->
    per-fragment alpha -> 'over' blend.  Two instances, one transformed."""
    import np_reference as npr
    n = 1500
    sc = synth.make_scene(n, seed=77)
    ps = ob.PreparedSet(sc)
    M1 = _trs([0.8, 1.2, 1.0], [0.1, 1.0, 0.3], -0.6, [1.0, 0.1, -0.5])
    inst = ob.make_instances([(ps, None), (ps, M1)])
    W, H = 160, 100
    eye = np.array([3.8, 1.4, 1.0], np.float32)
    V, P = lookat(eye, [0, 0, 0], [0, 1, 0]), persp(60, W / H, 0.1, 2000)
    fr = ob.make_frame(V, P, eye, W, H)  # fp32 target: no per-blend fp16 rounding on either side
    keys, ids = ob.key_cull(fr, inst)
    ks, vs = ob.sort_stable(keys, ids)
    oimg, st = ob.render(fr, inst, order=vs)
    pr = [npr.project(ps.positions, ps.cov6, ps.rgba, ps.sh.reshape(n, -1), ps.sh_degree, m, V, P, eye, W, H)
          for m in (np.eye(4), M1)]
    merged = {k: np.concatenate([pr[0][k], pr[1][k]]) for k in ("valid", "center_px", "b1", "b2", "rgba")}
    nimg = npr.render(merged, vs, W, H)
    assert st["fragments"] > 20000
    err = np.abs(oimg - nimg)
    # identical draw order and discards; the two sides differ by fp32-vs-fp64 rounding (fragments exactly on a
    # discard threshold are the only place a visible difference could come from)
    assert ob.psnr_rgb(oimg, nimg.astype(np.float32)) > 80.0
    assert err[..., :3].max() < 5e-3 and np.percentile(err, 99.9) < 1e-4


def test_gut_projection_and_response_against_independent_numpy_fp64(ob):
    """3DGUT: the oracle's unscented projection (7 sigma points, weights, covariance, conic extent) and its
    per-fragment particle response vs the float64 row-vector restatement in np_reference"""
    import np_reference as npr
    n = 2000
    sc = synth.make_scene(n, seed=17)
    ps = ob.PreparedSet(sc)
    M = _trs([1.2, 0.8, 1.0], [0.2, 1.0, 0.1], 0.5, [0.3, 0.1, -0.2])
    inst = ob.make_instances([(ps, M)])
    W, H = 320, 200
    eye = np.array([3.5, 1.2, 0.8], np.float32)
    V, P = lookat(eye, [0, 0, 0], [0, 1, 0]), persp(55, W / H, 0.1, 2000)
    fr = ob.make_frame(V, P, eye, W, H)
    g = npr.gut_project(ps.positions, sc["scale"], sc["rotation"], ps.rgba, M, V, P, W, H)
    got_valid = np.zeros(n, bool)
    checked = frag = 0
    rng = np.random.default_rng(3)
    for i in range(n):
        q = ob.project_gut(fr, inst, 0, i)
        got_valid[i] = bool(q.valid)
        if q.valid and g["valid"][i]:
            assert np.allclose(list(q.center_px), g["center_px"][i], rtol=1e-4, atol=5e-3), i
            assert np.isclose(q.half1[0], g["half_x"][i], rtol=2e-3, atol=2e-3) and np.isclose(q.half2[1], g["half_y"][i], rtol=2e-3, atol=2e-3), i
            checked += 1
            if checked % 7 == 0:  # a few fragments around the centre of every 7th splat
                for _ in range(4):
                    px = int(np.clip(q.center_px[0] + rng.integers(-3, 4), 0, W - 1))
                    py = int(np.clip(q.center_px[1] + rng.integers(-3, 4), 0, H - 1))
                    a = ob.gut_fragment(fr, inst, 0, q, px, py)
                    b = npr.gut_opacity(g, i, float(q.rgba[3]), M, V, P, W, H, px, py)
                    if a is None or b is None:
                        assert (a is None and (b is None or b < 6e-3)) or (b is None and a < 6e-3), (i, a, b)  # threshold fragments
                    else:
                        assert np.isclose(a, b, rtol=2e-3, atol=2e-4), (i, px, py, a, b)
                        frag += 1
    # the numpy side omits the z clip of the quad (ndc z in [0,1]) : it can only be MORE permissive
    assert (got_valid & ~g["valid"]).sum() <= 2 and checked > 500 and frag > 100


def test_gut_single_large_splat_matches_the_ewa_splat(ob):
    """a splat many pixels wide: the exact ray response (3DGUT) and the EWA footprint (3DGS) must nearly coincide — a
    sanity check that the two independent restatements describe the same Gaussian"""
    sc = dict(positions=np.array([[0.1, -0.05, 0]], np.float32), f_dc=np.array([[1.0, 0.0, -1.0]], np.float32),
              f_rest=np.zeros((1, 0), np.float32), opacity=np.array([2.0], np.float32),
              scale=np.log(np.array([[0.10, 0.05, 0.02]], np.float32)), rotation=np.array([[0.9, 0.1, 0.3, 0.2]], np.float32))
    ps = ob.PreparedSet(sc)
    inst = ob.make_instances([(ps, None)])
    eye = np.array([0.3, 0.2, 2], np.float32)
    V, P = lookat(eye, [0, 0, 0], [0, 1, 0]), persp(60, 1.0, 0.1, 100)
    fr = ob.make_frame(V, P, eye, 256, 256)
    a, _ = ob.render(fr, inst, order=np.array([0], np.uint32))
    g, st = ob.render_gut(fr, inst, np.array([0], np.uint32))
    assert st["quads"] == 1 and st["fragments"] > 1000
    # the reference adds 0.5 to SV_Position before generating the ray (frag.slang:105 with cameras.h.slang:37), so its 3DGUT
    # image sits half a pixel off the 3DGS one: a slope term on top of the (small) EWA-vs-exact difference
    assert np.abs(a[..., 3] - g[..., 3]).max() < 0.08 and abs(a[..., 3].max() - g[..., 3].max()) < 2e-3


def test_vulkan_reproducible_fixture_is_what_the_oracle_renders(ob):
    """tests/golden/vkrepro (scene.ply + repro.vkgs + camera.json + expected frames): a fixture the reference itself can
    consume on a Vulkan box.  Here: the committed expected frames are exactly what the oracle renders from the committed
    .ply through the product's own loader and the .vkgs project's camera."""
    import json
    from vk_gaussian_splatting_amd import project
    d = os.path.join(GOLDEN, "vkrepro")
    pr = project.load_project(os.path.join(d, "repro.vkgs"))
    cj = json.load(open(os.path.join(d, "camera.json")))
    W, H = cj["width"], cj["height"]
    V, P = pr.camera.matrices(W, H, False)
    assert np.allclose(np.asarray(V, np.float32).T.reshape(-1), cj["view_glm_column_major"], atol=1e-6)
    assert np.allclose(np.asarray(P, np.float32).T.reshape(-1), cj["proj_glm_column_major"], atol=1e-6)
    a = mgs.SplatSet.load(pr.splat_sets[0]).arrays()
    ps = ob.PreparedSet({k: a[k] for k in ("positions", "f_dc", "f_rest", "opacity", "scale", "rotation")})
    inst = ob.make_instances([(ps, None)])
    fr = ob.make_frame(V, P, pr.camera.eye, W, H, target_fp16=1)
    ks, vs = ob.sort_stable(*ob.key_cull(fr, inst))
    img, _ = ob.render(fr, inst, order=vs)
    assert np.array_equal(img.astype(np.float16), np.load(os.path.join(d, "expected_rgba16f.npy")))
    g, _ = ob.render_gut(fr, inst, vs)
    assert np.array_equal(g.astype(np.float16), np.load(os.path.join(d, "expected_3dgut_rgba16f.npy")))


def _vkrepro_generator():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_vkrepro", os.path.join(GOLDEN, "make_vkrepro.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


@pytest.mark.parametrize("case", ["u8_storage", "fisheye150_3dgut", "msaa_3dgs", "two_instances_trs"])
def test_vulkan_reproducible_cases_are_what_the_oracle_renders(ob, case):
    """tests/golden/vkrepro/<case>/ (round 4: uint8 storage — the reference's default —, fisheye 3DGUT at 150 degrees, Mip-Splatting
    antialiasing, two instances with translation / rotation / scale): the committed project file read back through the
    product's reader gives the camera of camera.json, and the committed expected frame is exactly what the oracle renders from it.
    (Expected = our oracle: unvalidated against the reference until a Vulkan box runs camera.json's reference_command.)"""
    import json
    from vk_gaussian_splatting_amd import project
    gen = _vkrepro_generator()
    d = os.path.join(GOLDEN, "vkrepro", case)
    pr = project.load_project(os.path.join(d, "repro.vkgs"))
    cj = json.load(open(os.path.join(d, "camera.json")))
    assert (cj["width"], cj["height"]) == (gen.CW, gen.CH) and pr.camera.model == cj["camera_model"]
    assert (pr.sh_format, pr.rgba_format) == (cj["frame"]["shFormat"], cj["frame"]["rgbaFormat"])
    assert os.path.samefile(pr.splat_sets[0], os.path.join(GOLDEN, "vkrepro", "scene.ply"))
    img, st, (V, P) = gen.render_case(case, pr)
    assert np.allclose(np.asarray(V, np.float32).T.reshape(-1), cj["view_glm_column_major"], atol=1e-6)
    assert np.allclose(np.asarray(P, np.float32).T.reshape(-1), cj["proj_glm_column_major"], atol=1e-6)
    want = np.load(os.path.join(d, "expected_rgba16f.npy"))
    assert np.array_equal(img, want) and st["fragments"] > 50_000 and float(want[..., 3].max()) > 0.5


# ---- stochastic paths: random numbers, stochastic splats, depth of field, temporal accumulation ----------------------------
def _np_xxhash32(x, y, z):
    M = 0xFFFFFFFF
    p0, p1, p2, p3 = 2246822519, 3266489917, 668265263, 374761393
    rot = lambda h: ((h << 17) | (h >> 15)) & M
    h = (z + p3 + x * p1) & M
    h = (p2 * rot(h)) & M
    h = (h + y * p1) & M
    h = (p2 * rot(h)) & M
    h = (p0 * (h ^ (h >> 15))) & M
    h = (p1 * (h ^ (h >> 13))) & M
    return h ^ (h >> 16)


def _np_rand(seed):
    M = 0xFFFFFFFF
    prev = (seed * 747796405 + 2891336453) & M
    word = (((prev >> ((prev >> 28) + 4)) ^ prev) * 277803737) & M
    r = (word >> 22) ^ word
    v = np.array([0x3F800000 | (r >> 9)], np.uint32).view(np.float32)[0] - np.float32(1.0)
    return float(v), prev


def test_xxhash32_restatement_against_the_published_xxhash_library(ob):
    """A pin from outside the three restatements: Jarzynski-Olano's xxhash32(uvec3) — what nvshaders/random.h.slang ships, absent
    from the reference tree — is the published XXH32 of the eight bytes (x, y) with the seed word taking z's place:
    XXH32 starts a short input at seed + PRIME32_5 + length, the shader variant at z + PRIME32_5, then both run the same two
    word rounds and the same avalanche — so xxhash32(x, y, z) == XXH32(le32(x) || le32(y), seed = z - 8 mod 2^32).  Checked
    against the `xxhash` wheel (Yann Collet's reference implementation behind it), including XXH32's own published vector."""
    import struct
    xxhash = pytest.importorskip("xxhash")
    assert xxhash.xxh32(b"", seed=0).intdigest() == 0x02CC5D05  # the library is the real thing (xxHash's documented test vector)
    rng = np.random.default_rng(11)
    for x, y, z in rng.integers(0, 2**32, (500, 3), dtype=np.uint64).tolist() + [[0, 0, 0], [1919, 1079, 199], [2**32 - 1] * 3, [5, 7, 3]]:
        want = xxhash.xxh32(struct.pack("<II", x, y), seed=(z - 8) & 0xFFFFFFFF).intdigest()
        assert ob.xxhash32(x, y, z) == want == _np_xxhash32(x, y, z)


def test_random_numbers_against_independent_python_restatement(ob):
    """nvshaders/random.h.slang is not in the reference tree (nvpro_core2): the oracle's C restatement is checked against a second,
    integer-only python restatement of the published functions, and for the properties the shaders rely on"""
    rng = np.random.default_rng(3)
    for x, y, z in rng.integers(0, 2**32, (200, 3), dtype=np.uint64).tolist() + [[0, 0, 0], [1919, 1079, 199]]:
        assert ob.xxhash32(x, y, z) == _np_xxhash32(x, y, z)
    vals = []
    for s in rng.integers(0, 2**32, 4000, dtype=np.uint64).tolist():
        v, nxt = ob.rand(s)
        v2, nxt2 = _np_rand(s)
        assert v == v2 and nxt == nxt2 and 0.0 <= v < 1.0
        vals.append(v)
    vals = np.array(vals)
    assert abs(vals.mean() - 0.5) < 0.02 and abs((vals < 0.25).mean() - 0.25) < 0.03
    # depthOfField draws twice from one state: the second draw differs from the first
    a, s1 = ob.rand(12345)
    b, _ = ob.rand(s1)
    assert a != b


def test_post_accumulate_is_the_running_mean(ob):
    rng = np.random.default_rng(5)
    frames = rng.random((9, 50), dtype=np.float32)
    main = np.zeros(50, np.float32)  # sample 0: lerp(main, aux1, 1) == aux1
    for k in range(9):
        main = ob.post_accumulate(main, frames[k], k)
        assert np.allclose(main, frames[:k + 1].mean(axis=0), atol=2e-6)


@pytest.mark.parametrize("gut", [False, True])
def test_stochastic_splats_converge_to_the_sorted_blend(ob, gut):
    """frag.slang:265-290 with depth test + write: the expected colour of the nearest accepted fragment is the sorted alpha
    blend; and the depth buffer makes the result independent of the draw order (that is the point of the mode)"""
    sc = synth.make_scene(900, seed=13)
    ps = ob.PreparedSet(sc)
    inst = ob.make_instances([(ps, None)])
    eye = np.array([2.5, 1.0, 2.0], np.float32)
    V, P = lookat(eye, [0, 0, 0], [0, 1, 0]), persp(60, 4 / 3, 0.1, 2000)
    Wd, Hd = 96, 72
    ok, oi = ob.key_cull(ob.make_frame(V, P, eye, Wd, Hd), inst)
    _, order = ob.sort_stable(ok, oi)
    rend = ob.render_gut if gut else ob.render
    blended, _ = rend(ob.make_frame(V, P, eye, Wd, Hd), inst, order)
    main = np.zeros_like(blended)
    psnr = {}
    for k in range(48):
        fr = ob.make_frame(V, P, eye, Wd, Hd, stochastic=1, frame_sample_id=k)
        img, st = rend(fr, inst, order)
        if k == 0:
            shuffled = order.copy()
            np.random.default_rng(1).shuffle(shuffled)
            img2, _ = rend(fr, inst, shuffled)
            # equal depths are resolved by draw order (LESS_OR_EQUAL): identical apart from such ties
            assert (np.abs(img - img2).max(axis=-1) > 0).mean() < 0.01
            assert set(np.unique(img[..., 3])) <= {0.0, 1.0}
        main = ob.post_accumulate(main, img, k)
        if k + 1 in (3, 12, 48):
            psnr[k + 1] = ob.psnr_rgb(main, blended)
    print("oracle stochastic mean vs blend:", psnr)
    assert psnr[12] > psnr[3] + 4.0 and psnr[48] > psnr[12] + 4.0 and psnr[48] > 28.0


def test_gut_depth_of_field_oracle_properties(ob):
    """cameras.h.slang:85-108: aperture 0 is the pinhole ray; the lens offset is bounded by sqrt(aperture); the same
    (pixel, sample) gives the same ray"""
    sc = synth.make_scene(600, seed=19)
    ps = ob.PreparedSet(sc)
    inst = ob.make_instances([(ps, None)])
    eye = np.array([2.5, 1.0, 2.0], np.float32)
    V, P = lookat(eye, [0, 0, 0], [0, 1, 0]), persp(60, 4 / 3, 0.1, 2000)
    Wd, Hd = 96, 72
    ok, oi = ob.key_cull(ob.make_frame(V, P, eye, Wd, Hd), inst)
    _, order = ob.sort_stable(ok, oi)
    sharp, _ = ob.render_gut(ob.make_frame(V, P, eye, Wd, Hd), inst, order)
    closed, _ = ob.render_gut(ob.make_frame(V, P, eye, Wd, Hd, dof_mode=1, focus_dist=3.0, aperture=0.0, frame_sample_id=4), inst, order)
    assert ob.psnr_rgb(sharp, closed) > 70.0
    a, _ = ob.render_gut(ob.make_frame(V, P, eye, Wd, Hd, dof_mode=1, focus_dist=3.0, aperture=0.05, frame_sample_id=4), inst, order)
    b, _ = ob.render_gut(ob.make_frame(V, P, eye, Wd, Hd, dof_mode=1, focus_dist=3.0, aperture=0.05, frame_sample_id=4), inst, order)
    c, _ = ob.render_gut(ob.make_frame(V, P, eye, Wd, Hd, dof_mode=1, focus_dist=3.0, aperture=0.05, frame_sample_id=5), inst, order)
    assert np.array_equal(a, b) and not np.array_equal(a, c)
    assert ob.psnr_rgb(sharp, a) < 40.0
    # the mean over many lens samples is a blurred image: lower gradient energy than the pinhole frame
    main = np.zeros_like(sharp)
    for k in range(24):
        img, _ = ob.render_gut(ob.make_frame(V, P, eye, Wd, Hd, dof_mode=1, focus_dist=3.0, aperture=0.05, frame_sample_id=k), inst, order)
        main = ob.post_accumulate(main, img, k)
    grad = lambda im: float(np.abs(np.diff(im[..., :3], axis=1)).mean())
    assert grad(main) < 0.9 * grad(sharp)


@pytest.mark.parametrize("degree", [0, 1, 3, 4, 5, 8])
def test_gut_kernel_degrees_against_the_closed_form(ob, degree):
    """particleRayMaxKernelResponse<KERNEL_DEGREE> (threedgrt.h.slang:83-127): the oracle's table of constants against the
    formula its comment states — exp(-4.5 / 3^n * d^n) of the distance d — in float64"""
    import np_reference as npr
    n = 600
    sc = synth.make_scene(n, seed=23)
    ps = ob.PreparedSet(sc)
    inst = ob.make_instances([(ps, None)])
    W, H = 256, 160
    eye = np.array([3.0, 1.0, 1.5], np.float32)
    V, P = lookat(eye, [0, 0, 0], [0, 1, 0]), persp(55, W / H, 0.1, 2000)
    fr = ob.make_frame(V, P, eye, W, H, kernel_degree=degree)
    g = npr.gut_project(ps.positions, sc["scale"], sc["rotation"], ps.rgba, np.eye(4), V, P, W, H)
    rng = np.random.default_rng(degree)
    frag = 0
    for i in range(n):
        q = ob.project_gut(fr, inst, 0, i)
        if not (q.valid and g["valid"][i]):
            continue
        for _ in range(2):
            px = int(np.clip(q.center_px[0] + rng.integers(-4, 5), 0, W - 1))
            py = int(np.clip(q.center_px[1] + rng.integers(-4, 5), 0, H - 1))
            a = ob.gut_fragment(fr, inst, 0, q, px, py)
            b = npr.gut_opacity(g, i, float(q.rgba[3]), np.eye(4), V, P, W, H, px, py, degree=degree)
            if a is None or b is None:
                assert (a is None and (b is None or b < 8e-3)) or (b is None and a < 8e-3), (i, a, b)
            else:
                assert np.isclose(a, b, rtol=3e-3, atol=3e-4), (i, px, py, a, b)
                frag += 1
    assert frag > 150


@pytest.mark.parametrize("thin", [1e-6, 0.03])
def test_gut_iso_surface_normal_against_the_quadric_gradient(ob, thin):
    """NORMAL_METHOD_ISO_SURFACE (threedgrt.h.slang:423-540): the oracle's shader-step restatement (canonical ray, sphere of radius
    3, normal / scale, rotate) against the gradient of the kernel ellipsoid's quadric at the entry point, float64
    (np_reference.gut_iso_normal); thin = 0.03 exercises the flat (axis normal) and degenerate (minus the ray) particles"""
    import np_reference as npr
    n = 500
    sc = synth.make_scene(n, seed=29)
    ps = ob.PreparedSet(sc)
    inst = ob.make_instances([(ps, None)])
    W, H = 256, 160
    eye = np.array([2.5, 1.2, -1.8], np.float32)
    V, P = lookat(eye, [0, 0, 0], [0, 1, 0]), persp(55, W / H, 0.1, 2000)
    fr = ob.make_frame(V, P, eye, W, H, normal_method=1)
    g = npr.gut_project(ps.positions, sc["scale"], sc["rotation"], ps.rgba, np.eye(4), V, P, W, H)
    rng = np.random.default_rng(5)
    frag, kinds, errs = 0, set(), []
    for i in range(n):
        q = ob.project_gut(fr, inst, 0, i)
        if not (q.valid and g["valid"][i]):
            continue
        s = g["scale"][i]
        kinds.add(int((s < max(0.02 * s.max(), thin)).sum()))
        for _ in range(3):
            px = int(np.clip(q.center_px[0] + rng.integers(-3, 4), 0, W - 1))
            py = int(np.clip(q.center_px[1] + rng.integers(-3, 4), 0, H - 1))
            a = ob.gut_fragment_iso(fr, inst, 0, q, px, py, thin)
            if a is None:
                continue
            op, nrm = a
            assert op == ob.gut_fragment(fr, inst, 0, q, px, py)  # the normal method does not touch the opacity
            want = npr.gut_iso_normal(g, i, V, P, W, H, px, py, thin)
            assert abs(np.linalg.norm(nrm) - 1.0) < 1e-5
            # fp32 conditioning: the canonical origin lies hundreds of radii from a small particle, so b^2 - 4ac of the sphere
            # intersection cancels (the shader has the same arithmetic)
            errs.append(np.abs(nrm - want).max())
            assert errs[-1] < 1e-2, (i, px, py, nrm, want)
            frag += 1
    print(f"iso-surface normal vs quadric gradient (thin {thin}): {frag} fragments, max {max(errs):.2e} mean {np.mean(errs):.2e}, small-axis counts {kinds}")
    assert frag > 150 and np.mean(errs) < 3e-4 and (thin < 1e-3 or {0, 1} <= kinds)


def test_gut_depth_of_field_against_independent_numpy_fp64(ob):
    """depthOfField (cameras.h.slang:85-108) in the oracle's 3DGUT fragment vs the float64 restatement, the lens sample drawn with
    the integer-only restatement of the random numbers: seed = xxhash32(px, py, sample), r1 = rand * 2 pi, r2 = rand * aperture"""
    import np_reference as npr
    n = 800
    sc = synth.make_scene(n, seed=29)
    ps = ob.PreparedSet(sc)
    inst = ob.make_instances([(ps, None)])
    W, H = 256, 160
    eye = np.array([3.0, 1.0, 1.5], np.float32)
    V, P = lookat(eye, [0, 0, 0], [0, 1, 0]), persp(55, W / H, 0.1, 2000)
    focus, aperture, sample = 2.5, 0.03, 11
    fr = ob.make_frame(V, P, eye, W, H, dof_mode=1, focus_dist=focus, aperture=aperture, frame_sample_id=sample)
    g = npr.gut_project(ps.positions, sc["scale"], sc["rotation"], ps.rgba, np.eye(4), V, P, W, H)
    rng = np.random.default_rng(4)
    frag = 0
    for i in range(n):
        q = ob.project_gut(fr, inst, 0, i)
        if not (q.valid and g["valid"][i]):
            continue
        px = int(np.clip(q.center_px[0] + rng.integers(-3, 4), 0, W - 1))
        py = int(np.clip(q.center_px[1] + rng.integers(-3, 4), 0, H - 1))
        seed = _np_xxhash32(px, py, sample)
        u1, seed = _np_rand(seed)
        u2, seed = _np_rand(seed)
        a = ob.gut_fragment(fr, inst, 0, q, px, py)
        b = npr.gut_opacity(g, i, float(q.rgba[3]), np.eye(4), V, P, W, H, px, py, lens=(u1 * 2.0 * np.pi, u2 * aperture, focus))
        if a is None or b is None:
            assert (a is None and (b is None or b < 8e-3)) or (b is None and a < 8e-3), (i, a, b)
        else:
            assert np.isclose(a, b, rtol=3e-3, atol=3e-4), (i, px, py, a, b)
            frag += 1
    assert frag > 60


# ---- CAMERA_FISHEYE branch of the dist stage (dist.comp.slang:75-90) --------------------------------------------------------
def test_deterministic_atan2_is_an_accurate_atan2(ob):
    """the fixed polynomial that stands in for the reference's implementation-defined atan2 in the fisheye cull: within 4e-7
    (3 ulp at pi/2) of the float64 arctangent over every quadrant the cull can reach (y = rho > 0), incl. x = 0, +-inf"""
    import ctypes as C
    L = ob.lib()
    L.orc_atan2_det.restype = C.c_float
    L.orc_atan2_det.argtypes = [C.c_float, C.c_float]
    rng = np.random.default_rng(5)
    ys = np.concatenate([10.0 ** rng.uniform(-7, 4, 4000), [1e-7, 1.0, 3.0]]).astype(np.float32)
    xs = np.concatenate([rng.standard_normal(2000) * 10.0 ** rng.uniform(-6, 4, 2000), 10.0 ** rng.uniform(-7, 4, 2000) * rng.choice([-1, 1], 2000),
                         [0.0, -0.0, 1.0]]).astype(np.float32)
    got = np.array([L.orc_atan2_det(float(y), float(x)) for y, x in zip(ys, xs)], np.float64)
    want = np.arctan2(ys.astype(np.float64), xs.astype(np.float64))
    assert np.abs(got - want).max() <= 4e-7, np.abs(got - want).max()
    assert abs(L.orc_atan2_det(1.0, float("inf")) - 0.0) <= 1e-7 and abs(L.orc_atan2_det(1.0, float("-inf")) - np.pi) <= 4e-7
    assert abs(L.orc_atan2_det(2.0, 0.0) - np.pi / 2) <= 2e-7
    assert np.isnan(L.orc_atan2_det(float("nan"), 1.0)) and np.isnan(L.orc_atan2_det(1.0, float("nan")))


@pytest.mark.parametrize("fov,gut,inside", [(60.0, 1, False), (100.0, 0, False), (150.0, 1, True), (170.0, 0, True), (175.0, 1, True)])
def test_fisheye_dist_cull_against_independent_numpy_fp64(ob, fov, gut, inside):
    """orc_key_cull with camera_model = fisheye vs np_reference.dist_cull (float64, row-vector form, numpy's arctan2): the same
    survivors except for splats within 1e-5 of a threshold; frameInfo.focal is the fisheye focal only on a 3DGUT pipeline; the
    set differs from the pinhole box's; non-finite centres are culled"""
    import np_reference as npr
    from vk_gaussian_splatting_amd import synth
    import vk_gaussian_splatting_amd as mgs
    sc = synth.make_scene(30000, seed=77)
    sc["positions"][:5] = [[np.nan, 0, 0], [0, np.inf, 0], [0, 0, -np.inf], [np.nan] * 3, [1e30, 1e30, 1e30]]
    W, H = 640, 360
    eye = np.array([0.3, 0.1, -0.2] if inside else [3.0, 1.2, 2.5], np.float32)
    ctr = [eye[0] + 0.2, eye[1], eye[2] + 1.0] if inside else [0, 0, 0]
    V, P = mgs.camera_lookat_perspective(eye, ctr, [0, 1, 0], fov, 0.1, 2000.0, W, H)
    M, _ = mgs.compute_transform([1.1, 0.9, 1.0], [5.0, -20.0, 12.0], [0.2, -0.1, 0.3])
    ps = ob.PreparedSet(sc)
    inst = ob.make_instances([(ps, M)])
    fr = ob.make_frame(V, P, eye, W, H, camera_model=1, pipeline_3dgut=gut)
    _, ids = ob.key_cull(fr, inst)
    fov_rad = 2.0 * np.arctan(1.0 / abs(float(np.float32(P[1][1]))))
    focal = [W / np.float32(fov_rad), -H / np.float32(fov_rad)] if gut else None
    with np.errstate(invalid="ignore", over="ignore"):
        ok, margin = npr.dist_cull(sc["positions"], M, V, P, W, H, 0.2, fisheye=True, focal=focal)
    got = np.zeros(ok.size, bool)
    got[ids] = True
    sure = np.nan_to_num(margin, nan=1.0) > 1e-5
    assert np.array_equal(got[sure], ok[sure]), int((got[sure] != ok[sure]).sum())
    assert not got[:5].any()                      # non-finite / absurd centres never survive the fisheye validity test
    assert (~sure).sum() < 30 and 100 < ids.size < ok.size
    pin_ok, _ = npr.dist_cull(sc["positions"], M, V, P, W, H, 0.2)
    print(f"fisheye cull fov {fov} gut {gut}: {ids.size} survive, pinhole box keeps {int(pin_ok.sum())}, differ on {int((pin_ok != got).sum())}")
    assert (pin_ok != got).sum() > 50
