"""A SECOND, independent restatement of the VK3DGSR per-splat and per-fragment math — float64 numpy, written
from the reference's Slang sources in the shaders' own ROW-VECTOR form (v' = mul(v, M) with M = glm memory read
row-major), i.e. along a different derivation path than oracle/mgs_oracle.cpp (column vectors, unfused fp32).
It exists to cross-check the oracle: a misreading shared by the oracle and the HIP kernels would show up here.

Test infrastructure only.  Follows:
  shaders/threedgs_raster.mesh.slang:162-289     fetch, alpha cull, view/clip centre, cull at raster, SH, quad
  shaders/threedgs.h.slang:26-121                covariance projection, extent basis
  shaders/threedgs_particle_storage.h.slang:103-159   SH radiance
  shaders/threedgs_raster.frag.slang:236-309     per-fragment alpha, discards, blend source
  src/gaussian_splatting.cpp:2066-2087           'over' blend state, cleared RGBA target
  shaders/dist.comp.slang:55-91                  dist-stage cull, both CAMERA_TYPE branches (dist_cull below)
"""
import numpy as np

SH_C1 = 0.4886025119029199
SH_C2 = [1.0925484, -1.0925484, 0.3153916, -1.0925484, 0.5462742]
SH_C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
         1.445305721320277, -0.5900435899266435]


def slang(m_math):
    """glm column-major memory of the math matrix, read by Slang as row-major  ==  transpose"""
    return np.asarray(m_math, np.float64).T


def sh_radiance(sh, degree, d):
    """sh: [n][15][3] ([coef][rgb]); d: [n][3] unit view directions"""
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    rgb = np.zeros((d.shape[0], 3))
    if degree >= 1:
        rgb += SH_C1 * (-sh[:, 0] * y + sh[:, 1] * z - sh[:, 2] * x)
    if degree >= 2:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        rgb += ((SH_C2[0] * xy) * sh[:, 3] + (SH_C2[1] * yz) * sh[:, 4] + (SH_C2[2] * (2.0 * zz - xx - yy)) * sh[:, 5]
                + (SH_C2[3] * xz) * sh[:, 6] + (SH_C2[4] * (xx - yy)) * sh[:, 7])
    if degree >= 3:
        rgb += (SH_C3[0] * sh[:, 8] * (3.0 * x * x - y * y) * y + SH_C3[1] * sh[:, 9] * x * y * z
                + SH_C3[2] * sh[:, 10] * (4.0 * z * z - x * x - y * y) * y
                + SH_C3[3] * sh[:, 11] * z * (2.0 * z * z - 3.0 * x * x - 3.0 * y * y)
                + SH_C3[4] * sh[:, 12] * x * (4.0 * z * z - x * x - y * y) + SH_C3[5] * sh[:, 13] * (x * x - y * y) * z
                + SH_C3[6] * sh[:, 14] * x * (x * x - 3.0 * y * y))
    return rgb


def project(centers, cov6, rgba, sh, set_degree, M, V, P, cam, W, H, splat_scale=1.0, alpha_cull=1.0 / 255.0,
            sh_degree=3, frustum_dilation=0.2, cull_at_raster=False, ms_aa=False):
    """every splat of one instance -> dict(valid, center_px[n,2], ndc_z, b1[n,2], b2[n,2], rgba[n,4], cov2[n,3])"""
    c = np.asarray(centers, np.float64).reshape(-1, 3)
    n = c.shape[0]
    col = np.asarray(rgba, np.float64).reshape(-1, 4).copy()
    valid = ~(col[:, 3] < alpha_cull)  # mesh.slang:164-170
    S_M, S_V, S_P = slang(M), slang(V), slang(P)
    mv = S_M @ S_V  # mul(desc.transform, frameInfo.viewMatrix), :175
    h = np.concatenate([c, np.ones((n, 1))], 1)
    view_c = h @ mv  # :178
    clip_c = view_c @ S_P  # :179
    if cull_at_raster:  # :181-190
        lim = (1.0 + frustum_dilation) * clip_c[:, 3]
        valid &= ~((np.abs(clip_c[:, 0]) > lim) | (np.abs(clip_c[:, 1]) > lim)
                   | (clip_c[:, 2] < (0.0 - frustum_dilation) * clip_c[:, 3]) | (clip_c[:, 2] > clip_c[:, 3]))
    # view-dependent colour, :240-243 (camera position through transformInverse, as a row vector)
    cam_model = (np.append(np.asarray(cam, np.float64), 1.0) @ np.linalg.inv(S_M))[:3]
    d = c - cam_model
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    deg = min(int(set_degree), int(sh_degree))
    if deg >= 1:
        col[:, :3] += sh_radiance(np.asarray(sh, np.float64).reshape(n, -1)[:, :45].reshape(n, 15, 3), deg, d)
    # covariance projection, threedgs.h.slang:26-56
    focal = np.array([P[0][0] * 0.5 * W, P[1][1] * 0.5 * H], np.float64)  # gaussian_splatting.cpp:1248-1250
    c6 = np.asarray(cov6, np.float64).reshape(-1, 6)
    cov3 = np.zeros((n, 3, 3))
    cov3[:, 0, 0], cov3[:, 0, 1], cov3[:, 0, 2] = c6[:, 0], c6[:, 1], c6[:, 2]
    cov3[:, 1, 0], cov3[:, 1, 1], cov3[:, 1, 2] = c6[:, 1], c6[:, 3], c6[:, 4]
    cov3[:, 2, 0], cov3[:, 2, 1], cov3[:, 2, 2] = c6[:, 2], c6[:, 4], c6[:, 5]
    zx, zy, zz = view_c[:, 0], view_c[:, 1], view_c[:, 2]
    s = 1.0 / (zz * zz)
    J = np.zeros((n, 3, 3))
    J[:, 0, 0] = focal[0] / zz
    J[:, 0, 2] = -(focal[0] * zx) * s
    J[:, 1, 1] = focal[1] / zz
    J[:, 1, 2] = -(focal[1] * zy) * s
    Wm = mv[:3, :3].T  # transpose(float3x3(modelViewTransform))
    T = J @ Wm  # mul(J, W)
    cov2 = T @ cov3 @ np.transpose(T, (0, 2, 1))
    a, b, dd = cov2[:, 0, 0].copy(), cov2[:, 0, 1].copy(), cov2[:, 1, 1].copy()
    # extent basis, threedgs.h.slang:60-121
    if ms_aa:
        det_orig = a * dd - b * b
    a += 0.3
    dd += 0.3
    if ms_aa:
        col[:, 3] *= np.sqrt(np.maximum(det_orig / (a * dd - b * b), 0.0))
    D = a * dd - b * b
    half = 0.5 * (a + dd)
    term2 = np.sqrt(np.maximum(0.1, half * half - D))
    ev1, ev2 = half + term2, half - term2
    valid &= ~(ev2 <= 0.0)
    e1 = np.stack([np.where(np.abs(b) < 0.001, 1.0, b), ev1 - a], 1)
    e1 /= np.linalg.norm(e1, axis=1, keepdims=True)
    e2 = np.stack([e1[:, 1], -e1[:, 0]], 1)
    sq8 = np.sqrt(8.0)
    with np.errstate(invalid="ignore"):
        b1 = e1 * (splat_scale * np.minimum(sq8 * np.sqrt(ev1), 2048.0))[:, None]
        b2 = e2 * (splat_scale * np.minimum(sq8 * np.sqrt(np.maximum(ev2, 0.0)), 2048.0))[:, None]
    ndc = clip_c[:, :3] / clip_c[:, 3:4]
    # the emitted quad sits at z = ndc.z, w = 1: Vulkan's fixed-function clip (0 <= z <= w, no depth clamp) removes it
    # entirely outside [0,1] — the mechanism emitDegeneratedQuad (z = 2) relies on, mesh.slang:102-109
    valid &= (ndc[:, 2] >= 0.0) & (ndc[:, 2] <= 1.0)
    # quad vertex = ndcCenter.xy + (fx*b1 + fy*b2) * basisViewport * 2 (:279-286), basisViewport = 1/viewport:
    # in pixels that is centre_px + fx*b1 + fy*b2 with the viewport transform (ndc+1)/2 * size
    center_px = np.stack([(ndc[:, 0] + 1.0) * 0.5 * W, (ndc[:, 1] + 1.0) * 0.5 * H], 1)
    return dict(valid=valid, center_px=center_px, ndc_z=ndc[:, 2], b1=b1, b2=b2, rgba=col,
                cov2=np.stack([a, b, dd], 1), ev=np.stack([ev1, ev2], 1))


def render(proj, order, W, H):
    """back-to-front 'over' (Src*As + Dst*(1-As); A = As + Ad) of the valid splats of `proj` in draw order `order`,
    float64, cleared target.  Fragment = pixel centre inside the quad; fragPos = sqrt8 * (u, v)."""
    img = np.zeros((H, W, 4))
    for i in order:
        if not proj["valid"][i]:
            continue
        c, b1, b2, col = proj["center_px"][i], proj["b1"][i], proj["b2"][i], proj["rgba"][i]
        ex, ey = abs(b1[0]) + abs(b2[0]), abs(b1[1]) + abs(b2[1])
        x0, x1 = int(max(0, np.floor(c[0] - ex - 0.5))), int(min(W - 1, np.ceil(c[0] + ex - 0.5)))
        y0, y1 = int(max(0, np.floor(c[1] - ey - 0.5))), int(min(H - 1, np.ceil(c[1] + ey - 0.5)))
        if x1 < x0 or y1 < y0:
            continue
        yy, xx = np.mgrid[y0:y1 + 1, x0:x1 + 1]
        dx, dy = xx + 0.5 - c[0], yy + 0.5 - c[1]
        u = (dx * b1[0] + dy * b1[1]) / (b1 @ b1)  # pixel = centre + u*b1 + v*b2, b1 ⟂ b2
        v = (dx * b2[0] + dy * b2[1]) / (b2 @ b2)
        A = 8.0 * (u * u + v * v)  # dot(fragPos, fragPos), frag.slang:236
        op = np.exp(-0.5 * A) * col[3]
        keep = (A <= 8.0) & (op > 1.0 / 255.0)  # :242-262
        op = np.where(keep, op, 0.0)[..., None]
        dst = img[y0:y1 + 1, x0:x1 + 1]
        dst[..., :3] = col[:3] * op + dst[..., :3] * (1.0 - op)
        dst[..., 3:4] = op + dst[..., 3:4]
    return img


# ---- 3DGUT (threedgut_raster.{mesh,frag}.slang), float64, the shaders' row-vector form ---------------------------------
def gut_project(centers, scales_log, rotations_wxyz, rgba, M, V, P, W, H, extent_conic=True, alpha_cull=1.0 / 255.0):
    """unscented projection of every splat (perfect pinhole, global shutter): dict(valid, center_px, cov, half_x, half_y,
    axes[n,3,3] (principal axes as rows), scale[n,3]).  SH is not repeated here (same function as 3DGS)."""
    c = np.asarray(centers, np.float64).reshape(-1, 3)
    n = c.shape[0]
    s = np.exp(np.asarray(scales_log, np.float64).reshape(-1, 3))
    q = np.asarray(rotations_wxyz, np.float64).reshape(-1, 4)
    q = q / np.linalg.norm(q, axis=1, keepdims=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    # quatToMat3 (quaternions.h.slang:39-58): the Slang rows; row i = i-th principal axis of the ellipsoid
    R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y + w * z), 2 * (x * z - w * y)], 1),
                  np.stack([2 * (x * y - w * z), 1 - 2 * (x * x + z * z), 2 * (y * z + w * x)], 1),
                  np.stack([2 * (x * z + w * y), 2 * (y * z - w * x), 1 - 2 * (x * x + y * y)], 1)], 1)  # [n, row, col]
    S_M, S_V = slang(M), slang(V)
    focal = np.array([P[0][0] * 0.5 * W, P[1][1] * 0.5 * H], np.float64)
    pp = np.array([W / 2.0, H / 2.0])

    def proj(pts):  # model-space points [n,3] -> (pixels [n,2], valid)
        h = np.concatenate([pts, np.ones((pts.shape[0], 1))], 1)
        v = (h @ S_M) @ S_V
        cam = np.stack([v[:, 0], v[:, 1], -v[:, 2]], 1)  # RUB -> RUF
        ok = cam[:, 2] > 0
        with np.errstate(divide="ignore", invalid="ignore"):
            uv = np.where(ok[:, None], cam[:, :2] / cam[:, 2:3], 0.0)
        px = np.where(ok[:, None], uv * focal + pp, 0.0)
        inside = (px[:, 0] > -0.1 * W) & (px[:, 1] > -0.1 * H) & (px[:, 0] < 1.1 * W) & (px[:, 1] < 1.1 * H)
        return px, ok & inside

    delta = np.sqrt(3.0)
    pts = [c] + [c + sgn * delta * s[:, i:i + 1] * R[:, i, :] for i in range(3) for sgn in (+1, -1)]
    pr = [proj(p) for p in pts]
    nvalid = sum(ok.astype(int) for _, ok in pr)
    wI = 1.0 / 6.0
    center = sum(wI * px for px, _ in pr[1:])  # weight of the mean is lambda / (D + lambda) = 0
    d0 = pr[0][0] - center
    cov = 2.0 * np.stack([d0[:, 0] ** 2, d0[:, 0] * d0[:, 1], d0[:, 1] ** 2], 1)  # weight0 = 0 + (1 - 1 + 2)
    for px, _ in pr[1:]:
        d = px - center
        cov += wI * np.stack([d[:, 0] ** 2, d[:, 0] * d[:, 1], d[:, 1] ** 2], 1)
    a = np.asarray(rgba, np.float64).reshape(-1, 4)[:, 3]
    valid = (nvalid > 0) & ~(a < alpha_cull)
    dx, dy, dz = cov[:, 0] + 0.3, cov[:, 1], cov[:, 2] + 0.3
    det = dx * dz - dy * dy
    valid &= (det != 0) & ~(a < 0.01)
    with np.errstate(invalid="ignore", divide="ignore"):
        factor = np.minimum(3.33, np.sqrt(2.0 * np.log(np.maximum(a, 1e-30) / 0.01)))
        mid = 0.5 * (dx + dz)
        radius = factor * np.sqrt(mid + np.sqrt(np.maximum(0.01, mid * mid - det)))
        hx, hy = np.minimum(factor * np.sqrt(dx), radius), np.minimum(factor * np.sqrt(dz), radius)
    return dict(valid=valid, center_px=center, cov=cov, half_x=hx, half_y=hy, axes=R, scale=s, position=c)


def gut_opacity(g, i, density, M, V, P, W, H, px, py, alpha_clamp=0.99, min_response=0.0113, degree=2, lens=None):
    """opacity of splat i at pixel (px, py) or None: ray of the fragment (SV_Position + 0.5, as the reference writes it),
    canonical ray, quadratic kernel"""
    S_V, S_P, S_M = slang(V), slang(P), slang(M)
    vi, pi, mi = np.linalg.inv(S_V), np.linalg.inv(S_P), np.linalg.inv(S_M)
    origin = (np.array([0, 0, 0, 1.0]) @ vi)[:3]
    inuv = (np.array([px + 0.5, py + 0.5]) + 0.5) / np.array([W, H])
    d = inuv * 2.0 - 1.0
    target = np.array([d[0], d[1], 1.0, 1.0]) @ pi
    rd = (np.array([target[0], target[1], target[2], 0.0]) @ vi)[:3]
    rd /= np.linalg.norm(rd)
    if lens is not None:
        # depthOfField (cameras.h.slang:85-108): lens = (r1, r2, focus distance) with r1 = rand * 2 pi, r2 = rand * aperture
        r1, r2, focus = lens
        right = (np.array([1.0, 0, 0, 0]) @ vi)[:3]
        up = (np.array([0, 1.0, 0, 0]) @ vi)[:3]
        ap = (np.cos(r1) * right + np.sin(r1) * up) * np.sqrt(r2)
        nd = rd * focus - ap
        origin = origin + ap
        rd = nd / np.linalg.norm(nd)
    mo = (np.append(origin, 1.0) @ mi)[:3]
    md = rd @ mi[:3, :3]
    md /= np.linalg.norm(md)
    A = g["axes"][i]  # rows = principal axes; mul(v, invRotation) = v @ A.T
    ro = ((mo - g["position"][i]) @ A.T) / g["scale"][i]
    r = (md @ A.T) / g["scale"][i]
    r /= np.linalg.norm(r)
    cr = np.cross(r, ro)
    d2 = cr @ cr
    if degree == 0:      # generalised Gaussian of degree n in the distance: exp(-4.5 / 3^n * dist^n); n = 0 is the linear kernel
        resp = max(1.0 - 0.329630334487 * np.sqrt(d2), 0.0)
    else:
        resp = np.exp(-4.5 / 3.0 ** degree * np.sqrt(d2) ** degree)
    alpha = min(alpha_clamp, resp * density)
    return alpha if (alpha > 1.0 / 255.0 and resp > min_response) else None


# ---- dist stage: which splats survive (dist.comp.slang:55-91), float64, the shader's row-vector form ------------------
def gut_iso_normal(g, i, V, P, W, H, px, py, thin=1e-6):
    """NORMAL_METHOD_ISO_SURFACE for splat i at pixel (px, py), identity instance transform, float64, written from the geometry
    rather than from the shader's steps: the kernel ellipsoid is the quadric (x - c)^T Q (x - c) = 9 with Q = A^T diag(1/s^2) A;
    the normal where the pixel's ray enters it is the quadric's gradient Q (x - c).  One small axis (< max(0.02 max scale, thin)):
    that axis, towards the camera; two or three: minus the ray."""
    S_V, S_P = slang(V), slang(P)
    vi, pi = np.linalg.inv(S_V), np.linalg.inv(S_P)
    origin = (np.array([0, 0, 0, 1.0]) @ vi)[:3]
    inuv = (np.array([px + 0.5, py + 0.5]) + 0.5) / np.array([W, H])
    d = inuv * 2.0 - 1.0
    target = np.array([d[0], d[1], 1.0, 1.0]) @ pi
    rd = (np.array([target[0], target[1], target[2], 0.0]) @ vi)[:3]
    rd /= np.linalg.norm(rd)
    A, sc, c = g["axes"][i].astype(np.float64), g["scale"][i].astype(np.float64), g["position"][i].astype(np.float64)
    small = sc < max(0.02 * sc.max(), thin)
    if small.sum() >= 2:
        return -rd
    if small.sum() == 1:
        n = A[int(np.argmax(small))]
        return n if n @ (origin - c) >= 0 else -n
    Q = A.T @ np.diag(1.0 / sc**2) @ A
    o = origin - c
    qa, qb, qc = rd @ Q @ rd, 2.0 * (rd @ Q @ o), o @ Q @ o - 9.0
    disc = qb * qb - 4 * qa * qc
    if disc < 0:
        return -rd
    t1, t2 = (-qb - np.sqrt(disc)) / (2 * qa), (-qb + np.sqrt(disc)) / (2 * qa)
    t = t1 if t1 >= 0 else (t2 if t2 >= 0 else None)
    if t is None:
        return -rd
    n = Q @ (o + t * rd)
    return n / np.linalg.norm(n)


def dist_cull(centers, M, V, P, W, H, dilation=0.2, fisheye=False, focal=None):
    """returns (survives[n] bool, margin[n]): margin = distance of the decision from its nearest threshold, relative, so that a
    caller comparing against an fp32 implementation can set borderline splats aside.
    pinhole: :65-73.  fisheye: :75-90 = initPerfectFisheyeCamera(viewport, frameInfo.focal) (threedgut_camera_models.h.slang:
    87-136) + projectPointFisheye (threedgut_camera_projections.h.slang:149-171) on (1,1,-1) * viewPos, tolerance 0.1, then the
    z test.  `focal` = frameInfo.focal (gaussian_splatting.cpp:1239-1251); default: the pinhole focal."""
    c = np.asarray(centers, np.float64).reshape(-1, 3)
    h = np.concatenate([c, np.ones((c.shape[0], 1))], 1)
    view = (h @ slang(M)) @ slang(V)        # mul(mul(splatPos, transform), viewMatrix)   :58
    clip = view @ slang(P)                  #                                             :60
    with np.errstate(divide="ignore", invalid="ignore"):
        ndc = clip / clip[:, 3:4]           #                                             :61
    zin = ~((ndc[:, 2] < 0.0 - dilation) | (ndc[:, 2] > 1.0))
    zmargin = np.minimum(np.abs(ndc[:, 2] + dilation), np.abs(ndc[:, 2] - 1.0))
    if not fisheye:
        lim = 1.0 + dilation
        ok = ~((np.abs(ndc[:, 0]) > lim) | (np.abs(ndc[:, 1]) > lim)) & zin
        margin = np.minimum(np.minimum(np.abs(np.abs(ndc[:, 0]) - lim), np.abs(np.abs(ndc[:, 1]) - lim)), zmargin)
        return ok, margin
    f = np.asarray(focal if focal is not None else [P[0][0] * 0.5 * W, P[1][1] * 0.5 * H], np.float64)
    res = np.array([W, H], np.float64)
    pp = res / 2.0
    # computeMaxAngle: max distance from the principal point to a border per axis (centre: half the size), radius = length
    max_r = np.linalg.norm(np.where(pp > 0.5 * res, pp, res - pp))
    max_angle = max(2.0 * max_r / f[0], 2.0 * max_r / f[1]) / 2.0
    pos = view[:, :3] * np.array([1.0, 1.0, -1.0])
    rho = np.maximum(np.hypot(pos[:, 0], pos[:, 1]), 1e-7)
    theta_full = np.arctan2(rho, pos[:, 2])
    theta = np.minimum(theta_full, max_angle)
    delta = theta / rho                     # radial coefficients are all zero
    px = f * pos[:, :2] * delta[:, None] + pp
    tol = res * 0.1
    within = (px[:, 0] > -tol[0]) & (px[:, 1] > -tol[1]) & (px[:, 0] < res[0] + tol[0]) & (px[:, 1] < res[1] + tol[1])
    ok = (theta < max_angle) & within & zin
    edge = np.minimum(np.minimum(np.abs(px[:, 0] + tol[0]), np.abs(px[:, 0] - res[0] - tol[0])) / res[0],
                      np.minimum(np.abs(px[:, 1] + tol[1]), np.abs(px[:, 1] - res[1] - tol[1])) / res[1])
    margin = np.minimum(np.minimum(np.abs(theta_full - max_angle), edge), zmargin)
    return ok, margin
