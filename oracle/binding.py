"""ctypes binding of oracle/liboracle.so and oracle/_ref/libref_ingest.so — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
F32P = C.POINTER(C.c_float)
U32P = C.POINTER(C.c_uint32)


class OrcFrame(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("proj", C.c_float * 16), ("camera_pos", C.c_float * 3),
                ("width", C.c_int), ("height", C.c_int), ("splat_scale", C.c_float), ("frustum_dilation", C.c_float),
                ("alpha_cull_threshold", C.c_float), ("sh_degree", C.c_int), ("front_to_back", C.c_int),
                ("frustum_culling", C.c_int), ("target_fp16", C.c_int), ("ms_antialiasing", C.c_int),
                ("size_culling", C.c_int), ("size_culling_min_pixels", C.c_float), ("debug_flags", C.c_int),
                ("camera_model", C.c_int), ("extent_method", C.c_int), ("fov_rad", C.c_float), ("alpha_clamp", C.c_float),
                ("kernel_min_response", C.c_float), ("stochastic", C.c_int), ("dof_mode", C.c_int), ("focus_dist", C.c_float),
                ("aperture", C.c_float), ("frame_sample_id", C.c_int), ("kernel_degree", C.c_int), ("pipeline_3dgut", C.c_int),
                ("normal_method", C.c_int)]


class OrcInstance(C.Structure):
    _fields_ = [("centers", F32P), ("cov6", F32P), ("rgba", F32P), ("sh", F32P), ("scales", F32P), ("count", C.c_uint32),
                ("sh_degree", C.c_int), ("sh_stride", C.c_int), ("transform", C.c_float * 16),
                ("transform_inv", C.c_float * 16), ("rotations", F32P)]


class OrcProjected(C.Structure):
    _fields_ = [("valid", C.c_int), ("center_px", C.c_float * 2), ("ndc_z", C.c_float), ("basis1", C.c_float * 2),
                ("basis2", C.c_float * 2), ("rgba", C.c_float * 4), ("opacity_disabled", C.c_int)]


class OrcGutProjected(C.Structure):
    _fields_ = [("valid", C.c_int), ("center_px", C.c_float * 2), ("ndc_z", C.c_float), ("half1", C.c_float * 2),
                ("half2", C.c_float * 2), ("rgba", C.c_float * 4), ("position", C.c_float * 3), ("scale", C.c_float * 3),
                ("inv_rot", C.c_float * 9)]


class OrcSortInstance(C.Structure):
    _fields_ = [("positions", F32P), ("count", C.c_uint32), ("global_offset", C.c_uint32),
                ("transform", C.c_float * 16)]


def build(force=False):
    """compile the oracle (and, when /root/reference exists, oracle/_ref) via oracle/Makefile"""
    so = os.path.join(HERE, "liboracle.so")
    src_newer = (not os.path.exists(so)) or any(
        os.path.getmtime(os.path.join(HERE, f)) > os.path.getmtime(so) for f in ("mgs_oracle.cpp", "mgs_oracle.h"))
    if force or src_newer:
        subprocess.run(["make", "-C", HERE, "liboracle.so"], check=True, capture_output=True)
    ref = os.path.join(HERE, "_ref", "libref_ingest.so")
    if os.path.isdir("/root/reference/src") and (force or not os.path.exists(ref)):
        subprocess.run(["make", "-C", HERE, "ref"], check=True, capture_output=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(os.path.join(HERE, "liboracle.so"))
        L.orc_max_sh_degree.restype = C.c_int
        L.orc_max_sh_degree.argtypes = [C.c_size_t, C.c_size_t]
        L.orc_key_cull.restype = C.c_uint32
        L.orc_key_cull.argtypes = [C.POINTER(OrcFrame), C.POINTER(OrcInstance), C.c_int, U32P, U32P]
        L.orc_sort_stable.argtypes = [U32P, U32P, C.c_uint32]
        L.orc_project.argtypes = [C.POINTER(OrcFrame), C.POINTER(OrcInstance), C.c_uint32, C.POINTER(OrcProjected)]
        L.orc_render.restype = C.c_uint64
        L.orc_render.argtypes = [C.POINTER(OrcFrame), C.POINTER(OrcInstance), C.c_int, F32P, C.POINTER(C.c_uint64)]
        L.orc_render_order.restype = C.c_uint64
        L.orc_render_order.argtypes = [C.POINTER(OrcFrame), C.POINTER(OrcInstance), C.c_int, U32P, C.c_uint32, F32P,
                                       C.POINTER(C.c_uint64)]
        L.orc_psnr_rgb.restype = C.c_double
        L.orc_psnr_rgb.argtypes = [F32P, F32P, C.c_int, C.c_int]
        L.orc_cpu_sort.restype = C.c_int
        L.orc_cpu_sort.argtypes = [F32P, F32P, C.POINTER(OrcSortInstance), C.c_int, C.c_uint32, C.c_int, C.c_int,
                                   F32P, U32P, C.POINTER(C.c_double), C.POINTER(C.c_double)]
        L.orc_cov3d.argtypes = [F32P, F32P, C.c_size_t, F32P]
        L.orc_rgba.argtypes = [F32P, F32P, C.c_size_t, F32P]
        L.orc_sh_interleave.argtypes = [F32P, C.c_size_t, C.c_int, F32P]
        L.orc_sh_stride.restype = C.c_int
        L.orc_sh_stride.argtypes = [C.c_int]
        L.orc_quantize_roundtrip.argtypes = [F32P, C.c_size_t, C.c_int, C.c_int]
        L.orc_convert_rdf_to_rub.argtypes = [F32P, F32P, F32P, C.c_size_t, C.c_size_t]
        L.orc_flip_sh_rdf_to_rub.argtypes = [F32P]
        L.orc_mat4_inverse.argtypes = [F32P, F32P]
        L.orc_mat4_mul.argtypes = [F32P, F32P, F32P]
        L.orc_encode_key.restype = C.c_uint32
        L.orc_encode_key.argtypes = [C.c_float]
        L.orc_float_to_half.restype = C.c_uint16
        L.orc_float_to_half.argtypes = [C.c_float]
        L.orc_half_to_float.restype = C.c_float
        L.orc_half_to_float.argtypes = [C.c_uint16]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(F32P)


def _u(a):
    return a.ctypes.data_as(U32P)


def f32(a):
    return np.ascontiguousarray(a, np.float32)


class PreparedSet:
    """the device-side buffers of a splat set as the shaders read them (upload transform + dequant)"""

    def __init__(self, arrays, sh_format=0, rgba_format=0):
        L = lib()
        self.positions = f32(arrays["positions"]).reshape(-1)
        n = self.positions.size // 3
        self.count = n
        scale, rot = f32(arrays["scale"]).reshape(-1), f32(arrays["rotation"]).reshape(-1)
        self.scales = scale
        self.rotations = rot
        f_dc, op = f32(arrays["f_dc"]).reshape(-1), f32(arrays["opacity"]).reshape(-1)
        f_rest = f32(arrays["f_rest"]).reshape(-1) if arrays.get("f_rest") is not None else np.zeros(0, np.float32)
        self.cov6 = np.zeros(6 * n, np.float32)
        L.orc_cov3d(_p(scale), _p(rot), n, _p(self.cov6))
        self.rgba = np.zeros(4 * n, np.float32)
        L.orc_rgba(_p(f_dc), _p(op), n, _p(self.rgba))
        L.orc_quantize_roundtrip(_p(self.rgba), self.rgba.size, rgba_format, 0)
        per = f_rest.size // n if n else 0
        cpc = per // 3
        self.sh_degree = max(0, L.orc_max_sh_degree(f_rest.size, n))
        self.sh_stride = L.orc_sh_stride(cpc)
        self.sh = np.zeros(max(1, self.sh_stride * n), np.float32)
        if self.sh_stride:
            L.orc_sh_interleave(_p(f_rest), n, cpc, _p(self.sh))
            L.orc_quantize_roundtrip(_p(self.sh), self.sh_stride * n, sh_format, 1)


def make_instances(prepared_and_transforms):
    """[(PreparedSet, 4x4 math-convention matrix or None)] -> ctypes array of OrcInstance"""
    L = lib()
    arr = (OrcInstance * len(prepared_and_transforms))()
    keep = []
    for i, (ps, m) in enumerate(prepared_and_transforms):
        mm = np.eye(4, dtype=np.float32) if m is None else f32(m)
        col = np.ascontiguousarray(mm.T.reshape(-1))
        inv = np.zeros(16, np.float32)
        L.orc_mat4_inverse(_p(col), _p(inv))
        I = arr[i]
        I.centers, I.cov6, I.rgba, I.sh = _p(ps.positions), _p(ps.cov6), _p(ps.rgba), _p(ps.sh)
        I.scales = _p(ps.scales)
        I.rotations = _p(ps.rotations)
        I.count, I.sh_degree, I.sh_stride = ps.count, ps.sh_degree, ps.sh_stride
        for k in range(16):
            I.transform[k] = float(col[k])
            I.transform_inv[k] = float(inv[k])
        keep.append((ps, col, inv))
    arr._keep = keep
    return arr


def make_frame(view, proj, camera_pos, width, height, splat_scale=1.0, frustum_dilation=0.2,
               alpha_cull=1.0 / 255.0, sh_degree=3, front_to_back=0, frustum_culling=1, target_fp16=0,
               ms_antialiasing=0, debug_flags=0, size_culling=0, size_culling_min_pixels=1.0,
               camera_model=0, extent_method=1, fov_rad=None, alpha_clamp=0.99, kernel_min_response=0.0113,
               stochastic=0, dof_mode=0, focus_dist=1.3, aperture=0.001, frame_sample_id=0, kernel_degree=2, pipeline_3dgut=0,
               normal_method=0):
    f = OrcFrame()
    v = f32(view).T.reshape(-1)
    p = f32(proj).T.reshape(-1)
    for i in range(16):
        f.view[i] = float(v[i])
        f.proj[i] = float(p[i])
    for i in range(3):
        f.camera_pos[i] = float(camera_pos[i])
    f.width, f.height = width, height
    f.splat_scale, f.frustum_dilation, f.alpha_cull_threshold = splat_scale, frustum_dilation, alpha_cull
    f.sh_degree, f.front_to_back, f.frustum_culling = sh_degree, front_to_back, frustum_culling
    f.target_fp16, f.ms_antialiasing, f.debug_flags = target_fp16, ms_antialiasing, debug_flags
    f.size_culling, f.size_culling_min_pixels = size_culling, size_culling_min_pixels
    f.camera_model, f.extent_method = camera_model, extent_method
    # fovRad of the perspective matrix (what cameraManip->getRadFov() returns for the camera that produced proj)
    f.fov_rad = float(fov_rad) if fov_rad is not None else float(2.0 * np.arctan(1.0 / abs(float(p[5]))))
    f.alpha_clamp, f.kernel_min_response = alpha_clamp, kernel_min_response
    f.kernel_degree = kernel_degree
    f.pipeline_3dgut = pipeline_3dgut
    f.normal_method = normal_method
    f.stochastic, f.dof_mode, f.focus_dist, f.aperture, f.frame_sample_id = stochastic, dof_mode, focus_dist, aperture, frame_sample_id
    return f


def xxhash32(x, y, z):
    L = lib()
    L.orc_xxhash32.restype = C.c_uint32
    return int(L.orc_xxhash32(C.c_uint32(x), C.c_uint32(y), C.c_uint32(z)))


def rand(seed):
    """returns (value in [0,1), next seed)"""
    L = lib()
    L.orc_rand.restype = C.c_float
    s = C.c_uint32(seed)
    v = float(L.orc_rand(C.byref(s)))
    return v, int(s.value)


def post_accumulate(main_image, aux1, frame_sample_id):
    """in place on main_image (float32)"""
    m = np.ascontiguousarray(main_image, np.float32)
    a = np.ascontiguousarray(aux1, np.float32)
    lib().orc_post_accumulate(m.ctypes.data_as(F32P), a.ctypes.data_as(F32P), C.c_size_t(m.size), C.c_int(frame_sample_id))
    return m


def key_cull(frame, inst):
    total = sum(inst[i].count for i in range(len(inst)))
    keys = np.zeros(max(total, 1), np.uint32)
    ids = np.zeros(max(total, 1), np.uint32)
    v = lib().orc_key_cull(C.byref(frame), inst, len(inst), _u(keys), _u(ids))
    return keys[:v].copy(), ids[:v].copy()


def sort_stable(keys, ids):
    k = np.ascontiguousarray(keys, np.uint32).copy()
    v = np.ascontiguousarray(ids, np.uint32).copy()
    lib().orc_sort_stable(_u(k), _u(v), k.size)
    return k, v


def project(frame, inst, k, local_idx):
    P = OrcProjected()
    lib().orc_project(C.byref(frame), C.byref(inst[k]), local_idx, C.byref(P))
    return P


def render(frame, inst, order=None):
    img = np.zeros((frame.height, frame.width, 4), np.float32)
    stats = (C.c_uint64 * 2)()
    if order is None:
        frags = lib().orc_render(C.byref(frame), inst, len(inst), _p(img), stats)
    else:
        o = np.ascontiguousarray(order, np.uint32)
        frags = lib().orc_render_order(C.byref(frame), inst, len(inst), _u(o), o.size, _p(img), stats)
    return img, dict(fragments=int(frags), visible=int(stats[0]), quads=int(stats[1]))


def project_gut(frame, inst, k, local_idx):
    P = OrcGutProjected()
    fn = lib().orc_project_gut
    fn.restype = None
    fn.argtypes = [C.POINTER(OrcFrame), C.POINTER(OrcInstance), C.c_uint32, C.POINTER(OrcGutProjected)]
    fn(C.byref(frame), C.byref(inst[k]), local_idx, C.byref(P))
    return P


def gut_fragment(frame, inst, k, P, px, py):
    """opacity of the fragment of projected splat P at pixel (px, py), or None when the hit is rejected"""
    fn = lib().orc_gut_fragment
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(OrcFrame), C.POINTER(OrcInstance), C.POINTER(OrcGutProjected), C.c_int, C.c_int, F32P]
    op = C.c_float()
    ok = fn(C.byref(frame), C.byref(inst[k]), C.byref(P), int(px), int(py), C.byref(op))
    return float(op.value) if ok else None


def gut_fragment_iso(frame, inst, k, P, px, py, thin_particle_threshold=1e-6):
    """(opacity, world normal under NORMAL_METHOD_ISO_SURFACE) of the fragment, or None when the hit is rejected"""
    fn = lib().orc_gut_fragment_iso
    fn.restype = C.c_int
    fn.argtypes = [C.POINTER(OrcFrame), C.POINTER(OrcInstance), C.POINTER(OrcGutProjected), C.c_int, C.c_int, F32P, C.c_float, C.c_void_p]
    op = C.c_float()
    n = np.zeros(3, np.float32)
    ok = fn(C.byref(frame), C.byref(inst[k]), C.byref(P), int(px), int(py), C.byref(op), float(thin_particle_threshold), n.ctypes.data)
    return (float(op.value), n) if ok else None


def render_gut(frame, inst, order):
    """3DGUT frame (threedgut_raster.{mesh,frag}.slang) in the supplied draw order"""
    img = np.zeros((frame.height, frame.width, 4), np.float32)
    stats = (C.c_uint64 * 2)()
    o = np.ascontiguousarray(order, np.uint32)
    fn = lib().orc_render_gut_order
    fn.restype = C.c_uint64
    fn.argtypes = [C.POINTER(OrcFrame), C.POINTER(OrcInstance), C.c_int, U32P, C.c_uint32, F32P, C.POINTER(C.c_uint64)]
    frags = fn(C.byref(frame), inst, len(inst), _u(o), o.size, _p(img), stats)
    return img, dict(fragments=int(frags), visible=int(stats[0]), quads=int(stats[1]))


def render_window(frame, inst, order, window):
    """orc_render_order restricted to the inclusive pixel window (x0, y0, x1, y1) -> [h][w][4] float32"""
    x0, y0, x1, y1 = [int(v) for v in window]
    o = np.ascontiguousarray(order, np.uint32)
    img = np.zeros((y1 - y0 + 1, x1 - x0 + 1, 4), np.float32)
    fn = lib().orc_render_window
    fn.restype = C.c_uint64
    fn.argtypes = [C.POINTER(OrcFrame), C.POINTER(OrcInstance), C.c_int, U32P, C.c_uint32, C.POINTER(C.c_int), F32P]
    win = (C.c_int * 4)(x0, y0, x1, y1)
    frags = fn(C.byref(frame), inst, len(inst), _u(o), o.size, win, _p(img))
    return img, int(frags)


def render_surface(frame, inst, order_front_to_back, depth_iso_threshold=0.7, thin_particle_threshold=1e-6,
                   quantize_normals=True, normals=False):
    """FTB side outputs: (depth[H,W] float32, splat_id[H,W] uint32[, normal[H,W,4] float32]) — picked depth, the
    splat that set it and (normals=True) the integrated normal attachment"""
    o = np.ascontiguousarray(order_front_to_back, np.uint32)
    depth = np.zeros((frame.height, frame.width), np.float32)
    ids = np.zeros((frame.height, frame.width), np.uint32)
    nrm = np.zeros((frame.height, frame.width, 4), np.float32) if normals else None
    fn = lib().orc_render_surface
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_int, C.c_void_p,
                   C.c_void_p, C.c_void_p]
    fn(C.cast(C.byref(frame), C.c_void_p), C.cast(inst, C.c_void_p), len(inst), o.ctypes.data, o.size,
       float(depth_iso_threshold), float(thin_particle_threshold), int(bool(quantize_normals)), depth.ctypes.data,
       ids.ctypes.data, nrm.ctypes.data if normals else None)
    return (depth, ids, nrm) if normals else (depth, ids)


def render_surface_gut(frame, inst, order_front_to_back, depth_iso_threshold=0.7, thin_particle_threshold=1e-6, normals=False):
    """3DGUT pipeline: (depth, splat_id[, normal]) as render_surface"""
    o = np.ascontiguousarray(order_front_to_back, np.uint32)
    depth = np.zeros((frame.height, frame.width), np.float32)
    ids = np.zeros((frame.height, frame.width), np.uint32)
    nrm = np.zeros((frame.height, frame.width, 4), np.float32) if normals else None
    fn = lib().orc_render_surface_gut
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_uint32, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]
    fn(C.cast(C.byref(frame), C.c_void_p), C.cast(inst, C.c_void_p), len(inst), o.ctypes.data, o.size,
       float(depth_iso_threshold), float(thin_particle_threshold), depth.ctypes.data, ids.ctypes.data,
       nrm.ctypes.data if normals else None)
    return (depth, ids, nrm) if normals else (depth, ids)


def splat_normal(frame, inst, k, local_idx, thin_particle_threshold=1e-6, quantize=False):
    out = np.zeros(3, np.float32)
    fn = lib().orc_splat_normal
    fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_float, C.c_int, C.c_void_p]
    fn(C.cast(C.byref(frame), C.c_void_p), C.cast(C.byref(inst[k]), C.c_void_p), int(local_idx),
       float(thin_particle_threshold), int(bool(quantize)), out.ctypes.data)
    return out


def oct_roundtrip(n):
    n = f32(n).reshape(3)
    out = np.zeros(3, np.float32)
    L = lib()
    L.orc_oct_encode.restype = C.c_uint32
    L.orc_oct_encode.argtypes = [C.c_void_p]
    L.orc_oct_decode.restype = None
    L.orc_oct_decode.argtypes = [C.c_uint32, C.c_void_p]
    code = L.orc_oct_encode(n.ctypes.data)
    L.orc_oct_decode(code, out.ctypes.data)
    return int(code), out


def psnr_rgb(a, b):
    a, b = f32(a), f32(b)
    return float(lib().orc_psnr_rgb(_p(a), _p(b), a.shape[1], a.shape[0]))


def cpu_sort(view_dir, cop, sets_and_transforms, front_to_back=False, threads=0):
    """[(positions[3n], 4x4 math matrix or None)] -> (distances, indices, dist_ms, sort_ms)"""
    n_inst = len(sets_and_transforms)
    arr = (OrcSortInstance * n_inst)()
    keep, off = [], 0
    for i, (pos, m) in enumerate(sets_and_transforms):
        pos = f32(pos).reshape(-1)
        mm = np.eye(4, dtype=np.float32) if m is None else f32(m)
        col = np.ascontiguousarray(mm.T.reshape(-1))
        arr[i].positions, arr[i].count, arr[i].global_offset = _p(pos), pos.size // 3, off
        for k in range(16):
            arr[i].transform[k] = float(col[k])
        off += pos.size // 3
        keep.append(pos)
    dist = np.zeros(max(off, 1), np.float32)
    idx = np.zeros(max(off, 1), np.uint32)
    d_ms, s_ms = C.c_double(), C.c_double()
    d, c = f32(view_dir), f32(cop)
    rc = lib().orc_cpu_sort(_p(d), _p(c), arr, n_inst, off, int(front_to_back), threads, _p(dist), _u(idx),
                            C.byref(d_ms), C.byref(s_ms))
    if rc != 0:
        raise RuntimeError("orc_cpu_sort failed")
    return dist[:off], idx[:off], d_ms.value, s_ms.value


# ---- reference-compiled ingest (oracle/_ref), present only where it was built --------------------
_ref = None


def ref_lib():
    global _ref
    if _ref is None:
        path = os.path.join(HERE, "_ref", "libref_ingest.so")
        if not os.path.exists(path):
            build()
        if not os.path.exists(path):
            return None
        R = C.CDLL(path)
        R.ref_ply_load.restype = C.c_void_p
        R.ref_ply_load.argtypes = [C.c_char_p]
        R.ref_spz_load.restype = C.c_void_p
        R.ref_spz_load.argtypes = [C.c_char_p]
        R.ref_spz_save.restype = C.c_int
        R.ref_spz_save.argtypes = [C.c_char_p, C.c_int, C.c_int, F32P, F32P, F32P, F32P, F32P, F32P, C.c_int]
        R.ref_set_size.restype = C.c_size_t
        R.ref_set_size.argtypes = [C.c_void_p, C.c_int]
        R.ref_set_copy.argtypes = [C.c_void_p, C.c_int, F32P]
        R.ref_set_max_sh_degree.restype = C.c_int
        R.ref_set_max_sh_degree.argtypes = [C.c_void_p]
        R.ref_set_free.argtypes = [C.c_void_p]
        R.ref_max_sh_degree.restype = C.c_int
        R.ref_max_sh_degree.argtypes = [C.c_size_t, C.c_size_t]
        R.ref_flip_sh.argtypes = [C.c_int, C.c_int, F32P, F32P, F32P]
        _ref = R
    return _ref


REF_FIELDS = ["positions", "f_dc", "f_rest", "opacity", "scale", "rotation"]


def ref_load(path):
    R = ref_lib()
    fn = R.ref_spz_load if path.lower().endswith(".spz") else R.ref_ply_load
    h = fn(os.fsencode(path))
    if not h:
        return None
    out = {}
    for i, name in enumerate(REF_FIELDS):
        n = R.ref_set_size(h, i)
        a = np.zeros(n, np.float32)
        if n:
            R.ref_set_copy(h, i, _p(a))
        out[name] = a
    out["sh_degree"] = R.ref_set_max_sh_degree(h)
    R.ref_set_free(h)
    return out
