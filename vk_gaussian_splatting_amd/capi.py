"""ctypes mirror of include/mgs.h (the C ABI of csrc/libmgs.so).

Names, argument meaning and error behaviour follow the header one to one; see the header for
the reference interface (file:line) each entry point replaces.
"""
import ctypes as C
import os
import numpy as np

FORMAT_FLOAT32, FORMAT_FLOAT16, FORMAT_UINT8 = 0, 1, 2
SORT_GPU_RADIX, SORT_CPU_ASYNC, SORT_STOCHASTIC = 0, 1, 3
DOF_DISABLED, DOF_FIXED_FOCUS = 0, 1
CULL_NONE, CULL_AT_DIST, CULL_AT_RASTER = 0, 1, 2
TARGET_RGBA16F, TARGET_RGBA32F, TARGET_RGBA8 = 0, 1, 2
ALPHA_COVERAGE, ALPHA_SUM = 0, 1
DEBUG_POINT_CLOUD, DEBUG_SH_ONLY, DEBUG_OPACITY_GAUSSIAN_DISABLED = 1, 2, 4
PIPELINE_3DGS, PIPELINE_3DGUT = 0, 1
NORMAL_MAX_DENSITY_PLANE, NORMAL_ISO_SURFACE = 0, 1
CAMERA_PINHOLE, CAMERA_FISHEYE = 0, 1
EXTENT_EIGEN, EXTENT_CONIC = 0, 1
STAGE_NAMES = ["project", "sort", "bin", "pairsort", "composite", "total"]


# status codes of include/mgs.h
OK, ERR_INVALID_ARG, ERR_IO, ERR_FORMAT, ERR_DEVICE, ERR_OOM, ERR_STATE, ERR_OVERFLOW, ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5, -6, -7, -8


class MgsError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"mgs error {code}: {msg}")
        self.code = code


class SplatSetView(C.Structure):
    _fields_ = [("positions", C.POINTER(C.c_float)), ("f_dc", C.POINTER(C.c_float)),
                ("f_rest", C.POINTER(C.c_float)), ("opacity", C.POINTER(C.c_float)),
                ("scale", C.POINTER(C.c_float)), ("rotation", C.POINTER(C.c_float)),
                ("splat_count", C.c_uint64), ("f_rest_per_splat", C.c_uint32), ("sh_degree", C.c_int32)]


class FrameParams(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("proj", C.c_float * 16), ("camera_pos", C.c_float * 3),
                ("width", C.c_int32), ("height", C.c_int32),
                ("splat_scale", C.c_float), ("frustum_dilation", C.c_float), ("alpha_cull_threshold", C.c_float),
                ("sh_degree", C.c_int32), ("sort_mode", C.c_int32), ("frustum_culling", C.c_int32),
                ("target_format", C.c_int32), ("alpha_mode", C.c_int32), ("ms_antialiasing", C.c_int32),
                ("strip_row_begin", C.c_int32), ("strip_row_end", C.c_int32),
                ("collect_timings", C.c_int32), ("cpu_sort_blocking", C.c_int32), ("debug_flags", C.c_int32),
                ("size_culling", C.c_int32), ("size_culling_min_pixels", C.c_float),
                ("surface_outputs", C.c_int32), ("depth_iso_threshold", C.c_float), ("cpu_lazy_sort", C.c_int32),
                ("thin_particle_threshold", C.c_float), ("quantize_normals", C.c_int32),
                ("pipeline", C.c_int32), ("camera_model", C.c_int32), ("extent_method", C.c_int32), ("fov_rad", C.c_float),
                ("alpha_clamp", C.c_float), ("kernel_min_response", C.c_float),
                ("dof_mode", C.c_int32), ("focus_dist", C.c_float), ("aperture", C.c_float),
                ("frame_sample_id", C.c_int32), ("temporal_sampling", C.c_int32), ("kernel_degree", C.c_int32),
                ("normal_method", C.c_int32), ("reserved_", C.c_int32 * 1)]


class FrameOut(C.Structure):
    _fields_ = [("rgba_device", C.c_void_p), ("rgba_bytes", C.c_uint64),
                ("frustum_count", C.c_uint32), ("sorted_count", C.c_uint32), ("tile_pairs", C.c_uint64),
                ("error_flags", C.c_uint32), ("shaded_count", C.c_uint32), ("scanned_entries", C.c_uint64),
                ("stage_ms", C.c_float * 8), ("escape_count", C.c_uint32), ("reserved0", C.c_uint32)]


class SortOut(C.Structure):
    _fields_ = [("count", C.c_uint32), ("key_ms", C.c_float), ("sort_ms", C.c_float), ("hist_ms", C.c_float),
                ("passes", C.c_uint32), ("reserved", C.c_uint32 * 3)]


def lib_path():
    if os.environ.get("MGS_LIB"):  # experiments: a variant build of the same library (tools/build_variant.sh); never a fallback
        return os.path.abspath(os.environ["MGS_LIB"])
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc", "libmgs.so")


_lib = None


def load_library():
    """Load csrc/libmgs.so.  Fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = lib_path()
    if not os.path.exists(path):
        raise MgsError(-4, f"{path} is missing — run __graft_entry__.build() (hipcc --offload-arch=gfx950); "
                           "there is no CPU fallback for the MI355X path")
    # torch bundles its own libamdhip64 (same SONAME as /opt/rocm's).  Load torch first so that this
    # process ends up with exactly ONE HIP runtime, whichever library asks for it later.
    try:
        import torch  # noqa: F401
    except Exception:
        pass
    lib = C.CDLL(path)
    F, P = C.c_float, C.POINTER
    vp = C.c_void_p
    sig = {
        "mgs_last_error": (C.c_char_p, []),
        "mgs_version": (C.c_char_p, []),
        "mgs_splatset_load": (C.c_int, [C.c_char_p, P(vp)]),
        "mgs_splatset_from_arrays": (C.c_int, [P(SplatSetView), P(vp)]),
        "mgs_splatset_view": (C.c_int, [vp, P(SplatSetView)]),
        "mgs_splatset_destroy": (None, [vp]),
        "mgs_scene_create": (C.c_int, [C.c_int, P(vp)]),
        "mgs_scene_destroy": (None, [vp]),
        "mgs_scene_set_stream": (C.c_int, [vp, vp]),
        "mgs_instance_add": (C.c_int, [vp, vp, P(F), P(C.c_int)]),
        "mgs_instance_set_transform": (C.c_int, [vp, C.c_int, P(F)]),
        "mgs_scene_commit": (C.c_int, [vp, C.c_int, C.c_int]),
        "mgs_scene_splat_count": (C.c_uint64, [vp]),
        "mgs_frame_context_create": (C.c_int, [vp, P(vp)]),
        "mgs_frame_context_destroy": (None, [vp]),
        "mgs_scene_memory_usage": (C.c_int, [vp, P(C.c_uint64), P(C.c_uint64)]),
        "mgs_scene_set_list_capacity": (C.c_int, [vp, C.c_uint64]),
        "mgs_scene_download_set": (C.c_int, [vp, C.c_int, C.c_int, P(F), C.c_size_t]),
        "mgs_scene_storage_order": (C.c_int, [vp, C.c_int, P(C.c_uint32), C.c_size_t]),
        "mgs_frame_params_default": (None, [P(FrameParams)]),
        "mgs_render": (C.c_int, [vp, P(FrameParams), P(FrameOut)]),
        "mgs_frame_stats": (C.c_int, [vp, P(FrameOut)]),
        "mgs_timings_query": (C.c_int, [vp, C.c_uint32, P(F)]),
        "mgs_frame_download": (C.c_int, [vp, vp, C.c_size_t]),
        "mgs_frame_download_surface": (C.c_int, [vp, C.c_int, vp, C.c_size_t]),
        "mgs_frame_copy_strip": (C.c_int, [vp, vp, C.c_size_t]),
        "mgs_sync": (C.c_int, [vp]),
        "mgs_frame_download_projected": (C.c_int, [vp, P(C.c_uint32), C.c_size_t, P(F), P(C.c_uint32)]),
        "mgs_loader_create": (C.c_int, [P(vp)]),
        "mgs_loader_destroy": (None, [vp]),
        "mgs_loader_push": (C.c_int, [vp, C.c_char_p]),
        "mgs_loader_status": (C.c_int, [vp, P(C.c_int), P(C.c_uint32), C.c_char_p, C.c_size_t]),
        "mgs_loader_take": (C.c_int, [vp, P(vp)]),
        "mgs_comm_unique_id": (C.c_int, [vp]),
        "mgs_scene_comm_init": (C.c_int, [vp, C.c_int, C.c_int, vp]),
        "mgs_scene_comm_destroy": (C.c_int, [vp]),
        "mgs_scene_set_strip_rows": (C.c_int, [vp, P(C.c_int32), C.c_int]),
        "mgs_render_gathered": (C.c_int, [vp, P(FrameParams), P(FrameOut)]),
        "mgs_frame_row_costs": (C.c_int, [vp, P(C.c_uint32), C.c_size_t]),
        "mgs_sort_keys": (C.c_int, [vp, P(FrameParams), P(SortOut)]),
        "mgs_sort_download": (C.c_int, [vp, P(C.c_uint32), P(C.c_uint32), C.c_uint32]),
        "mgs_radix_sort_u32": (C.c_int, [vp, vp, vp, C.c_uint32, C.c_int, C.c_int, P(F)]),
        "mgs_radix_sort_host": (C.c_int, [vp, P(C.c_uint32), P(C.c_uint32), C.c_uint32, C.c_int, C.c_int, P(F)]),
        "mgs_camera_lookat_perspective": (None, [P(F), P(F), P(F), F, F, F, C.c_int, C.c_int, C.c_int, P(F), P(F)]),
        "mgs_compute_transform": (None, [P(F), P(F), P(F), P(F), P(F)]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)  # AttributeError here == the library does not export what mgs.h declares
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


EXPORTED_SYMBOLS = [
    "mgs_last_error", "mgs_version", "mgs_splatset_load", "mgs_splatset_from_arrays", "mgs_splatset_view",
    "mgs_splatset_destroy", "mgs_scene_create", "mgs_scene_destroy", "mgs_scene_set_stream", "mgs_instance_add",
    "mgs_instance_set_transform", "mgs_scene_commit", "mgs_scene_splat_count", "mgs_scene_storage_order", "mgs_scene_download_set",
    "mgs_frame_context_create", "mgs_frame_context_destroy", "mgs_scene_memory_usage", "mgs_scene_set_list_capacity",
    "mgs_frame_params_default", "mgs_render", "mgs_frame_stats", "mgs_timings_query", "mgs_frame_download", "mgs_frame_download_surface", "mgs_frame_copy_strip",
    "mgs_frame_download_projected", "mgs_sync", "mgs_comm_unique_id", "mgs_scene_comm_init", "mgs_scene_comm_destroy",
    "mgs_scene_set_strip_rows", "mgs_render_gathered", "mgs_frame_row_costs",
    "mgs_loader_create", "mgs_loader_destroy", "mgs_loader_push", "mgs_loader_status", "mgs_loader_take", "mgs_sort_keys", "mgs_sort_download", "mgs_radix_sort_u32", "mgs_radix_sort_host",
    "mgs_camera_lookat_perspective", "mgs_compute_transform"]


def _check(rc):
    if rc != 0:
        raise MgsError(rc, load_library().mgs_last_error().decode(errors="replace"))


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


IDENTITY = np.eye(4, dtype=np.float32)


class SplatSet:
    """RAM splat model (struct SplatSet, src/splat_set.h:33-48)."""

    def __init__(self, handle):
        self._h = handle

    @classmethod
    def load(cls, path):
        lib = load_library()
        h = C.c_void_p()
        _check(lib.mgs_splatset_load(os.fsencode(path), C.byref(h)))
        return cls(h)

    @classmethod
    def from_arrays(cls, positions, f_dc, f_rest, opacity, scale, rotation):
        lib = load_library()
        keep = [_f32(positions).reshape(-1), _f32(f_dc).reshape(-1),
                None if f_rest is None else _f32(f_rest).reshape(-1),
                _f32(opacity).reshape(-1), _f32(scale).reshape(-1), _f32(rotation).reshape(-1)]
        n = keep[0].size // 3
        v = SplatSetView()
        v.positions, v.f_dc = _fp(keep[0]), _fp(keep[1])
        v.f_rest = _fp(keep[2]) if keep[2] is not None and keep[2].size else None
        v.opacity, v.scale, v.rotation = _fp(keep[3]), _fp(keep[4]), _fp(keep[5])
        v.splat_count = n
        v.f_rest_per_splat = 0 if keep[2] is None or n == 0 else keep[2].size // n
        h = C.c_void_p()
        _check(lib.mgs_splatset_from_arrays(C.byref(v), C.byref(h)))
        return cls(h)

    def arrays(self):
        """dict of numpy copies of the six SoA arrays + sh_degree."""
        lib = load_library()
        v = SplatSetView()
        _check(lib.mgs_splatset_view(self._h, C.byref(v)))
        n = v.splat_count

        def cp(ptr, cnt):
            return np.ctypeslib.as_array(ptr, shape=(cnt,)).copy() if cnt and ptr else np.zeros(0, np.float32)
        return dict(positions=cp(v.positions, 3 * n), f_dc=cp(v.f_dc, 3 * n),
                    f_rest=cp(v.f_rest, v.f_rest_per_splat * n), opacity=cp(v.opacity, n), scale=cp(v.scale, 3 * n),
                    rotation=cp(v.rotation, 4 * n), count=n, f_rest_per_splat=v.f_rest_per_splat,
                    sh_degree=v.sh_degree)

    def close(self):
        if self._h:
            load_library().mgs_splatset_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


LOADER_READY, LOADER_LOADING, LOADER_LOADED, LOADER_FAILURE = 1, 2, 3, 4


class Loader:
    """PlyLoaderAsync + the scene-load queue: push files, poll, take the loaded SplatSets in order"""

    def __init__(self):
        self._lib = load_library()
        h = C.c_void_p()
        _check(self._lib.mgs_loader_create(C.byref(h)))
        self._h = h

    def push(self, path):
        _check(self._lib.mgs_loader_push(self._h, os.fsencode(path)))

    def status(self):
        """(state, requests queued behind the head, head's path)"""
        st, q = C.c_int(), C.c_uint32()
        buf = C.create_string_buffer(1024)
        _check(self._lib.mgs_loader_status(self._h, C.byref(st), C.byref(q), buf, 1024))
        return st.value, q.value, buf.value.decode(errors="replace")

    def take(self):
        h = C.c_void_p()
        _check(self._lib.mgs_loader_take(self._h, C.byref(h)))
        return SplatSet(h)

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mgs_loader_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def default_params(width=1920, height=1080):
    p = FrameParams()
    load_library().mgs_frame_params_default(C.byref(p))
    p.width, p.height = width, height
    return p


def set_camera(p, view, proj, camera_pos):
    """view/proj: 4x4 numpy in math (row, col) convention; stored glm column-major."""
    v = _f32(view).T.reshape(-1)
    pr = _f32(proj).T.reshape(-1)
    for i in range(16):
        p.view[i] = float(v[i])
        p.proj[i] = float(pr[i])
    for i in range(3):
        p.camera_pos[i] = float(camera_pos[i])


def camera_lookat_perspective(eye, center, up, fov_deg, z_near, z_far, width, height, flip_y=False):
    """returns (view, proj) as 4x4 numpy arrays in math convention (row, col)."""
    lib = load_library()
    e, c, u = _f32(eye), _f32(center), _f32(up)
    v = np.zeros(16, np.float32)
    p = np.zeros(16, np.float32)
    lib.mgs_camera_lookat_perspective(_fp(e), _fp(c), _fp(u), fov_deg, z_near, z_far, width, height, int(flip_y),
                                      _fp(v), _fp(p))
    return v.reshape(4, 4).T.copy(), p.reshape(4, 4).T.copy()


def comm_unique_id():
    """ncclGetUniqueId through libmgs: 128 bytes; call on one rank and distribute"""
    buf = C.create_string_buffer(128)
    _check(load_library().mgs_comm_unique_id(C.cast(buf, C.c_void_p)))
    return bytes(buf.raw)


def compute_transform(scale, rotation_deg, translation):
    lib = load_library()
    m = np.zeros(16, np.float32)
    mi = np.zeros(16, np.float32)
    lib.mgs_compute_transform(_fp(_f32(scale)), _fp(_f32(rotation_deg)), _fp(_f32(translation)), _fp(m), _fp(mi))
    return m.reshape(4, 4).T.copy(), mi.reshape(4, 4).T.copy()


class Scene:
    """Device scene (SplatSetManagerVk + renderer buffers) on one MI355X."""

    def __init__(self, device=0, _context_of=None):
        lib = load_library()
        self._lib = lib
        h = C.c_void_p()
        if _context_of is None:
            _check(lib.mgs_scene_create(device, C.byref(h)))
        else:
            _check(lib.mgs_frame_context_create(_context_of._h, C.byref(h)))
        self._h = h
        self._sets = [] if _context_of is None else _context_of._sets

    def frame_context(self):
        """a frame in flight over this scene's committed data: own stream, working buffers and graphs (mgs_frame_context_create)"""
        return Scene(_context_of=self)

    def set_list_capacity(self, entries):
        _check(self._lib.mgs_scene_set_list_capacity(self._h, int(entries)))

    def memory_usage(self):
        """(scene_bytes, working_bytes): the committed data shared by all contexts, and this handle's working set"""
        a, b = C.c_uint64(), C.c_uint64()
        _check(self._lib.mgs_scene_memory_usage(self._h, C.byref(a), C.byref(b)))
        return int(a.value), int(b.value)

    def set_stream(self, stream_ptr):
        _check(self._lib.mgs_scene_set_stream(self._h, C.c_void_p(stream_ptr)))

    def add_instance(self, splat_set, transform=None):
        m = _f32(IDENTITY if transform is None else transform).T.reshape(-1).copy()  # -> glm column-major
        idx = C.c_int()
        _check(self._lib.mgs_instance_add(self._h, splat_set._h, _fp(m), C.byref(idx)))
        self._sets.append(splat_set)
        return idx.value

    def set_transform(self, instance, transform):
        m = _f32(transform).T.reshape(-1).copy()
        _check(self._lib.mgs_instance_set_transform(self._h, instance, _fp(m)))

    def commit(self, sh_format=FORMAT_FLOAT32, rgba_format=FORMAT_FLOAT32):
        _check(self._lib.mgs_scene_commit(self._h, sh_format, rgba_format))

    @property
    def splat_count(self):
        return int(self._lib.mgs_scene_splat_count(self._h))

    def storage_order(self, instance, count):
        """permutation storage index -> caller's index of the instance's splat set"""
        out = np.zeros(count, np.uint32)
        _check(self._lib.mgs_scene_storage_order(self._h, instance, out.ctypes.data_as(C.POINTER(C.c_uint32)), count))
        return out

    def download_set(self, instance, which, count):
        out = np.zeros(count, np.float32)
        _check(self._lib.mgs_scene_download_set(self._h, instance, which, _fp(out), out.size))
        return out

    def render(self, params, want_stats=False):
        out = FrameOut()
        self._sort_only = False
        _check(self._lib.mgs_render(self._h, C.byref(params), C.byref(out)))
        if want_stats and not params.collect_timings:
            _check(self._lib.mgs_frame_stats(self._h, C.byref(out)))
        return out

    def frame_stats(self):
        out = FrameOut()
        _check(self._lib.mgs_frame_stats(self._h, C.byref(out)))
        return out

    def timings(self, frames_back=0):
        ms = (C.c_float * 8)()
        _check(self._lib.mgs_timings_query(self._h, frames_back, ms))
        return [float(x) for x in ms[:6]]

    def timings_all(self, frames_back=0):
        """all MGS_STAGE_COUNT slots: the six of timings() + [6] = MGS_STAGE_CULL (the head of the project stage)"""
        ms = (C.c_float * 8)()
        _check(self._lib.mgs_timings_query(self._h, frames_back, ms))
        return [float(x) for x in ms]

    def download_frame(self, params):
        if params.target_format == TARGET_RGBA8:
            img = np.zeros((params.height, params.width, 4), np.uint8)
        elif params.target_format == TARGET_RGBA16F:
            img = np.zeros((params.height, params.width, 4), np.float16)
        else:
            img = np.zeros((params.height, params.width, 4), np.float32)
        _check(self._lib.mgs_frame_download(self._h, img.ctypes.data_as(C.c_void_p), img.nbytes))
        return img

    def download_surface(self, params, normals=False):
        """FTB side outputs of a frame rendered with params.surface_outputs = 1: (picked depth float32[H,W],
        splat id uint32[H,W] in the caller's id space, 0xFFFFFFFF = none[, integrated normal float32[H,W,4]])"""
        depth = np.zeros((params.height, params.width), np.float32)
        ids = np.zeros((params.height, params.width), np.uint32)
        _check(self._lib.mgs_frame_download_surface(self._h, 0, depth.ctypes.data_as(C.c_void_p), depth.nbytes))
        _check(self._lib.mgs_frame_download_surface(self._h, 1, ids.ctypes.data_as(C.c_void_p), ids.nbytes))
        if not normals:
            return depth, ids
        nrm = np.zeros((params.height, params.width, 4), np.float32)
        _check(self._lib.mgs_frame_download_surface(self._h, 2, nrm.ctypes.data_as(C.c_void_p), nrm.nbytes))
        return depth, ids, nrm

    def copy_strip(self, device_ptr, nbytes):
        _check(self._lib.mgs_frame_copy_strip(self._h, C.c_void_p(device_ptr), nbytes))

    def sync(self):
        _check(self._lib.mgs_sync(self._h))

    # ---- multi-GPU strips (RCCL inside libmgs) ----
    def comm_init(self, rank, world_size, unique_id):
        """unique_id: 128 bytes from comm_unique_id() of ONE rank, distributed out of band"""
        buf = C.create_string_buffer(bytes(unique_id), 128)
        _check(self._lib.mgs_scene_comm_init(self._h, rank, world_size, C.cast(buf, C.c_void_p)))

    def comm_destroy(self):
        _check(self._lib.mgs_scene_comm_destroy(self._h))

    def set_strip_rows(self, bounds):
        if bounds is None:
            _check(self._lib.mgs_scene_set_strip_rows(self._h, None, 0))
            return
        b = np.ascontiguousarray(bounds, np.int32)
        _check(self._lib.mgs_scene_set_strip_rows(self._h, b.ctypes.data_as(C.POINTER(C.c_int32)), b.size))

    def render_gathered(self, params):
        out = FrameOut()
        self._sort_only = False
        _check(self._lib.mgs_render_gathered(self._h, C.byref(params), C.byref(out)))
        return out

    def row_costs(self, height):
        rows = (height + 15) // 16
        out = np.zeros(rows, np.uint32)
        _check(self._lib.mgs_frame_row_costs(self._h, out.ctypes.data_as(C.POINTER(C.c_uint32)), rows))
        return out

    def download_projected(self, global_ids):
        """debug hook: (records[n,10] float32, rect[n] uint32) of the last frame for the given global ids"""
        ids = np.ascontiguousarray(global_ids, np.uint32)
        out = np.zeros((ids.size, 10), np.float32)
        rect = np.zeros(ids.size, np.uint32)
        _check(self._lib.mgs_frame_download_projected(self._h, ids.ctypes.data_as(C.POINTER(C.c_uint32)), ids.size,
                                                      _fp(out), rect.ctypes.data_as(C.POINTER(C.c_uint32))))
        return out, rect

    def sort_keys(self, params):
        out = SortOut()
        _check(self._lib.mgs_sort_keys(self._h, C.byref(params), C.byref(out)))
        self._sort_only = True
        return out

    def sort_download(self, count):
        """(keys, ids) of the last sort.  The sorted keys exist after sort_keys() only: a frame's last sort pass writes the ids
        alone, and keys is None then."""
        ids = np.zeros(max(count, 1), np.uint32)
        keys = np.zeros(max(count, 1), np.uint32) if getattr(self, "_sort_only", False) else None
        kp = keys.ctypes.data_as(C.POINTER(C.c_uint32)) if keys is not None else None
        _check(self._lib.mgs_sort_download(self._h, kp, ids.ctypes.data_as(C.POINTER(C.c_uint32)), ids.size))
        return (keys[:count] if keys is not None else None), ids[:count]

    def radix_sort_host(self, keys, values, begin_bit=0, end_bit=32):
        k = np.ascontiguousarray(keys, np.uint32).copy()
        v = np.ascontiguousarray(values, np.uint32).copy()
        ms = C.c_float()
        _check(self._lib.mgs_radix_sort_host(self._h, k.ctypes.data_as(C.POINTER(C.c_uint32)),
                                             v.ctypes.data_as(C.POINTER(C.c_uint32)), k.size, begin_bit, end_bit,
                                             C.byref(ms)))
        return k, v, ms.value

    def radix_sort_device(self, keys_ptr, vals_ptr, count, begin_bit=0, end_bit=32):
        ms = C.c_float()
        _check(self._lib.mgs_radix_sort_u32(self._h, C.c_void_p(keys_ptr), C.c_void_p(vals_ptr), count, begin_bit,
                                            end_bit, C.byref(ms)))
        return ms.value

    def close(self):
        if getattr(self, "_h", None):
            self._lib.mgs_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
