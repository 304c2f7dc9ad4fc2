"""Camera description and INRIA cameras.json import — host-side callers of the path.

Mirrors struct Camera (src/camera_set.h:44-63) and importCamerasINRIA (src/camera_set.h:219-270):
INRIA stores world-to-camera rotations and positions in the RDF frame; the renderer works in RUB, so the
y and z components of the eye are negated and the rotation is re-signed exactly as the reference does,
then `up` and `at` are the normalised 2nd and 3rd columns of that matrix.  The imported cameras keep the
default field of view and clip planes (the reference reads fx/fy but does not use them).
"""
import json
from dataclasses import dataclass, field

import numpy as np


@dataclass
class Camera:
    # defaults of src/camera_set.h:48-53
    eye: np.ndarray = field(default_factory=lambda: np.array([1.7, 1.5, 1.7], np.float32))
    ctr: np.ndarray = field(default_factory=lambda: np.array([0.0, 0.0, 0.0], np.float32))
    up: np.ndarray = field(default_factory=lambda: np.array([0.0, 1.0, 0.0], np.float32))
    fov: float = 60.0
    clip: tuple = (0.1, 2000.0)
    model: int = 0        # CAMERA_PINHOLE 0 / CAMERA_FISHEYE 1 (camera_set.h:46)
    dof_mode: int = 0     # DOF_DISABLED (camera_set.h:55-57)
    focus_dist: float = 1.3
    aperture: float = 0.001
    name: str = ""
    width: int = 0
    height: int = 0

    def same_view(self, o):
        """Camera::operator== (camera_set.h:59-62): model, eye, ctr, up, fov, clip — the depth-of-field fields do not count"""
        return (self.model == o.model and np.array_equal(np.asarray(self.eye, np.float32), np.asarray(o.eye, np.float32))
                and np.array_equal(np.asarray(self.ctr, np.float32), np.asarray(o.ctr, np.float32))
                and np.array_equal(np.asarray(self.up, np.float32), np.asarray(o.up, np.float32))
                and float(self.fov) == float(o.fov) and tuple(map(float, self.clip)) == tuple(map(float, o.clip)))

    def apply(self, params):
        """the camera's share of MgsFrameParams beyond the matrices: sensor model and depth of field (3DGUT pipeline)"""
        params.camera_model = int(self.model)
        params.dof_mode = 1 if int(self.dof_mode) != 0 else 0   # DOF_AUTO_FOCUS (2) = fixed focus at a picked distance
        params.focus_dist = float(self.focus_dist)
        params.aperture = float(self.aperture)
        return params

    def matrices(self, width, height, flip_y=False):
        """(view, proj) through the C ABI helper mgs_camera_lookat_perspective"""
        from . import capi
        return capi.camera_lookat_perspective(self.eye, self.ctr, self.up, self.fov, self.clip[0], self.clip[1], width,
                                              height, flip_y)


def import_cameras_inria(path):
    """list[Camera] from an INRIA cameras.json; raises on malformed files (the reference returns false)"""
    with open(path) as f:
        data = json.load(f)
    out = []
    for item in data:
        pos = np.asarray(item["position"], np.float32)
        R = np.asarray(item["rotation"], np.float32)
        if pos.shape != (3,) or R.shape != (3, 3):
            raise ValueError("cameras.json: position must have 3 and rotation 3x3 entries")
        # glm::mat3(c0..., c1..., c2...) is column-major: columns as written in camera_set.h:248-250
        c1 = np.array([-R[0, 1], R[1, 1], R[2, 1]], np.float32)
        c2 = np.array([R[0, 2], -R[1, 2], -R[2, 2]], np.float32)
        up = c1 / np.linalg.norm(c1)
        at = c2 / np.linalg.norm(c2)
        eye = np.array([pos[0], -pos[1], -pos[2]], np.float32)
        out.append(Camera(eye=eye, ctr=eye + at, up=up, name=str(item.get("img_name", "")),
                          width=int(item.get("width", 0)), height=int(item.get("height", 0))))
    return out


class CameraSet:
    """class CameraSet (src/camera_set.h:65-190) without the manipulator: the active camera and the list of presets, with the
    reference's rules — preset 0 is the home preset, createPreset returns the index of an equal preset instead of adding a
    duplicate, the last preset cannot be erased."""

    def __init__(self):
        self.camera = Camera()
        self.presets = []

    def reset(self):
        self.camera = Camera()
        self.presets = []

    def set_camera(self, camera):
        self.camera = camera

    def set_home_preset(self, camera):
        if not self.presets:
            self.presets.append(camera)
        else:
            self.presets[0] = camera

    def __len__(self):
        return len(self.presets)

    def clear_presets(self):
        self.presets = []

    def create_preset(self, camera):
        for i, p in enumerate(self.presets):
            if p.same_view(camera):
                return i
        self.presets.append(camera)
        return len(self.presets) - 1

    def store_current_camera(self):
        return self.create_preset(self.camera)

    def load_preset(self, index):
        if index < 0 or index >= len(self.presets):
            return False
        self.camera = self.presets[index]
        return True

    def erase_preset(self, index):
        if len(self.presets) <= 1 or index < 0 or index >= len(self.presets):
            return False
        del self.presets[index]
        return True

    def get_preset(self, index):
        return self.presets[index]

    def set_preset(self, index, camera):
        if index < 0 or index >= len(self.presets):
            return False
        self.presets[index] = camera
        return True

    def import_inria(self, path):
        """importCamerasINRIA: every camera of the file becomes a preset (duplicates collapse); returns how many were read"""
        cams = import_cameras_inria(path)
        for c in cams:
            self.create_preset(c)
        return len(cams)
