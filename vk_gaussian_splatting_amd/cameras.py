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
    name: str = ""
    width: int = 0
    height: int = 0

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
