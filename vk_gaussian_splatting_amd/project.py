""".vkgs project files — the reference's own scene description.

Format = the JSON written by VkgsProjectWriter (src/vkgs_project_writer.cpp:75-330, PROJECT_FILE_VERSION 5) and
read by VkgsProjectReader (src/vkgs_project_reader.cpp:55-345).  Only the sections the VK3DGSR path consumes are
interpreted: "renderer" (raster knobs), "camera" / "cameras", "splatsGlobals" (storage formats), "splatSets"
(assets: id + path relative to the project file) and "splats" (instances: splatSetId + position / rotation in
degrees / scale, composed as T*R*S like computeTransform, src/utilities.h:170-199).  Lights, meshes, RTX and
DLSS settings are carried through untouched on save but otherwise ignored (out of scope).
"""
import json
import os
from dataclasses import dataclass, field

import numpy as np

from .cameras import Camera

PROJECT_FILE_VERSION = 5


@dataclass
class SplatInstance:
    splat_set_id: int
    name: str = ""
    position: tuple = (0.0, 0.0, 0.0)
    rotation: tuple = (0.0, 0.0, 0.0)   # Euler degrees
    scale: tuple = (1.0, 1.0, 1.0)


@dataclass
class Project:
    version: int = PROJECT_FILE_VERSION
    renderer: dict = field(default_factory=dict)
    camera: Camera = field(default_factory=Camera)
    cameras: list = field(default_factory=list)
    sh_format: int = 0
    rgba_format: int = 0
    splat_sets: dict = field(default_factory=dict)      # id -> absolute path
    instances: list = field(default_factory=list)       # [SplatInstance]
    extra: dict = field(default_factory=dict)           # sections we do not interpret (lights, meshes, ...)

    # ---- mapping onto the C ABI ------------------------------------------------------------------
    def frame_params(self, width, height, flip_y=False):
        """MgsFrameParams filled from "renderer" + "camera" (defaults where a key is absent, like the reader's LOAD1)"""
        from . import capi
        r = self.renderer
        p = capi.default_params(width, height)
        V, P = self.camera.matrices(width, height, flip_y)
        capi.set_camera(p, V, P, self.camera.eye)
        p.sh_degree = int(r.get("maxShDegree", 3))
        p.frustum_culling = int(r.get("frustumCulling", capi.CULL_AT_DIST))
        p.size_culling = int(r.get("sizeCulling", 0))
        p.size_culling_min_pixels = float(r.get("sizeCullingMinPixels", 1.0))
        # shaderio.h:24-27: 0 GPU radix, 1/2 CPU async (mono / multi), 3 stochastic splat
        sm = int(r.get("sortingMethod", 0))
        p.sort_mode = {0: capi.SORT_GPU_RADIX, 3: capi.SORT_STOCHASTIC}.get(sm, capi.SORT_CPU_ASYNC)
        # "pipeline" (shaderio.h:61-66): 4 / 5 = the 3DGUT mesh pipeline (5 = hybrid with ray-traced secondary rays: its raster part);
        # everything else renders through the 3DGS raster path here
        p.pipeline = capi.PIPELINE_3DGUT if int(r.get("pipeline", 1)) in (4, 5) else capi.PIPELINE_3DGS
        p.kernel_degree = int(r.get("kernelDegree", 2))
        p.kernel_min_response = float(r.get("kernelMinResponse", 0.0113))
        p.temporal_sampling = int(bool(r.get("temporalSampling", False)))
        self.camera.apply(p)
        if p.pipeline != capi.PIPELINE_3DGUT:
            p.dof_mode = capi.DOF_DISABLED   # the 3DGS raster pipeline has no per-pixel rays
        p.cpu_lazy_sort = int(bool(r.get("cpuLazySort", True)))
        p.thin_particle_threshold = float(r.get("thinParticleThreshold", 1e-6))
        p.debug_flags = ((capi.DEBUG_POINT_CLOUD if r.get("pointCloudModeEnabled", False) else 0)
                         | (capi.DEBUG_SH_ONLY if r.get("showShOnly", False) else 0)
                         | (capi.DEBUG_OPACITY_GAUSSIAN_DISABLED if r.get("opacityGaussianDisabled", False) else 0))
        return p

    def build_scene(self, device=0):
        """load the assets, add the instances in file order and commit (needs an MI355X)"""
        from . import capi
        sets = {sid: capi.SplatSet.load(path) for sid, path in self.splat_sets.items()}
        scene = capi.Scene(device)
        for inst in self.instances:
            if inst.splat_set_id not in sets:
                continue  # "Invalid splatSetId reference" is skipped by the reader (vkgs_project_reader.cpp:268-270)
            M, _ = capi.compute_transform(inst.scale, inst.rotation, inst.position)
            scene.add_instance(sets[inst.splat_set_id], M)
        scene.commit(self.sh_format, self.rgba_format)
        return scene


def _cam_from(item):
    c = Camera()
    if "eye" in item: c.eye = np.asarray(item["eye"], np.float32)
    if "ctr" in item: c.ctr = np.asarray(item["ctr"], np.float32)
    if "up" in item: c.up = np.asarray(item["up"], np.float32)
    if "fov" in item: c.fov = float(item["fov"])
    if "clip" in item: c.clip = (float(item["clip"][0]), float(item["clip"][1]))
    if "model" in item: c.model = int(item["model"])
    # old files carry "dofEnabled" (bool), newer ones "dofMode", which wins (vkgs_project_reader.cpp:579-582)
    if "dofEnabled" in item: c.dof_mode = int(bool(item["dofEnabled"]))
    if "dofMode" in item: c.dof_mode = int(item["dofMode"])
    if "focusDist" in item: c.focus_dist = float(item["focusDist"])
    if "aperture" in item: c.aperture = float(item["aperture"])
    return c


def _cam_to(c):
    return {"model": int(c.model), "ctr": [float(x) for x in c.ctr], "eye": [float(x) for x in c.eye], "up": [float(x) for x in c.up],
            "fov": float(c.fov), "clip": [float(c.clip[0]), float(c.clip[1])], "dofMode": int(c.dof_mode),
            "focusDist": float(c.focus_dist), "aperture": float(c.aperture)}


def load_project(path):
    with open(path) as f:
        data = json.load(f)
    base = os.path.dirname(os.path.abspath(path))
    pr = Project(version=int(data.get("version", 0)))
    if pr.version > PROJECT_FILE_VERSION:
        raise ValueError(f".vkgs version {pr.version} is newer than the supported {PROJECT_FILE_VERSION}")
    pr.renderer = dict(data.get("renderer", {}))
    if "camera" in data:
        pr.camera = _cam_from(data["camera"])
    pr.cameras = [_cam_from(c) for c in data.get("cameras", [])]
    g = data.get("splatsGlobals", {})
    pr.sh_format, pr.rgba_format = int(g.get("shFormat", 0)), int(g.get("rgbaFormat", 0))
    if "splatSets" in data:
        for item in data["splatSets"]:
            pr.splat_sets[int(item["id"])] = os.path.normpath(os.path.join(base, item["path"]))
        for item in data.get("splats", []):
            inst = SplatInstance(int(item["splatSetId"]), item.get("name", ""))
            if all(k in item for k in ("position", "rotation", "scale")):
                inst.position, inst.rotation, inst.scale = tuple(item["position"]), tuple(item["rotation"]), tuple(item["scale"])
            pr.instances.append(inst)
    else:  # legacy (version 0): every splat entry carries its own path
        for i, item in enumerate(data.get("splats", [])):
            pr.splat_sets[i] = os.path.normpath(os.path.join(base, item["path"]))
            inst = SplatInstance(i, item.get("name", ""))
            if all(k in item for k in ("position", "rotation", "scale")):
                inst.position, inst.rotation, inst.scale = tuple(item["position"]), tuple(item["rotation"]), tuple(item["scale"])
            pr.instances.append(inst)
    pr.extra = {k: v for k, v in data.items()
                if k not in ("version", "renderer", "camera", "cameras", "splatsGlobals", "splatSets", "splats")}
    return pr


def save_project(pr, path):
    base = os.path.dirname(os.path.abspath(path))
    data = {"version": PROJECT_FILE_VERSION, "renderer": dict(pr.renderer), "camera": _cam_to(pr.camera),
            "cameras": [_cam_to(c) for c in pr.cameras],
            "splatsGlobals": {"shFormat": pr.sh_format, "rgbaFormat": pr.rgba_format},
            "splatSets": [{"id": sid, "path": os.path.relpath(p, base), "storage": 0, "shFormat": pr.sh_format,
                           "rgbaFormat": pr.rgba_format} for sid, p in sorted(pr.splat_sets.items())],
            "splats": [{"splatSetId": i.splat_set_id, "name": i.name, "position": list(i.position),
                        "rotation": list(i.rotation), "scale": list(i.scale)} for i in pr.instances]}
    data.update(pr.extra)
    with open(path, "w") as f:
        json.dump(data, f, indent=4)
