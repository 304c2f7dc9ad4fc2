"""vk_gaussian_splatting_amd — MI355X-native (gfx950) drop-in for the VK3DGSR hot path of
nvpro-samples/vk_gaussian_splatting: depth key + cull -> radix sort -> project/SH -> tile binning
-> per-pixel compositing, behind the C ABI declared in include/mgs.h.

The product is csrc/libmgs.so (hand-written HIP + C++ host).  This package is only the thin
ctypes mirror of that ABI plus the synthetic-scene generator used by tests and bench.py.
There is NO CPU fallback: importing works anywhere, but creating a Scene without the HIP
library or without a GPU raises.
"""
from .capi import (MgsError, SplatSet, Scene, Loader, FrameParams, FrameOut, SortOut, lib_path, load_library,
                   camera_lookat_perspective, compute_transform,
                   FORMAT_FLOAT32, FORMAT_FLOAT16, FORMAT_UINT8, SORT_GPU_RADIX, SORT_CPU_ASYNC, SORT_STOCHASTIC, DOF_DISABLED, DOF_FIXED_FOCUS,
                   CULL_NONE, CULL_AT_DIST, CULL_AT_RASTER, TARGET_RGBA16F, TARGET_RGBA32F,
                   ALPHA_COVERAGE, ALPHA_SUM)
from . import synth

__all__ = ["MgsError", "SplatSet", "Scene", "Loader", "FrameParams", "FrameOut", "SortOut", "lib_path", "load_library",
           "camera_lookat_perspective", "compute_transform", "synth"]
