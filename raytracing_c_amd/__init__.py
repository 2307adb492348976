"""MI355X-native render hot path behind the raytracing_c scene.h / raytracer.h boundary.

Python here is plumbing only (ctypes mirror of include/*.h, asset loading, the
multi-GPU launcher); every pixel is computed by librt_hip.so on gfx950.
"""
from . import ctypes_abi as abi          # noqa: F401
from .native import lib, diag, NativeLibraryMissing, last_error   # noqa: F401
from .scene import HostScene, build_scene, Material   # noqa: F401
from .background import procedural_background   # noqa: F401
from .loaders import load_model, default_camera, camera_from_trs   # noqa: F401
from .render import render_frame, render_context, frame_begin, frame_end, Counters   # noqa: F401
