"""Loads librt_hip.so (built in-tree by raytracing_c_amd/csrc/Makefile) and declares prototypes.

There is no Python or CPU implementation behind this module: if the shared library is missing
the import of `lib` raises NativeLibraryMissing.
"""
import ctypes as C
import os

from . import ctypes_abi as abi

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_LIB_PATH: A/B a differently built librt_hip.so (tools/exp_ab.sh); the default is the in-tree build.
LIB_PATH = os.environ.get("RT_LIB_PATH") or os.path.join(_HERE, "librt_hip.so")
# The diagnostic build (make diag): unit-test entry points (include/rt_hip_diag.h), the wavefront pipeline, the superseded
# kernel generations.  Tests and experiments load it BESIDE the product library as `raytracing_c_amd.diag`.
DIAG_PATH = os.path.join(_HERE, "librt_hip_diag.so")


class NativeLibraryMissing(RuntimeError):
    pass


def _preload_hip_runtime():
    """One HIP runtime per process.

    librt_hip.so needs `libamdhip64.so.7`; PyTorch-ROCm ships its own copy of that library and
    loads it by the unversioned name, so whichever is mapped first must serve both (two HSA
    runtimes in one process cannot both own the GPU: the second reports "No HIP GPUs").
    When torch is installed, map ITS copy first -- the multi-GPU launcher and the tests share
    device buffers with torch.  RT_HIP_RUNTIME=system keeps the ROCm installation's runtime
    (for processes that never import torch).
    """
    import importlib.util
    import sys
    if os.environ.get("RT_HIP_RUNTIME", "torch") != "torch":
        return
    if "torch" in sys.modules:
        return
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if not spec or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


class _Lazy:
    """Resolves the shared library on first attribute access."""

    def __init__(self, path, diag=False):
        self._dll = None
        self._path = path
        self._diag = diag

    def _load(self):
        if self._dll is not None:
            return self._dll
        if not os.path.exists(self._path):
            raise NativeLibraryMissing(
                f"{self._path} not found: build it with `make -C raytracing_c_amd/csrc{' diag' if self._diag else ''}` "
                "(or __graft_entry__.build()); there is no CPU fallback for the render path")
        _preload_hip_runtime()
        dll = C.CDLL(self._path)
        _declare(dll)
        if self._diag or hasattr(dll, "rt_test_math"):
            _declare_diag(dll)
        if self._diag and os.path.realpath(self._path) != os.path.realpath(LIB_PATH):
            # scenes are built with the PRODUCT library's material tokens (scene.py): make the diagnostic library accept them
            dll.rt_diag_set_tokens(symbol_address("disney_shader_proc"), symbol_address("debug_shader_proc"),
                                   symbol_address("sample_background"))
        self._dll = dll
        return dll

    def __getattr__(self, name):
        return getattr(self._load(), name)


def _declare(d):
    P = C.POINTER
    vp = C.c_void_p
    d.rt_last_error.restype = C.c_char_p
    d.rt_clear_error.restype = None
    d.rt_init.argtypes = [C.c_int]
    d.rt_set_seed.argtypes = [C.c_uint32]
    d.rt_set_seed.restype = None
    d.rt_get_seed.restype = C.c_uint32
    d.scene_init.argtypes = [P(abi.Scene), abi.Triangle_Slice, abi.Allocator]
    d.scene_init.restype = None
    d.scene_init_sah.argtypes = [P(abi.Scene), abi.Triangle_Slice, abi.Allocator]
    d.scene_init_sah.restype = None
    d.scene_init_gpu.argtypes = [P(abi.Scene), abi.Triangle_Slice, abi.Allocator]
    d.rt_scene_alloc.argtypes = [P(abi.Scene), abi.isize, abi.Allocator]
    d.rt_scene_alloc.restype = C.c_bool
    d.rt_scene_free.argtypes = [P(abi.Scene)]
    d.rt_scene_free.restype = None
    d.scene_load_bytes.argtypes = [abi.Byte_Slice, P(abi.Scene)]
    d.scene_load_bytes.restype = C.c_bool
    d.scene_save_bytes.argtypes = [P(abi.Scene), vp, C.c_ssize_t]
    d.scene_save_bytes.restype = C.c_ssize_t
    d.scene_file_size.argtypes = [P(abi.Scene)]
    d.scene_file_size.restype = C.c_ssize_t
    d.rt_scene_upload.argtypes = [P(abi.Scene)]
    d.rt_scene_upload.restype = vp
    d.rt_scene_release.argtypes = [vp]
    d.rt_scene_release.restype = None
    d.rt_scene_invalidate.argtypes = [P(abi.Scene)]
    d.rt_scene_invalidate.restype = None
    d.rt_scene_device_bytes.argtypes = [vp]
    d.rt_scene_device_bytes.restype = C.c_int64
    d.rt_set_camera.argtypes = [vp, P(abi.Camera)]
    d.rt_chunk_count.argtypes = [C.c_int32, C.c_int32]
    d.rt_local_chunk_count.argtypes = [C.c_int32] * 4
    d.rt_chunk_owner.argtypes = [C.c_int32] * 4
    d.rt_max_local_chunk_count.argtypes = [C.c_int32] * 3
    d.rt_local_chunk_list.argtypes = [C.c_int32] * 4 + [P(C.c_int32), C.c_int32]
    d.rt_render_accumulate.argtypes = [vp, P(abi.RT_Render_Params), vp, vp]
    d.rt_resolve.argtypes = [P(abi.RT_Render_Params), vp, vp, vp, vp, vp]
    d.rt_untile.argtypes = [C.c_int32, C.c_int32, C.c_int32, vp, vp, vp]
    d.rt_render_frame.argtypes = [P(abi.Scene), P(abi.Image), abi.isize, abi.isize, vp, vp]
    if hasattr(d, "rt_frame_begin"):                       # (absent from older builds loaded as A/B partners)
        d.rt_frame_begin.argtypes = [P(abi.Scene), P(abi.Image), abi.isize, abi.isize]
        d.rt_frame_end.argtypes = [C.c_int]
    d.rt_get_counters.argtypes = [P(abi.RT_Counters)]
    if hasattr(d, "rt_get_skipped_root_visits"):           # (absent from older builds that tools/exp_small_ab.sh loads as A/B partners)
        d.rt_get_skipped_root_visits.argtypes = [P(C.c_uint64)]
    d.rt_math_contract.restype = C.c_int
    d.rt_last_kernel_ms.restype = C.c_float
    d.rt_kernel_timing_reset.restype = None
    d.rt_kernel_timing_mean_ms.argtypes = [P(C.c_int32)]
    d.rt_kernel_timing_mean_ms.restype = C.c_float
    d.rt_set_devices.argtypes = [C.c_int32, C.c_int32]
    d.rt_device_count.restype = C.c_int32
    d.rt_scene_verify.argtypes = [P(abi.Scene)]
    d.rt_scene_touch.argtypes = [P(abi.Scene), vp, C.c_size_t]
    d.rt_scene_set_static.argtypes = [P(abi.Scene), C.c_int32]
    d.rt_scene_set_static.restype = None
    d.rt_get_frame_timing.argtypes = [P(abi.RT_Frame_Timing)]
    d.render_thread_proc.argtypes = [P(abi.Rendering_Context)]
    d.render_thread_proc.restype = None
    d.rendering_context_is_finished.argtypes = [P(abi.Rendering_Context)]
    d.rendering_context_is_finished.restype = C.c_bool
    d.rendering_context_finish.argtypes = [P(abi.Rendering_Context)]
    d.rendering_context_finish.restype = None
    d.render.argtypes = [P(abi.Scene), P(abi.Image), abi.isize, abi.isize]
    d.lightmap_bake.argtypes = [P(abi.Image), P(abi.Scene), abi.isize]
    d.lightmap_bake.restype = None
    d.denoise_image.argtypes = [P(abi.Image), P(abi.Image), abi.isize]
    d.denoise_image.restype = None
    d.rt_denoise.argtypes = [C.c_int32, C.c_int32, vp, vp, vp]


def _declare_diag(d):
    """include/rt_hip_diag.h"""
    vp = C.c_void_p
    d.rt_test_math.argtypes = [C.c_int32, C.c_int32, vp, vp, vp]
    d.rt_test_rcp_sweep.argtypes = [vp]
    d.rt_test_srgb_sweep.argtypes = [vp]
    d.rt_test_quantize_sweep.argtypes = [vp]
    d.rt_test_trace.argtypes = [vp, C.c_int32, vp, vp, vp, vp]
    d.rt_test_texture.argtypes = [vp, C.c_int32, C.c_int32, vp, vp]
    d.rt_test_trace_stream.argtypes = [vp, C.c_int32, vp, vp, C.c_int32, C.c_int32, vp, vp, vp, vp]
    d.rt_test_tile_order.argtypes = [C.c_int32, vp, vp]
    d.rt_set_pipeline.argtypes = [C.c_int32]
    d.rt_get_pipeline.restype = C.c_int32
    d.rt_set_wavefront_capacity.argtypes = [C.c_int64]
    d.rt_set_wavefront_capacity.restype = None
    d.rt_get_sched_stats.argtypes = [vp]
    d.rt_get_wave_times.argtypes = [vp, C.c_int32]
    d.rt_get_ledger.argtypes = [vp, C.c_int32]
    d.rt_diag_multi_fault.argtypes = [C.c_int32, C.c_int32]
    d.rt_diag_multi_fault.restype = None
    d.rt_diag_set_tokens.argtypes = [vp, vp, vp]
    d.rt_diag_set_tokens.restype = None


lib = _Lazy(LIB_PATH)
diag = _Lazy(DIAG_PATH, diag=True)


def symbol_address(name):
    """Address of an exported function (used as Shader.proc / Background.proc token)."""
    return C.cast(getattr(lib, name), C.c_void_p).value


def last_error(which=None):
    return ((which or lib).rt_last_error() or b"").decode("utf-8", "replace")
