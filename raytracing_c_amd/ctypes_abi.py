"""ctypes mirror of include/rt_types.h, rt_scene.h, rt_raytracer.h, rt_materials.h, rt_hip.h.

Field order and sizes are asserted against the values SURVEY.md section 8 quotes for the
reference structs (scene.h:37-97, raytracer.h:23-49, driver.c:191-198).
"""
import ctypes as C

import numpy as np

isize = C.c_ssize_t
f32 = C.c_float


class Vec2(C.Structure):
    _fields_ = [("x", f32), ("y", f32)]


class Vec3(C.Structure):
    _fields_ = [("x", f32), ("y", f32), ("z", f32)]


class Matrix_4x4(C.Structure):
    _fields_ = [("rows", (f32 * 4) * 4)]


class Byte_Slice(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", isize)]


class Image(C.Structure):  # rt_types.h (driver.c:747-754)
    _fields_ = [("components", isize), ("pixel_type", C.c_int), ("width", isize),
                ("stride", isize), ("height", isize), ("pixels", Byte_Slice)]


class Allocator(C.Structure):
    _fields_ = [("proc", C.c_void_p), ("user", C.c_void_p)]


class Shader_Input(C.Structure):  # scene.h:19-22
    _fields_ = [("direction", Vec3), ("normal", Vec3), ("normal_geo", Vec3), ("tangent", Vec3),
                ("bitangent", Vec3), ("position", Vec3), ("tex_coords", Vec2)]


class Shader_Output(C.Structure):  # scene.h:24-28
    _fields_ = [("direction", Vec3), ("tint", Vec3), ("emission", Vec3), ("terminate", C.c_bool)]


class Shader(C.Structure):  # scene.h:32-35
    _fields_ = [("data", C.c_void_p), ("proc", C.c_void_p)]


class Background(C.Structure):  # scene.h:67-70
    _fields_ = [("proc", C.c_void_p), ("data", C.c_void_p)]


class Camera(C.Structure):  # scene.h:14-17
    _fields_ = [("view_matrix", Matrix_4x4), ("fov", f32), ("focal_length", f32)]


class Triangle(C.Structure):  # scene.h:37-42
    _fields_ = [("positions", Vec3 * 3), ("normals", Vec3 * 3), ("tex_coords", Vec2 * 3), ("shader", Shader)]


class Triangle_Slice(C.Structure):
    _fields_ = [("data", C.c_void_p), ("len", isize)]


class Triangle_AOS(C.Structure):  # scene.h:46-51
    _fields_ = [("normal", Vec3), ("normal_a", Vec3), ("normal_b", Vec3), ("normal_c", Vec3),
                ("tangent", Vec3), ("bitangent", Vec3),
                ("tex_coords_a", Vec2), ("tex_coords_b", Vec2), ("tex_coords_c", Vec2), ("shader", Shader)]


class Triangles(C.Structure):  # scene.h:53-60
    _fields_ = [("x", C.POINTER(f32) * 3), ("y", C.POINTER(f32) * 3), ("z", C.POINTER(f32) * 3),
                ("aos", C.POINTER(Triangle_AOS)), ("len", C.c_int32)]


class BVH_Node(C.Structure):  # scene.h:72-76
    _fields_ = [("min_x", f32 * 8), ("min_y", f32 * 8), ("min_z", f32 * 8),
                ("max_x", f32 * 8), ("max_y", f32 * 8), ("max_z", f32 * 8)]


class Node_Slice(C.Structure):
    _fields_ = [("data", C.POINTER(BVH_Node)), ("len", isize)]


class BVH(C.Structure):  # scene.h:86-90
    _fields_ = [("nodes", Node_Slice), ("depth", isize), ("last_row_offset", isize)]


class Scene(C.Structure):  # scene.h:92-97
    _fields_ = [("bvh", BVH), ("camera", Camera), ("triangles", Triangles), ("background", Background)]


class Ray(C.Structure):  # raytracer.h:23-26
    _fields_ = [("position", Vec3), ("direction", Vec3)]


class Hit(C.Structure):  # raytracer.h:28-33
    _fields_ = [("distance", f32), ("normal", Vec3), ("normal_geo", Vec3), ("point", Vec3),
                ("tangent", Vec3), ("bitangent", Vec3), ("tex_coords", Vec2), ("shader", Shader)]


class Rendering_Context(C.Structure):  # raytracer.h:44-49
    _fields_ = [("image", Image), ("scene", C.POINTER(Scene)), ("samples", isize), ("max_bounces", isize),
                ("n_threads", C.c_int32), ("_current_chunk", C.c_int32)]


class PBR_Shader_Data(C.Structure):  # driver.c:191-198
    _fields_ = [("base_color", Vec3), ("emission", Vec3),
                ("roughness", f32), ("metalness", f32), ("normal_map_strength", f32),
                ("sheen", f32), ("sheen_tint", f32), ("anisotropic_strength", f32),
                ("texture_albedo", C.POINTER(Image)), ("texture_normal", C.POINTER(Image)),
                ("texture_metal_roughness", C.POINTER(Image)), ("texture_emission", C.POINTER(Image))]


class RT_Counters(C.Structure):  # rt_hip.h
    _fields_ = [(n, C.c_uint64) for n in
                ("paths", "rays", "node_visits", "leaf_visits", "shades", "backgrounds", "textured")]


class RT_Frame_Timing(C.Structure):  # rt_hip.h
    _fields_ = [(n, C.c_float) for n in
                ("stamp_ms", "upload_ms", "enqueue_ms", "gpu_prep_ms", "gpu_path_ms", "gpu_resolve_ms", "gpu_copy_ms", "total_ms",
                 "verify_ms", "gather_ms")] + [("n_devices", C.c_int32), ("slowest_device", C.c_int32)]


class RT_Render_Params(C.Structure):  # rt_hip.h
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("samples", C.c_int32), ("max_bounces", C.c_int32),
                ("seed", C.c_uint32), ("rank", C.c_int32), ("world", C.c_int32), ("slab", C.c_int32),
                ("flags", C.c_int32), ("sample_first", C.c_int32), ("sample_count", C.c_int32)]


assert C.sizeof(BVH_Node) == 192
assert C.sizeof(Triangle) == 112
assert C.sizeof(Triangle_AOS) == 112
assert C.sizeof(Hit) == 88
assert C.sizeof(PBR_Shader_Data) == 80

# numpy view of Triangle[] (112 bytes each) for bulk filling
TRIANGLE_DTYPE = np.dtype([("positions", "<f4", (3, 3)), ("normals", "<f4", (3, 3)), ("tex_coords", "<f4", (3, 2)),
                           ("shader_data", "<u8"), ("shader_proc", "<u8")])
assert TRIANGLE_DTYPE.itemsize == 112

# every symbol include/*.h declares for librt_hip.so
EXPORTED_SYMBOLS = [
    # rt_raytracer.h (reference raytracer.h:51-56)
    "render_thread_proc", "rendering_context_is_finished", "rendering_context_finish", "lightmap_bake", "render",
    "denoise_image",
    # rt_scene.h (reference scene.h:101)
    "scene_init", "scene_init_sah", "scene_init_gpu", "rt_scene_alloc", "rt_scene_free", "scene_load_bytes", "scene_save_bytes", "scene_file_size",
    # rt_materials.h (reference driver.c:95,350,411)
    "disney_shader_proc", "debug_shader_proc", "sample_background",
    # rt_hip.h
    "rt_last_error", "rt_clear_error", "rt_init", "rt_set_seed", "rt_get_seed",
    "rt_set_devices", "rt_device_count", "rt_math_contract",
    "rt_scene_verify", "rt_scene_touch", "rt_scene_set_static", "rt_get_frame_timing",
    "rt_scene_upload", "rt_scene_release", "rt_scene_invalidate", "rt_scene_device_bytes", "rt_set_camera",
    "rt_chunk_count", "rt_chunk_owner", "rt_local_chunk_count", "rt_max_local_chunk_count", "rt_local_chunk_list", "rt_render_accumulate", "rt_resolve", "rt_untile",
    "rt_denoise", "rt_render_frame", "rt_frame_begin", "rt_frame_end", "rt_get_counters", "rt_get_skipped_root_visits", "rt_last_kernel_ms", "rt_kernel_timing_reset", "rt_kernel_timing_mean_ms",
]

# include/rt_hip_diag.h: exported by librt_hip_diag.so only (which also exports everything above); the product library must
# NOT carry them (tests/test_abi.py)
DIAG_ONLY_SYMBOLS = [
    "rt_diag_set_tokens", "rt_diag_multi_fault", "rt_set_pipeline", "rt_get_pipeline", "rt_set_wavefront_capacity", "rt_get_sched_stats", "rt_get_wave_times", "rt_get_ledger",
    "rt_test_math", "rt_test_rcp_sweep", "rt_test_srgb_sweep", "rt_test_quantize_sweep", "rt_test_trace", "rt_test_trace_stream",
    "rt_test_tile_order", "rt_test_texture",
]
