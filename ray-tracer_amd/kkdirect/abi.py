"""ctypes mirror of include/kkdirect.h (the C ABI of libkkdirect.so).

Field order and types follow the header exactly; tests/test_abi.py checks sizeof() of every struct
against the values the C compiler reports (kkd_abi_sizes).
"""
import ctypes as C

KKD_ABI_VERSION = 1

KKD_OK, KKD_ERR_INVALID, KKD_ERR_DEVICE, KKD_ERR_CANCELLED, KKD_ERR_NOMEM, KKD_ERR_UNSUPPORTED = 0, -1, -2, -3, -4, -5

BSDF_DIFFUSE, BSDF_ROUGHCONDUCTOR, BSDF_ROUGHPLASTIC, BSDF_DIELECTRIC, BSDF_TWOSIDED = range(5)
DISTR_BECKMANN, DISTR_GGX, DISTR_PHONG = range(3)
EMITTER_AREA, EMITTER_POINT, EMITTER_ENVMAP = range(3)
RFILTER_GAUSSIAN, RFILTER_BOX, RFILTER_TENT, RFILTER_MITCHELL, RFILTER_CATMULLROM, RFILTER_LANCZOS = range(6)
SHAPE_HAS_NORMALS, SHAPE_HAS_TEXCOORDS = 1, 2
FLAG_COUNT_TRAVERSAL = 1
FLAG_PROJECTION_TEST = 2
FLAG_WATERTIGHT_TEST = 16
FLAG_ONE_LANE = 4
FLAG_GPU_BUILD = 8
NOALLOC_MAX_RAYS = 1 << 30      # KKD_NOALLOC_MAX_RAYS
PIXEL_LUMINANCE, PIXEL_LUMINANCE_ALPHA, PIXEL_RGB, PIXEL_RGBA, PIXEL_XYZ, PIXEL_XYZA = range(6)
COMPONENT_FLOAT16, COMPONENT_FLOAT32 = 0, 1
EXR_NO_COMPRESSION, EXR_ZIP_COMPRESSION = 0, 3
PIXEL_CHANNELS = {PIXEL_LUMINANCE: 1, PIXEL_LUMINANCE_ALPHA: 2, PIXEL_RGB: 3, PIXEL_RGBA: 4, PIXEL_XYZ: 3, PIXEL_XYZA: 4}
PIXEL_FORMATS = {"luminance": PIXEL_LUMINANCE, "luminancealpha": PIXEL_LUMINANCE_ALPHA, "rgb": PIXEL_RGB, "rgba": PIXEL_RGBA,
                 "xyz": PIXEL_XYZ, "xyza": PIXEL_XYZA}
RAYS_PRIMARY, RAYS_BSDF, RAYS_SHADOW = 0, 1, 2
STAGE_NAMES = ("primary", "shade", "bsdf", "shadow", "resolve")

f32, u32, i32, u64 = C.c_float, C.c_uint32, C.c_int32, C.c_uint64
pf32, pu32 = C.POINTER(C.c_float), C.POINTER(C.c_uint32)


class Shape(C.Structure):
    _fields_ = [("first_prim", u32), ("n_prims", u32), ("material_id", i32), ("emitter_id", i32),
                ("flags", u32), ("reserved", u32)]


class Material(C.Structure):
    _fields_ = [("type", i32), ("distribution", i32), ("sample_visible", i32), ("nonlinear", i32),
                ("nested", i32 * 2), ("alpha_u", f32), ("alpha_v", f32), ("eta", f32),
                ("reflectance", f32 * 3), ("specular_reflectance", f32 * 3),
                ("specular_transmittance", f32 * 3), ("cond_eta", f32 * 3), ("cond_k", f32 * 3)]


class Emitter(C.Structure):
    _fields_ = [("type", i32), ("shape_id", i32), ("sampling_weight", f32), ("radiance", f32 * 3),
                ("position", f32 * 3), ("reserved", f32)]


class EnvMap(C.Structure):
    _fields_ = [("width", u32), ("height", u32), ("texels", pf32), ("scale", f32), ("to_world", f32 * 9),
                ("bsphere_center", f32 * 3), ("bsphere_radius", f32)]


class Sensor(C.Structure):
    _fields_ = [("to_world", f32 * 16), ("xfov_deg", f32), ("near_clip", f32), ("far_clip", f32),
                ("film_width", u32), ("film_height", u32), ("crop_offset_x", u32), ("crop_offset_y", u32),
                ("crop_width", u32), ("crop_height", u32), ("film_has_alpha", i32), ("rfilter_type", i32),
                ("rfilter_radius", f32), ("rfilter_param", f32 * 2), ("film_high_quality_edges", i32), ("reserved", i32)]


class SceneDesc(C.Structure):
    _fields_ = [("abi_version", u32), ("reserved0", u32), ("n_verts", u64), ("n_tris", u64),
                ("positions", pf32), ("normals", pf32), ("texcoords", pf32), ("indices", pu32),
                ("n_shapes", u32), ("shapes", C.POINTER(Shape)),
                ("n_materials", u32), ("materials", C.POINTER(Material)),
                ("n_emitters", u32), ("emitters", C.POINTER(Emitter)),
                ("envmap", EnvMap), ("sensor", Sensor)]


class Config(C.Structure):
    _fields_ = [("device", i32), ("flags", u32), ("seed", u64), ("max_samples_in_flight", u64)]


class MultiConfig(C.Structure):
    _fields_ = [("n_devices", u32), ("split", u32), ("devices", C.POINTER(i32)), ("flags", u32), ("reserved", u32),
                ("seed", u64), ("max_samples_in_flight", u64)]


SPLIT_TILE, SPLIT_SAMPLE = 0, 1

PROGRESS_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint64, C.c_uint64)
FLUSH_FN = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_float), C.c_uint32, C.c_uint32, C.c_uint64, C.c_uint64)


class RenderParams(C.Structure):
    _fields_ = [("spp", u32), ("bsdf_samples", u32), ("light_samples", u32), ("hide_emitters", i32),
                ("block_size", u32), ("tile_index", u32), ("tile_count", u32), ("sample_first", u32),
                ("integrator", u32), ("strict_normals", i32), ("max_depth", i32), ("rr_depth", i32),
                ("sampler", u32), ("sampler_dimension", u32), ("sample_count_total", u32), ("reserved", u32)]


INTEGRATOR_KK, INTEGRATOR_DIRECT, INTEGRATOR_PATH = 0, 1, 2
SAMPLER_INDEPENDENT, SAMPLER_LDSAMPLER, SAMPLER_STRATIFIED, SAMPLER_HALTON, SAMPLER_HALTON_PLAIN = 0, 1, 2, 3, 4
SAMPLER_HAMMERSLEY, SAMPLER_HAMMERSLEY_PLAIN = 5, 6
SAMPLER_SOBOL = 7
MFILM_MATLAB, MFILM_MATHEMATICA = 0, 1


class Stats(C.Structure):
    _fields_ = [("samples", u64), ("rays_closest", u64), ("rays_shadow", u64), ("seconds", C.c_double),
                ("seconds_trace", C.c_double), ("algorithmic_bytes", C.c_double),
                ("nodes_visited", u64), ("tris_tested", u64),
                ("class_rays", u64 * 3), ("class_nodes", u64 * 3), ("class_tris", u64 * 3),
                ("stage_seconds", C.c_double * 5), ("stage_launches", u64 * 5)]


class Hit(C.Structure):
    _fields_ = [("t", f32), ("u", f32), ("v", f32), ("prim", u32)]


# every entry point include/kkdirect.h declares: name -> (restype, argtypes)
class Mesh(C.Structure):
    _fields_ = [("n_verts", u64), ("n_tris", u64), ("positions", pf32), ("normals", pf32), ("texcoords", pf32), ("colors", pf32),
                ("indices", pu32), ("flags", u32), ("reserved", u32), ("aabb_min", f32 * 3), ("aabb_max", f32 * 3), ("name", C.c_char * 256)]


MESH_FACE_NORMALS = 1
OBJ_COLLAPSE, OBJ_KEEP_TEXCOORD_V = 1, 2

ENTRY_POINTS = {
    "kkd_device_count": (C.c_int, []),
    "kkd_create": (C.c_void_p, [C.POINTER(Config)]),
    "kkd_scene_upload": (C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    "kkd_render": (C.c_int, [C.c_void_p, C.POINTER(RenderParams), pf32, C.POINTER(Stats)]),
    "kkd_render_device": (C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.POINTER(Stats)]),
    "kkd_trace": (C.c_int, [C.c_void_p, pf32, u64, C.c_int, C.POINTER(Hit)]),
    "kkd_trace_device": (C.c_int, [C.c_void_p, C.c_void_p, u64, C.c_int, C.c_void_p, C.POINTER(Stats)]),
    "kkd_trace_device_noalloc": (C.c_int, [C.c_void_p, C.c_void_p, u64, C.c_int, C.c_void_p, C.POINTER(Stats)]),
    "kkd_trace_counted": (C.c_int, [C.c_void_p, pf32, u64, C.c_int, C.POINTER(Hit), C.POINTER(Stats)]),
    "kkd_accel_info": (C.c_int, [C.c_void_p, pu32, pu32, C.POINTER(u64), C.POINTER(u64)]),
    "kkd_probe_rough_transmittance": (C.c_int, [C.c_int, C.c_float, C.c_float, pf32, pf32]),
    "kkd_probe_ldsampler": (C.c_int, [u64, u32, C.c_int, u32, u32, u32, u32, pf32]),
    "kkd_probe_stratified": (C.c_int, [u64, u32, C.c_int, u32, u32, u32, u32, u32, pf32]),
    "kkd_probe_halton": (C.c_int, [u32, u32, C.c_int32, C.c_int32, C.c_int, C.c_int, u32, u32, u32, u32, pf32]),
    "kkd_probe_faure_permutation": (C.c_int, [u32, C.POINTER(C.c_uint16)]),
    "kkd_probe_sobol": (C.c_int, [u32, u32, i32, i32, C.c_int, u32, u32, u32, u32, pf32]),
    "kkd_probe_sobol_raw": (C.c_int, [u64, u32, u32, u32, u32, u32, C.POINTER(u32), C.POINTER(u64), C.POINTER(u32)]),
    "kkd_probe_rfilter": (C.c_int, [C.c_int, C.c_float, pf32, pf32, pf32, C.POINTER(C.c_int)]),
    "kkd_develop_device": (C.c_int, [C.c_void_p, C.c_void_p, u32, u32, i32, C.c_void_p]),
    "kkd_develop": (C.c_int, [C.c_void_p, pf32, u32, u32, i32, pf32]),
    "kkd_write_pfm": (C.c_int, [C.c_char_p, pf32, u32, u32, i32]),
    "kkd_write_npy": (C.c_int, [C.c_char_p, pf32, u32, u32, u32]),
    "kkd_write_mfilm_text": (C.c_int, [C.c_char_p, pf32, u32, u32, u32, C.c_int32, C.c_int32, C.c_char_p]),
    "kkd_write_rgbe": (C.c_int, [C.c_char_p, pf32, u32, u32, i32]),
    "kkd_write_exr": (C.c_int, [C.c_char_p, pf32, u32, u32, i32, i32, i32, C.c_char_p]),
    "kkd_parse_bsdf_stream": (C.c_int, [C.c_void_p, u64, C.POINTER(Material), u32, pu32]),
    "kkd_mesh_count_serialized": (C.c_int, [C.c_char_p, pu32]),
    "kkd_mesh_load_serialized": (C.c_int, [C.c_char_p, i32, pf32, C.POINTER(Mesh)]),
    "kkd_mesh_save_serialized": (C.c_int, [C.c_char_p, C.POINTER(Mesh), u32]),
    "kkd_mesh_load_ply": (C.c_int, [C.c_char_p, pf32, i32, C.POINTER(Mesh)]),
    "kkd_mesh_count_obj": (C.c_int, [C.c_char_p, u32, pu32, pu32]),
    "kkd_mesh_load_obj": (C.c_int, [C.c_char_p, u32, pf32, u32, C.POINTER(Mesh), C.c_char_p, u32]),
    "kkd_mesh_rebuild_topology": (C.c_int, [C.POINTER(Mesh), C.c_float]),
    "kkd_mesh_free": (None, [C.POINTER(Mesh)]),
    "kkd_request_cancel": (None, [C.c_void_p]),
    "kkd_set_progress_callback": (C.c_int, [C.c_void_p, PROGRESS_FN, C.c_void_p]),
    "kkd_set_flush_callback": (C.c_int, [C.c_void_p, FLUSH_FN, C.c_void_p]),
    "kkd_request_flush": (None, [C.c_void_p]),
    "kkd_multi_create": (C.c_void_p, [C.POINTER(MultiConfig)]),
    "kkd_multi_scene_upload": (C.c_int, [C.c_void_p, C.POINTER(SceneDesc)]),
    "kkd_multi_render": (C.c_int, [C.c_void_p, C.POINTER(RenderParams), pf32, C.POINTER(Stats)]),
    "kkd_multi_render_device": (C.c_int, [C.c_void_p, C.POINTER(RenderParams), C.c_void_p, C.POINTER(Stats)]),
    "kkd_multi_request_cancel": (None, [C.c_void_p]),
    "kkd_multi_set_progress_callback": (C.c_int, [C.c_void_p, PROGRESS_FN, C.c_void_p]),
    "kkd_multi_merge_path": (C.c_char_p, [C.c_void_p]),
    "kkd_multi_diagnostics": (C.c_char_p, [C.c_void_p]),
    "kkd_multi_set_flush_callback": (C.c_int, [C.c_void_p, FLUSH_FN, C.c_void_p]),
    "kkd_multi_request_flush": (None, [C.c_void_p]),
    "kkd_multi_device_count": (u32, [C.c_void_p]),
    "kkd_multi_last_error": (C.c_char_p, [C.c_void_p]),
    "kkd_multi_destroy": (None, [C.c_void_p]),
    "kkd_last_error": (C.c_char_p, [C.c_void_p]),
    "kkd_destroy": (None, [C.c_void_p]),
}


def bind(lib):
    """Attach restype/argtypes for every declared entry point; raises AttributeError if one is missing."""
    for name, (res, args) in ENTRY_POINTS.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    return lib
