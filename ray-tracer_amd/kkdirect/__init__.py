"""kkdirect: Python host side over the C ABI of libkkdirect.so (include/kkdirect.h).

`DirectIntegrator` mirrors the reference plugin's interface for this path -- same plugin name
("kkIntegrator", reference src/integrators/SConscript:44) and the same three parameters
(`bsdfSamples`, `lightSamples`, `hideEmitters`; reference src/integrators/kk/direct.cpp:17-21) -- and
renders through the gfx950 library.  There is no CPU fallback: without the library or a GPU every call
raises.
"""
import ctypes as C
import os
import numpy as np
from . import abi
from .scene import Scene, look_at, compute_vertex_normals  # noqa: F401

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_PKG), "lib", "libkkdirect.so")

HIT_DTYPE = np.dtype([("t", np.float32), ("u", np.float32), ("v", np.float32), ("prim", np.uint32)])


class KKDError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libkkdirect error %d: %s" % (code, msg))
        self.code = code


_lib = None


def load_library(path=None):
    """dlopen libkkdirect.so and bind every entry point the header declares."""
    global _lib
    if _lib is None or path is not None:
        p = path or os.environ.get("KKD_LIB") or LIB_PATH   # $KKD_LIB: A/B runs of experimental builds
        if not os.path.exists(p):
            raise FileNotFoundError("libkkdirect.so not built: run __graft_entry__.build() (looked at %s)" % p)
        _lib = abi.bind(C.CDLL(p))
    return _lib


def device_count():
    return load_library().kkd_device_count()


class Context:
    """One kkd_ctx: one GPU, one uploaded scene."""

    def __init__(self, device=0, seed=1, max_samples_in_flight=0, count_traversal=False, projection_test=True, one_lane=False, gpu_build=False):
        self.lib = load_library()
        cfg = abi.Config(device, (abi.FLAG_COUNT_TRAVERSAL if count_traversal else 0) | (0 if projection_test else abi.FLAG_WATERTIGHT_TEST) | (abi.FLAG_ONE_LANE if one_lane else 0) | (abi.FLAG_GPU_BUILD if gpu_build else 0),
                         seed, max_samples_in_flight)
        self.h = self.lib.kkd_create(C.byref(cfg))
        if not self.h:
            raise KKDError(abi.KKD_ERR_DEVICE, self.lib.kkd_last_error(None).decode())
        self.width = self.height = 0
        self._scene = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.kkd_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc, allow=()):
        if rc != abi.KKD_OK and rc not in allow:
            raise KKDError(rc, self.lib.kkd_last_error(self.h).decode())
        return rc

    def upload(self, scene):
        d = scene.desc()
        self._check(self.lib.kkd_scene_upload(self.h, C.byref(d)))
        self._scene = scene
        self.width, self.height = d.sensor.crop_width, d.sensor.crop_height
        return self

    def accel_info(self):
        nb, tb, nn, nt = abi.u32(), abi.u32(), abi.u64(), abi.u64()
        self._check(self.lib.kkd_accel_info(self.h, C.byref(nb), C.byref(tb), C.byref(nn), C.byref(nt)))
        return dict(node_bytes=nb.value, tri_bytes=tb.value, n_nodes=nn.value, n_leaf_tris=nt.value)

    @staticmethod
    def _params(spp, bsdf_samples, light_samples, hide_emitters, block_size, tile_index, tile_count, sample_first,
                integrator=abi.INTEGRATOR_KK, strict_normals=False, max_depth=-1, rr_depth=5,
                sampler=abi.SAMPLER_INDEPENDENT, sampler_dimension=0, sample_count_total=0):
        return abi.RenderParams(spp, bsdf_samples, light_samples, int(hide_emitters), block_size, tile_index, tile_count, sample_first,
                                integrator, int(strict_normals), max_depth, rr_depth, sampler, sampler_dimension, sample_count_total, 0)

    def set_progress_callback(self, fn):
        """fn(samples_done, samples_total) is called between batches of a render (kkd_set_progress_callback); None removes it."""
        self._progress = abi.PROGRESS_FN(lambda user, done, total: fn(done, total)) if fn else abi.PROGRESS_FN()
        self._check(self.lib.kkd_set_progress_callback(self.h, self._progress, None))

    def set_flush_callback(self, fn):
        """fn(film[H, W, 5] copy, samples_done, samples_total) answers request_flush() between two batches of a render: the film
        so far, what Mitsuba's FlushThread develops (kkd_set_flush_callback); None removes it."""
        def thunk(user, film, w, h, done, total):
            fn(np.ctypeslib.as_array(film, shape=(h, w, 5)).copy(), done, total)
        self._flush = abi.FLUSH_FN(thunk) if fn else abi.FLUSH_FN()
        self._check(self.lib.kkd_set_flush_callback(self.h, self._flush, None))

    def request_flush(self):
        """asynchronous, thread-safe, allowed inside the progress callback (kkd_request_flush)"""
        self.lib.kkd_request_flush(self.h)

    def render(self, spp, bsdf_samples=1, light_samples=1, hide_emitters=False, block_size=0, tile_index=0, tile_count=1,
               sample_first=0, integrator=abi.INTEGRATOR_KK, strict_normals=False, max_depth=-1, rr_depth=5,
               sampler=abi.SAMPLER_INDEPENDENT, sampler_dimension=0, sample_count_total=0):
        """-> (film[H, W, 5] float32 = {R*w, G*w, B*w, alpha*w, w}, abi.Stats)"""
        p = self._params(spp, bsdf_samples, light_samples, hide_emitters, block_size, tile_index, tile_count, sample_first,
                         integrator, strict_normals, max_depth, rr_depth, sampler, sampler_dimension, sample_count_total)
        film = np.zeros((self.height, self.width, 5), np.float32)
        st = abi.Stats()
        self._check(self.lib.kkd_render(self.h, C.byref(p), film.ctypes.data_as(abi.pf32), C.byref(st)))
        return film, st

    def render_device(self, dev_ptr, spp, bsdf_samples=1, light_samples=1, hide_emitters=False, block_size=0, tile_index=0,
                      tile_count=1, sample_first=0, integrator=abi.INTEGRATOR_KK, strict_normals=False, max_depth=-1, rr_depth=5,
                      sampler=abi.SAMPLER_INDEPENDENT, sampler_dimension=0, sample_count_total=0):
        """Render into caller-owned device memory (H*W*5 floats, e.g. tensor.data_ptr()); -> abi.Stats"""
        p = self._params(spp, bsdf_samples, light_samples, hide_emitters, block_size, tile_index, tile_count, sample_first,
                         integrator, strict_normals, max_depth, rr_depth, sampler, sampler_dimension, sample_count_total)
        st = abi.Stats()
        self._check(self.lib.kkd_render_device(self.h, C.byref(p), C.c_void_p(dev_ptr), C.byref(st)))
        return st

    def trace(self, rays, any_hit=False, counted=False):
        """rays: n x 8 float32 {o, mint, d, maxt} -> structured array (t, u, v, prim) [, abi.Stats]"""
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros(len(rays), HIT_DTYPE)
        po = out.ctypes.data_as(C.POINTER(abi.Hit))
        if counted:
            st = abi.Stats()
            self._check(self.lib.kkd_trace_counted(self.h, rays.ctypes.data_as(abi.pf32), len(rays), int(any_hit), po, C.byref(st)))
            return out, st
        self._check(self.lib.kkd_trace(self.h, rays.ctypes.data_as(abi.pf32), len(rays), int(any_hit), po))
        return out

    def trace_device(self, dev_rays, n, dev_out, any_hit=False):
        st = abi.Stats()
        self._check(self.lib.kkd_trace_device(self.h, C.c_void_p(dev_rays), n, int(any_hit), C.c_void_p(dev_out), C.byref(st)))
        return st

    def trace_device_noalloc(self, dev_rays, n, dev_out, any_hit=False):
        """kkd_trace_device_noalloc: the same queries without any allocation inside the call (counters in the library's
        static device array)."""
        st = abi.Stats()
        self._check(self.lib.kkd_trace_device_noalloc(self.h, C.c_void_p(dev_rays), n, int(any_hit), C.c_void_p(dev_out), C.byref(st)))
        return st

    def develop(self, film, pixel_format="rgb"):
        """HDRFilm::develop / MFilm::develop on the GPU: film H x W x 5 -> H x W x channels float32 (kkd_develop)."""
        fmt = abi.PIXEL_FORMATS[pixel_format.lower()] if isinstance(pixel_format, str) else int(pixel_format)
        film = np.ascontiguousarray(film, np.float32)
        h, w = film.shape[0], film.shape[1]
        out = np.zeros((h, w, abi.PIXEL_CHANNELS.get(fmt, 4)), np.float32)   # an unknown format is the library's to reject
        self._check(self.lib.kkd_develop(self.h, film.ctypes.data_as(abi.pf32), w, h, fmt, out.ctypes.data_as(abi.pf32)))
        return out

    def develop_device(self, dev_film, width, height, dev_out, pixel_format="rgb"):
        fmt = abi.PIXEL_FORMATS[pixel_format.lower()] if isinstance(pixel_format, str) else int(pixel_format)
        self._check(self.lib.kkd_develop_device(self.h, C.c_void_p(dev_film), width, height, fmt, C.c_void_p(dev_out)))

    def request_cancel(self):
        self.lib.kkd_request_cancel(self.h)


class MultiContext:
    """One kkd_multi: several GPUs of one node behind one handle (tile-sharded or sample-split, include/kkdirect.h)."""

    def __init__(self, devices, split="tile", seed=1, max_samples_in_flight=0, projection_test=True, gpu_build=False):
        self.lib = load_library()
        self._devs = (abi.i32 * len(devices))(*devices)
        cfg = abi.MultiConfig(len(devices), {"tile": abi.SPLIT_TILE, "sample": abi.SPLIT_SAMPLE}[split], self._devs,
                              (0 if projection_test else abi.FLAG_WATERTIGHT_TEST) | (abi.FLAG_GPU_BUILD if gpu_build else 0), 0, seed, max_samples_in_flight)
        self.h = self.lib.kkd_multi_create(C.byref(cfg))
        if not self.h:
            raise KKDError(abi.KKD_ERR_DEVICE, self.lib.kkd_multi_last_error(None).decode())
        self.width = self.height = 0
        self._scene = None

    def close(self):
        if getattr(self, "h", None):
            self.lib.kkd_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def _check(self, rc):
        if rc != abi.KKD_OK:
            raise KKDError(rc, self.lib.kkd_multi_last_error(self.h).decode())

    def upload(self, scene):
        d = scene.desc()
        self._check(self.lib.kkd_multi_scene_upload(self.h, C.byref(d)))
        self._scene = scene
        self.width, self.height = d.sensor.crop_width, d.sensor.crop_height
        return self

    def set_progress_callback(self, fn):
        """fn(work_items_done, work_items_total) over all devices (kkd_multi_set_progress_callback); None removes it."""
        self._progress = abi.PROGRESS_FN(lambda user, done, total: fn(done, total)) if fn else abi.PROGRESS_FN()
        self._check(self.lib.kkd_multi_set_progress_callback(self.h, self._progress, None))

    def set_flush_callback(self, fn):
        """fn(film[H, W, 5] copy, work_items_done, work_items_total) answers request_flush(): the job's film so far, summed over
        the devices (kkd_multi_set_flush_callback); None removes it."""
        def thunk(user, film, w, h, done, total):
            fn(np.ctypeslib.as_array(film, shape=(h, w, 5)).copy(), done, total)
        self._flush = abi.FLUSH_FN(thunk) if fn else abi.FLUSH_FN()
        self._check(self.lib.kkd_multi_set_flush_callback(self.h, self._flush, None))

    def request_flush(self):
        """asynchronous, thread-safe, allowed inside the progress callback (kkd_multi_request_flush)"""
        self.lib.kkd_multi_request_flush(self.h)

    @property
    def merge_path(self):
        return self.lib.kkd_multi_merge_path(self.h).decode()

    @property
    def diagnostics(self):
        """what kkd_multi_create found: peer access per device pair, librccl and its version (kkd_multi_diagnostics)"""
        return self.lib.kkd_multi_diagnostics(self.h).decode()

    def render(self, spp, bsdf_samples=1, light_samples=1, hide_emitters=False, block_size=0, tile_index=0, tile_count=1,
               sample_first=0, integrator=abi.INTEGRATOR_KK, strict_normals=False, max_depth=-1, rr_depth=5,
               sampler=abi.SAMPLER_INDEPENDENT, sampler_dimension=0, sample_count_total=0):
        p = Context._params(spp, bsdf_samples, light_samples, hide_emitters, block_size, tile_index, tile_count, sample_first,
                            integrator, strict_normals, max_depth, rr_depth, sampler, sampler_dimension, sample_count_total)
        film = np.zeros((self.height, self.width, 5), np.float32)
        st = abi.Stats()
        self._check(self.lib.kkd_multi_render(self.h, C.byref(p), film.ctypes.data_as(abi.pf32), C.byref(st)))
        return film, st

    def render_device(self, dev_ptr, spp, bsdf_samples=1, light_samples=1, hide_emitters=False, block_size=0, tile_index=0,
                      tile_count=1, sample_first=0, integrator=abi.INTEGRATOR_KK, strict_normals=False, max_depth=-1, rr_depth=5,
                      sampler=abi.SAMPLER_INDEPENDENT, sampler_dimension=0, sample_count_total=0):
        p = Context._params(spp, bsdf_samples, light_samples, hide_emitters, block_size, tile_index, tile_count, sample_first,
                            integrator, strict_normals, max_depth, rr_depth, sampler, sampler_dimension, sample_count_total)
        st = abi.Stats()
        self._check(self.lib.kkd_multi_render_device(self.h, C.byref(p), C.c_void_p(dev_ptr), C.byref(st)))
        return st

    def request_cancel(self):
        self.lib.kkd_multi_request_cancel(self.h)


def develop(film):
    """HDRFilm::develop's normalisation on the host (reference src/libcore/fmtconv.cpp:978-989): RGB * (w != 0 ? 1/w : 0)."""
    film = np.asarray(film, np.float32)
    w = film[..., 4:5]
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = np.where(w != 0, np.float32(1.0) / w, w).astype(np.float32)
    return (film[..., :3] * inv).astype(np.float32)


def write_pfm(path, pixels, pixel_format=None):
    """Bitmap::writePFM (reference src/libcore/bitmap.cpp:3804-3843) through kkd_write_pfm; pixels H x W [x C]."""
    px = np.ascontiguousarray(pixels, np.float32)
    if px.ndim == 2:
        px = px[..., None]
    h, w, c = px.shape
    fmt = {1: abi.PIXEL_LUMINANCE, 3: abi.PIXEL_RGB, 4: abi.PIXEL_RGBA}.get(c, -1) if pixel_format is None else abi.PIXEL_FORMATS[pixel_format.lower()]
    rc = load_library().kkd_write_pfm(os.fsencode(path), px.ctypes.data_as(abi.pf32), w, h, fmt)
    if rc != 0:
        raise KKDError(rc, "kkd_write_pfm(%s) failed" % path)


def write_rgbe(path, pixels):
    """Bitmap::writeRGBE (reference src/libcore/bitmap.cpp:3679-3736) through kkd_write_rgbe; pixels H x W x {3, 4}."""
    px = np.ascontiguousarray(pixels, np.float32)
    if px.ndim != 3 or px.shape[2] not in (3, 4):
        raise KKDError(abi.KKD_ERR_INVALID, "writeRGBE(): pixel format must be ERGB or ERGBA")
    h, w, c = px.shape
    rc = load_library().kkd_write_rgbe(os.fsencode(path), px.ctypes.data_as(abi.pf32), w, h, abi.PIXEL_RGB if c == 3 else abi.PIXEL_RGBA)
    if rc != 0:
        raise KKDError(rc, "kkd_write_rgbe(%s) failed" % path)


def write_exr(path, pixels, pixel_format=None, component_format="float16", compression="zip", generated_by=None):
    """Bitmap::writeOpenEXR (reference src/libcore/bitmap.cpp:3171-3318) through kkd_write_exr; pixels H x W [x C];
    pixel_format: name as in HDRFilm's 'pixelFormat' (default by channel count: luminance / luminanceAlpha / rgb / rgba)."""
    px = np.ascontiguousarray(pixels, np.float32)
    if px.ndim == 2:
        px = px[..., None]
    h, w, c = px.shape
    fmt = abi.PIXEL_FORMATS[pixel_format.lower()] if pixel_format else {1: abi.PIXEL_LUMINANCE, 2: abi.PIXEL_LUMINANCE_ALPHA, 3: abi.PIXEL_RGB, 4: abi.PIXEL_RGBA}.get(c, -1)
    if fmt < 0 or abi.PIXEL_CHANNELS[fmt] != c:
        raise KKDError(abi.KKD_ERR_INVALID, "writeOpenEXR(): %d channels do not fit the pixel format" % c)
    comp = {"float16": abi.COMPONENT_FLOAT16, "float32": abi.COMPONENT_FLOAT32}[component_format]
    zc = {"zip": abi.EXR_ZIP_COMPRESSION, "none": abi.EXR_NO_COMPRESSION}[compression]
    rc = load_library().kkd_write_exr(os.fsencode(path), px.ctypes.data_as(abi.pf32), w, h, fmt, comp, zc, generated_by.encode() if generated_by else None)
    if rc != 0:
        raise KKDError(rc, "kkd_write_exr(%s) failed" % path)


def write_npy(path, pixels):
    """MFilm's numpy output (reference src/films/mfilm.cpp:336-352) through kkd_write_npy; pixels H x W [x C]."""
    px = np.ascontiguousarray(pixels, np.float32)
    if px.ndim == 2:
        px = px[..., None]
    h, w, c = px.shape
    rc = load_library().kkd_write_npy(os.fsencode(path), px.ctypes.data_as(abi.pf32), w, h, c)
    if rc != 0:
        raise KKDError(rc, "kkd_write_npy(%s) failed" % path)


def write_mfilm_text(path, pixels, file_format="matlab", digits=4, variable="data"):
    """MFilm's MATLAB (its default) / Mathematica output (reference src/films/mfilm.cpp:269-334) through kkd_write_mfilm_text;
    pixels H x W [x C]; digits and variable as the plugin's parameters (mfilm.cpp:134-135)."""
    px = np.ascontiguousarray(pixels, np.float32)
    if px.ndim == 2:
        px = px[..., None]
    h, w, c = px.shape
    fmt = {"matlab": abi.MFILM_MATLAB, "mathematica": abi.MFILM_MATHEMATICA}.get(str(file_format).lower())
    if fmt is None:                                                                                # mfilm.cpp:129-132
        raise KKDError(abi.KKD_ERR_INVALID, 'The "fileFormat" parameter must either be equal to "matlab", "mathematica" or "numpy"')
    rc = load_library().kkd_write_mfilm_text(os.fsencode(path), px.ctypes.data_as(abi.pf32), w, h, c, fmt, int(digits), variable.encode())
    if rc != 0:
        raise KKDError(rc, "kkd_write_mfilm_text(%s) failed" % path)


def _mesh_to_dict(m):
    nv, nt = int(m.n_verts), int(m.n_tris)

    def arr(ptr, n, k, dt):
        return np.ctypeslib.as_array(ptr, shape=(n * k,)).astype(dt).reshape(n, k).copy() if ptr else None
    return {"name": m.name.decode(errors="replace"), "positions": arr(m.positions, nv, 3, np.float32), "normals": arr(m.normals, nv, 3, np.float32),
            "texcoords": arr(m.texcoords, nv, 2, np.float32), "colors": arr(m.colors, nv, 3, np.float32),
            "indices": arr(m.indices, nt, 3, np.uint32), "face_normals": bool(m.flags & abi.MESH_FACE_NORMALS),
            "aabb": (np.array(m.aabb_min[:], np.float32), np.array(m.aabb_max[:], np.float32))}


def _xf(to_world):
    if to_world is None:
        return None, None
    a = np.ascontiguousarray(to_world, np.float32).reshape(16)
    return a, a.ctypes.data_as(abi.pf32)


def _io_check(rc, what):
    if rc != 0:
        raise KKDError(rc, "%s: %s" % (what, load_library().kkd_last_error(None).decode()))


def serialized_mesh_count(path):
    """TriMesh::readOffsetDictionary (reference src/librender/trimesh.cpp:296-337)."""
    n = C.c_uint32(0)
    _io_check(load_library().kkd_mesh_count_serialized(os.fsencode(path), C.byref(n)), "kkd_mesh_count_serialized")
    return int(n.value)


def _smooth(lib, m, max_smooth_angle):
    """'maxSmoothAngle' (reference src/shapes/ply.cpp:124-128, serialized.cpp:204-208): TriMesh::rebuildTopology on the loaded mesh"""
    if max_smooth_angle is not None:
        _io_check(lib.kkd_mesh_rebuild_topology(C.byref(m), float(max_smooth_angle)), "kkd_mesh_rebuild_topology")


def load_serialized(path, shape_index=0, to_world=None, max_smooth_angle=None):
    """The serialized shape plugin (reference src/shapes/serialized.cpp:150-208) through kkd_mesh_load_serialized."""
    lib = load_library(); m = abi.Mesh(); keep, ptr = _xf(to_world)
    _io_check(lib.kkd_mesh_load_serialized(os.fsencode(path), shape_index, ptr, C.byref(m)), "kkd_mesh_load_serialized")
    try:
        _smooth(lib, m, max_smooth_angle)
        return _mesh_to_dict(m)
    finally:
        lib.kkd_mesh_free(C.byref(m))


def load_ply(path, to_world=None, srgb=True, max_smooth_angle=None):
    """The ply shape plugin (reference src/shapes/ply.cpp:96-330) through kkd_mesh_load_ply."""
    lib = load_library(); m = abi.Mesh(); keep, ptr = _xf(to_world)
    _io_check(lib.kkd_mesh_load_ply(os.fsencode(path), ptr, int(srgb), C.byref(m)), "kkd_mesh_load_ply")
    try:
        _smooth(lib, m, max_smooth_angle)
        return _mesh_to_dict(m)
    finally:
        lib.kkd_mesh_free(C.byref(m))


def _obj_flags(collapse, flip_tex_coords):
    return (abi.OBJ_COLLAPSE if collapse else 0) | (0 if flip_tex_coords else abi.OBJ_KEEP_TEXCOORD_V)


def obj_mesh_count(path, collapse=False):
    """-> (number of meshes the obj plugin creates from the file with shapeIndex = -1, whether the last one is closed by the end
    of the file); reference src/shapes/obj.cpp:244-334"""
    n = C.c_uint32(0); eof = C.c_uint32(0)
    _io_check(load_library().kkd_mesh_count_obj(os.fsencode(path), _obj_flags(collapse, True), C.byref(n), C.byref(eof)), "kkd_mesh_count_obj")
    return int(n.value), bool(eof.value)


def obj_shape_index_meshes(count, eof_mesh, shape_index):
    """Which of a file's meshes the plugin's 'shapeIndex' parameter creates (reference src/shapes/obj.cpp:277, 293, 332): a mesh
    closed by a "g" / "usemtl" line when its number equals shapeIndex; the mesh the end of the file closes when the number of
    meshes closed BEFORE it, minus one, equals shapeIndex -- so it comes with mesh count - 2 and never under its own number."""
    if shape_index < 0:
        return list(range(count))
    closed = count - 1 if eof_mesh else count
    out = [shape_index] if shape_index < closed else []
    if eof_mesh and closed - 1 == shape_index:
        out.append(count - 1)
    return out


def load_obj(path, mesh_index=0, to_world=None, collapse=False, flip_tex_coords=True, max_smooth_angle=None):
    """The obj shape plugin's reader (reference src/shapes/obj.cpp:165-349, 610-717) through kkd_mesh_load_obj; the returned
    dict also carries "material", the usemtl name in force (what WavefrontOBJ::addChild matches BSDF names against)."""
    lib = load_library(); m = abi.Mesh(); keep, ptr = _xf(to_world)
    mat = C.create_string_buffer(256)
    _io_check(lib.kkd_mesh_load_obj(os.fsencode(path), int(mesh_index), ptr, _obj_flags(collapse, flip_tex_coords), C.byref(m), mat, 256), "kkd_mesh_load_obj")
    try:
        _smooth(lib, m, max_smooth_angle)
        d = _mesh_to_dict(m)
        d["material"] = mat.value.decode(errors="replace")
        return d
    finally:
        lib.kkd_mesh_free(C.byref(m))


def save_serialized(path, meshes):
    """TriMesh::serialize(Stream*) (reference src/librender/trimesh.cpp:1060-1108); meshes: dicts as load_serialized returns."""
    lib = load_library(); arr = (abi.Mesh * len(meshes))(); keep = []
    for k, d in enumerate(meshes):
        m = arr[k]
        pos = np.ascontiguousarray(d["positions"], np.float32); idx = np.ascontiguousarray(d["indices"], np.uint32); keep += [pos, idx]
        m.n_verts, m.n_tris = len(pos), len(idx)
        m.positions = pos.ctypes.data_as(abi.pf32); m.indices = idx.ctypes.data_as(abi.pu32)
        for key in ("normals", "texcoords", "colors"):
            if d.get(key) is not None:
                a = np.ascontiguousarray(d[key], np.float32); keep.append(a)
                setattr(m, key, a.ctypes.data_as(abi.pf32))
        m.flags = abi.MESH_FACE_NORMALS if d.get("face_normals") else 0
        m.name = d.get("name", "").encode()[:255]
    _io_check(lib.kkd_mesh_save_serialized(os.fsencode(path), arr, len(meshes)), "kkd_mesh_save_serialized")


class IndependentSampler:
    """Host mirror of the 'independent' sampler plugin (reference src/samplers/independent.cpp:30-60): sampleCount."""
    PLUGIN_NAME = "independent"

    def __init__(self, props=None, **kw):
        props = dict(props or {}, **kw)
        self.sampleCount = int(props.pop("sampleCount", 4))
        if props:
            raise ValueError("independent: unreferenced properties %s" % sorted(props))

    def render_args(self):
        return dict(sampler=abi.SAMPLER_INDEPENDENT)


class LowDiscrepancySampler:
    """Host mirror of the 'ldsampler' plugin, Mitsuba's default (reference src/samplers/ldsampler.cpp:76-98): sampleCount is
    rounded up to a power of two with a warning, 'dimension' low-discrepancy requests per sample (default 4)."""
    PLUGIN_NAME = "ldsampler"

    def __init__(self, props=None, **kw):
        props = dict(props or {}, **kw)
        self.sampleCount = int(props.pop("sampleCount", 4))
        self.dimension = int(props.pop("dimension", 4))
        if props:
            raise ValueError("ldsampler: unreferenced properties %s" % sorted(props))
        if self.sampleCount < 1 or self.dimension < 0:
            raise ValueError("ldsampler: sampleCount must be positive, dimension non-negative")
        if self.sampleCount & (self.sampleCount - 1):
            import warnings
            rounded = 1 << self.sampleCount.bit_length()
            warnings.warn("Sample count should be a power of two -- rounding to %d" % rounded)
            self.sampleCount = rounded

    def render_args(self):
        # dimension 0 means "the default" at the C ABI; a sampler without low-discrepancy requests is the independent one
        return dict(sampler=abi.SAMPLER_LDSAMPLER, sampler_dimension=self.dimension) if self.dimension else dict(sampler=abi.SAMPLER_INDEPENDENT)


class StratifiedSampler:
    """Host mirror of the 'stratified' plugin (reference src/samplers/stratified.cpp:60-91): sampleCount is rounded up to a
    perfect square with a warning, 'dimension' stratified requests per sample (default 4)."""
    PLUGIN_NAME = "stratified"

    def __init__(self, props=None, **kw):
        props = dict(props or {}, **kw)
        desired = int(props.pop("sampleCount", 4))
        self.dimension = int(props.pop("dimension", 4))
        if props:
            raise ValueError("stratified: unreferenced properties %s" % sorted(props))
        if desired < 1 or self.dimension < 0:
            raise ValueError("stratified: sampleCount must be positive, dimension non-negative")
        res = 1
        while res * res < desired:
            res += 1
        self.resolution, self.sampleCount = res, res * res
        if self.sampleCount != desired:
            import warnings
            warnings.warn("Sample count should be a perfect square -- rounding to %d" % self.sampleCount)

    def render_args(self):
        # dimension 0 means "the default" at the C ABI; without stratified requests the sampler is passed as the independent one
        # (the reference would still hand out Latin hypercube arrays for bsdfSamples / lightSamples > 1 then)
        return dict(sampler=abi.SAMPLER_STRATIFIED, sampler_dimension=self.dimension) if self.dimension else dict(sampler=abi.SAMPLER_INDEPENDENT)


class HaltonSampler:
    """Host mirror of the 'halton' plugin (reference src/samplers/halton.cpp:113-130): sampleCount (any number), scramble = -1
    (Faure permutations, the default) or 0 (none); random permutations (scramble > 0) are drawn from Mitsuba's SFMT and are
    not available."""
    PLUGIN_NAME = "halton"

    def __init__(self, props=None, **kw):
        props = dict(props or {}, **kw)
        self.sampleCount = int(props.pop("sampleCount", 4))
        self.scramble = int(props.pop("scramble", -1))
        if props:
            raise ValueError("halton: unreferenced properties %s" % sorted(props))
        if self.sampleCount < 1:
            raise ValueError("halton: sampleCount must be positive")
        if self.scramble not in (-1, 0):
            raise ValueError("halton: scramble must be -1 (Faure permutations) or 0 (none)")

    def render_args(self):
        return dict(sampler=abi.SAMPLER_HALTON if self.scramble == -1 else abi.SAMPLER_HALTON_PLAIN)


class HammersleySampler(HaltonSampler):
    """Host mirror of the 'hammersley' plugin (reference src/samplers/hammersley.cpp:93-103): sampleCount, scramble = -1 or 0;
    no sample arrays, so bsdfSamples / lightSamples must be 1 (hammersley.cpp:296-303)."""
    PLUGIN_NAME = "hammersley"

    def render_args(self):
        return dict(sampler=abi.SAMPLER_HAMMERSLEY if self.scramble == -1 else abi.SAMPLER_HAMMERSLEY_PLAIN)


class SobolSampler:
    """Host mirror of the 'sobol' plugin (reference src/samplers/sobol.cpp:82-107): sampleCount (any number), scramble = 0 (the
    default; other values run through sampleTEA in the reference and are not offered)."""
    PLUGIN_NAME = "sobol"

    def __init__(self, props=None, **kw):
        props = dict(props or {}, **kw)
        self.sampleCount = int(props.pop("sampleCount", 4))
        self.scramble = int(props.pop("scramble", 0))
        if props:
            raise ValueError("sobol: unreferenced properties %s" % sorted(props))
        if self.sampleCount < 1:
            raise ValueError("sobol: sampleCount must be positive")
        if self.scramble != 0:
            raise ValueError("sobol: only scramble = 0 is available")

    def render_args(self):
        return dict(sampler=abi.SAMPLER_SOBOL)


class DirectIntegrator:
    """Host mirror of the reference's DirectIntegrator plugin (src/integrators/kk/direct.cpp:15-48).

    props: dict with the XML parameter names; unknown keys raise like Mitsuba's unqueried-property check."""
    PLUGIN_NAME = "kkIntegrator"
    DESCRIPTION = "KK - Direct illumination integrator"

    def __init__(self, props=None, **kw):
        props = dict(props or {}, **kw)
        self.bsdfSamples = int(props.pop("bsdfSamples", 1))
        self.lightSamples = int(props.pop("lightSamples", 1))
        self.hideEmitters = bool(props.pop("hideEmitters", False))
        if props:
            raise ValueError("kkIntegrator: unreferenced properties %s" % sorted(props))
        if self.bsdfSamples < 0 or self.lightSamples < 0:
            raise ValueError("kkIntegrator: sample counts must be non-negative")
        self._cancel_targets = []

    def serialize(self):
        """Field order of DirectIntegrator::serialize (kk/direct.cpp:31-36): size, size, bool."""
        import struct
        return struct.pack("<QQ?", self.bsdfSamples, self.lightSamples, self.hideEmitters)

    @classmethod
    def unserialize(cls, blob):
        import struct
        b, l, h = struct.unpack("<QQ?", blob)
        return cls(bsdfSamples=b, lightSamples=l, hideEmitters=h)

    def render(self, scene, spp=None, device=0, seed=1, block_size=0, tile_index=0, tile_count=1, sample_first=0, ctx=None, sampler=None):
        """Integrator::render for this path: returns (film, stats); film is what Film::put receives.  sampler: an
        IndependentSampler / LowDiscrepancySampler (its sampleCount is the spp, integrator.cpp:103-104) or None."""
        own = ctx is None
        if own:
            ctx = Context(device=device, seed=seed).upload(scene)
        self._cancel_targets.append(ctx)
        try:
            if spp is None:
                if sampler is None:
                    raise ValueError("kkIntegrator.render: spp or a sampler is needed")
                spp = sampler.sampleCount
            return ctx.render(spp, self.bsdfSamples, self.lightSamples, self.hideEmitters, block_size, tile_index, tile_count, sample_first,
                              **(sampler.render_args() if sampler is not None else {}))
        finally:
            self._cancel_targets.remove(ctx)
            if own:
                ctx.close()

    def cancel(self):
        """SamplingIntegrator::cancel (reference src/librender/integrator.cpp:90-93)."""
        for c in list(self._cancel_targets):
            c.request_cancel()


class MIDirectIntegrator:
    """Host mirror of the stock 'direct' plugin (reference src/integrators/direct/direct.cpp:91-146), a sibling of the kk
    integrator that runs on the same kernels: shadingSamples / emitterSamples / bsdfSamples /
    strictNormals / hideEmitters, power heuristic."""
    PLUGIN_NAME = "direct"
    DESCRIPTION = "Direct illumination integrator"

    def __init__(self, props=None, **kw):
        props = dict(props or {}, **kw)
        shading = int(props.pop("shadingSamples", 1))
        self.emitterSamples = int(props.pop("emitterSamples", shading))
        self.bsdfSamples = int(props.pop("bsdfSamples", shading))
        self.strictNormals = bool(props.pop("strictNormals", False))
        self.hideEmitters = bool(props.pop("hideEmitters", False))
        if props:
            raise ValueError("direct: unreferenced properties %s" % sorted(props))
        if self.emitterSamples < 0 or self.bsdfSamples < 0 or self.emitterSamples + self.bsdfSamples <= 0:
            raise ValueError("direct: emitterSamples + bsdfSamples must be > 0")     # Assert, direct.cpp:107

    def serialize(self):
        """Field order of MIDirectIntegrator::serialize (direct/direct.cpp:120-126): size, size, bool, bool."""
        import struct
        return struct.pack("<QQ??", self.emitterSamples, self.bsdfSamples, self.strictNormals, self.hideEmitters)

    @classmethod
    def unserialize(cls, blob):
        import struct
        e, b, s, h = struct.unpack("<QQ??", blob)
        return cls(emitterSamples=e, bsdfSamples=b, strictNormals=s, hideEmitters=h)

    def render(self, scene, spp, device=0, seed=1, ctx=None, **kw):
        own = ctx is None
        if own:
            ctx = Context(device=device, seed=seed).upload(scene)
        try:
            return ctx.render(spp, self.bsdfSamples, self.emitterSamples, self.hideEmitters, integrator=abi.INTEGRATOR_DIRECT,
                              strict_normals=self.strictNormals, **kw)
        finally:
            if own:
                ctx.close()


class MIPathTracer:
    """Host mirror of the stock 'path' plugin (reference src/integrators/path/path.cpp:110-118), the next sibling on the same
    kernels: parameters of MonteCarloIntegrator (src/librender/integrator.cpp:190-233) -- rrDepth (5),
    maxDepth (-1 = unbounded), strictNormals, hideEmitters."""
    PLUGIN_NAME = "path"
    DESCRIPTION = "MI path tracer"

    def __init__(self, props=None, **kw):
        props = dict(props or {}, **kw)
        self.rrDepth = int(props.pop("rrDepth", 5))
        self.maxDepth = int(props.pop("maxDepth", -1))
        self.strictNormals = bool(props.pop("strictNormals", False))
        self.hideEmitters = bool(props.pop("hideEmitters", False))
        if props:
            raise ValueError("path: unreferenced properties %s" % sorted(props))
        if self.rrDepth <= 0:
            raise ValueError("'rrDepth' must be set to a value greater than zero!")                      # integrator.cpp:220-221
        if self.maxDepth <= 0 and self.maxDepth != -1:
            raise ValueError("'maxDepth' must be set to -1 (infinite) or a value greater than zero!")    # integrator.cpp:223-224

    def serialize(self):
        """Field order of MonteCarloIntegrator::serialize (librender/integrator.cpp:229-241): int, int, bool, bool."""
        import struct
        return struct.pack("<ii??", self.rrDepth, self.maxDepth, self.strictNormals, self.hideEmitters)

    @classmethod
    def unserialize(cls, blob):
        import struct
        r, m, s, h = struct.unpack("<ii??", blob)
        return cls(rrDepth=r, maxDepth=m, strictNormals=s, hideEmitters=h)

    def render(self, scene, spp, device=0, seed=1, ctx=None, **kw):
        own = ctx is None
        if own:
            ctx = Context(device=device, seed=seed).upload(scene)
        try:
            return ctx.render(spp, 1, 1, self.hideEmitters, integrator=abi.INTEGRATOR_PATH, strict_normals=self.strictNormals,
                              max_depth=self.maxDepth, rr_depth=self.rrDepth, **kw)
        finally:
            if own:
                ctx.close()
