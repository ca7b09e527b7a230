"""ORACLE (test infrastructure only): ctypes loader for oracle/build/libkkoracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the
product (ray-tracer_amd/, libkkdirect.so) never does.
"""
import ctypes as C
import os
import subprocess
import sys
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_REPO = os.path.dirname(_HERE)
sys.path.insert(0, os.path.join(_REPO, "ray-tracer_amd"))
from kkdirect import abi  # noqa: E402  (struct layouts only)

LIB_PATH = os.path.join(_HERE, "build", "libkkoracle.so")


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        L = C.CDLL(LIB_PATH)
        L.ora_create.restype = C.c_void_p
        L.ora_create.argtypes = [C.POINTER(abi.SceneDesc), C.c_int]
        L.ora_destroy.argtypes = [C.c_void_p]
        L.ora_trace.argtypes = [C.c_void_p, abi.pf32, C.c_uint64, C.c_int, C.c_int, C.POINTER(abi.Hit), C.POINTER(C.c_uint64)]
        L.ora_intersect_record.argtypes = [C.c_void_p, abi.pf32, abi.pf32]
        L.ora_render.argtypes = [C.c_void_p, C.POINTER(abi.RenderParams), C.c_uint64, C.c_int, abi.pf32, C.POINTER(abi.Stats)]
        L.ora_li_samples.argtypes = [C.c_void_p, C.POINTER(abi.RenderParams), C.c_uint64, abi.pu32, C.c_uint64, abi.pf32]
        L.ora_camera_rays.argtypes = [C.c_void_p, abi.pf32, C.c_uint64, abi.pf32]
        L.ora_philox.argtypes = [abi.pu32, abi.pu32, abi.pu32]
        L.ora_sample_stream.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_int, abi.pf32]
        L.ora_ldsampler.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, abi.pf32]
        L.ora_ldsampler.restype = None
        L.ora_stratified.argtypes = [C.c_uint64, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, abi.pf32]
        L.ora_stratified.restype = None
        L.ora_halton.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_int, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, abi.pf32]
        L.ora_halton.restype = None
        L.ora_sobol.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.c_int32, C.c_int, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, abi.pf32]
        L.ora_sobol.restype = None
        L.ora_sobol_bits.argtypes = [C.c_uint64, C.c_uint32]; L.ora_sobol_bits.restype = C.c_uint32
        L.ora_sobol_look_up.argtypes = [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]; L.ora_sobol_look_up.restype = C.c_uint64
        L.ora_clip_aabb.argtypes = [abi.pf32, abi.pf32, abi.pf32]
        L.ora_bsdf_eval.argtypes = [C.c_void_p, C.c_int, abi.pf32, abi.pf32, abi.pf32, abi.pf32]
        L.ora_bsdf_sample.argtypes = [C.c_void_p, C.c_int, abi.pf32, abi.pf32, C.c_uint64, abi.pf32]
        L.ora_bsdf_type.restype = C.c_uint32
        L.ora_bsdf_type.argtypes = [C.c_void_p, C.c_int]
        L.ora_microfacet_sample.argtypes = [C.c_int, C.c_float, C.c_float, C.c_int, abi.pf32, abi.pf32, C.c_uint64, abi.pf32]
        L.ora_microfacet_eval.argtypes = [C.c_int, C.c_float, C.c_float, C.c_int, abi.pf32, abi.pf32, C.c_uint64, abi.pf32]
        L.ora_rough_transmittance.argtypes = [C.c_int, C.c_float, C.c_float, abi.pf32, abi.pf32]
        L.ora_rough_transmittance_at.restype = C.c_double
        L.ora_rough_transmittance_at.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double]
        L.ora_rfilter_table.argtypes = [C.c_int, C.c_float, abi.pf32, abi.pf32, abi.pf32, C.POINTER(C.c_int)]
        L.ora_sample_emitter_direct.argtypes = [C.c_void_p, abi.pf32, abi.pf32, abi.pf32, C.c_uint64, abi.pf32]
        L.ora_env_probe.argtypes = [C.c_void_p, abi.pf32, C.c_uint64, abi.pf32]
        L.ora_kd_info.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(abi.pf32)


HIT_DTYPE = np.dtype([("t", np.float32), ("u", np.float32), ("v", np.float32), ("prim", np.uint32)])


class Oracle:
    """CPU restatement of ShapeKDTree + kk DirectIntegrator for one scene."""

    def __init__(self, scene, brute_force=False):
        self.scene = scene
        self._desc = scene.desc()
        self.h = lib().ora_create(C.byref(self._desc), int(brute_force))
        if not self.h:
            raise RuntimeError("oracle: scene rejected")
        s = self._desc.sensor
        self.width, self.height = s.crop_width, s.crop_height

    def close(self):
        if self.h:
            lib().ora_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def trace(self, rays, any_hit=False, brute=False, counts=False):
        rays = np.ascontiguousarray(rays, np.float32).reshape(-1, 8)
        out = np.zeros(len(rays), HIT_DTYPE)
        cnt = (C.c_uint64 * 2)()
        lib().ora_trace(self.h, _fp(rays), len(rays), int(any_hit), int(brute), out.ctypes.data_as(C.POINTER(abi.Hit)), cnt)
        return (out, (cnt[0], cnt[1])) if counts else out

    def intersect_record(self, ray):
        ray = np.ascontiguousarray(ray, np.float32).reshape(8)
        out = np.zeros(21, np.float32)
        ok = lib().ora_intersect_record(self.h, _fp(ray), _fp(out))
        if not ok:
            return None
        return dict(p=out[0:3], uv=out[3:5], geoN=out[5:8], shN=out[8:11], dpdu=out[11:14], dpdv=out[14:17], wi=out[17:20], t=out[20])

    def camera_rays(self, sample_pos):
        sp = np.ascontiguousarray(sample_pos, np.float32).reshape(-1, 2)
        rays = np.zeros((len(sp), 8), np.float32)
        lib().ora_camera_rays(self.h, _fp(sp), len(sp), _fp(rays))
        return rays

    def render(self, spp, bsdf_samples=1, light_samples=1, hide_emitters=False, seed=1, threads=None, block_size=0,
               tile_index=0, tile_count=1, sample_first=0, integrator=0, strict_normals=False, max_depth=-1, rr_depth=5,
               sampler=0, sampler_dimension=0, sample_count_total=0):
        p = abi.RenderParams(spp, bsdf_samples, light_samples, int(hide_emitters), block_size, tile_index, tile_count, sample_first,
                             integrator, int(strict_normals), max_depth, rr_depth, sampler, sampler_dimension, sample_count_total, 0)
        film = np.zeros((self.height, self.width, 5), np.float32)
        st = abi.Stats()
        rc = lib().ora_render(self.h, C.byref(p), seed, threads or os.cpu_count() or 1, _fp(film), C.byref(st))
        if rc != 0:
            raise RuntimeError("oracle render failed: %d" % rc)
        return film, st

    def li_samples(self, pix, bsdf_samples=1, light_samples=1, hide_emitters=False, seed=1):
        """pix: n x (x, y, sampleIndex) -> n x (r, g, b, alpha, sx, sy)"""
        pix = np.ascontiguousarray(pix, np.uint32).reshape(-1, 3)
        p = abi.RenderParams(1, bsdf_samples, light_samples, int(hide_emitters), 0, 0, 1, 0)
        out = np.zeros((len(pix), 6), np.float32)
        lib().ora_li_samples(self.h, C.byref(p), seed, pix.ctypes.data_as(abi.pu32), len(pix), _fp(out))
        return out

    def bsdf_eval(self, mat, wi, wo):
        wi = np.ascontiguousarray(wi, np.float32); wo = np.ascontiguousarray(wo, np.float32)
        f = np.zeros(3, np.float32); pdf = np.zeros(1, np.float32)
        lib().ora_bsdf_eval(self.h, mat, _fp(wi), _fp(wo), _fp(f), _fp(pdf))
        return f, float(pdf[0])

    def bsdf_sample(self, mat, wi, u2):
        wi = np.ascontiguousarray(wi, np.float32); u2 = np.ascontiguousarray(u2, np.float32).reshape(-1, 2)
        out = np.zeros((len(u2), 8), np.float32)
        lib().ora_bsdf_sample(self.h, mat, _fp(wi), _fp(u2), len(u2), _fp(out))
        return out

    def bsdf_type(self, mat):
        return lib().ora_bsdf_type(self.h, mat)

    def sample_emitter_direct(self, ref, ref_n, u2):
        ref = np.ascontiguousarray(ref, np.float32); ref_n = np.ascontiguousarray(ref_n, np.float32)
        u2 = np.ascontiguousarray(u2, np.float32).reshape(-1, 2)
        out = np.zeros((len(u2), 9), np.float32)
        lib().ora_sample_emitter_direct(self.h, _fp(ref), _fp(ref_n), _fp(u2), len(u2), _fp(out))
        return out

    def env_probe(self, dirs):
        dirs = np.ascontiguousarray(dirs, np.float32).reshape(-1, 3)
        out = np.zeros((len(dirs), 4), np.float32)
        lib().ora_env_probe(self.h, _fp(dirs), len(dirs), _fp(out))
        return out

    def kd_info(self):
        a, b, d = C.c_uint64(), C.c_uint64(), C.c_int()
        lib().ora_kd_info(self.h, C.byref(a), C.byref(b), C.byref(d))
        return a.value, b.value, d.value


def philox(ctr, key):
    ctr = np.ascontiguousarray(ctr, np.uint32); key = np.ascontiguousarray(key, np.uint32)
    out = np.zeros(4, np.uint32)
    lib().ora_philox(ctr.ctypes.data_as(abi.pu32), key.ctypes.data_as(abi.pu32), out.ctypes.data_as(abi.pu32))
    return out


def sample_stream(seed, pixel, sample, n):
    out = np.zeros(n, np.float32)
    lib().ora_sample_stream(seed, pixel, sample, n, _fp(out))
    return out


def ldsampler(seed, pixel, kind, index, array_n, set_size, count):
    """values of samples 0 ... count-1: kind 0 -> (count, 2), 1 -> (count,), 2 -> (count, array_n, 2)"""
    out = np.zeros(count * (1 if kind == 1 else 2 * (array_n if kind == 2 else 1)), np.float32)
    lib().ora_ldsampler(seed, pixel, kind, index, array_n, set_size, count, _fp(out))
    return out if kind == 1 else out.reshape((count, 2) if kind == 0 else (count, array_n, 2))


def stratified(seed, pixel, kind, index, array_n, res, flat_dim, count):
    """stratified sampler, shapes as ldsampler(); flat_dim: dimension of the independent stream that jitters the request"""
    out = np.zeros(count * (1 if kind == 1 else 2 * (array_n if kind == 2 else 1)), np.float32)
    lib().ora_stratified(seed, pixel, kind, index, array_n, res, flat_dim, count, _fp(out))
    return out if kind == 1 else out.reshape((count, 2) if kind == 0 else (count, array_n, 2))


def sobol(res_x, res_y, x, y, kind, dim, array_n, n_arrays, count):
    """sobol sampler, shapes as halton()"""
    out = np.zeros(count * (1 if kind == 1 else 2 * (array_n if kind == 2 else 1)), np.float32)
    lib().ora_sobol(res_x, res_y, x, y, kind, dim, array_n, n_arrays, count, _fp(out))
    return out if kind == 1 else out.reshape((count, 2) if kind == 0 else (count, array_n, 2))


def sobol_bits(index, dim):
    return int(lib().ora_sobol_bits(index, dim))


def sobol_look_up(m, frame, px, py):
    return int(lib().ora_sobol_look_up(m, frame, px, py))


def halton(res_x, res_y, x, y, faure, kind, dim, array_n, n_arrays, count):
    """halton sampler, shapes as ldsampler(); dim = running dimension (kinds 0, 1) or array number (kind 2)"""
    out = np.zeros(count * (1 if kind == 1 else 2 * (array_n if kind == 2 else 1)), np.float32)
    lib().ora_halton(res_x, res_y, x, y, int(faure), kind, dim, array_n, n_arrays, count, _fp(out))
    return out if kind == 1 else out.reshape((count, 2) if kind == 0 else (count, array_n, 2))


def clip_aabb(tri, box):
    tri = np.ascontiguousarray(tri, np.float32).reshape(9); box = np.ascontiguousarray(box, np.float32).reshape(6)
    out = np.zeros(6, np.float32)
    lib().ora_clip_aabb(_fp(tri), _fp(box), _fp(out))
    return out.reshape(2, 3)


def microfacet_sample(distr, au, av, visible, wi, u2):
    wi = np.ascontiguousarray(wi, np.float32); u2 = np.ascontiguousarray(u2, np.float32).reshape(-1, 2)
    out = np.zeros((len(u2), 4), np.float32)
    lib().ora_microfacet_sample(distr, au, av, int(visible), _fp(wi), _fp(u2), len(u2), _fp(out))
    return out


def microfacet_eval(distr, au, av, visible, wi, m):
    wi = np.ascontiguousarray(wi, np.float32); m = np.ascontiguousarray(m, np.float32).reshape(-1, 3)
    out = np.zeros((len(m), 3), np.float32)
    lib().ora_microfacet_eval(distr, au, av, int(visible), _fp(wi), _fp(m), len(m), _fp(out))
    return out


def rough_transmittance(distr, alpha, eta):
    t = np.zeros(100, np.float32); d = np.zeros(1, np.float32)
    lib().ora_rough_transmittance(distr, alpha, eta, _fp(t), _fp(d))
    return t, float(d[0])


def rough_transmittance_at(distr, alpha, eta, cos_theta):
    return lib().ora_rough_transmittance_at(distr, alpha, eta, cos_theta)


def rfilter_table(ftype, radius, param=(0.5, 0.0)):
    p = np.asarray(param, np.float32); v = np.zeros(32, np.float32); s = np.zeros(1, np.float32); b = C.c_int()
    lib().ora_rfilter_table(ftype, radius, _fp(p), _fp(v), _fp(s), C.byref(b))
    return v, float(s[0]), b.value


# ---- film develop and output (the step right after the path; ORACLE = test infrastructure only) -----------------
def develop_film(film, pixel_format="rgb"):
    """Bitmap::convert(ESpectrumAlphaWeight -> pixel_format, float32, gamma 1) as HDRFilm::develop / MFilm::develop use it
    (reference src/films/hdrfilm.cpp:481-497, src/films/mfilm.cpp:264-265; conversion src/libcore/fmtconv.cpp:955-1034;
    Rec. 709 weights src/libcore/spectrum.cpp:229-234).  film: H x W x 5 float32 {R*w, G*w, B*w, alpha*w, w}.
    Every operation is a separate float32 numpy operation (no fusion), in the reference's order."""
    f = np.float32
    film = np.asarray(film, f)
    r, g, b, a, w = (film[..., k] for k in range(5))
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = np.where(w != 0, f(1.0) / w, w).astype(f)
    fmt = pixel_format.lower()
    if fmt in ("luminance", "luminancealpha"):
        lum = ((r * f(0.212671) + g * f(0.715160)) + b * f(0.072169)) * inv
        ch = [lum] + ([a * inv] if fmt == "luminancealpha" else [])
    elif fmt in ("rgb", "rgba"):
        ch = [r * inv, g * inv, b * inv] + ([a * inv] if fmt == "rgba" else [])
    elif fmt in ("xyz", "xyza"):
        x, y, z = r * inv, g * inv, b * inv
        ch = [(x * f(0.412453) + y * f(0.357580)) + z * f(0.180423),
              (x * f(0.212671) + y * f(0.715160)) + z * f(0.072169),
              (x * f(0.019334) + y * f(0.119193)) + z * f(0.950227)] + ([a * inv] if fmt == "xyza" else [])
    else:
        raise ValueError(pixel_format)
    return np.stack(ch, -1).astype(f)


def pfm_bytes(pixels):
    """Bitmap::writePFM (reference src/libcore/bitmap.cpp:3804-3843): header 'PF'|'Pf', 'W H', '-1' (little endian), then the
    scanlines bottom-up; an alpha channel is stripped.  pixels: H x W x {1,3,4} float32."""
    px = np.asarray(pixels, np.float32)
    if px.ndim == 2:
        px = px[..., None]
    h, w, c = px.shape
    if c not in (1, 3, 4):
        raise ValueError("writePFM(): pixel format must be ERGB, ERGBA, ELuminance")
    head = ("P%s\n%d %d\n-1\n" % ("f" if c == 1 else "F", w, h)).encode()
    body = px[::-1, :, :3] if c == 4 else px[::-1]
    return head + np.ascontiguousarray(body).astype("<f4").tobytes()


def rgbe_bytes(pixels):
    """Bitmap::writeRGBE (reference src/libcore/bitmap.cpp:3679-3736) with RGBE_FromFloat (:3492-3508) and RGBE_WriteBytes_RLE
    (:3524-3572), no metadata.  pixels: H x W x {3, 4} float32.  Pure Python: small images only."""
    px = np.asarray(pixels, np.float32)
    h, w, c = px.shape
    if c not in (3, 4):
        raise ValueError("writeRGBE(): pixel format must be ERGB or ERGBA")
    out = bytearray(("#?RGBE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (h, w)).encode())

    def from_float(d):
        mx = np.float32(max(d[0], d[1], d[2]))
        if mx < np.float32(1e-32):
            return (0, 0, 0, 0)
        m, e = np.frexp(mx)
        s = np.float32(np.float32(np.float32(m) * np.float32(256)) / mx)
        return (int(np.float32(d[0] * s)) & 255, int(np.float32(d[1] * s)) & 255, int(np.float32(d[2] * s)) & 255, (int(e) + 128) & 255)

    def rle(data):
        n = len(data); cur = 0
        while cur < n:
            beg = cur; run = 0; old = 0
            while run < 4 and beg < n:
                beg += run; old = run; run = 1
                while beg + run < n and run < 127 and data[beg] == data[beg + run]:
                    run += 1
            if old > 1 and old == beg - cur:
                out.extend((128 + old, data[cur])); cur = beg
            while cur < beg:
                k = min(beg - cur, 128)
                out.append(k); out.extend(data[cur:cur + k]); cur += k
            if run >= 4:
                out.extend((128 + run, data[beg])); cur += run

    if w < 8 or w > 0x7fff:
        for y in range(h):
            for x in range(w):
                out.extend(from_float(px[y, x]))
    else:
        for y in range(h):
            out.extend((2, 2, w >> 8, w & 0xFF))
            q = [from_float(px[y, x]) for x in range(w)]
            for i in range(4):
                rle([v[i] for v in q])
    return bytes(out)


def rgbe_decode(blob):
    """reader for the files above (RGBE_ToFloat, bitmap.cpp:3510-3518, plus the standard scanline RLE) -> H x W x 3 float32"""
    pos = blob.index(b"-Y ")
    end = blob.index(b"\n", pos)
    h, w = [int(t) for t in blob[pos:end].decode().replace("-Y", "").replace("+X", "").split()]
    p = end + 1
    img = np.zeros((h, w, 4), np.uint8)
    if w < 8 or w > 0x7fff:
        img[:] = np.frombuffer(blob[p:p + 4 * w * h], np.uint8).reshape(h, w, 4); p += 4 * w * h
    else:
        for y in range(h):
            assert blob[p] == 2 and blob[p + 1] == 2 and (blob[p + 2] << 8 | blob[p + 3]) == w
            p += 4
            for i in range(4):
                x = 0
                while x < w:
                    k = blob[p]; p += 1
                    if k > 128:
                        img[y, x:x + k - 128, i] = blob[p]; p += 1; x += k - 128
                    else:
                        img[y, x:x + k, i] = np.frombuffer(blob[p:p + k], np.uint8); p += k; x += k
    assert p == len(blob)
    f = np.where(img[..., 3:4] > 0, np.ldexp(np.float32(1.0), img[..., 3:4].astype(np.int32) - 136), 0).astype(np.float32)
    return img[..., :3].astype(np.float32) * f


_EXR_CHANNELS = {"luminance": [("Y", 0)], "luminancealpha": [("A", 1), ("Y", 0)], "rgb": [("B", 2), ("G", 1), ("R", 0)],
                 "rgba": [("A", 3), ("B", 2), ("G", 1), ("R", 0)], "xyz": [("B", 2), ("G", 1), ("R", 0)], "xyza": [("A", 3), ("B", 2), ("G", 1), ("R", 0)]}


def exr_bytes(pixels, pixel_format="rgb", component_format="float16", compression="zip", generated_by="kkdirect"):
    """Bitmap::writeOpenEXR (reference src/libcore/bitmap.cpp:3171-3318) without the OpenEXR library (third party, absent):
    the published single-part scanline layout, attributes in the order of Imf::Header's map, ZIP in blocks of 16 lines.
    numpy's float16 cast rounds to nearest even like half.h.  Parity with files written by OpenEXR itself is unpinned."""
    import struct, zlib
    px = np.asarray(pixels, np.float32)
    if px.ndim == 2:
        px = px[..., None]
    h, w, _ = px.shape
    chans = _EXR_CHANNELS[pixel_format]
    half = component_format == "float16"

    def attr(name, typ, value):
        return name.encode() + b"\0" + typ.encode() + b"\0" + struct.pack("<i", len(value)) + value

    out = bytearray(b"\x76\x2f\x31\x01" + struct.pack("<i", 2))
    out += attr("channels", "chlist", b"".join(n.encode() + b"\0" + struct.pack("<iB3xii", 1 if half else 2, 0, 1, 1) for n, _ in chans) + b"\0")
    if pixel_format in ("rgb", "rgba"):
        out += attr("chromaticities", "chromaticities", struct.pack("<8f", 0.64, 0.33, 0.30, 0.60, 0.15, 0.06, 0.3127, 0.3290))
    elif pixel_format in ("xyz", "xyza"):
        out += attr("chromaticities", "chromaticities", struct.pack("<8f", 1.0, 0.0, 0.0, 1.0, 0.0, 0.0, np.float32(1.0) / np.float32(3.0), np.float32(1.0) / np.float32(3.0)))
    comp = {"none": 0, "zip": 3}[compression]
    out += attr("compression", "compression", bytes([comp]))
    box = struct.pack("<4i", 0, 0, w - 1, h - 1)
    out += attr("dataWindow", "box2i", box) + attr("displayWindow", "box2i", box)
    out += attr("generatedBy", "string", generated_by.encode())
    out += attr("lineOrder", "lineOrder", b"\0") + attr("pixelAspectRatio", "float", struct.pack("<f", 1.0))
    out += attr("screenWindowCenter", "v2f", struct.pack("<2f", 0.0, 0.0)) + attr("screenWindowWidth", "float", struct.pack("<f", 1.0)) + b"\0"
    lines = 16 if comp == 3 else 1
    n_chunks = (h + lines - 1) // lines
    table = len(out)
    out += b"\0" * (8 * n_chunks)
    for c in range(n_chunks):
        y0, y1 = c * lines, min(h, (c + 1) * lines)
        with np.errstate(over="ignore"):                       # values beyond the half range become inf, as in half.h
            raw = b"".join(np.ascontiguousarray(px[y, :, src]).astype("<f2" if half else "<f4").tobytes() for y in range(y0, y1) for _, src in chans)
        data = raw
        if comp == 3:
            a = np.frombuffer(raw, np.uint8)
            t = np.concatenate([a[0::2], a[1::2]]).astype(np.int32)
            t[1:] = (t[1:] - t[:-1] + 384) & 255
            z = zlib.compress(t.astype(np.uint8).tobytes())
            if len(z) < len(raw):
                data = z
        struct.pack_into("<Q", out, table + 8 * c, len(out))
        out += struct.pack("<ii", y0, len(data)) + data
    return bytes(out)


def exr_decode(blob):
    """reader for single-part scanline files with no or ZIP compression -> (dict of attributes, {channel: H x W float32})"""
    import struct, zlib
    assert blob[:4] == b"\x76\x2f\x31\x01" and struct.unpack_from("<i", blob, 4)[0] == 2
    p = 8; attrs = {}
    while blob[p] != 0:
        e = blob.index(b"\0", p); name = blob[p:e].decode(); p = e + 1
        e = blob.index(b"\0", p); typ = blob[p:e].decode(); p = e + 1
        n = struct.unpack_from("<i", blob, p)[0]; p += 4
        attrs[name] = (typ, blob[p:p + n]); p += n
    p += 1
    chans = []; v = attrs["channels"][1]; q = 0
    while v[q] != 0:
        e = v.index(b"\0", q); nm = v[q:e].decode(); q = e + 1
        pt, lin, xs, ys = struct.unpack_from("<iB3xii", v, q); q += 16
        assert xs == 1 and ys == 1
        chans.append((nm, pt))
    assert [c[0] for c in chans] == sorted(c[0] for c in chans)
    x0, y0_, x1, y1_ = struct.unpack("<4i", attrs["dataWindow"][1]); w, h = x1 - x0 + 1, y1_ - y0_ + 1
    comp = attrs["compression"][1][0]; lines = 16 if comp == 3 else 1
    n_chunks = (h + lines - 1) // lines
    offs = struct.unpack_from("<%dQ" % n_chunks, blob, p)
    img = {nm: np.zeros((h, w), np.float32) for nm, _ in chans}
    end = p + 8 * n_chunks
    for c, o in enumerate(offs):
        assert o == end                                    # chunks follow the table back to back, in increasing y
        y, n = struct.unpack_from("<ii", blob, o); data = blob[o + 8:o + 8 + n]; end = o + 8 + n
        assert y == c * lines
        ny = min(h, y + lines) - y
        size = ny * sum(w * (2 if pt == 1 else 4) for _, pt in chans)
        if comp == 3 and n < size:
            t = np.frombuffer(zlib.decompress(data), np.uint8).astype(np.int32)
            t = (np.cumsum(np.concatenate([t[:1], t[1:] - 128])) & 255).astype(np.uint8)
            half_n = (size + 1) // 2
            a = np.empty(size, np.uint8); a[0::2] = t[:half_n]; a[1::2] = t[half_n:]
            data = a.tobytes()
        assert len(data) == size
        q = 0
        for yy in range(y, y + ny):
            for nm, pt in chans:
                k = w * (2 if pt == 1 else 4)
                img[nm][yy] = np.frombuffer(data[q:q + k], "<f2" if pt == 1 else "<f4").astype(np.float32); q += k
    assert end == len(blob)
    return attrs, img


# ---- on-disk mesh formats (the step right before the path; ORACLE = test infrastructure only) ---------------------
import struct as _struct
import zlib as _zlib

_SER_HEADER, _SER_V3, _SER_V4 = 0x041C, 3, 4                    # reference src/librender/trimesh.cpp:33-35
_F_NORMALS, _F_UV, _F_COLORS, _F_FACEN, _F_SINGLE, _F_DOUBLE = 0x1, 0x2, 0x8, 0x10, 0x1000, 0x2000   # trimesh.cpp:88-96


def _xform(to_world, pos, nrm, idx=None):
    """SerializedMesh's toWorld handling (reference src/shapes/serialized.cpp:183-201), float32 like the reference's Float"""
    if to_world is None or np.array_equal(np.asarray(to_world, np.float32).reshape(4, 4), np.eye(4, dtype=np.float32)):
        return pos, nrm, idx
    m = np.asarray(to_world, np.float32).reshape(4, 4)
    f = np.float32
    x = ((m[0, 0] * pos[:, 0] + m[0, 1] * pos[:, 1]) + m[0, 2] * pos[:, 2]) + m[0, 3]
    y = ((m[1, 0] * pos[:, 0] + m[1, 1] * pos[:, 1]) + m[1, 2] * pos[:, 2]) + m[1, 3]
    z = ((m[2, 0] * pos[:, 0] + m[2, 1] * pos[:, 1]) + m[2, 2] * pos[:, 2]) + m[2, 3]
    w = ((m[3, 0] * pos[:, 0] + m[3, 1] * pos[:, 1]) + m[3, 2] * pos[:, 2]) + m[3, 3]
    out = np.stack([x, y, z], 1).astype(f)
    div = w != f(1)
    out[div] = (out[div] / w[div, None]).astype(f)
    if nrm is not None:
        inv_t = np.linalg.inv(m[:3, :3].astype(np.float64)).T.astype(f)
        n = (nrm @ inv_t.T).astype(f)
        ln = np.sqrt((n * n).sum(1, keepdims=True))
        nrm = np.where(ln > 0, n / np.where(ln > 0, ln, 1), n).astype(f)
    if idx is not None and np.linalg.det(m[:3, :3].astype(np.float64)) < 0:
        idx = idx.copy(); idx[:, [0, 1]] = idx[:, [1, 0]]
    return out, nrm, idx


def load_serialized(path, shape_index=0, to_world=None):
    """TriMesh::loadCompressed + readOffsetDictionary (reference src/librender/trimesh.cpp:174-337) and the serialized plugin's
    post-processing (src/shapes/serialized.cpp:183-201)."""
    raw = open(path, "rb").read()
    fmt, version = _struct.unpack_from("<HH", raw, 0)
    if fmt != _SER_HEADER or version not in (_SER_V3, _SER_V4):
        raise ValueError("invalid serialized mesh header")
    count = _struct.unpack_from("<I", raw, len(raw) - 4)[0]
    min_size = 4 + count * (2 * 2 + 4 + 1 + 2 * 8 + 3 * 4 + 3 * 4)
    offset = 0
    if count > 0 and len(raw) >= min_size:
        if shape_index >= count:
            raise ValueError("shape index out of range")
        if version == _SER_V4:
            offset = _struct.unpack_from("<Q", raw, len(raw) - 8 * count - 4 + 8 * shape_index)[0]
        else:
            offset = _struct.unpack_from("<I", raw, len(raw) - 4 * (count + 1) + 4 * shape_index)[0]
    elif shape_index != 0:
        raise ValueError("shape index out of range")
    fmt, version = _struct.unpack_from("<HH", raw, offset)
    data = _zlib.decompressobj(15).decompress(raw[offset + 4:])
    p = 0
    flags = _struct.unpack_from("<I", data, p)[0]; p += 4
    name = ""
    if version == _SER_V4:
        e = data.index(b"\0", p); name = data[p:e].decode(errors="replace"); p = e + 1
    nv, nt = _struct.unpack_from("<QQ", data, p); p += 16
    ft = "<f8" if flags & _F_DOUBLE else "<f4"

    def floats(k):
        nonlocal p
        a = np.frombuffer(data, ft, nv * k, p).astype(np.float32).reshape(nv, k); p += nv * k * (8 if flags & _F_DOUBLE else 4)
        return a
    pos = floats(3)
    nrm = floats(3) if flags & _F_NORMALS else None
    uv = floats(2) if flags & _F_UV else None
    col = floats(3) if flags & _F_COLORS else None
    idx = np.frombuffer(data, "<u4", nt * 3, p).reshape(nt, 3).copy()
    pos, nrm, idx = _xform(to_world, pos, nrm, idx)
    return {"name": name, "positions": pos, "normals": nrm, "texcoords": uv, "colors": col, "indices": idx, "face_normals": bool(flags & _F_FACEN)}


def save_serialized(path, meshes, version=4, double=False, dictionary=True):
    """TriMesh::serialize(Stream*) (reference src/librender/trimesh.cpp:1060-1108); version 3 = the same stream without the
    name and with a 32-bit dictionary (what readOffset / loadCompressed accept, trimesh.cpp:190-191, 285-289)."""
    out = bytearray(); offsets = []
    ft = "<f8" if double else "<f4"
    for m in meshes:
        offsets.append(len(out))
        out += _struct.pack("<HH", _SER_HEADER, version)
        flags = _F_DOUBLE if double else _F_SINGLE
        flags |= (_F_NORMALS if m.get("normals") is not None else 0) | (_F_UV if m.get("texcoords") is not None else 0)
        flags |= (_F_COLORS if m.get("colors") is not None else 0) | (_F_FACEN if m.get("face_normals") else 0)
        body = _struct.pack("<I", flags)
        if version == _SER_V4:
            body += m.get("name", "").encode() + b"\0"
        body += _struct.pack("<QQ", len(m["positions"]), len(m["indices"]))
        for key in ("positions", "normals", "texcoords", "colors"):
            if m.get(key) is not None:
                body += np.ascontiguousarray(m[key]).astype(ft).tobytes()
        body += np.ascontiguousarray(m["indices"]).astype("<u4").tobytes()
        out += _zlib.compress(body)
    if dictionary:
        for o in offsets:
            out += _struct.pack("<Q" if version == _SER_V4 else "<I", o)
        out += _struct.pack("<I", len(meshes))
    open(path, "wb").write(bytes(out))


_PLY_TYPES = {"char": "i1", "int8": "i1", "uchar": "u1", "uint8": "u1", "short": "i2", "int16": "i2", "ushort": "u2", "uint16": "u2",
              "int": "i4", "int32": "i4", "uint": "u4", "uint32": "u4", "float": "f4", "float32": "f4", "double": "f8", "float64": "f8"}


def load_ply(path, to_world=None, srgb=True):
    """PLYLoader (reference src/shapes/ply.cpp:96-330, 326-470), value by value (slow; small test files only)."""
    raw = open(path, "rb").read()
    head_end = raw.index(b"end_header") + len(b"end_header")
    head_end = raw.index(b"\n", head_end) + 1
    lines = [l.strip() for l in raw[:head_end].decode("ascii", "replace").splitlines()]
    assert lines[0] == "ply"
    mode = None; elems = []
    for l in lines[1:]:
        t = l.split()
        if not t:
            continue
        if t[0] == "format":
            mode = t[1]
        elif t[0] == "element":
            elems.append([t[1], int(t[2]), []])
        elif t[0] == "property":
            elems[-1][2].append(("list", _PLY_TYPES[t[2]], _PLY_TYPES[t[3]], t[4]) if t[1] == "list" else ("scalar", _PLY_TYPES[t[1]], None, t[2]))
    body = raw[head_end:]
    if mode == "ascii":
        toks = body.split(); ti = 0

        def rd(tp):
            nonlocal ti
            v = toks[ti]; ti += 1
            return float(v) if tp[0] == "f" else int(v)
    else:
        bo = "<" if mode == "binary_little_endian" else ">"; off = 0

        def rd(tp):
            nonlocal off
            v = np.frombuffer(body, bo + tp, 1, off)[0]; off += int(tp[1])
            return v
    slots = {"x": 0, "y": 1, "z": 2, "nx": 3, "ny": 4, "nz": 5, "u": 6, "texture_u": 6, "s": 6, "v": 7, "texture_v": 7, "t": 7,
             "red": 8, "diffuse_red": 8, "green": 9, "diffuse_green": 9, "blue": 10, "diffuse_blue": 10}
    pos = nrm = uv = col = None; tris = []
    for name, count, props in elems:
        if name == "vertex":
            used = [slots.get(pn, -1) if (tp == "f4" or (tp == "u1" and slots.get(pn, -1) >= 8)) and kind == "scalar" else -1 for kind, tp, _, pn in props]
            cur = np.zeros(11, np.float32); rows = np.zeros((count, 11), np.float32)
            for v in range(count):
                for (kind, tp, it, pn), s in zip(props, used):
                    if kind == "list":
                        for _ in range(int(rd(tp))):
                            rd(it)
                        continue
                    val = rd(tp)
                    if s >= 0:
                        cur[s] = np.float32(val) / np.float32(255.0) if tp == "u1" else np.float32(val)
                rows[v] = cur
            pos = rows[:, 0:3].copy()
            nrm = rows[:, 3:6].copy() if any(3 <= s <= 5 for s in used) else None
            uv = rows[:, 6:8].copy() if any(s in (6, 7) for s in used) else None
            col = rows[:, 8:11].copy() if any(s >= 8 for s in used) else None
        elif name == "face":
            for _ in range(count):
                for kind, tp, it, pn in props:
                    if kind == "scalar":
                        rd(tp); continue
                    n = int(rd(tp)); vals = [int(rd(it)) for _ in range(n)]
                    if pn in ("vertex_indices", "vertex_index") and tp in ("u1", "u4") and it in ("i4", "u4"):
                        if n not in (3, 4):
                            raise ValueError("Only triangle and quad-based PLY meshes are supported")
                        tris.append(vals[:3])
                        if n == 4:
                            tris.append([vals[3], vals[0], vals[2]])
        else:
            for _ in range(count):
                for kind, tp, it, pn in props:
                    if kind == "scalar":
                        rd(tp)
                    else:
                        for _ in range(int(rd(tp))):
                            rd(it)
    if pos is None or not tris:
        raise ValueError("no triangles or vertices found")
    if nrm is not None:
        ident = to_world is None or np.array_equal(np.asarray(to_world, np.float32).reshape(4, 4), np.eye(4, dtype=np.float32))
        if ident:
            ln = np.sqrt((nrm * nrm).sum(1, keepdims=True)); nrm = np.where(ln > 0, nrm / np.where(ln > 0, ln, 1), nrm).astype(np.float32)
    pos, nrm, _ = _xform(to_world, pos, nrm)
    if col is not None and srgb:
        c = col
        col = np.where(c <= np.float32(0.04045), c / np.float32(12.92), ((c + np.float32(0.055)) / np.float32(1.0 + 0.055)) ** np.float32(2.4)).astype(np.float32)
    return {"name": os.path.splitext(os.path.basename(path))[0], "positions": pos, "normals": nrm, "texcoords": uv, "colors": col,
            "indices": np.array(tris, np.uint32), "face_normals": False}


def load_obj(path, to_world=None, collapse=False, flip_tex_coords=True):
    """WavefrontOBJ's reader (reference src/shapes/obj.cpp:165-349, 373-392, 586-717), written for clarity, not speed: returns
    (meshes, eof_mesh) -- every mesh the plugin creates with shapeIndex = -1, in order, as dicts with "name" and "material", and
    whether the last one was closed by the end of the file.  Raises ValueError where the plugin logs an error."""
    f32 = np.float32
    text = open(path, "rb").read().decode("latin-1")
    raw = text.split("\n")
    if raw and raw[-1] == "":
        raw.pop()                                   # std::getline does not return an empty line after the final newline
    stem = os.path.basename(path)
    if "." in stem[1:]:
        stem = stem[:stem.rindex(".")]
    pos_in = [0]

    def fetch():                                    # fetch_line, obj.cpp:165-187
        if pos_in[0] >= len(raw):
            return None
        line = raw[pos_in[0]]; pos_in[0] += 1
        if line == "":
            return line
        stripped = line.rstrip("\r\n\t ")
        if stripped.endswith("\\"):
            nxt = fetch()
            return stripped[:-1] + (nxt if nxt is not None else "")
        return stripped

    def num(tok, k):
        try:
            return f32(float(tok[k]))
        except (IndexError, ValueError):
            return f32(0)

    V, N, T, tris, geoms, seen = [], [], [], [], [], []
    name, material, geom_index, name_before = stem, "", 0, False

    def corner(tok):                                # parse(), obj.cpp:373-392
        parts = [q for q in tok.split("/") if q != ""]
        def ai(x):                                  # atoi: leading integer, 0 when there is none
            import re
            mm = re.match(r"\s*[+-]?\d+", x)
            return int(mm.group(0)) if mm else 0
        if len(parts) == 1:
            return ai(parts[0]), 0, 0
        if len(parts) == 2:
            return (ai(parts[0]), ai(parts[1]), 0) if "//" not in tok else (ai(parts[0]), 0, ai(parts[1]))
        if len(parts) == 3:
            return ai(parts[0]), ai(parts[1]), ai(parts[2])
        raise ValueError("Invalid OBJ face format!")

    def close(as_name):
        geoms.append({"name": as_name, "material": material, "tris": list(tris), "counts": (len(V), len(N), len(T))})
        del tris[:]

    while True:
        line = fetch()
        if line is None:
            break
        tok = line.split()
        if not tok:
            continue
        key = tok[0]
        if key == "v":
            V.append([num(tok, 1), num(tok, 2), num(tok, 3)])
        elif key == "vn":
            N.append([num(tok, 1), num(tok, 2), num(tok, 3)])
        elif key == "g" and not collapse:
            new_name = line[1:].strip(" \t\r\n")
            target = name if name_before else new_name
            if tris:
                if target in seen:
                    target = "%s_%i" % (target, geom_index)
                geom_index += 1; seen.append(target); close(target)
            else:
                name_before = True
            name = new_name
        elif key == "usemtl":
            if tris and not collapse:
                if name in seen:
                    name = "%s_%i" % (name, geom_index)
                geom_index += 1; seen.append(name); close(name)
                name = stem
            material = line[6:].strip(" \t\r\n")
        elif key == "vt":
            u, v = num(tok, 1), num(tok, 2)
            T.append([u, f32(1) - v if flip_tex_coords else v])
        elif key == "f":
            cs = tok[1:]
            last = ""
            first3 = []
            for k in range(3):                      # a missing token repeats the one before (the stream leaves tmp untouched)
                if k < len(cs):
                    last = cs[k]
                first3.append(corner(last))
            tris.append(tuple(first3))
            cur = list(first3)
            for extra in cs[3:]:
                cur = [cur[0], cur[2], corner(extra)]
                tris.append(tuple(cur))
    if name in seen:
        name = "%s_%i" % (stem, geom_index)
    eof_mesh = bool(tris)
    if tris:
        close(name)

    meshes = []
    for g in geoms:                                 # createMesh, obj.cpp:610-717
        nv, nn, nt = g["counts"]
        table, verts, idx = {}, [], []
        has_n = has_uv = False
        for tri in g["tris"]:
            for (vi, ti, ni) in tri:
                if vi < 0: vi += nv + 1
                if ni < 0: ni += nn + 1
                if ti < 0: ti += nt + 1
                if vi > nv or vi <= 0:
                    raise ValueError("Out of bounds: tried to access vertex %i (max: %i)" % (vi, nv))
                p = np.array([V[vi - 1]], f32); n = np.zeros((1, 3), f32); uv = (f32(0), f32(0))
                if ni != 0:
                    if ni > nn or ni < 0:
                        raise ValueError("Out of bounds: tried to access normal %i (max: %i)" % (ni, nn))
                    n = np.array([N[ni - 1]], f32); has_n = True
                if to_world is not None and not np.array_equal(np.asarray(to_world, f32).reshape(4, 4), np.eye(4, dtype=f32)):
                    p, n2, _ = _xform(to_world, p, n if ni != 0 else None)
                    if ni != 0: n = n2
                elif ni != 0:
                    ln = np.sqrt((n * n).sum(dtype=f32), dtype=f32)
                    if ln > 0: n = (n / ln).astype(f32)
                if ti != 0:
                    if ti > nt or ti < 0:
                        raise ValueError("Out of bounds: tried to access uv %i (max: %i)" % (ti, nt))
                    uv = tuple(T[ti - 1]); has_uv = True
                k = tuple(float(x) for x in p[0]) + tuple(float(x) for x in n[0]) + (float(uv[0]), float(uv[1]))
                if k not in table:
                    table[k] = len(verts); verts.append(k)
                idx.append(table[k])
        va = np.array(verts, f32).reshape(-1, 8)
        meshes.append({"name": g["name"], "material": g["material"], "positions": va[:, 0:3].copy(),
                       "normals": va[:, 3:6].copy() if has_n else None, "texcoords": va[:, 6:8].copy() if has_uv else None,
                       "colors": None, "indices": np.array(idx, np.uint32).reshape(-1, 3), "face_normals": False})
    return meshes, eof_mesh


def rebuild_topology(mesh, max_angle_deg):
    """TriMesh::rebuildTopology (reference src/librender/trimesh.cpp:472-610), the 'maxSmoothAngle' parameter of the mesh
    plugins, corner by corner in float32 like the reference's Float (slow; small test meshes only).  mesh: dict as
    load_serialized / load_ply return; -> new dict, normals dropped."""
    pos = np.asarray(mesh["positions"], np.float32); idx = np.asarray(mesh["indices"], np.uint32)
    uv = None if mesh.get("texcoords") is None else np.asarray(mesh["texcoords"], np.float32)
    col = None if mesh.get("colors") is None else np.asarray(mesh["colors"], np.float32)
    thresh = np.float32(np.cos(np.float32(max_angle_deg) * np.float32(np.pi / 180.0)))
    face_n = np.zeros((len(idx), 3), np.float32)
    entries = []                                    # multimap<Vertex, TopoData>: (key, insertion order) -> [tri, clustered]
    for i, tri in enumerate(idx):
        for j in range(3):
            v = int(tri[j])
            key = tuple(float(x) for x in pos[v]) + (tuple(float(x) for x in uv[v]) if uv is not None else (0.0, 0.0)) \
                + (tuple(float(x) for x in col[v]) if col is not None else (0.0, 0.0, 0.0))
            entries.append([key, len(entries), i, False])
        a = pos[tri[1]] - pos[tri[0]]; b = pos[tri[2]] - pos[tri[0]]
        n = np.array([a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]], np.float32)
        l = np.float32(np.sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2]))
        face_n[i] = n / l if l > np.float32(2.0 ** -128) else 0
    entries.sort(key=lambda e: (e[0], e[1]))
    new_pos, new_uv, new_col = [], [], []
    new_idx = np.full((len(idx), 3), 0xFFFFFFFF, np.uint32)
    start = 0
    while start < len(entries):
        end = start
        while end < len(entries) and entries[end][0] == entries[start][0]:
            end += 1
        for i2 in range(start, end):
            if entries[i2][3]:
                continue
            key = entries[i2][0]; n1 = face_n[entries[i2][2]]
            vid = len(new_pos)
            new_pos.append(key[0:3]); new_uv.append(key[3:5]); new_col.append(key[5:8])
            for i3 in range(i2, end):
                e = entries[i3]
                if e[3]:
                    continue
                n2 = face_n[e[2]]
                d = np.float32(np.float32(np.float32(n1[0] * n2[0]) + np.float32(n1[1] * n2[1])) + np.float32(n1[2] * n2[2]))
                if np.array_equal(n1, n2) or d > thresh:
                    for k in range(3):
                        if tuple(float(x) for x in pos[idx[e[2], k]]) == key[0:3]:
                            new_idx[e[2], k] = vid
                    e[3] = True
        start = end
    assert (new_idx != 0xFFFFFFFF).all()
    out = dict(mesh)
    out["positions"] = np.array(new_pos, np.float32).reshape(-1, 3); out["indices"] = new_idx; out["normals"] = None
    out["texcoords"] = None if uv is None else np.array(new_uv, np.float32).reshape(-1, 2)
    out["colors"] = None if col is None else np.array(new_col, np.float32).reshape(-1, 3)
    return out
