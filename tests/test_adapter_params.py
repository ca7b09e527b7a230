"""Parameter extraction for the Mitsuba-side adapter: kkd_parse_bsdf_stream decodes what
InstanceManager::serialize writes for the BSDFs of the path.  The byte streams are built here from the layouts of the
reference's serialize() methods (cited at each writer); the reference cannot run, so there are no captured streams:
parity of the layout is by reading src/libcore/serialization.cpp:73-87 and the plugins."""
import ctypes as C
import struct
import numpy as np
import pytest
import kkdirect
from kkdirect import abi


class Writer:
    """InstanceManager + Stream, write side (reference src/libcore/serialization.cpp:73-87, stream.cpp:222-224)"""
    def __init__(self):
        self.b = bytearray(); self.ids = {}; self.counter = 0

    def obj(self, key, cls, body):
        if key is None:
            self.b += struct.pack("<I", 0); return
        if key in self.ids:
            self.b += struct.pack("<I", self.ids[key]); return
        self.counter += 1; self.ids[key] = self.counter
        self.b += struct.pack("<I", self.counter) + cls.encode() + b"\0"
        body()

    def spectrum_tex(self, key, rgb):      # src/libhw/basicshader.cpp:29-33
        self.obj(key, "ConstantSpectrumTexture", lambda: self.b.extend(struct.pack("<3f", *rgb)))

    def float_tex(self, key, v):           # src/libhw/basicshader.cpp:46-49
        self.obj(key, "ConstantFloatTexture", lambda: self.b.extend(struct.pack("<f", v)))

    def diffuse(self, key, refl):          # src/bsdfs/diffuse.cpp:161-165
        self.obj(key, "SmoothDiffuse", lambda: (self.b.extend(b"\0"), self.spectrum_tex(key + ".r", refl)))

    def roughconductor(self, key, distr, sv, au, av, spec, eta, k):      # src/bsdfs/roughconductor.cpp:216-226
        def body():
            self.b.extend(b"\0" + struct.pack("<I?", distr, sv))
            self.float_tex(key + ".au", au); self.float_tex(key + ".au" if au == av else key + ".av", av)
            self.spectrum_tex(key + ".s", spec); self.b.extend(struct.pack("<3f3f", *eta, *k))
        self.obj(key, "RoughConductor", body)

    def roughplastic(self, key, distr, sv, spec, diff, alpha, eta, nonlinear):   # src/bsdfs/roughplastic.cpp:243-253
        def body():
            self.b.extend(b"\1" + struct.pack("<I?", distr, sv))
            self.spectrum_tex(key + ".s", spec); self.spectrum_tex(key + ".d", diff); self.float_tex(key + ".a", alpha)
            self.b.extend(struct.pack("<f?", eta, nonlinear))
        self.obj(key, "RoughPlastic", body)

    def dielectric(self, key, eta, spec, trans):                              # src/bsdfs/dielectric.cpp:176-182
        def body():
            self.b.extend(b"\0" + struct.pack("<f", eta)); self.spectrum_tex(key + ".s", spec); self.spectrum_tex(key + ".t", trans)
        self.obj(key, "SmoothDielectric", body)

    def twosided(self, key, first, second=None):                              # src/bsdfs/twosided.cpp:77-82
        def body():
            self.b.extend(b"\0"); first(); second() if second else self.obj(None, "", None)
        self.obj(key, "TwoSidedBRDF", body)


def parse(buf, cap=16):
    out = (abi.Material * cap)(); n = C.c_uint32(0)
    rc = kkdirect.load_library().kkd_parse_bsdf_stream(bytes(buf), len(buf), out, cap, C.byref(n))
    if rc != 0:
        raise kkdirect.KKDError(rc, kkdirect.load_library().kkd_last_error(None).decode())
    return [out[i] for i in range(n.value)]


def test_every_bsdf_of_the_path():
    w = Writer()
    w.twosided("ts", lambda: w.diffuse("white", (0.5, 0.6, 0.7)))
    w.roughconductor("gold", abi.DISTR_GGX, True, 0.2, 0.2, (1, 1, 1), (0.143, 0.375, 1.442), (3.983, 2.386, 1.603))
    w.roughconductor("aniso", abi.DISTR_BECKMANN, False, 0.25, 0.08, (0.9, 0.8, 0.7), (0.2, 0.92, 1.1), (3.9, 2.45, 2.14))
    w.roughplastic("plastic", abi.DISTR_BECKMANN, True, (1, 1, 1), (0.2, 0.4, 0.6), 0.1, 1.49 / 1.000277, True)
    w.dielectric("glass", 1.5046, (1, 1, 1), (0.9, 0.95, 1.0))
    w.twosided("ts2", lambda: w.roughplastic("plastic", 0, True, (1, 1, 1), (0.2, 0.4, 0.6), 0.1, 1.0, False), lambda: w.diffuse("white", (0, 0, 0)))
    m = parse(w.b)
    assert [x.type for x in m] == [abi.BSDF_TWOSIDED, abi.BSDF_DIFFUSE, abi.BSDF_ROUGHCONDUCTOR, abi.BSDF_ROUGHCONDUCTOR,
                                   abi.BSDF_ROUGHPLASTIC, abi.BSDF_DIELECTRIC, abi.BSDF_TWOSIDED]
    assert list(m[0].nested) == [1, -1] and np.allclose(list(m[1].reflectance), (0.5, 0.6, 0.7))
    assert m[2].distribution == abi.DISTR_GGX and m[2].sample_visible == 1 and np.allclose((m[2].alpha_u, m[2].alpha_v), 0.2)
    assert np.allclose(list(m[2].cond_eta), (0.143, 0.375, 1.442)) and np.allclose(list(m[2].cond_k), (3.983, 2.386, 1.603))
    assert m[3].sample_visible == 0 and np.allclose((m[3].alpha_u, m[3].alpha_v), (0.25, 0.08)) and np.allclose(list(m[3].specular_reflectance), (0.9, 0.8, 0.7))
    assert m[4].nonlinear == 1 and np.isclose(m[4].eta, 1.49 / 1.000277) and np.allclose(list(m[4].reflectance), (0.2, 0.4, 0.6)) and np.isclose(m[4].alpha_u, 0.1)
    assert np.isclose(m[5].eta, 1.5046) and np.allclose(list(m[5].specular_transmittance), (0.9, 0.95, 1.0))
    assert list(m[6].nested) == [4, 1]                     # both children were serialised before: referenced by id, not repeated


def test_parsed_materials_render_like_hand_built_ones():
    """the parsed blocks are the blocks the Scene builder would have made (same ABI struct)"""
    sc = kkdirect.Scene()
    ref = sc.materials[sc.roughplastic((0.2, 0.4, 0.6), alpha=0.1)] if hasattr(sc, "materials") else None
    w = Writer(); w.roughplastic("p", abi.DISTR_BECKMANN, True, (1, 1, 1), (0.2, 0.4, 0.6), 0.1, 1.49 / 1.000277, False)
    got = parse(w.b)[0]
    if ref is not None:
        for f in ("type", "distribution", "sample_visible", "nonlinear", "alpha_u", "alpha_v", "eta", "reflectance", "specular_reflectance"):
            a, b = getattr(got, f), getattr(ref, f)
            assert np.allclose(list(a) if hasattr(a, "__len__") else a, list(b) if hasattr(b, "__len__") else b, rtol=1e-6), f


@pytest.mark.parametrize("mutate,msg", [(lambda b: b[:len(b) - 3], "truncated"), (lambda b: b.replace(b"SmoothDiffuse", b"SmoothPlastic"), "not on the kk direct path"),
                                        (lambda b: b.replace(b"ConstantSpectrumTexture", b"BitmapTextureXXXXXXXXXXX"), "not constant")])
def test_rejects_what_the_device_path_cannot_take(mutate, msg):
    w = Writer(); w.diffuse("d", (0.5, 0.5, 0.5))
    with pytest.raises(kkdirect.KKDError, match=msg):
        parse(mutate(bytes(w.b)))
    with pytest.raises(kkdirect.KKDError, match="more materials"):
        w2 = Writer(); w2.twosided("t", lambda: w2.diffuse("d", (1, 1, 1))); parse(w2.b, cap=1)


def _adapter_source():
    import os
    root = os.path.join(os.path.dirname(__file__), "..", "adapter")
    return open(os.path.join(root, "kkintegrator_gpu.h")).read() + open(os.path.join(root, "kk_direct_additions.inl")).read()


def test_adapter_source_uses_only_declared_abi():
    """The adapter cannot be compiled here (no Mitsuba header tree); at least every kkd_* / KKD_* name and every struct
    field it touches must exist in include/kkdirect.h."""
    import os, re
    root = os.path.join(os.path.dirname(__file__), "..")
    src = _adapter_source()
    hdr = open(os.path.join(root, "include", "kkdirect.h")).read()
    for name in sorted(set(re.findall(r"\b(?:kkd_[a-z_0-9]+|KKD_[A-Z_0-9]+)\b", src))):
        assert re.search(r"\b%s\b" % name, hdr), name
    fields = {"sh": "kkd_shape", "ke": "kkd_emitter", "env": "kkd_envmap", "ks": "kkd_sensor", "sd": "kkd_scene_desc",
              "cfg": "kkd_multi_config", "rp": "kkd_render_params", "st": "kkd_stats"}
    for var, struct in fields.items():
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (struct, struct), hdr, re.S).group(1)
        for f in sorted(set(re.findall(r"\b%s\.([a-z_0-9]+)" % var, src))):
            assert re.search(r"\b%s\b" % f, body), (struct, f)


REFERENCE_SRC = "/root/reference/src"


@pytest.mark.skipif(not __import__("os").path.isdir(REFERENCE_SRC), reason="the reference checkout only exists in the build container")
def test_adapter_source_uses_names_the_reference_has():
    """The Mitsuba side of the adapter, checked by grep against the reference's sources: the class it
    extends is the one kk/direct.cpp defines, with the members that file declares, and every Mitsuba method, type and enum
    it names occurs in the reference tree."""
    import os, re, subprocess
    direct = open(os.path.join(REFERENCE_SRC, "integrators", "kk", "direct.cpp")).read()
    assert not os.path.exists(os.path.join(REFERENCE_SRC, "integrators", "kk", "direct.h"))       # hence no #include of it
    inl = open(os.path.join(os.path.dirname(__file__), "..", "adapter", "kk_direct_additions.inl")).read()
    code = re.sub(r"/\*.*?\*/", "", inl, flags=re.S)
    for member in ("brdfSamplesNo", "lightSamplesNo", "hideEmitters"):
        assert re.search(r"\b(size_t|bool) %s;" % member, direct), member
        assert re.search(r"\b%s\b" % member, code), member
    assert "class DirectIntegrator : public SamplingIntegrator" in direct and "SamplingIntegrator::cancel()" in code
    assert '"direct.h"' not in re.sub(r"/\*.*?\*/", "", _adapter_source(), flags=re.S)
    src = re.sub(r"/\*.*?\*/", "", _adapter_source(), flags=re.S)
    src = re.sub(r'"(?:[^"\\]|\\.)*"', '""', src)
    names = set(re.findall(r"(?:->|\.|::)([A-Za-z_][A-Za-z_0-9]*)\s*\(", src)) | set(re.findall(r"\b(E[A-Z][A-Za-z0-9]+)\b", src)) \
        | set(re.findall(r"\b([A-Z][A-Za-z0-9]+)\s*(?:\*|&|::|<|\()", src))
    own = {"KKDGpuState"}
    missing = [n for n in sorted(names - own)
               if subprocess.run(["grep", "-rwq", n, REFERENCE_SRC, "--include=*.cpp", "--include=*.h"]).returncode != 0]
    assert not missing, missing
