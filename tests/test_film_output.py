"""Film develop and output (the step right after the path).
CPU tests: the host-only writers against the oracle's byte-level restatement of Bitmap::writePFM and against numpy's own
.npy reader.  GPU tests: kkd_develop / kkd_develop_device bit-exact against the oracle's restatement of
Bitmap::convert(ESpectrumAlphaWeight -> fmt)."""
import numpy as np
import pytest
import kkoracle
import kkdirect
from kkdirect import abi, scenes

FORMATS = ["luminance", "luminanceAlpha", "rgb", "rgba", "xyz", "xyza"]


def _film(h, w, seed):
    rng = np.random.RandomState(seed)
    wgt = rng.rand(h, w, 1).astype(np.float32) * 40
    wgt[rng.rand(h, w) < 0.1] = 0.0                       # pixels no sample touched: invWeight = 0
    rgb = (rng.rand(h, w, 3).astype(np.float32) ** 4) * 1e3
    alpha = rng.rand(h, w, 1).astype(np.float32)
    film = np.concatenate([rgb * wgt, alpha * wgt, wgt], -1).astype(np.float32)
    film[0, 0] = [1, 2, 3, 1, 0]                           # colour without weight stays colour * 0
    return film


@pytest.mark.parametrize("shape", [(7, 5, 3), (4, 9, 1), (3, 3, 4), (1, 1, 3)])
def test_write_pfm_bytes(tmp_path, shape):
    img = np.random.RandomState(1).randn(*shape).astype(np.float32)
    path = tmp_path / "out.pfm"
    kkdirect.write_pfm(str(path), img)
    assert path.read_bytes() == kkoracle.pfm_bytes(img)


@pytest.mark.parametrize("shape", [(5, 12, 3), (4, 9, 4), (3, 5, 3), (2, 300, 3)])
def test_write_rgbe_bytes(tmp_path, shape):
    """HDRFilm fileFormat = rgbe (reference src/libcore/bitmap.cpp:3679-3736): byte-equal with the restatement of the
    reference's writer, for RLE scanlines (runs, short runs, literals over 128 bytes) and the flat layout of narrow images;
    decoded again, every pixel is within the format's resolution (8-bit mantissa under the shared exponent)"""
    rng = np.random.RandomState(sum(shape))
    img = (rng.rand(*shape) ** 4 * 50).astype(np.float32)
    img[0, : shape[1] // 2] = 0.25                      # a long run
    img[1, 2:4] = img[1, 1]                             # a short run before literals
    img[-1, -1, :3] = 0                                 # a black pixel: exponent byte 0
    path = tmp_path / "x.rgbe"
    kkdirect.write_rgbe(str(path), img)
    blob = path.read_bytes()
    assert blob == kkoracle.rgbe_bytes(img)
    back = kkoracle.rgbe_decode(blob)
    mx = img[..., :3].max(-1, keepdims=True)
    assert (np.abs(back - img[..., :3]) <= mx / 128 + 1e-30).all() and (back <= img[..., :3] + 1e-30).all()   # truncating quantiser
    assert blob.startswith(b"#?RGBE\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n" % (shape[0], shape[1]))


def test_write_rgbe_rejects_other_formats(tmp_path):
    with pytest.raises(kkdirect.KKDError):
        kkdirect.write_rgbe(str(tmp_path / "x.rgbe"), np.zeros((2, 2, 1), np.float32))


@pytest.mark.parametrize("fmt,ch", [("rgb", 3), ("rgba", 4), ("luminance", 1), ("luminancealpha", 2), ("xyz", 3), ("xyza", 4)])
@pytest.mark.parametrize("component", ["float16", "float32"])
@pytest.mark.parametrize("compression", ["zip", "none"])
def test_write_exr(tmp_path, fmt, ch, component, compression):
    """HDRFilm's default output (reference src/libcore/bitmap.cpp:3171-3318, src/films/hdrfilm.cpp:213-223).  The OpenEXR library
    is absent, and so is any third-party reader: the file is compared byte for byte with the oracle's independent writer of
    the published layout and read back with the oracle's reader -- header attributes, channel order, offsets, ZIP blocks of
    16 lines with a partial last block, values equal to the float16 / float32 rounding of the input.  Parity unpinned."""
    rng = np.random.RandomState(ch)
    img = (rng.rand(37, 21, ch) ** 3 * 20).astype(np.float32)
    img[5:30, :, 0] = 0.5                                   # compressible rows
    img[0, 0] = 0; img[1, 1] = 70000.0                      # zero and a value beyond the half range (-> inf in float16)
    path = tmp_path / "x.exr"
    kkdirect.write_exr(str(path), img, pixel_format=fmt, component_format=component, compression=compression, generated_by="test")
    blob = path.read_bytes()
    assert blob == kkoracle.exr_bytes(img, fmt, component, compression, generated_by="test")
    attrs, chans = kkoracle.exr_decode(blob)
    names = {"rgb": "RGB", "rgba": "RGBA", "luminance": "Y", "luminancealpha": "YA", "xyz": "RGB", "xyza": "RGBA"}[fmt]
    assert sorted(chans) == sorted(names)
    with np.errstate(over="ignore"):
        want = img.astype(np.float16).astype(np.float32) if component == "float16" else img
    for k, nm in enumerate(names):
        np.testing.assert_array_equal(chans[nm], want[..., k])
    assert attrs["generatedBy"] == ("string", b"test") and attrs["lineOrder"][1] == b"\0" and attrs["compression"][1][0] == (3 if compression == "zip" else 0)
    assert ("chromaticities" in attrs) == (ch >= 3)
    if compression == "zip":
        assert len(blob) < len(kkoracle.exr_bytes(img, fmt, component, "none", generated_by="test"))


def test_write_exr_rejects_bad_arguments(tmp_path):
    with pytest.raises(kkdirect.KKDError):
        kkdirect.write_exr(str(tmp_path / "x.exr"), np.zeros((2, 2, 3), np.float32), pixel_format="rgba")
    with pytest.raises(kkdirect.KKDError):
        kkdirect.write_exr(str(tmp_path / "nodir" / "x.exr"), np.zeros((2, 2, 3), np.float32))


def test_write_pfm_rejects_other_formats(tmp_path):
    with pytest.raises(kkdirect.KKDError):
        kkdirect.write_pfm(str(tmp_path / "x.pfm"), np.zeros((2, 2, 2), np.float32))          # luminanceAlpha: not a PFM format
    with pytest.raises(kkdirect.KKDError):
        kkdirect.write_pfm(str(tmp_path / "nodir" / "x.pfm"), np.zeros((2, 2, 3), np.float32))


@pytest.mark.parametrize("shape", [(6, 4, 3), (5, 8, 1), (2, 3, 4), (300, 200, 2)])
def test_write_npy_roundtrip(tmp_path, shape):
    img = np.random.RandomState(2).randn(*shape).astype(np.float32)
    path = tmp_path / "out.npy"
    kkdirect.write_npy(str(path), img)
    back = np.load(str(path), allow_pickle=False)
    assert back.dtype == np.float32 and not np.isfortran(back)
    assert back.shape == (shape[:2] if shape[2] == 1 else shape)                              # mfilm.cpp:343-344: N = 2 for one channel
    assert np.array_equal(back.reshape(shape), img)
    raw = path.read_bytes()
    assert raw[:8] == b"\x93NUMPY\x01\x00" and (10 + int.from_bytes(raw[8:10], "little")) % 16 == 0


def test_oracle_develop_known_answers():
    film = np.array([[[2, 4, 6, 1, 2], [1, 1, 1, 1, 0], [0.5, 0.25, 0.125, 0.5, 0.5]]], np.float32)
    rgba = kkoracle.develop_film(film, "rgba")
    assert np.array_equal(rgba, np.array([[[1, 2, 3, 0.5], [0, 0, 0, 0], [1, 0.5, 0.25, 1]]], np.float32))
    lum = kkoracle.develop_film(film, "luminance")
    np.testing.assert_allclose(lum[0, :, 0], [0.212671 + 2 * 0.715160 + 3 * 0.072169, 0, 0.212671 + 0.5 * 0.715160 + 0.25 * 0.072169], rtol=1e-6)
    xyz = kkoracle.develop_film(film, "xyz")
    assert np.array_equal(xyz[..., 1], lum[..., 0]) or np.allclose(xyz[..., 1], lum[..., 0], rtol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt", FORMATS)
def test_develop_bit_exact(gpu, fmt):
    film = _film(97, 131, seed=3)
    with gpu.Context(device=0) as ctx:
        out = ctx.develop(film, fmt)
    ref = kkoracle.develop_film(film, fmt)
    assert out.shape == ref.shape
    assert np.array_equal(out.view(np.uint32), ref.view(np.uint32))


@pytest.mark.gpu
def test_develop_device_on_rendered_film(gpu, tmp_path):
    import torch
    sc = scenes.cornell_box(64, 64)
    with gpu.Context(device=0).upload(sc) as ctx:
        film = torch.zeros(64 * 64 * 5, dtype=torch.float32, device="cuda")
        ctx.render_device(film.data_ptr(), spp=16)
        rgb = torch.zeros(64 * 64 * 3, dtype=torch.float32, device="cuda")
        ctx.develop_device(film.data_ptr(), 64, 64, rgb.data_ptr(), "rgb")
        host = film.cpu().numpy().reshape(64, 64, 5)
        out = rgb.cpu().numpy().reshape(64, 64, 3)
        assert np.array_equal(out.view(np.uint32), kkoracle.develop_film(host, "rgb").view(np.uint32))
        assert np.array_equal(out, kkdirect.develop(host))                                    # the Python mirror agrees
        with pytest.raises(kkdirect.KKDError):
            ctx.develop(host, 17)
    kkdirect.write_pfm(str(tmp_path / "cbox.pfm"), out)
    assert (tmp_path / "cbox.pfm").read_bytes() == kkoracle.pfm_bytes(out)


# ---- Film 'highQualityEdges' (librender/film.cpp:54; BlockedRenderProcess::bindResource, librender/renderproc.cpp:163-171) ----
def _hq_scene(hq, w=40, h=24):
    from kkdirect import scenes, look_at
    sc = scenes.cornell_box(w, h)
    sc.set_sensor(look_at((278, 273, -800), (278, 273, -799), (0, 1, 0)), 39.3077, w, h, near=10.0, far=2800.0, high_quality_edges=hq)
    return sc


def test_high_quality_edges_oracle_samples_the_border():
    import kkoracle
    w, h, spp, b = 40, 24, 8, 2                      # default Gaussian: border 2
    plain, sp = kkoracle.Oracle(_hq_scene(False)).render(spp)
    hq, sh = kkoracle.Oracle(_hq_scene(True)).render(spp)
    assert sp.samples == w * h * spp and sh.samples == (w + 2 * b) * (h + 2 * b) * spp
    # the samples taken outside the film only add filter weight near the edges: interior weights agree in expectation,
    # edge pixels gain weight, and the film keeps its size
    assert hq.shape == plain.shape
    interior = plain[6:-6, 6:-6, 4].mean()
    for edge_plain, edge_hq in ((plain[0, :, 4], hq[0, :, 4]), (plain[:, 0, 4], hq[:, 0, 4]), (plain[-1, :, 4], hq[-1, :, 4]), (plain[:, -1, 4], hq[:, -1, 4])):
        assert edge_plain.mean() < 0.9 * interior            # half of the footprints that would reach an edge pixel are missing
        assert abs(edge_hq.mean() / interior - 1) < 0.06     # with the border sampled, edge pixels weigh like interior ones
    assert abs(hq[6:-6, 6:-6, 4].mean() / plain[6:-6, 6:-6, 4].mean() - 1) < 0.05


@pytest.mark.gpu
def test_high_quality_edges_gpu_matches_oracle():
    import kkoracle, kkdirect, helpers
    for blk in (0, 16):
        sc = _hq_scene(True)
        ref, so = kkoracle.Oracle(sc).render(32, seed=1, block_size=blk)
        with kkdirect.Context(device=0, seed=1).upload(sc) as ctx:
            film, sg = ctx.render(32, block_size=blk)
        assert sg.samples == so.samples == 44 * 28 * 32
        np.testing.assert_allclose(film[..., 4], ref[..., 4], rtol=1e-4, atol=1e-4)
        assert helpers.rel_mse(helpers.develop(film), helpers.develop(ref)) < 1e-5


# ---- MFilm text output (reference src/films/mfilm.cpp:269-334) ----
# expected bytes written out by hand from the reference's loop: "os << float" at setprecision(4) prints 0.5, 1.234e-05, 1235, 1e+10

_MF_IMG = np.array([[[0.5, 2, 3], [1.2344e-5, 5, 6]],
                    [[1234.56, 8, 9], [1e10, 11, 12]]], np.float32)          # H = 2, W = 2, C = 3


def test_write_mfilm_matlab_known_bytes(tmp_path):
    p = tmp_path / "a.m"
    kkdirect.write_mfilm_text(str(p), _MF_IMG)
    assert p.read_bytes() == (b"data = [0.5, 1.234e-05;\n\t1235, 1e+10];\n"
                              b"\ndata(:, :, 2) = [2, 5;\n\t8, 11];\n"
                              b"\ndata(:, :, 3) = [3, 6;\n\t9, 12];\n")
    kkdirect.write_mfilm_text(str(p), _MF_IMG[..., 0], digits=2, variable="img")                # one channel, other parameters
    assert p.read_bytes() == b"img = [0.5, 1.2e-05;\n\t1.2e+03, 1e+10];\n"


def test_write_mfilm_mathematica_known_bytes(tmp_path):
    p = tmp_path / "b.m"
    kkdirect.write_mfilm_text(str(p), _MF_IMG, file_format="mathematica")
    assert p.read_bytes() == (b"data = Transpose[{{{0.5, 1.234*^-05},\n\t{1235, 1*^+10}},\n\n"
                              b"\t{{2, 5},\n\t{8, 11}},\n\n"
                              b"\t{{3, 6},\n\t{9, 12}}}, {3,1,2}];\n")
    kkdirect.write_mfilm_text(str(p), _MF_IMG[..., :1], file_format="Mathematica")              # fileFormat is lower-cased, mfilm.cpp:93-94
    assert p.read_bytes() == b"data = {{0.5, 1.234*^-05},\n\t{1235, 1*^+10}};\n"


def test_write_mfilm_text_parses_back_and_edge_values(tmp_path):
    img = np.random.RandomState(5).rand(7, 5, 4).astype(np.float32) * 100
    img[0, 0, 0] = np.inf; img[1, 2, 3] = 0.0; img[2, 1, 1] = -3.5e-30
    p = tmp_path / "c.m"
    kkdirect.write_mfilm_text(str(p), img, digits=9)                                            # 9 digits identify a float32
    text = p.read_text()
    chans = [blk[blk.index("[") + 1:blk.index("]")] for blk in text.split("data")[1:]]
    assert len(chans) == 4
    back = np.stack([np.array([[float(v) for v in row.split(",")] for row in ch.split(";")], np.float32) for ch in chans], -1)
    assert np.array_equal(back, img)
    kkdirect.write_mfilm_text(str(p), np.full((1, 1), 0.125, np.float32), digits=0)             # 0 digits at the C ABI = the plugin's default 4
    assert p.read_bytes() == b"data = [0.125];\n"
    with pytest.raises(kkdirect.KKDError):
        kkdirect.write_mfilm_text(str(p), img, file_format="numpy")
    with pytest.raises(kkdirect.KKDError):
        kkdirect.write_mfilm_text(str(tmp_path / "nodir" / "x.m"), img)
    with pytest.raises(kkdirect.KKDError):
        kkdirect.write_mfilm_text(str(p), img, digits=-1)
