"""On-disk mesh formats (the step right before the path).  CPU tests: the host-only loaders behind the
C ABI against the oracle's Python restatement of TriMesh::loadCompressed / serialize and PLYLoader, on files generated
here (the reference ships no mesh files: data/tests/*.ply are absent).  GPU test: a scene assembled from files
renders the same image as the scene assembled from arrays."""
import struct
import numpy as np
import pytest
import kkoracle
import kkdirect
from kkdirect import Scene, scenes, look_at

KEYS = ("positions", "normals", "texcoords", "colors", "indices")


def _mesh(seed, n=9, normals=True, uv=True, colors=False, name="mesh"):
    rng = np.random.RandomState(seed)
    xs = np.linspace(-1, 1, n, dtype=np.float32)
    X, Z = np.meshgrid(xs, xs, indexing="ij")
    pos = np.stack([X, 0.2 * np.sin(3 * X + seed) * np.cos(2 * Z), Z], -1).reshape(-1, 3).astype(np.float32)
    idx = scenes._grid(n - 1, n - 1)
    m = {"name": name, "positions": pos, "indices": idx, "normals": None, "texcoords": None, "colors": None, "face_normals": False}
    if normals:
        nr = rng.normal(size=pos.shape).astype(np.float32); m["normals"] = nr / np.linalg.norm(nr, axis=1, keepdims=True).astype(np.float32)
    if uv:
        m["texcoords"] = rng.rand(len(pos), 2).astype(np.float32)
    if colors:
        m["colors"] = rng.rand(len(pos), 3).astype(np.float32)
    return m


def _same(a, b, exact=True):
    for k in KEYS:
        assert (a[k] is None) == (b[k] is None), k
        if a[k] is not None:
            if (exact and k != "colors") or k == "indices":      # colours pass through pow(): library-dependent last bit
                assert np.array_equal(a[k], b[k]), k
            else:
                np.testing.assert_allclose(a[k], b[k], rtol=2e-6, atol=1e-6, err_msg=k)


@pytest.mark.parametrize("version,double", [(4, False), (4, True), (3, False), (3, True)])
def test_serialized_load_matches_oracle(tmp_path, version, double):
    meshes = [_mesh(1, name="first"), _mesh(2, n=5, normals=False, uv=False, colors=True, name="second"), _mesh(3, n=7, uv=False, name="third")]
    meshes[2]["face_normals"] = True
    path = str(tmp_path / "m.serialized")
    kkoracle.save_serialized(path, meshes, version=version, double=double)
    assert kkdirect.serialized_mesh_count(path) == 3
    for i, src in enumerate(meshes):
        got = kkdirect.load_serialized(path, i)
        ref = kkoracle.load_serialized(path, i)
        _same(got, ref)
        _same(got, src, exact=True)              # float32 -> float64 -> float32 is lossless
        assert got["face_normals"] == src["face_normals"]
        assert got["name"] == (src["name"] if version == 4 else "")
        assert np.array_equal(got["aabb"][0], src["positions"].min(0)) and np.array_equal(got["aabb"][1], src["positions"].max(0))


def test_serialized_single_mesh_without_dictionary(tmp_path):
    path = str(tmp_path / "one.serialized")
    kkoracle.save_serialized(path, [_mesh(4)], dictionary=False)
    assert kkdirect.serialized_mesh_count(path) == 1
    _same(kkdirect.load_serialized(path, 0), kkoracle.load_serialized(path, 0))
    with pytest.raises(kkdirect.KKDError):
        kkdirect.load_serialized(path, 1)


def test_serialized_to_world_and_mirroring(tmp_path):
    path = str(tmp_path / "m.serialized")
    kkoracle.save_serialized(path, [_mesh(5)])
    rot = look_at((1, 2, 3), (0, 0.5, 0), (0, 1, 0)).astype(np.float32)
    rot[:3, :3] *= np.float32(1.7)
    mirror = np.diag([-1, 1, 1, 1]).astype(np.float32) @ rot
    for tw in (rot, mirror, np.eye(4, dtype=np.float32)):
        got = kkdirect.load_serialized(path, 0, tw)
        ref = kkoracle.load_serialized(path, 0, tw)
        _same(got, ref, exact=False)
    flipped = kkdirect.load_serialized(path, 0, mirror)["indices"]
    plain = kkdirect.load_serialized(path, 0, rot)["indices"]
    assert np.array_equal(flipped[:, [1, 0, 2]], plain)          # det < 0: idx[0] <-> idx[1], serialized.cpp:196-201


def test_serialized_writer_roundtrip_and_oracle_reads_it(tmp_path):
    meshes = [_mesh(6, colors=True, name="a"), _mesh(7, n=4, normals=False, name="b")]
    path = str(tmp_path / "w.serialized")
    kkdirect.save_serialized(path, meshes)
    raw = open(path, "rb").read()
    assert struct.unpack_from("<HH", raw, 0) == (0x041C, 4) and struct.unpack_from("<I", raw, len(raw) - 4)[0] == 2
    for i, src in enumerate(meshes):
        _same(kkoracle.load_serialized(path, i), src)
        _same(kkdirect.load_serialized(path, i), src)
        assert kkoracle.load_serialized(path, i)["name"] == src["name"]


def test_serialized_errors(tmp_path):
    bad = tmp_path / "bad.serialized"
    bad.write_bytes(b"\x04\x1c\x04\x00" + b"\0" * 64)            # the byte-swapped header of old files, trimesh.cpp:253-257
    with pytest.raises(kkdirect.KKDError, match="old version"):
        kkdirect.load_serialized(str(bad))
    bad.write_bytes(b"\x1c\x04\x09\x00" + b"\0" * 64)
    with pytest.raises(kkdirect.KKDError, match="incompatible"):
        kkdirect.load_serialized(str(bad))
    good = str(tmp_path / "g.serialized")
    kkoracle.save_serialized(good, [_mesh(8)])
    raw = open(good, "rb").read()
    (tmp_path / "trunc.serialized").write_bytes(raw[:len(raw) // 2])
    with pytest.raises(kkdirect.KKDError):
        kkdirect.load_serialized(str(tmp_path / "trunc.serialized"))
    with pytest.raises(kkdirect.KKDError):
        kkdirect.load_serialized(str(tmp_path / "missing.serialized"))
    with pytest.raises(kkdirect.KKDError):
        kkdirect.load_serialized(good, 5)


def _write_ply(path, mode, mesh, quads=False, color_type="uchar", count_type="uchar", index_type="int", extra=True):
    pos, nrm, uv = mesh["positions"], mesh["normals"], mesh["texcoords"]
    nv = len(pos)
    faces = [list(t) for t in mesh["indices"]]
    if quads:   # merge pairs of grid triangles (a, b, c) + (d, a, c) back into quads (a, b, c, d)
        faces = [[0, 1, 2, 3], [4, 5, 6]] + faces[2:]
    cols = (np.random.RandomState(9).rand(nv, 3) * 255).astype(np.uint8)
    props = ["property float x", "property float y", "property float z"]
    if extra:
        props.append("property double confidence")      # parsed and dropped: no float64 callback
    if nrm is not None:
        props += ["property float nx", "property float ny", "property float nz"]
    if uv is not None:
        props += ["property float s", "property float t"]
    props += ["property %s red" % color_type, "property %s green" % color_type, "property %s blue" % color_type]
    fmt = {"ascii": "ascii", "le": "binary_little_endian", "be": "binary_big_endian"}[mode]
    head = ["ply", "format %s 1.0" % fmt, "comment made by tests", "element vertex %d" % nv] + props + \
           ["element face %d" % len(faces), "property list %s %s vertex_indices" % (count_type, index_type), "property uchar flags",
            "element edge 1", "property int a", "property int b", "end_header"]
    bo = ">" if mode == "be" else "<"
    tp = {"uchar": "u1", "float": "f4", "int": "i4", "uint": "u4"}
    with open(path, "wb") as f:
        f.write(("\n".join(head) + "\n").encode())
        for v in range(nv):
            vals = [(pos[v, 0], "f4"), (pos[v, 1], "f4"), (pos[v, 2], "f4")]
            if extra:
                vals.append((0.5, "f8"))
            if nrm is not None:
                vals += [(nrm[v, k], "f4") for k in range(3)]
            if uv is not None:
                vals += [(uv[v, k], "f4") for k in range(2)]
            vals += [((cols[v, k] if color_type == "uchar" else cols[v, k] / 255.0), tp[color_type]) for k in range(3)]
            if mode == "ascii":
                f.write((" ".join(repr(float(x)) if t[0] == "f" else str(int(x)) for x, t in vals) + "\n").encode())
            else:
                for x, t in vals:
                    f.write(np.array([x], bo + t).tobytes())
        for fc in faces:
            if mode == "ascii":
                f.write(("%d %s 7\n" % (len(fc), " ".join(str(i) for i in fc))).encode())
            else:
                f.write(np.array([len(fc)], bo + tp[count_type]).tobytes() + np.array(fc, bo + tp[index_type]).tobytes() + b"\x07")
        f.write(b"3 4\n" if mode == "ascii" else np.array([3, 4], bo + "i4").tobytes())


@pytest.mark.parametrize("mode", ["ascii", "le", "be"])
@pytest.mark.parametrize("variant", [dict(), dict(quads=True, count_type="uint", index_type="uint"), dict(color_type="float", extra=False)])
def test_ply_load_matches_oracle(tmp_path, mode, variant):
    mesh = _mesh(11, n=6)
    path = str(tmp_path / "bunny.ply")
    _write_ply(path, mode, mesh, **variant)
    tw = look_at((0, 1, 4), (0, 0, 0), (0, 1, 0)).astype(np.float32)
    for to_world in (None, tw):
        for srgb in (True, False):
            got = kkdirect.load_ply(path, to_world, srgb)
            ref = kkoracle.load_ply(path, to_world, srgb)
            _same(got, ref, exact=(to_world is None and mode != "ascii"))
            assert got["name"] == "bunny" and got["colors"] is not None
    got = kkdirect.load_ply(path)
    if variant.get("quads"):
        assert np.array_equal(got["indices"][:3], [[0, 1, 2], [3, 0, 2], [4, 5, 6]])          # ply.cpp:296-306
    else:
        assert np.array_equal(got["indices"], mesh["indices"])
    if mode != "ascii":
        assert np.array_equal(got["positions"], mesh["positions"]) and np.array_equal(got["texcoords"], mesh["texcoords"])


def test_ply_errors(tmp_path):
    mesh = _mesh(12, n=4)
    p = str(tmp_path / "m.ply")
    _write_ply(p, "le", mesh)
    raw = open(p, "rb").read()
    (tmp_path / "trunc.ply").write_bytes(raw[:-40])
    with pytest.raises(kkdirect.KKDError):
        kkdirect.load_ply(str(tmp_path / "trunc.ply"))
    (tmp_path / "notply.ply").write_bytes(b"solid\n")
    with pytest.raises(kkdirect.KKDError):
        kkdirect.load_ply(str(tmp_path / "notply.ply"))
    # double-precision positions have no callback in the reference loader: nothing is stored, the load fails on purpose? no:
    # positions stay zero but faces load; pentagons are rejected
    (tmp_path / "penta.ply").write_bytes(b"ply\nformat ascii 1.0\nelement vertex 5\nproperty float x\nproperty float y\nproperty float z\n"
                                         b"element face 1\nproperty list uchar int vertex_indices\nend_header\n"
                                         b"0 0 0\n1 0 0\n1 1 0\n0 1 0\n0.5 2 0\n5 0 1 2 3 4\n")
    with pytest.raises(kkdirect.KKDError, match="triangle and quad"):
        kkdirect.load_ply(str(tmp_path / "penta.ply"))


def test_scene_from_files_equals_scene_from_arrays(tmp_path):
    a = _mesh(13, n=12, normals=False, uv=False)
    b = _mesh(14, n=6, normals=True, uv=True)
    kkdirect.save_serialized(str(tmp_path / "s.serialized"), [a, b])
    _write_ply(str(tmp_path / "p.ply"), "le", b)
    s1 = Scene(); m = s1.diffuse()
    s1.add_serialized(str(tmp_path / "s.serialized"), m, shape_index=0)
    s1.add_ply(str(tmp_path / "p.ply"), m)
    s2 = Scene(); m = s2.diffuse()
    s2.add_mesh(a["positions"], a["indices"], m, normals=kkdirect.compute_vertex_normals(a["positions"], a["indices"]))
    s2.add_mesh(b["positions"], b["indices"], m, normals=b["normals"], texcoords=b["texcoords"])
    for x, y in zip(s1.arrays(), s2.arrays()):
        # the ply loader re-normalises stored normals (ply.cpp:229): last-bit differences against the raw arrays
        assert (x is None) == (y is None) and (x is None or np.allclose(x, y, rtol=0, atol=2e-7))


@pytest.mark.gpu
def test_render_scene_loaded_from_files(gpu, tmp_path):
    cb = scenes.cornell_box(64, 64)
    pos, idx = cb.arrays()[:2]
    ground = _mesh(15, n=20, normals=False, uv=False)
    ground["positions"] = (ground["positions"] * np.float32(120) + np.float32([278, 60, 280])).astype(np.float32)
    kkdirect.save_serialized(str(tmp_path / "g.serialized"), [ground])

    def build(from_file):
        sc = scenes.cornell_box(64, 64)
        m = sc.diffuse((0.3, 0.5, 0.7))
        if from_file:
            sc.add_serialized(str(tmp_path / "g.serialized"), m)
        else:
            sc.add_mesh(ground["positions"], ground["indices"], m, normals=kkdirect.compute_vertex_normals(ground["positions"], ground["indices"]))
        return sc
    with gpu.Context(device=0).upload(build(True)) as c1, gpu.Context(device=0).upload(build(False)) as c2:
        f1, _ = c1.render(8)
        f2, _ = c2.render(8)
    # same samples, same scene; float atomics of neighbouring footprints commute only up to rounding
    assert f1[..., 4].sum() > 0 and np.allclose(f1, f2, rtol=1e-4, atol=1e-4)


def _write_obj(path, meshes):
    """positions / texcoords / normals per corner index, one "g" + "usemtl" per mesh"""
    nv = 0
    with open(path, "w") as f:
        for m in meshes:
            for q in m["positions"]:
                f.write("v %.9g %.9g %.9g\n" % tuple(q))
            if m["texcoords"] is not None:
                for q in m["texcoords"]:
                    f.write("vt %.9g %.9g\n" % (q[0], 1.0 - q[1]))
            if m["normals"] is not None:
                for q in m["normals"]:
                    f.write("vn %.9g %.9g %.9g\n" % tuple(q))
        nt = nn = 0
        for m in meshes:
            f.write("g %s\nusemtl %s\n" % (m["name"], m["name"] + "_mat"))
            for t in m["indices"]:
                def corner(i):
                    a = str(nv + i + 1)
                    tt = str(nt + i + 1) if m["texcoords"] is not None else ""
                    nn_ = str(nn + i + 1) if m["normals"] is not None else ""
                    return a if not (tt or nn_) else (a + "/" + tt if not nn_ else a + "/" + tt + "/" + nn_)
                f.write("f %s %s %s\n" % tuple(corner(int(i)) for i in t))
            nv += len(m["positions"])
            nt += len(m["texcoords"]) if m["texcoords"] is not None else 0
            nn += len(m["normals"]) if m["normals"] is not None else 0


def test_scene_from_obj_equals_scene_from_arrays(tmp_path):
    """two groups with their own materials: the shapes, their order, their materials and their arrays are what add_mesh gives
    for the same data (vertices come out in first-use order, so the comparison is per triangle corner)"""
    a = _mesh(21, n=7, normals=False, uv=False, name="floor")
    b = _mesh(22, n=5, normals=True, uv=True, name="wall")
    _write_obj(str(tmp_path / "two.obj"), [a, b])
    s1 = Scene(); ma, mb = s1.diffuse((0.2, 0.3, 0.4)), s1.diffuse((0.7, 0.6, 0.5))
    ids = s1.add_obj(str(tmp_path / "two.obj"), {"floor_mat": ma, "wall_mat": mb})
    assert len(ids) == 2
    assert [kkdirect.load_obj(str(tmp_path / "two.obj"), k)["material"] for k in range(2)] == ["floor_mat", "wall_mat"]
    lm = [kkdirect.load_obj(str(tmp_path / "two.obj"), k) for k in range(2)]
    for got, want in zip(lm, (a, b)):
        assert np.array_equal(got["positions"][got["indices"]], want["positions"][want["indices"]])
        if want["texcoords"] is not None:
            np.testing.assert_allclose(got["texcoords"][got["indices"]], want["texcoords"][want["indices"]], atol=1e-7)
            np.testing.assert_allclose(got["normals"][got["indices"]], want["normals"][want["indices"]], atol=2e-7)
    assert s1.add_obj(str(tmp_path / "two.obj"), ma, shape_index=0) and len(s1.add_obj(str(tmp_path / "two.obj"), ma, shape_index=0)) == 2   # obj.cpp:332: the last mesh comes with number count - 2
    with pytest.raises(ValueError, match="no material"):
        Scene().add_obj(str(tmp_path / "two.obj"), {"floor_mat": 0})


@pytest.mark.gpu
def test_render_scene_loaded_from_obj(gpu, tmp_path):
    ground = _mesh(15, n=20, normals=False, uv=False, name="ground")
    ground["positions"] = (ground["positions"] * np.float32(120) + np.float32([278, 60, 280])).astype(np.float32)
    _write_obj(str(tmp_path / "g.obj"), [ground])

    def build(from_file):
        sc = scenes.cornell_box(64, 64)
        m = sc.diffuse((0.3, 0.5, 0.7))
        if from_file:
            sc.add_obj(str(tmp_path / "g.obj"), m)
        else:
            sc.add_mesh(ground["positions"], ground["indices"], m, normals=kkdirect.compute_vertex_normals(ground["positions"], ground["indices"]))
        return sc
    with gpu.Context(device=0).upload(build(True)) as c1, gpu.Context(device=0).upload(build(False)) as c2:
        f1, _ = c1.render(8)
        f2, _ = c2.render(8)
    # same triangles in the same order (the OBJ reader numbers vertices by first use: generated normals agree to rounding)
    assert f1[..., 4].sum() > 0 and np.allclose(f1, f2, rtol=2e-3, atol=2e-3)


# ---- maxSmoothAngle: TriMesh::rebuildTopology (reference src/librender/trimesh.cpp:472-610) ---------------------------------
def _cube_mesh(shared=True):
    """a unit cube: 8 shared vertices / 12 triangles, or the same triangles with one private vertex per corner"""
    p = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], np.float32)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    idx = np.array([t for a, b, c, d in quads for t in ((a, b, c), (a, c, d))], np.uint32)
    if not shared:
        p = p[idx.reshape(-1)]; idx = np.arange(36, dtype=np.uint32).reshape(12, 3)
    return {"name": "cube", "positions": p, "indices": idx, "normals": None, "texcoords": None, "colors": None, "face_normals": False}


@pytest.mark.parametrize("shared", [True, False])
@pytest.mark.parametrize("angle,n_verts", [(30.0, 24), (100.0, 8)])
def test_max_smooth_angle_on_a_cube(tmp_path, shared, angle, n_verts):
    """The documented behaviour (reference src/shapes/obj.cpp:126-140): faces meeting at more than maxSmoothAngle get vertices of
    their own (cube at 30 degrees: 6 faces x 4 corners), below it they share them (100 degrees: the 8 corners) -- whatever the
    input's own sharing was."""
    path = str(tmp_path / "cube.serialized")
    kkdirect.save_serialized(path, [_cube_mesh(shared)])
    got = kkdirect.load_serialized(path, max_smooth_angle=angle)
    assert len(got["positions"]) == n_verts and got["normals"] is None and len(got["indices"]) == 12
    src = _cube_mesh(shared)
    np.testing.assert_array_equal(got["positions"][got["indices"]], src["positions"][src["indices"]])     # same triangles
    if n_verts == 24:       # every vertex belongs to triangles of ONE face: its generated normal is the face normal
        sc = Scene(); sc.add_serialized(path, sc.diffuse(), max_smooth_angle=angle)
        pos, idx, nrm, _ = sc.arrays()
        fn = np.cross(pos[idx[:, 1]] - pos[idx[:, 0]], pos[idx[:, 2]] - pos[idx[:, 0]]); fn /= np.linalg.norm(fn, axis=1, keepdims=True)
        for k in range(3):
            np.testing.assert_allclose(nrm[idx[:, k]], fn, atol=1e-6)


@pytest.mark.parametrize("seed,uv,colors,angle", [(21, False, False, 20.0), (22, True, False, 12.0), (23, True, True, 60.0), (24, False, True, 5.0), (25, True, True, 0.5)])
def test_max_smooth_angle_matches_the_restatement(tmp_path, seed, uv, colors, angle):
    """library vs the oracle's corner-by-corner restatement on a bumpy grid with duplicated seam vertices (same positions,
    other uv / colours), a degenerate triangle and a triangle that uses one vertex twice"""
    m = _mesh(seed, n=8, normals=True, uv=uv, colors=colors)
    m["positions"] = np.concatenate([m["positions"], m["positions"][:8]])                 # a seam: 8 positions a second time
    for k in ("normals", "texcoords", "colors"):
        if m[k] is not None:
            extra = m[k][:8] if k == "normals" else np.random.RandomState(seed).rand(8, m[k].shape[1]).astype(np.float32)
            m[k] = np.concatenate([m[k], extra])
    n0 = 64
    m["indices"] = np.concatenate([m["indices"], np.array([[n0, n0 + 1, 9], [n0 + 1, n0 + 2, 10], [3, 3, 12], [5, 5, 5]], np.uint32)])
    path = str(tmp_path / "m.ply")
    _write_ply(path, "le", m)
    got = kkdirect.load_ply(path, max_smooth_angle=angle)
    want = kkoracle.rebuild_topology(kkoracle.load_ply(path), angle)
    _same(got, want)


def test_max_smooth_angle_excludes_face_normals(tmp_path):
    path = str(tmp_path / "cube.serialized")
    kkdirect.save_serialized(path, [_cube_mesh()])
    sc = Scene()
    with pytest.raises(ValueError, match="maxSmoothAngle"):
        sc.add_serialized(path, sc.diffuse(), face_normals=True, max_smooth_angle=30.0)


# ---- Wavefront OBJ (reference src/shapes/obj.cpp) ------------------------------------------------------------------

def _obj_all(path, **kw):
    n, eof = kkdirect.obj_mesh_count(path, collapse=kw.get("collapse", False))
    return [kkdirect.load_obj(path, i, **kw) for i in range(n)], eof


def _same_obj(lib_meshes, ora_meshes, exact=True):
    assert len(lib_meshes) == len(ora_meshes)
    for a, b in zip(lib_meshes, ora_meshes):
        assert a["name"] == b["name"] and a["material"] == b["material"], (a["name"], b["name"], a["material"], b["material"])
        _same(a, b, exact)


def test_obj_known_answer_quad(tmp_path):
    """written out by hand from obj.cpp: a quad is a fan of two triangles, equal corners are merged, v is stored as 1 - v,
    the mesh takes the file's stem as its name, normals are normalised"""
    p = tmp_path / "quad.obj"
    p.write_text("# a quad\nv 0 0 0\nv 1 0 0\nv 1 1 0\nv 0 1 0\nvt 0 0\nvt 1 0\nvt 1 1\nvt 0 1\nvn 0 0 2\nf 1/1/1 2/2/1 3/3/1 4/4/1\n")
    assert kkdirect.obj_mesh_count(str(p)) == (1, True)
    m = kkdirect.load_obj(str(p))
    assert m["name"] == "quad" and m["material"] == ""
    assert np.array_equal(m["indices"], [[0, 1, 2], [0, 2, 3]])
    assert np.array_equal(m["positions"], np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32))
    assert np.array_equal(m["texcoords"], np.array([[0, 1], [1, 1], [1, 0], [0, 0]], np.float32))
    assert np.array_equal(m["normals"], np.tile(np.array([[0, 0, 1]], np.float32), (4, 1)))
    m = kkdirect.load_obj(str(p), flip_tex_coords=False)
    assert np.array_equal(m["texcoords"], np.array([[0, 0], [1, 0], [1, 1], [0, 1]], np.float32))


def _obj_text(seed, crlf=False):
    """a file that uses every form the reader knows: the four corner forms, negative indices, n-gons, both "g" naming
    conventions are exercised by `lead_g`, usemtl splits, repeated names, a continued line, comments, unknown statements"""
    rng = np.random.RandomState(seed)
    L = ["# generated", "mtllib ignored.mtl", "o ignored"]
    nv = 0

    def grid(n):
        nonlocal nv
        pts = rng.rand(n * n, 3).round(3)
        for q in pts:
            L.append("v %g %g %g" % tuple(q))
        base = nv; nv += n * n
        return base
    for k in range(6):
        L.append("vn %g %g %g" % tuple(rng.normal(size=3).round(3)))
    L.append("vn 0 0 0")
    for k in range(5):
        L.append("vt %g %g" % tuple(rng.rand(2).round(3)))
    b = grid(3)
    L.append("g first")
    L.append("usemtl red")
    L.append("f %d %d %d" % (b + 1, b + 2, b + 4))
    L.append("f %d/1 %d/2 %d/3 %d/4" % (b + 2, b + 3, b + 6, b + 5))
    L.append("f %d//1 %d//2 \\" % (b + 4, b + 5))
    L.append("  %d//7" % (b + 7))
    L.append("usemtl blue")
    L.append("f %d/1/2 %d/2/3 %d/5/6 %d/3/1 %d/4/4" % (b + 5, b + 6, b + 9, b + 8, b + 7))
    L.append("g second")
    b2 = grid(2)
    L.append("f -1 -2 -3")
    L.append("f -4/-1/-1 -3/-2/-2 -1/-3/-3")
    L.append("s off")
    L.append("g first")
    L.append("f %d %d %d" % (b2 + 1, b2 + 2, b2 + 3))
    L.append("f %d %d" % (b2 + 1, b2 + 4))            # two corners: the last one repeats
    L.append("g")
    L.append("\tf %d %d %d" % (b + 1, b + 5, b + 9))
    return ("\r\n" if crlf else "\n").join(L) + ("\r\n" if crlf else "\n")


@pytest.mark.parametrize("seed,crlf,kw", [(1, False, {}), (2, True, {}), (3, False, {"collapse": True}), (4, False, {"flip_tex_coords": False}),
                                          (5, True, {"to_world": "trs"}), (6, False, {"to_world": "mirror"})])
def test_obj_load_matches_oracle(tmp_path, seed, crlf, kw):
    p = tmp_path / "scan.model.obj"
    p.write_bytes(_obj_text(seed, crlf).encode())
    kw = dict(kw)
    if kw.get("to_world") == "trs":
        kw["to_world"] = (look_at((1, 2, 3), (0, 0.5, 0), (0, 1, 0)) @ np.diag([2.0, 0.5, 1.5, 1.0])).astype(np.float32)
    elif kw.get("to_world") == "mirror":
        kw["to_world"] = np.diag([-1.0, 1.0, 1.0, 1.0]).astype(np.float32)
    lib, eof = _obj_all(str(p), **kw)
    ora, ora_eof = kkoracle.load_obj(str(p), **kw)
    assert eof == ora_eof
    # under a transform the normals go through an inverse transpose that the two sides form in different precisions: last bit
    _same_obj(lib, ora, exact="to_world" not in kw)
    if kw.get("collapse"):
        assert len(lib) == 1 and lib[0]["name"] == "scan.model" and lib[0]["material"] == "blue"
    else:
        # names as the plugin forms them: "g first" before any face = the name-before-geometry convention, so a mesh is saved
        # under the name in force when its faces were read; the usemtl split falls back to the stem; the second "first" is
        # numbered; the bare "g" leaves an empty name, which is not taken yet, so the last mesh keeps it (traced by hand through
        # obj.cpp:258-334)
        assert [m["name"] for m in lib] == ["first", "scan.model", "second", "first_3", ""], [m["name"] for m in lib]
        assert [m["material"] for m in lib] == ["red", "blue", "blue", "blue", "blue"]
        assert lib[0]["normals"] is not None and lib[0]["texcoords"] is not None     # one corner form with each is enough for the mesh
        assert lib[3]["normals"] is None and lib[3]["texcoords"] is None
        assert lib[3]["indices"].shape == (2, 3) and lib[3]["indices"][1][1] == lib[3]["indices"][1][2]


def test_obj_name_after_geometry_convention(tmp_path):
    """faces first, then "g name": the mesh that was just read takes the NEW name (obj.cpp:262-269)"""
    p = tmp_path / "late.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nv 1 1 0\nf 1 2 3\ng alpha\nf 2 4 3\ng beta\n")
    lib, eof = _obj_all(str(p)); ora, ora_eof = kkoracle.load_obj(str(p))
    _same_obj(lib, ora)
    assert [m["name"] for m in lib] == ["alpha", "beta"] and eof is False and ora_eof is False
    # the positions list is global: the second mesh only holds the three vertices it uses
    assert lib[1]["positions"].shape == (3, 3)


def test_obj_shape_index_rule():
    """obj.cpp:277, 293, 332: the mesh the end of the file closes comes with number count - 2, never under its own"""
    f = kkdirect.obj_shape_index_meshes
    assert f(3, True, -1) == [0, 1, 2]
    assert f(3, True, 0) == [0] and f(3, True, 1) == [1, 2] and f(3, True, 2) == []
    assert f(1, True, 0) == [] and f(1, True, -1) == [0]
    assert f(2, False, 1) == [1] and f(2, False, 0) == [0]


def test_obj_errors(tmp_path):
    def load(text, **kw):
        p = tmp_path / "bad.obj"; p.write_text(text)
        return kkdirect.load_obj(str(p), **kw)
    tri = "v 0 0 0\nv 1 0 0\nv 0 1 0\n"
    with pytest.raises(kkdirect.KKDError, match="tried to access vertex 4"):
        load(tri + "f 1 2 4\n")
    with pytest.raises(kkdirect.KKDError, match="tried to access vertex 0"):
        load(tri + "f 1 2 0\n")
    with pytest.raises(kkdirect.KKDError, match="tried to access normal 1"):
        load(tri + "f 1//1 2//1 3//1\n")
    with pytest.raises(kkdirect.KKDError, match="tried to access uv 2"):
        load(tri + "vt 0 0\nf 1/2 2/1 3/1\n")
    with pytest.raises(kkdirect.KKDError, match="Invalid OBJ face format"):
        load(tri + "f 1/1/1/1 2 3\n")
    with pytest.raises(kkdirect.KKDError, match="Invalid OBJ face format"):
        load(tri + "f\n")
    with pytest.raises(kkdirect.KKDError, match="out of range"):
        load(tri + "f 1 2 3\n", mesh_index=1)
    with pytest.raises(kkdirect.KKDError, match="cannot open"):
        kkdirect.load_obj(str(tmp_path / "absent.obj"))
    # a vertex defined AFTER the geometry was closed is out of bounds for it (the lists are read as they stand then)
    with pytest.raises(kkdirect.KKDError, match="tried to access vertex 4"):
        load(tri + "f 1 2 4\ng later\nv 1 1 0\nf 1 2 3\n")
    for text in (tri + "f 1 2 4\n", tri + "f 1/1/1/1 2 3\n"):
        p = tmp_path / "bad2.obj"; p.write_text(text)
        with pytest.raises(ValueError):
            kkoracle.load_obj(str(p))


def test_obj_max_smooth_angle(tmp_path):
    """a cube written face by face with shared corners: 8 vertices as read, 24 after maxSmoothAngle = 30 (obj.cpp:336-343)"""
    c = [(x, y, z) for x in (0, 1) for y in (0, 1) for z in (0, 1)]
    quads = [(1, 2, 4, 3), (5, 7, 8, 6), (1, 5, 6, 2), (3, 4, 8, 7), (1, 3, 7, 5), (2, 6, 8, 4)]
    p = tmp_path / "cube.obj"
    p.write_text("".join("v %d %d %d\n" % q for q in c) + "".join("f %d %d %d %d\n" % q for q in quads))
    assert kkdirect.load_obj(str(p))["positions"].shape == (8, 3)
    m = kkdirect.load_obj(str(p), max_smooth_angle=30)
    assert m["positions"].shape == (24, 3) and m["indices"].shape == (12, 3) and m["normals"] is None
