"""Host-side flattened scene (what a Mitsuba adapter would produce from Scene/TriMesh/BSDF/Emitter,
SURVEY 8b) -> kkd_scene_desc.  Shape order = Scene::addShape order (reference
src/librender/scene.cpp:558-591), global primitive id = first_prim + primIndex
(src/librender/skdtree.cpp:57-58,69-70).
"""
import ctypes as C
import math
import numpy as np
from . import abi


def look_at(origin, target, up):
    """Transform::lookAt (reference src/libcore/transform.cpp:191-214): camera-to-world, +z = view dir."""
    o = np.asarray(origin, np.float64); t = np.asarray(target, np.float64); u = np.asarray(up, np.float64)
    d = t - o; d /= np.linalg.norm(d)
    left = np.cross(u, d); left /= np.linalg.norm(left)
    new_up = np.cross(d, left)
    m = np.eye(4)
    m[:3, 0], m[:3, 1], m[:3, 2], m[:3, 3] = left, new_up, d, o
    return m.astype(np.float32)


def compute_vertex_normals(pos, idx):
    """TriMesh::computeNormals, angle-weighted (reference src/librender/trimesh.cpp:612-685)."""
    pos = np.asarray(pos, np.float32); idx = np.asarray(idx, np.int64)
    n = np.zeros_like(pos, dtype=np.float64)
    p = pos.astype(np.float64)
    v = [p[idx[:, k]] for k in range(3)]
    fn = np.cross(v[1] - v[0], v[2] - v[0])
    ln = np.linalg.norm(fn, axis=1, keepdims=True)
    ok = ln[:, 0] > 0
    fn = np.where(ln > 0, fn / np.maximum(ln, 1e-300), 0)
    for k in range(3):
        d0 = v[(k + 1) % 3] - v[k]; d1 = v[(k + 2) % 3] - v[k]
        d0 /= np.maximum(np.linalg.norm(d0, axis=1, keepdims=True), 1e-300)
        d1 /= np.maximum(np.linalg.norm(d1, axis=1, keepdims=True), 1e-300)
        ang = np.arccos(np.clip(np.sum(d0 * d1, axis=1), -1, 1))
        np.add.at(n, idx[ok, k], fn[ok] * ang[ok, None])
    l = np.linalg.norm(n, axis=1, keepdims=True)
    n = np.where(l > 0, n / np.maximum(l, 1e-300), np.array([1.0, 0, 0]))
    return n.astype(np.float32)


class Scene:
    """Accumulates shapes / materials / emitters; desc() yields a kkd_scene_desc whose pointers stay
    valid as long as this object lives."""

    def __init__(self):
        self._pos, self._nrm, self._uv, self._idx = [], [], [], []
        self.shapes, self.materials, self.emitters = [], [], []
        self.envmap = None
        self.sensor = None
        self._nv = 0
        self._nt = 0
        self._keep = None

    # ---- materials (defaults = the plugins' defaults) ----
    def _mat(self, **kw):
        m = abi.Material()
        m.type = kw.get("type", abi.BSDF_DIFFUSE)
        m.distribution = kw.get("distribution", abi.DISTR_BECKMANN)
        m.sample_visible = int(kw.get("sample_visible", True))
        m.nonlinear = int(kw.get("nonlinear", False))
        n = kw.get("nested", (-1, -1))
        m.nested[0], m.nested[1] = n
        m.alpha_u = kw.get("alpha_u", 0.1); m.alpha_v = kw.get("alpha_v", 0.1)
        m.eta = kw.get("eta", 1.0)
        for name, default in (("reflectance", (0.5,) * 3), ("specular_reflectance", (1.0,) * 3),
                              ("specular_transmittance", (1.0,) * 3), ("cond_eta", (0.0,) * 3), ("cond_k", (1.0,) * 3)):
            v = kw.get(name, default)
            if np.isscalar(v):
                v = (v,) * 3
            arr = getattr(m, name)
            for i in range(3):
                arr[i] = float(v[i])
        self.materials.append(m)
        return len(self.materials) - 1

    def diffuse(self, reflectance=(0.5, 0.5, 0.5)):
        return self._mat(type=abi.BSDF_DIFFUSE, reflectance=reflectance)

    def roughconductor(self, eta, k, alpha=0.1, alpha_v=None, distribution=abi.DISTR_BECKMANN, sample_visible=True,
                       specular_reflectance=1.0, ext_eta=1.000277):
        eta = np.asarray(eta, np.float32) / np.float32(ext_eta); k = np.asarray(k, np.float32) / np.float32(ext_eta)
        return self._mat(type=abi.BSDF_ROUGHCONDUCTOR, cond_eta=eta, cond_k=k, alpha_u=alpha,
                         alpha_v=alpha if alpha_v is None else alpha_v, distribution=distribution,
                         sample_visible=sample_visible, specular_reflectance=specular_reflectance)

    def roughplastic(self, diffuse_reflectance=(0.5, 0.5, 0.5), alpha=0.1, int_ior=1.49, ext_ior=1.000277,
                     distribution=abi.DISTR_BECKMANN, nonlinear=False, specular_reflectance=1.0, sample_visible=True):
        return self._mat(type=abi.BSDF_ROUGHPLASTIC, reflectance=diffuse_reflectance, alpha_u=alpha, alpha_v=alpha,
                         eta=np.float32(int_ior) / np.float32(ext_ior), distribution=distribution, nonlinear=nonlinear,
                         specular_reflectance=specular_reflectance, sample_visible=sample_visible)

    def dielectric(self, int_ior=1.5046, ext_ior=1.000277, specular_reflectance=1.0, specular_transmittance=1.0):
        return self._mat(type=abi.BSDF_DIELECTRIC, eta=np.float32(int_ior) / np.float32(ext_ior),
                         specular_reflectance=specular_reflectance, specular_transmittance=specular_transmittance)

    def twosided(self, front, back=-1):
        return self._mat(type=abi.BSDF_TWOSIDED, nested=(front, back))

    # ---- shapes ----
    def add_mesh(self, positions, indices, material, normals=None, texcoords=None, radiance=None, sampling_weight=1.0):
        """One TriMesh shape; radiance != None attaches an area emitter (reference src/emitters/area.cpp)."""
        positions = np.ascontiguousarray(positions, np.float32).reshape(-1, 3)
        indices = np.ascontiguousarray(indices, np.uint32).reshape(-1, 3)
        nv, nt = len(positions), len(indices)
        assert nt > 0 and indices.max() < nv
        sh = abi.Shape()
        sh.first_prim = self._nt; sh.n_prims = nt; sh.material_id = material; sh.emitter_id = -1; sh.flags = 0
        self._pos.append(positions); self._idx.append(indices + np.uint32(self._nv))
        if normals is not None:
            self._nrm.append((self._nv, np.ascontiguousarray(normals, np.float32).reshape(nv, 3)))
            sh.flags |= abi.SHAPE_HAS_NORMALS
        if texcoords is not None:
            self._uv.append((self._nv, np.ascontiguousarray(texcoords, np.float32).reshape(nv, 2)))
            sh.flags |= abi.SHAPE_HAS_TEXCOORDS
        self._nv += nv; self._nt += nt
        self.shapes.append(sh)
        sid = len(self.shapes) - 1
        if radiance is not None:
            e = abi.Emitter()
            e.type = abi.EMITTER_AREA; e.shape_id = sid; e.sampling_weight = sampling_weight
            for i in range(3):
                e.radiance[i] = float(radiance[i])
            self.emitters.append(e)
            sh.emitter_id = len(self.emitters) - 1
        return sid

    def add_loaded_mesh(self, mesh, material, face_normals=None, flip_normals=False, **kw):
        """A mesh dict from kkdirect.load_serialized / load_ply, configured like TriMesh::configure (reference
        src/librender/trimesh.cpp:366-390): stored normals are used unless faceNormals is set; a mesh without normals gets
        smooth vertex normals (computeNormals); flipNormals negates them."""
        fn = mesh["face_normals"] if face_normals is None else face_normals
        nrm = None
        if not fn:
            nrm = mesh["normals"] if mesh["normals"] is not None else compute_vertex_normals(mesh["positions"], mesh["indices"])
            if flip_normals:
                nrm = -nrm
        return self.add_mesh(mesh["positions"], mesh["indices"], material, normals=nrm, texcoords=mesh["texcoords"], **kw)

    @staticmethod
    def _check_smooth(face_normals, max_smooth_angle):
        if max_smooth_angle is not None and face_normals:      # ply.cpp:125-127, serialized.cpp:205-207
            raise ValueError("The properties 'maxSmoothAngle' and 'faceNormals' can't be specified at the same time!")

    def add_serialized(self, path, material, shape_index=0, to_world=None, face_normals=None, flip_normals=False, max_smooth_angle=None, **kw):
        """<shape type="serialized"> (reference src/shapes/serialized.cpp:150-208)."""
        from . import load_serialized
        self._check_smooth(face_normals, max_smooth_angle)
        return self.add_loaded_mesh(load_serialized(path, shape_index, to_world, max_smooth_angle), material, face_normals, flip_normals, **kw)

    def add_ply(self, path, material, to_world=None, srgb=True, face_normals=None, flip_normals=False, max_smooth_angle=None, **kw):
        """<shape type="ply"> (reference src/shapes/ply.cpp:96-136)."""
        from . import load_ply
        self._check_smooth(face_normals, max_smooth_angle)
        return self.add_loaded_mesh(load_ply(path, to_world, srgb, max_smooth_angle), material, face_normals, flip_normals, **kw)

    def add_obj(self, path, material, shape_index=-1, to_world=None, collapse=False, flip_tex_coords=True, face_normals=None,
                flip_normals=False, max_smooth_angle=None, **kw):
        """<shape type="obj"> (reference src/shapes/obj.cpp:189-349): one shape per mesh the plugin creates for `shapeIndex` (its
        rule, quirk included: kkdirect.obj_shape_index_meshes).  material: one material id for every mesh (an unnamed <bsdf>
        child), or a dict usemtl-name -> id with "" as the fallback, which is how WavefrontOBJ::addChild hands named BSDFs to
        the meshes (obj.cpp:738-757).  Returns the shape ids."""
        from . import load_obj, obj_mesh_count, obj_shape_index_meshes
        self._check_smooth(face_normals, max_smooth_angle)
        count, eof = obj_mesh_count(path, collapse)
        ids = []
        for k in obj_shape_index_meshes(count, eof, shape_index):
            m = load_obj(path, k, to_world, collapse, flip_tex_coords, max_smooth_angle)
            if isinstance(material, dict):
                if m["material"] not in material and "" not in material:
                    raise ValueError("no material for the OBJ material name %r" % m["material"])
                mat = material.get(m["material"], material.get(""))
            else:
                mat = material
            ids.append(self.add_loaded_mesh(m, mat, face_normals, flip_normals, **kw))
        return ids

    def add_point_light(self, position, intensity, sampling_weight=1.0):
        e = abi.Emitter()
        e.type = abi.EMITTER_POINT; e.shape_id = -1; e.sampling_weight = sampling_weight
        for i in range(3):
            e.radiance[i] = float(intensity[i]); e.position[i] = float(position[i])
        self.emitters.append(e)
        return len(self.emitters) - 1

    def set_envmap(self, texels, scale=1.0, to_world=None, sampling_weight=1.0):
        """texels: (H, W, 3) float32 lat-long image, row 0 = +Y pole (reference src/emitters/envmap.cpp)."""
        texels = np.ascontiguousarray(texels, np.float32)
        assert texels.ndim == 3 and texels.shape[2] == 3
        self.envmap = dict(texels=texels, scale=float(scale),
                           to_world=np.eye(3, dtype=np.float32) if to_world is None else np.asarray(to_world, np.float32))
        e = abi.Emitter()
        e.type = abi.EMITTER_ENVMAP; e.shape_id = -1; e.sampling_weight = sampling_weight
        self.emitters.append(e)
        return len(self.emitters) - 1

    def set_sensor(self, to_world, xfov_deg, width, height, near=1e-2, far=1e4, crop=None, has_alpha=False,
                   rfilter=abi.RFILTER_GAUSSIAN, rfilter_radius=None, rfilter_param=None, high_quality_edges=False):
        s = abi.Sensor()
        tw = np.asarray(to_world, np.float32).reshape(16)
        for i in range(16):
            s.to_world[i] = float(tw[i])
        s.xfov_deg = xfov_deg; s.near_clip = near; s.far_clip = far
        s.film_width, s.film_height = width, height
        cx, cy, cw, ch = crop if crop else (0, 0, width, height)
        s.crop_offset_x, s.crop_offset_y, s.crop_width, s.crop_height = cx, cy, cw, ch
        s.film_has_alpha = int(has_alpha)
        s.film_high_quality_edges = int(high_quality_edges)
        s.rfilter_type = rfilter
        defaults = {abi.RFILTER_GAUSSIAN: (2.0, (0.5, 0.0)), abi.RFILTER_BOX: (0.5 + 1e-5, (0.0, 0.0)),
                    abi.RFILTER_TENT: (1.0, (0.0, 0.0)), abi.RFILTER_MITCHELL: (2.0, (1 / 3.0, 1 / 3.0)),
                    abi.RFILTER_CATMULLROM: (2.0, (0.0, 0.0)), abi.RFILTER_LANCZOS: (3.0, (3.0, 0.0))}
        r, p = defaults[rfilter]
        s.rfilter_radius = r if rfilter_radius is None else rfilter_radius
        p = p if rfilter_param is None else rfilter_param
        s.rfilter_param[0], s.rfilter_param[1] = p
        self.sensor = s

    # ---- flatten ----
    def arrays(self):
        pos = np.concatenate(self._pos) if self._pos else np.zeros((0, 3), np.float32)
        idx = np.concatenate(self._idx) if self._idx else np.zeros((0, 3), np.uint32)
        nrm = uv = None
        if self._nrm:
            nrm = np.zeros((self._nv, 3), np.float32)
            for off, a in self._nrm:
                nrm[off:off + len(a)] = a
        if self._uv:
            uv = np.zeros((self._nv, 2), np.float32)
            for off, a in self._uv:
                uv[off:off + len(a)] = a
        return np.ascontiguousarray(pos), np.ascontiguousarray(idx), nrm, uv

    def aabb(self):
        pos, idx, _, _ = self.arrays()
        used = pos[np.unique(idx)]
        return used.min(0), used.max(0)

    def desc(self):
        assert self.sensor is not None, "set_sensor() first"
        pos, idx, nrm, uv = self.arrays()
        d = abi.SceneDesc()
        d.abi_version = abi.KKD_ABI_VERSION
        d.n_verts, d.n_tris = len(pos), len(idx)
        d.positions = pos.ctypes.data_as(abi.pf32)
        d.indices = idx.ctypes.data_as(abi.pu32)
        d.normals = nrm.ctypes.data_as(abi.pf32) if nrm is not None else None
        d.texcoords = uv.ctypes.data_as(abi.pf32) if uv is not None else None
        shapes = (abi.Shape * max(1, len(self.shapes)))(*self.shapes)
        mats = (abi.Material * max(1, len(self.materials)))(*self.materials)
        ems = (abi.Emitter * max(1, len(self.emitters)))(*self.emitters)
        d.n_shapes, d.shapes = len(self.shapes), shapes
        d.n_materials, d.materials = len(self.materials), mats
        d.n_emitters, d.emitters = len(self.emitters), ems
        keep = [pos, idx, nrm, uv, shapes, mats, ems]
        if self.envmap is not None:
            tex = self.envmap["texels"]
            d.envmap.height, d.envmap.width = tex.shape[0], tex.shape[1]
            d.envmap.texels = tex.ctypes.data_as(abi.pf32)
            d.envmap.scale = self.envmap["scale"]
            tw = self.envmap["to_world"].reshape(9)
            for i in range(9):
                d.envmap.to_world[i] = float(tw[i])
            # EnvironmentMap::createShape (reference src/emitters/envmap.cpp:331-341): bounding sphere of the
            # scene AABB (center, radius = |max - center|), radius * 1.5
            lo, hi = self.aabb()
            c = ((lo.astype(np.float32) + hi.astype(np.float32)) * np.float32(0.5)).astype(np.float32)
            r = np.float32(np.linalg.norm((hi.astype(np.float32) - c).astype(np.float32))) * np.float32(1.5)
            for i in range(3):
                d.envmap.bsphere_center[i] = float(c[i])
            d.envmap.bsphere_radius = float(r)
            keep.append(tex)
        d.sensor = self.sensor
        self._keep = keep
        return d
