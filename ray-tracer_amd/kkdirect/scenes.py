"""Seeded synthetic scenes for the five BASELINE.json configs (no assets ship with the reference;
SURVEY 8d).  All generators are deterministic functions of their arguments.

  cornell_box()       C1  ~32 tris, one area light, diffuse (twosided)
  sponza_class()      C2  ~260k tris colonnade interior, 4 area + 2 point lights, diffuse / roughplastic
  many_lights()       C3  interior + 10k single-triangle area emitters (10k-entry emitter CDF)
  scanned_mesh()      C4  >=10M-tri displaced cube-sphere under a procedural HDR envmap, roughconductor + dielectric
"""
import math
import numpy as np
from . import abi
from .scene import Scene, look_at, compute_vertex_normals


def _quad(p):
    """4 corners -> (positions[4,3], indices[2,3])"""
    return np.asarray(p, np.float32), np.array([[0, 1, 2], [0, 2, 3]], np.uint32)


def _add_quads(sc, quads, material, **kw):
    pos = np.concatenate([np.asarray(q, np.float32) for q in quads])
    idx = np.concatenate([np.array([[0, 1, 2], [0, 2, 3]], np.uint32) + 4 * i for i in range(len(quads))])
    return sc.add_mesh(pos, idx, material, **kw)


# ------------------------------------------------------------------------------------------------ C1
def cornell_box(width=512, height=512, rfilter=abi.RFILTER_GAUSSIAN, twosided=True, has_alpha=False):
    sc = Scene()
    def mat(rgb):
        m = sc.diffuse(rgb)
        return sc.twosided(m) if twosided else m
    white, red, green = mat((0.725, 0.71, 0.68)), mat((0.63, 0.065, 0.05)), mat((0.14, 0.45, 0.091))
    light_bsdf = sc.diffuse((0.78, 0.78, 0.78))
    _add_quads(sc, [[(552.8, 0, 0), (0, 0, 0), (0, 0, 559.2), (549.6, 0, 559.2)]], white)                     # floor
    _add_quads(sc, [[(556, 548.8, 0), (556, 548.8, 559.2), (0, 548.8, 559.2), (0, 548.8, 0)]], white)         # ceiling
    _add_quads(sc, [[(549.6, 0, 559.2), (0, 0, 559.2), (0, 548.8, 559.2), (556, 548.8, 559.2)]], white)       # back
    _add_quads(sc, [[(0, 0, 559.2), (0, 0, 0), (0, 548.8, 0), (0, 548.8, 559.2)]], green)                     # right
    _add_quads(sc, [[(552.8, 0, 0), (549.6, 0, 559.2), (556, 548.8, 559.2), (556, 548.8, 0)]], red)           # left
    _add_quads(sc, [[(343, 548.7, 227), (343, 548.7, 332), (213, 548.7, 332), (213, 548.7, 227)]], light_bsdf,
               radiance=(17.0, 12.0, 4.0))
    short = [[(130, 165, 65), (82, 165, 225), (240, 165, 272), (290, 165, 114)],
             [(290, 0, 114), (290, 165, 114), (240, 165, 272), (240, 0, 272)],
             [(130, 0, 65), (130, 165, 65), (290, 165, 114), (290, 0, 114)],
             [(82, 0, 225), (82, 165, 225), (130, 165, 65), (130, 0, 65)],
             [(240, 0, 272), (240, 165, 272), (82, 165, 225), (82, 0, 225)]]
    tall = [[(423, 330, 247), (265, 330, 296), (314, 330, 456), (472, 330, 406)],
            [(423, 0, 247), (423, 330, 247), (472, 330, 406), (472, 0, 406)],
            [(472, 0, 406), (472, 330, 406), (314, 330, 456), (314, 0, 456)],
            [(314, 0, 456), (314, 330, 456), (265, 330, 296), (265, 0, 296)],
            [(265, 0, 296), (265, 330, 296), (423, 330, 247), (423, 0, 247)]]
    _add_quads(sc, short, white)
    _add_quads(sc, tall, white)
    sc.set_sensor(look_at((278, 273, -800), (278, 273, -799), (0, 1, 0)), 39.3077, width, height,
                  near=10.0, far=2800.0, rfilter=rfilter, has_alpha=has_alpha)
    return sc


# ------------------------------------------------------------------------------------------------ helpers
def _grid(nu, nv):
    """indices of an (nu+1) x (nv+1) vertex grid -> 2*nu*nv triangles"""
    i, j = np.meshgrid(np.arange(nu), np.arange(nv), indexing="ij")
    a = (i * (nv + 1) + j).ravel(); b = a + (nv + 1); c = b + 1; d = a + 1
    return np.stack([np.stack([a, b, c], 1), np.stack([a, c, d], 1)], 1).reshape(-1, 3).astype(np.uint32)


def _cylinder(center, radius, y0, y1, nseg, nrings, flute=0.0, nflutes=0):
    th = np.linspace(0, 2 * np.pi, nseg + 1); ys = np.linspace(y0, y1, nrings + 1)
    T, Y = np.meshgrid(th, ys, indexing="ij")
    R = radius * (1.0 + flute * np.cos(nflutes * T)) * (1.0 - 0.08 * (Y - y0) / max(y1 - y0, 1e-6))
    pos = np.stack([center[0] + R * np.cos(T), Y, center[1] + R * np.sin(T)], -1).reshape(-1, 3)
    return pos.astype(np.float32), _grid(nseg, nrings)[:, ::-1].copy()


def _sheet(origin, du, dv, nu, nv, amp, waves, normal):
    u = np.linspace(0, 1, nu + 1); v = np.linspace(0, 1, nv + 1)
    U, V = np.meshgrid(u, v, indexing="ij")
    disp = amp * np.sin(2 * np.pi * waves * U) * (0.3 + 0.7 * V)
    pos = (np.asarray(origin)[None, None] + U[..., None] * np.asarray(du) + V[..., None] * np.asarray(dv)
           + disp[..., None] * np.asarray(normal))
    return pos.reshape(-1, 3).astype(np.float32), _grid(nu, nv)


def _arch(x0, x1, z, y_spring, thickness, depth, nseg, nrad):
    """half-torus-like arch band between two columns: semicircular strip extruded in z"""
    r = 0.5 * (x1 - x0); cx = 0.5 * (x0 + x1)
    a = np.linspace(0, np.pi, nseg + 1); w = np.linspace(-0.5, 0.5, nrad + 1)
    A, W = np.meshgrid(a, w, indexing="ij")
    rr = r - thickness * 0.5 * (1 + np.cos(2 * np.pi * W))  # curved soffit
    pos = np.stack([cx - rr * np.cos(A), y_spring + rr * np.sin(A), z + depth * W], -1).reshape(-1, 3)
    return pos.astype(np.float32), _grid(nseg, nrad)


# ------------------------------------------------------------------------------------------------ C2
def sponza_class(width=1920, height=1080, target_tris=260_000, seed=1, rfilter=abi.RFILTER_GAUSSIAN):
    """Procedural colonnade interior: 30 x 12 x 14 hall, two rows of fluted columns with arches,
    wavy drapes, tessellated floor; 4 area lights + 2 point lights; diffuse / roughplastic."""
    rng = np.random.RandomState(seed)
    sc = Scene()
    L, H, Wd = 30.0, 12.0, 14.0
    stone = sc.twosided(sc.diffuse((0.62, 0.58, 0.52)))
    floor_m = sc.roughplastic((0.45, 0.40, 0.34), alpha=0.15)
    col_m = sc.roughplastic((0.70, 0.68, 0.62), alpha=0.1)
    ceil_m = sc.twosided(sc.diffuse((0.75, 0.73, 0.70)))
    drape_ms = [sc.twosided(sc.diffuse(c)) for c in ((0.55, 0.08, 0.07), (0.08, 0.32, 0.12), (0.09, 0.12, 0.45))]
    drape_p = sc.twosided(sc.roughplastic((0.5, 0.35, 0.1), alpha=0.2))
    light_m = sc.diffuse((0.0, 0.0, 0.0))
    # budget split: columns 45 %, drapes 30 %, arches 15 %, floor/walls 10 %
    n_cols = 2 * 8
    seg = 48; rings = max(8, int(0.45 * target_tris / (n_cols * 2 * seg)))
    n_dr = 6
    dn = max(8, int(math.sqrt(0.30 * target_tris / (n_dr * 2))))
    n_arch = 2 * 7
    an = max(4, int(math.sqrt(0.15 * target_tris / (n_arch * 2 * 4)))); aseg = 4 * an
    fn = max(4, int(math.sqrt(0.10 * target_tris / (5 * 2))))
    # shell
    p, i = _sheet((0, 0, 0), (L, 0, 0), (0, 0, Wd), fn, fn, 0.0, 0, (0, 1, 0)); sc.add_mesh(p, i[:, ::-1].copy(), floor_m)
    p, i = _sheet((0, H, 0), (L, 0, 0), (0, 0, Wd), fn, fn, 0.0, 0, (0, 1, 0)); sc.add_mesh(p, i, ceil_m)
    p, i = _sheet((0, 0, 0), (L, 0, 0), (0, H, 0), fn, fn, 0.0, 0, (0, 0, 1)); sc.add_mesh(p, i, stone)
    p, i = _sheet((0, 0, Wd), (L, 0, 0), (0, H, 0), fn, fn, 0.0, 0, (0, 0, 1)); sc.add_mesh(p, i, stone)
    p, i = _sheet((L, 0, 0), (0, 0, Wd), (0, H, 0), fn, fn, 0.0, 0, (1, 0, 0)); sc.add_mesh(p, i, stone)
    p, i = _sheet((0, 0, 0), (0, 0, Wd), (0, H, 0), 2, 2, 0.0, 0, (1, 0, 0)); sc.add_mesh(p, i, stone)
    # columns + arches
    xs = np.linspace(3.0, L - 3.0, 8)
    for z in (3.5, Wd - 3.5):
        for x in xs:
            p, i = _cylinder((x, z), 0.55, 0.0, 7.0, seg, rings, flute=0.06, nflutes=12)
            sc.add_mesh(p, i, col_m, normals=compute_vertex_normals(p, i))
        for k in range(7):
            p, i = _arch(xs[k], xs[k + 1], z, 7.0, 0.5, 1.1, aseg, an)
            sc.add_mesh(p, i, stone, normals=compute_vertex_normals(p, i))
    # drapes between columns on the upper level
    for k in range(n_dr):
        x0 = 4.0 + 4.0 * k; z = 3.5 if k % 2 == 0 else Wd - 3.5
        p, i = _sheet((x0, 10.5, z), (3.2, 0, 0), (0, -5.5, 0), dn, dn, 0.18 + 0.05 * rng.rand(), 5 + k % 3, (0, 0, 1))
        sc.add_mesh(p, i, drape_ms[k % 3] if k % 4 else drape_p, normals=compute_vertex_normals(p, i))
    # lights: 4 downward-facing quads under the ceiling, 2 point lights along the aisle
    for (x, z, rad) in ((6.0, 7.0, (18, 16, 13)), (14.0, 7.0, (16, 16, 16)), (22.0, 7.0, (13, 15, 18)), (27.0, 7.0, (17, 14, 10))):
        q = [(x + 0.9, H - 0.05, z - 0.9), (x + 0.9, H - 0.05, z + 0.9), (x - 0.9, H - 0.05, z + 0.9), (x - 0.9, H - 0.05, z - 0.9)]
        _add_quads(sc, [q], light_m, radiance=rad)
    sc.add_point_light((10.0, 3.0, 7.0), (30.0, 24.0, 16.0))
    sc.add_point_light((20.0, 2.0, 4.5), (12.0, 18.0, 28.0))
    sc.set_sensor(look_at((1.5, 4.2, 7.6), (16.0, 4.6, 6.4), (0, 1, 0)), 62.0, width, height, near=0.05, far=200.0, rfilter=rfilter)
    return sc


# ------------------------------------------------------------------------------------------------ C3
def many_lights(width=1920, height=1080, n_lights=10_000, interior_tris=60_000, seed=1, single_mesh=False):
    """Interior + n_lights small emissive triangles.  single_mesh=False: every triangle is its own
    shape + area emitter (scene emitter CDF with n_lights entries, reference scene.cpp:376-380);
    single_mesh=True: one emissive mesh (TriMesh area CDF, reference trimesh.cpp:392-407)."""
    sc = sponza_class(width, height, target_tris=interior_tris, seed=seed)
    # drop the big lights' dominance: keep them, add the small ones
    rng = np.random.RandomState(seed + 7)
    c = np.stack([rng.uniform(1, 29, n_lights), rng.uniform(7.5, 11.5, n_lights), rng.uniform(1, 13, n_lights)], 1)
    s = 0.04 + 0.05 * rng.rand(n_lights)
    ang = rng.uniform(0, 2 * np.pi, n_lights)
    # triangles facing down (normal -y): order chosen so cross(p1-p0, p2-p0).y < 0
    p0 = c + np.stack([s * np.cos(ang), 0 * s, s * np.sin(ang)], 1)
    p1 = c + np.stack([s * np.cos(ang + 4.18879), 0 * s, s * np.sin(ang + 4.18879)], 1)
    p2 = c + np.stack([s * np.cos(ang + 2.0944), 0 * s, s * np.sin(ang + 2.0944)], 1)
    pos = np.stack([p0, p1, p2], 1).astype(np.float32)
    hue = rng.rand(n_lights, 3) * 0.8 + 0.2
    rad = (hue * 600.0).astype(np.float32)
    lm = sc.diffuse((0, 0, 0))
    if single_mesh:
        idx = np.arange(3 * n_lights, dtype=np.uint32).reshape(-1, 3)
        sc.add_mesh(pos.reshape(-1, 3), idx, lm, radiance=(500.0, 450.0, 400.0))
    else:
        tri = np.array([[0, 1, 2]], np.uint32)
        for k in range(n_lights):
            sc.add_mesh(pos[k], tri, lm, radiance=rad[k])
    return sc


# ------------------------------------------------------------------------------------------------ C4
def _fbm(d, seed, octaves=6):
    """seeded fBm on unit directions: sum of random plane waves per octave"""
    rng = np.random.RandomState(seed)
    out = np.zeros(len(d), np.float64)
    amp, freq = 1.0, 2.0
    for _ in range(octaves):
        for _ in range(6):
            k = rng.normal(size=3); k *= freq / np.linalg.norm(k)
            out += amp * np.sin(d @ k + rng.uniform(0, 2 * np.pi)) / 6.0
        amp *= 0.5; freq *= 2.1
    return out


def cube_sphere(n, radius=1.0, displace=0.08, seed=1, center=(0, 0, 0)):
    """6 faces x n x n x 2 triangles, displaced radially by fBm ('scan-like')."""
    u = np.linspace(-1, 1, n + 1)
    U, V = np.meshgrid(np.tan(u * np.pi / 4), np.tan(u * np.pi / 4), indexing="ij")
    one = np.ones_like(U)
    faces = [(one, U, V), (-one, V, U), (V, one, U), (U, -one, V), (U, V, one), (V, U, -one)]
    P, I = [], []
    base = _grid(n, n)
    for f, (x, y, z) in enumerate(faces):
        d = np.stack([x, y, z], -1).reshape(-1, 3)
        d /= np.linalg.norm(d, axis=1, keepdims=True)
        r = radius * (1.0 + displace * _fbm(d, seed))
        P.append((d * r[:, None] + np.asarray(center)).astype(np.float32))
        I.append(base + np.uint32(f * (n + 1) * (n + 1)))
    return np.concatenate(P), np.concatenate(I)


def procedural_envmap(width=4096, height=2048, seed=1):
    """sky gradient + ground + 3 suns with 1e4 peak (lat-long, row 0 = +Y)"""
    rng = np.random.RandomState(seed)
    th = (np.arange(height) + 0.5) * np.pi / height; ph = (np.arange(width) + 0.5) * 2 * np.pi / width
    T, Pp = np.meshgrid(th, ph, indexing="ij")
    d = np.stack([np.sin(Pp) * np.sin(T), np.cos(T), -np.cos(Pp) * np.sin(T)], -1)
    up = np.clip(d[..., 1], -1, 1)
    sky = np.stack([0.25 + 0.35 * (1 - up), 0.4 + 0.35 * (1 - up), 0.9 + 0.0 * up], -1) * (0.6 + 0.4 * np.clip(up, 0, 1))[..., None]
    ground = np.array([0.18, 0.15, 0.12])[None, None] * (0.5 + 0.5 * np.clip(-up, 0, 1))[..., None]
    img = np.where(up[..., None] > 0, sky, ground)
    for k in range(3):
        sd = np.array([np.cos(1.1 + 2.3 * k) * 0.8, 0.35 + 0.25 * k, np.sin(1.1 + 2.3 * k) * 0.8]); sd /= np.linalg.norm(sd)
        c = np.clip(d @ sd, -1, 1)
        col = np.array([1.0, 0.9 - 0.1 * k, 0.7 - 0.1 * k]) * 1e4 / (1 + k)
        img = img + np.exp((c - 1.0) * (4000.0 / (1 + k)))[..., None] * col
    return img.astype(np.float32)


def scanned_mesh(width=3840, height=2160, n=936, env_size=(4096, 2048), seed=1):
    """n=936 -> 6*936*936*2 = 10.5M triangles (roughconductor gold), a dielectric displaced sphere,
    a diffuse ground plane, lit only by the HDR envmap."""
    sc = Scene()
    gold = sc.roughconductor(eta=(0.143, 0.375, 1.442), k=(3.983, 2.386, 1.603), alpha=0.2, distribution=abi.DISTR_GGX)
    glass = sc.dielectric(1.5046)
    ground = sc.twosided(sc.diffuse((0.4, 0.4, 0.4)))
    p, i = cube_sphere(n, 1.0, 0.08, seed)
    sc.add_mesh(p, i, gold, normals=compute_vertex_normals(p, i))
    n2 = max(8, n // 8)
    p, i = cube_sphere(n2, 0.45, 0.03, seed + 1, center=(1.7, -0.5, 0.6))
    sc.add_mesh(p, i, glass, normals=compute_vertex_normals(p, i))
    _add_quads(sc, [[(-8, -1.15, -8), (-8, -1.15, 8), (8, -1.15, 8), (8, -1.15, -8)]], ground)
    sc.set_envmap(procedural_envmap(env_size[0], env_size[1], seed))
    sc.set_sensor(look_at((0.6, 0.9, -3.6), (0.3, -0.1, 0.0), (0, 1, 0)), 45.0, width, height, near=0.01, far=1000.0)
    return sc
