"""The scenes of the converged-image goldens (tests/golden/make_golden.py generates the images with the oracle, tests/test_gpu_converged.py
renders them on the GPU with another seed).  Config-shaped, but small enough for the oracle to converge in a minute:
  materials   every BSDF and emitter type of the path in one frame (test_gpu_parity._material_scene)
  c4_shaped   displaced cube-sphere (GGX roughconductor) + dielectric sphere + ground under the procedural HDR environment map
              with its 1e4 suns (kkdirect.scenes.scanned_mesh, 27 k triangles, 512 x 256 map)
  c3_shaped   a room lit by 10 000 single-triangle area emitters -- a 10 001-entry scene emitter CDF, sampled uniformly like the
              reference does (librender/scene.cpp:376-380, 828-852) -- hanging at least two units away from every surface, so that
              the image converges at the sample counts a test can afford (C3 proper, with its lights among the arches, does not)"""
import numpy as np
from kkdirect import Scene, look_at, abi, compute_vertex_normals, scenes


def c3_shaped(width=64, height=36, n_lights=10_000, seed=3):
    rng = np.random.RandomState(seed)
    sc = Scene()
    white, red, green = sc.twosided(sc.diffuse((0.7, 0.7, 0.7))), sc.twosided(sc.diffuse((0.6, 0.1, 0.1))), sc.twosided(sc.diffuse((0.1, 0.5, 0.15)))
    X, Y, Z = 10.0, 7.0, 10.0
    quad = [[0, 1, 2], [0, 2, 3]]
    sc.add_mesh([[0, 0, 0], [0, 0, Z], [X, 0, Z], [X, 0, 0]], quad, white)                       # floor
    sc.add_mesh([[0, Y, 0], [X, Y, 0], [X, Y, Z], [0, Y, Z]], quad, white)                       # ceiling
    sc.add_mesh([[0, 0, Z], [0, Y, Z], [X, Y, Z], [X, 0, Z]], quad, white)                       # back
    sc.add_mesh([[0, 0, 0], [0, Y, 0], [0, Y, Z], [0, 0, Z]], quad, red)                         # left
    sc.add_mesh([[X, 0, 0], [X, 0, Z], [X, Y, Z], [X, Y, 0]], quad, green)                       # right
    for k, (m, c) in enumerate(((sc.roughplastic((0.2, 0.3, 0.6), alpha=0.15), (3.2, 1.0, 6.0)), (sc.twosided(sc.diffuse((0.6, 0.5, 0.3))), (6.6, 1.2, 4.6)))):
        p, i = scenes.cube_sphere(12, 1.0 + 0.2 * k, 0.05, seed=k, center=c)
        sc.add_mesh(p, i, m, normals=compute_vertex_normals(p, i))
    # the lights: down-facing triangles in a slab two units under the ceiling, at least 2.6 above the objects
    c = np.stack([rng.uniform(0.8, X - 0.8, n_lights), rng.uniform(4.8, 5.0, n_lights), rng.uniform(0.8, Z - 0.8, n_lights)], 1)
    s = 0.05 + 0.03 * rng.rand(n_lights)
    ang = rng.uniform(0, 2 * np.pi, n_lights)
    p0 = c + np.stack([s * np.cos(ang), 0 * s, s * np.sin(ang)], 1)
    p1 = c + np.stack([s * np.cos(ang + 4.18879), 0 * s, s * np.sin(ang + 4.18879)], 1)
    p2 = c + np.stack([s * np.cos(ang + 2.0944), 0 * s, s * np.sin(ang + 2.0944)], 1)
    pos = np.stack([p0, p1, p2], 1).astype(np.float32)
    rad = ((rng.rand(n_lights, 3) * 0.8 + 0.2) * 40.0).astype(np.float32)
    lm = sc.diffuse((0, 0, 0))
    tri = np.array([[0, 1, 2]], np.uint32)
    for k in range(n_lights):
        sc.add_mesh(pos[k], tri, lm, radiance=rad[k])
    sc.set_sensor(look_at((5.0, 2.6, -7.5), (5.0, 2.4, 5.0), (0, 1, 0)), 50.0, width, height, near=0.1, far=100.0)
    return sc


def c4_shaped(width=64, height=36):
    return scenes.scanned_mesh(width, height, n=48, env_size=(512, 256))


def materials(width=96, height=64):
    from test_gpu_parity import _material_scene
    return _material_scene(width, height)


GOLDENS = {   # name -> (scene factory, samples per pixel, seed of the oracle render)
    "materials_96x64": (materials, 16384, 4321),
    "c4_shaped_64x36": (c4_shaped, 16384, 4322),
    "c3_shaped_64x36": (c3_shaped, 16384, 4323),
}
