"""Synthetic stand-in scenes.

The reference's scene assets (example-scenes-cg24/: cornell-box, veach-mis, bathroom2) are not
available, so the benchmark / parity workloads are deterministic procedural stand-ins that use the
reference's material names (Source/Model.cpp:16-51) and image sizes (read from Results/*.png):

  S1 cornell_box()   1024x1024, 5 wall quads + ceiling quad light + tessellated ball
  S2 veach_mis()     1280x720, 4 sphere lights, 4 Phong plates, Lambert-like floor/back wall
  S3 bathroom()      1280x720, >=120k triangles, textured Lambertian, mirrors, Empty, window light
  S4 triangle_soup() >=8M small random triangles (HBM-resident stress for the roofline claim)
  S0 random_rays()   seeded incoherent rays inside a bounding box

Everything is generated with numpy from fixed seeds; no files are read.
"""
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from . import _abi


@dataclass
class Material:
    name: str
    type: int
    kd: tuple = (0.0, 0.0, 0.0)
    ks: tuple = (0.0, 0.0, 0.0)
    ns: float = 0.0
    emission: tuple = (0.0, 0.0, 0.0)
    eta: tuple = (1.0, 1.0, 1.0)
    k: tuple = (0.0, 0.0, 0.0)
    alpha_x: float = 0.3
    alpha_y: float = 0.3
    texture: int = -1


@dataclass
class Camera:
    width: int
    height: int
    fovy: float
    eye: tuple
    look_at: tuple
    up: tuple = (0.0, 1.0, 0.0)


@dataclass
class SceneData:
    name: str
    vertices: np.ndarray  # (n,3,3)
    texcoords: Optional[np.ndarray]  # (n,3,2)
    normals: Optional[np.ndarray]  # (n,3,3)
    mesh_first_tri: np.ndarray  # (m+1,)
    mesh_material: np.ndarray  # (m,)
    mesh_names: List[str]
    materials: List[Material]
    camera: Camera
    textures: List[np.ndarray] = field(default_factory=list)
    light_meshes: Optional[List[int]] = None  # explicit `lights` list (mesh indices); None = every emissive mesh (main.cpp:40-45)

    @property
    def n_tris(self):
        return int(self.vertices.shape[0])

    def bounds(self):
        v = self.vertices.reshape(-1, 3)
        return v.min(axis=0), v.max(axis=0)


class _Builder:
    def __init__(self, name):
        self.name = name
        self.v, self.t, self.n = [], [], []
        self.first = [0]
        self.mesh_mat, self.mesh_names = [], []
        self.materials: List[Material] = []
        self.textures: List[np.ndarray] = []

    def material(self, m: Material) -> int:
        for i, x in enumerate(self.materials):
            if x.name == m.name:
                return i
        self.materials.append(m)
        return len(self.materials) - 1

    def mesh(self, name, mat_index, verts, uvs=None, normals=None):
        verts = np.asarray(verts, dtype=np.float64).reshape(-1, 3, 3)
        n = verts.shape[0]
        if uvs is None:
            uvs = np.zeros((n, 3, 2))
        if normals is None:
            normals = np.zeros((n, 3, 3))
        self.v.append(verts)
        self.t.append(np.asarray(uvs, dtype=np.float64).reshape(n, 3, 2))
        self.n.append(np.asarray(normals, dtype=np.float64).reshape(n, 3, 3))
        self.first.append(self.first[-1] + n)
        self.mesh_mat.append(mat_index)
        self.mesh_names.append(name)

    def build(self, camera) -> SceneData:
        return SceneData(
            name=self.name,
            vertices=np.concatenate(self.v, axis=0),
            texcoords=np.concatenate(self.t, axis=0),
            normals=np.concatenate(self.n, axis=0),
            mesh_first_tri=np.asarray(self.first, dtype=np.uint64),
            mesh_material=np.asarray(self.mesh_mat, dtype=np.int32),
            mesh_names=self.mesh_names,
            materials=self.materials,
            camera=camera,
            textures=self.textures,
        )


def quad(p0, p1, p2, p3):
    """Two triangles (p0,p1,p2), (p0,p2,p3) with unit-square UVs; normal = (p1-p0)x(p2-p0)."""
    p0, p1, p2, p3 = (np.asarray(p, dtype=np.float64) for p in (p0, p1, p2, p3))
    verts = np.array([[p0, p1, p2], [p0, p2, p3]])
    uvs = np.array([[[0, 0], [1, 0], [1, 1]], [[0, 0], [1, 1], [0, 1]]], dtype=np.float64)
    return verts, uvs


def grid_quad(p0, pu, pv, nu, nv, displace=None):
    """A (nu x nv)-cell tessellated parallelogram p0 + s*pu + t*pv -> 2*nu*nv triangles."""
    p0, pu, pv = (np.asarray(p, dtype=np.float64) for p in (p0, pu, pv))
    s = np.linspace(0.0, 1.0, nu + 1)
    t = np.linspace(0.0, 1.0, nv + 1)
    S, T = np.meshgrid(s, t, indexing="ij")
    P = p0 + S[..., None] * pu + T[..., None] * pv
    if displace is not None:
        nrm = np.cross(pu, pv)
        nrm = nrm / np.linalg.norm(nrm)
        P = P + displace(S, T)[..., None] * nrm
    UV = np.stack([S, T], axis=-1)
    a, b, c, d = P[:-1, :-1], P[1:, :-1], P[1:, 1:], P[:-1, 1:]
    ua, ub, uc, ud = UV[:-1, :-1], UV[1:, :-1], UV[1:, 1:], UV[:-1, 1:]
    verts = np.concatenate([np.stack([a, b, c], axis=-2).reshape(-1, 3, 3), np.stack([a, c, d], axis=-2).reshape(-1, 3, 3)])
    uvs = np.concatenate([np.stack([ua, ub, uc], axis=-2).reshape(-1, 3, 2), np.stack([ua, uc, ud], axis=-2).reshape(-1, 3, 2)])
    return verts, uvs


def icosphere(subdiv, radius=1.0, center=(0.0, 0.0, 0.0)):
    """Icosphere with 20*4^subdiv outward-wound triangles, spherical UVs and vertex normals."""
    t = (1.0 + 5.0 ** 0.5) / 2.0
    v = np.array(
        [[-1, t, 0], [1, t, 0], [-1, -t, 0], [1, -t, 0], [0, -1, t], [0, 1, t], [0, -1, -t], [0, 1, -t],
         [t, 0, -1], [t, 0, 1], [-t, 0, -1], [-t, 0, 1]], dtype=np.float64)
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    f = np.array(
        [[0, 11, 5], [0, 5, 1], [0, 1, 7], [0, 7, 10], [0, 10, 11], [1, 5, 9], [5, 11, 4], [11, 10, 2], [10, 7, 6],
         [7, 1, 8], [3, 9, 4], [3, 4, 2], [3, 2, 6], [3, 6, 8], [3, 8, 9], [4, 9, 5], [2, 4, 11], [6, 2, 10],
         [8, 6, 7], [9, 8, 1]], dtype=np.int64)
    tri = v[f]  # (20,3,3)
    for _ in range(subdiv):
        a, b, c = tri[:, 0], tri[:, 1], tri[:, 2]
        ab = a + b
        bc = b + c
        ca = c + a
        ab /= np.linalg.norm(ab, axis=1, keepdims=True)
        bc /= np.linalg.norm(bc, axis=1, keepdims=True)
        ca /= np.linalg.norm(ca, axis=1, keepdims=True)
        tri = np.concatenate([np.stack([a, ab, ca], 1), np.stack([b, bc, ab], 1), np.stack([c, ca, bc], 1),
                              np.stack([ab, bc, ca], 1)])
    nrm = tri.copy()
    u = np.arctan2(tri[..., 2], tri[..., 0]) / (2 * np.pi) + 0.5
    w = np.arccos(np.clip(tri[..., 1], -1.0, 1.0)) / np.pi
    uvs = np.stack([u, w], axis=-1)
    verts = tri * radius + np.asarray(center, dtype=np.float64)
    return verts, uvs, nrm


def box(lo, hi):
    """Axis-aligned box, 12 outward-wound triangles."""
    x0, y0, z0 = lo
    x1, y1, z1 = hi
    faces = [
        ((x0, y0, z1), (x1, y0, z1), (x1, y1, z1), (x0, y1, z1)),  # +z
        ((x1, y0, z0), (x0, y0, z0), (x0, y1, z0), (x1, y1, z0)),  # -z
        ((x1, y0, z1), (x1, y0, z0), (x1, y1, z0), (x1, y1, z1)),  # +x
        ((x0, y0, z0), (x0, y0, z1), (x0, y1, z1), (x0, y1, z0)),  # -x
        ((x0, y1, z1), (x1, y1, z1), (x1, y1, z0), (x0, y1, z0)),  # +y
        ((x0, y0, z0), (x1, y0, z0), (x1, y0, z1), (x0, y0, z1)),  # -y
    ]
    vs, us = [], []
    for fq in faces:
        v, u = quad(*fq)
        vs.append(v)
        us.append(u)
    return np.concatenate(vs), np.concatenate(us)


# --------------------------------------------------------------------------------------- S1
def cornell_box(ball_subdiv=5, width=1024, height=1024, symmetric_camera=False, ball_cooktorrance_alpha=None) -> SceneData:
    """S1: open-front box [-1,1]^3, ceiling quad light (17,12,4), tessellated DiffuseBall.
    `ball_cooktorrance_alpha`: the ball becomes a CookTorrance conductor with that roughness and the loader's
    default gold eta / k (Source/Model.cpp:306-315) — the configuration behind the reference's published
    Results/cornell-box_..._alpha0.1.png."""
    b = _Builder("cornell-box")
    white = b.material(Material("DiffuseWhite", _abi.MAT_LAMBERTIAN, kd=(0.725, 0.71, 0.68)))
    red = b.material(Material("LeftWall", _abi.MAT_LAMBERTIAN, kd=(0.63, 0.065, 0.05)))
    green = b.material(Material("RightWall", _abi.MAT_LAMBERTIAN, kd=(0.14, 0.45, 0.091)))
    light = b.material(Material("Light", _abi.MAT_DIFFUSE_LIGHT, emission=(17.0, 12.0, 4.0)))
    if ball_cooktorrance_alpha is None:
        ball = b.material(Material("DiffuseBall", _abi.MAT_LAMBERTIAN, kd=(0.5, 0.5, 0.8)))
    else:
        ball = b.material(Material("DiffuseBall", _abi.MAT_COOKTORRANCE, kd=(0.5, 0.5, 0.8), eta=(0.1, 0.5, 1.5), k=(4.0, 0.02, 0.3),
                                   alpha_x=ball_cooktorrance_alpha, alpha_y=ball_cooktorrance_alpha))
    b.mesh("floor", white, *quad((-1, -1, 1), (1, -1, 1), (1, -1, -1), (-1, -1, -1)))
    b.mesh("ceiling", white, *quad((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1)))
    b.mesh("backWall", white, *quad((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1)))
    b.mesh("leftWall", red, *quad((-1, -1, 1), (-1, -1, -1), (-1, 1, -1), (-1, 1, 1)))
    b.mesh("rightWall", green, *quad((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1)))
    # light faces down (-y): (p1-p0)x(p2-p0) = (+x) x (+z)... wound so the normal is -y
    b.mesh("light", light, *quad((-0.25, 0.998, -0.25), (0.25, 0.998, -0.25), (0.25, 0.998, 0.25), (-0.25, 0.998, 0.25)))
    v, uv, n = icosphere(ball_subdiv, radius=0.45, center=(0.2, -0.55, 0.1))
    b.mesh("ball", ball, v, uv, n)
    # The eye sits slightly off the box axis on purpose: with eye=(0,0,3.4) the pixel-centre rays of
    # both image diagonals pass EXACTLY through the wall/ceiling/floor edges and hit two triangles of
    # different materials at bit-identical t; which one wins such a tie depends on BVH test order
    # (in the reference as well: Interval::Contains is inclusive and AABB::Hit culls t.max <= t.min),
    # so it is not a property any other tree can reproduce.  tests/test_gpu_parity.py keeps a
    # symmetric-camera case that documents this.
    eye = (0.0, 0.0, 3.4) if symmetric_camera else (0.0127, 0.0311, 3.4)
    cam = Camera(width, height, 39.3077, eye=eye, look_at=(0.0, 0.0, 0.0))
    return b.build(cam)


# --------------------------------------------------------------------------------------- S2
def veach_mis(width=1280, height=720, light_subdiv=3, plate_cells=8) -> SceneData:
    """S2: four sphere lights of growing radius / equal power over four tilted Phong plates."""
    b = _Builder("veach-mis")
    radii = [0.03, 0.1, 0.3, 0.9]
    xs = [-3.75, -1.25, 1.25, 3.75]
    cols = [(1.0, 0.25, 0.25), (1.0, 1.0, 0.25), (0.25, 1.0, 0.25), (0.25, 0.25, 1.0)]
    for i, (r, x, c) in enumerate(zip(radii, xs, cols)):
        scale = 800.0 * (radii[0] / r) ** 2  # equal power: radiance ~ 1/r^2
        m = b.material(Material(f"light{i + 1}", _abi.MAT_DIFFUSE_LIGHT, emission=tuple(scale * np.array(c))))
        v, uv, n = icosphere(light_subdiv, radius=r, center=(x, 0.0, 0.0))
        b.mesh(f"light{i + 1}", m, v, uv, n)
    ns_values = [20.0, 80.0, 400.0, 2000.0]
    for i, ns in enumerate(ns_values):
        m = b.material(Material(f"material{i}", _abi.MAT_PHONG, kd=(0.07, 0.09, 0.13), ks=(0.35, 0.35, 0.35), ns=ns))
        # plates step away from the camera and tilt up so each reflects the lights toward the eye
        z0 = 4.0 - 1.6 * i
        y0 = -4.0 + 0.45 * i
        tilt = np.radians(12.0 + 9.0 * i)
        pu = np.array([10.0, 0.0, 0.0])
        pv = np.array([0.0, 1.3 * np.sin(tilt), -1.3 * np.cos(tilt)])
        v, uv = grid_quad((-5.0, y0, z0), pu, pv, plate_cells * 4, plate_cells)
        b.mesh(f"plate{i}", m, v, uv)
    wall = b.material(Material("material4", _abi.MAT_PHONG, kd=(0.4, 0.4, 0.4), ks=(0.0, 0.0, 0.0), ns=1.0))
    v1, u1 = quad((-12, -4.5, 8), (12, -4.5, 8), (12, -4.5, -6), (-12, -4.5, -6))
    v2, u2 = quad((-12, -4.5, -6), (12, -4.5, -6), (12, 10, -6), (-12, 10, -6))
    b.mesh("floorAndWall", wall, np.concatenate([v1, v2]), np.concatenate([u1, u2]))
    # fill light; the reference's name table only knows light1..4 / Light (Model.cpp:16-51)
    extra = b.material(Material("Light", _abi.MAT_DIFFUSE_LIGHT, emission=(30.0, 30.0, 30.0)))
    v, uv, n = icosphere(light_subdiv, radius=0.5, center=(10.0, 10.0, 4.0))
    b.mesh("fill", extra, v, uv, n)
    cam = Camera(width, height, 28.0, eye=(0.0, 2.0, 15.0), look_at=(0.0, -2.0, 2.5))
    return b.build(cam)


# --------------------------------------------------------------------------------------- S3
def _wood_texture(size=512, seed=7):
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:size, 0:size].astype(np.float64) / size
    rings = np.sin((x * 14.0 + 0.6 * np.sin(y * 9.0) + 0.05 * rng.standard_normal((size, size))) * np.pi)
    base = 0.55 + 0.25 * rings
    img = np.stack([base * 0.75, base * 0.5, base * 0.3], axis=-1)
    return np.clip(img * 255.0, 0, 255).astype(np.uint8)


def _tile_texture(size=512):
    y, x = np.mgrid[0:size, 0:size]
    grout = ((x % 64) < 3) | ((y % 64) < 3)
    img = np.where(grout[..., None], np.array([120, 120, 120]), np.array([225, 228, 232]))
    return img.astype(np.uint8)


def bathroom(width=1280, height=720, detail=1.0) -> SceneData:
    """S3: closed room with displaced/tessellated fixtures, >= 120k triangles at detail=1."""
    b = _Builder("bathroom2")
    b.textures.append(_wood_texture())
    b.textures.append(_tile_texture())
    wall = b.material(Material("Wall", _abi.MAT_LAMBERTIAN, kd=(0.8, 0.8, 0.78), texture=1))
    floor = b.material(Material("WoodFloor", _abi.MAT_LAMBERTIAN, kd=(0.5, 0.35, 0.2), texture=0))
    wood = b.material(Material("Wood", _abi.MAT_LAMBERTIAN, kd=(0.5, 0.35, 0.2), texture=0))
    ceramic = b.material(Material("Ceramic", _abi.MAT_LAMBERTIAN, kd=(0.9, 0.9, 0.88)))
    towel = b.material(Material("Towel", _abi.MAT_LAMBERTIAN, kd=(0.6, 0.2, 0.25)))
    mirror = b.material(Material("Mirror", _abi.MAT_MIRROR))
    binm = b.material(Material("Bin", _abi.MAT_MIRROR))
    empty = b.material(Material("quad1", _abi.MAT_EMPTY))
    light = b.material(Material("Light", _abi.MAT_DIFFUSE_LIGHT, emission=(12.0, 11.0, 10.0)))
    plastic = b.material(Material("Plastic", _abi.MAT_LAMBERTIAN, kd=(0.2, 0.3, 0.6)))
    k = lambda n: max(2, int(round(n * detail)))  # noqa: E731
    # room: x in [-3,3], y in [0,3], z in [-4,4]; camera near +z looking toward -z
    b.mesh("floor", floor, *grid_quad((-3, 0, 4), (6, 0, 0), (0, 0, -8), k(64), k(96)))
    b.mesh("ceiling", wall, *grid_quad((-3, 3, -4), (6, 0, 0), (0, 0, 8), k(24), k(32)))
    b.mesh("wallBack", wall, *grid_quad((-3, 0, -4), (6, 0, 0), (0, 3, 0), k(48), k(24)))
    b.mesh("wallLeft", wall, *grid_quad((-3, 0, 4), (0, 0, -8), (0, 3, 0), k(64), k(24)))
    b.mesh("wallRight", wall, *grid_quad((3, 0, -4), (0, 0, 8), (0, 3, 0), k(64), k(24)))
    b.mesh("wallFront", wall, *grid_quad((3, 0, 4), (-6, 0, 0), (0, 3, 0), k(24), k(12)))
    # window light on the left wall + ceiling lamp
    b.mesh("window", light, *quad((-2.99, 1.2, -1.0), (-2.99, 1.2, 1.0), (-2.99, 2.6, 1.0), (-2.99, 2.6, -1.0)))
    b.mesh("lamp", light, *quad((-0.5, 2.99, -0.5), (0.5, 2.99, -0.5), (0.5, 2.99, 0.5), (-0.5, 2.99, 0.5)))
    # mirror on the back wall, black absorber strip under it
    b.mesh("mirror", mirror, *quad((-1.5, 1.0, -3.98), (1.5, 1.0, -3.98), (1.5, 2.6, -3.98), (-1.5, 2.6, -3.98)))
    b.mesh("absorber", empty, *quad((-1.5, 0.9, -3.97), (1.5, 0.9, -3.97), (1.5, 1.0, -3.97), (-1.5, 1.0, -3.97)))
    # bathtub: displaced tessellated basin (ceramic)
    bump = lambda S, T: -0.45 * np.sin(np.pi * S) * np.sin(np.pi * T) + 0.01 * np.sin(40 * S) * np.sin(40 * T)  # noqa: E731
    b.mesh("tubTop", ceramic, *grid_quad((0.8, 0.7, -3.6), (2.0, 0, 0), (0, 0, 2.6), k(128), k(176), displace=bump))
    vb, ub = box((0.8, 0.0, -3.6), (2.8, 0.25, -1.0))
    b.mesh("tubBase", ceramic, vb, ub)
    # sink + cabinet
    vb, ub = box((-2.9, 0.0, -3.5), (-1.7, 0.85, -2.3))
    b.mesh("cabinet", wood, vb, ub)
    v, uv, n = icosphere(max(1, int(round(5 * min(1.0, detail) + 0.01))), radius=0.35, center=(-2.3, 1.05, -2.9))
    b.mesh("sinkBowl", ceramic, v, uv, n)
    # towel: wavy cloth
    wave = lambda S, T: 0.06 * np.sin(12 * np.pi * S) * (0.3 + T)  # noqa: E731
    b.mesh("towel", towel, *grid_quad((2.95, 0.9, 0.0), (0, 0, 1.2), (0, 1.2, 0), k(128), k(96), displace=wave))
    # bin (mirror sphere) + plastic bottles
    v, uv, n = icosphere(max(1, int(round(4 * min(1.0, detail) + 0.01))), radius=0.3, center=(-2.3, 0.3, 1.5))
    b.mesh("bin", binm, v, uv, n)
    for i in range(6):
        v, uv, n = icosphere(max(1, int(round(3 * min(1.0, detail) + 0.01))), radius=0.09, center=(-2.75 + 0.22 * i, 0.94, -3.2))
        b.mesh(f"bottle{i}", plastic, v, uv, n)
    cam = Camera(width, height, 55.0, eye=(0.6, 1.6, 3.6), look_at=(-0.2, 1.2, -2.0))
    return b.build(cam)


# --------------------------------------------------------------------------------------- S4
def triangle_soup(n_tris=8_000_000, seed=4, extent=0.01, with_light=True) -> SceneData:
    """S4: n small random triangles in the unit cube — BVH + triangles far larger than L2/MALL."""
    rng = np.random.default_rng(seed)
    c = rng.random((n_tris, 1, 3))
    v = c + (rng.random((n_tris, 3, 3)) - 0.5) * extent
    b = _Builder("triangle-soup")
    m = b.material(Material("DiffuseWhite", _abi.MAT_LAMBERTIAN, kd=(0.7, 0.7, 0.7)))
    uv = np.tile(np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0]]), (n_tris, 1, 1))
    b.mesh("soup", m, v, uv)
    if with_light:
        lm = b.material(Material("Light", _abi.MAT_DIFFUSE_LIGHT, emission=(10.0, 10.0, 10.0)))
        b.mesh("light", lm, *quad((0.25, 1.2, 0.25), (0.75, 1.2, 0.25), (0.75, 1.2, 0.75), (0.25, 1.2, 0.75)))
    cam = Camera(512, 512, 40.0, eye=(0.5, 0.5, 3.0), look_at=(0.5, 0.5, 0.5))
    return b.build(cam)


# --------------------------------------------------------------------------------------- S0
def random_rays(n, lo, hi, seed=12345, tmin=1e-4, tmax=np.inf):
    """Incoherent rays: origin ~ U(box), direction uniform on S^2 (z = 2u-1, phi = 2 pi v)."""
    rng = np.random.default_rng(seed)
    rays = np.zeros(n, dtype=_abi.RAY_DTYPE)
    lo = np.asarray(lo, dtype=np.float64)
    hi = np.asarray(hi, dtype=np.float64)
    rays["o"] = lo + rng.random((n, 3)) * (hi - lo)
    z = 2.0 * rng.random(n) - 1.0
    phi = 2.0 * np.pi * rng.random(n)
    r = np.sqrt(np.maximum(0.0, 1.0 - z * z))
    rays["d"] = np.stack([r * np.cos(phi), r * np.sin(phi), z], axis=-1)
    rays["tmin"] = tmin
    rays["tmax"] = tmax
    return rays


def camera_rays(cam, n, seed=12345, tmin=1e-4, tmax=np.inf):
    """Coherent rays (SURVEY.md §8d, S0 "coherent" variant): n primary rays of `cam` in pixel order — pixel k % (W*H),
    jittered inside the pixel (uniform, seeded) so that repeated passes over the image are not identical rays.
    Camera::Initialize / GetRay arithmetic (Camera.cpp:75-117), directions left unnormalised like the reference's."""
    W, H = cam.width, cam.height
    eye, look, up = (np.asarray(v, dtype=np.float64) for v in (cam.eye, cam.look_at, cam.up))
    el = eye - look
    focal = np.sqrt(el @ el)
    vh = 2.0 * np.tan(np.radians(cam.fovy) / 2.0) * focal
    vw = vh * (W / H)
    w = el / focal
    u = np.cross(up, w)
    u /= np.sqrt(u @ u)
    v = np.cross(w, u)
    du, dv = vw * u / W, vh * (-v) / H
    p00 = eye - focal * w - vw * u / 2.0 - vh * (-v) / 2.0 + 0.5 * (du + dv)
    k = np.arange(n) % (W * H)
    rng = np.random.default_rng(seed)
    fx = (k % W) + rng.random(n) - 0.5
    fy = (k // W) + rng.random(n) - 0.5
    rays = np.zeros(n, dtype=_abi.RAY_DTYPE)
    rays["o"] = eye
    rays["d"] = p00 + fx[:, None] * du + fy[:, None] * dv - eye
    rays["tmin"] = tmin
    rays["tmax"] = tmax
    return rays


def tiny_scene() -> SceneData:
    """A 64x64 cornell with a coarse ball: the smoke / unit-test workload (runs in ms on the CPU)."""
    return cornell_box(ball_subdiv=1, width=64, height=64)


def mixed_materials(width=48, height=48) -> SceneData:
    """Small closed box exercising every material kind + an image texture (parity-test scene)."""
    b = _Builder("mixed")
    b.textures.append(_wood_texture(64))
    white = b.material(Material("DiffuseWhite", _abi.MAT_LAMBERTIAN, kd=(0.7, 0.7, 0.7)))
    woodm = b.material(Material("Wood", _abi.MAT_LAMBERTIAN, kd=(0.5, 0.3, 0.2), texture=0))
    light = b.material(Material("Light", _abi.MAT_DIFFUSE_LIGHT, emission=(15.0, 15.0, 15.0)))
    phong_hi = b.material(Material("material0", _abi.MAT_PHONG, kd=(0.2, 0.3, 0.4), ks=(0.5, 0.5, 0.5), ns=60.0))
    phong_lo = b.material(Material("material4", _abi.MAT_PHONG, kd=(0.5, 0.5, 0.3), ks=(0.1, 0.1, 0.1), ns=1.0))
    phong_tex = b.material(Material("material1", _abi.MAT_PHONG, kd=(0.2, 0.3, 0.4), ks=(0.5, 0.5, 0.5), ns=25.0, texture=0))
    mirror = b.material(Material("Mirror", _abi.MAT_MIRROR))
    gold = b.material(Material("Gold", _abi.MAT_COOKTORRANCE, kd=(0.8, 0.6, 0.2), eta=(0.1, 0.5, 1.5), k=(4.0, 0.02, 0.3),
                               alpha_x=0.3, alpha_y=0.3))
    empty = b.material(Material("quad1", _abi.MAT_EMPTY))
    debug = b.material(Material("Debug", _abi.MAT_DEBUG, kd=(0.1, 0.4, 0.1)))
    b.mesh("floor", woodm, *quad((-1, -1, 1), (1, -1, 1), (1, -1, -1), (-1, -1, -1)))
    b.mesh("ceiling", white, *quad((-1, 1, -1), (1, 1, -1), (1, 1, 1), (-1, 1, 1)))
    b.mesh("back", phong_lo, *quad((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1)))
    b.mesh("left", mirror, *quad((-1, -1, 1), (-1, -1, -1), (-1, 1, -1), (-1, 1, 1)))
    b.mesh("right", phong_hi, *quad((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1)))
    b.mesh("front", white, *quad((1, -1, 1.0), (-1, -1, 1.0), (-1, 1, 1.0), (1, 1, 1.0)))
    b.mesh("light", light, *quad((-0.3, 0.995, -0.3), (0.3, 0.995, -0.3), (0.3, 0.995, 0.3), (-0.3, 0.995, 0.3)))
    v, uv, n = icosphere(2, radius=0.3, center=(-0.4, -0.7, -0.2))
    b.mesh("goldBall", gold, v, uv, n)
    v, uv, n = icosphere(2, radius=0.25, center=(0.45, -0.75, 0.1))
    b.mesh("texBall", phong_tex, v, uv, n)
    b.mesh("absorber", empty, *quad((-0.2, -0.999, 0.3), (0.2, -0.999, 0.3), (0.2, -0.999, 0.6), (-0.2, -0.999, 0.6)))
    b.mesh("debugPatch", debug, *quad((0.5, -0.3, -0.999), (0.8, -0.3, -0.999), (0.8, 0.0, -0.999), (0.5, 0.0, -0.999)))
    cam = Camera(width, height, 60.0, eye=(0.0173, 0.0091, 0.95), look_at=(0.0, -0.2, 0.0))
    return b.build(cam)


def dump_scene(scene: SceneData, path: str):
    """Flat binary dump read by examples/render_scene.cpp (the C++ host-API driver)."""
    import struct
    with open(path, "wb") as f:
        cam = scene.camera
        f.write(struct.pack("<Iii d", 0x50525431, cam.width, cam.height, cam.fovy))
        f.write(struct.pack("<9d", *cam.eye, *cam.look_at, *cam.up))
        f.write(struct.pack("<I", len(scene.textures)))
        for t in scene.textures:
            t = np.ascontiguousarray(t, dtype=np.uint8)
            f.write(struct.pack("<iii", t.shape[1], t.shape[0], t.shape[2] if t.ndim == 3 else 1))
            f.write(t.tobytes())
        f.write(struct.pack("<I", len(scene.materials)))
        for m in scene.materials:
            f.write(struct.pack("<ii", m.type, m.texture))
            f.write(struct.pack("<18d", *m.kd, *m.ks, m.ns, *m.emission, *m.eta, *m.k, m.alpha_x, m.alpha_y))
        f.write(struct.pack("<I", len(scene.mesh_material)))
        for i, mat in enumerate(scene.mesh_material):
            name = scene.mesh_names[i].encode()
            a, b = int(scene.mesh_first_tri[i]), int(scene.mesh_first_tri[i + 1])
            f.write(struct.pack("<I", len(name)) + name + struct.pack("<iQ", int(mat), b - a))
            rec = np.concatenate([scene.vertices[a:b].reshape(b - a, 9), scene.normals[a:b].reshape(b - a, 9),
                                  scene.texcoords[a:b].reshape(b - a, 6)], axis=1)
            f.write(np.ascontiguousarray(rec, dtype=np.float64).tobytes())


def apply_loader_uv_fixup(scene: SceneData) -> SceneData:
    """The reference's OBJ loader replaces the texcoords of a triangle whose three UVs are not pairwise
    distinct by (0,0),(1,0),(1,1) (Source/Model.cpp:170-175).  Returns a copy with that rule applied,
    i.e. the scene as the C++ Model class hands it to the renderer."""
    import copy
    out = copy.copy(scene)
    tc = scene.texcoords.copy()
    eq = lambda a, b: (tc[:, a] == tc[:, b]).all(axis=1)  # noqa: E731
    bad = eq(0, 1) | eq(1, 2) | eq(0, 2)
    tc[bad] = np.array([[0.0, 0.0], [1.0, 0.0], [1.0, 1.0]])
    out.texcoords = tc
    return out


def export_obj(scene: SceneData, resources_dir: str, texture_format="ppm"):
    """Write <resources_dir>/<name>/<name>.obj, .mtl, .xml (+ P6 .ppm textures) in the layout the
    reference reads (main.cpp:17-33).  Numbers are written with 17 significant
    digits so they round-trip exactly."""
    import os
    d = os.path.join(resources_dir, scene.name)
    os.makedirs(d, exist_ok=True)
    fmt = lambda xs: " ".join(repr(float(x)) for x in xs)  # noqa: E731
    with open(os.path.join(d, scene.name + ".mtl"), "w") as f:
        for m in scene.materials:
            f.write(f"newmtl {m.name}\nKd {fmt(m.kd)}\nKs {fmt(m.ks)}\nNs {repr(float(m.ns))}\n")
            if m.texture >= 0:
                f.write(f"map_Kd tex{m.texture}.{texture_format}\n")
            f.write("\n")
    for i, t in enumerate(scene.textures):
        t = np.ascontiguousarray(t, dtype=np.uint8)
        if texture_format == "png":
            from PIL import Image
            Image.fromarray(t).save(os.path.join(d, f"tex{i}.png"))  # zlib-compressed, adaptive filters
            continue
        with open(os.path.join(d, f"tex{i}.ppm"), "wb") as f:
            f.write(f"P6\n{t.shape[1]} {t.shape[0]}\n255\n".encode())
            f.write(t.tobytes())
    with open(os.path.join(d, scene.name + ".obj"), "w") as f:
        f.write(f"mtllib {scene.name}.mtl\n")
        n = 0
        for i, mat in enumerate(scene.mesh_material):
            a, b = int(scene.mesh_first_tri[i]), int(scene.mesh_first_tri[i + 1])
            f.write(f"g {scene.mesh_names[i]}\nusemtl {scene.materials[int(mat)].name}\n")
            lines = []
            for t in range(a, b):
                for k in range(3):
                    lines.append("v " + fmt(scene.vertices[t, k]))
                    lines.append("vt " + fmt(scene.texcoords[t, k]))
                    lines.append("vn " + fmt(scene.normals[t, k]))
                j = n * 3
                lines.append(f"f {j + 1}/{j + 1}/{j + 1} {j + 2}/{j + 2}/{j + 2} {j + 3}/{j + 3}/{j + 3}")
                n += 1
            f.write("\n".join(lines) + "\n")
    cam = scene.camera
    with open(os.path.join(d, scene.name + ".xml"), "w") as f:
        f.write('<?xml version="1.0" encoding="utf-8"?>\n')
        f.write(f'<camera type="perspective" width="{cam.width}" height="{cam.height}" fovy="{repr(float(cam.fovy))}">\n')
        f.write('  <eye x="%r" y="%r" z="%r"/>\n' % tuple(float(x) for x in cam.eye))
        f.write('  <lookat x="%r" y="%r" z="%r"/>\n' % tuple(float(x) for x in cam.look_at))
        f.write('  <up x="%r" y="%r" z="%r"/>\n</camera>\n' % tuple(float(x) for x in cam.up))
        for m in scene.materials:
            if m.type == _abi.MAT_DIFFUSE_LIGHT:
                f.write('<light mtlname="%s" radiance="%r,%r,%r"/>\n' % ((m.name,) + tuple(float(x) for x in m.emission)))
    return d
