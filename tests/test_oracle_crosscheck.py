"""A SECOND, independent restatement of the reference's central routine, checked against the oracle sample by sample.

`oracle/pt_oracle.cpp` cannot be pinned against the reference (no fixtures, unbuildable here).  What can be done is to
write `Camera::RayColor` and what it calls a second time, from the reference's text and in another language, and require
both restatements to agree on every sample: a transcription slip in either one shows up as a difference.  This file is
that second restatement — plain Python / numpy, brute-force closest hit over all triangles (no BVH), recursion exactly as
in the reference:

    Camera::GetRay / RayColor            Source/Camera.cpp:108-204
    Triangle ctor, Hit, IsInterior, Sample   Source/Triangle.cpp:11-113
    HitRecord::SetFaceNormal             Source/Hittable.cpp:8-13
    HittableList::Sample, BVHNode::Sample    Source/HittableList.h:44-59, Source/BVH.cpp:62-67 (one light triangle: the
                                         area-CDF descent has nothing to choose; its two draws are still consumed)
    Material::WorldToLocal / LocalToWorld    Source/Material.h:76-98
    Lambertian, PhoneReflectance, PerfectMirror, DiffuseLight, EmptyMaterial   Source/Material.h:101-366,537-540
    ImageTexture::Value / GetPixel / SRGBToLinear   Source/Texture.cpp:22-71
    SampleCosineHemisphere / SampleUniformDiskConcentric   Source/RandomNumberGenerator.h:39-64

It takes from the oracle ONLY the random numbers (`oracle.rng_stream`: the keyed stream both sides must share, departure
B2) and compares with `orc.render_samples` per (pixel, sample) to 1e-12.  Documented departures are applied here as
there: B9 (an escaping shadow ray is unoccluded), B13 (Phong's unassigned attenuation is 0), B20 (`dvec2(xi, xi)`: y is
drawn first).  Phong is restated with the library calls the reference makes (acos, sin, cos, pow), so agreement to 1e-9 here
also covers the algebraic equivalents the GPU uses through the oracle.  CookTorrance is pinned separately by closed
forms in tests/test_materials.py.
"""
import math

import numpy as np
import pytest

import oracle
from pooraytracer_amd import _abi, scenes

INF = float("inf")


class Tri:
    """Triangle::Triangle (Triangle.cpp:11-53)."""

    def __init__(self, v, uv, mat):
        self.v = [np.array(x, dtype=np.float64) for x in v]
        self.uv = [np.array(x, dtype=np.float64) for x in uv]
        self.e0, self.e1 = self.v[1] - self.v[0], self.v[2] - self.v[0]
        n = np.cross(self.e0, self.e1)
        self.normal = n * (1.0 / math.sqrt(n @ n))
        d0, d1 = np.array(uv[1]) - np.array(uv[0]), np.array(uv[2]) - np.array(uv[0])
        with np.errstate(divide="ignore", invalid="ignore"):
            f = 1.0 / (d0[0] * d1[1] - d1[0] * d0[1])
            t = f * (d1[1] * self.e0 - d0[1] * self.e1)
            t = t * (1.0 / np.sqrt(t @ t))
        if np.isnan(t).any():
            helper = np.array([1.0, 0, 0]) if abs(self.normal[0]) < 0.9 else np.array([0, 1.0, 0])
            t = np.cross(self.normal, helper)
            t = t * (1.0 / math.sqrt(t @ t))
        self.tangent = t
        self.area = math.sqrt(n @ n) * 0.5
        self.D = self.normal @ self.v[0]
        self.w = n / (n @ n)
        self.mat = mat

    def hit(self, o, d, tmin, tmax):
        """Triangle::Hit + IsInterior (Triangle.cpp:54-83,100-113); returns t or None."""
        denom = self.normal @ d
        if abs(denom) < 1e-8:
            return None
        t = (self.D - self.normal @ o) / denom
        if not (tmin <= t <= tmax):
            return None
        p = o + d * t
        v0p = p - self.v[0]
        alpha = self.w @ np.cross(v0p, self.e1)
        beta = self.w @ np.cross(self.e0, v0p)
        if alpha != alpha or beta != beta or alpha < 0 or beta < 0 or alpha + beta > 1:
            return None
        self.last_uv = (1.0 - alpha - beta) * self.uv[0] + alpha * self.uv[1] + beta * self.uv[2]   # IsInterior, :111
        return t


class Tracer:
    def __init__(self, data, rr, background, sample_lights=True):
        self.tris = []
        for m in range(len(data.mesh_material)):
            mat = data.materials[int(data.mesh_material[m])]
            for t in range(int(data.mesh_first_tri[m]), int(data.mesh_first_tri[m + 1])):
                self.tris.append(Tri(data.vertices[t], data.texcoords[t], mat))
        self.lights = [t for t in self.tris if t.mat.type == _abi.MAT_DIFFUSE_LIGHT]
        assert len(self.lights) == 1, "the cross-check scene has exactly one light triangle"
        self.rr, self.background, self.sample_lights = rr, np.array(background, dtype=np.float64), sample_lights
        self.textures = data.textures
        self.rng = None
        self.seen = set()   # materials a path vertex landed on (the comparison must not be vacuous)

    def tex(self, ti, u, v):
        """ImageTexture::Value / GetPixel / SRGBToLinear (Texture.cpp:22-71)."""
        img = self.textures[ti]
        H, W = img.shape[0], img.shape[1]
        ch = img.shape[2] if img.ndim == 3 else 1
        u, v = min(max(u, 0.0), 1.0), min(max(v, 0.0), 1.0)
        x, y = u * (W - 1.0), (1.0 - v) * (H - 1.0)
        x0, y0 = int(x), int(y)
        x1, y1 = min(x0 + 1, W - 1), min(y0 + 1, H - 1)
        tx, ty = x - x0, y - y0

        def px(xx, yy):
            if ch >= 3:
                c = img[yy, xx, :3].astype(np.float64) * (1.0 / 255.0)
                return np.array([cc * (1.0 / 12.92) if cc <= 0.04045 else math.pow((cc + 0.055) * (1.0 / 1.055), 2.4) for cc in c])
            g = float(img[yy, xx] if img.ndim == 2 else img[yy, xx, 0]) * (1.0 / 255.0)
            return np.array([g, g, g])
        c0 = px(x0, y0) * (1 - tx) + px(x1, y0) * tx
        c1 = px(x0, y1) * (1 - tx) + px(x1, y1) * tx
        return c0 * (1 - ty) + c1 * ty

    def kd(self, mat, rec):
        return self.tex(mat.texture, rec["uv"][0], rec["uv"][1]) if mat.texture >= 0 else np.array(mat.kd, dtype=np.float64)

    def ks(self, mat, rec):                                  # Phong(mapKd, ...) stores the map in Ks too (Material.h:178-181)
        return self.tex(mat.texture, rec["uv"][0], rec["uv"][1]) if mat.texture >= 0 else np.array(mat.ks, dtype=np.float64)

    @staticmethod
    def phong_split(mat):                                    # SetProbabilitiesByNs, Material.h:318-327
        return (1.0, 0.0) if mat.ns <= 9.0 else (0.6, 0.4)

    @staticmethod
    def mirror_dir(wo):                                      # normalize(Reflect(wo, (0,0,1)))
        nz = np.array([0.0, 0.0, 1.0])
        r = -wo + 2.0 * (wo @ nz) * nz
        return r * (1.0 / math.sqrt(r @ r))

    def phong_eval(self, mat, wi, wo, rec):                  # Material.h:227-248 — draws one number
        pkd, pks = self.phong_split(mat)
        u = self.xi()
        if u < pkd:
            return np.zeros(3) if wi[2] <= 0 else self.kd(mat, rec) / math.pi
        if pkd <= u < pkd + pks:
            if wi[2] <= 0:
                return np.zeros(3)
            ca = max(0.0, wi @ self.mirror_dir(wo))
            if ca <= 0.0:
                return np.zeros(3)
            return self.ks(mat, rec) * (mat.ns + 2.0) / (2 * math.pi) * math.pow(ca, mat.ns)
        return np.zeros(3)

    def phong_scatter(self, mat, d, rec):                    # Material.h:183-226,263-285
        wo = self.to_local(-d, rec)
        pkd, pks = self.phong_split(mat)
        wi, f, pdf = np.zeros(3), np.zeros(3), 0.0
        u = self.xi()
        if u < pkd:
            wi = self.cosine_hemisphere()
            while wi[2] <= 0.0:
                wi = self.cosine_hemisphere()
            pdf = wi[2] / math.pi
            f = self.kd(mat, rec) / math.pi
        elif pkd <= u < pkd + pks:
            u1, u2 = self.xi(), self.xi()
            alpha = math.acos(math.pow(u1, 1.0 / (mat.ns + 1.0)))
            phi = 2.0 * math.pi * u2
            rw = np.array([math.sin(alpha) * math.cos(phi), math.sin(alpha) * math.sin(phi), math.cos(alpha)])
            lr = self.mirror_dir(wo)
            V = np.array([0.0, 1.0, 0.0]) if abs(lr[0]) > 0.9 else np.array([1.0, 0.0, 0.0])
            T = np.cross(V, lr)
            T = T * (1.0 / math.sqrt(T @ T))
            B = np.cross(lr, T)
            wi = rw[0] * T + rw[1] * B + rw[2] * lr
            pdf = 0.0 if wi[2] <= 0 else (mat.ns + 1.0) / (2 * math.pi) * math.pow(wi @ lr, mat.ns)
            lca = max(0.0, wi @ lr)
            if wi[2] > 0 and lca > 0:
                f = self.ks(mat, rec) * (mat.ns + 2.0) / (2 * math.pi) * math.pow(lca, mat.ns)
        att = f * wi[2] / pdf if (pdf > 0 and wi[2] > 0) else np.zeros(3)   # unassigned upstream -> 0 (B13)
        return True, att, self.to_world(wi, rec)

    def xi(self):
        return next(self.rng)

    # world.Hit: closest accepted triangle (HittableList.h:26-39 semantics; no exact ties in this scene)
    def world_hit(self, o, d, tmin, tmax):
        best, bt, buv = None, tmax, None
        for tr in self.tris:
            t = tr.hit(o, d, tmin, bt)
            if t is not None:
                best, bt, buv = tr, t, tr.last_uv
        if best is None:
            return None
        front = (d @ best.normal) < 0.0                      # SetFaceNormal, Hittable.cpp:8-13
        return dict(t=bt, p=o + d * bt, n=best.normal if front else -best.normal, tangent=best.tangent, tri=best, uv=buv)

    @staticmethod
    def to_local(w, rec):                                    # Material.h:84-92
        bit = np.cross(rec["tangent"], rec["n"])
        return np.array([w @ rec["tangent"], w @ bit, w @ rec["n"]])

    @staticmethod
    def to_world(l, rec):                                    # Material.h:76-83 (normalises)
        bit = np.cross(rec["tangent"], rec["n"])
        v = l[0] * rec["tangent"] + l[1] * bit + l[2] * rec["n"]
        return v * (1.0 / math.sqrt(v @ v))

    def cosine_hemisphere(self):                             # RandomNumberGenerator.h:39-64, B20: u.y is the first draw
        first, second = self.xi(), self.xi()
        ox, oy = 2.0 * second - 1.0, 2.0 * first - 1.0
        if ox == 0.0 and oy == 0.0:
            dx = dy = 0.0
        else:
            if abs(ox) > abs(oy):
                r, theta = ox, (math.pi / 4) * (oy / ox)
            else:
                r, theta = oy, math.pi / 2 - (math.pi / 4) * (ox / oy)
            dx, dy = r * math.cos(theta), r * math.sin(theta)
        return np.array([dx, dy, math.sqrt(max(0.0, 1.0 - dx * dx - dy * dy))])

    def scatter(self, mat, d, rec):
        """Material::Scatter -> (ok, attenuation, direction)."""
        if mat.type == _abi.MAT_LAMBERTIAN:                  # Material.h:106-151
            wi = self.cosine_hemisphere()
            while wi[2] <= 0.0:
                wi = self.cosine_hemisphere()
            pdf = wi[2] / math.pi
            f = self.kd(mat, rec) / math.pi
            return True, f * wi[2] / pdf, self.to_world(wi, rec)
        if mat.type == _abi.MAT_PHONG:
            return self.phong_scatter(mat, d, rec)
        if mat.type == _abi.MAT_MIRROR:                      # Material.h:334-363
            wo = self.to_local(-d, rec)
            nz = np.array([0.0, 0.0, 1.0])
            wi = -wo + 2.0 * (wo @ nz) * nz
            f = np.ones(3) / wi[2]
            return True, f * wi[2] / 1.0, self.to_world(wi, rec)
        return False, None, None                             # DiffuseLight / Empty: Material.h:57-59

    def ray_color(self, o, d, depth):                        # Camera.cpp:119-204, line by line
        if depth < 0:
            return np.zeros(3)
        rec = self.world_hit(o, d, 0.0001, INF)
        if rec is None:
            return self.background.copy()
        mat = rec["tri"].mat
        self.seen.add(mat.name)
        if mat.type == _abi.MAT_DIFFUSE_LIGHT:
            return np.array(mat.emission, dtype=np.float64)
        skip = mat.type in (_abi.MAT_MIRROR, _abi.MAT_EMPTY) or (mat.type == _abi.MAT_PHONG and mat.ns > 1.0)   # Material.h:328,365,539
        ps = rec["p"]
        direct, scat = np.zeros(3), np.zeros(3)
        if self.sample_lights and not skip:
            self.xi()                                        # HittableList::Sample's draw (one child)
            self.xi()                                        # BVHNode::Sample's sqrt(xi) * area (one leaf: nothing to choose)
            lt = self.lights[0]
            x, y = math.sqrt(self.xi()), self.xi()           # Triangle::Sample
            pl = lt.v[0] * (1.0 - x) + lt.v[1] * (x * (1.0 - y)) + lt.v[2] * (x * y)
            lfront = ((pl - ps) @ lt.normal) < 0.0
            lnormal = lt.normal if lfront else -lt.normal
            pdf = 1.0 / lt.area                              # one triangle: pdf * area / total area = 1 / area
            v = pl - ps
            dist = math.sqrt(v @ v)
            ldir = v * (1.0 / dist)
            sh = self.world_hit(ps, ldir, 0.001, 1.7976931348623157e308)
            visible = True if sh is None else (dist - math.sqrt((ps - sh["p"]) @ (ps - sh["p"]))) < 0.001   # B9
            if rec["n"] @ ldir > 0.0 and lfront and visible:
                lwi = self.to_local(ldir, rec)
                lln = self.to_local(lnormal, rec)
                if mat.type == _abi.MAT_PHONG:                # the draw happens only once the three conditions hold
                    fr = self.phong_eval(mat, lwi, self.to_local(-d, rec), rec)
                else:
                    fr = self.kd(mat, rec) / math.pi         # Lambertian::Eval
                direct = np.array(lt.mat.emission) * fr * lwi[2] * (lln @ -lwi) / (dist * dist) / pdf
        if self.xi() < self.rr:
            ok, att, wdir = self.scatter(mat, d, rec)
            if ok:
                if self.sample_lights:
                    nxt = self.world_hit(ps, wdir, 0.0001, INF)
                    if nxt is not None:
                        if nxt["tri"].mat.type != _abi.MAT_DIFFUSE_LIGHT or skip:
                            scat = att * self.ray_color(ps, wdir, depth - 1) / self.rr
                else:
                    scat = att * self.ray_color(ps, wdir, depth - 1) / self.rr
        return direct + scat

    def sample(self, cam, i, j, s, seed, depth):
        stream = oracle.rng_stream(seed, j * cam.width + i, s, 4096)
        self.rng = iter(stream)
        # Camera::Initialize + GetRay (Camera.cpp:75-117)
        eye, look, up = (np.array(x, dtype=np.float64) for x in (cam.eye, cam.look_at, cam.up))
        focal = math.sqrt((eye - look) @ (eye - look))
        h = math.tan(math.radians(cam.fovy) / 2)
        vh = 2 * h * focal
        vw = vh * (cam.width / cam.height)
        w = (eye - look) / focal
        u = np.cross(up, w)
        u = u / math.sqrt(u @ u)
        v = np.cross(w, u)
        vu, vv = vw * u, vh * -v
        du, dv = vu / cam.width, vv / cam.height
        ul = eye - focal * w - vu / 2 - vv / 2
        p00 = ul + 0.5 * (du + dv)
        return self.ray_color(eye, p00 + i * du + j * dv - eye, depth)


def crosscheck_scene():
    b = scenes._Builder("crosscheck")
    M = scenes.Material
    white = b.material(M("DiffuseWhite", _abi.MAT_LAMBERTIAN, kd=(0.73, 0.71, 0.68)))
    red = b.material(M("LeftWall", _abi.MAT_LAMBERTIAN, kd=(0.63, 0.065, 0.05)))
    light = b.material(M("Light", _abi.MAT_DIFFUSE_LIGHT, emission=(17.0, 12.0, 4.0)))
    mirror = b.material(M("Mirror", _abi.MAT_MIRROR))
    empty = b.material(M("quad1", _abi.MAT_EMPTY))
    b.mesh("floor", white, *scenes.quad((-1, -1, 1), (1, -1, 1), (1, -1, -1), (-1, -1, -1)))
    b.mesh("back", white, *scenes.quad((-1, -1, -1), (1, -1, -1), (1, 1, -1), (-1, 1, -1)))
    b.mesh("left", red, *scenes.quad((-1, -1, 1), (-1, -1, -1), (-1, 1, -1), (-1, 1, 1)))
    b.mesh("mirror", mirror, *scenes.quad((1, -1, -1), (1, -1, 1), (1, 1, 1), (1, 1, -1)))
    b.mesh("absorber", empty, *scenes.quad((-0.6, -0.999, 0.2), (-0.1, -0.999, 0.2), (-0.1, -0.999, -0.3), (-0.6, -0.999, -0.3)))
    b.mesh("blocker", white, *scenes.quad((0.1, 0.1, -0.2), (0.6, 0.1, -0.2), (0.6, 0.1, 0.3), (0.1, 0.1, 0.3)))
    rng = np.random.default_rng(3)
    b.textures.append(rng.integers(0, 256, size=(6, 5, 3), dtype=np.uint8))
    wood = b.material(M("Wood", _abi.MAT_LAMBERTIAN, kd=(1, 1, 1), texture=0))
    matte = b.material(M("material4", _abi.MAT_PHONG, kd=(0.4, 0.5, 0.3), ks=(0.2, 0.2, 0.2), ns=1.0))       # NEE + stochastic Eval
    glossy = b.material(M("material1", _abi.MAT_PHONG, kd=(0.2, 0.25, 0.3), ks=(0.5, 0.45, 0.4), ns=40.0))   # 0.6 / 0.4 lobes, no NEE
    b.mesh("panel", wood, *scenes.quad((-0.95, -0.6, -0.99), (-0.1, -0.6, -0.99), (-0.1, 0.4, -0.99), (-0.95, 0.4, -0.99)))
    b.mesh("matte", matte, *scenes.quad((0.2, -0.998, 0.4), (0.9, -0.998, 0.4), (0.9, -0.998, -0.1), (0.2, -0.998, -0.1)))
    b.mesh("glossy", glossy, *scenes.quad((-0.99, -0.9, 0.9), (-0.99, -0.9, 0.1), (-0.99, 0.2, 0.1), (-0.99, 0.2, 0.9)))
    lv = np.array([[[-0.3, 0.95, -0.3], [0.4, 0.95, -0.2], [-0.1, 0.95, 0.35]]])   # one triangle, facing down
    b.mesh("light", light, lv)
    return b.build(scenes.Camera(14, 12, 45.0, eye=(0.05, 0.07, 3.2), look_at=(0, 0, 0)))


@pytest.mark.parametrize("sample_lights,depth,spp", [(True, 6, 6), (True, 0, 3), (False, 5, 4)])
def test_python_raycolor_equals_oracle_per_sample(sample_lights, depth, spp):
    data = crosscheck_scene()
    cam = data.camera
    bg = (0.02, 0.03, 0.05)
    orc = oracle.Oracle(data)
    py = Tracer(data, rr=0.8, background=bg, sample_lights=sample_lights)
    # the light must face the room for NEE to contribute
    assert py.lights[0].normal[1] < 0
    px = [(i, j) for j in range(cam.height) for i in range(cam.width)]
    want = orc.render_samples(px, spp=spp, max_depth=depth, seed=9, rr=0.8, background=bg, sample_lights=sample_lights)
    worst, seen_direct, seen_mirror = 0.0, 0, 0
    for k, (i, j) in enumerate(px):
        for s in range(spp):
            got = py.sample(cam, i, j, s, 9, depth)
            ref = want[k, s]
            err = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
            worst = max(worst, err)
            assert err <= 1e-12, (i, j, s, got, ref)
            seen_direct += bool(got.any())
    assert seen_direct > 0.3 * len(px) * spp        # the comparison is not vacuous
    if depth > 0:
        assert {"DiffuseWhite", "LeftWall", "Light", "Mirror", "quad1", "Wood", "material4", "material1"} <= py.seen, py.seen
    assert worst <= 1e-12
