"""A SECOND, independent restatement of the reference's central routine, checked against the oracle sample by sample.

`oracle/pt_oracle.cpp` cannot be pinned against the reference (no fixtures, unbuildable here).  What can be done is to
write `Camera::RayColor` and what it calls a second time, from the reference's text and in another language, and require
both restatements to agree on every sample: a transcription slip in either one shows up as a difference.  This file is
that second restatement — plain Python / numpy, brute-force closest hit over all triangles (no BVH), recursion exactly as
in the reference:

    Camera::GetRay / RayColor            Source/Camera.cpp:108-204
    Triangle ctor, Hit, IsInterior, Sample   Source/Triangle.cpp:11-113
    HitRecord::SetFaceNormal             Source/Hittable.cpp:8-13
    AABB / Interval (union, PadToMinimus, LongestAxis)   Source/AABB.cpp:10-36,66-82, Source/Interval.h:8-37
    BVHNode::BVHNode (median split, std::sort IN PLACE on the mesh's own triangle list)   Source/BVH.cpp:6-49
    the two-level lights object graph    main.cpp:36-45 (every mesh is built twice: for `world`, then for `lights`)
    HittableList::Sample, BVHNode::Sample, TraverseSample (float p)   Source/HittableList.h:44-59, Source/BVH.cpp:62-67,86-100
    std::sort                            libstdc++ 11 (bits/stl_algo.h, bits/stl_heap.h): introsort = median-of-3 quicksort to
                                         depth 2*log2(n), heapsort below that, one final insertion sort — restated here because
                                         the order of triangles with EQUAL keys (which the unstable sort leaves wherever its
                                         swaps put them) decides which light triangle a given random number picks
    Material::WorldToLocal / LocalToWorld    Source/Material.h:76-98
    Lambertian, PhoneReflectance, PerfectMirror, DiffuseLight, DebugMaterial, EmptyMaterial   Source/Material.h:101-366,523-540
    CookTorrance (D, Lambda, G1, G, visible-normal SampleWm, Sample, Eval, Scatter), Complex, FrComplex
                                         Source/Material.h:368-521, Source/MaterialUtils.h:6-111, RandomNumberGenerator.h:69-73
    ImageTexture::Value / GetPixel / SRGBToLinear   Source/Texture.cpp:22-71
    SampleCosineHemisphere / SampleUniformDiskConcentric   Source/RandomNumberGenerator.h:39-64

It takes from the oracle ONLY the random numbers (`oracle.rng_stream`: the keyed stream both sides must share, departure
B2) and compares with `orc.render_samples` per (pixel, sample) to 1e-12.  Documented departures are applied here as
there: B9 (an escaping shadow ray is unoccluded), B13 (Phong's unassigned attenuation is 0), B20 (`dvec2(xi, xi)`: y is
drawn first).  Phong is restated with the library calls the reference makes (acos, sin, cos, pow), so agreement to 1e-9 here
also covers the algebraic equivalents the GPU uses through the oracle.  The light selection (tree build with the sort's side
effects, float-p descent) is compared BIT FOR BIT with `orc.sample_lights` on the multi-mesh light lists of the veach-mis and
bathroom stand-ins.
"""
import math

import numpy as np
import pytest

import oracle
from pooraytracer_amd import _abi, scenes

INF = float("inf")


def dot3(a, b):                                              # glm::dot: a.x*b.x + a.y*b.y + a.z*b.z, left to right
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def cross3(x, y):                                            # glm::cross
    return np.array([x[1] * y[2] - y[1] * x[2], x[2] * y[0] - y[2] * x[0], x[0] * y[1] - y[0] * x[1]])


def normalize3(v):                                           # glm::normalize = v * inversesqrt(dot(v, v))
    with np.errstate(divide="ignore", invalid="ignore"):
        return v * (1.0 / np.sqrt(np.float64(dot3(v, v))))


# ---------------------------------------------------------------------------- Interval / AABB (Interval.h, AABB.cpp)
class Box:
    """AABB: three [min, max] intervals."""

    def __init__(self, lo, hi):
        self.lo, self.hi = list(lo), list(hi)

    def pad(self):                                           # AABB::PadToMinimus, AABB.cpp:76-82 (Interval::Expand, Interval.h:33-36)
        for a in range(3):
            if self.hi[a] - self.lo[a] < 0.0001:
                self.lo[a], self.hi[a] = self.lo[a] - 0.0001 / 2.0, self.hi[a] + 0.0001 / 2.0
        return self

    @staticmethod
    def empty():                                             # AABB::empty, AABB.cpp:7 with Interval::empty = [+inf, -inf] (departure B1)
        return Box([INF] * 3, [-INF] * 3).pad()

    @staticmethod
    def of_points(a, b):                                     # AABB(const vec3&, const vec3&), AABB.cpp:16-22
        return Box([a[k] if a[k] <= b[k] else b[k] for k in range(3)], [b[k] if a[k] <= b[k] else a[k] for k in range(3)]).pad()

    @staticmethod
    def union(p, q):                                         # AABB(const AABB&, const AABB&), AABB.cpp:24-29 + Interval(a, b), Interval.h:14-17: no padding
        return Box([p.lo[k] if p.lo[k] <= q.lo[k] else q.lo[k] for k in range(3)], [p.hi[k] if p.hi[k] >= q.hi[k] else q.hi[k] for k in range(3)])

    def longest_axis(self):                                  # AABB.cpp:66-74
        lx, ly, lz = (self.hi[k] - self.lo[k] for k in range(3))
        if lx > ly:
            return 0 if lx > lz else 2
        return 1 if ly > lz else 2


# ---------------------------------------------------------------------------- std::sort of libstdc++ (GCC 11)
def std_sort(a, first, last, less):
    """std::sort(a + first, a + last, less) exactly as libstdc++ does it (bits/stl_algo.h: __sort, __introsort_loop,
    __unguarded_partition_pivot, __move_median_to_first, __final_insertion_sort; bits/stl_heap.h for the fallback).  Only
    the order of elements that compare equal depends on these details — and that order is what BVH.cpp:31 leaves behind."""
    def lg(n):
        return n.bit_length() - 1

    def swap(i, j):
        a[i], a[j] = a[j], a[i]

    def move_median_to_first(result, x, y, z):
        if less(a[x], a[y]):
            if less(a[y], a[z]):
                swap(result, y)
            elif less(a[x], a[z]):
                swap(result, z)
            else:
                swap(result, x)
        elif less(a[x], a[z]):
            swap(result, x)
        elif less(a[y], a[z]):
            swap(result, z)
        else:
            swap(result, y)

    def unguarded_partition(f, l, pivot):
        while True:
            while less(a[f], a[pivot]):
                f += 1
            l -= 1
            while less(a[pivot], a[l]):
                l -= 1
            if not f < l:
                return f
            swap(f, l)
            f += 1

    def push_heap(base, hole, top, value):
        parent = (hole - 1) // 2
        while hole > top and less(a[base + parent], value):
            a[base + hole] = a[base + parent]
            hole = parent
            parent = (hole - 1) // 2
        a[base + hole] = value

    def adjust_heap(base, hole, length, value):
        top = hole
        child = hole
        while child < (length - 1) // 2:
            child = 2 * (child + 1)
            if less(a[base + child], a[base + child - 1]):
                child -= 1
            a[base + hole] = a[base + child]
            hole = child
        if (length & 1) == 0 and child == (length - 2) // 2:
            child = 2 * (child + 1)
            a[base + hole] = a[base + child - 1]
            hole = child - 1
        push_heap(base, hole, top, value)

    def heap_sort(f, l):                                     # std::partial_sort(f, l, l) = __heap_select (make_heap only) + __sort_heap
        length = l - f
        if length >= 2:
            parent = (length - 2) // 2
            while True:
                adjust_heap(f, parent, length, a[f + parent])
                if parent == 0:
                    break
                parent -= 1
        while l - f > 1:
            l -= 1
            value = a[l]
            a[l] = a[f]
            adjust_heap(f, 0, l - f, value)

    def introsort_loop(f, l, depth):
        while l - f > 16:
            if depth == 0:
                heap_sort(f, l)
                return
            depth -= 1
            mid = f + (l - f) // 2
            move_median_to_first(f, f + 1, mid, l - 1)
            cut = unguarded_partition(f + 1, l, f)
            introsort_loop(cut, l, depth)
            l = cut

    def unguarded_linear_insert(i):
        val = a[i]
        nxt = i - 1
        while less(val, a[nxt]):
            a[i] = a[nxt]
            i = nxt
            nxt -= 1
        a[i] = val

    def insertion_sort(f, l):
        for i in range(f + 1, l):
            if less(a[i], a[f]):
                val = a[i]
                a[f + 1:i + 1] = a[f:i]
                a[f] = val
            else:
                unguarded_linear_insert(i)

    if first == last:
        return
    introsort_loop(first, last, 2 * lg(last - first))
    if last - first > 16:
        insertion_sort(first, first + 16)
        for i in range(first + 16, last):
            unguarded_linear_insert(i)
    else:
        insertion_sort(first, last)


class Tri:
    """Triangle::Triangle (Triangle.cpp:11-53)."""

    def __init__(self, v, uv, mat, prim=-1):
        self.v = [np.array(x, dtype=np.float64) for x in v]
        self.uv = [np.array(x, dtype=np.float64) for x in uv]
        self.e0, self.e1 = self.v[1] - self.v[0], self.v[2] - self.v[0]
        n = cross3(self.e0, self.e1)
        self.normal = normalize3(n)
        d0, d1 = self.uv[1] - self.uv[0], self.uv[2] - self.uv[0]
        with np.errstate(divide="ignore", invalid="ignore"):
            f = np.float64(1.0) / (d0[0] * d1[1] - d1[0] * d0[1])
            t = normalize3(f * (d1[1] * self.e0 - d0[1] * self.e1))
        if np.isnan(t).any():
            helper = np.array([1.0, 0, 0]) if abs(self.normal[0]) < float(np.float32(0.9)) else np.array([0, 1.0, 0])
            t = normalize3(cross3(self.normal, helper))
        self.tangent = t
        self.area = math.sqrt(dot3(n, n)) * 0.5               # length(n) * 0.5
        self.D = dot3(self.normal, self.v[0])
        self.w = n / dot3(n, n)
        self.mat = mat
        self.prim = prim
        # SetBoundingBox, Triangle.cpp:94-99
        self.bbox = Box.union(Box.of_points(self.v[0], self.v[1]), Box.of_points(self.v[0], self.v[2]))

    def hit(self, o, d, tmin, tmax):
        """Triangle::Hit + IsInterior (Triangle.cpp:54-83,100-113); returns t or None."""
        denom = dot3(self.normal, d)
        if abs(denom) < 1e-8:
            return None
        t = (self.D - dot3(self.normal, o)) / denom
        if not (tmin <= t <= tmax):
            return None
        p = o + d * t
        v0p = p - self.v[0]
        alpha = dot3(self.w, cross3(v0p, self.e1))
        beta = dot3(self.w, cross3(self.e0, v0p))
        if alpha != alpha or beta != beta or alpha < 0 or beta < 0 or alpha + beta > 1:
            return None
        self.last_uv = (1.0 - alpha - beta) * self.uv[0] + alpha * self.uv[1] + beta * self.uv[2]   # IsInterior, :111
        return t

    def sample(self, origin, xi):
        """Triangle::Sample (Triangle.cpp:84-93): position, face-forwarded normal, bFrontFace, pdf = 1 / area."""
        x = math.sqrt(xi())
        y = xi()
        p = self.v[0] * (1.0 - x) + self.v[1] * (x * (1.0 - y)) + self.v[2] * (x * y)
        front = dot3(p - origin, self.normal) < 0.0          # SetFaceNormal(Ray(origin, p - origin), normal), Hittable.cpp:8-13
        return p, (self.normal if front else -self.normal), front, 1.0 / self.area


# ---------------------------------------------------------------------------- BVHNode (BVH.cpp) and the lights list (main.cpp:36-45)
class Node:
    """BVHNode::BVHNode(objects, start, end), BVH.cpp:7-48.  `objects` is sorted IN PLACE, like the reference's vector."""

    def __init__(self, objects, start, end):
        self.bbox = Box.empty()
        for k in range(start, end):
            self.bbox = Box.union(self.bbox, objects[k].bbox)
        axis = self.bbox.longest_axis()
        span = end - start
        if span == 1:
            self.left = self.right = objects[start]
            self.area = objects[start].area
        elif span == 2:
            self.left, self.right = objects[start], objects[start + 1]
            self.area = objects[start].area + objects[start + 1].area
        else:
            std_sort(objects, start, end, lambda p, q: p.bbox.lo[axis] < q.bbox.lo[axis])   # BoxCompare, BVH.cpp:68-73
            mid = start + span // 2
            self.left = Node(objects, start, mid)
            self.right = Node(objects, mid, end)
            self.area = self.left.area + self.right.area     # both children are BVHNodes (BVH.cpp:38-40)

    def sample(self, origin, xi):
        """BVHNode::Sample (BVH.cpp:62-67) -> TraverseSample (BVH.cpp:86-100) with its `float p`."""
        p = np.float32(math.sqrt(xi()) * self.area)          # double, converted to float at the call
        node = self
        while isinstance(node, Node):
            if float(p) < node.left.area:                    # float promoted to double for the comparison
                node = node.left
            else:
                p = np.float32(float(p) - node.left.area)    # double arithmetic, converted to float at the call
                node = node.right
        pos, n, front, pdf = node.sample(origin, xi)
        pdf *= node.area
        pdf /= self.area
        return node, pos, n, front, pdf


def is_emissive(mat):                                        # HasEmission(): DiffuseLight (Material.h:166) and DebugMaterial (:528)
    return mat.type in (_abi.MAT_DIFFUSE_LIGHT, _abi.MAT_DEBUG)


def emission_of(mat):                                        # GetEmission(): the light's radiance / the debug material's albedo
    return np.array(mat.emission if mat.type == _abi.MAT_DIFFUSE_LIGHT else mat.kd, dtype=np.float64)


def build_lights(meshes):
    """main.cpp:36-45.  `meshes` = list of (triangle list, material).  Every mesh is handed to BVHNode(shared_ptr<Mesh>)
    once for `world` — whose std::sort permutes mesh->objects — and, if emissive, a second time for `lights`, which
    therefore starts from the order the first build left behind.  Then lights = HittableList(BVHNode(lights)): a tree over
    the per-mesh trees, built from a COPY of the list (BVHNode(HittableList list) takes it by value)."""
    lights = []
    for tris, mat in meshes:
        if not tris:
            continue
        Node(tris, 0, len(tris))                             # world.Add(make_shared<BVHNode>(mesh))
        if is_emissive(mat):
            lights.append(Node(tris, 0, len(tris)))          # lights.Add(make_shared<BVHNode>(mesh))
    if not lights:
        return None
    return Node(list(lights), 0, len(lights))


def lights_sample(top, origin, xi):
    """lights.Sample(): HittableList::Sample (HittableList.h:44-59) over its one object, then BVHNode::Sample."""
    area_sum = 0.0 + top.area
    p = xi() * area_sum
    assert p <= area_sum                                     # the only object is always the one picked
    return top.sample(origin, xi)


def mesh_lists(data):
    out = []
    for m in range(len(data.mesh_material)):
        mat = data.materials[int(data.mesh_material[m])]
        out.append(([Tri(data.vertices[t], data.texcoords[t], mat, prim=t)
                     for t in range(int(data.mesh_first_tri[m]), int(data.mesh_first_tri[m + 1]))], mat))
    return out


# ---------------------------------------------------------------------------- MaterialUtils.h
class Cx:
    """Complex<double> (MaterialUtils.h:6-44)."""

    def __init__(self, re, im=0.0):
        self.re, self.im = re, im

    def __add__(self, z):
        return Cx(self.re + z.re, self.im + z.im)

    def __sub__(self, z):
        return Cx(self.re - z.re, self.im - z.im)

    def __mul__(self, z):
        return Cx(self.re * z.re - self.im * z.im, self.re * z.im + self.im * z.re)

    def __truediv__(self, z):
        scale = 1 / (z.re * z.re + z.im * z.im)
        return Cx(scale * (self.re * z.re + self.im * z.im), scale * (self.im * z.re - self.re * z.im))


def cx_norm(z):
    return z.re * z.re + z.im * z.im


def cx_sqrt(z):                                              # MaterialUtils.h:54-65
    n = math.sqrt(cx_norm(z))
    t1 = math.sqrt(.5 * (n + abs(z.re)))
    if n == 0:
        return Cx(0.0)
    t2 = .5 * z.im / t1
    if z.re >= 0:
        return Cx(t1, t2)
    return Cx(abs(t2), math.copysign(t1, z.im))


def sqr(v):
    return v * v


def clamp(v, lo, hi):
    return lo if v < lo else (hi if v > hi else v)


def fr_complex(cos_i, eta):                                  # MaterialUtils.h:100-111
    cos_i = clamp(cos_i, 0, 1)
    sin2_i = 1 - cos_i * cos_i
    sin2_t = Cx(sin2_i) / (eta * eta)
    cos_t = cx_sqrt(Cx(1) - sin2_t)
    r_parl = (eta * Cx(cos_i) - cos_t) / (eta * Cx(cos_i) + cos_t)
    r_perp = (Cx(cos_i) - eta * cos_t) / (Cx(cos_i) + eta * cos_t)
    return (cx_norm(r_parl) + cx_norm(r_perp)) / 2


def cos2theta(w):
    return w[2] * w[2]


def sin2theta(w):
    return max(0., 1 - cos2theta(w))


def tan2theta(w):
    s, c = sin2theta(w), cos2theta(w)
    return s / c if c != 0 else (INF if s > 0 else float("nan"))


def cosphi(w):
    st = math.sqrt(sin2theta(w))
    return 1 if st == 0 else clamp(w[0] / st, -1, 1)


def sinphi(w):
    st = math.sqrt(sin2theta(w))
    return 0 if st == 0 else clamp(w[1] / st, -1, 1)


class CookTorrance:
    """Source/Material.h:368-521 on local directions."""

    def __init__(self, mat):
        self.ax, self.ay = mat.alpha_x, mat.alpha_y
        self.eta, self.k = mat.eta, mat.k

    def D(self, wm):                                         # :373-380
        t2 = tan2theta(wm)
        if math.isinf(t2):
            return 0
        cos4 = sqr(cos2theta(wm))
        e = t2 * (sqr(cosphi(wm) / self.ax) + sqr(sinphi(wm) / self.ay))
        return 1 / (math.pi * self.ax * self.ay * cos4 * sqr(1 + e))

    def Lambda(self, w):                                     # :381-386
        t2 = tan2theta(w)
        if math.isinf(t2):
            return 0
        a2 = sqr(cosphi(w) * self.ax) + sqr(sinphi(w) * self.ay)
        return (math.sqrt(1 + a2 * t2) - 1) / 2

    def G1(self, w):
        return 1 / (1 + self.Lambda(w))

    def G(self, wo, wi):
        return 1 / (1 + self.Lambda(wo) + self.Lambda(wi))

    def Dv(self, w, wm):                                     # D(w, wm), :391-393
        return self.G1(w) / abs(w[2]) * self.D(wm) * abs(dot3(w, wm))

    def fresnel(self, wo, wm):
        c = abs(dot3(wo, wm))
        return np.array([fr_complex(c, Cx(self.eta[k], self.k[k])) for k in range(3)])

    def sample_wm(self, w, u):                               # :412-435, u = (u.x, u.y)
        wh = normalize3(np.array([self.ax * w[0], self.ay * w[1], w[2]]))
        if wh[2] < 0:
            wh = -wh
        T1 = normalize3(cross3(np.array([0., 0., 1.]), wh)) if wh[2] < 0.99999 else np.array([1., 0, 0])
        T2 = cross3(wh, T1)
        r, theta = math.sqrt(u[0]), 2 * math.pi * u[1]       # SampleUniformDiskPolar, RandomNumberGenerator.h:69-73
        px, py = r * math.cos(theta), r * math.sin(theta)
        h = math.sqrt(1 - px * px)
        x = (1 + wh[2]) / 2
        py = (1 - x) * h + x * py                            # Lerp((1 + wh.z) / 2, h, p.y)
        pz = math.sqrt(max(0., 1. - (px * px + py * py)))
        nh = px * T1 + py * T2 + pz * wh
        return normalize3(np.array([self.ax * nh[0], self.ay * nh[1], max(1e-6, nh[2])]))

    def sample(self, wo, xi):                                # :437-472 -> (f, pdf, wi) or None (flags Unset)
        if wo[2] == 0:
            return None
        first, second = xi(), xi()                           # vec2(RandomDouble(), RandomDouble()): g++ evaluates right to left (B20)
        wm = self.sample_wm(wo, (second, first))
        wi = -wo + 2.0 * dot3(wo, wm) * wm                   # Reflect(wo, wm), Material.h:94-98
        if not wo[2] * wi[2] > 0:
            return None
        pdf = self.Dv(wo, wm) / (4. * abs(dot3(wo, wm)))
        co, ci = abs(wo[2]), abs(wi[2])
        if ci == 0 or co == 0:
            return None
        f = self.D(wm) * self.fresnel(wo, wm) * self.G(wo, wi) / (4. * ci * co)
        return f, pdf, wi

    def eval(self, wi, wo):                                  # :474-496
        if not wo[2] * wi[2] > 0:
            return np.zeros(3)
        co, ci = abs(wo[2]), abs(wi[2])
        if ci == 0 or co == 0:
            return np.zeros(3)
        wm = wi + wo
        if wm[0] * wm[0] + wm[1] * wm[1] + wm[2] * wm[2] == 0:
            return np.zeros(3)
        wm = normalize3(wm)
        return self.D(wm) * self.fresnel(wo, wm) * self.G(wo, wi) / (4 * ci * co)


class Tracer:
    def __init__(self, data, rr, background, sample_lights=True):
        meshes = mesh_lists(data)
        self.tris = [t for tris, _ in meshes for t in tris]  # description order (the brute-force closest hit needs no tree)
        self.lights = build_lights([(list(tris), mat) for tris, mat in meshes])
        self.rr, self.background, self.sample_lights = rr, np.array(background, dtype=np.float64), sample_lights
        self.textures = data.textures
        self.rng = None
        self.seen = set()   # materials a path vertex landed on (the comparison must not be vacuous)
        # the triangles' constants side by side, for Triangle::Hit over all of them at once (same expressions, elementwise)
        T = self.tris
        self.N = np.array([t.normal for t in T])
        self.Dd = np.array([t.D for t in T])
        self.W = np.array([t.w for t in T])
        self.V0 = np.array([t.v[0] for t in T])
        self.E0 = np.array([t.e0 for t in T])
        self.E1 = np.array([t.e1 for t in T])

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

    # world.Hit: closest accepted triangle (HittableList.h:26-39 semantics; no exact ties in these scenes).  Triangle::Hit +
    # IsInterior (Triangle.cpp:54-83,100-113) evaluated for every triangle at once: the same expressions as Tri.hit, elementwise.
    def world_hit(self, o, d, tmin, tmax):
        N, W, E0, E1 = self.N, self.W, self.E0, self.E1
        with np.errstate(divide="ignore", invalid="ignore", over="ignore"):
            denom = N[:, 0] * d[0] + N[:, 1] * d[1] + N[:, 2] * d[2]
            t = (self.Dd - (N[:, 0] * o[0] + N[:, 1] * o[1] + N[:, 2] * o[2])) / denom
            P = o[None, :] + d[None, :] * t[:, None]
            Q = P - self.V0
            cx = np.stack([Q[:, 1] * E1[:, 2] - E1[:, 1] * Q[:, 2], Q[:, 2] * E1[:, 0] - E1[:, 2] * Q[:, 0], Q[:, 0] * E1[:, 1] - E1[:, 0] * Q[:, 1]], axis=1)
            alpha = W[:, 0] * cx[:, 0] + W[:, 1] * cx[:, 1] + W[:, 2] * cx[:, 2]
            cy = np.stack([E0[:, 1] * Q[:, 2] - Q[:, 1] * E0[:, 2], E0[:, 2] * Q[:, 0] - Q[:, 2] * E0[:, 0], E0[:, 0] * Q[:, 1] - Q[:, 0] * E0[:, 1]], axis=1)
            beta = W[:, 0] * cy[:, 0] + W[:, 1] * cy[:, 1] + W[:, 2] * cy[:, 2]
            ok = (np.abs(denom) >= 1e-8) & (tmin <= t) & (t <= tmax) & (alpha >= 0) & (beta >= 0) & (alpha + beta <= 1)
        if not ok.any():
            return None
        tt = np.where(ok, t, INF)
        k = int(len(tt) - 1 - np.argmin(tt[::-1]))           # of equal t the later-tested triangle wins (inclusive interval)
        best, bt = self.tris[k], float(tt[k])
        a, b = float(alpha[k]), float(beta[k])
        buv = (1.0 - a - b) * best.uv[0] + a * best.uv[1] + b * best.uv[2]
        front = dot3(d, best.normal) < 0.0                   # SetFaceNormal, Hittable.cpp:8-13
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
        if mat.type == _abi.MAT_COOKTORRANCE:                # Material.h:497-516
            wo = normalize3(self.to_local(-d, rec))
            r = CookTorrance(mat).sample(wo, self.xi)
            if r is None:
                return False, None, None
            f, pdf, wi = r
            return True, f * wi[2] / pdf, self.to_world(wi, rec)
        return False, None, None                             # DiffuseLight / Debug / Empty: Material.h:57-59

    def ray_color(self, o, d, depth):                        # Camera.cpp:119-204, line by line
        if depth < 0:
            return np.zeros(3)
        rec = self.world_hit(o, d, 0.0001, INF)
        if rec is None:
            return self.background.copy()
        mat = rec["tri"].mat
        self.seen.add(mat.name)
        if is_emissive(mat):
            return emission_of(mat)
        skip = mat.type in (_abi.MAT_MIRROR, _abi.MAT_EMPTY) or (mat.type == _abi.MAT_PHONG and mat.ns > 1.0)   # Material.h:328,365,539
        ps = rec["p"]
        direct, scat = np.zeros(3), np.zeros(3)
        if self.sample_lights and self.lights is not None and not skip:
            lt, pl, lnormal, lfront, pdf = lights_sample(self.lights, ps, self.xi)
            v = pl - ps
            ldir = normalize3(v)
            dist = math.sqrt(dot3(v, v))
            sh = self.world_hit(ps, ldir, 0.001, 1.7976931348623157e308)
            visible = True if sh is None else (dist - math.sqrt(dot3(ps - sh["p"], ps - sh["p"]))) < 0.001   # B9
            if dot3(rec["n"], ldir) > 0.0 and lfront and visible:
                lwi = self.to_local(ldir, rec)
                lln = self.to_local(lnormal, rec)
                if mat.type == _abi.MAT_PHONG:                # the draw happens only once the three conditions hold
                    fr = self.phong_eval(mat, lwi, self.to_local(-d, rec), rec)
                elif mat.type == _abi.MAT_COOKTORRANCE:       # context.wo = WorldToLocal(-ray.direction), NOT normalised here (Camera.cpp:163)
                    fr = CookTorrance(mat).eval(lwi, self.to_local(-d, rec))
                else:
                    fr = self.kd(mat, rec) / math.pi         # Lambertian::Eval
                direct = emission_of(lt.mat) * fr * lwi[2] * dot3(lln, -lwi) / (dist * dist) / pdf
        if self.xi() < self.rr:
            ok, att, wdir = self.scatter(mat, d, rec)
            if ok:
                if self.sample_lights:
                    nxt = self.world_hit(ps, wdir, 0.0001, INF)
                    if nxt is not None:
                        if not is_emissive(nxt["tri"].mat) or skip:
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
    assert py.lights.left.left.normal[1] < 0                 # top node -> the mesh's node -> its one triangle
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


def test_std_sort_restatement_sorts_and_is_the_library_s_on_ties():
    """The restated introsort is a sort (all three phases exercised), and on keys with many ties it leaves the permutation
    libstdc++'s std::sort leaves — checked through the one place the oracle exposes it: the light order of a mesh whose
    triangles tie on the split axis (test below).  Here: plain sortedness on adversarial inputs, incl. the heapsort fallback."""
    rng = np.random.default_rng(2)
    calls = {"n": 0}
    for n in (1, 2, 3, 16, 17, 100, 1000):
        for keys in (rng.integers(0, 5, n), rng.random(n), np.arange(n)[::-1], np.arange(n), np.zeros(n)):
            a = [(float(k), i) for i, k in enumerate(keys)]
            std_sort(a, 0, n, lambda p, q: p[0] < q[0])
            assert [x[0] for x in a] == sorted(float(k) for k in keys)
            assert sorted(x[1] for x in a) == list(range(n))
    # median-of-3 killer-ish input: organ pipe keys force deep recursion; the result must still be sorted
    n = 4000
    keys = list(range(n // 2)) + list(range(n // 2, 0, -1))
    a = list(keys)
    std_sort(a, 0, n, lambda p, q: (calls.__setitem__("n", calls["n"] + 1) or p < q))
    assert a == sorted(keys) and calls["n"] < 40 * n * 12


@pytest.mark.parametrize("scene_fn,n_origins", [("veach_mis", 100_000), ("bathroom", 100_000), ("mixed_materials", 20_000)])
def test_light_selection_equals_oracle_bit_for_bit(scene_fn, n_origins):
    """BVHNode::BVHNode + main.cpp:36-45 + HittableList::Sample + BVHNode::Sample / TraverseSample (float p) +
    Triangle::Sample, restated above from the reference's text, against `orc.sample_lights`: for every seeded origin the
    same triangle, the same point, normal and face flag, and the same pdf — to the last bit.  veach-mis has five emissive
    meshes of 1280 triangles (sphere lights: many equal keys per split), bathroom two quads, mixed-materials a quad light
    and an emissive debug material."""
    data = getattr(scenes, scene_fn)()
    orc = oracle.Oracle(data)
    top = build_lights(mesh_lists(data))
    order = []

    def walk(n):
        if isinstance(n, Node):
            walk(n.left)
            if n.right is not n.left:
                walk(n.right)
        else:
            order.append(n.prim)
    walk(top)
    assert order == orc.light_order().tolist()               # leaf order of the area-CDF descent
    lo, hi = data.bounds()
    rng = np.random.default_rng(17)
    origins = lo + rng.random((n_origins, 3)) * (hi - lo)
    want = orc.sample_lights(origins, seed=4)
    picked = set()
    for i in range(n_origins):
        st = iter(oracle.rng_stream(4, i, 0, 4))
        tri, pos, nrm, front, pdf = lights_sample(top, origins[i], lambda: next(st))
        w = want[i]
        assert tri.prim == w["prim"] and pdf == w["pdf"] and bool(front) == bool(w["front"]), (i, tri.prim, w)
        assert (pos == w["position"]).all() and (nrm == w["normal"]).all(), (i, pos, w)
        picked.add(tri.prim)
    assert len(picked) >= min(len(order), 3000) * 0.9        # the descent reached (nearly) every light triangle


def test_python_raycolor_equals_oracle_on_mixed_materials():
    """The same per-sample comparison on the stand-in that has every material kind — incl. CookTorrance (Scatter and the
    NEE Eval), the emissive debug material, image textures, both Phong regimes — and a light list of two meshes."""
    data = scenes.mixed_materials()
    cam = data.camera
    orc = oracle.Oracle(data)
    py = Tracer(data, rr=0.8, background=(0.0, 0.0, 0.0), sample_lights=True)
    rng = np.random.default_rng(8)
    px = [(int(i), int(j)) for i, j in zip(rng.integers(0, cam.width, 260), rng.integers(0, cam.height, 260))]
    spp, depth = 3, 6
    want = orc.render_samples(px, spp=spp, max_depth=depth, seed=11, rr=0.8)
    worst, lit = 0.0, 0
    for k, (i, j) in enumerate(px):
        for s in range(spp):
            got = py.sample(cam, i, j, s, 11, depth)
            ref = want[k, s]
            err = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
            worst = max(worst, err)
            assert err <= 1e-12, (i, j, s, got, ref)
            lit += bool(got.any())
    assert lit > 0.3 * len(px) * spp
    assert {"Gold", "Debug", "Wood", "Mirror", "Light"} <= py.seen, py.seen


@pytest.mark.parametrize("seed", [9001, 9002, 9003, 9004])
def test_python_raycolor_equals_oracle_on_random_scenes(seed):
    """... and on scenes nobody designed: the random soups of the GPU fuzz test (all material kinds with random parameters,
    1- / 3- / 4-channel textures, uv outside [0, 1], one or two light meshes, random camera).  tools/crosscheck_campaign.py
    runs the same comparison over hundreds of seeds (profiles/r03_crosscheck_campaign.json)."""
    try:
        from tests.test_gpu_parity import _random_scene
    except ImportError:
        from test_gpu_parity import _random_scene
    data = _random_scene(seed)
    cam = data.camera
    orc = oracle.Oracle(data)
    py = Tracer(data, rr=0.8, background=(0.1, 0.2, 0.3), sample_lights=True)
    rng = np.random.default_rng(seed)
    px = [(int(i), int(j)) for i, j in zip(rng.integers(0, cam.width, 30), rng.integers(0, cam.height, 30))]
    spp, depth = 2, 8
    want = orc.render_samples(px, spp=spp, max_depth=depth, seed=seed + 3, rr=0.8, background=(0.1, 0.2, 0.3))
    for k, (i, j) in enumerate(px):
        for s in range(spp):
            got = py.sample(cam, i, j, s, seed + 3, depth)
            err = np.abs(got - want[k, s]).max() / max(1.0, np.abs(want[k, s]).max())
            assert err <= 1e-12, (seed, i, j, s, got, want[k, s])
