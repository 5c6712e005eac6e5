"""CPU tests of the oracle (the restated reference algorithm): hand-derived known answers,
invariants and the committed golden fixtures.  PARITY UNPINNED: the reference ships no vectors."""
import json
import os

import numpy as np
import pytest

import oracle
from pooraytracer_amd import _abi, scenes

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def one_triangle(material=None):
    b = scenes._Builder("one")
    m = b.material(material or scenes.Material("DiffuseWhite", _abi.MAT_LAMBERTIAN, kd=(0.5, 0.5, 0.5)))
    v = np.array([[[0, 0, 0], [1, 0, 0], [0, 1, 0]]], dtype=np.float64)
    uv = np.array([[[0, 0], [1, 0], [0, 1]]], dtype=np.float64)
    b.mesh("t", m, v, uv)
    return b.build(scenes.Camera(4, 4, 30.0, (0.25, 0.25, 2.0), (0.25, 0.25, 0.0)))


def mkrays(o, d, tmin=1e-4, tmax=np.inf):
    o = np.atleast_2d(np.asarray(o, dtype=np.float64))
    d = np.atleast_2d(np.asarray(d, dtype=np.float64))
    r = np.zeros(o.shape[0], dtype=_abi.RAY_DTYPE)
    r["o"], r["d"], r["tmin"], r["tmax"] = o, d, tmin, tmax
    return r


def test_triangle_hit_known_answers():
    """Triangle::Hit (Triangle.cpp:54-83): t, barycentrics, face flag, inclusive interval, parallel reject."""
    o = oracle.Oracle(one_triangle())
    h = o.trace_closest(mkrays([[0.25, 0.25, 2.0]], [[0, 0, -1]]))
    assert h["prim"][0] == 0 and h["t"][0] == 2.0 and h["alpha"][0] == 0.25 and h["beta"][0] == 0.25 and h["front"][0] == 1
    # unnormalised direction: t is in units of |d|
    h = o.trace_closest(mkrays([[0.25, 0.25, 2.0]], [[0, 0, -4]]))
    assert h["t"][0] == 0.5
    # from behind: back face
    h = o.trace_closest(mkrays([[0.25, 0.25, -1.0]], [[0, 0, 1]]))
    assert h["prim"][0] == 0 and h["front"][0] == 0
    # outside the triangle (alpha+beta > 1), and parallel ray (|n.d| < 1e-8)
    h = o.trace_closest(mkrays([[0.75, 0.75, 2.0], [0.25, 0.25, 1.0]], [[0, 0, -1], [1, 0, 0]]))
    assert (h["prim"] == -1).all() and np.isinf(h["t"]).all()
    # inclusive interval (Interval.h:21-23): t == tmax and t == tmin are hits, just outside is not
    h = o.trace_closest(mkrays([[0.25, 0.25, 2.0]] * 3, [[0, 0, -1]] * 3, tmin=[1e-4, 2.0, 2.0000001], tmax=[2.0, 5.0, 5.0]))
    assert h["prim"].tolist() == [0, 0, -1]
    # edge/vertex hits are accepted (alpha == 0 or beta == 0)
    h = o.trace_closest(mkrays([[0.5, 0.0, 1.0], [0.0, 0.0, 1.0]], [[0, 0, -1]] * 2))
    assert h["prim"].tolist() == [0, 0]


def test_closest_of_many_is_minimum_t():
    sc = scenes.tiny_scene()
    o = oracle.Oracle(sc)
    lo, hi = sc.bounds()
    rays = scenes.random_rays(2000, lo, hi, seed=3)
    h = o.trace_closest(rays)
    # brute force in numpy over all triangles with the same formulas
    v = sc.vertices
    e0, e1 = v[:, 1] - v[:, 0], v[:, 2] - v[:, 0]
    n = np.cross(e0, e1)
    nn = n / np.linalg.norm(n, axis=1, keepdims=True)
    D = (nn * v[:, 0]).sum(1)
    w = n / (n * n).sum(1, keepdims=True)
    for i in range(0, 2000, 37):
        oo, dd = rays["o"][i], rays["d"][i]
        den = nn @ dd
        with np.errstate(divide="ignore", invalid="ignore"):
            t = (D - nn @ oo) / den
        p = oo + dd * t[:, None]
        v0p = p - v[:, 0]
        al = (w * np.cross(v0p, e1)).sum(1)
        be = (w * np.cross(e0, v0p)).sum(1)
        ok = (np.abs(den) >= 1e-8) & (t >= 1e-4) & (al >= 0) & (be >= 0) & (al + be <= 1)
        if ok.any():
            tt = np.where(ok, t, np.inf)
            assert h["prim"][i] >= 0 and abs(h["t"][i] - tt.min()) <= 1e-12 * max(1, tt.min())
        else:
            assert h["prim"][i] == -1


def test_rng_stream_properties():
    """Keyed RNG: 31-bit granularity like rand()/(RAND_MAX+1.0), in [0,1), keyed independence."""
    a = oracle.rng_stream(1, 5, 7, 4096)
    assert (a >= 0).all() and (a < 1).all()
    assert np.array_equal(a * 2147483648.0, np.floor(a * 2147483648.0))
    assert abs(a.mean() - 0.5) < 0.03
    b = oracle.rng_stream(1, 5, 8, 4096)
    c = oracle.rng_stream(2, 5, 7, 4096)
    assert not np.array_equal(a, b) and not np.array_equal(a, c)
    assert np.array_equal(a, oracle.rng_stream(1, 5, 7, 4096))


def test_seeds_share_no_streams():
    """The seed is hashed on its own before it meets (pixel, sample): no (sample) stream of one seed reappears under a
    neighbouring seed (folded linearly into one key, seed 2 had reused seed 1's streams with sample indices swapped in
    pairs — averaging frames of different seeds then reduced no variance)."""
    first = {s: {oracle.rng_stream(s, 5, k, 1)[0] for k in range(512)} for s in (1, 2, 3, 1 << 32, (1 << 32) + 1)}
    seeds = list(first)
    for i, a in enumerate(seeds):
        assert len(first[a]) >= 510  # 31-bit values: a chance repeat among 512 is possible, a systematic one is not
        for b in seeds[i + 1:]:
            assert len(first[a] & first[b]) <= 1, (a, b)
    # and on the quantity that matters: per-sample radiance of one pixel under seed 1 and seed 2
    sc = scenes.tiny_scene()
    o = oracle.Oracle(sc)
    px = [[sc.camera.width // 2, sc.camera.height // 2 + 5]]
    r1 = o.render_samples(px, spp=256, max_depth=6, seed=1)[0]
    r2 = o.render_samples(px, spp=256, max_depth=6, seed=2)[0]
    lit1 = {tuple(v) for v in r1 if v.any()}
    lit2 = {tuple(v) for v in r2 if v.any()}
    assert len(lit1) > 100 and len(lit1 & lit2) == 0


def test_camera_rays_pinhole():
    """Camera::Initialize/GetRay (Camera.cpp:75-117): pixel-centre rays, unnormalised, symmetric."""
    cam = scenes.Camera(8, 4, 90.0, (0, 0, 0), (0, 0, -1))
    r = oracle.camera_rays(cam)
    assert np.allclose(r[..., :3], 0)
    d = r[..., 3:]
    assert np.allclose(d[..., 2], -1.0)  # focal length 1
    # fovy 90 => viewport height 2, width 4; pixel (0,0) centre = (-2+0.25, 1-0.25)
    assert np.allclose(d[0, 0, :2], [-1.75, 0.75])
    assert np.allclose(d[3, 7, :2], [1.75, -0.75])
    assert np.allclose(d[:, :, 0], -d[:, ::-1, 0]) and np.allclose(d[:, :, 1], -d[::-1, :, 1])


def test_light_sampling_pdf_and_skew():
    """lights.Sample: pdf = 1/total area; points lie on light triangles; sqrt-skewed pick (BVH.cpp:64)."""
    sc = scenes.tiny_scene()
    o = oracle.Oracle(sc)
    org = np.zeros((20000, 3))
    s = o.sample_lights(org, seed=9)
    assert np.allclose(s["pdf"], 1.0 / 0.25)  # 0.5 x 0.5 quad
    assert np.allclose(s["position"][:, 1], 0.998)
    assert (np.abs(s["position"][:, [0, 2]]) <= 0.25 + 1e-12).all()
    assert (s["front"] == 1).all() and np.allclose(s["normal"], [0, -1, 0])
    order = o.light_order()
    # p = sqrt(xi)*A: P(first leaf) = P(sqrt(xi) < 1/2) = 1/4 for two equal-area triangles
    frac_first = (s["prim"] == order[0]).mean()
    assert abs(frac_first - 0.25) < 0.02


def test_render_furnace_and_determinism():
    """White-furnace style check: inside a closed emissive box every path returns Le at its first hit."""
    b = scenes._Builder("furnace")
    m = b.material(scenes.Material("Light", _abi.MAT_DIFFUSE_LIGHT, emission=(0.5, 0.25, 2.0)))
    v, uv = scenes.box((-1, -1, -1), (1, 1, 1))
    b.mesh("box", m, v, uv)
    sc = b.build(scenes.Camera(8, 8, 60.0, (0, 0, 0), (0, 0, -1)))
    o = oracle.Oracle(sc)
    img, cnt = o.render(spp=3, max_depth=5)
    assert np.allclose(img, [0.5, 0.25, 2.0], rtol=1e-15)
    assert cnt["rays_closest"] == 8 * 8 * 3 and cnt["rays_shadow"] == 0


def test_render_background_and_depth_zero():
    sc = one_triangle()
    o = oracle.Oracle(sc)
    img, _ = o.render(spp=2, max_depth=0, background=(0.1, 0.2, 0.3), sample_lights=False)
    # pixels that miss the triangle see the background; hit pixels: depth 0 => scatter recursion returns 0
    miss = np.isclose(img, [0.1, 0.2, 0.3]).all(-1)
    assert miss.any() and (~miss).any()
    assert np.allclose(img[~miss], 0.0)


def test_pixel_jitter_softens_edges_only():
    """pixel_jitter=1 (the AA commented out at Camera.cpp:110-111): with depth 0 and a background, a pixel's
    value is background x (fraction of jittered rays that miss).  Interior/exterior pixels are unchanged,
    edge pixels become fractional, and the default (0) stays the reference's pixel-centre behaviour."""
    sc = one_triangle()
    o = oracle.Oracle(sc)
    kw = dict(spp=64, max_depth=0, background=(1.0, 1.0, 1.0), sample_lights=False)
    hard, _ = o.render(**kw)
    soft, _ = o.render(pixel_jitter=True, **kw)
    assert set(np.unique(hard)) <= {0.0, 1.0}
    frac = (soft > 1e-9) & (soft < 1 - 1e-9)
    assert frac.any()
    # fractional pixels sit on the silhouette: a 3x3 neighbourhood of the hard image holds both values
    jj, ii = np.nonzero(frac[..., 0])
    for j, i in zip(jj, ii):
        nb = hard[max(j - 1, 0):j + 2, max(i - 1, 0):i + 2, 0]
        assert nb.min() == 0.0 and nb.max() == 1.0
    assert abs(soft.mean() - hard.mean()) < 0.1  # tiny image: coverage is the same up to edge quantisation
    # the offset uses the first two numbers of the sample stream: y first, then x
    u = oracle.rng_stream(1, 0, 0, 2)
    assert 0.0 <= u[0] < 1.0 and 0.0 <= u[1] < 1.0


def test_peek_reuse_is_identical_and_threads_do_not_matter():
    sc = scenes.mixed_materials(24, 24)
    o = oracle.Oracle(sc)
    a, ca = o.render(spp=4, max_depth=6, reuse_peek=True, nthreads=1)
    b, cb = o.render(spp=4, max_depth=6, reuse_peek=False, nthreads=3)
    assert np.array_equal(a, b)
    assert ca["rays_closest"] == cb["rays_closest"] and cb["hit_calls"] > ca["hit_calls"]


@pytest.mark.parametrize("name", ["tiny_cornell", "mixed"])
def test_golden_fixture(name):
    """Regression fixtures generated by tests/golden/make_golden.py from this oracle (self-pinned)."""
    path = os.path.join(GOLD, f"{name}.npz")
    g = np.load(path)
    meta = json.loads(str(g["meta"]))
    sc = scenes.tiny_scene() if name == "tiny_cornell" else scenes.mixed_materials()
    o = oracle.Oracle(sc)
    img, _ = o.render(spp=meta["spp"], max_depth=meta["max_depth"], seed=meta["seed"])
    assert np.allclose(img, g["image"], rtol=1e-12, atol=1e-14)
    h = o.trace_closest(g["rays"].view(_abi.RAY_DTYPE).reshape(-1))
    assert np.array_equal(h["prim"], g["hit_prim"])
    assert np.allclose(h["t"], g["hit_t"], rtol=1e-13)
    assert np.array_equal(o.light_order(), g["light_order"])


def _direct_light_scene():
    """A Lambertian floor (kd 0.5) at y=0 under ONE emissive triangle at y=1 facing down, seen from above."""
    b = scenes._Builder("direct")
    floor = b.material(scenes.Material("DiffuseWhite", _abi.MAT_LAMBERTIAN, kd=(0.5, 0.5, 0.5)))
    light = b.material(scenes.Material("Light", _abi.MAT_DIFFUSE_LIGHT, emission=(10.0, 8.0, 6.0)))
    fv = np.array([[[-4, 0, -4], [-4, 0, 4], [4, 0, 4]], [[-4, 0, -4], [4, 0, 4], [4, 0, -4]]], dtype=np.float64)
    b.mesh("floor", floor, fv, np.zeros((2, 3, 2)))
    lv = np.array([[[-0.5, 1, -0.5], [0.5, 1, -0.5], [0.0, 1, 0.5]]], dtype=np.float64)  # normal (0,-1,0)
    b.mesh("light", light, lv, np.zeros((1, 3, 2)))
    # the eye sits beside the light, so no camera ray of the sampled pixels meets the light itself
    return b.build(scenes.Camera(8, 8, 20.0, (2.5, 3.0, 0.0), (0.6, 0.0, 0.1), up=(0.0, 0.0, -1.0))), lv[0]


def test_direct_lighting_matches_the_area_integral():
    """Physics check that does not lean on the reference: with maxDepth 0 a pixel's radiance is the next-event
    estimate alone (Camera.cpp:137-172), whose expectation for one Lambertian point under one emissive triangle is
    L = integral over the light of Le * (kd/pi) * cos(theta_x) * cos(theta_l) / r^2 dA.  The oracle's mean over
    many samples must agree with a dense quadrature of that integral (a single light triangle, so the sqrt-skewed
    CDF of BVH.cpp:62-67 cannot bias the pick)."""
    sc, tri = _direct_light_scene()
    o = oracle.Oracle(sc)
    rays = oracle.camera_rays(sc.camera)
    spp = 20000
    for (i, j) in ((3, 3), (5, 2)):
        org, d = rays[j, i, :3], rays[j, i, 3:]
        t = -org[1] / d[1]                       # floor y = 0
        x = org + t * d
        h = o.trace_closest(mkrays([org], [d]))
        assert h["prim"][0] in (0, 1) and abs(h["t"][0] - t) < 1e-12
        # quadrature: n x n sub-triangles of the light, midpoint rule
        n = 400
        u, v = np.meshgrid((np.arange(n) + 0.5) / n, (np.arange(n) + 0.5) / n, indexing="ij")
        keep = u + v < 1.0
        u, v = u[keep], v[keep]
        p = tri[0] + u[:, None] * (tri[1] - tri[0]) + v[:, None] * (tri[2] - tri[0])
        w = (x - p)
        r2 = (w * w).sum(1)
        cos_l = np.abs(w[:, 1]) / np.sqrt(r2)    # light normal (0,-1,0), floor normal (0,1,0)
        cos_x = cos_l
        area = 0.5 * np.linalg.norm(np.cross(tri[1] - tri[0], tri[2] - tri[0]))
        geom = (cos_x * cos_l / r2).mean() * area * (keep.sum() / (0.5 * n * n))  # cells cut by the hypotenuse
        expect = np.array([10.0, 8.0, 6.0]) * (0.5 / np.pi) * geom
        s = o.render_samples([(i, j)], spp=spp, max_depth=0, seed=5)[0]
        mean, sem = s.mean(0), s.std(0) / np.sqrt(spp)
        assert np.all(np.abs(mean - expect) < 4 * sem + 2e-3 * expect), (mean, expect, sem)


def test_explicit_lights_list_changes_only_the_light_selection():
    """OrcSceneDesc.light_meshes = the `lights` argument of Camera::Render (Camera.cpp:137-139 samples the list it is
    given).  With only one of two emitters in the list, NEE must pick points on that emitter only and with pdf
    1 / its area; the default (NULL) is main.cpp:40-45's list of every emissive mesh, in mesh order."""
    import dataclasses
    b = scenes._Builder("two-lights")
    white = b.material(scenes.Material("DiffuseWhite", 0, kd=(0.5, 0.5, 0.5)))
    l1 = b.material(scenes.Material("light1", 4, emission=(3, 3, 3)))
    l2 = b.material(scenes.Material("light2", 4, emission=(1, 2, 3)))
    b.mesh("floor", white, *scenes.quad((-2, 0, -2), (2, 0, -2), (2, 0, 2), (-2, 0, 2)))
    b.mesh("a", l1, *scenes.quad((-1, 2, -1), (0, 2, -1), (0, 2, 0), (-1, 2, 0)))          # area 1
    b.mesh("b", l2, *scenes.quad((0.5, 2, 0.5), (2.5, 2, 0.5), (2.5, 2, 2.5), (0.5, 2, 2.5)))  # area 4
    data = b.build(scenes.Camera(8, 8, 40.0, eye=(0, 1, 6), look_at=(0, 1, 0)))
    org = np.zeros((4000, 3))
    both = oracle.Oracle(data).sample_lights(org, seed=5)
    assert set(np.unique(both["prim"])) == {2, 3, 4, 5} and np.allclose(both["pdf"], 1 / 5.0)
    assert np.array_equal(oracle.Oracle(dataclasses.replace(data, light_meshes=[1, 2])).sample_lights(org, seed=5)["prim"], both["prim"])
    only_b = oracle.Oracle(dataclasses.replace(data, light_meshes=[2])).sample_lights(org, seed=5)
    assert set(np.unique(only_b["prim"])) == {4, 5} and np.allclose(only_b["pdf"], 1 / 4.0)
    assert (only_b["position"][:, 0] >= 0.5).all()
    assert oracle.Oracle(dataclasses.replace(data, light_meshes=[])).light_order().size == 0
    assert list(oracle.Oracle(dataclasses.replace(data, light_meshes=[2, 1])).light_order()) != list(oracle.Oracle(data).light_order())
