"""CPU emulation (numpy float32, one rounding per operation like the device code) of the fp32 slab test of the 4-wide
BVH node (pooraytracer_amd/csrc/prt_device.h: slab_axis / Trav::box4).  Two properties:

* conservative: whenever the exact slab test (AABB::Hit, AABB.cpp:38-64, in float64 on the dequantised box) accepts a
  box, the fp32 test accepts it too — for scenes at the origin, scenes translated far away and rays starting far outside;
* an unused child slot (inverted range lo = 0xffff, hi = 0) is rejected by its range for every ray within ~2^18 grid
  extents of the scene, and can be accepted beyond that — which is why the traversal checks the slot's ref as well.
"""
import numpy as np

f32 = np.float32
PAD = f32(9.5367432e-7)  # 2^-20


def fma32(a, b, c):
    # a, b, c float32: the product of two float32 is exact in float64; one rounding to float32 at the end
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(f32)


def slab_axis(o, d, E, g0, gs):
    """prt_device.h slab_axis (fp64 origin): returns idq, c_lo (entry), c_hi (exit), rot."""
    df = d.astype(f32)
    with np.errstate(divide="ignore"):
        idv = (f32(1.0) / df).astype(f32)  # v_rcp_f32 is within 1 ulp of this
    idv = np.where(np.abs(idv) <= f32(1e28), idv, np.copysign(f32(1e28), df)).astype(f32)
    r = (np.float64(g0) - o).astype(f32)
    c = (r * idv).astype(f32)
    pad = ((np.abs(idv) * (np.abs(r) + f32(E)).astype(f32)).astype(f32) * PAD).astype(f32)
    return (f32(gs) * idv).astype(f32), (c - pad).astype(f32), (c + pad).astype(f32), idv < 0


def box_test(o, d, lo_q, hi_q, E, g0, gs, tmin=f32(1e-4), tmax=f32(np.inf)):
    """Trav::box4 with PRT_BOX_ROTATE: entry / exit per axis from the rotated range, n <= f accepts."""
    n = np.full(o.shape[0], tmin, dtype=f32)
    f = np.full(o.shape[0], tmax, dtype=f32)
    for a in range(3):
        idq, c_lo, c_hi, neg = slab_axis(o[:, a], d[:, a], E, g0[a], gs[a])
        ent = np.where(neg, hi_q[:, a], lo_q[:, a]).astype(f32)
        ext = np.where(neg, lo_q[:, a], hi_q[:, a]).astype(f32)
        n = np.maximum(n, fma32(ent, idq, c_lo))
        f = np.minimum(f, fma32(ext, idq, c_hi))
    return n <= f


def exact_test(o, d, lo, hi, tmin=1e-4):
    """AABB::Hit in float64 on the dequantised box."""
    t0 = np.full(o.shape[0], tmin)
    t1 = np.full(o.shape[0], np.inf)
    with np.errstate(divide="ignore", invalid="ignore"):
        for a in range(3):
            inv = 1.0 / d[:, a]
            ta, tb = (lo[:, a] - o[:, a]) * inv, (hi[:, a] - o[:, a]) * inv
            t0 = np.maximum(t0, np.minimum(ta, tb))
            t1 = np.minimum(t1, np.maximum(ta, tb))
    return t1 > t0


def grid_for(center, extent):
    g0 = (np.asarray(center, dtype=np.float64) - extent / 2).astype(f32)
    gs = np.full(3, np.nextafter(f32(extent / 65535.0), f32(np.inf)), dtype=f32)
    E = np.nextafter(f32(65535.0 * float(gs.max())), f32(np.inf))
    return g0, gs, E


def rays_towards(rng, n, center, extent, dist):
    """Origins `dist` away from the scene centre (0 = inside the scene), aimed at points inside the scene."""
    tgt = center + (rng.random((n, 3)) - 0.5) * extent
    if dist == 0:
        o = center + (rng.random((n, 3)) - 0.5) * extent
    else:
        u = rng.normal(size=(n, 3))
        o = center + u / np.linalg.norm(u, axis=1, keepdims=True) * dist
    d = tgt - o
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    return o, d


def test_fp32_slab_test_is_conservative_near_far_and_translated():
    rng = np.random.default_rng(7)
    n = 20000
    for center in ((0.0, 0.0, 0.0), (1e6, -3e5, 2e6), (1e9, 1e9, 1e9)):
        center = np.asarray(center)
        for extent in (2.0, 1e-2):
            g0, gs, E = grid_for(center, extent)
            for dist in (0.0, 10 * extent, 1e4 * extent, 1e7 * extent):
                o, d = rays_towards(rng, n, center, extent, dist)
                lo_q = rng.integers(0, 65000, size=(n, 3))
                hi_q = lo_q + rng.integers(1, 535, size=(n, 3))
                lo = g0.astype(np.float64) + lo_q * gs.astype(np.float64)
                hi = g0.astype(np.float64) + hi_q * gs.astype(np.float64)
                ex = exact_test(o, d, lo, hi)
                got = box_test(o, d, lo_q, hi_q, E, g0, gs)
                assert not (ex & ~got).any(), (center, extent, dist, int((ex & ~got).sum()))
                if dist <= 10 * extent and np.abs(center).max() <= 1e6 * extent:
                    # ... and still culls: relative to the grid origin a far-away scene is tested like one at the origin
                    assert got.mean() < ex.mean() + 0.05, (center, extent, dist, got.mean(), ex.mean())


def test_unused_slot_range_is_rejected_near_the_scene_and_not_beyond():
    rng = np.random.default_rng(11)
    n = 20000
    center = np.array([1e6, 1e6, 1e6])
    extent = 2.0
    g0, gs, E = grid_for(center, extent)
    lo_q = np.full((n, 3), 0xFFFF)
    hi_q = np.zeros((n, 3), dtype=np.int64)
    for dist in (0.0, 100.0, 2.0 ** 17 * extent):
        o, d = rays_towards(rng, n, center, extent, dist)
        assert not box_test(o, d, lo_q, hi_q, E, g0, gs).any(), dist
    # from 2^21 extents away the pad exceeds the whole grid on every axis: the range alone no longer rejects the slot
    o, d = rays_towards(rng, n, center, extent, 2.0 ** 21 * extent)
    assert box_test(o, d, lo_q, hi_q, E, g0, gs).any()
    # the formula this replaces padded by the WORLD coordinate: a unit scene at 1e6 accepted the empty slot for rays inside it
    o, d = rays_towards(rng, n, center, extent, 0.0)
    B = f32(np.abs(center).max() + extent)
    acc = np.ones(n, dtype=bool)
    nn = np.full(n, f32(1e-4))
    ff = np.full(n, f32(np.inf))
    for a in range(3):
        df = d[:, a].astype(f32)
        idv = (f32(1.0) / df).astype(f32)
        of = o[:, a].astype(f32)
        c = ((g0[a] - of).astype(f32) * idv).astype(f32)
        pad = ((np.abs(idv) * (np.abs(of) + B).astype(f32)).astype(f32) * PAD).astype(f32)
        idq = (gs[a] * idv).astype(f32)
        neg = idv < 0
        ent = np.where(neg, 0, 0xFFFF).astype(f32)
        ext = np.where(neg, 0xFFFF, 0).astype(f32)
        nn = np.maximum(nn, fma32(ent, idq, (c - pad).astype(f32)))
        ff = np.minimum(ff, fma32(ext, idq, (c + pad).astype(f32)))
    acc = nn <= ff
    assert acc.any()
