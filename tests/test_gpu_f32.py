"""GPU tests (-m gpu) of the fp32 fast mode (PRT_PRECISION_F32, prt_kernels_f32.hip) at the SECOND tolerance tier
(SURVEY.md §8d): the same kernels with every real number a float cannot agree with the fp64 oracle to 1e-9 — a hit
parameter carries fp32 rounding, and a path whose branch sits on a knife edge takes the other side — so the checks are

  * closest hits: |dt| / max(1, t) <= 1e-5 and the same primitive (or a neighbour at the same t) for all but 1e-4 of
    the rays (a ray that grazes an edge may slip past it in fp32 and hit what lies behind);
  * images at equal seeds: per pixel and channel |d| <= 3 sigma / sqrt(spp) + 1e-3 max(1, |x|), sigma = the pixel's
    per-sample standard deviation from the oracle's own samples, for all but 0.1 % of the pixels; image mean within 1e-3;
  * frames are reproducible bit for bit; tile shards are disjoint and sum to the full frame to fp32 rounding.

The fp64 path's parity tests (test_gpu_parity.py) are untouched by this mode: it has its own translation unit.
"""
import numpy as np
import pytest

import oracle
from pooraytracer_amd import _abi, api, scenes

pytestmark = pytest.mark.gpu
F32 = _abi.PRECISION_F32


def trace(sc, rays, precision):
    import torch
    rays = np.ascontiguousarray(rays, dtype=_abi.RAY_DTYPE)
    d_r = torch.from_numpy(rays.view(np.float64).reshape(-1, 8)).cuda()
    d_h = torch.zeros((rays.shape[0], 4), dtype=torch.float64, device="cuda")
    sc.trace_closest_device(d_r.data_ptr(), rays.shape[0], d_h.data_ptr(), precision=precision)
    torch.cuda.synchronize()
    return d_h.cpu().numpy().view(_abi.HIT_DTYPE).reshape(-1)


def check_hits_tier2(g, o, max_bad=1e-4):
    n = o.shape[0]
    flip = (g["prim"] >= 0) != (o["prim"] >= 0)
    both = (g["prim"] >= 0) & (o["prim"] >= 0)
    rel = np.abs(g["t"][both] - o["t"][both]) / np.maximum(1.0, o["t"][both])
    far = rel > 1e-5
    assert flip.sum() + far.sum() <= max_bad * n, (int(flip.sum()), int(far.sum()), n)
    # where t agrees the primitive is the same one, or its neighbour across a shared edge / a coplanar twin (same t)
    near = ~far
    same = g["prim"][both][near] == o["prim"][both][near]
    # (bathroom's fixtures share edges and stack coplanar faces: 0.4 % of its random rays end on such a pair)
    assert (~same).mean() <= 1e-2, float((~same).mean())
    ok = both.copy()
    ok[both] = near & (g["prim"][both] == o["prim"][both])
    assert np.allclose(g["alpha"][ok], o["alpha"][ok], atol=2e-3) and np.allclose(g["beta"][ok], o["beta"][ok], atol=2e-3)
    return float(rel[near].max()) if near.any() else 0.0


@pytest.mark.parametrize("scene_fn,nrays", [
    (lambda: scenes.cornell_box(), 200_000),
    (lambda: scenes.bathroom(detail=0.5), 100_000),
    (lambda: scenes.mixed_materials(40, 40), 50_000),
])
def test_f32_closest_hits_within_tier2_of_the_oracle(gpu, scene_fn, nrays):
    data = scene_fn()
    lo, hi = data.bounds()
    rays = scenes.random_rays(nrays, lo, hi, seed=2024)
    want = oracle.Oracle(data).trace_closest(rays)
    sc = api.Scene(data).upload(gpu)
    got = trace(sc, rays, F32)
    check_hits_tier2(got, want)
    # and the fp64 kernel on the same scene object is still the exact one (both table sets are resident)
    exact = trace(sc, rays, _abi.PRECISION_F64)
    hit = want["prim"] >= 0
    assert np.array_equal(exact["prim"] >= 0, hit)
    assert (np.abs(exact["t"][hit] - want["t"][hit]) <= 1e-12 * np.maximum(1.0, want["t"][hit])).all()


@pytest.mark.parametrize("device_bvh", [False, True])
def test_f32_hits_with_either_builder_and_a_deep_tree(gpu, device_bvh):
    """A soup whose 4-wide tree can need more than 32 stack entries: the host builder hands the fp32 kernels its second,
    shallower collapse; a device-built tree keeps the 40-entry stacks.  Either way the hits are the oracle's."""
    data = scenes.triangle_soup(n_tris=150_000, seed=5)
    lo, hi = data.bounds()
    rays = scenes.random_rays(60_000, lo, hi, seed=77)
    want = oracle.Oracle(data).trace_closest(rays)
    sc = api.Scene(data, device_bvh=device_bvh).upload(gpu)
    check_hits_tier2(trace(sc, rays, F32), want)


@pytest.mark.parametrize("scene_fn,spp,depth", [
    (lambda: scenes.cornell_box(ball_subdiv=2, width=48, height=48), 64, 10),
    (lambda: scenes.mixed_materials(40, 40), 64, 12),
    (lambda: scenes.veach_mis(64, 36, light_subdiv=2, plate_cells=2), 48, 20),
    (lambda: scenes.bathroom(48, 28, detail=0.12), 32, 12),
])
def test_f32_images_within_tier2_of_the_oracle(gpu, scene_fn, spp, depth):
    data = scene_fn()
    cam = data.camera
    orc = oracle.Oracle(data)
    px = np.stack(np.meshgrid(np.arange(cam.width), np.arange(cam.height)), -1).reshape(-1, 2)
    samples = orc.render_samples(px, spp=spp, max_depth=depth, seed=9)          # (pixels, spp, 3) radiance of every sample
    ref = samples.mean(axis=1).reshape(cam.height, cam.width, 3)
    sigma = samples.std(axis=1, ddof=1).reshape(cam.height, cam.width, 3)
    sc = api.Scene(data).upload(gpu)
    img = sc.render(spp=spp, max_depth=depth, seed=9, precision=F32)
    assert np.isfinite(img).all()
    tol = 3.0 * sigma / np.sqrt(spp) + 1e-3 * np.maximum(1.0, np.abs(ref))
    bad = (np.abs(img - ref) > tol).any(-1)
    assert bad.mean() <= 1e-3, f"{bad.sum()} of {bad.size} pixels outside 3 sigma / sqrt(spp) + 1e-3"
    assert abs(img.mean() - ref.mean()) <= 1e-3 * abs(ref.mean())
    # the bulk of the pixels took exactly the oracle's paths: they agree to fp32 rounding, not merely statistically
    close = (np.abs(img - ref) <= 1e-4 * np.maximum(1.0, np.abs(ref))).all(-1)
    assert close.mean() >= 0.97, float(close.mean())
    cnt = sc.counters()
    assert cnt["samples"] == cam.width * cam.height * spp


def test_f32_full_size_frame_against_the_fp64_kernels(gpu):
    """BASELINE config 2's frame size at reduced spp: the fp32 and the fp64 frame of the same seeds differ in a few
    pixels (diverged paths) and agree in the mean; tile shards of the fp32 frame sum to it (fp32 rounding); it is reproducible."""
    data = scenes.cornell_box()
    sc = api.Scene(data).upload(gpu)
    kw = dict(spp=4, max_depth=20, seed=1)
    a = sc.render(**kw)
    b = sc.render(precision=F32, **kw)
    assert abs(b.mean() - a.mean()) <= 1e-3 * a.mean()
    far = (np.abs(a - b) > 1e-3 * np.maximum(1.0, np.abs(a))).any(-1)
    assert far.mean() <= 0.01, float(far.mean())
    assert np.array_equal(b, sc.render(precision=F32, **kw))
    # tile shards: a share of the frame cuts a pixel's samples into different chunks, and a chunk's sum is an fp32 one here
    # (the fp64 kernels' shards agree with the full frame to the last fp32 bit; these to fp32 rounding)
    parts = [sc.render(precision=F32, rank=r, nranks=3, tile_size=16, **kw) for r in range(3)]
    whole = sc.render(precision=F32, tile_size=16, **kw)
    assert np.allclose(parts[0] + parts[1] + parts[2], whole, rtol=1e-5, atol=1e-9)
    owner = [(p != 0).any(-1) for p in parts]
    assert not (owner[0] & owner[1]).any() and not (owner[0] & owner[2]).any() and not (owner[1] & owner[2]).any()
    info = sc.bvh_info()
    assert info["width"] == 4


def test_f32_after_moving_geometry_and_bad_precision(gpu):
    data = scenes.cornell_box(ball_subdiv=2, width=32, height=32)
    sc = api.Scene(data).upload(gpu)
    first = sc.render(spp=8, max_depth=6, seed=2, precision=F32)
    moved = data.vertices * 1.0
    moved[..., 1] += 0.05
    sc.update_vertices(moved)                       # re-upload: the float tables are rebuilt on the next fp32 call
    second = sc.render(spp=8, max_depth=6, seed=2, precision=F32)
    assert np.isfinite(second).all() and not np.array_equal(first, second)
    exact = sc.render(spp=8, max_depth=6, seed=2)
    assert abs(second.mean() - exact.mean()) <= 2e-3 * exact.mean()
    with pytest.raises(api.PrtError):
        sc.render(spp=1, precision=7)


def test_f32_through_the_cpp_camera(gpu, tmp_path):
    """Camera::bFloatPrecision (include/pooraytracer/Camera.h) selects the fp32 kernels: the main.cpp-style driver must
    produce the frame the binding produces with precision=F32 (same C ABI, same inputs: bit-identical)."""
    import os
    import subprocess
    from pooraytracer_amd import build
    exe = build.build_host_example()
    data = scenes.mixed_materials(40, 32)
    dump = str(tmp_path / "scene.bin")
    scenes.dump_scene(data, dump)
    out = str(tmp_path / "out.f64")
    r = subprocess.run([exe, dump, "6", "8", out], capture_output=True, text=True, timeout=300, env=dict(os.environ, PRT_EXAMPLE_F32="1"))
    assert r.returncode == 0, r.stderr + r.stdout
    img = np.fromfile(out, dtype=np.float64).reshape(32, 40, 3)
    sc = api.Scene(data).upload(gpu)
    assert np.array_equal(img, sc.render(spp=6, max_depth=8, seed=1, precision=F32))
    assert not np.array_equal(img, sc.render(spp=6, max_depth=8, seed=1))


def test_f32_render_options_follow_the_fp64_kernels(gpu):
    """Every switch of PrtRenderParams means the same thing in either precision: depth 0, no NEE, a background colour,
    Russian roulette 1.0, pixel jitter, explicit sample chunks — compared with the fp64 kernels' frame of the same seeds
    (which test_gpu_parity.py holds against the oracle)."""
    data = scenes.mixed_materials(48, 40)
    sc = api.Scene(data).upload(gpu)
    for kw in (dict(max_depth=0), dict(sample_lights=False, background=(0.2, 0.3, 0.5)), dict(rr=1.0, max_depth=4),
               dict(pixel_jitter=True), dict(sample_chunks=3), dict(max_depth=-1)):
        kw = dict(dict(spp=32, max_depth=8, seed=11), **kw)
        a, b = sc.render(**kw), sc.render(precision=F32, **kw)
        assert np.isfinite(b).all()
        if a.mean() == 0.0:
            assert b.mean() == 0.0
            continue
        assert abs(b.mean() - a.mean()) <= 2e-3 * a.mean(), kw
        close = (np.abs(a - b) <= 1e-3 * np.maximum(1.0, np.abs(a))).all(-1)
        assert close.mean() >= 0.95, (kw, float(close.mean()))
