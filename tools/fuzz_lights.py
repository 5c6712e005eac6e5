"""Light-pick campaign (round 4: the O(1) threshold / bucket tables of prt_device.h, sample_lights): N random scenes with 1-7
emissive meshes of 1-4000 triangles each — spheres, displaced grids, single triangles, duplicated vertices, zero-area slivers,
meshes whose areas differ by 10^6 — and, per scene, 60,000 light samples from the GPU (prt_sample_lights: K3's own device
function) against the oracle's restatement of HittableList::Sample -> BVHNode::Sample -> TraverseSample -> Triangle::Sample
(BVH.cpp:62-67,86-100, Triangle.cpp:84-93): the same triangle, face flag and pdf BIT FOR BIT (position to 1e-13, normal to 1e-15: the GPU's fast sqrt); plus one frame
per scene against the oracle at 1e-9.  Also reports how many scenes got tables and how many picks went through one.
Writes gpurun_out/r04_fuzz_lights.json."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch  # noqa: F401
from pooraytracer_amd import _abi, api, scenes
from pooraytracer_amd.scenes import Camera, Material, _Builder, grid_quad, icosphere, quad
import oracle


def random_light_scene(seed):
    rng = np.random.default_rng(seed)
    b = _Builder(f"lights{seed}")
    wall = b.material(Material("DiffuseWhite", _abi.MAT_LAMBERTIAN, kd=(0.7, 0.7, 0.7)))
    v, uv = grid_quad((-4, -1.5, 4), (8, 0, 0), (0, 0, -8), 6, 6)
    b.mesh("floor", wall, v, uv)
    n_lights = int(rng.integers(1, 8))
    for i in range(n_lights):
        m = b.material(Material(f"light{i % 4 + 1}" if i < 4 else "Light", _abi.MAT_DIFFUSE_LIGHT, emission=tuple(rng.uniform(0.5, 20.0, 3))))
        kind = int(rng.integers(0, 5))
        c = rng.uniform(-3, 3, 3) + np.array([0, 2.0, 0])
        scale = float(10.0 ** rng.uniform(-2.5, 0.5))
        if kind == 0:
            vv, uu, nn = icosphere(int(rng.integers(0, 5)), radius=scale, center=tuple(c))
            b.mesh(f"L{i}", m, vv, uu, nn)
        elif kind == 1:
            nu, nv = int(rng.integers(1, 45)), int(rng.integers(1, 45))
            amp = float(rng.uniform(0, 0.3)) * scale
            disp = (lambda S, T, a=amp: a * np.sin(7 * S) * np.cos(5 * T))
            vv, uu = grid_quad(tuple(c), tuple(rng.normal(size=3) * scale), tuple(rng.normal(size=3) * scale), nu, nv, displace=disp if amp > 0 else None)
            b.mesh(f"L{i}", m, vv, uu)
        elif kind == 2:  # a single triangle (span-1 node over a triangle) or a pair
            vv, uu = quad(tuple(c), tuple(c + rng.normal(size=3) * scale), tuple(c + rng.normal(size=3) * scale), tuple(c + rng.normal(size=3) * scale))
            k = int(rng.integers(1, 3))
            b.mesh(f"L{i}", m, vv[:k], uu[:k])
        elif kind == 3:  # many copies of the same few triangles: equal keys for std::sort, equal thresholds
            vv, uu = grid_quad(tuple(c), (scale, 0, 0), (0, 0, scale), 2, 2)
            reps = int(rng.integers(2, 40))
            b.mesh(f"L{i}", m, np.concatenate([vv] * reps), np.concatenate([uu] * reps))
        else:  # slivers: areas down to ~1e-14 next to ordinary triangles
            vv, uu = grid_quad(tuple(c), (scale, 0, 0), (0, scale * 1e-7, scale), int(rng.integers(2, 30)), 3)
            vv = vv.copy()
            vv[::3, 2] = vv[::3, 0]  # collapse one vertex of every third triangle towards another
            b.mesh(f"L{i}", m, vv, uu)
    cam = Camera(40, 30, 50.0, eye=(0.3, 2.5, 9.0), look_at=(0.0, 0.5, 0.0))
    return b.build(cam)


n = int(os.environ.get("FUZZ_N", "200"))
first = int(os.environ.get("FUZZ_FIRST", "7000"))
tot = bad = scenes_with_tables = px_tot = px_bad = 0
tris_seen = []
failed = {}
t0 = time.time()
for seed in range(first, first + n):
    data = random_light_scene(seed)
    orc = oracle.Oracle(data)
    sc = api.Scene(data, device_bvh=bool(seed & 1)).upload(0)
    order = sc.light_order()
    assert np.array_equal(order, orc.light_order()), seed
    info = sc.bvh_info()
    tris_seen.append(int(order.shape[0]))
    scenes_with_tables += int(order.shape[0] >= 16)
    for k in range(3):
        org = np.random.default_rng(seed * 3 + k).uniform(-4, 4, size=(20000, 3))
        g, c = sc.sample_lights(org, seed=seed + 11 * k), orc.sample_lights(org, seed=seed + 11 * k)
        # the PICK is bit-exact: same triangle, same face flag, same pdf (a host-computed constant of the triangle); position and
        # normal carry the GPU's fast sqrt of the barycentric draw (tests/test_gpu_parity.py: 1e-13 / 1e-15 absolute at unit scale)
        scale = np.maximum(1.0, np.abs(c["position"]).max(-1))
        same = (g["prim"] == c["prim"]) & (g["front"] == c["front"]) & (g["pdf"].view(np.uint64) == c["pdf"].view(np.uint64))
        same &= (np.abs(g["position"] - c["position"]).max(-1) <= 1e-13 * scale) & (np.abs(g["normal"] - c["normal"]).max(-1) <= 1e-15)
        tot += org.shape[0]
        if not same.all():
            bad += int((~same).sum())
            failed[seed] = {"differing": int((~same).sum()), "first": int(np.nonzero(~same)[0][0]), "light_tris": int(order.shape[0])}
            print("seed", seed, failed[seed], flush=True)
    img = sc.render(spp=8, max_depth=6, seed=seed)
    ref, _ = orc.render(spp=8, max_depth=6, seed=seed, nthreads=16)
    rel = np.abs(img - ref) / np.maximum(1.0, np.abs(ref))
    px_tot += rel.shape[0] * rel.shape[1]
    px_bad += int((rel > 1e-9).any(-1).sum())
    sc.close()
    if (seed - first) % 20 == 19:
        print(f"{seed - first + 1} scenes, {tot} picks, {bad} differing, {px_bad} / {px_tot} pixels beyond 1e-9, {time.time() - t0:.0f} s", flush=True)
out = {"scenes": n, "first_seed": first, "light_picks": tot, "light_picks_differing (triangle, face flag, pdf bits; position 1e-13, normal 1e-15)": bad, "scenes_with_at_least_16_light_triangles": scenes_with_tables,
       "light_triangles_min_median_max": [int(np.min(tris_seen)), int(np.median(tris_seen)), int(np.max(tris_seen))],
       "pixels": px_tot, "pixels_beyond_1e-9": px_bad, "seeds_with_differences": failed}
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_fuzz_lights.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "seeds_with_differences"}), flush=True)
