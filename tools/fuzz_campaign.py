"""Parity campaign on random scenes (the generator of tests/test_gpu_parity.py::_random_scene): N seeds, frame against the
oracle at 1e-9 per channel, every pixel; lists the seeds with pixels beyond it.  Writes gpurun_out/r04_fuzz_campaign.json."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from pooraytracer_amd import api, scenes
import oracle
from test_gpu_parity import _random_scene

n = int(os.environ.get("FUZZ_N", "200"))
first = int(os.environ.get("FUZZ_FIRST", "1000"))
spp, depth = int(os.environ.get("FUZZ_SPP", "16")), 12
tot_px = bad_px = 0
ray_tot = ray_bad = pick_tot = pick_bad = 0
smp_tot = smp_bad = sig_bad = 0
bad = {}
t0 = time.time()
for seed in range(first, first + n):
    data = _random_scene(seed)
    orc = oracle.Oracle(data)
    cpu, _ = orc.render(spp=spp, max_depth=depth, seed=seed + 1, background=(0.1, 0.2, 0.3), nthreads=16)
    sc = api.Scene(data, device_bvh=bool(seed & 1)).upload(0)
    img = sc.render(spp=spp, max_depth=depth, seed=seed + 1, background=(0.1, 0.2, 0.3))
    rel = np.abs(img - cpu) / np.maximum(1.0, np.abs(cpu))
    b = (rel > 1e-9).any(-1)
    tot_px += b.size
    if b.any():
        bad_px += int(b.sum())
        ys, xs = np.nonzero(b)
        bad[seed] = {"pixels": [[int(x), int(y)] for x, y in zip(xs[:8], ys[:8])], "count": int(b.sum()), "max_rel": float(rel.max())}
        print("seed", seed, bad[seed], flush=True)
    if os.environ.get("FUZZ_RAYS", "1") == "1":  # closest hits of random rays and light picks, as the per-seed test does
        lo, hi = data.bounds()
        rays = scenes.random_rays(20000, lo - 0.3, hi + 0.3, seed=seed + 100)
        want, got = orc.trace_closest(rays), sc.trace_closest(rays)
        same = (got["prim"] == want["prim"]) | (got["t"] == want["t"])  # exact ties may name the other triangle
        hit = want["prim"] >= 0
        terr = np.abs(got["t"][hit] - want["t"][hit]) / np.maximum(1.0, want["t"][hit])
        ray_tot += rays.shape[0]
        ray_bad += int((~same).sum()) + int((terr > 1e-12).sum())
        org = np.random.default_rng(seed).uniform(-1.5, 1.5, size=(2000, 3))
        g, c = sc.sample_lights(org, seed=seed), orc.sample_lights(org, seed=seed)
        pick_tot += org.shape[0]
        pick_bad += int((g["prim"] != c["prim"]).sum())
    if os.environ.get("FUZZ_SAMPLES", "1") == "1":  # per camera sample: radiance and path signature (prt_render_samples hook)
        cam = data.camera
        r2 = np.random.default_rng(seed + 7)
        px = np.stack([r2.integers(0, cam.width, 48), r2.integers(0, cam.height, 48)], axis=1)
        g, gt = sc.render_samples(px, spp=24, max_depth=depth, seed=seed + 2, trace=True, background=(0.1, 0.2, 0.3))
        o, ot = orc.render_samples(px, spp=24, max_depth=depth, seed=seed + 2, trace=True, background=(0.1, 0.2, 0.3))
        relS = (np.abs(g - o) / np.maximum(1.0, np.abs(o))).max(-1)
        smp_tot += relS.size
        smp_bad += int((relS > 1e-9).sum())
        sig_bad += int((~(gt == ot).all(-1)).sum())
    sc.close()
    if (seed - first) % 25 == 24:
        print(f"{seed - first + 1} scenes, {tot_px} pixels, {bad_px} beyond 1e-9, {time.time() - t0:.0f} s", flush=True)
out = {"scenes": n, "first_seed": first, "spp": spp, "max_depth": depth, "pixels": tot_px, "pixels_beyond_1e-9": bad_px,
       "random_rays": ray_tot, "rays_with_another_primitive_or_t_beyond_1e-12": ray_bad, "light_picks": pick_tot, "light_picks_differing": pick_bad,
       "camera_samples_compared_one_by_one": smp_tot, "samples_beyond_1e-9": smp_bad, "samples_with_another_path_signature": sig_bad,
       "seeds_with_differences": bad}
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r04_fuzz_campaign.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "seeds_with_differences"}), flush=True)
