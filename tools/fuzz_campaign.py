"""Parity campaign on random scenes (the generator of tests/test_gpu_parity.py::_random_scene): N seeds, frame against the
oracle at 1e-9 per channel, every pixel; lists the seeds with pixels beyond it.  Writes gpurun_out/r03_fuzz_campaign.json."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa: F401
from pooraytracer_amd import api
import oracle
from test_gpu_parity import _random_scene

n = int(os.environ.get("FUZZ_N", "200"))
first = int(os.environ.get("FUZZ_FIRST", "1000"))
spp, depth = int(os.environ.get("FUZZ_SPP", "16")), 12
tot_px = bad_px = 0
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
    sc.close()
    if (seed - first) % 25 == 24:
        print(f"{seed - first + 1} scenes, {tot_px} pixels, {bad_px} beyond 1e-9, {time.time() - t0:.0f} s", flush=True)
out = {"scenes": n, "first_seed": first, "spp": spp, "max_depth": depth, "pixels": tot_px, "pixels_beyond_1e-9": bad_px, "seeds_with_differences": bad}
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r03_fuzz_campaign.json"), "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "seeds_with_differences"}), flush=True)
