"""Whole-frame parity at the BASELINE configurations: every pixel of the fp64 GPU frame against the oracle's frame of the
same (spp, depth, seed) — not only the rows bench.py samples.  Writes gpurun_out/r03_full_frame_parity.json.
The oracle runs on the box's 16 host threads (about a minute per frame)."""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch  # noqa: F401
from pooraytracer_amd import api, scenes
import oracle

out = {}
CONFIGS = [("cornell-box spp 500 depth 20", scenes.cornell_box, {}, 500, 20),
           ("veach-mis spp 300 depth 100", scenes.veach_mis, {}, 300, 100),
           ("bathroom2 spp 100 depth 50", scenes.bathroom, {}, 100, 50),
           ("cornell-ct spp 100 depth 10", scenes.cornell_box, {"ball_cooktorrance_alpha": 0.1}, 100, 10)]
if os.environ.get("FF_FULL") == "1":  # the remaining BASELINE configurations at their full sample counts (minutes of oracle time each)
    CONFIGS = [("cornell-box spp 10 depth 10 (config 1)", scenes.cornell_box, {}, 10, 10),
               ("veach-mis spp 3000 depth 100 (config 3)", scenes.veach_mis, {}, 3000, 100),
               ("bathroom2 spp 500 depth 50 (config 5's frame)", scenes.bathroom, {}, 500, 50)]
for name, fn, kw, spp, depth in CONFIGS:
    data = fn(**kw)
    cam = data.camera
    sc = api.Scene(data).upload(0)
    orc = oracle.Oracle(data)
    t = time.time()
    ref, cnt = orc.render(spp=spp, max_depth=depth, seed=1, nthreads=16)
    dt = time.time() - t
    img = sc.render(spp=spp, max_depth=depth, seed=1)
    rel = np.abs(img - ref) / np.maximum(1.0, np.abs(ref))
    bad = (rel > 1e-9).any(-1)
    out[name] = {"pixels": int(bad.size), "pixels_beyond_1e-9": int(bad.sum()), "max_rel": float(rel.max()),
                 "image_mean_gpu": img.mean(axis=(0, 1)).tolist(), "image_mean_oracle": ref.mean(axis=(0, 1)).tolist(),
                 "oracle_seconds_16_threads": round(dt, 1), "oracle_mpaths_per_s": round(cnt["samples"] / dt / 1e6, 2),
                 "width": cam.width, "height": cam.height}
    print(name, out[name], flush=True)
    del sc
json.dump(out, open("gpurun_out/r03_full_frame_parity%s.json" % ("_full_spp" if os.environ.get("FF_FULL") == "1" else ""), "w"), indent=1)
