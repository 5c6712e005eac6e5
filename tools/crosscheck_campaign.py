"""CPU-only: the oracle (oracle/pt_oracle.cpp) against the second, independent restatement in Python
(tests/test_oracle_crosscheck.py::Tracer) on RANDOM scenes — the generator of the GPU fuzz test: all material kinds with
random parameters, textures, one or two lights (a light list of one or two meshes), random cameras — per camera sample,
tolerance 1e-12.  Writes profiles/r03_crosscheck_campaign.json (run here, in the build container: no GPU involved)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
from test_gpu_parity import _random_scene
from test_oracle_crosscheck import Tracer

n = int(os.environ.get("XC_N", "60"))
first = int(os.environ.get("XC_FIRST", "9000"))
spp, depth = 2, 8
samples = bad = 0
worst = 0.0
seen = set()
t0 = time.time()
for seed in range(first, first + n):
    data = _random_scene(seed)
    cam = data.camera
    orc = oracle.Oracle(data)
    py = Tracer(data, rr=0.8, background=(0.1, 0.2, 0.3), sample_lights=True)
    rng = np.random.default_rng(seed)
    px = [(int(i), int(j)) for i, j in zip(rng.integers(0, cam.width, 40), rng.integers(0, cam.height, 40))]
    want = orc.render_samples(px, spp=spp, max_depth=depth, seed=seed + 3, rr=0.8, background=(0.1, 0.2, 0.3))
    for k, (i, j) in enumerate(px):
        for s in range(spp):
            got = py.sample(cam, i, j, s, seed + 3, depth)
            err = float(np.abs(got - want[k, s]).max() / max(1.0, np.abs(want[k, s]).max()))
            worst = max(worst, err)
            samples += 1
            if not err <= 1e-12:
                bad += 1
                print("DIFF seed", seed, "pixel", (i, j), "sample", s, got, want[k, s], flush=True)
    seen |= py.seen
    if (seed - first) % 10 == 9:
        print(f"{seed - first + 1} scenes, {samples} samples, {bad} beyond 1e-12, worst {worst:.2e}, {time.time() - t0:.0f} s", flush=True)
out = {"scenes": n, "first_seed": first, "camera_samples": samples, "samples_beyond_1e-12": bad, "worst_relative_difference": worst,
       "max_depth": depth, "materials_met": sorted(seen)}
json.dump(out, open(os.path.join(ROOT, "profiles", "r03_crosscheck_campaign.json"), "w"), indent=1)
print(json.dumps(out), flush=True)
