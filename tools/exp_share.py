"""Developer tool: K3 time of one rank's 1/N tile share under different item schedules (PRT_TUNE_* overrides)."""
import os as _os; _os.environ.setdefault("PRT_DEV_LIB", "1")  # the PRT_TUNE_* hooks exist in libprt_hip_dev.so only
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pooraytracer_amd import api, scenes
data = scenes.cornell_box(); sc = api.Scene(data).upload(0); cam = data.camera
fb = torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda")
sc.render_device(None, fb.data_ptr(), spp=8, max_depth=20); torch.cuda.synchronize()

def run(label, reps=3, **kw):
    best = 1e9
    for _ in range(reps):
        sc.render_device(None, fb.data_ptr(), max_depth=20, **kw); torch.cuda.synchronize()
        c = sc.counters(); best = min(best, c["kernel_ms"])
    r = c["rays_closest"] + c["rays_shadow"]
    print(f"{label}: {best:.2f} ms  {r / best / 1e3:.0f} Mrays/s  ideal {r / 11.1e6:.2f} ms  overhead {best - r / 11.1e6:.2f} ms", flush=True)

nr = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for env in ({}, {"PRT_TUNE_VAR": "1"}, {"PRT_TUNE_VAR": "2"}, {"PRT_TUNE_VAR": "4"}, {"PRT_TUNE_VAR": "6"},
            {"PRT_TUNE_BODY": "32"}, {"PRT_TUNE_BODY": "128"}, {"PRT_TUNE_VAR": "4", "PRT_TUNE_BODY": "64"}):
    for k in ("PRT_TUNE_VAR", "PRT_TUNE_BODY", "PRT_TUNE_KEEP"):
        os.environ.pop(k, None)
    os.environ.update(env)
    run(f"1/{nr} share {env}", spp=500, rank=0, nranks=nr, tile_size=16)
for k in ("PRT_TUNE_VAR", "PRT_TUNE_BODY", "PRT_TUNE_KEEP"):
    os.environ.pop(k, None)
for env in ({}, {"PRT_TUNE_VAR": "1.5"}, {"PRT_TUNE_VAR": "6"}):
    os.environ.update(env)
    run(f"full frame spp500 {env}", spp=500)
    run(f"full frame spp62 {env}", spp=62)
    run(f"1/2 share {env}", spp=500, rank=0, nranks=2, tile_size=16)
    os.environ.pop("PRT_TUNE_VAR", None)
