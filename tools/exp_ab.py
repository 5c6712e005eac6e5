"""Developer tool: quick A/B of library builds (PRT_LIB=...) on the three render workloads at reduced spp."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pooraytracer_amd import api, scenes
out = [os.environ.get("PRT_LIB", "default").split("libprt_")[-1]]
for name, fn, spp, depth in (("cornell", scenes.cornell_box, 250, 20), ("bathroom", scenes.bathroom, 50, 50), ("veach", scenes.veach_mis, 400, 100)):
    data = fn(); sc = api.Scene(data).upload(0); cam = data.camera
    fb = torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda")
    sc.render_device(None, fb.data_ptr(), spp=4, max_depth=depth); torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        sc.render_device(None, fb.data_ptr(), max_depth=depth, spp=spp); torch.cuda.synchronize()
        c = sc.counters(); best = min(best, c["kernel_ms"])
    r = c["rays_closest"] + c["rays_shadow"]
    out.append(f"{name} {r/best/1e3:.0f}")
    del sc
print(" | ".join(out), flush=True)
