import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pooraytracer_amd import api, scenes
data = scenes.cornell_box(); sc = api.Scene(data).upload(0); cam = data.camera
fb = torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda")
sc.render_device(None, fb.data_ptr(), spp=8, max_depth=20); torch.cuda.synchronize()
for spp in (500,):
    for n in (1, 2, 4, 8):
        sc.render_device(None, fb.data_ptr(), spp=spp, max_depth=20, rank=0, nranks=n); torch.cuda.synchronize()
        c = sc.counters()
        print("spp", spp, "nranks", n, "ms", round(c["kernel_ms"], 2), "Mrays/s", round((c["rays_closest"] + c["rays_shadow"]) / c["kernel_ms"] / 1e3, 1))
