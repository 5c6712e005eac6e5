import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pooraytracer_amd import api, scenes
data = scenes.cornell_box(); sc = api.Scene(data).upload(0); cam = data.camera
fb = torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda")
sc.render_device(None, fb.data_ptr(), spp=8, max_depth=20); torch.cuda.synchronize()
def run(label, **kw):
    sc.render_device(None, fb.data_ptr(), max_depth=20, **kw); torch.cuda.synchronize()
    c = sc.counters(); r = c["rays_closest"] + c["rays_shadow"]
    print(label, "ms", round(c["kernel_ms"], 2), "Mrays/s", round(r / c["kernel_ms"] / 1e3, 1), "rays", r)
for n in (1, 8, 16, 64):
    run(f"share 1/{n} spp500", spp=500, rank=0, nranks=n)
small = scenes.Camera(362, 362, cam.fovy, cam.eye, cam.look_at)
run("full 362x362 spp500", spp=500, camera=small)
small2 = scenes.Camera(512, 256, cam.fovy, cam.eye, cam.look_at)
run("full 512x256 spp500", spp=500, camera=small2)
run("full 1024x1024 spp62", spp=62)
run("full 1024x1024 spp125", spp=125)
run("full 1024x1024 spp250", spp=250)
