import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pooraytracer_amd import api, scenes
data = scenes.bathroom(); sc = api.Scene(data).upload(0); cam = data.camera
fb = torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda")
for prec in (0, 1):
    sc.render_device(None, fb.data_ptr(), spp=4, max_depth=50, precision=prec); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        sc.render_device(None, fb.data_ptr(), spp=50, max_depth=50, precision=prec); torch.cuda.synchronize()
        best = min(best, sc.counters()["kernel_ms"])
    print("precision", prec, best, "ms", flush=True)
