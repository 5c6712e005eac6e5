"""Developer tool: the counting build's wave-level counters of one library (PRT_LIB) on the three render scenes."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pooraytracer_amd import api, scenes
tag = os.environ.get("PRT_LIB", "default").split("libprt_")[-1]
for name, fn, depth in (("cornell", scenes.cornell_box, 20), ("bathroom", scenes.bathroom, 50), ("veach", scenes.veach_mis, 100)):
    data = fn(); sc = api.Scene(data).upload(0); cam = data.camera
    fb = torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda")
    sc.render_device(None, fb.data_ptr(), max_depth=depth, spp=16, count_work=True); torch.cuda.synchronize()
    c = sc.counters(); r = c["rays_closest"] + c["rays_shadow"]
    print(tag, name, {k: c[k] for k in ("node_fetches", "tri_tests", "tri_full", "inner_rounds", "leaf_rounds", "refills")}, "rays", r, flush=True)
    del sc
