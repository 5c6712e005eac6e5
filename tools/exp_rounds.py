"""Developer tool: wave-level round statistics of K3 (counting instantiation)."""
import os as _os; _os.environ.setdefault("PRT_DEV_LIB", "1")  # the PRT_TUNE_* hooks exist in libprt_hip_dev.so only
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pooraytracer_amd import api, scenes
for name, fn, spp, depth in (("cornell", scenes.cornell_box, 64, 20), ("bathroom", scenes.bathroom, 32, 50), ("veach", scenes.veach_mis, 64, 100)):
    data = fn(); sc = api.Scene(data).upload(0); cam = data.camera
    fb = torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda")
    for keep in (None, "16", "32", "40"):
        if keep: os.environ["PRT_TUNE_KEEP"] = keep
        else: os.environ.pop("PRT_TUNE_KEEP", None)
        sc.render_device(None, fb.data_ptr(), spp=spp, max_depth=depth, count_work=True); torch.cuda.synchronize()
        c = sc.counters(); r = c["rays_closest"] + c["rays_shadow"]
        print(f"{name} keep {keep}: per ray: node visits {c['node_fetches']/r:.2f} tri tests {c['tri_tests']/r:.2f} | wave rounds per 64 rays: inner {c['inner_rounds']*64/r:.1f} leaf {c['leaf_rounds']*64/r:.2f} refill {c['refills']*64/r:.2f} | lane util inner {c['node_fetches']/max(1,c['inner_rounds'])/64:.2f} | {c['kernel_ms']:.1f} ms", flush=True)
    os.environ.pop("PRT_TUNE_KEEP", None)
    del sc
