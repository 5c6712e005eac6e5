import os; os.environ["PRT_DEV_LIB"]="1"
import sys; sys.path.insert(0, os.getcwd())
import torch
from pooraytracer_amd import api, scenes
for name, fn, spp, depth in (("bathroom", scenes.bathroom, 100, 50), ("veach", scenes.veach_mis, 600, 100)):
    data = fn(); sc = api.Scene(data).upload(0); cam = data.camera
    fb = torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda")
    sc.render_device(None, fb.data_ptr(), spp=4, max_depth=depth); torch.cuda.synchronize()
    res = {}
    for rep in range(4):
        for lb, im in ((40, 12), (32, 12), (32, 8)):
            os.environ["PRT_TUNE_LEAF_BATCH"] = str(lb); os.environ["PRT_TUNE_INNER_MIN"] = str(im)
            sc.render_device(None, fb.data_ptr(), max_depth=depth, spp=spp); torch.cuda.synchronize()
            res.setdefault((lb, im), []).append(round(sc.counters()["kernel_ms"], 2))
    print(name, res, flush=True)
