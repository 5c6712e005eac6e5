"""Developer tool: K3 scheduling thresholds (keep / leaf_batch / inner_min) swept for the fp32 kernels (PRT_PREC=1) or the fp64 ones."""
import os as _os; _os.environ.setdefault("PRT_DEV_LIB", "1")  # the PRT_TUNE_* hooks exist in libprt_hip_dev.so only
import os, sys, itertools
sys.path.insert(0, os.getcwd())
import torch
from pooraytracer_amd import api, scenes
prec = int(os.environ.get("PRT_PREC", "1"))
which = os.environ.get("AB_SCENES", "cornell,bathroom,veach").split(",")
grid = [(k, lb, im) for k in (16, 24, 32, 40) for lb, im in ((40, 12), (48, 20), (56, 28))]
for name, fn, spp, depth in (("cornell", scenes.cornell_box, 125, 20), ("bathroom", scenes.bathroom, 25, 50), ("veach", scenes.veach_mis, 200, 100)):
    if name not in which:
        continue
    data = fn(); sc = api.Scene(data).upload(0); cam = data.camera
    fb = torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda")
    def run():
        best = 1e9
        for _ in range(2):
            sc.render_device(None, fb.data_ptr(), max_depth=depth, spp=spp, precision=prec); torch.cuda.synchronize()
            c = sc.counters(); best = min(best, c["kernel_ms"])
        return (c["rays_closest"] + c["rays_shadow"]) / best / 1e3
    for k in ("PRT_TUNE_KEEP", "PRT_TUNE_LEAF_BATCH", "PRT_TUNE_INNER_MIN"):
        os.environ.pop(k, None)
    run()
    res = [("default", run())]
    for k, lb, im in grid:
        os.environ.update(PRT_TUNE_KEEP=str(k), PRT_TUNE_LEAF_BATCH=str(lb), PRT_TUNE_INNER_MIN=str(im))
        res.append((f"{k}/{lb}/{im}", run()))
    print(name, f"prec={prec}", " ".join(f"{t}:{v:.0f}" for t, v in res), flush=True)
    del sc
