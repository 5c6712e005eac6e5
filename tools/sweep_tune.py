"""Developer tool: sweep the wave-scheduling thresholds of K3 (PRT_TUNE_KEEP / LEAF_BATCH / INNER_MIN are read by
prt_render_device at every call) for one library build, on the three render scenes at reduced spp."""
import os as _os; _os.environ.setdefault("PRT_DEV_LIB", "1")  # the PRT_TUNE_* hooks exist in libprt_hip_dev.so only
import itertools, os, sys
sys.path.insert(0, os.getcwd())
import torch
from pooraytracer_amd import api, scenes
tag = os.environ.get("PRT_LIB", "default").split("libprt_")[-1]
keeps = [int(x) for x in os.environ.get("SW_KEEP", "16,20,24,28,32").split(",")]
lbs = [int(x) for x in os.environ.get("SW_LB", "32").split(",")]
ims = [int(x) for x in os.environ.get("SW_IM", "12").split(",")]
cms = [int(x) for x in os.environ.get("SW_CM", "65").split(",")]
rms = [int(x) for x in os.environ.get("SW_RM", "12").split(",")]
which = os.environ.get("SW_SCENES", "cornell,bathroom,veach").split(",")
for name, fn, spp, depth in (("cornell", scenes.cornell_box, 125, 20), ("bathroom", scenes.bathroom, 50, 50), ("veach", scenes.veach_mis, 200, 100)):
    if name not in which:
        continue
    data = fn(); sc = api.Scene(data).upload(0); cam = data.camera
    fb = torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda")
    sc.render_device(None, fb.data_ptr(), spp=4, max_depth=depth); torch.cuda.synchronize()
    res = []
    for k, lb, im, cm, rm in itertools.product(keeps, lbs, ims, cms, rms):
        os.environ["PRT_TUNE_KEEP"], os.environ["PRT_TUNE_LEAF_BATCH"], os.environ["PRT_TUNE_INNER_MIN"], os.environ["PRT_TUNE_CACHED_MIN"] = str(k), str(lb), str(im), str(cm)
        os.environ["PRT_TUNE_RESTART_MIN"] = str(rm)
        best = 1e9
        for _ in range(2):
            sc.render_device(None, fb.data_ptr(), max_depth=depth, spp=spp); torch.cuda.synchronize()
            c = sc.counters(); best = min(best, c["kernel_ms"])
        r = c["rays_closest"] + c["rays_shadow"]
        res.append((best, k, lb, im, cm, rm))
    res.sort()
    print(tag, name, " ".join(f"{m:.2f}ms@k{k}/lb{lb}/im{im}/cm{cm}/rm{rm}" for m, k, lb, im, cm, rm in res[:14]), "| worst", f"{res[-1][0]:.2f}", flush=True)
    del sc
