"""Developer experiment (cross-wave regrouping): what would PERFECT regrouping of the traversal buy?  K3's own
ray stream (every traversal it starts, in the order its waves start them: PRT_TUNE_DUMP_RAYS) is replayed through K1 — pure
traversal, every lane refilled the moment it finishes, nothing else in the wave's way — and the frame's traversal time at
that rate is set against the frame's K3 time.  Also replayed shuffled (incoherent) and sorted by origin (coherent)."""
import os as _os; _os.environ.setdefault("PRT_DEV_LIB", "1")  # the PRT_TUNE_* hooks exist in libprt_hip_dev.so only
import json, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from pooraytracer_amd import api, scenes, _abi
out = {}
for name, fn, kw, spp, depth in (("cornell-box", scenes.cornell_box, {}, 16, 20), ("veach-mis", scenes.veach_mis, {}, 32, 100), ("bathroom2", scenes.bathroom, {}, 8, 50)):
    data = fn(**kw); sc = api.Scene(data).upload(0); cam = data.camera
    fb = torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda")
    path = f"/tmp/rays_{name}.bin"
    os.environ["PRT_TUNE_DUMP_RAYS"] = f"120000000,{path}"
    sc.render_device(None, fb.data_ptr(), spp=spp, max_depth=depth, count_work=True); torch.cuda.synchronize()
    del os.environ["PRT_TUNE_DUMP_RAYS"]
    best = 1e9
    for _ in range(3):
        sc.render_device(None, fb.data_ptr(), spp=spp, max_depth=depth); torch.cuda.synchronize()
        c = sc.counters(); best = min(best, c["kernel_ms"])
    n_rays = c["rays_closest"] + c["rays_shadow"]
    rays = np.fromfile(path, dtype=_abi.RAY_DTYPE)
    os.remove(path)
    n = len(rays)
    res = {"k3_ms": best, "k3_rays": int(n_rays), "dumped": int(n), "k3_mrays_per_s": n_rays / best / 1e3, "shadow_fraction": float((rays["tmin"] > 5e-4).mean())}
    d_h = torch.zeros((n, 4), dtype=torch.float64, device="cuda")
    rng = np.random.default_rng(1)
    for tag, order in (("k3_order", None), ("shuffled", rng.permutation(n))):
        r = rays if order is None else rays[order]
        d_r = torch.from_numpy(r.view(np.float64).reshape(-1, 8)).cuda()
        b = 1e9
        for _ in range(3):
            sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr()); torch.cuda.synchronize(); b = min(b, sc.counters()["kernel_ms"])
        res[f"k1_{tag}_ms"] = b
        res[f"k1_{tag}_mrays_per_s"] = n / b / 1e3
        del d_r
    res["traversal_only_share_of_k3_time"] = res["k1_k3_order_ms"] / best * (n_rays / n)
    out[name] = res
    print(name, json.dumps(res), flush=True)
    del sc
json.dump(out, open("gpurun_out/r03_replay.json", "w"), indent=1)
