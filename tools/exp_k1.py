"""Developer tool: K1 (closest hit, 2^24 incoherent rays) on the cornell geometry and on the 8M-triangle soup, both precisions."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from pooraytracer_amd import api, scenes
tag = os.environ.get("PRT_LIB", "default").split("libprt_")[-1]
out = [tag]
for name, data, dev in (("s0", scenes.cornell_box(), False), ("s4", scenes.triangle_soup(n_tris=int(os.environ.get("K1_SOUP", "8000000"))), True)):
    sc = api.Scene(data, device_bvh=dev).upload(0)
    lo, hi = data.bounds(); n = 1 << 24
    rays = scenes.random_rays(n, lo, hi, seed=12345)
    d_r = torch.from_numpy(rays.view(np.float64).reshape(-1, 8)).cuda(); d_h = torch.zeros((n, 4), dtype=torch.float64, device="cuda")
    for prec in (0, 1):
        best = 1e9
        for _ in range(4):
            sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr(), precision=prec); torch.cuda.synchronize(); best = min(best, sc.counters()["kernel_ms"])
        out.append(f"{name} {'f32' if prec else 'f64'} {n/best/1e3:.0f}")
    h = d_h.cpu().numpy(); out.append(f"chk {np.nansum(np.where(np.isfinite(h[:,0]), h[:,0], 0)):.6f}")
    del sc
print(" | ".join(out), flush=True)
