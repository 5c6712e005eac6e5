"""Developer tool: K4 (sorted ray batches) against plain K1 on the 8M-triangle soup; run under rocprofv3 --kernel-trace --stats
to see the phases (k_ray_keys, rocprim's sort kernels, k_trace_closest)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pooraytracer_amd import api, scenes
n_tris = int(os.environ.get("SORT_TRIS", "8000000"))
data = scenes.triangle_soup(n_tris=n_tris)
sc = api.Scene(data, device_bvh=True).upload(0)
lo, hi = data.bounds(); n = 1 << 24
rays = scenes.random_rays(n, lo, hi, seed=12345)
d_r = torch.from_numpy(rays.view(np.float64).reshape(-1, 8)).cuda(); d_h = torch.zeros((n, 4), dtype=torch.float64, device="cuda")
for sort in (False, True):
    best = 1e9
    for _ in range(3):
        sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr(), sort=sort); torch.cuda.synchronize(); best = min(best, sc.counters()["kernel_ms"])
    print(f"sort={sort}: {best:.2f} ms, {n / best / 1e3:.0f} Mrays/s", flush=True)
