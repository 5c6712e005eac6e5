"""Developer tool: quick A/B of library builds (PRT_LIB=...) on the render workloads at reduced spp and on the K1
ray microbenchmarks.  AB_S4=1 adds the 8M-triangle soup (device-built tree), AB_COUNT=1 the per-ray work counters."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from pooraytracer_amd import api, scenes
tag = os.environ.get("PRT_LIB", "default").split("libprt_")[-1]
out = [tag]
count = os.environ.get("AB_COUNT") == "1"
which = os.environ.get("AB_SCENES", "cornell,bathroom,veach").split(",")
for name, fn, spp, depth in (("cornell", scenes.cornell_box, 250, 20), ("bathroom", scenes.bathroom, 50, 50), ("veach", scenes.veach_mis, 400, 100)):
    if name not in which:
        continue
    data = fn(); sc = api.Scene(data).upload(0); cam = data.camera
    fb = torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda")
    sc.render_device(None, fb.data_ptr(), spp=4, max_depth=depth); torch.cuda.synchronize()
    best = 1e9
    for _ in range(2):
        sc.render_device(None, fb.data_ptr(), max_depth=depth, spp=spp); torch.cuda.synchronize()
        c = sc.counters(); best = min(best, c["kernel_ms"])
    r = c["rays_closest"] + c["rays_shadow"]
    txt = f"{name} {best:.2f}ms {r/best/1e3:.0f}"
    if count:
        sc.render_device(None, fb.data_ptr(), max_depth=depth, spp=8, count_work=True); torch.cuda.synchronize()
        c = sc.counters(); r = c["rays_closest"] + c["rays_shadow"]
        txt += f" (n/r {c['node_fetches']/r:.2f} t/r {c['tri_tests']/r:.2f} full/r {c['tri_full']/r:.2f} util {c['node_fetches']/max(1,64*c['inner_rounds']):.2f})"
    # K1 on the same geometry
    lo, hi = data.bounds(); n = 1 << 22
    rays = scenes.random_rays(n, lo, hi, seed=12345)
    d_r = torch.from_numpy(rays.view(np.float64).reshape(-1, 8)).cuda(); d_h = torch.zeros((n, 4), dtype=torch.float64, device="cuda")
    best = 1e9
    for _ in range(3):
        sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr()); torch.cuda.synchronize(); best = min(best, sc.counters()["kernel_ms"])
    txt += f" k1 {n/best/1e3:.0f}"
    out.append(txt)
    del sc
if os.environ.get("AB_S4") == "1":
    data = scenes.triangle_soup(n_tris=8_000_000)
    for dev in (True, False) if os.environ.get("AB_S4_HOST") == "1" else (True,):
        t0 = time.time(); sc = api.Scene(data, device_bvh=dev).upload(0); t1 = time.time() - t0
        lo, hi = data.bounds(); n = 1 << 24
        rays = scenes.random_rays(n, lo, hi, seed=12345)
        d_r = torch.from_numpy(rays.view(np.float64).reshape(-1, 8)).cuda(); d_h = torch.zeros((n, 4), dtype=torch.float64, device="cuda")
        best = 1e9
        for _ in range(3):
            sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr()); torch.cuda.synchronize(); best = min(best, sc.counters()["kernel_ms"])
        srt = ""
        if os.environ.get("AB_SORT") == "1":
            bs = 1e9
            for _ in range(3):
                sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr(), sort=True); torch.cuda.synchronize(); bs = min(bs, sc.counters()["kernel_ms"])
            srt = f" sorted(K4) {n/bs/1e3:.0f} [{bs:.2f} ms incl. sort]"
        sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr(), count_work=True); torch.cuda.synchronize(); c = sc.counters()
        bi = sc.bvh_info()
        out.append(f"s4[{'gpu' if dev else 'host'} build {t1:.1f}s nodes {bi['n_nodes']} depth {bi['depth']}] {n/best/1e3:.0f} (n/r {c['node_fetches']/n:.1f} t/r {c['tri_tests']/n:.2f} full/r {c['tri_full']/n:.2f}){srt}")
        del sc
print(" | ".join(out), flush=True)
