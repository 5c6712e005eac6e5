"""Developer tool: fp32 fast mode against the fp64 path — hit agreement on random rays, image agreement at equal seeds,
and frame times at reduced spp."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch
from pooraytracer_amd import api, scenes, _abi

which = os.environ.get("AB_SCENES", "cornell,bathroom,veach").split(",")
for name, fn, spp, depth in (("cornell", scenes.cornell_box, 250, 20), ("bathroom", scenes.bathroom, 50, 50), ("veach", scenes.veach_mis, 400, 100)):
    if name not in which:
        continue
    data = fn(); sc = api.Scene(data).upload(0); cam = data.camera
    lo, hi = data.bounds(); n = 1 << 20
    rays = scenes.random_rays(n, lo, hi, seed=12345)
    d_r = torch.from_numpy(rays.view(np.float64).reshape(-1, 8)).cuda()
    res = {}
    for prec in (0, 1):
        d_h = torch.zeros((n, 4), dtype=torch.float64, device="cuda")
        best = 1e9
        for _ in range(3):
            sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr(), precision=prec); torch.cuda.synchronize()
            best = min(best, sc.counters()["kernel_ms"])
        res[prec] = (d_h.cpu().numpy().view(_abi.HIT_DTYPE).reshape(-1), n / best / 1e3)
    a, b = res[0][0], res[1][0]
    hit = (a["prim"] >= 0) & (b["prim"] >= 0)
    dt = np.abs(a["t"][hit] - b["t"][hit]) / np.maximum(1.0, a["t"][hit])
    print(f"{name} K1: f64 {res[0][1]:.0f} f32 {res[1][1]:.0f} Mrays/s | hit/miss disagree {(a['prim'] >= 0).sum() - hit.sum() + (b['prim'] >= 0).sum() - hit.sum()} "
          f"prim differ {(a['prim'][hit] != b['prim'][hit]).sum()} of {hit.sum()} | dt/t max {dt.max():.2e} p99.9 {np.quantile(dt, 0.999):.2e} (>1e-5: {(dt > 1e-5).sum()})", flush=True)
    fb = {p: torch.zeros((cam.height, cam.width, 3), dtype=torch.float64, device="cuda") for p in (0, 1)}
    ms = {}
    for prec in (0, 1):
        sc.render_device(fb[prec].data_ptr(), None, spp=4, max_depth=depth, precision=prec); torch.cuda.synchronize()
        best = 1e9
        for _ in range(2):
            sc.render_device(fb[prec].data_ptr(), None, max_depth=depth, spp=spp, precision=prec); torch.cuda.synchronize()
            c = sc.counters(); best = min(best, c["kernel_ms"])
        ms[prec] = (best, (c["rays_closest"] + c["rays_shadow"]) / best / 1e3)
    x, y = fb[0].cpu().numpy(), fb[1].cpu().numpy()
    mean_rel = abs(y.mean() - x.mean()) / x.mean()
    d = np.abs(y - x)
    print(f"{name} K3 spp {spp}: f64 {ms[0][0]:.1f} ms {ms[0][1]:.0f} Mrays/s | f32 {ms[1][0]:.1f} ms {ms[1][1]:.0f} Mrays/s (x{ms[0][0] / ms[1][0]:.2f}) | "
          f"image mean {x.mean():.6f} vs {y.mean():.6f} rel {mean_rel:.2e} | px |d| mean {d.mean():.2e} max {d.max():.2e} nan {np.isnan(y).sum()}", flush=True)
    del sc
