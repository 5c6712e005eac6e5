"""Developer micro-benchmark (not the judged bench.py): times K3 on S1 and K1 on random rays."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from pooraytracer_amd import api, scenes, _abi

def main():
    spp = int(os.environ.get("QB_SPP", "16"))
    scene_name = os.environ.get("QB_SCENE", "cornell")
    data = {"cornell": scenes.cornell_box, "veach": scenes.veach_mis, "bathroom": scenes.bathroom}[scene_name]()
    t0 = time.time(); sc = api.Scene(data); t1 = time.time(); sc.upload(0)
    cam = data.camera
    out = torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda")
    res = {"lib": os.path.basename(api.lib_path()), "scene": scene_name, "tris": data.n_tris, "build_s": round(t1 - t0, 3)}
    depth = int(os.environ.get("QB_DEPTH", "20"))
    for rep in range(2):
        torch.cuda.synchronize(); t = time.time()
        sc.render_device(None, out.data_ptr(), spp=spp, max_depth=depth)
        torch.cuda.synchronize(); dt = time.time() - t
        c = sc.counters()
    rays = c["rays_closest"] + c["rays_shadow"]
    res.update(render_s=round(dt, 4), kernel_ms=round(c["kernel_ms"], 2), mrays=round(rays / dt / 1e6, 1), rays_per_sample=round(rays / c["samples"], 2),
               mean=[round(float(x), 5) for x in out.mean(dim=(0, 1)).tolist()])
    if os.environ.get("QB_COUNT"):
        sc.render_device(None, out.data_ptr(), spp=spp, max_depth=depth, count_work=True); torch.cuda.synchronize()
        c = sc.counters(); r = c["rays_closest"] + c["rays_shadow"]
        res.update(nodes_per_ray=round(c["node_fetches"] / r, 2), tris_per_ray=round(c["tri_tests"] / r, 2),
                   inner_util=round(c["node_fetches"] / max(1, 64 * c["inner_rounds"]), 3), leaf_rounds_per_ray=round(64 * c["leaf_rounds"] / r, 2),
                   inner_rounds_x64_per_ray=round(64 * c["inner_rounds"] / r, 2), refills_x64_per_ray=round(64 * c["refills"] / r, 2))
    # K1 microbench: incoherent rays
    lo, hi = data.bounds()
    n = 1 << 22
    rays_np = scenes.random_rays(n, lo, hi)
    d_r = torch.from_numpy(rays_np.view(np.float64).reshape(-1, 8)).cuda()
    d_h = torch.zeros((n, 4), dtype=torch.float64, device="cuda")
    for rep in range(3):
        sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr()); torch.cuda.synchronize()
    res.update(k1_ms=round(sc.counters()["kernel_ms"], 3), k1_mrays=round(n / sc.counters()["kernel_ms"] / 1e3, 1))
    print(json.dumps(res))

main()
