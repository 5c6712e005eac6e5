"""Developer tool: wave-level statistics of K3's counting instantiation per workload (node rounds, leaf rounds, passes per
64 rays; VALU-relevant weights for tools/price_mix.py).  Writes gpurun_out/r04_wave_stats.json."""
import json, os, sys
sys.path.insert(0, os.getcwd())
import torch
from pooraytracer_amd import api, scenes
out = {}
for name, fn, kw, spp, depth in (("cornell-box", scenes.cornell_box, {}, 64, 20), ("veach-mis", scenes.veach_mis, {}, 96, 100),
                                 ("bathroom2", scenes.bathroom, {}, 32, 50), ("cornell-ct", scenes.cornell_box, {"ball_cooktorrance_alpha": 0.1}, 64, 10)):
    data = fn(**kw); sc = api.Scene(data).upload(0); cam = data.camera
    fb = torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda")
    for prec in (0, 1):
        sc.render_device(None, fb.data_ptr(), spp=spp, max_depth=depth, count_work=True, precision=prec); torch.cuda.synchronize()
        c = sc.counters(); r = c["rays_closest"] + c["rays_shadow"]
        out[name + ("-f32" if prec else "")] = dict(rays=r, samples=c["samples"], node_fetches=c["node_fetches"], tri_tests=c["tri_tests"], tri_full=c["tri_full"],
                                                   inner_rounds=c["inner_rounds"], leaf_rounds=c["leaf_rounds"], passes=c["refills"],
                                                   inner_rounds_per_64_rays=64 * c["inner_rounds"] / r, leaf_rounds_per_64_rays=64 * c["leaf_rounds"] / r,
                                                   passes_per_64_rays=64 * c["refills"] / r, node_round_lane_utilisation=c["node_fetches"] / max(1, 64 * c["inner_rounds"]))
    del sc
json.dump(out, open("gpurun_out/r04_wave_stats.json", "w"), indent=1)
print(json.dumps(out, indent=1))
