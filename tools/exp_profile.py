"""Developer tool (needs a -DPRT_K3_PROFILE=1 build, PRT_LIB=...): share of the wave cycles each section of K3's loop takes
(shader-clock stamps of lane 0 of every wave, summed): traversal rounds / consume (closest hit or shadow ray) / roulette +
Scatter / end of sample + fetch + new sample / traversal set-up.
Reading it: the stamps between "rounds" and "set-up" sit inside the divergent part of a pass, so a wave whose lane 0 is still
traversing books the WHOLE pass of its other lanes under "set-up" (r04z: cornell rounds 56.6 % | consume 11.8 | roulette +
Scatter 7.9 | end / fetch / new sample 5.0 | set-up 18.8 — i.e. rounds 57 %, pass 43 %, and the first three pass figures are
the split of the passes lane 0 takes part in; the set-up itself is 61 vector instructions)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pooraytracer_amd import api, scenes
for name, fn, spp, depth in (("cornell", scenes.cornell_box, 64, 20), ("bathroom", scenes.bathroom, 32, 50), ("veach", scenes.veach_mis, 64, 100)):
    data = fn(); sc = api.Scene(data).upload(0); cam = data.camera
    fb = torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda")
    sc.render_device(None, fb.data_ptr(), spp=4, max_depth=depth); torch.cuda.synchronize()
    sc.render_device(None, fb.data_ptr(), spp=spp, max_depth=depth, count_work=True); torch.cuda.synchronize()
    c = sc.counters()
    sec = [c["tri_tests"], c["inner_rounds"], c["leaf_rounds"], c["refills"], c["tri_full"]]
    tot = float(sum(sec))
    print(name, " | ".join(f"{n} {100 * v / tot:.1f}%" for n, v in zip(("rounds", "consume", "roulette+scatter", "end/fetch/new-sample", "set-up"), sec)),
          f"| total {tot / 1e9:.2f} Gcycles", flush=True)
    del sc
