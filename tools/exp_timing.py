"""Developer tool (needs a -DPRT_K3_TIMING=1 build, PRT_LIB=...): where a K3 launch spends its fixed cost."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from pooraytracer_amd import api, scenes
data = scenes.cornell_box(); sc = api.Scene(data).upload(0); cam = data.camera
fb = torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda")
sc.render_device(None, fb.data_ptr(), spp=8, max_depth=20); torch.cuda.synchronize()
M = (1 << 64) - 1
for label, kw in (("1/8 share spp500", dict(spp=500, rank=0, nranks=8, tile_size=16)),
                  ("1/8 share spp500 depth5", dict(spp=500, rank=0, nranks=8, tile_size=16, max_depth=5)),
                  ("1/64 share spp500", dict(spp=500, rank=0, nranks=64, tile_size=16)),
                  ("full spp62", dict(spp=62)), ("full spp500", dict(spp=500))):
    kw.setdefault("max_depth", 20)
    for count in (True, False):
        sc.render_device(None, fb.data_ptr(), count_work=count, **kw); torch.cuda.synchronize()
        c = sc.counters()
        if count:
            t0, tdry, tend, life = M - c["inner_rounds"], M - c["leaf_rounds"], c["refills"], c["tri_tests"]
            span = (tend - t0) / 1e5
            print(f"{label}: COUNT kernel {c['kernel_ms']:.2f} ms | span {span:.2f} ms, queue dry at {(tdry - t0) / 1e5:.2f} ms, "
                  f"drain {(tend - tdry) / 1e5:.2f} ms, sum of wave lifetimes {life / 1e5:.0f} wave-ms "
                  f"(= {life / 1e5 / span:.0f} waves alive on average)", flush=True)
        else:
            r = c["rays_closest"] + c["rays_shadow"]
            print(f"   plain kernel {c['kernel_ms']:.2f} ms, {r/1e6:.0f} Mrays", flush=True)
