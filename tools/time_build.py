"""Developer tool: time prt_scene_create / prt_scene_upload (host SAH build + flatten + upload) per scene."""
import os, sys, time
sys.path.insert(0, os.getcwd())
import torch  # noqa: F401  (HIP runtime first)
from pooraytracer_amd import api, scenes

for name, fn in (("cornell", scenes.cornell_box), ("bathroom", scenes.bathroom),
                 ("soup1m", lambda: scenes.triangle_soup(1_000_000)), ("soup8m", lambda: scenes.triangle_soup(8_000_002))):
    if len(sys.argv) > 1 and name not in sys.argv[1:]:
        continue
    t0 = time.time(); data = fn(); t1 = time.time()
    sc = api.Scene(data); t2 = time.time()
    sc.upload(0); t3 = time.time()
    c = sc.counters()
    print(f"{name}: tris {len(data.vertices)} gen {t1-t0:.2f}s | HOST build: create {t2-t1:.3f}s upload {t3-t2:.3f}s nodes {c['bvh_nodes']} depth {c['bvh_depth']} bvh_ms {sc.bvh_info()['build_ms']:.1f}", flush=True)
    lo, hi = data.vertices.reshape(-1, 3).min(0), data.vertices.reshape(-1, 3).max(0)
    rays = scenes.random_rays(1 << 22, lo, hi, seed=3)
    def cost(scene, tag):
        scene.trace_closest(rays[:1000])
        scene.trace_closest(rays, count_work=True)
        k = scene.counters()
        drays = torch.from_numpy(rays.view("u1").reshape(-1)).cuda()
        dhits = torch.empty(len(rays) * 32, dtype=torch.uint8, device="cuda")
        for _ in range(3):
            scene.trace_closest_device(drays.data_ptr(), len(rays), dhits.data_ptr())
            torch.cuda.synchronize()
        ms = scene.counters()["kernel_ms"]
        print(f"   {tag}: nodes/ray {k['node_fetches']/len(rays):.1f} tris/ray {k['tri_tests']/len(rays):.2f}  K1 {len(rays)/ms/1e3:.0f} Mrays/s", flush=True)
    cost(sc, "host tree")
    del sc
    t4 = time.time(); sd = api.Scene(data, device_bvh=True); t5 = time.time()
    sd.upload(0); t6 = time.time()
    i = sd.bvh_info()
    print(f"   DEVICE build: create {t5-t4:.3f}s upload {t6-t5:.3f}s nodes {i['n_nodes']} depth {i['depth']} build_ms {i['build_ms']:.2f} (sort {i['sort_ms']:.2f} tree {i['tree_ms']:.2f} split {i['split_ms']:.2f})", flush=True)
    cost(sd, "device tree")
    del sd
