"""Developer tool: node fetches of the far (scaled, translated) cornell against the near one, for one library (PRT_LIB)."""
import copy, os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import torch  # noqa: F401
from pooraytracer_amd import api, scenes
base = scenes.cornell_box(ball_subdiv=3, width=32, height=32)
near = api.Scene(base).upload(0)
r0 = scenes.random_rays(20000, *base.bounds(), seed=31)
near.trace_closest(r0, count_work=True)
n = near.counters()["node_fetches"]
print(os.environ.get("PRT_LIB", "default").split("libprt_")[-1], "near", n, near.bvh_info()["nodes"] if "nodes" in near.bvh_info() else "", flush=True)
for scale, off in ((0.5, (0, 0, 0)), (0.5, (1e3, 1e3, 1e3)), (0.5, (1e5, 1e5, 1e5)), (0.5, (1e6, 1e6, 1e6)), (5.0, (6e6, -6e6, 6e6))):
    data = copy.copy(base)
    data.vertices = base.vertices * scale + np.asarray(off)
    sc = api.Scene(data).upload(0)
    lo, hi = data.vertices.reshape(-1, 3).min(0), data.vertices.reshape(-1, 3).max(0)
    rays = scenes.random_rays(20000, lo, hi, seed=31)
    rays["tmin"] = 1e-4 * scale
    sc.trace_closest(rays, count_work=True)
    f = sc.counters()["node_fetches"]
    print("  scale", scale, "offset", off[0], "fetches", f, f"{f / n:.3f}", flush=True)
