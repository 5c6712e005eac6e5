"""Developer tool: renders each rank's tile share of one frame on ONE GPU, one after another, and
predicts the strong-scaling efficiency of the tile partition as mean(kernel_ms) / max(kernel_ms)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pooraytracer_amd import api, scenes

def main():
    name = os.environ.get("PS_SCENE", "cornell")
    spp = int(os.environ.get("PS_SPP", "500")); depth = int(os.environ.get("PS_DEPTH", "20"))
    data = {"cornell": scenes.cornell_box, "veach": scenes.veach_mis, "bathroom": scenes.bathroom}[name]()
    sc = api.Scene(data).upload(0)
    cam = data.camera
    fb = torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda")
    sc.render_device(None, fb.data_ptr(), spp=8, max_depth=depth); torch.cuda.synchronize()
    sc.render_device(None, fb.data_ptr(), spp=spp, max_depth=depth); torch.cuda.synchronize()
    full = sc.counters()["kernel_ms"]
    out = {"scene": name, "spp": spp, "full_ms": round(full, 2)}
    for n in (2, 4, 8):
        ms = []
        for r in range(n):
            sc.render_device(None, fb.data_ptr(), spp=spp, max_depth=depth, rank=r, nranks=n, tile_size=int(os.environ.get("PS_TILE", "32"))); torch.cuda.synchronize()
            ms.append(sc.counters()["kernel_ms"])
        out[f"n{n}"] = {"max_ms": round(max(ms), 2), "mean_ms": round(sum(ms) / n, 2),
                        "balance": round(sum(ms) / n / max(ms), 4), "speedup_vs_full": round(full / max(ms), 3)}
    print(json.dumps(out))
main()
