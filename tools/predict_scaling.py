"""Developer tool: predicts the strong scaling of the tile partition from ONE GPU.  For N in (2, 4, 8) every rank's
tile share of the frame is rendered on this GPU the way bench.py renders it on its own GPU — `frames` frames back to
back, alternating two HIP streams / framebuffers (PS_PIPELINE=0: one stream) — and the slowest rank's frame period
is compared with the full frame's.  The RCCL reduce (12.6 MB, overlapped with the next frame) is not included."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pooraytracer_amd import api, scenes

def main():
    name = os.environ.get("PS_SCENE", "cornell")
    spp = int(os.environ.get("PS_SPP", "500")); depth = int(os.environ.get("PS_DEPTH", "20"))
    frames = int(os.environ.get("PS_FRAMES", "6")); pipe = os.environ.get("PS_PIPELINE", "1") == "1"
    tile = int(os.environ.get("PS_TILE", "16"))
    data = {"cornell": scenes.cornell_box, "veach": scenes.veach_mis, "bathroom": scenes.bathroom}[name]()
    sc = api.Scene(data).upload(0)
    cam = data.camera
    fbs = [torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()] if pipe else [torch.cuda.current_stream()] * 2

    def period(**kw):
        for k in range(2):
            sc.render_device(None, fbs[k].data_ptr(), spp=spp, max_depth=depth, stream=streams[k].cuda_stream, **kw)
        torch.cuda.synchronize()
        t = time.perf_counter()
        for k in range(frames):
            sc.render_device(None, fbs[k % 2].data_ptr(), spp=spp, max_depth=depth, stream=streams[k % 2].cuda_stream, **kw)
        torch.cuda.synchronize()
        return (time.perf_counter() - t) / frames * 1e3

    full = period(tile_size=32)
    out = {"scene": name, "spp": spp, "depth": depth, "pipelined": pipe, "frames": frames, "full_ms": round(full, 2)}
    for n in (2, 4, 8):
        ms = [period(rank=r, nranks=n, tile_size=tile) for r in range(n)]
        out[f"n{n}"] = {"max_ms": round(max(ms), 2), "mean_ms": round(sum(ms) / n, 2),
                        "balance": round(sum(ms) / n / max(ms), 4), "speedup_vs_full": round(full / max(ms), 3)}
    print(json.dumps(out), flush=True)
main()
