"""Developer tool: find the pixels of a frame where the fp64 GPU image differs from the oracle's beyond 1e-9, then the
(sample) that differs and the two path signatures (prt_render_samples / orc_render_samples_trace)."""
import json, os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
import torch  # noqa: F401
from pooraytracer_amd import api, scenes
import oracle

name = os.environ.get("FL_SCENE", "cornell")
fn, kw, spp, depth = {"cornell": (scenes.cornell_box, {}, 500, 20), "cornell-ct": (scenes.cornell_box, {"ball_cooktorrance_alpha": 0.1}, 100, 10),
                      "veach": (scenes.veach_mis, {}, 300, 100), "bathroom": (scenes.bathroom, {}, 100, 50), "bathroom500": (scenes.bathroom, {}, 500, 50)}[name]
rows = int(os.environ.get("FL_ROWS", "112"))
data = fn(**kw)
cam = data.camera
sc = api.Scene(data).upload(0)
orc = oracle.Oracle(data)
y0 = cam.height // 2 - rows // 2
t = time.time()
ref, _ = orc.render(spp=spp, max_depth=depth, seed=1, rows=(y0, y0 + rows), nthreads=16)
print(f"oracle rows {y0}..{y0+rows}: {time.time()-t:.1f} s", flush=True)
img = sc.render(spp=spp, max_depth=depth, seed=1)
g, r = img[y0:y0 + rows], ref[y0:y0 + rows]
rel = np.abs(g - r) / np.maximum(1.0, np.abs(r))
bad = np.argwhere((rel > 1e-9).any(-1))
print("flipped pixels:", len(bad), "of", rows * cam.width, "max_rel", rel.max(), flush=True)
out = []
for (jj, i) in bad[:40]:
    j = int(jj) + y0
    px = [[int(i), j]]
    gs, gt = sc.render_samples(px, spp=spp, max_depth=depth, seed=1, trace=True)
    gp = sc.render_samples(px, spp=spp, max_depth=depth, seed=1)            # production instantiation
    os_, ot = orc.render_samples(px, spp=spp, max_depth=depth, seed=1, trace=True)
    srel = (np.abs(gs[0] - os_[0]) / np.maximum(1.0, np.abs(os_[0]))).max(-1)
    ds = np.argwhere(srel > 1e-9).ravel()
    print(f"pixel ({i},{j}): {len(ds)} differing samples {ds.tolist()[:8]}; count-build == production: {np.array_equal(gs, gp)}", flush=True)
    for s in ds[:4]:
        a, b = gt[0, s], ot[0, s]
        n = max(a[0], b[0])
        first = next((v for v in range(n) if a[1 + 2 * v] != b[1 + 2 * v] or a[2 + 2 * v] != b[2 + 2 * v]), None)
        print(f"   sample {s}: gpu {gs[0, s]} oracle {os_[0, s]}")
        print(f"      gpu    n={a[0]} {[(int(a[1+2*v]), int(a[2+2*v])) for v in range(a[0])]}")
        print(f"      oracle n={b[0]} {[(int(b[1+2*v]), int(b[2+2*v])) for v in range(b[0])]}")
        print(f"      first differing vertex: {first}", flush=True)
        st = oracle.rng_stream(1, j * cam.width + int(i), int(s), 256)
        print(f"      draws of exactly 0 in the sample's first 256 numbers: {np.nonzero(st == 0.0)[0].tolist()}", flush=True)
        out.append(dict(scene=name, pixel=[int(i), j], sample=int(s), gpu=gs[0, s].tolist(), oracle=os_[0, s].tolist(),
                        gpu_trace=a[:1 + 2 * a[0]].tolist(), oracle_trace=b[:1 + 2 * b[0]].tolist(), first_diff=first))
json.dump(out, open(f"gpurun_out/flips_{name}.json", "w"))
