"""Whole-frame parity at the BASELINE configurations: every pixel of the fp64 GPU frame against the oracle's frame of the
same (spp, depth, seed) — not only the rows bench.py samples.  Writes gpurun_out/r04_full_frame_parity.json.
The oracle runs on the box's 16 host threads (about a minute per frame)."""
import os as _os; _os.environ.setdefault("PRT_DEV_LIB", "1")  # the PRT_TUNE_* hooks exist in libprt_hip_dev.so only
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # noqa: F401
from pooraytracer_amd import api, scenes
import oracle

out = {}
CONFIGS = [("cornell-box spp 500 depth 20", scenes.cornell_box, {}, 500, 20),
           ("veach-mis spp 300 depth 100", scenes.veach_mis, {}, 300, 100),
           ("bathroom2 spp 100 depth 50", scenes.bathroom, {}, 100, 50),
           ("cornell-ct spp 100 depth 10", scenes.cornell_box, {"ball_cooktorrance_alpha": 0.1}, 100, 10)]
if os.environ.get("FF_FULL") == "1":  # the remaining BASELINE configurations at their full sample counts (minutes of oracle time each)
    CONFIGS = [("cornell-box spp 10 depth 10 (config 1)", scenes.cornell_box, {}, 10, 10),
               ("veach-mis spp 3000 depth 100 (config 3)", scenes.veach_mis, {}, 3000, 100),
               ("bathroom2 spp 500 depth 50 (config 5's frame)", scenes.bathroom, {}, 500, 50)]
if os.environ.get("FF_VARIANTS") == "1":
    # The headline frame through every route that must not change it: host- and device-built tree, eight tile shares (the
    # 8-GPU decomposition on one GPU), three replicas through prt_render_multi (fp32 sum), pixel order scrambled.
    data = scenes.cornell_box()
    cam = data.camera
    spp, depth = 500, 20
    orc = oracle.Oracle(data)
    t = time.time()
    ref, cnt = orc.render(spp=spp, max_depth=depth, seed=1, nthreads=16)
    dt = time.time() - t

    def cmp(img):
        rel = np.abs(img - ref) / np.maximum(1.0, np.abs(ref))
        return {"pixels_beyond_1e-9": int((rel > 1e-9).any(-1).sum()), "max_rel": float(rel.max())}

    res = {"oracle_seconds_16_threads": round(dt, 1), "pixels": cam.width * cam.height}
    host = api.Scene(data).upload(0)
    base, base32 = host.render(spp=spp, max_depth=depth, seed=1, f32=True)
    res["host-built tree"] = cmp(base)
    dev = api.Scene(data, device_bvh=True).upload(0)
    img = dev.render(spp=spp, max_depth=depth, seed=1)
    res["device-built tree"] = dict(cmp(img), equals_host_tree_bitwise=bool(np.array_equal(img, base)))
    acc = np.zeros_like(base)
    for r in range(8):
        acc += host.render(spp=spp, max_depth=depth, seed=1, rank=r, nranks=8)
    res["eight tile shares summed"] = dict(cmp(acc), equals_single_launch_bitwise=bool(np.array_equal(acc, base)),
                                           max_rel_vs_single_launch=float((np.abs(acc - base) / np.maximum(1.0, np.abs(base))).max()))
    one8 = host.render(spp=spp, max_depth=depth, seed=1, sample_chunks=8)
    acc8 = np.zeros_like(base)
    for r in range(8):  # with an explicit chunk table (8 equal chunks) the shares group the samples like the single launch does
        acc8 += host.render(spp=spp, max_depth=depth, seed=1, rank=r, nranks=8, sample_chunks=8)
    res["eight tile shares summed, explicit sample_chunks=8"] = {"equals_single_launch_with_the_same_chunks_bitwise": bool(np.array_equal(acc8, one8))}
    res["eight tile shares summed"]["note"] = ("a share deals its samples in chunks sized for ITS pixel count; the per-pixel sum of chunk sums "
                                               "therefore groups the samples differently from the single launch: same terms, fp64 rounding of the order 1e-16")
    reps = [host, api.Scene(data).upload(0), api.Scene(data).upload(0)]
    m = api.render_multi(reps, spp=spp, max_depth=depth, seed=1)
    res["prt_render_multi, three replicas (fp32)"] = {"equals_single_launch_fp32_bitwise": bool(np.array_equal(m, base32))}
    os.environ["PRT_TUNE_SCRAMBLE"] = "1"
    img = host.render(spp=spp, max_depth=depth, seed=1)
    del os.environ["PRT_TUNE_SCRAMBLE"]
    res["pixel order scrambled"] = dict(cmp(img), equals_single_launch_bitwise=bool(np.array_equal(img, base)))
    print(json.dumps(res, indent=1), flush=True)
    json.dump({"cornell-box spp 500 depth 20, every route": res}, open("gpurun_out/r04_full_frame_variants.json", "w"), indent=1)
    raise SystemExit(0)
for name, fn, kw, spp, depth in CONFIGS:
    data = fn(**kw)
    cam = data.camera
    sc = api.Scene(data).upload(0)
    orc = oracle.Oracle(data)
    t = time.time()
    ref, cnt = orc.render(spp=spp, max_depth=depth, seed=1, nthreads=16)
    dt = time.time() - t
    img = sc.render(spp=spp, max_depth=depth, seed=1)
    rel = np.abs(img - ref) / np.maximum(1.0, np.abs(ref))
    bad = (rel > 1e-9).any(-1)
    out[name] = {"pixels": int(bad.size), "pixels_beyond_1e-9": int(bad.sum()), "max_rel": float(rel.max()),
                 "image_mean_gpu": img.mean(axis=(0, 1)).tolist(), "image_mean_oracle": ref.mean(axis=(0, 1)).tolist(),
                 "oracle_seconds_16_threads": round(dt, 1), "oracle_mpaths_per_s": round(cnt["samples"] / dt / 1e6, 2),
                 "width": cam.width, "height": cam.height}
    print(name, out[name], flush=True)
    del sc
json.dump(out, open("gpurun_out/r04_full_frame_parity%s.json" % ("_full_spp" if os.environ.get("FF_FULL") == "1" else ""), "w"), indent=1)
