#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X path-tracing hot path.

Metric (BASELINE.json): Mrays/s (primary + secondary) and wall-clock s/frame at fixed spp.
Workload at N=1 (BASELINE.json configs[1]): synthetic "cornell-box" stand-in (scenes.cornell_box,
20,492 triangles — the reference's asset is not available), 1024x1024, spp=500, depth=20, RR 0.8,
bSampleLights, seed 1, fp64 arithmetic (the reference's).  One "step" = one full frame
(Camera::Render): K3 persistent path-tracing kernel + K5 finalize, inputs (scene, BVH) resident in HBM.

N>1: one process per GPU (torch.distributed, backend nccl = RCCL).  The SAME frame is cut into 16x16
tiles dealt diagonally over ranks (strong scaling: total work fixed); every rank renders its tiles
into a zero-initialised full-size fp32 framebuffer and one RCCL reduce(sum) to rank 0 assembles the
image (disjoint tiles => x + 0 + ... + 0, bit-identical to the 1-GPU image).  The reduce is inside
the timed region.  For N>1 consecutive frames alternate between two HIP streams / framebuffers so the
next frame fills the GPU while the previous one drains and is being reduced (--no-pipeline turns it
off; at N=1 frames run back to back on one stream unless --pipeline is given).

Prints ONE JSON line on rank 0.  Extra objects:
  roofline     dominant kernel k_render: algorithmic bytes per launch / mean launch duration measured
               live with HIP events on the launch stream around every launch of the timed region
               (overlapped launches, N>1 only: the launch period instead, see kernel_ms_source).
  cpu_baseline the CPU oracle (port of the reference algorithm, oracle/pt_oracle.cpp) timed on this
               box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E nominal (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate

WORKLOADS = {
    # name: (scene factory name, kwargs, spp, depth)
    "cornell-box": ("cornell_box", {}, 500, 20),
    "veach-mis": ("veach_mis", {}, 3000, 100),
    "bathroom2": ("bathroom", {}, 100, 50),
}
# K1 closest-hit microbenchmarks: 2^24 seeded incoherent rays resident in HBM
RAY_WORKLOADS = {
    "s0-rays-cornell": ("cornell_box", {}),                          # cache-resident geometry
    "s4-rays-soup8m": ("triangle_soup", {"n_tris": 8_000_000}),      # HBM-resident: 256 MB nodes + 1 GB triangles
}


def run_ray_microbench(args, torch, api, scenes):
    """One step = one K1 launch over 2^24 rays (rays and hits stay in HBM)."""
    import numpy as np
    fn, kw = RAY_WORKLOADS[args.workload]
    data = getattr(scenes, fn)(**kw)
    t0 = time.time()
    sc = api.Scene(data)
    build_s = time.time() - t0
    sc.upload(0)
    n = 1 << 24
    lo, hi = data.bounds()
    rays_np = scenes.random_rays(n, lo, hi, seed=12345)
    d_r = torch.from_numpy(rays_np.view(np.float64).reshape(-1, 8)).cuda()
    d_h = torch.zeros((n, 4), dtype=torch.float64, device="cuda")
    for _ in range(args.warmup):
        sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr())
    torch.cuda.synchronize()
    ms = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr())
        ms.append(sc.counters()["kernel_ms"])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr(), count_work=True)
    torch.cuda.synchronize()
    cc = sc.counters()
    npr, tpr = cc["node_fetches"] / n, cc["tri_tests"] / n
    bpr = bytes_per_ray(npr, tpr)
    mean_ms = sum(ms) / len(ms)
    achieved = bpr * n / (mean_ms * 1e-3) / 1e9
    hits = d_h.cpu().numpy().view(np.dtype([("t", "<f8"), ("a", "<f8"), ("b", "<f8"), ("prim", "<i4"), ("front", "<i4")])).reshape(-1)
    out = {
        "metric": "Mrays/s (closest-hit, incoherent rays)", "value": round(n * args.steps / elapsed / 1e6, 2), "unit": "Mrays/s",
        "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 3),
        "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {n} seeded random rays vs {data.n_tris} triangles "
                               f"({cc['bvh_nodes']} BVH nodes, depth {cc['bvh_depth']}, host build {build_s:.1f} s)",
                   "hit_fraction": round(float((hits["prim"] >= 0).mean()), 4)},
        "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                     "frac": round(achieved / HBM_PEAK_GBPS, 4), "traffic": None, "kernel": "k_trace_closest",
                     "kernel_ms": round(mean_ms, 3), "bytes_per_ray": round(bpr, 1), "nodes_per_ray": round(npr, 2),
                     "tris_per_ray": round(tpr, 2)},
    }
    if not args.no_cpu_baseline:
        import oracle
        orc = oracle.Oracle(data) if data.n_tris <= 200_000 else None
        if orc is not None:
            m = 200_000
            t0 = time.time()
            orc.trace_closest(rays_np[:m])
            dt = time.time() - t0
            out["cpu_baseline"] = {"value": round(m / dt / 1e6, 3), "unit": "Mrays/s", "cores": 1, "kind": "port",
                                   "sample": f"first {m} rays of the same batch, single thread ({dt:.2f} s)"}
    print(json.dumps(out), flush=True)



NODE_BYTES = 32.0   # DNode: both children's boxes on a 16-bit grid + 2 refs (csrc/prt_types.h)
TRI_BYTES = 128.0   # DTri: fp64 n, D, w, v0, e0, e1


def bytes_per_ray(nodes_per_ray, tris_per_ray):
    """SURVEY.md §8(d) with this build's record sizes: ray in (64 B fp64 o,tmin,d,tmax) + hit out (32 B)
    + 32 B per BVH node record fetched + 128 B per fp64 triangle record tested (counters from the
    counting instantiation of the same kernel)."""
    return 64.0 + 32.0 + NODE_BYTES * nodes_per_ray + TRI_BYTES * tris_per_ray


def cpu_baseline(scene_data, spp, depth, seed, target_s=8.0):
    """Time the CPU oracle (kind "port") on a bounded sample: centre rows of the same image at the
    same spp/depth, all host threads as independent row workers with keyed per-sample RNG."""
    import oracle  # test infrastructure used here only as the reported CPU baseline
    threads = max(1, min(len(os.sched_getaffinity(0)), 16))  # the GPU box gives one GPU a 16-core CPU share
    orc = oracle.Oracle(scene_data)
    cam = scene_data.camera
    mid = cam.height // 2
    # calibration: a few rows at low spp
    t = time.time()
    _, c = orc.render(spp=8, max_depth=depth, seed=seed, rows=(mid, mid + threads), nthreads=threads)
    dt = max(time.time() - t, 1e-3)
    paths_per_s = c["samples"] / dt
    rows = int(max(threads, min(cam.height, (paths_per_s * target_s) / (cam.width * spp))))
    rows = max(threads, (rows // threads) * threads)
    y0 = max(0, mid - rows // 2)
    t = time.time()
    _, c = orc.render(spp=spp, max_depth=depth, seed=seed, rows=(y0, y0 + rows), nthreads=threads)
    dt = time.time() - t
    rays = c["rays_closest"] + c["rays_shadow"]
    return {
        "value": round(rays / dt / 1e6, 3),
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "sample": f"rows {y0}..{y0 + rows} of the {cam.width}x{cam.height} frame at spp={spp} depth={depth} "
                  f"({c['samples']} camera samples, {rays} rays, {dt:.2f} s, {threads} row-worker threads)",
        "mpaths_per_s": round(c["samples"] / dt / 1e6, 4),
        "seconds": round(dt, 2),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell-box", choices=sorted(WORKLOADS) + sorted(RAY_WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override spp (0 = the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-pipeline", action="store_true", help="one stream, one framebuffer: frames strictly back to back")
    ap.add_argument("--pipeline", action="store_true", help="alternate two streams / framebuffers also at N=1 (default: only for N>1)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world != 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    nranks = world

    import torch
    import torch.distributed as dist
    from pooraytracer_amd import api, build, distributed, scenes

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    if rank == 0:
        build.build()  # no-op when libprt_hip.so is fresh
    # PRT_BENCH_REHEARSAL=1: exercise the N>1 path on a ONE-GPU box — every rank shares cuda:0 and the
    # reduce goes through gloo on host copies (RCCL refuses two ranks on one device).  Not a measurement.
    rehearsal = os.environ.get("PRT_BENCH_REHEARSAL") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    force_dist = os.environ.get("PRT_BENCH_FORCE_DIST") == "1"  # test hook: RCCL process group even at N=1
    if nranks > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if rehearsal:
            dist.init_process_group(backend="gloo", rank=rank, world_size=nranks)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=nranks,
                                    device_id=torch.device("cuda", local_rank))
        dist.barrier()

    if args.workload in RAY_WORKLOADS:
        if nranks != 1:
            raise SystemExit("ray microbenchmarks are single-GPU")
        run_ray_microbench(args, torch, api, scenes)
        return
    fn, kw, spp, depth = WORKLOADS[args.workload]
    if args.spp > 0:
        spp = args.spp
    seed = 1
    data = getattr(scenes, fn)(**kw)
    cam = data.camera
    sc = api.Scene(data).upload(local_rank)
    # Two framebuffers on two HIP streams: consecutive frames alternate, so frame k+1 starts filling the
    # GPU while the last long paths of frame k drain and its framebuffer is being reduced (RCCL overlaps
    # with compute).  Every step is still one complete frame; libprt_hip double-buffers its per-call state.
    # Pipelining is for N > 1 (a 1/N tile share makes the fixed fill+drain of a launch matter; at N = 1 it is
    # worth 2 %).  At N = 1 the frames run back to back on one stream so that the HIP events around each launch
    # bracket exactly that launch: overlapped launches would each be timed from submission, i.e. including the
    # wait for the previous frame's wave slots (measured: 834 ms per launch for a 418 ms launch period), and the
    # roofline below would be computed from a duration that is not the kernel's.  --pipeline forces it on.
    pipelined = (args.pipeline or nranks > 1) and not args.no_pipeline and not rehearsal
    fbs = [torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
    fb = fbs[0]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()] if pipelined else [torch.cuda.current_stream()] * 2
    stream = torch.cuda.current_stream().cuda_stream
    render_kw = dict(spp=spp, max_depth=depth, seed=seed, rank=rank, nranks=nranks, tile_size=16 if nranks > 1 else 32)

    def step(k, events=None):
        st, buf = streams[k % 2], fbs[k % 2]
        with torch.cuda.stream(st):
            if events is not None:
                events[0].record(st)
            sc.render_device(None, buf.data_ptr(), stream=st.cuda_stream, **render_kw)
            if events is not None:
                events[1].record(st)
            if rehearsal and nranks > 1:
                host = buf.cpu()
                distributed.reduce_framebuffer(host, dst=0)
                buf.copy_(host)
            else:
                distributed.reduce_framebuffer(buf, dst=0)

    def fence():
        if nranks > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k)
    fence()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k, evs[k])
    fence()
    elapsed = time.perf_counter() - t0
    # HIP events recorded on the stream each K3 was launched on (memsets + K3 + K5; K3 is > 99.9 % of it)
    kernel_ms = [a.elapsed_time(b) for a, b in evs]
    c = sc.counters()  # ray counters of the last frame; every frame traces exactly the same rays (keyed RNG)
    rays = (c["rays_closest"] + c["rays_shadow"]) * args.steps
    samples = c["samples"] * args.steps
    fb = fbs[(args.steps - 1) % 2]

    if nranks > 1:
        t = torch.tensor([elapsed, float(rays), float(samples)], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        rays, samples = int(t[1]), int(t[2])

    if rank == 0 and nranks > 1 and os.environ.get("PRT_BENCH_VERIFY") == "1":
        # outside the timed region: the reduced framebuffer must equal a single-rank render bit for bit
        assembled = fb.clone()
        sc.render_device(None, fb.data_ptr(), stream=stream, **dict(render_kw, rank=0, nranks=1))
        torch.cuda.synchronize()
        print("[bench] N=%d assembled image %s the single-rank image" %
              (nranks, "EQUALS" if torch.equal(assembled, fb) else "DIFFERS FROM"), file=sys.stderr, flush=True)
    if rank == 0:
        # counting instantiation (outside the timed region): mean node fetches / triangle tests per ray
        torch.cuda.synchronize()
        sc.render_device(None, fb.data_ptr(), count_work=True, stream=stream,
                         **dict(render_kw, spp=min(spp, 8), rank=0, nranks=1))
        torch.cuda.synchronize()
        cc = sc.counters()
        cr = max(1, cc["rays_closest"] + cc["rays_shadow"])
        npr, tpr = cc["node_fetches"] / cr, cc["tri_tests"] / cr
        bpr = bytes_per_ray(npr, tpr)
        rays_per_launch = rays / args.steps / nranks  # this rank's launch (tiles are balanced round-robin)
        mean_ms = sum(kernel_ms) / len(kernel_ms)
        if pipelined:
            # overlapped launches: an event pair spans the queueing behind the previous frame as well, so the
            # launch PERIOD (timed region / launches) is the per-launch duration the roofline is priced with
            mean_ms = elapsed * 1e3 / args.steps
        fb_bytes = 24.0 * cam.width * cam.height / nranks  # fp64 per-item partial sums written by K3
        achieved = (bpr * rays_per_launch + fb_bytes) / (mean_ms * 1e-3) / 1e9
        out = {
            "metric": "Mrays/s (primary+secondary) at fixed spp; s/frame in ms_per_step",
            "value": round(rays / elapsed / 1e6, 2),
            "unit": "Mrays/s",
            "n_gpus": nranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3),
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, gloo)" if rehearsal else ""),
            "config": {
                "workload": f"{args.workload} (synthetic stand-in, {data.n_tris} tris) {cam.width}x{cam.height} "
                            f"spp={spp} depth={depth} rr=0.8 bSampleLights seed={seed}",
                "parallelism": f"{render_kw['tile_size']}x{render_kw['tile_size']} tiles dealt diagonally over {nranks} GPU(s)" + ("; frames pipelined on 2 streams" if pipelined else "") + (" + RCCL reduce(sum) of fp32 framebuffer" if nranks > 1 else ""),
                "mpaths_per_s": round(samples / elapsed / 1e6, 3),
                "rays_per_frame": int(rays / args.steps),
                "s_per_frame": round(elapsed / args.steps, 4),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": None,
                "kernel": "k_render",
                "kernel_ms": round(mean_ms, 3),
                "kernel_ms_source": "launch period of overlapped launches" if pipelined else "HIP events around each launch",
                "bytes_per_ray": round(bpr, 1),
                "nodes_per_ray": round(npr, 2),
                "tris_per_ray": round(tpr, 2),
            },
        }
        prof = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(prof):  # PMC-measured HBM bytes per launch of this workload (separate rocprofv3 --pmc passes)
            try:
                tr = json.load(open(prof)).get(args.workload)
                if tr and tr.get("spp") == spp and nranks == 1:
                    out["roofline"]["traffic"] = tr["hbm_bytes_per_launch"]
            except Exception:
                pass
        if nranks == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(data, spp, depth, seed)
            out["cpu_baseline"]["gpu_over_cpu"] = round(out["value"] / max(out["cpu_baseline"]["value"], 1e-9), 1)
        print(json.dumps(out), flush=True)
    if nranks > 1 or force_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
