#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X path-tracing hot path.

Metric (BASELINE.json): Mrays/s (primary + secondary) and wall-clock s/frame at fixed spp.
Headline workload (BASELINE.json configs[1]): synthetic "cornell-box" stand-in (scenes.cornell_box,
20,492 triangles — the reference's asset is not available), 1024x1024, spp=500, depth=20, RR 0.8,
bSampleLights, seed 1, fp64 arithmetic (the reference's).  One "step" = one full frame
(Camera::Render): K3 persistent path-tracing kernel + K5 finalize, inputs (scene, BVH) resident in HBM.

`python bench.py --gpus N` is a complete N-rank run: with WORLD_SIZE unset and N > 1 this process starts
N rank processes (one per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment) BEFORE it
touches a GPU, waits for them and exits with their worst return code.  Under torch.distributed.run the
ranks already exist and the same code runs in each.  Ranks form an `nccl` (= RCCL) group and assert that
its size is N.

N>1: the SAME frame is cut into 16x16 tiles dealt diagonally over ranks (strong scaling: total work
fixed); every rank renders its tiles into a zero-initialised full-size fp32 framebuffer and one RCCL
reduce(sum) to rank 0 assembles the image (disjoint tiles => x + 0 + ... + 0: the reduce itself is exact; a share's
own chunking of the samples makes its fp64 sums differ from the 1-GPU launch's by ~1e-15, invisible in fp32 except
for about one value in 10^9).  The reduce is inside the timed region; consecutive frames alternate between two HIP streams /
framebuffers so the next frame fills the GPU while the previous one drains and is being reduced.  After
the timed region rank 0 renders the frame alone and compares (`assembled_matches_single_rank`, `assembled_exact`), and
BASELINE config 5 (bathroom2 spp=500 depth=50) is timed the same way (`config5`).

Rank 0 prints TWO JSON lines: first the detailed object (every workload's full entry; also written to bench_detail.json),
then — as the LAST line, the one the driver parses — a compact object of < 4 KB with the contract's keys, `roofline`
(bound / achieved / peak / unit / frac / traffic as the bench contract defines them), `cpu_baseline`, `parity_check`,
`checks_ok` and a one-entry-per-workload `workloads_summary`.  Objects of the detailed line:
  roofline      dominant kernel: what bounds it according to the PMC counters of the same build
                (profiles/pmc_summary.json, separate rocprofv3 --pmc passes), priced with the launch
                duration measured live with HIP events on the launch stream.
  cpu_baseline  the CPU oracle (port of the reference algorithm, oracle/pt_oracle.cpp) timed on this
                box's host cores on a bounded sample of the same workload (rank 0, N=1 only).
  parity_check  the rows the CPU baseline rendered, compared with an fp64 GPU frame of the same
                configuration (1e-9 per channel); a mismatch makes the bench fail.
  workloads     (N=1) the other single-GPU workloads, each with its own roofline / parity entry.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0    # MI355X HBM3E nominal (MI355X_MICROARCH.md); ~6300 GB/s is the measured copy rate
N_SIMD = 256 * 4          # 256 CUs x 4 SIMDs
PEAK_CLOCK_GHZ = 2.4
# The `valu` roofline's denominator is MEASURED: tools/valu_calib.hip times every vector instruction the kernels are made of
# at 1-4 waves per SIMD (profiles/r03_valu_calibration.json: plain fp32 / integer ALU ops issue every ~2.2-2.6 cycles per
# SIMD, everything else K1 / K3 use — fp64, conversions, v_alignbit, min / max, v_cndmask, compares, v_pk_fma_f32 — every
# ~4.2, reciprocals 8-16), tools/price_mix.py prices each permutation's ISA with that, weighted by the wave-level counters
# of the counting build (profiles/r04_valu_mix.json).  Fallback when that file has no entry: 4.2 cycles per instruction.
VALU_FALLBACK_GINST = N_SIMD * PEAK_CLOCK_GHZ / 4.2


def valu_peak(workload):
    try:
        mix = json.load(open(os.path.join(ROOT, "profiles", "r04_valu_mix.json")))["workloads"]
        e = mix.get(workload) or mix.get(workload.split("-spp")[0] + ("-f32" if workload.endswith("-f32") else ""))
        if e:
            return float(e["issue_ceiling_ginstr_per_s"]), float(e["avg_cycles_per_valu_instruction"]), "profiles/r04_valu_mix.json"
    except Exception:
        pass
    return VALU_FALLBACK_GINST, 4.2, "fallback: 4.2 cycles per vector instruction (profiles/r03_valu_calibration.json, fp64 / conversion class)"

WORKLOADS = {
    # name: (scene factory name, kwargs, spp, depth)
    "cornell-box": ("cornell_box", {}, 500, 20),
    "veach-mis": ("veach_mis", {}, 3000, 100),
    "bathroom2": ("bathroom", {}, 100, 50),
    "bathroom2-spp500": ("bathroom", {}, 500, 50),            # BASELINE config 5 (the 8-GPU configuration)
    "cornell-ct": ("cornell_box", {"ball_cooktorrance_alpha": 0.1}, 100, 10),  # Results/..._alpha0.1.png configuration
    # fp32 fast mode (PRT_PRECISION_F32, tolerance tier 2): the same frames through prt_kernels_f32.hip
    "cornell-box-f32": ("cornell_box", {}, 500, 20, 1),
    "veach-mis-f32": ("veach_mis", {}, 3000, 100, 1),
    "bathroom2-f32": ("bathroom", {}, 100, 50, 1),
    "bathroom2-spp500-f32": ("bathroom", {}, 500, 50, 1),
}
# K1 closest-hit microbenchmarks: 2^24 seeded incoherent rays resident in HBM
RAY_WORKLOADS = {
    "s0-rays-cornell": ("cornell_box", {}, False),                          # cache-resident geometry
    "s0-rays-cornell-coherent": ("cornell_box", {}, False),                 # ... and camera rays in pixel order (SURVEY §8d "coherent")
    "s4-rays-soup8m": ("triangle_soup", {"n_tris": 8_000_000}, True),       # HBM-resident; tree built on the GPU
}
EXTRA_AT_N1 = ["veach-mis", "bathroom2", "bathroom2-spp500", "cornell-ct", "s0-rays-cornell", "s0-rays-cornell-coherent", "s4-rays-soup8m",
               "cornell-box-f32", "veach-mis-f32", "bathroom2-f32"]
TOL = 1e-9  # per channel, relative to max(1, |x|): fp64 on both sides, differences = FMA contraction + libm ulps


# ------------------------------------------------------------------------------------------------ launcher
def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpus():
    """GPUs this process may use, counted WITHOUT torch or HIP (the launcher must not initialise a GPU runtime before it
    starts its rank processes): the KFD topology nodes that have SIMDs, narrowed by HIP_/ROCR_/CUDA_VISIBLE_DEVICES.
    None when the topology cannot be read — the ranks then fail by themselves if a device is missing."""
    try:
        base = "/sys/class/kfd/kfd/topology/nodes"
        n = 0
        for node in os.listdir(base):
            for line in open(os.path.join(base, node, "properties")):
                if line.startswith("simd_count") and int(line.split()[1]) > 0:
                    n += 1
    except Exception:
        return None
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def spawn_ranks(n):
    """Start n rank processes of this script, one per GPU.  Nothing here initialises a GPU: the children do."""
    rehearsal = os.environ.get("PRT_BENCH_REHEARSAL") == "1"
    if not rehearsal and os.environ.get("PRT_BENCH_LAUNCH_STUB") != "1":
        have = visible_gpus()
        if have is not None and have < n:
            raise SystemExit(f"bench.py --gpus {n}: only {have} GPU(s) visible")
    port = str(free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    for p in procs:
        p.wait()
        rc = rc or p.returncode
    raise SystemExit(rc)


# ------------------------------------------------------------------------------------------------ rooflines
def record_bytes(info):
    return float(info["node_bytes"])


def bytes_per_ray(cc, info, f32=False):
    """Algorithmic bytes per ray, SURVEY.md §8(d) with this build's record sizes and what the kernel really fetches
    (counters of the counting instantiation of the same kernel): ray in (64 B: fp64 o, tmin, d, tmax) + hit out (32 B)
    + one node record per node visit + 32 B (plane n, D) per triangle test + the rest of the record (64 B: the two
    edge functions A, a0, B, b0) per test that passed the plane / interval check."""
    rays = max(1, cc["rays_closest"] + cc["rays_shadow"])
    npr, tpr, fpr = cc["node_fetches"] / rays, cc["tri_tests"] / rays, cc["tri_full"] / rays
    if f32:  # fp32 fast mode: 48-byte records (16-byte plane + 32 bytes of edge functions), fp32 ray / hit
        return 32.0 + 16.0 + record_bytes(info) * npr + 16.0 * tpr + 32.0 * fpr, npr, tpr, fpr
    return 64.0 + 32.0 + record_bytes(info) * npr + 32.0 * tpr + (info["tri_bytes"] - 32.0) * fpr, npr, tpr, fpr


def load_pmc(workload):
    p = os.path.join(ROOT, "profiles", "pmc_summary.json")
    try:
        return json.load(open(p)).get(workload.split("-spp")[0])  # bathroom2-spp500 scales from bathroom2's per-ray counters
    except Exception:
        return None


def roofline(workload, kernel, bpr, npr, tpr, fpr, rays_per_launch, extra_bytes, kernel_ms, kernel_ms_source):
    """What bounds the dominant kernel.  The PMC counters (profiles/pmc_summary.json: per launch, with the rays of that
    launch) decide: a launch whose HBM-side traffic is a large share of the peak is HBM-bound and is priced in bytes;
    a cache-resident one is priced by the issue rate of its vector instructions."""
    sec = kernel_ms * 1e-3
    alg_gbps = (bpr * rays_per_launch + extra_bytes) / sec / 1e9
    pmc = load_pmc(workload)
    out = {"kernel": kernel, "kernel_ms": round(kernel_ms, 3), "kernel_ms_source": kernel_ms_source,
           "bytes_per_ray": round(bpr, 1), "nodes_per_ray": round(npr, 2), "tris_per_ray": round(tpr, 2),
           "tris_full_per_ray": round(fpr, 2), "algorithmic_gbps": round(alg_gbps, 1),
           "algorithmic_frac_of_hbm_peak": round(alg_gbps / HBM_PEAK_GBPS, 4)}
    scale = None
    if pmc and pmc.get("rays_per_launch"):
        scale = rays_per_launch / float(pmc["rays_per_launch"])  # counters scale with the rays of the launch
        out["pmc_round"] = pmc.get("round")
    traffic = pmc["hbm_bytes_per_launch"] * scale if scale and "hbm_bytes_per_launch" in pmc else None
    hbm_gbps = traffic / sec / 1e9 if traffic else None
    if hbm_gbps is not None:
        out["hbm_gbps_measured"] = round(hbm_gbps, 1)
        out["hbm_frac_measured"] = round(hbm_gbps / HBM_PEAK_GBPS, 4)
    valu = pmc["counters_per_launch"].get("SQ_INSTS_VALU") * scale if scale and "SQ_INSTS_VALU" in pmc.get("counters_per_launch", {}) else None
    if pmc:
        for k in ("valu_busy_frac", "wait_any_frac", "l2_hit_rate", "clock_ghz"):
            if k in pmc:
                out["pmc_" + k] = pmc[k]
    if hbm_gbps is not None and hbm_gbps >= 0.4 * HBM_PEAK_GBPS:
        out.update(bound="hbm", achieved=round(alg_gbps, 1), peak=HBM_PEAK_GBPS, unit="GB/s",
                   frac=round(alg_gbps / HBM_PEAK_GBPS, 4), traffic=traffic)
        # scattered reads are served per 128-byte line whatever part of it is used: the rate of L2 misses against the
        # random-line rate measured on this chip (tools/hbm_calib.hip, profiles/r02_hbm_calibration.json)
        misses = pmc["counters_per_launch"].get("TCC_MISS_sum")
        try:
            ceiling = json.load(open(os.path.join(ROOT, "profiles", "r02_hbm_calibration.json")))["conclusions"]["random_line_ceiling_per_us"] * 1e6
        except Exception:
            ceiling = None
        if misses and ceiling:
            lines_per_s = misses * scale / sec
            out.update(lines_per_ray=round(misses * scale / rays_per_launch, 2), random_lines_per_s=round(lines_per_s / 1e9, 2),
                       random_line_ceiling_per_s=round(ceiling / 1e9, 2), frac_of_line_ceiling=round(lines_per_s / ceiling, 4))
    elif valu is not None:
        ginst = valu / sec / 1e9
        peak, cyc, src = valu_peak(workload)
        out.update(bound="valu", achieved=round(ginst, 1), peak=round(peak, 1), unit="Ginstr/s",
                   frac=round(ginst / peak, 4), traffic=traffic,
                   valu_insts_per_ray=round(valu / rays_per_launch, 2), peak_cycles_per_instruction=cyc, peak_source=src)
    else:  # no counters for this build yet: the algorithmic figure only, flagged
        out.update(bound="hbm", achieved=round(alg_gbps, 1), peak=HBM_PEAK_GBPS, unit="GB/s",
                   frac=round(alg_gbps / HBM_PEAK_GBPS, 4), traffic=None, note="no PMC summary for this workload")
    # The object the bench contract asks for, whatever the counters say bounds the launch: ALGORITHMIC bytes of the launch
    # (SURVEY.md §8(d) per-ray figure x rays of the launch) / the launch's duration against the 8 TB/s HBM peak, with the
    # counter-measured HBM bytes of the launch beside it.  On cache-resident scenes traffic << algorithmic bytes: the byte
    # model counts cached bytes there, and the vector-issue fraction (`valu_frac`) is the figure that says how full the chip is.
    out["contract"] = {"bound": "hbm", "achieved": round(alg_gbps, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                       "frac": round(alg_gbps / HBM_PEAK_GBPS, 4), "traffic": traffic}
    return out


# ------------------------------------------------------------------------------------------------ CPU baseline + parity
def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_quota():
    """CPU quota of this process's cgroup in cores (None = unlimited / unreadable)."""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            return round(float(q) / float(per), 2)
    except Exception:
        pass
    try:
        q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
        per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        if q > 0:
            return round(q / per, 2)
    except Exception:
        pass
    return None


def cpu_baseline(scene_data, spp, depth, seed, target_s=8.0, threads_hint=None):
    """Time the CPU oracle (kind "port") on a bounded sample: centre rows of the same image at the same spp/depth,
    independent row workers with keyed per-sample RNG on the host cores this process may use ("all host
    cores, state N = nproc") — the worker count is the fastest of a short scan up to the whole affinity mask, nproc, the
    mask and the scan are reported; beside it the 16-thread figure earlier rounds reported and one row single-threaded.  Returns (json object, rendered rows image, (y0, y1))."""
    import oracle  # test infrastructure used here only as the reported CPU baseline and as the parity checker
    affinity = len(os.sched_getaffinity(0))
    quota = cpu_quota()
    orc = oracle.Oracle(scene_data)
    cam = scene_data.camera
    mid = cam.height // 2

    def band(nthreads, budget_s):
        """rows around the image centre sized for ~budget_s seconds on nthreads workers"""
        n0 = min(nthreads, cam.height)
        pps = scan[nthreads] * 1e6
        rows = int(max(n0, min(cam.height, (pps * budget_s) / (cam.width * spp))))
        rows = min(cam.height, max(n0, (rows // n0) * n0))
        y0 = max(0, min(cam.height - rows, mid - rows // 2))
        t = time.time()
        img, c = orc.render(spp=spp, max_depth=depth, seed=seed, rows=(y0, y0 + rows), nthreads=nthreads)
        return img, c, time.time() - t, (y0, y0 + rows), pps

    # Every core this process may use — and, because the oracle's row workers do not scale to every box's SMT threads
    # (r04a: 256 workers on the box's 256-thread mask ran at 0.6x the 16-worker rate), the thread count that is FASTEST
    # here: a short scan (spp 8 bands) over 16, 32, 64, 128 and the whole mask; the bounded sample runs on the winner.
    scan = {}
    counts = [threads_hint] if threads_hint else sorted({min(affinity, x) for x in (16, 32, 64, 128, affinity)})
    orc.render(spp=1, max_depth=depth, seed=seed, rows=(mid, mid + 1), nthreads=1)  # first call: page-in / lazy set-up, untimed
    scan_spp = min(spp, 64)  # the oracle has a fixed cost per pixel: a scan at spp 1 would not rank the thread counts of an spp-500 run
    for n in counts:         # two rows per worker, around the image centre
        r = min(cam.height, 2 * n)
        best = 0.0
        for _ in range(2):   # best of two: the first call with a new worker count pays for thread stacks / malloc arenas
            t = time.time()
            _, cs = orc.render(spp=scan_spp, max_depth=depth, seed=seed, rows=(mid - r // 2, mid - r // 2 + r), nthreads=n)
            best = max(best, cs["samples"] / max(time.time() - t, 1e-3) / 1e6)
        scan[n] = round(best, 3)
    threads = max(scan, key=scan.get)
    img, c, dt, rows, paths_per_s = band(threads, target_s)
    rays = c["rays_closest"] + c["rays_shadow"]
    # single thread: a bounded share of one row
    spp1 = max(1, min(spp, int(paths_per_s / threads * 2.0 / cam.width)))
    t = time.time()
    _, c1 = orc.render(spp=spp1, max_depth=depth, seed=seed, rows=(mid, mid + 1), nthreads=1)
    dt1 = max(time.time() - t, 1e-6)
    out = {
        "value": round(rays / dt / 1e6, 3),
        "unit": "Mrays/s",
        "cores": threads,
        "kind": "port",
        "sample": f"rows {rows[0]}..{rows[1]} of the {cam.width}x{cam.height} frame at spp={spp} depth={depth} "
                  f"({c['samples']} camera samples, {rays} rays, {dt:.2f} s, {threads} row-worker threads)",
        "mpaths_per_s": round(c["samples"] / dt / 1e6, 4),
        "seconds": round(dt, 2),
        "nproc": os.cpu_count(),
        "affinity_cores": affinity,
        "cgroup_cpu_quota": quota,
        "thread_scan_mpaths_per_s": scan,
        "cpu_model": cpu_model(),
        "single_thread": {"value": round((c1["rays_closest"] + c1["rays_shadow"]) / dt1 / 1e6, 3), "unit": "Mrays/s",
                          "mpaths_per_s": round(c1["samples"] / dt1 / 1e6, 4),
                          "sample": f"row {mid} at spp={spp1} ({dt1:.2f} s)"},
        "ray_count_note": "rays = BVH traversals actually executed.  The oracle counts every world.Hit the reference issues "
                          "(one camera ray per SAMPLE); the GPU traces the pixel's one camera ray once per work item and starts "
                          "every sample from that hit (Camera.cpp:53-57: the ray is the same for all samples), and does not trace "
                          "rays whose result the reference discards — so compare paths/s (gpu_over_cpu_paths) and s/frame, not rays/s",
    }
    if threads != 16 and 16 in scan and not threads_hint:  # the figure rounds 1-3 reported (16 row workers), on a quarter of the budget
        _, c16, dt16, rows16, _ = band(16, target_s / 4)
        out["threads16"] = {"value": round((c16["rays_closest"] + c16["rays_shadow"]) / dt16 / 1e6, 3), "unit": "Mrays/s", "cores": 16,
                            "mpaths_per_s": round(c16["samples"] / dt16 / 1e6, 4),
                            "sample": f"rows {rows16[0]}..{rows16[1]} ({dt16:.2f} s, 16 row-worker threads)"}
    return out, img, rows


def parity_rows(gpu_f64, ref_img, rows):
    import numpy as np
    y0, y1 = rows
    g, r = gpu_f64[y0:y1], ref_img[y0:y1]
    scale = np.maximum(1.0, np.abs(r))
    rel = np.abs(g - r) / scale
    bad = (rel > TOL).any(-1)
    return {"rows": [int(y0), int(y1)], "pixels": int(bad.size), "bad_px": int(bad.sum()),
            "max_rel": float(rel.max()) if rel.size else 0.0, "tolerance": TOL,
            "ok": bool(bad.mean() <= 2e-5 if bad.size else True)}  # observed: no pixel beyond the tolerance on any workload


# ------------------------------------------------------------------------------------------------ workloads
class Ctx:
    pass


def time_render(ctx, name, steps, warmup, spp_override=0, with_cpu=True, cpu_target_s=8.0):
    """Times `steps` frames of a render workload on the current rank set; returns the JSON object of the workload."""
    torch, dist, api, scenes, distributed = ctx.torch, ctx.dist, ctx.api, ctx.scenes, ctx.distributed
    rank, nranks, rehearsal = ctx.rank, ctx.nranks, ctx.rehearsal
    fn, kw, spp, depth = WORKLOADS[name][:4]
    precision = WORKLOADS[name][4] if len(WORKLOADS[name]) > 4 else 0
    if spp_override > 0:
        spp = spp_override
    seed = 1
    data = getattr(scenes, fn)(**kw)
    cam = data.camera
    sc = api.Scene(data).upload(ctx.local_rank)
    info = sc.bvh_info()
    # Two framebuffers on two HIP streams: consecutive frames alternate, so frame k+1 starts filling the GPU while
    # the last long paths of frame k drain and its framebuffer is being reduced (RCCL overlaps with compute).  Every
    # step is still one complete frame; libprt_hip double-buffers its per-call state.  At N = 1 the frames run back
    # to back on one stream so that the HIP events around each launch bracket exactly that launch.
    pipelined = (ctx.args.pipeline or nranks > 1) and not ctx.args.no_pipeline and not rehearsal
    fbs = [torch.zeros((cam.height, cam.width, 3), dtype=torch.float32, device="cuda") for _ in range(2)]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()] if pipelined else [torch.cuda.current_stream()] * 2
    stream = torch.cuda.current_stream().cuda_stream
    render_kw = dict(spp=spp, max_depth=depth, seed=seed, rank=rank, nranks=nranks, tile_size=16 if nranks > 1 else 32,
                     precision=precision)

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
        if ctx.in_group:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(warmup):
        step(k)
    fence()
    evs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(steps)]
    t0 = time.perf_counter()
    for k in range(steps):
        step(k, evs[k])
    fence()
    elapsed = time.perf_counter() - t0
    # HIP events recorded on the stream each K3 was launched on (memsets + K3 + K5; K3 is > 99.9 % of it)
    kernel_ms = [a.elapsed_time(b) for a, b in evs]
    c = sc.counters()  # ray counters of the last frame; every frame traces exactly the same rays (keyed RNG)
    rays = (c["rays_closest"] + c["rays_shadow"]) * steps
    samples = c["samples"] * steps
    fb = fbs[(steps - 1) % 2]

    if nranks > 1:
        t = torch.tensor([elapsed, float(rays), float(samples)], dtype=torch.float64, device="cpu" if rehearsal else "cuda")
        tmax = t.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        elapsed = float(tmax[0])
        rays, samples = int(t[1]), int(t[2])

    out = None
    if rank == 0:
        out = {"workload": name, "value": round(rays / elapsed / 1e6, 2), "unit": "Mrays/s", "steps": steps, "warmup": warmup,
               "ms_per_step": round(elapsed / steps * 1e3, 3), "mpaths_per_s": round(samples / elapsed / 1e6, 3),
               "rays_per_frame": int(rays / steps),
               "config": f"{name} (synthetic stand-in, {data.n_tris} tris, {info['width']}-wide BVH {info['n_nodes']} nodes) "
                         f"{cam.width}x{cam.height} spp={spp} depth={depth} rr=0.8 bSampleLights seed={seed}",
               "tile_size": render_kw["tile_size"], "pipelined": bool(pipelined), "dtype": "f32" if precision else "f64"}
        if nranks > 1:
            # outside the timed region: the reduced framebuffer must equal a single-rank render bit for bit
            assembled = fb.clone()
            sc.render_device(None, fb.data_ptr(), stream=stream, **dict(render_kw, rank=0, nranks=1))
            torch.cuda.synchronize()
            # A share deals a pixel's samples in chunks sized for ITS pixel count, so the per-pixel sum of chunk sums groups
            # the samples differently from the single launch: the fp64 sums agree to ~1e-15 (whole-frame measurement:
            # profiles/r03_full_frame_variants.json), and the fp32 framebuffer values are the same floats except where that
            # difference straddles a rounding boundary — about one value in 10^9.  Hence: equal, or within one float ulp
            # on at most a handful of values.  (fp32 fast mode: a chunk's sum is itself an fp32 one: rtol 1e-5.)
            differing = int((assembled != fb).sum())
            out["assembled_differing_values"] = differing
            out["assembled_exact"] = differing == 0
            out["assembled_matches_single_rank"] = bool(torch.allclose(assembled, fb, rtol=2.5e-7, atol=1e-12) and differing <= 16) if not precision else \
                bool(torch.allclose(assembled, fb, rtol=1e-5, atol=1e-9))
        # counting instantiation (outside the timed region): node fetches / triangle tests per ray
        torch.cuda.synchronize()
        sc.render_device(None, fb.data_ptr(), count_work=True, stream=stream, **dict(render_kw, spp=min(spp, 8), rank=0, nranks=1))
        torch.cuda.synchronize()
        bpr, npr, tpr, fpr = bytes_per_ray(sc.counters(), info, f32=bool(precision))
        rays_per_launch = rays / steps / nranks  # this rank's launch (tiles are balanced round-robin)
        mean_ms = sum(kernel_ms) / len(kernel_ms)
        src = "HIP events around each launch"
        if pipelined:
            # overlapped launches: an event pair spans the queueing behind the previous frame as well, so the launch
            # PERIOD (timed region / launches) is the per-launch duration the roofline is priced with
            mean_ms, src = elapsed * 1e3 / steps, "launch period of overlapped launches"
        fb_bytes = 24.0 * cam.width * cam.height / nranks  # fp64 per-item partial sums written by K3
        out["roofline"] = roofline(name, "k_render", bpr, npr, tpr, fpr, rays_per_launch, fb_bytes, mean_ms, src)
        if precision:
            out["note"] = ("fp32 fast mode (PRT_PRECISION_F32): an opt-in extra, NOT the headline — `value` of the JSON line and every "
                           "1e-9 parity claim are fp64; this entry is checked at the second tolerance tier only")
            # tolerance tier 2: against the fp64 kernels' frame of the same seeds (itself checked against the oracle above)
            ref = torch.zeros((cam.height, cam.width, 3), dtype=torch.float64, device="cuda")
            got = torch.zeros_like(ref)
            sc.render_device(ref.data_ptr(), None, stream=stream, **dict(render_kw, rank=0, nranks=1, precision=0))
            sc.render_device(got.data_ptr(), None, stream=stream, **dict(render_kw, rank=0, nranks=1))
            torch.cuda.synchronize()
            rel = ((got - ref).abs() / ref.abs().clamp(min=1.0)).amax(dim=-1)
            mean_rel = float(((got.mean() - ref.mean()).abs() / ref.mean().abs()).item())
            far = float((rel > 1e-2).double().mean().item())
            out["parity_check"] = {"against": "the fp64 kernels' frame of the same seeds", "mean_rel": mean_rel,
                                   "px_beyond_1e-2": far, "px_rel_p999": float(torch.quantile(rel.flatten()[:: max(1, rel.numel() // 1_000_000)], 0.999).item()),
                                   "tolerance": "image mean 1e-3; at most 0.1 % of the pixels beyond 1e-2 (diverged paths)",
                                   "ok": bool(mean_rel <= 1e-3 and far <= 1e-3)}
        elif nranks == 1 and with_cpu and not ctx.args.no_cpu_baseline:
            # the thread-count scan runs once, on the headline workload; the other workloads use its winner
            base, ref_img, rows = cpu_baseline(data, spp, depth, seed, target_s=cpu_target_s, threads_hint=getattr(ctx, "cpu_threads", None))
            ctx.cpu_threads = base["cores"]
            base["gpu_over_cpu"] = round(out["value"] / max(base["value"], 1e-9), 1)
            base["gpu_over_cpu_paths"] = round(out["mpaths_per_s"] / max(base["mpaths_per_s"], 1e-9), 1)
            out["cpu_baseline"] = base
            # parity on exactly this configuration: one fp64 frame of the same (spp, depth, seed), outside the timed region
            f64 = torch.zeros((cam.height, cam.width, 3), dtype=torch.float64, device="cuda")
            kw64 = dict(render_kw, rank=0, nranks=1)
            if os.environ.get("PRT_BENCH_FAULT") == "1":  # test hook: the check must notice a frame that is not the oracle's
                kw64["seed"] = seed + 1
            sc.render_device(f64.data_ptr(), None, stream=stream, **kw64)
            torch.cuda.synchronize()
            out["parity_check"] = parity_rows(f64.cpu().numpy(), ref_img, rows)
    sc.close()
    return out


def time_rays(ctx, name, steps, warmup):
    """One step = one K1 launch over 2^24 rays (rays and hits stay in HBM)."""
    import numpy as np
    torch, api, scenes = ctx.torch, ctx.api, ctx.scenes
    fn, kw, device_bvh = RAY_WORKLOADS[name]
    data = getattr(scenes, fn)(**kw)
    t0 = time.time()
    sc = api.Scene(data, device_bvh=device_bvh)
    sc.upload(0)
    build_s = time.time() - t0
    info = sc.bvh_info()
    n = 1 << 24
    lo, hi = data.bounds()
    coherent = name.endswith("-coherent")
    rays_np = scenes.camera_rays(data.camera, n, seed=12345) if coherent else scenes.random_rays(n, lo, hi, seed=12345)
    d_r = torch.from_numpy(rays_np.view(np.float64).reshape(-1, 8)).cuda()
    d_h = torch.zeros((n, 4), dtype=torch.float64, device="cuda")
    for _ in range(warmup):
        sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr())
    torch.cuda.synchronize()
    ms = []
    t0 = time.perf_counter()
    for _ in range(steps):
        sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr())
        ms.append(sc.counters()["kernel_ms"])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    sc.trace_closest_device(d_r.data_ptr(), n, d_h.data_ptr(), count_work=True)
    torch.cuda.synchronize()
    cc = sc.counters()
    cc["rays_closest"], cc["rays_shadow"] = n, 0
    bpr, npr, tpr, fpr = bytes_per_ray(cc, info)
    hits = d_h.cpu().numpy().view(np.dtype([("t", "<f8"), ("a", "<f8"), ("b", "<f8"), ("prim", "<i4"), ("front", "<i4")])).reshape(-1)
    out = {"workload": name, "value": round(n * steps / elapsed / 1e6, 2), "unit": "Mrays/s", "steps": steps, "warmup": warmup,
           "ms_per_step": round(elapsed / steps * 1e3, 3), "rays_per_launch": n,
           "config": f"{name}: {n} seeded {'camera rays in pixel order' if coherent else 'random rays'} vs {data.n_tris} triangles ({info['width']}-wide BVH, {info['n_nodes']} nodes, "
                     f"depth {info['depth']}, {'GPU' if info['built_on_device'] else 'host'} build, create+upload {build_s:.1f} s)",
           "hit_fraction": round(float((hits["prim"] >= 0).mean()), 4),
           "roofline": roofline(name, "k_trace_closest", bpr, npr, tpr, fpr, n, 0.0, sum(ms) / len(ms), "HIP events around each launch")}
    # the same batch through the fp32 fast mode (tolerance tier 2: |dt| / max(1, t) <= 1e-5 against the fp64 kernel's hits)
    d_h32 = torch.zeros_like(d_h)
    sc.trace_closest_device(d_r.data_ptr(), n, d_h32.data_ptr(), precision=1)
    torch.cuda.synchronize()
    ms32 = []
    t0 = time.perf_counter()
    for _ in range(steps):
        sc.trace_closest_device(d_r.data_ptr(), n, d_h32.data_ptr(), precision=1)
        ms32.append(sc.counters()["kernel_ms"])
    torch.cuda.synchronize()
    el32 = time.perf_counter() - t0
    h32 = d_h32.cpu().numpy().view(hits.dtype).reshape(-1)
    both = (hits["prim"] >= 0) & (h32["prim"] >= 0)
    rel = np.abs(h32["t"][both] - hits["t"][both]) / np.maximum(1.0, hits["t"][both])
    flips = int(((hits["prim"] >= 0) != (h32["prim"] >= 0)).sum())
    out["f32"] = {"note": "fp32 fast mode of the same batch: an opt-in extra, not the workload's `value` (which is fp64)",
                  "value": round(n * steps / el32 / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(el32 / steps * 1e3, 3),
                  "kernel_ms": round(sum(ms32) / len(ms32), 3), "dtype": "f32",
                  "parity_check": {"against": "the fp64 kernel's hits", "hit_miss_flips": flips, "beyond_1e-5": int((rel > 1e-5).sum()),
                                   "rel_t_p999": float(np.quantile(rel, 0.999)) if rel.size else 0.0, "tolerance": "1e-5 for all but 1e-4 of the rays",
                                   "ok": bool(flips + int((rel > 1e-5).sum()) <= 1e-4 * n)}}
    if data.n_tris >= 1_000_000:
        # K4: the same batch traced in a locality order (keys + radix sort + K1 through the permutation, all inside the time);
        # the hits must be the unsorted call's, bit for bit.  An extra: the workload's `value` is the batch as given.
        d_hs = torch.zeros_like(d_h)
        sc.trace_closest_device(d_r.data_ptr(), n, d_hs.data_ptr(), sort=True)
        torch.cuda.synchronize()
        mss = []
        t0 = time.perf_counter()
        for _ in range(steps):
            sc.trace_closest_device(d_r.data_ptr(), n, d_hs.data_ptr(), sort=True)
            mss.append(sc.counters()["kernel_ms"])
        torch.cuda.synchronize()
        els = time.perf_counter() - t0
        out["k4_sorted"] = {"note": "prt_trace_closest_sorted_device: keys + radix sort + K1 in locality order, all timed; an extra, not `value`",
                            "value": round(n * steps / els / 1e6, 2), "unit": "Mrays/s", "ms_per_step": round(els / steps * 1e3, 3),
                            "device_ms": round(sum(mss) / len(mss), 3),
                            "hits_equal_unsorted": bool(np.array_equal(d_hs.cpu().numpy().view(np.uint64), d_h.cpu().numpy().view(np.uint64))),
                            "reference_profile": {"note": "NOT measured by this run: L2 misses per ray from an earlier PMC profile",
                                                  "l2_misses_per_ray_as_given": 32.2, "l2_misses_per_ray_sorted": 21.8,
                                                  "source": "profiles/r03_k4_sorted_pmc.txt (TCC_MISS_sum per launch / 2^24)"}}
        del d_hs
    if not ctx.args.no_cpu_baseline and data.n_tris <= 200_000:
        import oracle
        orc = oracle.Oracle(data)
        m = 200_000
        t0 = time.time()
        ref = orc.trace_closest(rays_np[:m])
        dt = time.time() - t0
        out["cpu_baseline"] = {"value": round(m / dt / 1e6, 3), "unit": "Mrays/s", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
                               "sample": f"first {m} rays of the same batch, single thread ({dt:.2f} s)"}
        g = hits[:m]
        same = (g["prim"] == ref["prim"]) | (g["t"] == ref["t"])  # exact ties may name the other triangle
        hit = ref["prim"] >= 0
        terr = np.abs(g["t"][hit] - ref["t"][hit]) / np.maximum(1.0, ref["t"][hit])
        out["parity_check"] = {"rays": m, "prim_mismatch": int((~same).sum()), "max_rel_t": float(terr.max()) if terr.size else 0.0,
                               "tolerance": 1e-12, "ok": bool((~same).sum() == 0 and (terr.size == 0 or terr.max() <= 1e-12))}
    sc.close()
    return out


# ------------------------------------------------------------------------------------------------ output lines
COMPACT_LIMIT = 4000  # bytes; the driver keeps a bounded tail of stdout and parses the LAST line (a 24.8 KB line was cut)
RAY_DEFINITION = ("ray = one BVH traversal executed; the camera ray is traced once per work item since r03 "
                  "(compare ms_per_step / mpaths_per_s across rounds, not Mrays/s)")


def _short(s, n):
    s = str(s)
    return s if len(s) <= n else s[: n - 1] + "…"


def compact_roofline(rf):
    """bench contract: {"bound", "achieved", "peak", "unit", "frac", "traffic"} = algorithmic bytes per launch / launch duration
    against the HBM peak + the PMC-measured HBM bytes per launch; the counter-decided diagnosis rides along in a few scalars."""
    c = dict(rf.get("contract") or {"bound": "hbm", "achieved": rf.get("algorithmic_gbps"), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                                    "frac": rf.get("algorithmic_frac_of_hbm_peak"), "traffic": rf.get("traffic")})
    c.update(kernel=rf.get("kernel"), kernel_ms=rf.get("kernel_ms"), bytes_per_ray=rf.get("bytes_per_ray"),
             nodes_per_ray=rf.get("nodes_per_ray"), tris_per_ray=rf.get("tris_per_ray"))
    if rf.get("bound") == "valu":  # cache-resident launch: what the counters say bounds it
        c.update(limited_by="valu issue (cache-resident: traffic << algorithmic bytes)", valu_frac=rf.get("frac"),
                 valu_ginstr_per_s=rf.get("achieved"), valu_peak_ginstr_per_s=rf.get("peak"))
    elif "frac_of_line_ceiling" in rf:
        c.update(limited_by="random 128-B line rate", frac_of_line_ceiling=rf["frac_of_line_ceiling"], hbm_frac_measured=rf.get("hbm_frac_measured"))
    for k in ("pmc_round", "pmc_valu_busy_frac", "pmc_l2_hit_rate"):
        if k in rf:
            c[k] = rf[k]
    return c


def compact_cpu_baseline(b):
    c = {k: b[k] for k in ("value", "unit", "cores", "kind") if k in b}
    c["sample"] = _short(b.get("sample", ""), 200)
    for k in ("mpaths_per_s", "nproc", "affinity_cores", "cgroup_cpu_quota", "thread_scan_mpaths_per_s", "cpu_model", "gpu_over_cpu", "gpu_over_cpu_paths"):
        if k in b:
            c[k] = b[k]
    if "single_thread" in b:
        c["single_thread_mrays_per_s"] = b["single_thread"]["value"]
    if "threads16" in b:
        c["threads16_mrays_per_s"] = b["threads16"]["value"]
        c["threads16_mpaths_per_s"] = b["threads16"]["mpaths_per_s"]
    return c


def summary_entry(x):
    rf = x["roofline"]
    e = {"name": x["workload"], "mrays_per_s": x["value"], "ms_per_step": x["ms_per_step"]}
    if x.get("mpaths_per_s") is not None:
        e["mpaths_per_s"] = x["mpaths_per_s"]
    e["frac"] = rf["contract"]["frac"] if "contract" in rf else rf.get("algorithmic_frac_of_hbm_peak")
    if rf.get("bound") == "valu":
        e["valu_frac"] = rf.get("frac")
    pc = x.get("parity_check", {})
    e["parity_ok"] = pc.get("ok")
    if "bad_px" in pc:
        e["bad_px"] = pc["bad_px"]
    if "k4_sorted" in x:
        e["k4_sorted_mrays_per_s"] = x["k4_sorted"]["value"]
    return e


def compact_line(full):
    """The LAST stdout line: the bench contract's keys + roofline + cpu_baseline + the checks, in < COMPACT_LIMIT bytes.
    `full` is the detailed object (printed on an earlier line and written to bench_detail.json)."""
    out = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                "vs_baseline", "dtype", "data") if k in full}
    cfg = dict(full.get("config", {}))
    if "workload" in cfg:
        cfg["workload"] = _short(cfg["workload"], 220)
    out["config"] = cfg
    if "roofline" in full:
        out["roofline"] = compact_roofline(full["roofline"])
    if "cpu_baseline" in full:
        out["cpu_baseline"] = compact_cpu_baseline(full["cpu_baseline"])
    for k in ("parity_check", "assembled_matches_single_rank", "assembled_exact", "assembled_differing_values", "checks_ok"):
        if k in full:
            out[k] = full[k]
    if "config5" in full and full["config5"]:
        c5 = full["config5"]
        out["config5"] = {"workload": _short(c5.get("config", c5.get("workload")), 160), "value": c5["value"], "unit": c5["unit"],
                          "ms_per_step": c5["ms_per_step"], "mpaths_per_s": c5.get("mpaths_per_s"), "steps": c5.get("steps"),
                          "frac": c5["roofline"].get("algorithmic_frac_of_hbm_peak"),
                          "assembled_matches_single_rank": c5.get("assembled_matches_single_rank"),
                          "assembled_exact": c5.get("assembled_exact")}
    for k in ("f32", "k4_sorted"):
        if k in full:
            out[k] = {kk: full[k][kk] for kk in ("value", "unit", "ms_per_step") if kk in full[k]}
            pc = full[k].get("parity_check")
            if pc:
                out[k]["parity_ok"] = pc.get("ok")
            if "hits_equal_unsorted" in full[k]:
                out[k]["hits_equal_unsorted"] = full[k]["hits_equal_unsorted"]
    if "workloads_summary" in full:
        out["workloads_summary"] = full["workloads_summary"]
    out["detail"] = "bench_detail.json (also the previous stdout line)"
    # never exceed the limit: shed the optional parts in order of dispensability
    for shed in ("detail", ("workloads_summary", "optional"), ("cpu_baseline", "sample"), ("config", "parallelism"), "workloads_summary"):
        if len(json.dumps(out)) <= COMPACT_LIMIT:
            break
        if shed == ("workloads_summary", "optional"):
            out["workloads_summary"] = [{k: e[k] for k in ("name", "mrays_per_s", "ms_per_step", "frac", "parity_ok") if k in e}
                                        for e in out.get("workloads_summary", [])]
        elif isinstance(shed, tuple):
            if shed[0] in out and shed[1] in out[shed[0]]:
                out[shed[0]][shed[1]] = _short(out[shed[0]][shed[1]], 60)
        else:
            out.pop(shed, None)
    return json.dumps(out)


def emit(full):
    """Detailed object first (one line + bench_detail.json), the compact contract line LAST."""
    detail = json.dumps(full)
    try:
        with open(os.path.join(ROOT, "bench_detail.json"), "w") as f:
            f.write(detail + "\n")
    except OSError:
        pass
    print(detail, flush=True)
    line = compact_line(full)
    assert len(line) <= COMPACT_LIMIT + 96, len(line)
    print(line, flush=True)



def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default="cornell-box", choices=sorted(WORKLOADS) + sorted(RAY_WORKLOADS))
    ap.add_argument("--spp", type=int, default=0, help="override spp (0 = the workload's)")
    ap.add_argument("--precision", choices=["f64", "f32"], default="f64",
                    help="f64 = the reference's arithmetic (default, the headline); f32 = the fp32 fast mode of the same workload (tolerance tier 2)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="headline workload only (no `workloads` / `config5` entries)")
    ap.add_argument("--no-pipeline", action="store_true", help="one stream, one framebuffer: frames strictly back to back")
    ap.add_argument("--pipeline", action="store_true", help="alternate two streams / framebuffers also at N=1 (default: only for N>1)")
    args = ap.parse_args()
    if args.gpus < 1:
        raise SystemExit("--gpus must be >= 1")

    if args.precision == "f32" and not args.workload.endswith("-f32"):
        if args.workload + "-f32" not in WORKLOADS:
            raise SystemExit(f"--precision f32: no fp32 variant of {args.workload}")
        args.workload += "-f32"
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU")
    nranks = world
    if os.environ.get("PRT_BENCH_LAUNCH_STUB") == "1":
        # launcher test hook (no GPU, no torch): every rank reports what it was given; rank 0 prints the JSON line
        print(json.dumps({"stub": True, "rank": rank, "n_gpus": nranks, "master": os.environ.get("MASTER_ADDR"),
                          "port": os.environ.get("MASTER_PORT")}), flush=True)
        return

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
    in_group = nranks > 1 or force_dist
    if in_group:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        if rehearsal:
            dist.init_process_group(backend="gloo", rank=rank, world_size=nranks)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=nranks, device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
        assert rehearsal or dist.get_backend() == "nccl"
        dist.barrier()

    ctx = Ctx()
    ctx.args, ctx.torch, ctx.dist, ctx.api, ctx.scenes, ctx.distributed = args, torch, dist, api, scenes, distributed
    ctx.rank, ctx.local_rank, ctx.nranks, ctx.rehearsal, ctx.in_group = rank, local_rank, nranks, rehearsal, in_group

    ok = True
    if args.workload in RAY_WORKLOADS:
        if nranks != 1:
            raise SystemExit("ray microbenchmarks are single-GPU")
        w = time_rays(ctx, args.workload, args.steps, args.warmup)
        out = {"metric": "Mrays/s (closest-hit, %s rays)" % ("coherent camera" if args.workload.endswith("-coherent") else "incoherent"), "value": w["value"], "unit": "Mrays/s", "n_gpus": 1,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": w["ms_per_step"], "higher_is_better": True,
               "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": w["config"], "hit_fraction": w["hit_fraction"], "rays_per_launch": w["rays_per_launch"]},
               "roofline": w["roofline"]}
        for k in ("cpu_baseline", "parity_check", "f32", "k4_sorted"):
            if k in w:
                out[k] = w[k]
        ok = out.get("parity_check", {}).get("ok", True) and out.get("k4_sorted", {}).get("hits_equal_unsorted", True)
        out["checks_ok"] = bool(ok)
        emit(out)
    else:
        w = time_render(ctx, args.workload, args.steps, args.warmup, spp_override=args.spp)
        extras, config5 = [], None
        if not args.no_extra and args.workload in ("cornell-box", "cornell-box-f32") and args.spp == 0:
            if nranks > 1:
                # BASELINE config 5: bathroom2 spp 500 depth 50, tiles over the ranks + RCCL reduce
                config5 = time_render(ctx, "bathroom2-spp500" + ("-f32" if args.precision == "f32" else ""), max(1, min(args.steps, 3)), 1, with_cpu=False)
            elif args.precision == "f64":
                for name in EXTRA_AT_N1:
                    if name in RAY_WORKLOADS:
                        extras.append(time_rays(ctx, name, 5, 1))
                    else:
                        extras.append(time_render(ctx, name, 2 if name in ("veach-mis", "bathroom2-spp500") else 3, 1, cpu_target_s=3.0))
        if rank == 0:
            rf = w["roofline"]
            out = {
                "metric": "Mrays/s (primary+secondary) at fixed spp; s/frame in ms_per_step",
                "value": w["value"],
                "unit": "Mrays/s",
                "n_gpus": nranks,
                "steps": args.steps,
                "warmup": args.warmup,
                "ms_per_step": w["ms_per_step"],
                "higher_is_better": True,
                "scaling": "strong",
                "vs_baseline": None,
                "dtype": w["dtype"],
                "data": "synthetic" + (" (REHEARSAL: all ranks on one GPU, gloo)" if rehearsal else ""),
                "config": {
                    "workload": w["config"],
                    "parallelism": f"{w['tile_size']}x{w['tile_size']} tiles dealt diagonally over {nranks} GPU(s)"
                                   + ("; frames pipelined on 2 streams" if w["pipelined"] else "")
                                   + (" + RCCL reduce(sum) of fp32 framebuffer" if nranks > 1 else ""),
                    "mpaths_per_s": w["mpaths_per_s"],
                    "rays_per_frame": w["rays_per_frame"],
                    "s_per_frame": round(w["ms_per_step"] / 1e3, 4),
                },
                "roofline": rf,
            }
            for k in ("cpu_baseline", "parity_check", "assembled_matches_single_rank", "assembled_exact", "assembled_differing_values"):
                if k in w:
                    out[k] = w[k]
            out["config"]["ray_definition"] = RAY_DEFINITION
            if config5 is not None:
                out["config5"] = config5
            checks = [w] + extras + ([config5] if config5 else [])
            ok = all(x.get("parity_check", {}).get("ok", True) and x.get("assembled_matches_single_rank", True) and
                     x.get("f32", {}).get("parity_check", {}).get("ok", True) and x.get("k4_sorted", {}).get("hits_equal_unsorted", True) for x in checks)
            out["checks_ok"] = bool(ok)
            out["workloads_summary"] = [summary_entry(x) for x in checks]
            if extras:
                out["workloads"] = extras
            emit(out)
    if in_group:
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device="cpu" if rehearsal else "cuda")
        dist.broadcast(flag, src=0)
        ok = bool(int(flag[0]))
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        raise SystemExit("bench.py: a parity / assembly check failed (see parity_check / assembled_matches_single_rank)")


if __name__ == "__main__":
    main()
