"""GPU tests of the C++ drop-in host API (include/pooraytracer/*.h over the C ABI): the main.cpp-style
driver examples/render_scene.cpp must produce the same framebuffer as the Python binding (same C ABI,
same inputs => bit-identical) and match the oracle within the parity tolerance."""
import os
import subprocess

import numpy as np
import pytest

import oracle
from pooraytracer_amd import _abi, api, build, distributed, scenes

pytestmark = pytest.mark.gpu


def test_render_scene_driver_matches_binding_and_oracle(gpu, tmp_path):
    exe = build.build_host_example()
    data = scenes.mixed_materials(40, 32)
    dump = str(tmp_path / "scene.bin")
    scenes.dump_scene(data, dump)
    out, png = str(tmp_path / "out.f64"), str(tmp_path / "out.png")
    r = subprocess.run([exe, dump, "6", "8", out, png], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    assert "centre ray hits" in r.stdout
    img = np.fromfile(out, dtype=np.float64).reshape(32, 40, 3)
    ref = api.Scene(data).upload(gpu).render(spp=6, max_depth=8, seed=1)
    assert np.array_equal(img, ref)
    cpu, _ = oracle.Oracle(data).render(spp=6, max_depth=8, seed=1)
    bad = (np.abs(img - cpu) > 1e-9 * np.maximum(1.0, np.abs(cpu))).any(-1)
    assert bad.mean() <= 1e-4
    # PNG / HDR written by Camera::WriteColorAttachment (Camera.cpp:279-331 semantics)
    from PIL import Image
    im = np.asarray(Image.open(png))
    x = np.nan_to_num(img, nan=0.0)
    srgb = np.where(x <= 0.0031308, 12.92 * x, 1.055 * np.power(np.maximum(x, 0), 1 / 2.4) - 0.055)
    expect = (np.clip(srgb, 0, 0.9999) * 255).astype(np.uint8)
    assert im.shape == (32, 40, 3) and np.abs(im.astype(int) - expect.astype(int)).max() <= 1
    hdr = png[:-4] + ".hdr"
    assert os.path.getsize(hdr) > 32 * 40 * 4


def test_camera_devices_and_device_bvh(gpu, tmp_path):
    """Camera::devices cuts the frame into tiles over several scene replicas (here all on the one GPU of the box: tile
    shares summed on that device; on different GPUs the same call reduces over RCCL) through prt_render_multi, whose
    frame is the fp32 framebuffer: it must equal the single-device image rounded to float, bit for bit.
    Camera::bBuildBvhOnDevice (GPU-built BVH) must not change the image either."""
    exe = build.build_host_example()
    data = scenes.mixed_materials(56, 40)
    dump = str(tmp_path / "scene.bin")
    scenes.dump_scene(data, dump)
    outs = {}
    for tag, env in (("one", {}), ("three", {"PRT_EXAMPLE_DEVICES": f"{gpu},{gpu},{gpu}"}),
                     ("gpubvh", {"PRT_EXAMPLE_DEVICE_BVH": "1"})):
        out = str(tmp_path / f"{tag}.f64")
        r = subprocess.run([exe, dump, "5", "6", out], capture_output=True, text=True, timeout=300,
                           env=dict(os.environ, **env))
        assert r.returncode == 0, r.stderr + r.stdout
        outs[tag] = np.fromfile(out, dtype=np.float64)
    assert np.array_equal(outs["one"].astype(np.float32).astype(np.float64), outs["three"])
    assert np.array_equal(outs["one"], outs["gpubvh"])


def test_render_multi_on_one_device_and_over_rccl(gpu):
    """prt_render_multi (the C-ABI entry Camera::devices goes through).  Replicas on ONE device: tile shares summed there.
    Replicas on DIFFERENT devices (when the box has them): one ncclReduce(sum, float) to the first device.  Either way the
    frame equals prt_render's fp32 output bit for bit (disjoint tiles: x + 0 + ... + 0).  Mixed placements, a handle
    listed twice and scenes that are not uploaded are refused."""
    data = scenes.mixed_materials(72, 56)
    base = api.Scene(data).upload(gpu)
    _, want = base.render(spp=6, max_depth=6, seed=3, f32=True)
    reps = [api.Scene(data).upload(gpu) for _ in range(3)]
    got = api.render_multi(reps, spp=6, max_depth=6, seed=3)
    assert np.array_equal(got, want)
    assert np.array_equal(api.render_multi(reps[:1], spp=6, max_depth=6, seed=3), want)
    with pytest.raises(api.PrtError):
        api.render_multi([reps[0], reps[0]], spp=2, max_depth=2)
    with pytest.raises(api.PrtError):
        api.render_multi([reps[0], api.Scene(data)], spp=2, max_depth=2)
    # the caller's current device is restored, and prt_shutdown may be called any number of times
    api.shutdown()
    api.shutdown()
    assert np.array_equal(api.render_multi(reps, spp=6, max_depth=6, seed=3), want)
    ndev = api.device_count()
    if ndev >= 2:
        n = min(ndev, 4)
        spread = [api.Scene(data).upload(d) for d in range(n)]
        for _ in range(2):  # the second frame reuses the cached communicators
            assert np.array_equal(api.render_multi(spread, spp=6, max_depth=6, seed=3), want)
        if n >= 3:
            with pytest.raises(api.PrtError):  # two on device 0, one on device 1: neither all-distinct nor all-same
                api.render_multi([spread[0], reps[0], spread[1]], spp=2, max_depth=2)


def test_render_multi_rccl_branch_with_one_rank_and_injected_failures(gpu, dev_lib, monkeypatch):
    """The RCCL branch of prt_render_multi as far as ONE GPU allows, in the dev-hooks build of the library:
    PRT_TEST_FORCE_RCCL sends a single scene through it — ncclCommInitAll with one rank, the grouped ncclReduce on the device
    framebuffer, stream order, the copy back; PRT_TEST_FAIL_NCCL=init|reduce makes that step report a failure: the call must
    fail with PRT_E_HIP (group closed, failed communicators dropped), leave the scene usable, and the next frames — through
    RCCL again (communicators re-created) and on the plain single-device path — are bit-identical to prt_render's."""
    data = scenes.mixed_materials(72, 56)
    sc = api.Scene(data).upload(gpu)
    assert sc._L.prt_dev_hooks() == 1
    _, want = sc.render(spp=6, max_depth=6, seed=3, f32=True)
    kw = dict(spp=6, max_depth=6, seed=3)
    monkeypatch.setenv("PRT_TEST_FORCE_RCCL", "1")
    for _ in range(2):  # the second frame reuses the cached communicator
        assert np.array_equal(api.render_multi([sc], **kw), want)
    for step in ("reduce", "init"):
        if step == "init":
            sc._L.prt_shutdown()  # no cached communicator: the next call has to create one
        monkeypatch.setenv("PRT_TEST_FAIL_NCCL", step)
        with pytest.raises(api.PrtError) as e:
            api.render_multi([sc], **kw)
        assert e.value.code == _abi.PRT_E_HIP and ("ncclReduce failed" if step == "reduce" else "ncclCommInitAll failed") in str(e.value)
        monkeypatch.delenv("PRT_TEST_FAIL_NCCL")
        assert np.array_equal(api.render_multi([sc], **kw), want)          # RCCL branch again, fresh communicator
        assert np.array_equal(sc.render(f32=True, **kw)[1], want)          # and the scene's own render
    monkeypatch.delenv("PRT_TEST_FORCE_RCCL")
    assert np.array_equal(api.render_multi([sc], **kw), want)              # n == 1 without the hook: no collective
    sc._L.prt_shutdown()
    sc.close()


def test_shipped_library_reads_no_environment_hooks(gpu, monkeypatch):
    """the PRT_TUNE_* / PRT_TEST_* hooks are compiled into libprt_hip_dev.so only.  With the variables set,
    the shipped library uploads (no injected failure), keeps its LDS tables and its packed records, and renders the same
    frame as without them."""
    data = scenes.tiny_scene()
    ref = api.Scene(data).upload(gpu)
    assert ref._L.prt_dev_hooks() == 0
    want = ref.render(spp=3, max_depth=5, seed=2)
    for k, v in (("PRT_TEST_FAIL_UPLOAD", "0"), ("PRT_TUNE_TRI_STRIDE", "128"), ("PRT_TUNE_NO_LDS", "1"), ("PRT_TUNE_KEEP", "63"),
                 ("PRT_TUNE_CACHED_MIN", "0"), ("PRT_TUNE_SCRAMBLE", "2"), ("PRT_TEST_FAIL_NCCL", "init"), ("PRT_TEST_FORCE_RCCL", "1")):
        monkeypatch.setenv(k, v)
    sc = api.Scene(data).upload(gpu)
    assert sc.bvh_info()["tri_stride"] == 96
    assert np.array_equal(sc.render(spp=3, max_depth=5, seed=2), want)
    assert np.array_equal(api.render_multi([sc], spp=3, max_depth=5, seed=2), want.astype(np.float32))


def test_camera_xml_override(gpu, tmp_path):
    exe = build.build_host_example()
    data = scenes.tiny_scene()
    dump = str(tmp_path / "scene.bin")
    scenes.dump_scene(data, dump)
    xml = tmp_path / "cam.xml"
    xml.write_text('<?xml version="1.0"?>\n<camera type="perspective" width="48" height="24" fovy="50.5">\n'
                   '  <eye x="0.1" y="0.2" z="3.0"/>\n  <lookat x="0" y="-0.1" z="0"/>\n  <up x="0" y="1" z="0"/>\n</camera>\n'
                   '<light mtlname="Light" radiance="17,12,4"/>\n')
    out = str(tmp_path / "o.f64")
    r = subprocess.run([exe, dump, "2", "4", out, str(tmp_path / "o.png"), str(xml)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr + r.stdout
    img = np.fromfile(out, dtype=np.float64).reshape(24, 48, 3)
    cam = scenes.Camera(48, 24, 50.5, (0.1, 0.2, 3.0), (0, -0.1, 0))
    ref = api.Scene(data).upload(gpu).render(camera=cam, spp=2, max_depth=4, seed=1)
    assert np.array_equal(img, ref)


def test_device_tile_mapping_matches_host_owner_map(gpu):
    data = scenes.tiny_scene()
    cam = scenes.Camera(100, 72, 40.0, (0.01, 0.02, 3.4), (0, 0, 0))
    sc = api.Scene(data).upload(gpu)
    full = sc.render(camera=cam, spp=1, max_depth=0, background=(1.0, 1.0, 1.0), sample_lights=False)
    for tile, n in [(16, 3), (32, 4), (8, 5)]:
        owner = distributed.tile_owner_map(100, 72, tile, n)
        for r in range(n):
            part = sc.render(camera=cam, spp=1, max_depth=0, background=(1.0, 1.0, 1.0), sample_lights=False,
                             tile_size=tile, rank=r, nranks=n)
            mine = owner == r
            assert np.array_equal(part[mine], full[mine])      # owned pixels: the single-rank values
            assert (part[~mine] == 0).all()                     # everything else stays zero


@pytest.mark.parametrize("scene_fn", [
    lambda: scenes.cornell_box(ball_subdiv=2, width=48, height=40),
    lambda: scenes.veach_mis(64, 36, light_subdiv=1, plate_cells=2),
    lambda: scenes.bathroom(64, 36, detail=0.1),
])
def test_obj_mtl_xml_loader_main_flow(gpu, tmp_path, scene_fn):
    """Scene-loader row: export a stand-in scene as OBJ/MTL/XML(+PPM) in the
    reference's directory layout, run the main.cpp-style driver (Model -> BVHNode -> Camera::Render)
    and require the framebuffer the binding produces for the same (loader-normalised) scene, bit for bit."""
    build.build_host_example()
    data = scene_fn()
    res = str(tmp_path / "res")
    scenes.export_obj(data, res)
    out = str(tmp_path / "o.f64")
    r = subprocess.run([build.MAIN_EXE, res, data.name, "4", "6", str(tmp_path), out], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr + r.stdout
    cam = data.camera
    img = np.fromfile(out, dtype=np.float64).reshape(cam.height, cam.width, 3)
    fixed = scenes.apply_loader_uv_fixup(data)
    ref = api.Scene(fixed).upload(gpu).render(spp=4, max_depth=6, seed=1)
    assert np.array_equal(img, ref)
    assert any(f.endswith(".png") for f in os.listdir(tmp_path)) and any(f.endswith(".hdr") for f in os.listdir(tmp_path))
    # ... and the ORACLE's frame for the loader-normalised scene (the row was tested against itself only)
    cpu, _ = oracle.Oracle(fixed).render(spp=4, max_depth=6, seed=1)
    bad = (np.abs(img - cpu) > 1e-9 * np.maximum(1.0, np.abs(cpu))).any(-1)
    assert bad.mean() <= 1e-4, f"{bad.sum()} pixels differ from the oracle"


def test_loader_png_textures_match_ppm(gpu, tmp_path):
    """Texture ingestion row: the same scene exported with PNG textures and with raw PPM textures must
    render identically through the C++ loader (PNG decoded by the host library's own inflate)."""
    build.build_host_example()
    data = scenes.bathroom(48, 28, detail=0.08)
    outs = []
    for fmt in ("ppm", "png"):
        res = str(tmp_path / fmt)
        scenes.export_obj(data, res, texture_format=fmt)
        out = str(tmp_path / f"{fmt}.f64")
        r = subprocess.run([build.MAIN_EXE, res, data.name, "3", "5", str(tmp_path), out], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr + r.stdout
        assert "Failed" not in r.stderr
        outs.append(np.fromfile(out, dtype=np.float64))
    assert np.array_equal(outs[0], outs[1])


def test_camera_render_honours_its_lights_argument(gpu, tmp_path):
    """Camera::Render(world, lights) samples the list it is GIVEN (Source/Camera.cpp:137-139).  The host API expresses a
    caller's list as world-mesh indices (PrtSceneDesc.light_meshes); r1 replaced a list that was not main.cpp's with a
    warning.  A subset, a reordering and main.cpp's own list must each equal the binding and the oracle with the same
    list; a list holding geometry that is not in world is refused."""
    build.build_host_example()
    exe = str(tmp_path / "lights_arg")
    lib_dir = os.path.dirname(build.HOST_LIB)
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-pthread", "-I", os.path.join(os.path.dirname(lib_dir), "include"),
                           os.path.join(os.path.dirname(lib_dir), "tests", "cpp", "lights_arg.cpp"), "-L", lib_dir,
                           "-Wl,-rpath," + lib_dir, "-lpooraytracer_host", "-lprt_hip", "-o", exe])
    data = scenes.veach_mis(64, 36, light_subdiv=1, plate_cells=2)   # five emissive meshes
    res = str(tmp_path / "res")
    scenes.export_obj(data, res)
    fixed = scenes.apply_loader_uv_fixup(data)
    emissive = [i for i, m in enumerate(fixed.mesh_material) if fixed.materials[int(m)].type == 4]
    assert len(emissive) == 5
    cam = data.camera
    frames = {}
    for mode, lm in (("all", None), ("first", emissive[:1]), ("reversed", emissive[::-1])):
        out = str(tmp_path / f"{mode}.f64")
        r = subprocess.run([exe, res, data.name, mode, "6", "5", out], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        img = np.fromfile(out, dtype=np.float64).reshape(cam.height, cam.width, 3)
        import dataclasses
        sd = dataclasses.replace(fixed, light_meshes=lm)
        assert np.array_equal(img, api.Scene(sd).upload(gpu).render(spp=6, max_depth=5, seed=1)), mode
        cpu, _ = oracle.Oracle(sd).render(spp=6, max_depth=5, seed=1)
        bad = (np.abs(img - cpu) > 1e-9 * np.maximum(1.0, np.abs(cpu))).any(-1)
        assert bad.mean() <= 1e-4, (mode, int(bad.sum()))
        frames[mode] = img
    assert not np.array_equal(frames["all"], frames["first"])
    # the reversed list gives the SAME frame: BVHNode's constructor sorts its list along the longest axis
    # (BVH.cpp:12-33), and these five emitters have distinct positions along it — as the oracle agreed above
    assert np.array_equal(frames["all"], frames["reversed"])
    r = subprocess.run([exe, res, data.name, "foreign", "1", "1", str(tmp_path / "x.f64")], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "must be made of whole meshes" in r.stdout


def _bench(args, env_extra, timeout=600):
    import json
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, capture_output=True, text=True, env=env, timeout=timeout)
    raw = [x for x in r.stdout.splitlines() if x.startswith("{")]
    lines = [json.loads(x) for x in raw]
    if raw:
        # the LAST stdout line is the one the driver parses: compact, with the contract's objects
        assert r.stdout.rstrip().splitlines()[-1] == raw[-1] and len(raw[-1]) < 4096, len(raw[-1])
        assert {"metric", "value", "unit", "n_gpus", "ms_per_step", "config", "roofline", "checks_ok"} <= set(lines[-1])
        assert {"bound", "achieved", "peak", "unit", "frac", "traffic"} <= set(lines[-1]["roofline"])
    return r, lines


def test_bench_two_rank_rehearsal_on_one_gpu(gpu):
    """`python bench.py --gpus 2` starts its two ranks itself; with PRT_BENCH_REHEARSAL=1 both share this box's one GPU
    and the reduce runs over gloo (RCCL refuses two ranks on one device) — the whole N>1 path except RCCL: tile split,
    two-stream pipelining off, reduce, max-over-ranks timing, the assembled-image check, one JSON line from rank 0."""
    r, lines = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--spp", "6", "--no-cpu-baseline"], {"PRT_BENCH_REHEARSAL": "1"})
    assert r.returncode == 0, r.stderr[-3000:]
    assert len(lines) == 2  # the detailed object, then the compact line
    out = lines[-1]
    assert out["n_gpus"] == 2 and out["assembled_matches_single_rank"] is True and out["checks_ok"] is True
    assert out["config"]["parallelism"].startswith("16x16 tiles dealt diagonally over 2 GPU(s)")
    assert out["value"] > 0 and out["roofline"]["kernel"] == "k_render"


def test_bench_two_ranks_over_rccl_when_two_gpus_are_visible(gpu):
    """The real thing (RCCL reduce over xGMI, frames pipelined on two streams, BASELINE config 5 as `config5`) needs two
    GPUs; the one-GPU test box skips it, the driver's multi-GPU node does not."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU visible")
    r, lines = _bench(["--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], {}, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    out = lines[-1]
    assert out["n_gpus"] == 2 and out["assembled_matches_single_rank"] is True
    assert out["config5"]["assembled_matches_single_rank"] is True and out["checks_ok"] is True


def test_bench_parity_check_fails_the_run_when_pixels_differ(gpu, monkeypatch):
    """bench.py compares the rows its CPU baseline rendered with a GPU frame of the same configuration and must exit
    non-zero on a mismatch: PRT_BENCH_FAULT=1 makes it compare against a deliberately different seed."""
    r, lines = _bench(["--steps", "1", "--warmup", "0", "--spp", "4", "--no-extra"], {"PRT_BENCH_FAULT": "1"})
    assert r.returncode != 0 and lines and lines[-1]["parity_check"]["ok"] is False and lines[-1]["checks_ok"] is False
    r, lines = _bench(["--steps", "1", "--warmup", "0", "--spp", "4", "--no-extra"], {})
    pc = lines[-1]["parity_check"]
    cb = lines[-1]["cpu_baseline"]
    # the CPU baseline runs on the fastest worker count of a scan up to the whole affinity mask, and says which
    assert str(cb["cores"]) in {str(k) for k in cb["thread_scan_mpaths_per_s"]} and cb["affinity_cores"] == len(os.sched_getaffinity(0))
    assert lines[-1]["config"]["ray_definition"]
    # a knife-edge branch may flip in a handful of the ~10^6 pixels (tolerance: 0.1 % of them); everything else is within 1e-9
    assert r.returncode == 0 and pc["ok"] is True and pc["bad_px"] <= 1e-4 * pc["pixels"]
