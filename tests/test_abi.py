"""C-ABI surface: the library loads, exports every symbol include/prt.h declares, struct layouts
agree between prt.h, _abi.py and the oracle header, and host-only entry points work without a GPU."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from pooraytracer_amd import _abi, api, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_header_symbols_all_exported(prt_lib):
    hdr = open(os.path.join(ROOT, "include", "prt.h")).read()
    declared = set(re.findall(r"\b(prt_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_abi.EXPORTS), declared ^ set(_abi.EXPORTS)
    for name in declared:
        assert hasattr(prt_lib, name), f"{name} not exported by libprt_hip.so"
    assert prt_lib.prt_abi_version() == _abi.PRT_ABI_VERSION


def test_struct_layouts_match_header(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text(
        '#include <stdio.h>\n#include "prt.h"\n#include "oracle.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n",'
        "sizeof(PrtMaterial),sizeof(PrtTexture),sizeof(PrtSceneDesc),sizeof(PrtCamera),sizeof(PrtRenderParams),"
        "sizeof(PrtRay),sizeof(PrtHit),sizeof(PrtLightSample),sizeof(PrtCounters),sizeof(PrtBvhInfo));"
        'printf("%zu %zu %zu %zu %zu %zu %zu %zu\\n",sizeof(OrcMaterial),sizeof(OrcTexture),sizeof(OrcSceneDesc),'
        "sizeof(OrcCamera),sizeof(OrcRenderParams),sizeof(OrcRay),sizeof(OrcHit),sizeof(OrcLightSample));return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "oracle"), str(src), "-o", str(exe)])
    a, b = subprocess.check_output([str(exe)]).decode().strip().split("\n")
    prt_sizes = [int(x) for x in a.split()]
    orc_sizes = [int(x) for x in b.split()]
    py = [C.sizeof(t) for t in (_abi.PrtMaterial, _abi.PrtTexture, _abi.PrtSceneDesc, _abi.PrtCamera,
                                _abi.PrtRenderParams, _abi.PrtRay, _abi.PrtHit, _abi.PrtLightSample, _abi.PrtCounters,
                                _abi.PrtBvhInfo)]
    assert prt_sizes == py
    assert orc_sizes == py[:8]


def test_scene_create_and_light_order_without_gpu(prt_lib):
    sc = api.Scene(scenes.tiny_scene())
    order = sc.light_order()
    assert sorted(order.tolist()) == [10, 11]  # the two triangles of the ceiling light quad
    cnt = sc.counters()
    assert cnt["bvh_nodes"] >= 1 and cnt["bvh_depth"] <= 30
    info = sc.bvh_info()
    assert info["n_nodes"] == cnt["bvh_nodes"] and info["built_on_device"] == 0 and info["build_ms"] >= 0
    sc.close()
    # PRT_SCENE_DEVICE_BVH defers the build to upload(): create succeeds without a GPU, nothing is built yet
    sc = api.Scene(scenes.tiny_scene(), device_bvh=True)
    assert sc.bvh_info()["n_nodes"] == 0
    with pytest.raises(api.PrtError):
        sc.upload(0)
    sc.close()


def test_update_vertices_without_gpu(prt_lib):
    """prt_scene_update_vertices on a scene that is not uploaded: triangle precompute, light tree and the host
    BVH follow the new positions (an uploaded scene rebuilds on the GPU instead; tests/test_gpu_parity.py)."""
    data = scenes.tiny_scene()
    sc = api.Scene(data)
    before = sc.light_order().tolist()
    moved = data.vertices * np.array([2.0, 1.0, 0.5]) + np.array([0.1, -0.2, 0.3])
    sc.update_vertices(moved)
    assert sorted(sc.light_order().tolist()) == sorted(before)
    assert sc.bvh_info()["n_nodes"] >= 1 and sc.bvh_info()["built_on_device"] == 0
    with pytest.raises(AssertionError):
        sc.update_vertices(moved[:-1])
    sc.close()


def test_compute_fails_loudly_without_upload(prt_lib):
    sc = api.Scene(scenes.tiny_scene())
    rays = scenes.random_rays(4, (-1, -1, -1), (1, 1, 1))
    with pytest.raises(api.PrtError) as e:
        sc.trace_closest(rays)
    assert e.value.code == _abi.PRT_E_NO_DEVICE
    with pytest.raises(api.PrtError):
        sc.render(spp=1)
    # the round-3 entry points refuse the same way: no device, no CPU path behind them
    with pytest.raises(api.PrtError) as e:
        sc.render_samples([[1, 1]], spp=2)
    assert e.value.code == _abi.PRT_E_NO_DEVICE
    with pytest.raises(api.PrtError) as e:
        api.render_multi([sc, api.Scene(scenes.tiny_scene())], spp=1)
    assert e.value.code == _abi.PRT_E_NO_DEVICE


def test_scene_create_rejects_bad_input(prt_lib):
    data = scenes.tiny_scene()
    data.mesh_material = data.mesh_material.copy()
    data.mesh_material[0] = 99
    with pytest.raises(api.PrtError) as e:
        api.Scene(data)
    assert e.value.code == _abi.PRT_E_INVALID


def test_scene_create_rejects_non_finite_vertices(prt_lib):
    for bad in (np.nan, np.inf, -np.inf, 1e300):
        data = scenes.tiny_scene()
        data.vertices = data.vertices.copy()
        data.vertices[3, 1, 2] = bad
        with pytest.raises(api.PrtError) as e:
            api.Scene(data)
        assert e.value.code == _abi.PRT_E_INVALID
    sc = api.Scene(scenes.tiny_scene())
    v = scenes.tiny_scene().vertices.copy()
    v[0, 0, 0] = np.nan
    with pytest.raises(api.PrtError):
        sc.update_vertices(v)
    sc.close()


def test_product_does_not_import_oracle():
    """The product package and C sources must never reference the oracle."""
    pkg = os.path.join(ROOT, "pooraytracer_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert "import oracle" not in txt and "from oracle" not in txt and "oracle.h" not in txt, f


def test_cpp_driver_fails_loudly_without_gpu(tmp_path):
    """The C++ host path has no CPU fallback either: on a machine without a HIP device the
    main.cpp-style driver must exit non-zero with the library's error, after loading the scene."""
    from pooraytracer_amd import build
    build.build_host_example()
    if api.device_count() > 0:
        pytest.skip("a GPU is present")
    data = scenes.tiny_scene()
    res = str(tmp_path / "res")
    scenes.export_obj(data, res)
    r = subprocess.run([build.MAIN_EXE, res, data.name, "1", "1", str(tmp_path)], capture_output=True, text=True, timeout=120)
    assert r.returncode != 0
    assert "no HIP device" in r.stderr or "prt_scene_upload" in r.stderr


def test_cmake_build_produces_the_same_libraries(tmp_path):
    """north_star: "host code stays C++ (CMake)" (reference CMakeLists.txt:13-35).  CMakeLists.txt configures without
    a GPU (gfx950 cross-compile) and builds libprt_hip, the C++ host library, both drivers and the oracle; the
    resulting libprt_hip.so exports every symbol include/prt.h declares and reports the header's ABI version."""
    import ctypes
    import shutil
    import subprocess
    if not (shutil.which("cmake") and shutil.which("ninja")):
        pytest.skip("cmake / ninja not installed")
    bdir = str(tmp_path / "build")
    subprocess.check_call(["cmake", "-S", ROOT, "-B", bdir, "-G", "Ninja"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["cmake", "--build", bdir, "-j", "6"], stdout=subprocess.DEVNULL)
    for f in ("libprt_hip.so", "libpooraytracer_host.so", "libprt_oracle.so", "render_scene", "pooraytracer_main"):
        assert os.path.exists(os.path.join(bdir, f)), f
    nm = subprocess.run(["nm", "-D", "--defined-only", os.path.join(bdir, "libprt_hip.so")], capture_output=True, text=True).stdout
    exported = {line.split()[-1] for line in nm.splitlines() if line.strip()}
    assert not [s for s in _abi.EXPORTS if s not in exported]
    # the version is read in a child process: this one may already hold the in-tree library (and torch's HIP runtime)
    code = f"import ctypes; print(ctypes.CDLL({os.path.join(bdir, 'libprt_hip.so')!r}).prt_abi_version())"
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and int(out.stdout.strip()) == _abi.PRT_ABI_VERSION, out.stderr
    ldd = subprocess.run(["ldd", os.path.join(bdir, "render_scene")], capture_output=True, text=True).stdout
    assert "libpooraytracer_host.so" in ldd and "libprt_hip.so" in ldd


@pytest.mark.parametrize("flags", [["-DPRT_K3_PROFILE=1"], ["-DPRT_DEV_HOOKS=1"]])
def test_alternative_build_configurations_still_compile(flags):
    """Two build options of the product sources remain: the per-section cycle stamps of K3's wave loop (instrumentation for
    tools/exp_profile.py) and the developer / test environment hooks (libprt_hip_dev.so; exercised for RESULTS by the -m gpu
    tests that take the `dev_lib` fixture).  The rejected A/B experiments of rounds 1-3 live in experiments/, not behind
    switches.  Semantic analysis incl. every kernel instantiation, host and gfx950 passes; no code generation."""
    import shutil
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not installed")
    csrc = os.path.join(ROOT, "pooraytracer_amd", "csrc")
    srcs = [os.path.join(csrc, f) for f in ("prt_kernels.hip", "prt_kernels_f32.hip", "bvh_build_gpu.hip", "ray_sort.hip", "bvh_build.cpp", "prt_api.cpp")]
    r = subprocess.run([hipcc, "-std=c++17", "--offload-arch=gfx950", "-fsyntax-only", "-Wno-unused-function"] + flags + srcs,
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
