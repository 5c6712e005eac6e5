"""In-tree build of libprt_hip.so (hipcc, gfx950 only).  hipcc cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libprt_hip.so")
SOURCES = ["prt_api.cpp", "prt_kernels.hip", "prt_kernels_f32.hip", "bvh_build.cpp", "bvh_build_gpu.hip", "ray_sort.hip", "scene_setup.cpp"]
HEADERS = ["prt_types.h", "prt_device.h", "prt_host.h", os.path.join("..", "..", "include", "prt.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-shared", "-Wall", "-Wno-unused-function",
         "-fgpu-rdc" if False else "-fno-gpu-rdc"]


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, extra_flags=(), out=None):
    """Compile every HIP/C++ source of the hot path into pooraytracer_amd/libprt_hip.so."""
    if out is None and not force and not stale():
        return LIB
    out = out or LIB
    cmd = [HIPCC] + FLAGS + list(extra_flags) + ["-o", out] + [os.path.join(CSRC, f) for f in SOURCES] + ["-lrccl"]  # RCCL: prt_render_multi's reduce
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


HOST_LIB = os.path.join(HERE, "libpooraytracer_host.so")
HOST_EXE = os.path.join(HERE, "render_scene")
MAIN_EXE = os.path.join(HERE, "pooraytracer_main")
ROOT = os.path.dirname(HERE)


def build_host_example(force=False):
    """g++ build of the C++ host API (include/pooraytracer/*.h over the C ABI) and the main.cpp-style driver."""
    srcs = [os.path.join(HERE, "host", f) for f in ("host_api.cpp", "model.cpp", "png_decode.cpp", "jpeg_decode.cpp")]
    exe_src = os.path.join(ROOT, "examples", "render_scene.cpp")
    main_src = os.path.join(ROOT, "examples", "pooraytracer_main.cpp")
    inc = os.path.join(ROOT, "include")
    deps = srcs + [exe_src, main_src, os.path.join(inc, "prt.h")] + [os.path.join(inc, "pooraytracer", f) for f in os.listdir(os.path.join(inc, "pooraytracer"))]
    fresh = all(os.path.exists(x) and os.path.getmtime(x) >= max(os.path.getmtime(d) for d in deps) for x in (HOST_LIB, HOST_EXE, MAIN_EXE))
    if fresh and not force:
        return HOST_EXE
    build()
    common = ["-O2", "-std=c++17", "-Wall", "-pthread", "-I", inc, "-L", HERE, "-Wl,-rpath,$ORIGIN"]
    subprocess.check_call(["g++", "-shared", "-fPIC"] + common + ["-o", HOST_LIB] + srcs + ["-lprt_hip"])
    subprocess.check_call(["g++"] + common + ["-o", HOST_EXE, exe_src, "-lpooraytracer_host", "-lprt_hip"])
    subprocess.check_call(["g++"] + common + ["-o", MAIN_EXE, main_src, "-lpooraytracer_host", "-lprt_hip"])
    return HOST_EXE


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True,
          extra_flags=["-Rpass-analysis=kernel-resource-usage"] if "--resources" in sys.argv else [])
    print(LIB)
