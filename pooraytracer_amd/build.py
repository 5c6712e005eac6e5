"""In-tree build of libprt_hip.so (hipcc, gfx950 only).  hipcc cross-compiles without a GPU.

Every source is compiled to its own object (in parallel, only when it or a header changed) and the objects are linked into
  libprt_hip.so      the product: reads NO developer environment variable (PRT_DEV_HOOKS 0);
  libprt_hip_dev.so  the same objects except prt_api.o, which is compiled with -DPRT_DEV_HOOKS=1: the PRT_TUNE_* / PRT_TEST_*
                     / PRT_VALIDATE_BVH hooks exist only there (failure-injection tests, sweep tools: api.dev_hooks()).
"""
import concurrent.futures
import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libprt_hip.so")
DEV_LIB = os.path.join(HERE, "libprt_hip_dev.so")
OBJ = os.path.join(HERE, "_obj")
SOURCES = ["prt_api.cpp", "prt_kernels.hip", "prt_kernels_f32.hip", "bvh_build.cpp", "bvh_build_gpu.hip", "ray_sort.hip", "scene_setup.cpp"]
HEADERS = ["prt_types.h", "prt_device.h", "prt_host.h", os.path.join("..", "..", "include", "prt.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
CFLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Wall", "-Wno-unused-function", "-fno-gpu-rdc"]
DEV_FLAGS = ["-DPRT_DEV_HOOKS=1"]
JOBS = max(1, min(6, (os.cpu_count() or 2)))


def _deps(src):
    return [os.path.join(CSRC, src)] + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]


def _obj_path(src, flags):
    tag = hashlib.sha1(" ".join(flags).encode()).hexdigest()[:10] if flags else "std"
    return os.path.join(OBJ, tag, src + ".o")


def _compile(src, flags, force, verbose):
    out = _obj_path(src, flags)
    if not force and os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(d) for d in _deps(src)):
        return out
    os.makedirs(os.path.dirname(out), exist_ok=True)
    cmd = [HIPCC, "-c"] + CFLAGS + list(flags) + ["-o", out, os.path.join(CSRC, src)]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


def _link(objs, out, verbose):
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(o) for o in objs):
        return out
    cmd = [HIPCC, "-shared", "-fPIC", "--offload-arch=gfx950", "-fno-gpu-rdc", "-o", out] + objs + ["-lrccl"]  # RCCL: prt_render_multi's reduce
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


def stale():
    return not os.path.exists(LIB) or not os.path.exists(DEV_LIB) or any(
        os.path.getmtime(d) > min(os.path.getmtime(LIB), os.path.getmtime(DEV_LIB)) for src in SOURCES for d in _deps(src))


def build(force=False, verbose=False, extra_flags=(), out=None):
    """Compile every HIP/C++ source of the hot path into pooraytracer_amd/libprt_hip.so (+ libprt_hip_dev.so).
    With `extra_flags` / `out`: ONE library of that configuration at `out` (A/B builds of the tools)."""
    extra = list(extra_flags)
    jobs = [(src, extra) for src in SOURCES]
    if not extra and out is None:
        if not force and not stale():
            return LIB
        jobs.append(("prt_api.cpp", DEV_FLAGS))
    with concurrent.futures.ThreadPoolExecutor(JOBS) as ex:
        objs = list(ex.map(lambda j: _compile(j[0], j[1], force, verbose), jobs))
    if extra or out is not None:
        return _link(objs[:len(SOURCES)], out or LIB, verbose)
    _link(objs[:len(SOURCES)], LIB, verbose)
    _link([objs[-1]] + objs[1:len(SOURCES)], DEV_LIB, verbose)
    return LIB


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
