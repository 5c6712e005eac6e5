"""In-tree build of libprt_hip.so (hipcc, gfx950 only).  hipcc cross-compiles without a GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libprt_hip.so")
SOURCES = ["prt_api.cpp", "prt_kernels.hip", "bvh_build.cpp", "scene_setup.cpp"]
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
    cmd = [HIPCC] + FLAGS + list(extra_flags) + ["-o", out] + [os.path.join(CSRC, f) for f in SOURCES]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True,
          extra_flags=["-Rpass-analysis=kernel-resource-usage"] if "--resources" in sys.argv else [])
    print(LIB)
