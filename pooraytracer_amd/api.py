"""Python binding of libprt_hip.so (ctypes over the C ABI in include/prt.h).

This is test / benchmark plumbing: numpy (or torch device pointers) in, numpy out.  All compute
happens in the HIP kernels; there is no Python or CPU fallback — if the library or a GPU is missing
the calls raise PrtError.
"""
import ctypes as C
import os

import numpy as np

from . import _abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_DEV_LIB_PATH = os.path.join(_HERE, "libprt_hip_dev.so")
# PRT_LIB=<path>: another build of the library (A/B tools); PRT_DEV_LIB=1: the dev-hooks build (sweep tools that set PRT_TUNE_*)
_LIB_PATH = os.environ.get("PRT_LIB") or (_DEV_LIB_PATH if os.environ.get("PRT_DEV_LIB") == "1" else os.path.join(_HERE, "libprt_hip.so"))
_libs = {}


class PrtError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libprt_hip error {code}: {msg}")
        self.code = code


def lib_path():
    return _LIB_PATH


def load():
    """dlopen the in-tree libprt_hip.so (build it first with pooraytracer_amd.build.build())."""
    if _LIB_PATH in _libs:
        return _libs[_LIB_PATH]
    if not os.path.exists(_LIB_PATH):
        raise PrtError(-100, f"{_LIB_PATH} not built; run `python -m pooraytracer_amd.build` (needs hipcc)")
    # PyTorch bundles its own libamdhip64.so.7 / libhsa-runtime64.so.1 (same SONAMEs as /opt/rocm).
    # Two HIP runtimes cannot coexist in one process, and torch fails to initialise on the system
    # one, so when torch is importable it is imported first and this library binds to its runtime.
    if not os.environ.get("PRT_NO_TORCH_PRELOAD"):
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(_LIB_PATH)
    vp, sz, i32, u64 = C.c_void_p, C.c_size_t, C.c_int, C.c_uint64
    L.prt_abi_version.restype = C.c_int
    L.prt_last_error.restype = C.c_char_p
    L.prt_device_count.argtypes = [C.POINTER(C.c_int)]
    L.prt_scene_create.argtypes = [vp, C.POINTER(vp)]
    L.prt_scene_destroy.argtypes = [vp]
    L.prt_scene_destroy.restype = None
    L.prt_scene_upload.argtypes = [vp, i32]
    L.prt_scene_bvh_info.argtypes = [vp, vp]
    L.prt_scene_update_vertices.argtypes = [vp, vp, vp]
    L.prt_scene_light_count.argtypes = [vp, C.POINTER(u64)]
    L.prt_scene_light_order.argtypes = [vp, vp, u64]
    L.prt_trace_closest.argtypes = [vp, vp, sz, vp, i32]
    L.prt_trace_closest_device.argtypes = [vp, vp, sz, vp, i32, vp]
    L.prt_trace_closest_device_prec.argtypes = [vp, vp, sz, vp, i32, i32, vp]
    L.prt_trace_closest_sorted_device.argtypes = [vp, vp, sz, vp, i32, i32, vp]
    L.prt_sample_lights.argtypes = [vp, vp, sz, u64, vp]
    L.prt_render.argtypes = [vp, vp, vp, vp, vp]
    L.prt_render_device.argtypes = [vp, vp, vp, vp, vp, i32, vp]
    L.prt_get_counters.argtypes = [vp, vp]
    L.prt_tonemap_srgb8.argtypes = [vp, vp, i32, i32, vp, vp]
    L.prt_material_eval.argtypes = [vp, i32, sz, vp, vp, vp, u64, vp]
    L.prt_material_scatter.argtypes = [vp, i32, sz, vp, vp, vp, vp, u64, vp, vp, vp]
    L.prt_texture_value.argtypes = [vp, i32, sz, vp, vp]
    L.prt_render_samples.argtypes = [vp, vp, vp, vp, sz, i32, i32, vp, vp]
    L.prt_render_multi.argtypes = [vp, i32, vp, vp, vp]
    if L.prt_abi_version() != _abi.PRT_ABI_VERSION and os.environ.get("PRT_ABI_ANY") != "1":  # (PRT_ABI_ANY: A/B tools timing an older build)
        raise PrtError(-101, "ABI version mismatch between _abi.py and libprt_hip.so")
    try:
        L.prt_shutdown.restype = None
        L.prt_dev_hooks.restype = C.c_int
    except AttributeError:
        if os.environ.get("PRT_ABI_ANY") != "1":
            raise
    _libs[_LIB_PATH] = L
    return L


class dev_hooks:
    """`with api.dev_hooks():` — inside, scenes are created by libprt_hip_dev.so, the build of the same sources that reads the
    PRT_TUNE_* / PRT_TEST_* environment hooks (the shipped libprt_hip.so reads none).  A Scene keeps the library that made it."""

    def __enter__(self):
        global _LIB_PATH
        self._saved = _LIB_PATH
        _LIB_PATH = _DEV_LIB_PATH
        L = load()
        assert L.prt_dev_hooks() == 1
        return L

    def __exit__(self, *exc):
        global _LIB_PATH
        _LIB_PATH = self._saved
        return False


def shutdown():
    """prt_shutdown of every library loaded so far (cached RCCL communicators)."""
    for L in _libs.values():
        L.prt_shutdown()


def _check(rc, L=None):
    if rc != 0:
        raise PrtError(rc, (L or load()).prt_last_error().decode("utf-8", "replace"))


def _f64(a, k):
    return np.ascontiguousarray(a, dtype=np.float64).reshape(-1, k)


def device_count():
    n = C.c_int(0)
    rc = load().prt_device_count(C.byref(n))
    return n.value if rc == 0 else 0


class Scene:
    """A scene handle: host-side preparation at construction, `upload(device)` before any compute."""

    def __init__(self, scene_data, device_bvh=False):
        self.data = scene_data
        L = self._L = load()
        desc, keep = _abi.marshal_scene(scene_data)
        if device_bvh:
            desc.flags = _abi.PRT_SCENE_DEVICE_BVH  # tree built on the GPU in upload()
        h = C.c_void_p()
        _check(L.prt_scene_create(C.byref(desc), C.byref(h)), L)
        del keep  # the library copies everything it needs during create
        self._h = h
        self.device = None

    def close(self):
        if getattr(self, "_h", None):
            self._L.prt_scene_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def upload(self, device=0):
        _check(self._L.prt_scene_upload(self._h, device), self._L)
        self.device = device
        return self

    def update_vertices(self, vertices, normals=None):
        """New positions for the same triangles; an uploaded scene rebuilds its BVH on the GPU."""
        v = np.ascontiguousarray(vertices, dtype=np.float64)
        assert v.shape == self.data.vertices.shape
        n = None if normals is None else np.ascontiguousarray(normals, dtype=np.float64)
        _check(self._L.prt_scene_update_vertices(self._h, v.ctypes.data, None if n is None else n.ctypes.data), self._L)
        return self

    def bvh_info(self):
        b = _abi.PrtBvhInfo()
        _check(self._L.prt_scene_bvh_info(self._h, C.byref(b)), self._L)
        return {f: getattr(b, f) for f, _ in _abi.PrtBvhInfo._fields_}

    def light_order(self):
        n = C.c_uint64(0)
        _check(self._L.prt_scene_light_count(self._h, C.byref(n)), self._L)
        out = np.zeros(n.value, dtype=np.int32)
        _check(self._L.prt_scene_light_order(self._h, out.ctypes.data, n.value), self._L)
        return out

    def trace_closest(self, rays, count_work=False):
        rays = np.ascontiguousarray(rays, dtype=_abi.RAY_DTYPE)
        hits = np.zeros(rays.shape[0], dtype=_abi.HIT_DTYPE)
        _check(self._L.prt_trace_closest(self._h, rays.ctypes.data, rays.shape[0], hits.ctypes.data, int(count_work)), self._L)
        return hits

    def trace_closest_device(self, d_rays_ptr, n, d_hits_ptr, count_work=False, stream=None, precision=0, sort=False):
        """K1 on device buffers; precision = _abi.PRECISION_F64 (default) or PRECISION_F32 (fp32 fast mode).  sort=True: K4
        first — the batch is traced in a locality order (same hits, for scenes that do not fit the caches)."""
        fn = self._L.prt_trace_closest_sorted_device if sort else self._L.prt_trace_closest_device_prec
        _check(fn(self._h, d_rays_ptr, n, d_hits_ptr, int(count_work), int(precision), stream))

    def sample_lights(self, origins, seed=1):
        origins = np.ascontiguousarray(origins, dtype=np.float64).reshape(-1, 3)
        out = np.zeros(origins.shape[0], dtype=_abi.LIGHT_SAMPLE_DTYPE)
        _check(self._L.prt_sample_lights(self._h, origins.ctypes.data, origins.shape[0], seed, out.ctypes.data), self._L)
        return out

    # ---- test hooks for the material arithmetic (include/prt.h)
    def material_eval(self, material, wi, wo, uv=None, seed=1):
        wi, wo = _f64(wi, 3), _f64(wo, 3)
        uv = None if uv is None else _f64(uv, 2)
        out = np.zeros_like(wi)
        _check(self._L.prt_material_eval(self._h, material, wi.shape[0], wi.ctypes.data, wo.ctypes.data,
                                        None if uv is None else uv.ctypes.data, seed, out.ctypes.data), self._L)
        return out

    def material_scatter(self, material, rd, normal=(0, 0, 1), tangent=(1, 0, 0), uv=None, seed=1):
        rd = _f64(rd, 3)
        uv = None if uv is None else _f64(uv, 2)
        nrm, tan = _f64(normal, 3), _f64(tangent, 3)
        wi, att, ok = np.zeros_like(rd), np.zeros_like(rd), np.zeros(rd.shape[0], dtype=np.int32)
        _check(self._L.prt_material_scatter(self._h, material, rd.shape[0], rd.ctypes.data, nrm.ctypes.data, tan.ctypes.data,
                                           None if uv is None else uv.ctypes.data, seed, wi.ctypes.data, att.ctypes.data,
                                           ok.ctypes.data), self._L)
        return wi, att, ok.astype(bool)

    def texture_value(self, texture, uv):
        uv = _f64(uv, 2)
        out = np.zeros((uv.shape[0], 3))
        _check(self._L.prt_texture_value(self._h, texture, uv.shape[0], uv.ctypes.data, out.ctypes.data), self._L)
        return out

    def render(self, camera=None, f32=False, **kw):
        """Render one frame to host memory.  Returns (H,W,3) float64 (and float32 if f32=True)."""
        cam = camera or self.data.camera
        c, p = _abi.make_camera(cam), _abi.make_params(**kw)
        out64 = np.zeros((cam.height, cam.width, 3), dtype=np.float64)
        out32 = np.zeros((cam.height, cam.width, 3), dtype=np.float32) if f32 else None
        _check(self._L.prt_render(self._h, C.byref(c), C.byref(p), out64.ctypes.data,
                                 out32.ctypes.data if f32 else None), self._L)
        return (out64, out32) if f32 else out64

    def render_samples(self, pixels_xy, camera=None, sample_begin=0, sample_count=None, trace=False, **kw):
        """RayColor of single camera samples through K3 (test hook, include/prt.h): (n_pixels, count, 3) float64 and, with
        trace=True, the paths' signatures (n_pixels, count, TRACE_WORDS) int32.  kw as for render(); spp = the default count."""
        cam = camera or self.data.camera
        c, p = _abi.make_camera(cam), _abi.make_params(**kw)
        count = p.spp if sample_count is None else int(sample_count)
        px = np.ascontiguousarray(pixels_xy, dtype=np.int32).reshape(-1, 2)
        out = np.zeros((px.shape[0], count, 3), dtype=np.float64)
        tr = np.zeros((px.shape[0], count, _abi.TRACE_WORDS), dtype=np.int32) if trace else None
        _check(self._L.prt_render_samples(self._h, C.byref(c), C.byref(p), px.ctypes.data, px.shape[0], int(sample_begin), count,
                                         out.ctypes.data, tr.ctypes.data if trace else None), self._L)
        return (out, tr) if trace else out

    def render_device(self, d_f64_ptr, d_f32_ptr, camera=None, count_work=False, stream=None, **kw):
        """Asynchronous render into device buffers (raw device pointers, e.g. torch tensor.data_ptr())."""
        cam = camera or self.data.camera
        c, p = _abi.make_camera(cam), _abi.make_params(**kw)
        _check(self._L.prt_render_device(self._h, C.byref(c), C.byref(p), d_f64_ptr, d_f32_ptr, int(count_work), stream), self._L)

    def tonemap_srgb8(self, d_f32_ptr, width, height, d_u8_ptr, stream=None):
        _check(self._L.prt_tonemap_srgb8(self._h, d_f32_ptr, width, height, d_u8_ptr, stream), self._L)

    def counters(self):
        c = _abi.PrtCounters()
        _check(self._L.prt_get_counters(self._h, C.byref(c)), self._L)
        return {f: getattr(c, f) for f, _ in _abi.PrtCounters._fields_}


def render_multi(scene_list, camera=None, **kw):
    """prt_render_multi: one frame over several uploaded replicas of a scene (different GPUs: tiles + one RCCL reduce of the
    fp32 framebuffer; one GPU: tile shares summed on it).  Returns (H, W, 3) float32."""
    cam = camera or scene_list[0].data.camera
    c, p = _abi.make_camera(cam), _abi.make_params(**kw)
    hs = (C.c_void_p * len(scene_list))(*[s._h for s in scene_list])
    out = np.zeros((cam.height, cam.width, 3), dtype=np.float32)
    L = scene_list[0]._L
    _check(L.prt_render_multi(hs, len(scene_list), C.byref(c), C.byref(p), out.ctypes.data), L)
    return out
